const char __hip_fatbin_2602246389185c9d[32] __attribute__((aligned(4096))) = "__CLANG_OFFLOAD_BUNDLE__";
const char __hip_fatbin_3349d64916e1a571[32] __attribute__((aligned(4096))) = "__CLANG_OFFLOAD_BUNDLE__";
const char __hip_fatbin_4b879c5600fab75a[32] __attribute__((aligned(4096))) = "__CLANG_OFFLOAD_BUNDLE__";
const char __hip_fatbin_6d7e86ec32dbbc41[32] __attribute__((aligned(4096))) = "__CLANG_OFFLOAD_BUNDLE__";
const char __hip_fatbin_7b81974400f5c69c[32] __attribute__((aligned(4096))) = "__CLANG_OFFLOAD_BUNDLE__";
const char __hip_fatbin_8150a235d8ea9146[32] __attribute__((aligned(4096))) = "__CLANG_OFFLOAD_BUNDLE__";
const char __hip_fatbin_cc18bfca6287000c[32] __attribute__((aligned(4096))) = "__CLANG_OFFLOAD_BUNDLE__";
