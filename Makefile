# Repo-level convenience targets.  The product build is `python -c "import __graft_entry__ as g; g.build()"` (csrc/Makefile + oracle/Makefile).
ROCM_CLANG_RT := $(firstword $(wildcard /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so))

all:
	$(MAKE) -C three-mlagents_amd/csrc -j4
	$(MAKE) -C oracle

# an AddressSanitizer + UBSan build of the host code for the CPU tests -- the C oracle and the HOST half of
# libtma_hip.so (device code is not compiled: GPU ASAN is unavailable on this pool), then `pytest -m "not gpu"` on them.
# Python itself is not instrumented, so the runtime is preloaded and leak checking (CPython never frees everything) is off.
ASAN_TESTS ?= tests
asan-build:
	$(MAKE) -C oracle libtma_oracle_asan.so
	$(MAKE) -C three-mlagents_amd/csrc -j4 libtma_hip_asan.so
asan: asan-build
	TMA_IN_ASAN=1 LD_PRELOAD=$(ROCM_CLANG_RT) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	    TMA_LIB_PATH=$(CURDIR)/three-mlagents_amd/csrc/libtma_hip_asan.so TMA_ORACLE_PATH=$(CURDIR)/oracle/libtma_oracle_asan.so \
	    python -m pytest $(ASAN_TESTS) -x -q -m "not gpu" -p no:cacheprovider

.PHONY: all asan asan-build
