// tma_common.h -- shared host-side helpers for libtma_hip.so (error state, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/tma.h"

namespace tma {

char *err_buf();  // thread-local, 512 bytes (defined in tma_env.hip)

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define TMA_HIP(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess) return ::tma::fail(TMA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define TMA_LAUNCH_CHECK()                                                                                     \
    do {                                                                                                       \
        hipError_t _e = hipGetLastError();                                                                     \
        if (_e != hipSuccess) return ::tma::fail(TMA_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// counter-based hash shared by the action tape, the crawler reset and the policy sampler
__host__ __device__ inline uint32_t mix32(uint32_t seed, uint32_t i, uint32_t t) {
    uint32_t x = (seed * 0x9E3779B1u) ^ (i * 0x85EBCA77u) ^ (t * 0xC2B2AE3Du);
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace tma
