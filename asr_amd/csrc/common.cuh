// Shared device/host helpers for libasr_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/asr_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WAVE 64

// ---------------------------------------------------------------- errors --
void asr_set_error(const char *fmt, ...);

#define ASR_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            asr_set_error(__VA_ARGS__);          \
            return ASR_ERR_ARG;                  \
        }                                        \
    } while (0)

#define ASR_CHECK_LAUNCH(name)                                                 \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            asr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return ASR_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

#define ASR_CHECK_HIP(expr)                                                    \
    do {                                                                       \
        hipError_t e__ = (expr);                                               \
        if (e__ != hipSuccess) {                                               \
            asr_set_error("%s: %s", #expr, hipGetErrorString(e__));            \
            return ASR_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

// ------------------------------------------------------------ device math --
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// sigmoid as torch.nn.LSTM computes it (1/(1+exp(-x))), accurate expf
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// tanh with ~1e-7 absolute error: 1 - 2/(exp(2x)+1); saturates cleanly.
__device__ __forceinline__ float tanh_f(float x) {
    float ax = fabsf(x);
    if (ax < 0.04f) {  // odd Taylor series, avoids cancellation near 0
        float x2 = x * x;
        return x * (1.0f + x2 * (-0.33333333f + x2 * (0.13333333f - x2 * 0.053968254f)));
    }
    float e = expf(2.0f * ax);
    float r = 1.0f - 2.0f / (e + 1.0f);
    return copysignf(r, x);
}

__host__ __device__ __forceinline__ int64_t rowoff2(int64_t i, int d, int64_t s0, int64_t s1) {
    return (i / d) * s1 + (i % d) * s0;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
