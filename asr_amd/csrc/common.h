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

#define ASR_CHECK(expr)                  \
    do {                                 \
        int r__ = (expr);                \
        if (r__ != ASR_OK) return r__;   \
    } while (0)

// ------------------------------------------------------------ device math --
// Sum over the 16 lanes of a DPP row (lanes with equal lane>>4), result in every lane of the row.
// Data-parallel-primitive modifiers on the adds themselves (row rotate by 8 and 4, quad
// permutes): 4 VALU instructions, no trip through the LDS crossbar (__shfl_xor compiles to
// ds_bpermute_b32 + a wait on gfx9).
#define ASR_DPP(v, ctrl) \
    __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float row16_sum(float v) {
    v += ASR_DPP(v, 0x128);   // row_ror:8
    v += ASR_DPP(v, 0x124);   // row_ror:4
    v += ASR_DPP(v, 0x4E);    // quad_perm:[2,3,0,1]
    v += ASR_DPP(v, 0xB1);    // quad_perm:[1,0,3,2]
    return v;
}
// Sums of FOUR values over the 16 lanes of a DPP row in 5 DPP adds + 6 selects (instead of
// 4 x 4): a butterfly that halves the number of live values at each of the first two steps.
// Returns, in lane c of the row, the row total of v[c >> 2] (each total replicated on 4 lanes).
// Pairings: +8 (row_ror:8), mirror within 8 lanes (flips bit 2), then xor 2 and xor 1.
__device__ __forceinline__ float row16_sum4(float v0, float v1, float v2, float v3) {
    const int c = threadIdx.x & 15;
    const bool hi = (c & 8) != 0, mid = (c & 4) != 0;
    const float s0 = hi ? v0 : v2, s1 = hi ? v1 : v3;      // what the partner keeps
    const float k0 = hi ? v2 : v0, k1 = hi ? v3 : v1;      // what this lane keeps
    const float u0 = k0 + ASR_DPP(s0, 0x128);              // row_ror:8
    const float u1 = k1 + ASR_DPP(s1, 0x128);
    const float s = mid ? u0 : u1, k = mid ? u1 : u0;
    float w = k + ASR_DPP(s, 0x141);                       // row_half_mirror
    w += ASR_DPP(w, 0x4E);                                 // quad_perm:[2,3,0,1]
    w += ASR_DPP(w, 0xB1);                                 // quad_perm:[1,0,3,2]
    return w;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exp(x) through the hardware 2^x unit (1 ulp) -- x*log2(e) in fp32.
__device__ __forceinline__ float fast_exp(float x) {
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}

// tanh(x) = 1 - 2 / (1 + e^{2x}): 3 VALU + 2 transcendental instructions, no sign handling;
// saturates cleanly (e -> inf: rcp = 0 -> 1; e -> 0: 1 - 2 = -1), absolute error ~1e-7
// everywhere (the cancellation near 0 costs relative, not absolute, accuracy -- every use here
// feeds a sum or a product with O(1) terms).
__device__ __forceinline__ float tanh_f(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}

// 1 - tanh(x)^2 = 4 e / (1 + e)^2 with e = exp(-2|x|) in (0, 1]: no overflow, no tanh needed
// (3 VALU multiplies + add + 2 transcendental; |x| and the sign are free operand modifiers).
__device__ __forceinline__ float sech2_f(float x) {
    const float e = __builtin_amdgcn_exp2f(fabsf(x) * -2.88539008177792681472f);
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    return 4.0f * (e * r) * r;
}

// sigmoid as torch.nn.LSTM computes it, 1/(1+exp(-x))
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x));   // v_rcp_f32: 1 ulp, no division sequence
}

// sigmoid in the tanh form of Models/decoder.py:269-279
__device__ __forceinline__ float sigmoid_tanh_form(float x) { return 0.5f * tanh_f(0.5f * x) + 0.5f; }

// the counter-based hash behind every dropout mask: element i is kept iff
// asr_mix32(seed + i * golden) >= p * 2^32 (asr_dropout_fwd, misc.hip)
__device__ __forceinline__ uint32_t asr_mix32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return (uint32_t)x;
}
__device__ __forceinline__ bool asr_keep(uint64_t seed, int64_t i, uint32_t thresh) {
    return asr_mix32(seed + (uint64_t)i * 0x9E3779B97F4A7C15ULL) >= thresh;
}

__host__ __device__ __forceinline__ int64_t rowoff2(int64_t i, int d, int64_t s0, int64_t s1) {
    return (i / d) * s1 + (i % d) * s0;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Bound of every in-launch wait (a wave polling for data another workgroup of the same launch publishes): WALL TIME
// on the constant-rate counter s_memrealtime reads (100 MHz on gfx950), not an iteration count.  An iteration
// count measures nothing when the poll's own latency moves (a loaded fabric, a slow clock); what a bound has to
// tell apart is "the producer is late" -- it is not resident yet because another kernel (an RCCL channel) still
// holds its compute unit: milliseconds at most -- from "the producer will never come".  4 s is far beyond the first
// and still ends a launch that cannot make progress.  `spins` counts the caller's polls; the counter is read once
// per 256 of them (first at poll 255, which starts the clock), so the fast path pays nothing.
#define ASR_WAIT_LIMIT_TICKS 400000000ull
__device__ __forceinline__ bool asr_wait_expired(int spins, unsigned long long &t0) {
    if ((spins & 255) != 255) return false;
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (spins == 255) {
        t0 = now;
        return false;
    }
    return now - t0 > ASR_WAIT_LIMIT_TICKS;
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
