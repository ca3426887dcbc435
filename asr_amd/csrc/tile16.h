// Small-M matrix products of the recurrent steps on the fp32 matrix cores.
//
// Every per-step product of the path has M = batch (32) rows: the LSTM
// recurrences h_{t-1} W_hh^T (torch.nn.LSTM, Models/encoder.py:33), and the
// decoder's L_se / L_ss / L_gs products (Models/attention.py:30,
// Models/decoder.py:57) with their transposes in the backward.  One workgroup
// computes ONE 16x16 output tile
//
//     D[i][n] = sum over segments  sum_k X_seg[i][k] * W_seg[n][k]
//
// with v_mfma_f32_16x16x4_f32 (exact fp32).  K is split over the workgroup's
// waves and over the MFMA's four k-lanes: lane (i = lane&15, q = lane>>4) of
// wave w owns the contiguous k-range of slot w*4+q in both its A row and its
// B row, so both operands are read straight from global/L2 with 16-byte loads
// and no LDS staging (the operands are used once per workgroup).  The waves'
// partial tiles are summed through LDS.
#pragma once
#include "common.h"

#define T16_MAXW 16

struct Tile16Acc {
    f32x4 a0, a1;
};

__device__ __forceinline__ void tile16_init(Tile16Acc &acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc.a0[r] = 0.f; acc.a1[r] = 0.f; }
}

// fast: K % 4 == 0 and 16-byte aligned rows -> unconditional masked load (no branch, so
// all loads of an unrolled loop are in flight together); `safe` is any valid aligned address.
__device__ __forceinline__ float4 t16_load_fast(const float *row, int k, int ke, const float *safe) {
    const bool ok = row != nullptr && k < ke;
    const float4 q = ld4(ok ? row + k : safe);
    const unsigned m = ok ? 0xffffffffu : 0u;
    return make_float4(__uint_as_float(__float_as_uint(q.x) & m), __uint_as_float(__float_as_uint(q.y) & m),
                       __uint_as_float(__float_as_uint(q.z) & m), __uint_as_float(__float_as_uint(q.w) & m));
}

__device__ __forceinline__ float4 t16_load(const float *row, int k, int ke, bool vec) {
    float4 v = zero4();
    if (row != nullptr && k < ke) {
        if (vec && k + 3 < ke) {
            v = ld4(row + k);
        } else {
            v.x = row[k];
            if (k + 1 < ke) v.y = row[k + 1];
            if (k + 2 < ke) v.z = row[k + 2];
            if (k + 3 < ke) v.w = row[k + 3];
        }
    }
    return v;
}

// xrow: this lane's A row (X[i][:], i = lane&15) or nullptr for a zero row;
// wrow: this lane's B row (W[n][:], n = lane&15) or nullptr.
// nparts > 1: the A row is the sum of `nparts` rows `pstride` floats apart
// (partial sums left by an earlier kernel); if `sum_out` is non-null the summed
// A values are also stored there (same indexing as xrow).
// The trip count is uniform over the workgroup: every wave issues the same MFMAs.
// safe: any mapped, 16-byte aligned address (masked lanes load from it); parts: UNIFORM
// flag selecting the partial-sum form (nparts / sum_out may then differ per lane).
__device__ __forceinline__ void tile16_seg(Tile16Acc &acc, const float *xrow, const float *wrow, int K,
                                           bool vec, const float *safe, bool parts = false, int nparts = 1,
                                           int64_t pstride = 0, float *sum_out = nullptr, int nparts_max = 1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int slots = nw * 4;
    int kq = (K + slots - 1) / slots;
    kq = (kq + 3) & ~3;
    const int kb = (wave * 4 + (lane >> 4)) * kq;
    const int ke = min(K, kb + kq);
    if (vec && (K & 3) == 0 && !parts) {   // uniform
#pragma unroll 4
        for (int u = 0; u < kq; u += 4) {
            const int k = kb + u;
            const float4 a = t16_load_fast(xrow, k, ke, safe);
            const float4 b = t16_load_fast(wrow, k, ke, safe);
            acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc.a0, 0, 0, 0);
            acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc.a1, 0, 0, 0);
            acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc.a0, 0, 0, 0);
            acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc.a1, 0, 0, 0);
        }
        return;
    }
    if (vec && (K & 3) == 0 && parts) {   // uniform: branch-free partial-sum form, 8 parts in flight
        for (int u = 0; u < kq; u += 4) {
            const int k = kb + u;
            float4 a = zero4();
            for (int p0 = 0; p0 < nparts_max; p0 += 8) {   // uniform bound; lanes mask p >= nparts
                float4 t[8];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    t[c] = t16_load_fast((xrow != nullptr && p0 + c < nparts) ? xrow + (p0 + c) * pstride : nullptr,
                                         k, ke, safe);
#pragma unroll
                for (int c = 0; c < 8; ++c) { a.x += t[c].x; a.y += t[c].y; a.z += t[c].z; a.w += t[c].w; }
            }
            if (sum_out != nullptr && xrow != nullptr && k < ke) *reinterpret_cast<float4 *>(sum_out + k) = a;
            const float4 b = t16_load_fast(wrow, k, ke, safe);
            acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc.a0, 0, 0, 0);
            acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc.a1, 0, 0, 0);
            acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc.a0, 0, 0, 0);
            acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc.a1, 0, 0, 0);
        }
        return;
    }
#pragma unroll 2
    for (int u = 0; u < kq; u += 4) {
        const int k = kb + u;
        float4 a = t16_load(xrow, k, ke, vec);
        if (nparts > 1 && xrow != nullptr) {
            for (int p = 1; p < nparts; ++p) {
                float4 t = t16_load(xrow + p * pstride, k, ke, vec);
                a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
            }
        }
        if (sum_out != nullptr && xrow != nullptr && k < ke) {
            sum_out[k] = a.x;
            if (k + 1 < ke) sum_out[k + 1] = a.y;
            if (k + 2 < ke) sum_out[k + 2] = a.z;
            if (k + 3 < ke) sum_out[k + 3] = a.w;
        }
        const float4 b = t16_load(wrow, k, ke, vec);
        acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc.a0, 0, 0, 0);
        acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc.a1, 0, 0, 0);
        acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc.a0, 0, 0, 0);
        acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc.a1, 0, 0, 0);
    }
}

// C/D layout of the 16x16 MFMA: col = lane&15 (n), row = (lane>>4)*4 + r (i).
// After this call (it contains a __syncthreads) tile16_get(red, i, n) is the
// full sum for any thread of the workgroup.
__device__ __forceinline__ void tile16_reduce(const Tile16Acc &acc, float (*red)[16][17]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][lane & 15] = acc.a0[r] + acc.a1[r];
    __syncthreads();
}

__device__ __forceinline__ float tile16_get(float (*red)[16][17], int i, int n) {
    const int nw = blockDim.x >> 6;
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += red[w][i][n];
    return s;
}
