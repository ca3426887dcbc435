// Teacher-forced location-aware attention decoder: the L-step loop of
// Models/decoder.py:52-60 with Models/attention.py:24-40 and the LSTM cell of
// Models/decoder.py:269-279 inside it, and its closed-form backward (BPTT through
// s, c AND the alignment alpha -- Models/decoder.py:53 does not detach it).
//
// What is hoisted out of the loop (host side, asr_sgemm):
//   PH = L_he(hbatch)                       (the reference redoes it every step)
//   U  = tanh(L_gy g + L_sy s), Y = L_yy U  (not on the recurrent chain)
//   all weight gradients, dh = sum_s alpha_s (x) dg_s, dPH W_he.
//
// Per step, forward (4 launches):
//   sproj_s   = S_s W_se^T                                   tile16 MFMA
//   energy    : F = conv(alpha_{s-1}); e = w_ee . tanh(sproj + PH + F W_fe^T)
//   ctx       : alpha = masked softmax(e); g = sum_t alpha h
//   cell      : rec = Emb[y] + S W_ss^T + g W_gs^T + b; LSTM cell   tile16 MFMA
//
// The energy kernels put the FRAME on the lane (64 frames per workgroup, the
// waves split the attention dimension): the sum over a, the conv features and
// their gradient are then lane-local, and the per-a operands (sproj, w_ee,
// W_fe) are wave-uniform LDS broadcasts.  PH tiles are staged through LDS with
// full-line global reads and read back transposed, bank-conflict free.
// HBM/L3 traffic per step: PH once (energy) + hbatch once (ctx) = 2*B*T*A*4 B.
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include "tile16.h"

#define NCH 10     // conv channels  (Models/attention.py:22)
#define KW 100     // conv taps
#define PADW 50    // conv padding
#define TT 64      // frames per energy workgroup (= lanes)
#define EW 16      // waves per energy workgroup (4 per SIMD: the tanh chains are latency-bound)
#define ACH (EW * 16)   // attention-dim chunk staged per pass
#define TILE_FLOATS (EW * TT * 17)   // eight per-wave tiles of 64 x 17 floats

// ------------------------------------------------------------------ sproj --
__global__ __launch_bounds__(256) void sproj_kernel(const float *__restrict__ S_s,
                                                    const float *__restrict__ W_se,
                                                    float *__restrict__ sp, int B, int Hd, int A, int vec) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, i = lane & 15;
    const int n0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const float *xrow = (b0 + i) < B ? S_s + (int64_t)(b0 + i) * Hd : nullptr;
    const float *wrow = (n0 + i) < A ? W_se + (int64_t)(n0 + i) * Hd : nullptr;
    Tile16Acc acc;
    tile16_init(acc);
    tile16_seg(acc, xrow, wrow, Hd, vec != 0, W_se);
    tile16_reduce(acc, red);
    const int bi = threadIdx.x >> 4, ni = threadIdx.x & 15;
    if (b0 + bi < B && n0 + ni < A) sp[(int64_t)(b0 + bi) * A + n0 + ni] = tile16_get(red, bi, ni);
}

// ---------------------------------------------------- shared energy pieces --
// uni[a][12] = {sproj[a], w_ee[a], W_fe[a][0..9]}  (wave-uniform operands)
__device__ __forceinline__ void fill_uni(float *uni, const float *__restrict__ sp_b,
                                         const float *__restrict__ w_ee,
                                         const float *__restrict__ W_fe, int A) {
    for (int idx = threadIdx.x; idx < A * 12; idx += blockDim.x) {
        const int a = idx / 12, c = idx % 12;
        float v;
        if (c == 0) v = sp_b ? sp_b[a] : 0.f;
        else if (c == 1) v = w_ee[a];
        else v = W_fe[a * NCH + (c - 2)];
        uni[idx] = v;
    }
}

// Per-wave staging of PH: wave w consumes columns [ac + 16w, ac + 16w + 16) of the
// chunk for all 64 frames, so it stages exactly those 64 x 16 floats into its OWN LDS
// tile (row stride 17: conflict-free column reads).  No workgroup barrier is needed
// in the chunk loop -- the waves run decoupled and hide each other's latencies; LDS
// operations of one wave execute in order.  The global loads of chunk c+1 are issued
// before chunk c is consumed.  Loads are branch-free (masked), so all four are in
// flight together.
#define WCOLS 16
#define WSTR 17
__device__ __forceinline__ void wstage_load(float4 (&regs)[4], unsigned (&mask)[4],
                                            const float *__restrict__ PHb, int t0, int nrows, int a0, int A,
                                            bool vec) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int f = lane + 64 * it;
        const int row = f >> 2, a = a0 + (f & 3) * 4;
        if (vec) {
            const bool ok = row < nrows && a < A;      // A % 4 == 0: the whole float4 is in range
            regs[it] = ld4(ok ? PHb + (int64_t)(t0 + row) * A + a : PHb);
            mask[it] = ok ? 0xffffffffu : 0u;
        } else {
            float4 v = zero4();
            if (row < nrows && a < A) {
                const float *src = PHb + (int64_t)(t0 + row) * A + a;
                v.x = src[0];
                if (a + 1 < A) v.y = src[1];
                if (a + 2 < A) v.z = src[2];
                if (a + 3 < A) v.w = src[3];
            }
            regs[it] = v;
            mask[it] = 0xffffffffu;
        }
    }
}
__device__ __forceinline__ void wstage_write(float *wt, const float4 (&regs)[4], const unsigned (&mask)[4]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int f = lane + 64 * it;
        float *dst = wt + (f >> 2) * WSTR + (f & 3) * 4;
        const unsigned m = mask[it];
        dst[0] = __uint_as_float(__float_as_uint(regs[it].x) & m);
        dst[1] = __uint_as_float(__float_as_uint(regs[it].y) & m);
        dst[2] = __uint_as_float(__float_as_uint(regs[it].z) & m);
        dst[3] = __uint_as_float(__float_as_uint(regs[it].w) & m);
    }
}

// dynamic LDS layout of the energy kernels (floats)
//   uni  [A*12]   (rounded up to 4)
//   tile [TT*TSTR]
//   misc [TT*NCH + 64 + ...] per kernel
__device__ __forceinline__ int round4(int x) { return (x + 3) & ~3; }

// ------------------------------------------------------- forward: energies --
// grid (ceil(T/TT), B), block EW*64.  Frames t >= lengths[b] are skipped: their
// energies only enter the reference through the max shift of the softmax
// (Models/attention.py:33), which cancels in alpha.
__global__ __launch_bounds__(EW * 64) void energy_fwd_kernel(
    const float *__restrict__ PH, const float *__restrict__ alpha_prev /* (B,T) or null */,
    const float *__restrict__ sp /* (B,A) or null */, const float *__restrict__ w_ee,
    const float *__restrict__ W_fe, const float *__restrict__ conv_w,
    const int32_t *__restrict__ lengths, float *__restrict__ e_buf, float *__restrict__ Fc_s, int T, int A,
    int vec) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const int len = min(lengths[b], T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *uni = smem;
    float *tile = uni + round4(A * 12);
    float *Fs = tile + round4(TILE_FLOATS);         // [TT][NCH]
    float *aw = Fs + TT * NCH;                    // [TT + KW] alpha window
    float *cw = aw + round4(TT + KW);             // [NCH*KW]
    float *red = cw + NCH * KW;                   // [EW][64]
    float *Fout = Fc_s + ((int64_t)b * T + t0) * NCH;
    if (t0 >= len) {
        // still define Fc for the (unused) padded frames of this tile
        const int n = min(TT, T - t0) * NCH;
        for (int i = threadIdx.x; i < n; i += blockDim.x) Fout[i] = 0.f;
        return;
    }
    const int nrows = min(TT, len - t0);
    const float *PHb = PH + (int64_t)b * T * A;
    float4 pf[4];
    unsigned pm[4];
    float *wt = nullptr;
    wstage_load(pf, pm, PHb, t0, nrows, wave * WCOLS, A, vec != 0);   // in flight under the conv prologue

    fill_uni(uni, sp ? sp + (int64_t)b * A : nullptr, w_ee, W_fe, A);
    if (alpha_prev) {
        for (int i = threadIdx.x; i < TT + KW; i += blockDim.x) {
            const int t = t0 - PADW + i;
            aw[i] = (t >= 0 && t < T) ? alpha_prev[(int64_t)b * T + t] : 0.f;
        }
        for (int i = threadIdx.x; i < NCH * KW; i += blockDim.x) cw[i] = conv_w[i];
    }
    __syncthreads();
    // conv features: F[t][c] = sum_k cw[c][k] * alpha_prev[t + k - PADW]; waves 0..4 take 2 channels each
    if (alpha_prev) {
        if (wave < NCH / 2) {
            const int c0 = wave * 2;
            float f0 = 0.f, f1 = 0.f;
#pragma unroll 4
            for (int k = 0; k < KW; ++k) {
                const float av = aw[lane + k];
                f0 += cw[c0 * KW + k] * av;
                f1 += cw[(c0 + 1) * KW + k] * av;
            }
            Fs[lane * NCH + c0] = f0;
            Fs[lane * NCH + c0 + 1] = f1;
        }
    } else {
        for (int i = threadIdx.x; i < TT * NCH; i += blockDim.x) Fs[i] = 0.f;
    }
    __syncthreads();
    float F[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) F[c] = Fs[lane * NCH + c];
    {   // save the conv features for the backward (contiguous, coalesced)
        const int n = min(TT, T - t0) * NCH;
        for (int i = threadIdx.x; i < n; i += blockDim.x) Fout[i] = Fs[i];
    }

    float eacc = 0.f;
    wt = tile + wave * (TT * WSTR);
    for (int ac = 0; ac < A; ac += ACH) {
        __builtin_amdgcn_wave_barrier();
        wstage_write(wt, pf, pm);
        __builtin_amdgcn_wave_barrier();
        if (ac + ACH < A) wstage_load(pf, pm, PHb, t0, nrows, ac + ACH + wave * WCOLS, A, vec != 0);
        const int abase = ac + wave * 16;
#pragma unroll 8
        for (int aa = 0; aa < 16; ++aa) {
            const int a = abase + aa;
            if (a < A) {  // wave-uniform
                const float *u = uni + a * 12;
                float x = wt[lane * WSTR + aa] + u[0];
#pragma unroll
                for (int c = 0; c < NCH; ++c) x += F[c] * u[2 + c];
                eacc += u[1] * tanh_f(x);
            }
        }
    }
    red[wave * 64 + lane] = eacc;
    __syncthreads();
    if (wave == 0 && lane < nrows) {
        float e = 0.f;
#pragma unroll
        for (int w = 0; w < EW; ++w) e += red[w * 64 + lane];
        e_buf[(int64_t)b * T + t0 + lane] = e;
    }
}

static size_t energy_fwd_smem(int A) {
    return sizeof(float) * (size_t)(((A * 12 + 3) & ~3) + ((TILE_FLOATS + 3) & ~3) + TT * NCH +
                                    ((TT + KW + 3) & ~3) + NCH * KW + EW * 64);
}

// ------------------------------------------- forward: softmax + context --
// grid (ceil(E/64), B), block 256, dynamic LDS T floats.
__global__ __launch_bounds__(256) void ctx_kernel(const float *__restrict__ e_buf,
                                                  const float *__restrict__ hbatch,
                                                  const int32_t *__restrict__ lengths,
                                                  float *__restrict__ alpha_s, float *__restrict__ g_s, int T,
                                                  int E, int vec) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float sred[4];
    __shared__ float cred[16][65];
    float *al = smem;  // [T]
    const int b = blockIdx.y, tid = threadIdx.x;
    const int len = min(lengths[b], T);
    const float *eb = e_buf + (int64_t)b * T;
    float mx = -INFINITY;
    for (int t = tid; t < len; t += 256) mx = fmaxf(mx, eb[t]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) sred[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    __syncthreads();
    float sum = 0.f;
    for (int t = tid; t < len; t += 256) {
        const float p = fast_exp(eb[t] - mx);
        al[t] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) sred[tid >> 6] = sum;
    __syncthreads();
    sum = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    const float inv = 1.0f / sum;
    for (int t = tid; t < T; t += 256) {
        const float a = t < len ? al[t] * inv : 0.f;
        if (t < len) al[t] = a;
        if (blockIdx.x == 0) alpha_s[(int64_t)b * T + t] = a;
    }
    __syncthreads();
    const float *hb = hbatch + (int64_t)b * T * E;
    const int e0 = blockIdx.x * 64;
    if (vec) {
        const int tg = tid >> 4, l16 = tid & 15;
        const int e = e0 + l16 * 4;
        float4 acc = zero4();
        if (e < E) {
#pragma unroll 4
            for (int t = tg; t < len; t += 16) {
                const float a = al[t];
                const float4 h = ld4(hb + (int64_t)t * E + e);
                acc.x += a * h.x; acc.y += a * h.y; acc.z += a * h.z; acc.w += a * h.w;
            }
        }
        cred[tg][l16 * 4 + 0] = acc.x; cred[tg][l16 * 4 + 1] = acc.y;
        cred[tg][l16 * 4 + 2] = acc.z; cred[tg][l16 * 4 + 3] = acc.w;
        __syncthreads();
        if (tid < 64 && e0 + tid < E) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += cred[q][tid];
            g_s[(int64_t)b * E + e0 + tid] = s;
        }
    } else {
        const int tg = tid >> 6, l = tid & 63;
        const int e = e0 + l;
        float acc = 0.f;
        if (e < E)
            for (int t = tg; t < len; t += 4) acc += al[t] * hb[(int64_t)t * E + e];
        cred[tg][l] = acc;
        __syncthreads();
        if (tid < 64 && e0 + tid < E)
            g_s[(int64_t)b * E + e0 + tid] = (cred[0][tid] + cred[1][tid]) + (cred[2][tid] + cred[3][tid]);
    }
}

// ------------------------------------------------------- forward: LSTM cell --
// grid (ceil(Hd/4), ceil(B/16)), block 256
__global__ __launch_bounds__(256) void cell_fwd_kernel(
    const float *__restrict__ S_s, const float *__restrict__ C_s, const float *__restrict__ g_s,
    const int64_t *__restrict__ targets, int s, int L, int V, const float *__restrict__ Emb,
    const float *__restrict__ W_ss, const float *__restrict__ W_gs, const float *__restrict__ b_gs,
    float *__restrict__ gates_s, float *__restrict__ S_n, float *__restrict__ C_n, int B, int Hd, int E,
    int vec_s, int vec_g) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, i = lane & 15;
    const int j0 = blockIdx.x * 4, b0 = blockIdx.y * 16;
    {
        const int b = b0 + i;
        const int gate = i >> 2, j = j0 + (i & 3);
        const int64_t wr = (int64_t)gate * Hd + j;
        Tile16Acc acc;
        tile16_init(acc);
        tile16_seg(acc, b < B ? S_s + (int64_t)b * Hd : nullptr, j < Hd ? W_ss + wr * Hd : nullptr, Hd,
                   vec_s != 0, W_ss);
        tile16_seg(acc, b < B ? g_s + (int64_t)b * E : nullptr, j < Hd ? W_gs + wr * E : nullptr, E,
                   vec_g != 0, W_gs);
        tile16_reduce(acc, red);
    }
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int bi = tid >> 2, ju = tid & 3;
        const int b = b0 + bi, j = j0 + ju;
        if (b < B && j < Hd) {
            int64_t y = targets[(int64_t)b * L + s];
            y = y < 0 ? 0 : (y >= V ? V - 1 : y);
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pre[g] = tile16_get(red, bi, g * 4 + ju) + Emb[y * 4 * Hd + g * Hd + j] + b_gs[g * Hd + j];
            const float ig = sigmoid_tanh_form(pre[0]), fg = sigmoid_tanh_form(pre[1]);
            const float gg = tanh_f(pre[2]), og = sigmoid_tanh_form(pre[3]);
            const float c = fg * C_s[(int64_t)b * Hd + j] + ig * gg;
            float *gd = gates_s + (int64_t)b * 4 * Hd;
            gd[0 * Hd + j] = ig; gd[1 * Hd + j] = fg; gd[2 * Hd + j] = gg; gd[3 * Hd + j] = og;
            C_n[(int64_t)b * Hd + j] = c;
            S_n[(int64_t)b * Hd + j] = og * tanh_f(c);
        }
    }
}

// ------------------------------------------------ greedy: output + argmax --
// Batch-1 greedy search (beam_width == 1 path of Models/decoder.py:63-130):
// u = tanh(L_gy g + L_sy s), logits = L_yy u, token = argmax (log_softmax is
// monotone, so it does not change the argmax; ties resolve to the lowest index
// like torch.argmax).  One workgroup; the token is written to tokens[step] and
// consumed by cell_fwd_kernel of the same step, so no host sync per token.
__global__ __launch_bounds__(1024) void greedy_out_kernel(
    const float *__restrict__ S_s, const float *__restrict__ g_s, const float *__restrict__ W_sy,
    const float *__restrict__ W_gy, const float *__restrict__ b_gy, const float *__restrict__ W_yy,
    const float *__restrict__ b_yy, int64_t *__restrict__ tokens, int step, int Hd, int E, int V) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float bestv[16];
    __shared__ int besti[16];
    float *u = smem;           // [Hd]
    float *xs = u + Hd;        // [Hd] state, then [E] context
    float *xg = xs + Hd;
    const int tid = threadIdx.x;
    for (int i = tid; i < Hd; i += 1024) xs[i] = S_s[i];
    for (int i = tid; i < E; i += 1024) xg[i] = g_s[i];
    __syncthreads();
    for (int j = tid; j < Hd; j += 1024) {
        float acc = b_gy[j];
        const float *wg = W_gy + (int64_t)j * E;
        for (int k = 0; k < E; ++k) acc += wg[k] * xg[k];
        const float *wsr = W_sy + (int64_t)j * Hd;
        for (int k = 0; k < Hd; ++k) acc += wsr[k] * xs[k];
        u[j] = tanh_f(acc);
    }
    __syncthreads();
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = tid; v < V; v += 1024) {
        float acc = b_yy[v];
        const float *wy = W_yy + (int64_t)v * Hd;
        for (int k = 0; k < Hd; ++k) acc += wy[k] * u[k];
        if (acc > bv) { bv = acc; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { bestv[tid >> 6] = bv; besti[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) { bv = bestv[w]; bi = besti[w]; }
        tokens[step] = bi;
    }
}


// ======================================================================= v2 ==
// Energy kernels on the matrix cores.  The conv-feature term of the energy,
//     pre[t][a] = PH[t][a] + sproj[a] + sum_c F[t][c] W_fe[a][c],
// is a (16 a) x (16 t) x (K = 12) product per tile: three v_mfma_f32_16x16x4_f32 with
// the accumulator initialised to PH + sproj.  The C/D layout puts 4 consecutive a
// of one frame on a lane (rows a = 4q + r, column t), so PH is read straight from
// global memory as one 16-byte load per lane and tile (two adjacent tiles = one
// 128-byte line per frame) -- no LDS staging, no transposition -- and the ten FMAs per
// element of the scalar form disappear from the VALU.  In the backward the same
// registers are the B operand of dF^T[c][t] += W_fe^T[c][a] dpre[a][t] (four MFMAs
// per tile, accumulated over all tiles), so dF needs no cross-lane traffic either;
// only dsproj[a] = sum_t dpre[t][a] is reduced across the 16 lanes of a row.
// Workgroup = (b, 64 frames): 16 waves = 4 frame tiles x 4 slices of the attention
// dimension.  Requires A % 4 == 0 and 16-byte aligned PH (else the scalar kernels).
#define E2W 16
#define E2P (E2W / 4)
#define E2PF 4      // PH tiles loaded ahead per wave

struct E2Smem {
    float *sp, *wee, *wfe, *Fs, *aw, *cw, *red, *dspl, *dFl;
};
__device__ __forceinline__ int round16(int x) { return (x + 15) & ~15; }

// Padded copies of W_fe ([A16][12], channels 10/11 and rows >= A zero) and w_ee ([A16]),
// built once per forward/backward call so the per-step kernels fill their LDS tables with
// straight 16-byte copies (all loads issued before the first store).
__global__ __launch_bounds__(256) void e2_prep_kernel(const float *__restrict__ w_ee,
                                                      const float *__restrict__ W_fe, int A, int A16,
                                                      float *__restrict__ tab) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < A16 * 12) {
        const int a = idx / 12, c = idx - a * 12;
        tab[idx] = (a < A && c < NCH) ? W_fe[a * NCH + c] : 0.f;
    }
    if (idx < A16) tab[A16 * 12 + idx] = idx < A ? w_ee[idx] : 0.f;
}

// tab = [wfe_pad (A16*12) | wee_pad (A16)] in global memory; m.wfe and m.wee are contiguous in LDS
// in the same order.  sp_b: this utterance's sproj row or null.
// The padded tables (W_fe / w_ee, built once per call by e2_prep_kernel) and this step's state
// projection go global -> registers -> LDS.  Split in two so the caller can put its PH loads in
// flight BETWEEN the halves: vector loads return in issue order, so the tables (L2 hits, needed
// first) must be issued before the PH tiles (HBM, needed after the prologue), not behind them.
struct E2Tab {
    float4 r[4], s4;
};
__device__ __forceinline__ E2Tab e2_tables_load(const float *__restrict__ sp_b, const float *__restrict__ tab,
                                                int A, int A16) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n4 = A16 * 13 / 4;                       // A16 % 16 == 0
    E2Tab t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * nt;
        t.r[i] = ld4(tab + 4 * (f < n4 ? f : 0));
    }
    t.s4 = zero4();
    const int a4 = tid * 4;
    if (sp_b != nullptr && a4 < A) t.s4 = ld4(sp_b + a4);  // A % 4 == 0 on this path
    return t;
}
__device__ __forceinline__ void e2_tables_store(const E2Smem &m, const E2Tab &t, const float *__restrict__ sp_b,
                                                const float *__restrict__ tab, int A, int A16) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n4 = A16 * 13 / 4;
    const int a4 = tid * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * nt;
        if (f < n4) *reinterpret_cast<float4 *>(m.wfe + 4 * f) = t.r[i];
    }
    for (int f = tid + 4 * nt; f < n4; f += nt) *reinterpret_cast<float4 *>(m.wfe + 4 * f) = ld4(tab + 4 * f);
    if (a4 < A16) *reinterpret_cast<float4 *>(m.sp + a4) = t.s4;
    for (int a = a4 + 4 * nt; a < A16; a += 4 * nt)
        *reinterpret_cast<float4 *>(m.sp + a) = (sp_b != nullptr && a < A) ? ld4(sp_b + a) : zero4();
}
__device__ __forceinline__ float4 e2_load_ph(const float *__restrict__ PHrow, bool rowok, int a4, int A,
                                             const float *safe) {
    const bool ok = rowok && a4 < A;
    const float4 q = ld4(ok ? PHrow + a4 : safe);
    const unsigned mk = ok ? 0xffffffffu : 0u;
    return make_float4(__uint_as_float(__float_as_uint(q.x) & mk), __uint_as_float(__float_as_uint(q.y) & mk),
                       __uint_as_float(__float_as_uint(q.z) & mk), __uint_as_float(__float_as_uint(q.w) & mk));
}

// pre-activation tile: acc[r] = pre[t][16*mt + 4q + r]
__device__ __forceinline__ f32x4 e2_pre(const E2Smem &m, int mt, int q, int col, const float4 &ph,
                                        const float (&fB)[3]) {
    const float4 s4 = *reinterpret_cast<const float4 *>(m.sp + 16 * mt + 4 * q);
    f32x4 acc = {ph.x + s4.x, ph.y + s4.y, ph.z + s4.z, ph.w + s4.w};
    const float *wrow = m.wfe + (16 * mt + col) * 12 + q;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[4 * ks], fB[ks], acc, 0, 0, 0);
    return acc;
}

__global__ __launch_bounds__(E2W * 64) void energy_fwd2_kernel(
    const float *__restrict__ PH, const float *__restrict__ alpha_prev, const float *__restrict__ sp,
    const float *__restrict__ tab, const float *__restrict__ conv_w,
    const int32_t *__restrict__ lengths, float *__restrict__ e_buf, float *__restrict__ Fc_s, int T, int A,
    int dbg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const int len = min(lengths[b], T);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int A16 = round16(A);
    E2Smem m;
    m.sp = smem;
    m.wfe = m.sp + A16;
    m.wee = m.wfe + A16 * 12;
    m.Fs = m.wee + A16;               // [TT][12]
    m.aw = m.Fs + TT * 12;            // [TT + KW]
    m.cw = m.aw + round4(TT + KW);    // [NCH*KW]
    m.red = m.cw + NCH * KW;          // [E2P][64]
    float *Fout = Fc_s + ((int64_t)b * T + t0) * NCH;
    if (t0 >= len) {
        const int n = min(TT, T - t0) * NCH;
        for (int i = tid; i < n; i += blockDim.x) Fout[i] = 0.f;
        return;
    }
    const int nrows = min(TT, len - t0);
    const int q = lane >> 4, col = lane & 15;
    const int tt = wave & 3, part = wave >> 2;
    const int tl = 16 * tt + col;
    const bool rowok = tl < nrows && !(dbg & 1);
    const float *PHrow = PH + ((int64_t)b * T + t0 + tl) * A;
    const int MT = A16 / 16, MTP = (MT + E2P - 1) / E2P;
    const int mt0 = part * MTP, mt1 = (dbg & 2) ? mt0 : min(MT, mt0 + MTP);
    // table loads first, then the first PH tiles (they do not depend on the prologue), then the
    // LDS half of the tables
    const float *sp_b = sp ? sp + (int64_t)b * A : nullptr;
    const E2Tab tb = e2_tables_load(sp_b, tab, A, A16);
    __builtin_amdgcn_sched_barrier(0);
    float4 ph[E2PF];
#pragma unroll
    for (int jx = 0; jx < E2PF; ++jx)
        ph[jx] = e2_load_ph(PHrow, rowok && mt0 + jx < mt1, 16 * (mt0 + jx) + 4 * q, A, PH);
    __builtin_amdgcn_sched_barrier(0);
    e2_tables_store(m, tb, sp_b, tab, A, A16);
    if (alpha_prev) {
        for (int i = tid; i < TT + KW; i += blockDim.x) {
            const int t = t0 - PADW + i;
            m.aw[i] = (t >= 0 && t < T) ? alpha_prev[(int64_t)b * T + t] : 0.f;
        }
        for (int i = tid; i < NCH * KW; i += blockDim.x) m.cw[i] = conv_w[i];
    }
    __syncthreads();
    // conv features F[t][c] = sum_k cw[c][k] alpha_prev[t + k - PADW] as a Toeplitz product on the
    // matrix cores: waves 0..3 take 16 frames each, M = t, N = c (16, 10 used), K = 100 taps.
    if (wave < 4) {
        const int q = lane >> 4, col = lane & 15;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (alpha_prev) {
            const float *arow = m.aw + 16 * wave + col + q;                       // A[t][k] = aw[t + k]
            const float *brow = m.cw + (col < NCH ? col : 0) * KW + q;           // B[k][c] = cw[c][k]
#pragma unroll 5
            for (int ks = 0; ks < KW / 4; ++ks) {
                const float bv = col < NCH ? brow[4 * ks] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * ks], bv, acc, 0, 0, 0);
            }
        }
        // D[t = 4q + r][c = col]
        if (col < 12) {
#pragma unroll
            for (int r = 0; r < 4; ++r) m.Fs[(16 * wave + 4 * q + r) * 12 + col] = acc[r];
        }
    }
    __syncthreads();
    {
        const int n = min(TT, T - t0) * NCH;
        for (int i = tid; i < n; i += blockDim.x) Fout[i] = m.Fs[(i / NCH) * 12 + (i % NCH)];
    }
    float fB[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) fB[ks] = m.Fs[tl * 12 + 4 * ks + q];
    float e = 0.f;
    // all PH loads of a chunk of tiles are issued before the first one is consumed: the
    // kernel is bound by memory latency, not by arithmetic
    for (int mc = mt0; mc < mt1; mc += E2PF) {
        if (mc != mt0) {
#pragma unroll
            for (int jx = 0; jx < E2PF; ++jx)
                ph[jx] = e2_load_ph(PHrow, rowok && mc + jx < mt1, 16 * (mc + jx) + 4 * q, A, PH);
        }
#pragma unroll
        for (int jx = 0; jx < E2PF; ++jx) {
            const int mt = mc + jx;
            if (mt < mt1) {   // uniform
                const f32x4 acc = e2_pre(m, mt, q, col, ph[jx], fB);
                const float4 w4 = *reinterpret_cast<const float4 *>(m.wee + 16 * mt + 4 * q);
                e += w4.x * tanh_f(acc[0]) + w4.y * tanh_f(acc[1]) + w4.z * tanh_f(acc[2]) + w4.w * tanh_f(acc[3]);
            }
        }
    }
    e += __shfl_xor(e, 16, 64);
    e += __shfl_xor(e, 32, 64);
    if (q == 0) m.red[part * 64 + tl] = e;
    __syncthreads();
    if (tid < nrows) {
        float v = 0.f;
#pragma unroll
        for (int p = 0; p < E2P; ++p) v += m.red[p * 64 + tid];
        e_buf[(int64_t)b * T + t0 + tid] = v;
    }
}

static size_t energy_fwd2_smem(int A) {
    const int A16 = (A + 15) & ~15;
    return sizeof(float) * (size_t)(A16 * 14 + TT * 12 + ((TT + KW + 3) & ~3) + NCH * KW + E2P * 64);
}

__global__ __launch_bounds__(E2W * 64) void energy_bwd2_kernel(
    const float *__restrict__ PH, const float *__restrict__ alpha_s, const float *__restrict__ dalpha,
    const float *__restrict__ sp, const float *__restrict__ tab,
    const float *__restrict__ Fc_s, const int32_t *__restrict__ lengths, float *__restrict__ de_s,
    float *__restrict__ dF_s, float *__restrict__ dsp_part, int B, int T, int A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const int len = min(lengths[b], T);
    if (t0 >= len) return;
    const int nrows = min(TT, len - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int A16 = round16(A);
    E2Smem m;
    m.sp = smem;
    m.wfe = m.sp + A16;
    m.wee = m.wfe + A16 * 12;
    m.Fs = m.wee + A16;                // [TT][12]
    m.red = m.Fs + TT * 12;            // [E2W]
    m.dspl = m.red + 64;               // [4][A16]
    m.dFl = m.dspl + 4 * A16;          // [E2P][TT][16]
    const int q = lane >> 4, col = lane & 15;
    const int tt = wave & 3, prt = wave >> 2;
    const int tl = 16 * tt + col;
    const bool rowok = tl < nrows;
    const float *PHrow = PH + ((int64_t)b * T + t0 + tl) * A;
    const int MT = A16 / 16, MTP = (MT + E2P - 1) / E2P;
    const int mt0 = prt * MTP, mt1 = min(MT, mt0 + MTP);
    // loads in the order their data is needed: <alpha, dalpha> terms and tables (L2), then the
    // first PH tiles (HBM; in flight under the rest of the prologue)
    float part = 0.f;
    for (int t = tid; t < len; t += E2W * 64) part += alpha_s[(int64_t)b * T + t] * dalpha[(int64_t)b * T + t];
    const E2Tab tb = e2_tables_load(sp + (int64_t)b * A, tab, A, A16);
    __builtin_amdgcn_sched_barrier(0);
    float4 ph[E2PF];
#pragma unroll
    for (int jx = 0; jx < E2PF; ++jx)
        ph[jx] = e2_load_ph(PHrow, rowok && mt0 + jx < mt1, 16 * (mt0 + jx) + 4 * q, A, PH);
    __builtin_amdgcn_sched_barrier(0);
    part = wave_sum(part);
    if (lane == 0) m.red[wave] = part;
    e2_tables_store(m, tb, sp + (int64_t)b * A, tab, A, A16);
    for (int i = tid; i < TT * 12; i += blockDim.x) {
        const int r = i / 12, c = i - r * 12;
        m.Fs[i] = (r < nrows && c < NCH) ? Fc_s[((int64_t)b * T + t0 + r) * NCH + c] : 0.f;
    }
    __syncthreads();
    float dot = 0.f;
#pragma unroll
    for (int w = 0; w < E2W; ++w) dot += m.red[w];
    float de = 0.f;
    if (rowok) {
        const int64_t ti = (int64_t)b * T + t0 + tl;
        de = alpha_s[ti] * (dalpha[ti] - dot);
        if (prt == 0 && q == 0) de_s[ti] = de;
    }
    float fB[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) fB[ks] = m.Fs[tl * 12 + 4 * ks + q];
    f32x4 dFacc = {0.f, 0.f, 0.f, 0.f};     // dF^T[c = 4q + r][t = col], partial over this wave's a-range
    for (int mc = mt0; mc < mt1; mc += E2PF) {
        if (mc != mt0) {
#pragma unroll
            for (int jx = 0; jx < E2PF; ++jx)
                ph[jx] = e2_load_ph(PHrow, rowok && mc + jx < mt1, 16 * (mc + jx) + 4 * q, A, PH);
        }
#pragma unroll
        for (int jx = 0; jx < E2PF; ++jx) {
            const int mt = mc + jx;
            if (mt < mt1) {   // uniform
                const f32x4 acc = e2_pre(m, mt, q, col, ph[jx], fB);
                const float4 w4 = *reinterpret_cast<const float4 *>(m.wee + 16 * mt + 4 * q);
                float dp[4];
                dp[0] = (de * w4.x) * sech2_f(acc[0]);
                dp[1] = (de * w4.y) * sech2_f(acc[1]);
                dp[2] = (de * w4.z) * sech2_f(acc[2]);
                dp[3] = (de * w4.w) * sech2_f(acc[3]);
                // dF^T[c][t] += sum_a W_fe[a][c] dpre[t][a]: k-step r pairs a = 16mt + 4q + r on both operands
                // (columns 12..15 read into the next table row -- still inside the LDS tables, finite --
                // and only feed rows c >= 12 of dF^T, which nobody reads: no mask, no branch)
                const float *wcol = m.wfe + (16 * mt + 4 * q) * 12 + col;
        #pragma unroll
                for (int r = 0; r < 4; ++r)
                    dFacc = __builtin_amdgcn_mfma_f32_16x16x4f32(wcol[r * 12], dp[r], dFacc, 0, 0, 0);
                // dsproj[a] partial: sum over the 16 frames of this tile (the 16 lanes with equal q);
                // lane col ends up with the total of a = 16 mt + 4 q + (col >> 2)
                const float tot = row16_sum4(dp[0], dp[1], dp[2], dp[3]);
                if ((col & 3) == 0) m.dspl[tt * A16 + 16 * mt + 4 * q + (col >> 2)] = tot;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) m.dFl[(prt * TT + tl) * 16 + 4 * q + r] = dFacc[r];
    __syncthreads();
    float *dsp_out = dsp_part + ((int64_t)blockIdx.x * B + b) * A;
    for (int a = tid; a < A; a += blockDim.x)
        dsp_out[a] = (m.dspl[a] + m.dspl[A16 + a]) + (m.dspl[2 * A16 + a] + m.dspl[3 * A16 + a]);
    for (int i = tid; i < nrows * NCH; i += blockDim.x) {
        const int r = i / NCH, c = i - r * NCH;
        float v = 0.f;
#pragma unroll
        for (int p = 0; p < E2P; ++p) v += m.dFl[(p * TT + r) * 16 + c];
        dF_s[((int64_t)b * T + t0) * NCH + i] = v;
    }
}

static size_t energy_bwd2_smem(int A) {
    const int A16 = (A + 15) & ~15;
    return sizeof(float) * (size_t)(A16 * 14 + TT * 12 + 64 + 4 * A16 + E2P * TT * 16);
}

static bool al16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// ===================================================== persistent forward ==
// The whole teacher-forced loop of Models/decoder.py:52-60 in ONE launch, with everything a step re-reads kept
// ON CHIP for all L steps.  Workgroup (b, k) of the grid B x ceil(T/64) -- 512 threads, one per compute unit, all
// resident -- holds in REGISTERS
//   * the 64-frame tile k of PH[b] (all attention units): 20 float4 per lane in the C/D layout of the energy MFMAs,
//   * the column chunk k of hbatch[b] (all frames): 20 float4 per lane for the context sum,
// so a step's energies and context cost no memory traffic at all (the per-step kernels stream 2 x 41 MB per step).
// The M = 32 products of a step are 16 x 16 tile jobs on the same workgroups: cell jobs (4 hidden units x 16 batch
// rows) on workgroups [0, NJ*MTB), state-projection jobs (16 attention units x 16 rows) on the next NSP*MTB -- or, where
// the grid ends first (short utterances), the remaining ones on the first cell workgroups as a second job.
// Per label step the data crosses workgroups four times:
//   sproj_s[b]     -> energy (b, k)                           (edge P, 2.5 KB in per workgroup)
//   e[b, 64 frames]-> softmax + context chunk, all (b, *)     (edge E, 2 KB in)
//   g_s[b, chunk]  -> cell jobs of b's batch tile             (edge G, 41 KB in per job)
//   S_{s+1}        -> state-projection jobs, and the W_ss S term of the next step's cell jobs (edge S, 20 KB in)
// Hand-off: as in csrc/blstm.hip -- 16-byte write-through (sc1) stores of an EXCHANGED COPY whose every float carries
// the step's epoch bit (s >> 1) & 1 in its mantissa LSB (<= 1 ulp; the saved activations S, g, sproj, alpha are
// stored untouched beside it), two parity slots per edge pre-filled with 0xFF, and the consumer re-loads (sc1) until
// every tag matches: no flag, no store drain, no fence.  A slot is rewritten two steps later; by then every reader
// is past it (each edge's producer at step s + 2 depends, through the chain above, on every consumer of its step-s
// data having produced something later).  Bounded spins: a wait that runs out sets *err and the workgroup leaves,
// the others then run out the same way, so the grid always drains.  Weights of a job (static) are fetched BEFORE
// its wait, so only the handed-off operand's latency is on the chain.  alpha_{s-1} never leaves the workgroup:
// every workgroup of an utterance normalises the utterance's energies itself and keeps them in LDS for the
// next step's convolution.
#define ATTDEC_PERSIST_DEFAULT 1
#define PD_THREADS 512
#define PD_WAVES (PD_THREADS / 64)
#define PD_MTP 20          // 16-unit tiles of the attention dimension per wave (2 parts): A <= 640
#define PD_HF 20           // frames per thread of the context chunk
#define PD_HF_MAX 22       // ... of the forward instantiation for long utterances (T = 800: 39 frame groups x 21)
#define PD_AL (32 * PD_MTP)   // attention units the LDS tables are padded to (zeros past A)
#define PD_KF 5            // float4 per lane of a job's K slot: K <= 32 slots x 20 = 640
// in-launch waits are bounded by wall time: asr_wait_expired (common.h), 4 s on the 100 MHz counter
#define PD_OOB 0x7ffffff0
#define PD_NPACE 8
typedef unsigned int pd_u32x4 __attribute__((ext_vector_type(4)));

struct PDParams {
    asr_attdec_fwd_args a;
    const float *tab;      // padded W_fe / w_ee tables (e2_prep_kernel)
    float *xch;            // exchange area (tagged copies, two parity slots each): X_sp | X_e | X_g | X_S
    unsigned *err;         // sticky error word
    int NTILE, EC, NJ, NSP, MTB, nfg, A16, Tp, Tal;
    int o_e, o_g, o_S, xch_bytes;
    unsigned long long *stamps;   // diagnostics only (asr_attdec_debug_stamps): [workgroup][step][PD_NSTAMP] 100 MHz clock
    int pace[PD_NPACE];           // pause before the first gather attempt of each wait (asr_attdec_pace), units of 256 cycles
};
// A wait that starts polling long before its data can be there does harm: every poll of a few hundred workgroups is
// fabric traffic in front of the very stores being waited for.  pd_pause holds the wave off the memory system for
// about the time the producer still needs (s_sleep 4 = 256 cycles per unit).
__device__ __forceinline__ void pd_pause(int n) {
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);
}
#define PD_NSTAMP 10
#define PD_STAMP(i)                                                                                   \
    do {                                                                                              \
        if (STAMPS && p.stamps != nullptr && tid == 0)                                                \
            p.stamps[((int64_t)w * L + s) * PD_NSTAMP + (i)] = __builtin_amdgcn_s_memrealtime();      \
    } while (0)

__device__ __forceinline__ float4 pd_ld4_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    pd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16 /* sc1 */);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ int pd_opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ unsigned pd_tag(float v, unsigned e) { return (__float_as_uint(v) & ~1u) | e; }
__device__ __forceinline__ void pd_st4_tag(__amdgpu_buffer_rsrc_t r, int byte_off, const float4 &v, unsigned e) {
    const pd_u32x4 pk = {pd_tag(v.x, e), pd_tag(v.y, e), pd_tag(v.z, e), pd_tag(v.w, e)};
    __builtin_amdgcn_raw_buffer_store_b128(pk, r, byte_off, 0, 16 /* sc1 */);
}
__device__ __forceinline__ bool pd_tags_ok(const float4 &v, unsigned e) {
    return (((__float_as_uint(v.x) ^ e) | (__float_as_uint(v.y) ^ e) | (__float_as_uint(v.z) ^ e) |
             (__float_as_uint(v.w) ^ e)) & 1u) == 0u;
}
// wave-uniform tagged gather: (re)issues NF sc1 loads per lane until every tag of every lane carries `want`
// (offsets PD_OOB read zeros and count as valid).  While the data is not there yet the wave polls ONE granule per
// lane at most (`probe`: a granule of each producer the wave depends on, PD_OOB for lanes that need not look) --
// hundreds of workgroups re-reading their whole 20-40 KB operand every spin kept the fabric busy enough to delay
// the very stores they wait for.  The probe is only a hint: the full gather checks every tag and retries.
// On a timeout sets the abort word and the sticky error.
// PROBE = false (call sites that pass PD_OOB for every lane): no probe stage at all -- even a probe at PD_OOB is a
// load the wave has to wait for, and loads return in order, so it put everything issued before it (a job's weight
// rows) in front of the first gather.
// Returns the number of full-gather attempts that failed (0 = the first one found everything).
template <int NF, bool PROBE = true>
__device__ __forceinline__ int pd_gather(float4 (&v)[NF], __amdgpu_buffer_rsrc_t r, const int (&off)[NF], unsigned want,
                                         int probe, int *s_abort, unsigned *err) {
    int spins = 0;
    unsigned long long t0 = 0;
    bool late = false;
    for (; PROBE; ++spins) {
        const float4 t = pd_ld4_sc1(r, probe);
        if (__all(pd_tags_ok(t, want) | (probe == PD_OOB))) break;
        if (asr_wait_expired(spins, t0)) {
            late = true;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    // the full gather: normally one pass; a probe that ran out of time gives it one attempt
    unsigned long long t1 = 0;
    for (spins = 0;; ++spins) {
#pragma unroll
        for (int f = 0; f < NF; ++f) v[f] = pd_ld4_sc1(r, off[f]);
        __builtin_amdgcn_sched_barrier(0);
        bool ok = true;
#pragma unroll
        for (int f = 0; f < NF; ++f) ok = ok & (pd_tags_ok(v[f], want) | (off[f] == PD_OOB));
        if (__all(ok)) return spins;
        if (late || asr_wait_expired(spins, t1)) {
            if ((threadIdx.x & 63) == 0) {
                *s_abort = 1;
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return spins;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

static size_t attdec_persist_smem(int Tal) {
    const int A16 = PD_AL;
    return sizeof(float) * (size_t)(A16 * 14 + 2 * TT * 12 + ((TT + KW + 3) & ~3) + NCH * KW + 2 * 64 + 2 * Tal + 2048 +
                                    2 * PD_WAVES * 16 * 17 + 16 * 17 + 32 + 16);
}

// STAMPS: the diagnostics build (tools/dec_stamps.py); the product path instantiates it without the ten per-step branches
// HF: frames per thread of the context chunk (PD_HF; PD_HF_MAX where T needs more than nfg * PD_HF frames)
template <bool STAMPS, int HF>
__global__ __launch_bounds__(PD_THREADS) void attdec_fwd_persist_kernel(PDParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const asr_attdec_fwd_args &a = p.a;
    const int B = a.B, T = a.T, E = a.E, Hd = a.Hd, L = a.L, V = a.V, A = 2 * Hd;
    const int NTILE = p.NTILE, MTB = p.MTB, EC = p.EC, A16 = p.A16, Tp = p.Tp, Tal = p.Tal;
    const int w = blockIdx.x;
    const int b = w / NTILE, k = w - b * NTILE, t0 = k * TT;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int len = min(a.lengths[b], T);
    E2Smem m;
    m.sp = smem;                         // tables padded to PD_AL units: zeros past A
    m.wfe = m.sp + PD_AL;
    m.wee = m.wfe + PD_AL * 12;
    m.Fs = m.wee + PD_AL;                // [2][TT][12]: the two halves of the taps
    m.aw = m.Fs + 2 * TT * 12;           // [TT + KW]
    m.cw = m.aw + round4(TT + KW);       // [NCH*KW]
    m.red = m.cw + NCH * KW;             // [2][64]
    float *al = m.red + 2 * 64;          // [2][Tal] alpha of the previous / current step (zero past the length)
    float *cred = al + 2 * Tal;          // [nfg][EC] context partial sums
    float (*red16)[16][17] = reinterpret_cast<float (*)[16][17]>(cred + 2048);                        // phase D
    float (*red16e)[16][17] = reinterpret_cast<float (*)[16][17]>(cred + 2048 + PD_WAVES * 16 * 17);  // phase E
    float *accS = cred + 2048 + 2 * PD_WAVES * 16 * 17;     // [16][17]: W_ss S_s of this workgroup's cell job
    float *sred = accS + 16 * 17;                       // [32]
    int *s_abort = reinterpret_cast<int *>(sred + 32);
    float *wsum = sred + 34;

    const int64_t BH = (int64_t)B * Hd, BT = (int64_t)B * T, BA = (int64_t)B * A, BE = (int64_t)B * E;
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(p.xch, 0, p.xch_bytes, 0x00020000);

    // tables that stay in LDS for all L steps
    constexpr float K2L = 2.88539008177792681472f;     // 2 log2(e): tanh(x) = 1 - 2 / (1 + 2^(K2L x))
    for (int i = tid; i < PD_AL * 12; i += PD_THREADS) m.wfe[i] = i < A16 * 12 ? p.tab[i] * K2L : 0.f;
    for (int i = tid; i < PD_AL; i += PD_THREADS) m.wee[i] = i < A16 ? p.tab[A16 * 12 + i] : 0.f;
    for (int i = tid; i < NCH * KW; i += PD_THREADS) m.cw[i] = a.conv_w[i];
    for (int i = tid; i < PD_AL; i += PD_THREADS) m.sp[i] = 0.f;
    for (int i = tid; i < 2 * Tal; i += PD_THREADS) al[i] = 0.f;
    if (wave == 0) {    // sum of w_ee over the attention units (padded entries are zero)
        float v = 0.f;
        for (int i = tid; i < A16; i += 64) v += p.tab[A16 * 12 + i];
        v = wave_sum(v);
        if (tid == 0) *wsum = v;
    }
    for (int i = tid; i < 16 * 17; i += PD_THREADS) accS[i] = 0.f;
    if (tid == 0) *s_abort = 0;

    // Index arithmetic is re-derived from an OPAQUE copy of the thread id at the top of every phase: whatever the
    // compiler can prove invariant in s it hoists and keeps live across the whole step loop, and with 160 registers
    // of resident data per lane there is no room for that.
    const int nrows = max(0, min(TT, len - t0));
    const int MT = A16 / 16;
    const int ncg = EC / 4, nfg = p.nfg;
    const int c0 = k * EC;
    // ---- energy role (as energy_fwd2_kernel): 4 frame tiles x 2 parts of the attention dimension
    float4 phr[PD_MTP];
    {
        const int lane = tid & 63, q = lane >> 4, col = lane & 15, tt = wave & 3, part = wave >> 2;
        const int tl = 16 * tt + col;
        const int mt0 = part * PD_MTP, mt1 = min(MT, mt0 + PD_MTP);
        const float *PHrow = a.PH + ((int64_t)b * T + t0 + tl) * A;
#pragma unroll
        for (int j = 0; j < PD_MTP; ++j)
        {
            const float4 v = e2_load_ph(PHrow, tl < nrows && mt0 + j < mt1, 16 * (mt0 + j) + 4 * q, A, a.PH);
            phr[j] = make_float4(v.x * K2L, v.y * K2L, v.z * K2L, v.w * K2L);
        }
    }
    // ---- context role: column chunk k of hbatch[b], frames fg + nfg i
    float4 hreg[HF];
    {
        const int cg = tid % ncg, fg = tid / ncg;
        const bool cact = fg < nfg && c0 + 4 * cg < E;
        const float *hb = a.hbatch + (int64_t)b * T * E + c0 + 4 * cg;
#pragma unroll
        for (int i = 0; i < HF; ++i) {
            const int t = fg + nfg * i;
            const bool ok = cact && t < len;
            const float4 v = ld4(ok ? hb + (int64_t)t * E : a.hbatch);
            const unsigned mk = ok ? 0xffffffffu : 0u;
            hreg[i] = make_float4(__uint_as_float(__float_as_uint(v.x) & mk), __uint_as_float(__float_as_uint(v.y) & mk),
                                  __uint_as_float(__float_as_uint(v.z) & mk), __uint_as_float(__float_as_uint(v.w) & mk));
        }
    }
    // ---- job roles: cell jobs on workgroups [0, ncell); state projection jobs on the workgroups behind them, and,
    // when those run out (ncell + nsp > workgroups: short utterances), one each on the first cell workgroups as well
    const int ncell = p.NJ * MTB, nsp = p.NSP * MTB;
    const int nsp_own = min(nsp, (int)gridDim.x - ncell);     // state projection jobs with a workgroup of their own
    const bool is_cell = w < ncell;
    const int sp_job = !is_cell ? (w - ncell < nsp_own ? w - ncell : -1) : (nsp_own + w < nsp ? nsp_own + w : -1);
    const int jw = is_cell ? w : 0;
    const int jt = jw / MTB, jb0 = 16 * (jw - jt * MTB);     // cell job: tile index along the units, first batch row
    int kqE = (E + 31) / 32, kqH = (Hd + 31) / 32;
    kqE = (kqE + 3) & ~3;
    kqH = (kqH + 3) & ~3;
    float c_reg = 0.f;        // cell state of the owner threads (16 rows x 4 units on threads 0..63)
    int pace_d = p.pace[2], fails_d = 0;
    __syncthreads();

    for (int s = 0; s < L; ++s) {
        const int slot = s & 1, pslot = slot ^ 1;
        const unsigned ep = (unsigned)((s >> 1) & 1), pep = (unsigned)(((s - 1) >> 1) & 1);
        float *al_cur = al + slot * Tal;
        const float *al_prev = al + pslot * Tal;
        PD_STAMP(0);
        // embedding row + bias of this step's label for the cell owners: a dependent pair of loads (label, then
        // its row), issued here so that it is not in front of the context gather of phase D
        float eb[4] = {0.f, 0.f, 0.f, 0.f};
        if (is_cell) {
            const int tid = pd_opaque((int)threadIdx.x);
            const int ob = jb0 + (tid >> 2), oj = 4 * jt + (tid & 3);
            if (tid < 64 && ob < B && oj < Hd) {
                int64_t y = a.targets[(int64_t)ob * L + s];
                y = y < 0 ? 0 : (y >= V ? V - 1 : y);
#pragma unroll
                for (int g = 0; g < 4; ++g) eb[g] = a.Emb[y * 4 * Hd + g * Hd + oj] + a.b_gs[g * Hd + oj];
            }
        }
        // ---- A. the alignment window of the convolution, the conv features (they need the previous step's alignment
        // only, which this workgroup has had since its own softmax: done BEFORE the wait, in its shadow), then this
        // utterance's state projection (edge P)
        if (nrows > 0) {
            const int tid = pd_opaque((int)threadIdx.x);
            for (int i = tid; i < TT + KW; i += PD_THREADS) {
                const int t = t0 - PADW + i;
                m.aw[i] = (t >= 0 && t < T) ? al_prev[t] : 0.f;
            }
        }
        __syncthreads();
        if (nrows > 0) {
            const int tid = pd_opaque((int)threadIdx.x);
            const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, col = lane & 15, tt = wave & 3, part = wave >> 2;
            {   // conv features as a Toeplitz product, the 25 k-steps of the taps split over the two parts
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};     // two chains: MFMA latency
                const float *arow = m.aw + 16 * tt + col + q;
                const float *brow = m.cw + (col < NCH ? col : 0) * KW + q;
                const int ks0 = part ? 13 : 0;
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int ks = ks0 + 2 * u;
                    const float b0 = col < NCH ? brow[4 * ks] : 0.f, b1 = col < NCH ? brow[4 * ks + 4] : 0.f;
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * ks], b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * ks + 4], b1, acc1, 0, 0, 0);
                }
                if (!part) {
                    const float b0 = col < NCH ? brow[4 * 12] : 0.f;
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * 12], b0, acc0, 0, 0, 0);
                }
                if (col < 12) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) m.Fs[part * TT * 12 + (16 * tt + 4 * q + r) * 12 + col] = acc0[r] + acc1[r];
                }
            }
            if (s > 0 && (tid >> 6) * 256 < A) {          // wave-uniform: waves that hold a granule
                const int off[1] = {4 * tid < A ? ((pslot * B + b) * A16 + 4 * tid) * 4 : PD_OOB};
                float4 v[1];
                pd_pause(p.pace[0]);
                pd_gather<1, false>(v, rX, off, pep, PD_OOB, s_abort, p.err);
                if (4 * tid < A) {
                    float *d = m.wfe + 4 * tid * 12 + 10;
                    d[0] = v[0].x * K2L; d[12] = v[0].y * K2L; d[24] = v[0].z * K2L; d[36] = v[0].w * K2L;
                }
            }
        }
        __syncthreads();
        if (*s_abort) return;
        PD_STAMP(1);
        // ---- B. energies of this workgroup's 64 frames
        if (nrows > 0) {
            const int tid = pd_opaque((int)threadIdx.x);
            const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, col = lane & 15, tt = wave & 3, part = wave >> 2;
            const int tl = 16 * tt + col;
            const int mt0 = part * PD_MTP;
            float fB[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) fB[ks] = m.Fs[tl * 12 + 4 * ks + q] + m.Fs[TT * 12 + tl * 12 + 4 * ks + q];
            if (q == 2) fB[2] = 1.0f;     // k-slot 10 (ks = 2, q = 2) pairs with column 10 of the table: sproj rides there
            // PH (registers), sproj and W_fe (LDS) carry the factor 2 log2(e), so the MFMA result is the argument of
            // 2^x directly; sum_a w_a tanh(x_a) = sum_a w_a - 2 sum_a w_a / (1 + e^{2 x_a}): one FMA per element
            // No bounds: tiles past A read zero tables (w = 0) and zero PH.  Two tiles per round, the LDS operands of the
            // next round fetched before this round's arithmetic: two independent MFMA chains, eight independent 2^x / 1/x.
            struct PDTile { float4 w4; float a0, a1, a2; };
            auto ldt = [&](int mt) {
                PDTile t;
                const float *wr = m.wfe + (16 * mt + col) * 12 + q;
                t.a0 = wr[0]; t.a1 = wr[4]; t.a2 = wr[8];
                t.w4 = *reinterpret_cast<const float4 *>(m.wee + 16 * mt + 4 * q);
                return t;
            };
            PDTile c0 = ldt(mt0), c1 = ldt(mt0 + 1);
            float e = 0.f, e1 = 0.f;
            // Software pipeline over the tile pairs: the three-MFMA chains of pair j + 1 are ISSUED before the
            // transcendental work of pair j, so the matrix pipe runs under this wave's own 2^x / 1/x stream instead of
            // in front of it (an MFMA holds the vector issue port for 8 of its 32 cycles only).
            auto xmm = [&](const PDTile &t0_, const PDTile &t1_, int j, f32x4 &x0, f32x4 &x1) {
                x0 = f32x4{phr[j].x, phr[j].y, phr[j].z, phr[j].w};
                x1 = f32x4{phr[j + 1].x, phr[j + 1].y, phr[j + 1].z, phr[j + 1].w};
                x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t0_.a0, fB[0], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t1_.a0, fB[0], x1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t0_.a1, fB[1], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t1_.a1, fB[1], x1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t0_.a2, fB[2], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t1_.a2, fB[2], x1, 0, 0, 0);
            };
            f32x4 x0, x1;
            xmm(c0, c1, 0, x0, x1);
#pragma unroll
            for (int j = 0; j < PD_MTP; j += 2) {
                PDTile n0 = c0, n1 = c1;
                f32x4 y0 = x0, y1 = x1;
                if (j + 2 < PD_MTP) {
                    n0 = ldt(mt0 + j + 2); n1 = ldt(mt0 + j + 3);
                    xmm(n0, n1, j + 2, y0, y1);
                }
                float t[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { t[r] = __builtin_amdgcn_exp2f(x0[r]); t[4 + r] = __builtin_amdgcn_exp2f(x1[r]); }
#pragma unroll
                for (int r = 0; r < 8; ++r) t[r] = __builtin_amdgcn_rcpf(1.0f + t[r]);
                e += c0.w4.x * t[0]; e1 += c1.w4.x * t[4];
                e += c0.w4.y * t[1]; e1 += c1.w4.y * t[5];
                e += c0.w4.z * t[2]; e1 += c1.w4.z * t[6];
                e += c0.w4.w * t[3]; e1 += c1.w4.w * t[7];
                c0 = n0; c1 = n1;
                x0 = y0; x1 = y1;
            }
            e += e1;
            e += __shfl_xor(e, 16, 64);
            e += __shfl_xor(e, 32, 64);
            if (q == 0) m.red[part * 64 + tl] = e;
            __syncthreads();
            if (tid < 16 && 4 * tid < nrows) {
                const float4 r0 = *reinterpret_cast<const float4 *>(m.red + 4 * tid);
                const float4 r1 = *reinterpret_cast<const float4 *>(m.red + 64 + 4 * tid);
                const float ws_ = *wsum;
                pd_st4_tag(rX, (p.o_e + (slot * B + b) * Tp + t0 + 4 * tid) * 4,
                           make_float4(ws_ - 2.0f * (r0.x + r1.x), ws_ - 2.0f * (r0.y + r1.y), ws_ - 2.0f * (r0.z + r1.z),
                                       ws_ - 2.0f * (r0.w + r1.w)), ep);
            }
            // saved conv features: behind the hand-off store, nothing in the launch reads them
            float *Fout = a.Fc + (((int64_t)s * B + b) * T + t0) * NCH;
            const int n = min(TT, T - t0) * NCH;
            for (int i = tid; i < n; i += PD_THREADS) {
                const int r = i / NCH, c = i - r * NCH;
                Fout[i] = m.Fs[r * 12 + c] + m.Fs[TT * 12 + r * 12 + c];
            }
        } else if (t0 < T) {
            const int tid = pd_opaque((int)threadIdx.x);
            float *Fout = a.Fc + (((int64_t)s * B + b) * T + t0) * NCH;
            const int n = min(TT, T - t0) * NCH;
            for (int i = tid; i < n; i += PD_THREADS) Fout[i] = 0.f;
        }
        PD_STAMP(2);
        // ---- C. softmax over the utterance (every workgroup of b for itself), context chunk k
        {
            const int tid = pd_opaque((int)threadIdx.x);
            const int lane = tid & 63, wave = tid >> 6;
            const int ng = (len + 3) >> 2;
            float4 ev = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            if (wave * 64 < ng) {      // wave-uniform
                const int off[1] = {tid < ng ? (p.o_e + (slot * B + b) * Tp + 4 * tid) * 4 : PD_OOB};
                float4 v[1];
                pd_pause(p.pace[1]);
                pd_gather<1, false>(v, rX, off, ep, PD_OOB, s_abort, p.err);
                if (tid < ng) {
                    ev.x = v[0].x;
                    if (4 * tid + 1 < len) ev.y = v[0].y;
                    if (4 * tid + 2 < len) ev.z = v[0].z;
                    if (4 * tid + 3 < len) ev.w = v[0].w;
                }
            }
            PD_STAMP(3);
            float mx = wave_max(fmaxf(fmaxf(ev.x, ev.y), fmaxf(ev.z, ev.w)));
            if (lane == 0) sred[wave] = mx;
            __syncthreads();
            if (*s_abort) return;
            mx = sred[0];
#pragma unroll
            for (int i = 1; i < PD_WAVES; ++i) mx = fmaxf(mx, sred[i]);
            float4 pe = make_float4(fast_exp(ev.x - mx), fast_exp(ev.y - mx), fast_exp(ev.z - mx), fast_exp(ev.w - mx));
            float sum = wave_sum((pe.x + pe.y) + (pe.z + pe.w));
            if (lane == 0) sred[8 + wave] = sum;
            __syncthreads();
            sum = 0.f;
#pragma unroll
            for (int i = 0; i < PD_WAVES; ++i) sum += sred[8 + i];
            const float inv = 1.0f / sum;
            pe.x *= inv; pe.y *= inv; pe.z *= inv; pe.w *= inv;
            if (tid < ng) *reinterpret_cast<float4 *>(al_cur + 4 * tid) = pe;
            {   // this tile's 64 frames of the saved alignment (zeros past the length)
                float *alpha_s = a.alpha + (int64_t)s * BT + (int64_t)b * T;
                const int t = 4 * tid;
                if (t >= t0 && t < t0 + TT && t < T) {
                    if ((T & 3) == 0) {
                        *reinterpret_cast<float4 *>(alpha_s + t) = pe;
                    } else {
                        alpha_s[t] = pe.x;
                        if (t + 1 < T) alpha_s[t + 1] = pe.y;
                        if (t + 2 < T) alpha_s[t + 2] = pe.z;
                        if (t + 3 < T) alpha_s[t + 3] = pe.w;
                    }
                }
            }
            __syncthreads();
        }
        {
            const int tid = pd_opaque((int)threadIdx.x);
            const int cg = tid % ncg, fg = tid / ncg;
            if (fg < nfg) {
                float4 acc = zero4();
                const float *ap = al_cur + fg;
#pragma unroll
                for (int i = 0; i < HF; ++i) {
                    const float av = ap[nfg * i];
                    acc.x += av * hreg[i].x; acc.y += av * hreg[i].y; acc.z += av * hreg[i].z; acc.w += av * hreg[i].w;
                }
                *reinterpret_cast<float4 *>(cred + fg * EC + 4 * cg) = acc;
            }
            __syncthreads();
            if (tid < ncg && c0 + 4 * tid < E) {
                float4 g4 = zero4();
                for (int f = 0; f < nfg; ++f) {
                    const float4 v = *reinterpret_cast<const float4 *>(cred + f * EC + 4 * tid);
                    g4.x += v.x; g4.y += v.y; g4.z += v.z; g4.w += v.w;
                }
                pd_st4_tag(rX, (p.o_g + slot * B * E + ((c0 >> 2) + tid) * 4 * B + 4 * b) * 4, g4, ep);
                *reinterpret_cast<float4 *>(a.g + (int64_t)s * BE + (int64_t)b * E + c0 + 4 * tid) = g4;
            }
        }
        PD_STAMP(4);
        // ---- D. cell job: 4 hidden units (all four gates) x 16 batch rows; the W_ss S_s term waits in accS
#ifndef PD_NO_D
        if (is_cell) {
            const int tid = pd_opaque((int)threadIdx.x);
            const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, ji = lane & 15;
            const int jrow = jb0 + ji;                 // A row (batch) of this lane
            const int cu = 4 * jt + (ji & 3);          // B row: gate (ji >> 2) of unit cu
            const int kbE = (wave * 4 + q) * kqE, keE = min(E, kbE + kqE);
            const float *wrow_gs = cu < Hd ? a.W_gs + ((int64_t)(ji >> 2) * Hd + cu) * E : nullptr;
            float4 wv[PD_KF];
#pragma unroll
            for (int f = 0; f < PD_KF; ++f) wv[f] = t16_load_fast(wrow_gs, kbE + 4 * f, keE, a.W_gs);
            const int obi = tid >> 2, oju = tid & 3;
            const int ob = jb0 + obi, oj = 4 * jt + oju;
            const bool owner = tid < 64 && ob < B && oj < Hd;
            int off[PD_KF];
#pragma unroll
            for (int f = 0; f < PD_KF; ++f)
                off[f] = (jrow < B && kbE + 4 * f < keE) ? (p.o_g + slot * B * E + ((kbE >> 2) + f) * 4 * B + 4 * jrow) * 4 : PD_OOB;
            float4 av[PD_KF];
            // no probe: the other producers finish with this workgroup's own context chunk, the wait is a spin or two
            // paced first attempt (see PDParams.pace): p.pace[2] to start with, then by the wave's own record -- two
            // steps longer when two or more of the last eight first attempts failed, one shorter when none did
            pd_pause(pace_d);
            fails_d += pd_gather<PD_KF, false>(av, rX, off, ep, PD_OOB, s_abort, p.err) > 0;
            if ((s & 7) == 7) {
                if (p.pace[5] == 0) pace_d = fails_d >= 2 ? min(pace_d + 2, 64) : (fails_d == 0 ? max(pace_d - 1, 0) : pace_d);
                fails_d = 0;
            }
            PD_STAMP(5);
            Tile16Acc acc;
            tile16_init(acc);
#pragma unroll
            for (int f = 0; f < PD_KF; ++f) {
                acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f].x, wv[f].x, acc.a0, 0, 0, 0);
                acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f].y, wv[f].y, acc.a1, 0, 0, 0);
                acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f].z, wv[f].z, acc.a0, 0, 0, 0);
                acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f].w, wv[f].w, acc.a1, 0, 0, 0);
            }
            tile16_reduce(acc, red16);
            if (*s_abort) return;
            if (tid < 64) {
                float hval = 0.f;
                if (owner) {
                    float pre[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        pre[g] = tile16_get(red16, obi, g * 4 + oju) + accS[obi * 17 + g * 4 + oju] + eb[g];
                    const float ig = sigmoid_tanh_form(pre[0]), fgt = sigmoid_tanh_form(pre[1]);
                    const float gg = tanh_f(pre[2]), og = sigmoid_tanh_form(pre[3]);
                    c_reg = fgt * c_reg + ig * gg;
                    hval = og * tanh_f(c_reg);
                    float *gd = a.gates + (int64_t)s * 4 * BH + (int64_t)ob * 4 * Hd;
                    gd[0 * Hd + oj] = ig; gd[1 * Hd + oj] = fgt; gd[2 * Hd + oj] = gg; gd[3 * Hd + oj] = og;
                    a.C[(int64_t)(s + 1) * BH + (int64_t)ob * Hd + oj] = c_reg;
                    a.S[(int64_t)(s + 1) * BH + (int64_t)ob * Hd + oj] = hval;
                }
                // the 4 units of a batch row sit on 4 adjacent lanes: one 16-byte tagged store per row
                const float h1 = ASR_DPP(hval, 0x55), h2 = ASR_DPP(hval, 0xAA), h3 = ASR_DPP(hval, 0xFF);
                if (oju == 0 && ob < B && 4 * jt < Hd)
                    pd_st4_tag(rX, (p.o_S + slot * B * Hd + jt * 4 * B + 4 * ob) * 4, make_float4(hval, h1, h2, h3), ep);
            }
            PD_STAMP(6);
        }
#endif
        // ---- E. products with S_{s+1}: state projection of the next step; W_ss S_{s+1} of the next cell job
#ifndef PD_NO_E
        // one 16 x 16 product with S_{s+1}: cellmode -> W_ss rows of cell job (ejt, ejb0) into accS; otherwise the state
        // projection job (ejt, ejb0), published for the next step's phase A
        auto do_E = [&](bool cellmode, int ejt, int ejb0) {
            const int tid = pd_opaque((int)threadIdx.x);
            const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, ji = lane & 15;
            const int jrow = ejb0 + ji;
            const int cu = 4 * ejt + (ji & 3);
            const int kbH = (wave * 4 + q) * kqH, keH = min(Hd, kbH + kqH);
            // weight row of this lane: cell job -> gate (ji >> 2) of unit cu; state projection -> attention unit 16 ejt + ji
            const float *wrow_S = cellmode ? (cu < Hd ? a.W_ss + ((int64_t)(ji >> 2) * Hd + cu) * Hd : nullptr)
                                           : (16 * ejt + ji < A ? a.W_se + (int64_t)(16 * ejt + ji) * Hd : nullptr);
            constexpr int KFH = 3;         // Hd <= 32 slots x 12
            float4 wv[KFH];
#pragma unroll
            for (int f = 0; f < KFH; ++f) wv[f] = t16_load_fast(wrow_S, kbH + 4 * f, keH, a.W_se);
            int off[KFH];
#pragma unroll
            for (int f = 0; f < KFH; ++f)
                off[f] = (jrow < B && kbH + 4 * f < keH) ? (p.o_S + slot * B * Hd + ((kbH >> 2) + f) * 4 * B + 4 * jrow) * 4 : PD_OOB;
            float4 av[KFH];
            // probe: lanes 0..2 of each k-slot look at one granule of each of the slot's producers
            pd_pause(p.pace[cellmode ? 4 : 3]);
            pd_gather<KFH>(av, rX, off, ep, ji == 0 ? off[0] : (ji == 1 ? off[1] : (ji == 2 ? off[2] : PD_OOB)), s_abort, p.err);
            Tile16Acc acc;
            tile16_init(acc);
#pragma unroll
            for (int f = 0; f < KFH; ++f) {
                acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f].x, wv[f].x, acc.a0, 0, 0, 0);
                acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f].y, wv[f].y, acc.a1, 0, 0, 0);
                acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f].z, wv[f].z, acc.a0, 0, 0, 0);
                acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f].w, wv[f].w, acc.a1, 0, 0, 0);
            }
            tile16_reduce(acc, red16e);
            if (*s_abort) return;
            if (cellmode) {
                if (tid < 256) accS[(tid >> 4) * 17 + (tid & 15)] = tile16_get(red16e, tid >> 4, tid & 15);
            } else if (tid < 64) {
                const int bi = tid >> 2, n4 = (tid & 3) * 4;
                const int bb = ejb0 + bi, n0 = 16 * ejt + n4;
                if (bb < B && n0 < A) {
                    const float4 v = make_float4(tile16_get(red16e, bi, n4), tile16_get(red16e, bi, n4 + 1),
                                                 tile16_get(red16e, bi, n4 + 2), tile16_get(red16e, bi, n4 + 3));
                    pd_st4_tag(rX, ((slot * B + bb) * A16 + n0) * 4, v, ep);
                    *reinterpret_cast<float4 *>(a.sproj + (int64_t)(s + 1) * BA + (int64_t)bb * A + n0) = v;
                }
            }
        };
        if (s + 1 < L) {
            // a state projection job first (the next step's phase A waits for it), then the cell job's own W_ss term
            if (sp_job >= 0) {
                do_E(false, sp_job / MTB, 16 * (sp_job % MTB));
                if (*s_abort) return;
                PD_STAMP(7);
                if (is_cell) __syncthreads();        // red16e is read by the first 64 threads above
            }
            if (is_cell) {
                do_E(true, jt, jb0);
                if (*s_abort) return;
            }
            PD_STAMP(8);
        }
#endif
        PD_STAMP(9);
    }
}


// workspace layout (floats): e_buf [round4(B*T)] | 64 | tables [A16*13] | 64 | persistent kernel: exchange area
// X_sp [2][B][A16] | X_e [2][B][Tp] | X_g [2][B][E] | X_S [2][B][Hd]  (Tp = 64 ceil(T/64)) | 64 | error word | 63
static int64_t attdec_ws_xch_offset(int B, int T, int Hd) {
    const int64_t A16 = (2 * Hd + 15) & ~15;
    return (((int64_t)B * T + 3) & ~(int64_t)3) + 64 + A16 * 13 + 64;
}
static int64_t attdec_ws_xch_floats(int B, int T, int E, int Hd) {
    const int64_t A16 = (2 * Hd + 15) & ~15, Tp = (int64_t)cdiv(T, TT) * TT;
    return 2 * (int64_t)B * (A16 + Tp + E + Hd);
}
extern "C" int64_t asr_attdec_ws_floats(int B, int T, int E, int Hd, int L) {
    (void)L;
    return attdec_ws_xch_offset(B, T, Hd) + attdec_ws_xch_floats(B, T, E, Hd) + 64 + 64;
}

// Index (in floats) of the persistent kernel's sticky error word inside the workspace: set when an in-launch
// wait timed out (the outputs are then garbage); the Python wrapper folds it into the outputs (NaN).
extern "C" int64_t asr_attdec_err_offset(int B, int T, int E, int Hd, int L) {
    (void)L;
    return attdec_ws_xch_offset(B, T, Hd) + attdec_ws_xch_floats(B, T, E, Hd) + 64;
}

// diagnostics: a device buffer of B*ceil(T/64) * L * 10 uint64 that the next persistent launches fill with phase
// time stamps (tools/dec_stamps.py); NULL switches it off.  Never set by the product path.
static unsigned long long *g_attdec_stamps = nullptr;
extern "C" void asr_attdec_debug_stamps(void *buf) { g_attdec_stamps = static_cast<unsigned long long *>(buf); }
// pauses in front of the loop kernels' waits, units of 256 cycles: forward 0 = state projection (phase A), 1 = energies
// (C), 2 = context (D: the START value, each wave then adapts it to its own record of failed first attempts unless
// entry 5 is non-zero), 3 / 4 = state for the projection job / the cell job's W_ss term (E); backward 8.. (see there).
// asr_attdec_pace(which, units) sets one (timing experiments, tools/dec_pace.py) and returns the old value; units < 0 only reads.
// Defaults from tools/dec_pace.py at BASELINE configs 2, 4 and 5: only the waits that re-read a whole operand per attempt
// respond -- the cell job's context gather (160 workgroups x 40 KB per attempt: 2.03 -> 1.81 ms per forward pass at
// config 2, 6.15 -> 5.63 at config 4, optimum 12 units = 1.7 us at both, 15 at config 5) and, a little, the backward's
// second wait (4 units: -0.04 ms at configs 2 and 4).  The backward's first wait gains 0.015 ms at config 2 with 12
// units but loses 0.28 ms at config 4 and 0.17 at config 5: left at 0.
static int g_pd_pace[2 * PD_NPACE] = {0, 0, 12, 0, 0, 0, 0, 0, 0, 4, 0, 0, 0, 0, 0, 0};
extern "C" int asr_attdec_pace(int which, int units) {
    if (which < 0 || which >= 2 * PD_NPACE) return -1;
    const int old = g_pd_pace[which];
    if (units >= 0) g_pd_pace[which] = units > 4096 ? 4096 : units;
    return old;
}

static int g_attdec_persist = -1;           // asr_attdec_persist(0) selects the one-launch-per-kernel loop (A/B, tests)
static int attdec_persist_on() {
    if (g_attdec_persist < 0) {
        const char *e = getenv("ASR_ATTDEC_PERSIST");
        g_attdec_persist = (e && atoi(e) == 1) ? 1 : ATTDEC_PERSIST_DEFAULT;
        if (e && atoi(e) == 0) g_attdec_persist = 0;
    }
    return g_attdec_persist;
}
extern "C" int asr_attdec_persist(int on) {
    const int old = attdec_persist_on();
    if (on == 0 || on == 1) g_attdec_persist = on;
    return old;
}

// Shape test of the one-launch forward (everything but pointer alignment): the frames-per-thread instantiation to
// use (PD_HF or PD_HF_MAX), 0 when the shape does not fit -- B x ceil(T/64) workgroups must each get a compute unit.
static int attdec_fwd_persist_hf(int B, int T, int E, int Hd, int n_cu) {
    const int A = 2 * Hd, A16p = (A + 15) & ~15, NTILE = cdiv(T, TT), MTB = cdiv(B, 16);
    const int NJ = cdiv(Hd, 4), NSP = cdiv(A, 16), EC = ((cdiv(E, NTILE) + 3) / 4) * 4, nfg = PD_THREADS / (EC / 4);
    const int64_t nwg = (int64_t)B * NTILE;
    const int hf = (int64_t)nfg * PD_HF >= T ? PD_HF : PD_HF_MAX;
    const int Tal = (std::max(NTILE * TT, nfg * hf) + 3) & ~3;
    const bool ok = A % 4 == 0 && E % 4 == 0 && Hd % 4 == 0 && energy_fwd2_smem(A) <= 150 * 1024 && A % 16 == 0 &&
                    nwg <= n_cu && (int64_t)NJ * MTB <= nwg && (int64_t)NSP * MTB <= nwg && A16p <= 32 * PD_MTP &&
                    E <= 128 * PD_KF && Hd <= 384 && EC / 4 <= PD_THREADS && (int64_t)nfg * hf >= T &&
                    attdec_persist_smem(Tal) <= 150 * 1024 && attdec_ws_xch_floats(B, T, E, Hd) * 4 < (int64_t)0x7fff0000;
    return ok ? hf : 0;
}

// One launch instead of four or five memsets in front of a decoder pass: zero up to three float ranges and one word,
// fill the exchange area with 0xFF (every exchanged float starts with the "wrong" epoch tag).  Sizes in floats;
// pointers 16-byte aligned or null.
__global__ __launch_bounds__(256) void attdec_init_kernel(float *z0, int64_t n0, float *z1, int64_t n1, float *z2, int64_t n2,
                                                          unsigned *word, float *ff, int64_t nff) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n0) z0[i] = 0.f;
    if (i < n1) z1[i] = 0.f;
    if (i < n2) z2[i] = 0.f;
    if (i == 0 && word != nullptr) *word = 0u;
    if (4 * i + 3 < nff) reinterpret_cast<uint4 *>(ff)[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    else for (int64_t e = 4 * i; e < nff; ++e) reinterpret_cast<unsigned *>(ff)[e] = 0xffffffffu;
}
static void attdec_init(float *z0, int64_t n0, float *z1, int64_t n1, float *z2, int64_t n2, unsigned *word, float *ff,
                        int64_t nff, hipStream_t st) {
    int64_t n = std::max(std::max(n0, n1), std::max(n2, (nff + 3) / 4));
    if (n < 1) n = 1;
    hipLaunchKernelGGL(attdec_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z0, n0, z1, n1, z2, n2, word,
                       ff, nff);
}

struct GreedyParams {
    const float *W_sy, *W_gy, *b_gy, *W_yy, *b_yy;
    int64_t *tokens;
};

static int run_attdec_fwd(const asr_attdec_fwd_args *a, const GreedyParams *gp, void *stream) {
    ASR_CHECK_ARG(a != nullptr, "attdec_fwd: null args");
    const int B = a->B, T = a->T, E = a->E, Hd = a->Hd, L = a->L, V = a->V, A = 2 * Hd;
    ASR_CHECK_ARG(B > 0 && T > 0 && E > 0 && Hd > 0 && L > 0 && V > 0, "attdec_fwd: bad shape");
    ASR_CHECK_ARG(a->hbatch && a->PH && a->lengths && a->targets && a->W_se && a->w_ee && a->W_fe &&
                      a->conv_w && a->Emb && a->W_ss && a->W_gs && a->b_gs && a->S && a->C && a->alpha &&
                      a->sproj && a->g && a->gates && a->Fc && a->ws,
                  "attdec_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t esm = energy_fwd_smem(A);
    ASR_CHECK_ARG(esm <= 160 * 1024 && (size_t)T * 4 <= 128 * 1024, "attdec_fwd: T=%d / A=%d exceed the LDS budget", T, A);
    static thread_local bool attr_set = false;
    if (!attr_set) {
        ASR_CHECK_HIP(hipFuncSetAttribute((const void *)energy_fwd_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        ASR_CHECK_HIP(hipFuncSetAttribute((const void *)ctx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          128 * 1024));
        attr_set = true;
    }
    const int64_t BH = (int64_t)B * Hd, BT = (int64_t)B * T, BA = (int64_t)B * A, BE = (int64_t)B * E;
    const int A16p = (A + 15) & ~15;
    float *e_buf = a->ws;
    float *tab = a->ws + ((BT + 3) & ~(int64_t)3) + 64;   // padded W_fe / w_ee tables of the v2 energy kernel
    unsigned *err_word = reinterpret_cast<unsigned *>(a->ws + asr_attdec_err_offset(B, T, E, Hd, L));
    bool inited = false;        // S[0], C[0], sproj[0], the error word (and the exchange area): one launch, below
    const int vecPH = al16(a->PH) && A % 4 == 0;
    const size_t esm2 = energy_fwd2_smem(A);
    const bool use_v2 = vecPH && esm2 <= 150 * 1024;
    const char *dbg_env = getenv("ASR_ATT_DBG");   // timing experiments only
    const int att_dbg = dbg_env ? atoi(dbg_env) : 0;
    if (use_v2) {
        static thread_local bool attr2 = false;
        if (!attr2) {
            ASR_CHECK_HIP(hipFuncSetAttribute((const void *)energy_fwd2_kernel,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr2 = true;
        }
    }
    const int vecH = al16(a->hbatch) && E % 4 == 0;
    const int vecS = al16(a->S) && al16(a->W_se) && al16(a->W_ss) && Hd % 4 == 0;
    const int vecG = al16(a->g) && al16(a->W_gs) && E % 4 == 0;
    if (use_v2)
        hipLaunchKernelGGL(e2_prep_kernel, dim3(cdiv((int64_t)A16p * 12, 256)), dim3(256), 0, st, a->w_ee, a->W_fe,
                           A, A16p, tab);
    // one launch for the whole loop when every workgroup of the (B x frame tile) grid gets a compute unit of its own
    {
        static int n_cu = -1;
        if (n_cu < 0) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
                n_cu = 0;
        }
        const int NTILE = cdiv(T, TT), MTB = cdiv(B, 16);
        PDParams p;
        p.NTILE = NTILE;
        p.MTB = MTB;
        p.NJ = cdiv(Hd, 4);
        p.NSP = cdiv(A, 16);
        p.EC = ((cdiv(E, NTILE) + 3) / 4) * 4;
        p.nfg = PD_THREADS / (p.EC / 4);
        p.A16 = A16p;
        p.Tp = NTILE * TT;
        const int hf = attdec_fwd_persist_hf(B, T, E, Hd, n_cu);     // 0: the shape does not fit the one-launch form
        p.Tal = (std::max(p.Tp, p.nfg * std::max(hf, PD_HF)) + 3) & ~3;
        const size_t psm = attdec_persist_smem(p.Tal);
        const int64_t xfl = attdec_ws_xch_floats(B, T, E, Hd);
        const int64_t nwg = (int64_t)B * NTILE;
        if (attdec_persist_on() && !gp && use_v2 && vecH && vecS && vecG && hf > 0 &&
            al16(a->g) && al16(a->sproj) && al16(a->ws)) {
            static thread_local bool attr3 = false;
            if (!attr3) {
                ASR_CHECK_HIP(hipFuncSetAttribute((const void *)attdec_fwd_persist_kernel<false, PD_HF>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                ASR_CHECK_HIP(hipFuncSetAttribute((const void *)attdec_fwd_persist_kernel<true, PD_HF>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                ASR_CHECK_HIP(hipFuncSetAttribute((const void *)attdec_fwd_persist_kernel<false, PD_HF_MAX>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                attr3 = true;
            }
            p.a = *a;
            p.tab = tab;
            p.xch = a->ws + attdec_ws_xch_offset(B, T, Hd);
            p.o_e = 2 * B * A16p;
            p.o_g = p.o_e + 2 * B * p.Tp;
            p.o_S = p.o_g + 2 * B * E;
            p.xch_bytes = (int)(xfl * 4);
            p.err = reinterpret_cast<unsigned *>(a->ws + asr_attdec_err_offset(B, T, E, Hd, L));
            p.stamps = g_attdec_stamps;
            for (int i = 0; i < PD_NPACE; ++i) p.pace[i] = g_pd_pace[i];
            // every exchanged float starts with the "wrong" epoch tag (LSB = 1)
            attdec_init(a->S, BH, a->C, BH, a->sproj, BA, err_word, p.xch, xfl, st);
            inited = true;
            if (hf > PD_HF) hipLaunchKernelGGL((attdec_fwd_persist_kernel<false, PD_HF_MAX>), dim3((unsigned)nwg), dim3(PD_THREADS), psm, st, p);
            else if (p.stamps) hipLaunchKernelGGL((attdec_fwd_persist_kernel<true, PD_HF>), dim3((unsigned)nwg), dim3(PD_THREADS), psm, st, p);
            else hipLaunchKernelGGL((attdec_fwd_persist_kernel<false, PD_HF>), dim3((unsigned)nwg), dim3(PD_THREADS), psm, st, p);
            ASR_CHECK_LAUNCH("attdec_fwd_persist_kernel");
            return ASR_OK;
        }
    }
    if (!inited) attdec_init(a->S, BH, a->C, BH, a->sproj, BA, err_word, nullptr, 0, st);
    for (int s = 0; s < L; ++s) {
        const float *S_s = a->S + s * BH;
        float *sp_s = a->sproj + s * BA;
        if (s > 0) {
            hipLaunchKernelGGL(sproj_kernel, dim3(cdiv(A, 16), cdiv(B, 16)), dim3(256), 0, st, S_s, a->W_se,
                               sp_s, B, Hd, A, vecS);
        }
        if (use_v2) {
            hipLaunchKernelGGL(energy_fwd2_kernel, dim3(cdiv(T, TT), B), dim3(E2W * 64), esm2, st, a->PH,
                               s > 0 ? a->alpha + (s - 1) * BT : (const float *)nullptr,
                               s > 0 ? (const float *)sp_s : (const float *)nullptr, (const float *)tab,
                               a->conv_w, a->lengths, e_buf, a->Fc + (int64_t)s * BT * NCH, T, A, att_dbg);
        } else {
            hipLaunchKernelGGL(energy_fwd_kernel, dim3(cdiv(T, TT), B), dim3(EW * 64), esm, st, a->PH,
                               s > 0 ? a->alpha + (s - 1) * BT : (const float *)nullptr,
                               s > 0 ? (const float *)sp_s : (const float *)nullptr, a->w_ee, a->W_fe,
                               a->conv_w, a->lengths, e_buf, a->Fc + (int64_t)s * BT * NCH, T, A, vecPH);
        }
        hipLaunchKernelGGL(ctx_kernel, dim3(cdiv(E, 64), B), dim3(256), sizeof(float) * T, st, e_buf,
                           a->hbatch, a->lengths, a->alpha + s * BT, a->g + s * BE, T, E, vecH);
        if (gp) {
            hipLaunchKernelGGL(greedy_out_kernel, dim3(1), dim3(1024), sizeof(float) * (2 * Hd + E), st, S_s,
                               (const float *)(a->g + s * BE), gp->W_sy, gp->W_gy, gp->b_gy, gp->W_yy, gp->b_yy,
                               gp->tokens, s, Hd, E, V);
        }
        hipLaunchKernelGGL(cell_fwd_kernel, dim3(cdiv(Hd, 4), cdiv(B, 16)), dim3(256), 0, st, S_s,
                           a->C + s * BH, a->g + s * BE, a->targets, s, L, V, a->Emb, a->W_ss, a->W_gs,
                           a->b_gs, a->gates + s * 4 * BH, a->S + (s + 1) * BH, a->C + (s + 1) * BH, B, Hd, E,
                           vecS, vecG);
    }
    ASR_CHECK_LAUNCH("attdec_fwd kernels");
    return ASR_OK;
}

// ------------------------------------------------ single steps (beam search) --
// The two halves of one decoder step for callers that choose the fed-back token themselves
// (beam search, Models/decoder.py:79-123): attention + context from a given state and previous
// alignment, then -- after the caller has picked tokens -- the recurrent cell.  Same kernels as
// the training forward; a is read as one step (L = 1): S, C = state in ([B][Hd]), S + B*Hd,
// C + B*Hd = state out, alpha/sproj/g/gates/Fc = this step's slabs.
extern "C" int asr_attdec_step_att(const asr_attdec_fwd_args *a, const float *alpha_prev, void *stream) {
    ASR_CHECK_ARG(a != nullptr, "attdec_step_att: null args");
    const int B = a->B, T = a->T, E = a->E, Hd = a->Hd, A = 2 * Hd;
    ASR_CHECK_ARG(B > 0 && T > 0 && E > 0 && Hd > 0, "attdec_step_att: bad shape");
    ASR_CHECK_ARG(a->hbatch && a->PH && a->lengths && a->W_se && a->w_ee && a->W_fe && a->conv_w && a->S &&
                      a->alpha && a->sproj && a->g && a->Fc && a->ws,
                  "attdec_step_att: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t esm = energy_fwd_smem(A);
    ASR_CHECK_ARG(esm <= 160 * 1024 && (size_t)T * 4 <= 128 * 1024, "attdec_step_att: T=%d / A=%d exceed the LDS budget", T, A);
    static thread_local bool attr_set = false;
    if (!attr_set) {
        ASR_CHECK_HIP(hipFuncSetAttribute((const void *)energy_fwd_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        ASR_CHECK_HIP(hipFuncSetAttribute((const void *)ctx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          128 * 1024));
        ASR_CHECK_HIP(hipFuncSetAttribute((const void *)energy_fwd2_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set = true;
    }
    const int64_t BT = (int64_t)B * T;
    const int A16p = (A + 15) & ~15;
    float *e_buf = a->ws;
    float *tab = a->ws + ((BT + 3) & ~(int64_t)3) + 64;
    const int vecPH = al16(a->PH) && A % 4 == 0;
    const size_t esm2 = energy_fwd2_smem(A);
    const bool use_v2 = vecPH && esm2 <= 150 * 1024;
    const int vecH = al16(a->hbatch) && E % 4 == 0;
    const int vecS = al16(a->S) && al16(a->W_se) && Hd % 4 == 0;
    // W_se s: with the all-zero initial state this is zero, as in the first training step
    hipLaunchKernelGGL(sproj_kernel, dim3(cdiv(A, 16), cdiv(B, 16)), dim3(256), 0, st, (const float *)a->S, a->W_se,
                       a->sproj, B, Hd, A, vecS);
    if (use_v2) {
        hipLaunchKernelGGL(e2_prep_kernel, dim3(cdiv((int64_t)A16p * 12, 256)), dim3(256), 0, st, a->w_ee, a->W_fe,
                           A, A16p, tab);
        hipLaunchKernelGGL(energy_fwd2_kernel, dim3(cdiv(T, TT), B), dim3(E2W * 64), esm2, st, a->PH, alpha_prev,
                           (const float *)a->sproj, (const float *)tab, a->conv_w, a->lengths, e_buf, a->Fc, T, A, 0);
    } else {
        hipLaunchKernelGGL(energy_fwd_kernel, dim3(cdiv(T, TT), B), dim3(EW * 64), esm, st, a->PH, alpha_prev,
                           (const float *)a->sproj, a->w_ee, a->W_fe, a->conv_w, a->lengths, e_buf, a->Fc, T, A,
                           vecPH);
    }
    hipLaunchKernelGGL(ctx_kernel, dim3(cdiv(E, 64), B), dim3(256), sizeof(float) * T, st, e_buf, a->hbatch,
                       a->lengths, a->alpha, a->g, T, E, vecH);
    ASR_CHECK_LAUNCH("attdec_step_att kernels");
    return ASR_OK;
}

extern "C" int asr_attdec_step_cell(const asr_attdec_fwd_args *a, void *stream) {
    ASR_CHECK_ARG(a != nullptr, "attdec_step_cell: null args");
    const int B = a->B, E = a->E, Hd = a->Hd, V = a->V;
    ASR_CHECK_ARG(B > 0 && E > 0 && Hd > 0 && V > 0, "attdec_step_cell: bad shape");
    ASR_CHECK_ARG(a->targets && a->Emb && a->W_ss && a->W_gs && a->b_gs && a->S && a->C && a->g && a->gates,
                  "attdec_step_cell: null pointer");
    const int64_t BH = (int64_t)B * Hd;
    const int vecS = al16(a->S) && al16(a->W_ss) && Hd % 4 == 0;
    const int vecG = al16(a->g) && al16(a->W_gs) && E % 4 == 0;
    hipLaunchKernelGGL(cell_fwd_kernel, dim3(cdiv(Hd, 4), cdiv(B, 16)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)a->S, (const float *)a->C, (const float *)a->g, a->targets, 0, 1, V, a->Emb,
                       a->W_ss, a->W_gs, a->b_gs, a->gates, a->S + BH, a->C + BH, B, Hd, E, vecS, vecG);
    ASR_CHECK_LAUNCH("cell_fwd_kernel");
    return ASR_OK;
}

extern "C" int asr_attdec_fwd(const asr_attdec_fwd_args *a, void *stream) {
    return run_attdec_fwd(a, nullptr, stream);
}

extern "C" int asr_attdec_greedy(const asr_attdec_fwd_args *a, const float *W_sy, const float *W_gy,
                                 const float *b_gy, const float *W_yy, const float *b_yy, int64_t *tokens,
                                 void *stream) {
    ASR_CHECK_ARG(a != nullptr && W_sy && W_gy && b_gy && W_yy && b_yy && tokens, "attdec_greedy: null pointer");
    ASR_CHECK_ARG(a->B == 1, "attdec_greedy: batch must be 1 (Models/decoder.py:63-130 is batch-1)");
    ASR_CHECK_ARG((const void *)a->targets == (const void *)tokens,
                  "attdec_greedy: a->targets must alias the token buffer");
    ASR_CHECK_ARG(sizeof(float) * (size_t)(2 * a->Hd + a->E) <= 60 * 1024, "attdec_greedy: Hd/E too large");
    GreedyParams gp{W_sy, W_gy, b_gy, W_yy, b_yy, tokens};
    return run_attdec_fwd(a, &gp, stream);
}

// =================================================================== backward ==
// Per step s = L-1 .. 0 (4 launches):
//   dscell  : dS_{s+1} = drec_{s+1} W_ss + dyp_{s+1} W_sy + dsp_{s+1} W_se  (tile16),
//             then the cell backward of step s -> drec_s, dC carry
//   dg      : dg_s = drec_s W_gs + dyp_s W_gy                                (tile16)
//   dalpha  : dalpha = h . dg  +  conv^T(dF_{s+1})
//   energy_bwd : de = alpha (dalpha - <alpha,dalpha>); dpre = de w_ee (1 - th^2);
//                dsp partials (sum over t), dF = dpre W_fe (lane-local)
// After the loop: dPH / dW_fe / dw_ee in one pass over PH with the step loop
// inside (the tanh is recomputed; no read-modify-write of dPH per step), and
// the conv-filter gradient as a correlation of dF_s with alpha_{s-1}.

// grid (ceil(Hd/16), ceil(B/16)), block 1024; dynamic LDS: 16*A floats when lds_sum != 0
__global__ __launch_bounds__(1024) void dscell_kernel(
    int s, int L, const float *__restrict__ drec_n, const float *__restrict__ dyp_n,
    const float *__restrict__ dsp_part, int64_t part_stride, float *__restrict__ dsp_n,
    const float *__restrict__ W_ssT, const float *__restrict__ W_syT, const float *__restrict__ W_seT,
    const float *__restrict__ gates_s, const float *__restrict__ C_s, const float *__restrict__ C_n,
    const int32_t *__restrict__ lengths, float *__restrict__ dC, float *__restrict__ drec_s, int B, int Hd,
    int T, int vec, int lds_sum) {
    __shared__ float red[T16_MAXW][16][17];
    extern __shared__ __attribute__((aligned(16))) float dsps[];   // [16][A]: summed dsp partials
    const int lane = threadIdx.x & 63, i = lane & 15;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int A = 2 * Hd;
    const bool has_next = s + 1 < L;
    const int tid = threadIdx.x;
    // operands of the cell backward: fetched first, their latency hides under the products
    const int ebi = tid >> 4, eji = tid & 15;
    const int eb = b0 + ebi, ej = j0 + eji;
    const bool ev = tid < 256 && eb < B && ej < Hd;
    float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cn = 0.f, cp = 0.f, dCv = 0.f;
    if (ev) {
        const float *gd = gates_s + (int64_t)eb * 4 * Hd;
        ig = gd[ej]; fg = gd[Hd + ej]; gg = gd[2 * Hd + ej]; og = gd[3 * Hd + ej];
        cn = C_n[(int64_t)eb * Hd + ej]; cp = C_s[(int64_t)eb * Hd + ej];
        dCv = dC[(int64_t)eb * Hd + ej];
    }
    if (has_next) {
        const int b = b0 + i, j = j0 + i;
        const bool bv = b < B, jv = j < Hd;
        const int ntile = (T + TT - 1) / TT;
        if (lds_sum) {
            // the attention-state gradient arrives as one partial per frame tile: sum them
            // once per workgroup, every thread's loads in flight together, into LDS
            const int A4 = A / 4;
            for (int it = tid; it < 16 * A4; it += 1024) {
                const int r = it / A4, c4 = it % A4;
                const int bb = b0 + r;
                const int np = bb < B ? max(1, (min(lengths[bb], T) + TT - 1) / TT) : 0;
                const float *src = dsp_part + (int64_t)min(bb, B - 1) * A + c4 * 4;
                float4 acc = zero4();
                for (int p0 = 0; p0 < ntile; p0 += 8) {
                    float4 t[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const bool ok = p0 + c < np;
                        t[c] = ld4(ok ? src + (int64_t)(p0 + c) * part_stride : W_seT);
                        if (!ok) t[c] = zero4();
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) { acc.x += t[c].x; acc.y += t[c].y; acc.z += t[c].z; acc.w += t[c].w; }
                }
                *reinterpret_cast<float4 *>(dsps + r * A + c4 * 4) = acc;
                if (blockIdx.x == 0 && bb < B) *reinterpret_cast<float4 *>(dsp_n + (int64_t)bb * A + c4 * 4) = acc;
            }
        }
        Tile16Acc acc;
        tile16_init(acc);
        tile16_seg(acc, bv ? drec_n + (int64_t)b * 4 * Hd : nullptr, jv ? W_ssT + (int64_t)j * 4 * Hd : nullptr,
                   4 * Hd, vec != 0, W_ssT);
        tile16_seg(acc, bv ? dyp_n + (int64_t)b * Hd : nullptr, jv ? W_syT + (int64_t)j * Hd : nullptr, Hd,
                   vec != 0, W_syT);
        if (lds_sum) {
            __syncthreads();
            // same k-slot split as tile16_seg, A operand from LDS
            const int wave = tid >> 6, q = lane >> 4;
            const int slots = (blockDim.x >> 6) * 4;
            int kq = (A + slots - 1) / slots;
            kq = (kq + 3) & ~3;
            const int kb = (wave * 4 + q) * kq;
            const int ke = min(A, kb + kq);
            const float *wrow = jv ? W_seT + (int64_t)j * A : nullptr;
#pragma unroll 4
            for (int u0 = 0; u0 < kq; u0 += 4) {
                const int k = kb + u0;
                float4 a = zero4();
                if (k < ke) a = *reinterpret_cast<const float4 *>(dsps + i * A + k);
                const float4 bq = t16_load_fast(wrow, k, ke, W_seT);
                acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq.x, acc.a0, 0, 0, 0);
                acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq.y, acc.a1, 0, 0, 0);
                acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq.z, acc.a0, 0, 0, 0);
                acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq.w, acc.a1, 0, 0, 0);
            }
        } else {
            int nparts = 1;
            if (bv) nparts = max(1, (min(lengths[b], T) + TT - 1) / TT);
            tile16_seg(acc, bv ? dsp_part + (int64_t)b * A : nullptr, jv ? W_seT + (int64_t)j * A : nullptr, A,
                       vec != 0, W_seT, true, nparts, part_stride,
                       (blockIdx.x == 0 && bv) ? dsp_n + (int64_t)b * A : nullptr, ntile);
        }
        tile16_reduce(acc, red);
    }
    if (ev) {
        const float dS = has_next ? tile16_get(red, ebi, eji) : 0.f;
        const float tc = tanh_f(cn);
        const float dO = dS * tc;
        const float dc = dS * og * (1.f - tc * tc) + dCv;
        float *dr = drec_s + (int64_t)eb * 4 * Hd;
        dr[ej] = dc * gg * ig * (1.f - ig);
        dr[Hd + ej] = dc * cp * fg * (1.f - fg);
        dr[2 * Hd + ej] = dc * ig * (1.f - gg * gg);
        dr[3 * Hd + ej] = dO * og * (1.f - og);
        dC[(int64_t)eb * Hd + ej] = dc * fg;
    }
}

// grid (ceil(E/16), ceil(B/16)), block 1024
__global__ __launch_bounds__(1024) void dg_kernel(const float *__restrict__ drec_s,
                                                  const float *__restrict__ dyp_s,
                                                  const float *__restrict__ W_gsT,
                                                  const float *__restrict__ W_gyT, float *__restrict__ dg_s,
                                                  int B, int Hd, int E, int vec) {
    __shared__ float red[T16_MAXW][16][17];
    const int lane = threadIdx.x & 63, i = lane & 15;
    const int e0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int b = b0 + i, e = e0 + i;
    Tile16Acc acc;
    tile16_init(acc);
    tile16_seg(acc, b < B ? drec_s + (int64_t)b * 4 * Hd : nullptr, e < E ? W_gsT + (int64_t)e * 4 * Hd : nullptr,
               4 * Hd, vec != 0, W_gsT);
    tile16_seg(acc, b < B ? dyp_s + (int64_t)b * Hd : nullptr, e < E ? W_gyT + (int64_t)e * Hd : nullptr, Hd,
               vec != 0, W_gyT);
    tile16_reduce(acc, red);
    const int tid = threadIdx.x;
    if (tid < 256) {
        const int bi = tid >> 4, ei = tid & 15;
        if (b0 + bi < B && e0 + ei < E) dg_s[(int64_t)(b0 + bi) * E + e0 + ei] = tile16_get(red, bi, ei);
    }
}

// dalpha[b][t] = sum_e dg[b][e] h[b][t][e] + sum_{c,k} dF_next[b][t+PADW-k][c] cw[c][k]
// grid (ceil(T/32), B), block 256 (one wave per frame, 8 frames per wave).
// dynamic LDS: dgs[E] + dFw[(32+KW-1)][11] + cws[NCH*KW]
#define DA_TT 32
__global__ __launch_bounds__(256) void dalpha_kernel(const float *__restrict__ hbatch,
                                                     const float *__restrict__ dg_s,
                                                     const float *__restrict__ dF_next /* (B,T,NCH) or null */,
                                                     const float *__restrict__ conv_w,
                                                     const int32_t *__restrict__ lengths,
                                                     float *__restrict__ dalpha, int T, int E, int vec) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.y, t0 = blockIdx.x * DA_TT;
    const int len = min(lengths[b], T);
    if (t0 >= len) return;
    float *dgs = smem;                               // [round4(E)]
    float *dFw = dgs + ((E + 3) & ~3);               // [(DA_TT + KW - 1)][11]
    float *cws = dFw + (DA_TT + KW - 1) * 11 + 1;    // [NCH*KW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < E; i += 256) dgs[i] = dg_s[(int64_t)b * E + i];
    if (dF_next) {
        // window rows r = 0 .. DA_TT+KW-2  <->  frame t0 - (KW-1-PADW) + r
        for (int i = tid; i < (DA_TT + KW - 1) * NCH; i += 256) {
            const int r = i / NCH, c = i % NCH;
            const int t = t0 - (KW - 1 - PADW) + r;
            dFw[r * 11 + c] = (t >= 0 && t < len) ? dF_next[((int64_t)b * T + t) * NCH + c] : 0.f;
        }
        for (int i = tid; i < NCH * KW; i += 256) cws[i] = conv_w[i];
    }
    __syncthreads();
    const float *hb = hbatch + (int64_t)b * T * E;
    const int E4 = E / 4;
    if (vec && E4 <= 192) {
        // 4 frames per round: all their hbatch loads (<= 12 x 16 B per lane) are in flight
        // before the first dot product -- one memory round trip per round, not one per frame
        float4 dv[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e4 = lane + 64 * r;
            dv[r] = e4 < E4 ? *reinterpret_cast<const float4 *>(dgs + e4 * 4) : zero4();
        }
        for (int r0 = 0; r0 < DA_TT / 4; r0 += 4) {
            if (t0 + wave + 4 * r0 >= len) break;   // wave-uniform
            float4 hv[4][3];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int t = t0 + wave + 4 * (r0 + uu);
                const float *hr = hb + (int64_t)min(t, len - 1) * E;   // clamped: always a valid row
#pragma unroll
                for (int r = 0; r < 3; ++r) hv[uu][r] = ld4(hr + min(lane + 64 * r, E4 - 1) * 4);
            }
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int tt = wave + 4 * (r0 + uu);
                const int t = t0 + tt;
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r)   // dv is zero where lane + 64 r is past the row
                    acc += hv[uu][r].x * dv[r].x + hv[uu][r].y * dv[r].y + hv[uu][r].z * dv[r].z +
                           hv[uu][r].w * dv[r].w;
                if (dF_next) {
                    // frame of dF touched by tap k: t + PADW - k  ->  window row tt + (KW-1) - k
                    for (int k = lane; k < KW; k += 64) {
                        const float *fr = dFw + (tt + (KW - 1) - k) * 11;
#pragma unroll
                        for (int c = 0; c < NCH; ++c) acc += fr[c] * cws[c * KW + k];
                    }
                }
                acc = wave_sum(acc);
                if (lane == 0 && t < len) dalpha[(int64_t)b * T + t] = acc;
            }
        }
        return;
    }
    for (int tt = wave; tt < DA_TT; tt += 4) {
        const int t = t0 + tt;
        if (t >= len) break;  // wave-uniform
        float acc = 0.f;
        if (vec) {
            const float *hr = hb + (int64_t)t * E;
            for (int e4 = lane; e4 < E4; e4 += 64) {
                const float4 h = ld4(hr + e4 * 4);
                const float4 d = *reinterpret_cast<const float4 *>(dgs + e4 * 4);
                acc += h.x * d.x + h.y * d.y + h.z * d.z + h.w * d.w;
            }
        } else {
            for (int e = lane; e < E; e += 64) acc += hb[(int64_t)t * E + e] * dgs[e];
        }
        if (dF_next) {
            // frame of dF touched by tap k: t + PADW - k  ->  window row tt + (KW-1) - k
            for (int k = lane; k < KW; k += 64) {
                const float *fr = dFw + (tt + (KW - 1) - k) * 11;
#pragma unroll
                for (int c = 0; c < NCH; ++c) acc += fr[c] * cws[c * KW + k];
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) dalpha[(int64_t)b * T + t] = acc;
    }
}

static size_t dalpha_smem(int E) {
    return sizeof(float) * (size_t)(((E + 3) & ~3) + (DA_TT + KW - 1) * 11 + 1 + NCH * KW);
}

// grid (ceil(T/TT), B), block EW*64 (lane = frame)
__global__ __launch_bounds__(EW * 64) void energy_bwd_kernel(
    const float *__restrict__ PH, const float *__restrict__ alpha_s, const float *__restrict__ dalpha,
    const float *__restrict__ sp /* (B,A) */, const float *__restrict__ w_ee, const float *__restrict__ W_fe,
    const float *__restrict__ Fc_s, const int32_t *__restrict__ lengths, float *__restrict__ de_s,
    float *__restrict__ dF_s, float *__restrict__ dsp_part /* [tile][B][A] */, int B, int T, int A,
    int vec) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const int len = min(lengths[b], T);
    if (t0 >= len) return;
    const int nrows = min(TT, len - t0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *uni = smem;
    float *tile = uni + round4(A * 12);
    float *red = tile + round4(TILE_FLOATS);   // [EW][64][NCH] (also used for the dot reduction)
    const int tid = threadIdx.x;
    const float *PHb = PH + (int64_t)b * T * A;
    float4 pf[4];
    unsigned pm[4];
    wstage_load(pf, pm, PHb, t0, nrows, wave * WCOLS, A, vec != 0);   // in flight under the prologue
    float *wt = tile + wave * (TT * WSTR);

    // <alpha, dalpha> over the valid frames of this utterance
    float part = 0.f;
    for (int t = tid; t < len; t += EW * 64) part += alpha_s[(int64_t)b * T + t] * dalpha[(int64_t)b * T + t];
    part = wave_sum(part);
    if (lane == 0) red[wave] = part;
    fill_uni(uni, sp + (int64_t)b * A, w_ee, W_fe, A);
    __syncthreads();
    float dot = 0.f;
#pragma unroll
    for (int w = 0; w < EW; ++w) dot += red[w];
    const int t = t0 + lane;
    float de = 0.f;
    float F[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) F[c] = 0.f;
    if (lane < nrows) {
        const float al = alpha_s[(int64_t)b * T + t];
        de = al * (dalpha[(int64_t)b * T + t] - dot);
        const float *fp = Fc_s + ((int64_t)b * T + t) * NCH;
#pragma unroll
        for (int c = 0; c < NCH; ++c) F[c] = fp[c];
        if (wave == 0) de_s[(int64_t)b * T + t] = de;
    }
    float dFa[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) dFa[c] = 0.f;

    float *dsp_out = dsp_part + ((int64_t)blockIdx.x * B + b) * A;
    for (int ac = 0; ac < A; ac += ACH) {
        __builtin_amdgcn_wave_barrier();
        wstage_write(wt, pf, pm);
        __builtin_amdgcn_wave_barrier();
        if (ac + ACH < A) wstage_load(pf, pm, PHb, t0, nrows, ac + ACH + wave * WCOLS, A, vec != 0);
        const int abase = ac + wave * 16;
#pragma unroll 4
        for (int aa = 0; aa < 16; ++aa) {
            const int a = abase + aa;
            if (a < A) {  // wave-uniform
                const float *u = uni + a * 12;
                float x = wt[lane * WSTR + aa] + u[0];
#pragma unroll
                for (int c = 0; c < NCH; ++c) x += F[c] * u[2 + c];
                const float th = tanh_f(x);
                const float dpre = de * u[1] * (1.f - th * th);
#pragma unroll
                for (int c = 0; c < NCH; ++c) dFa[c] += dpre * u[2 + c];
                const float ssum = wave_sum(dpre);
                if (lane == 0) dsp_out[a] = ssum;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) red[(wave * 64 + lane) * NCH + c] = dFa[c];
    __syncthreads();
    // dF[s][b][t0 .. t0+nrows)[c]: contiguous
    for (int i = tid; i < nrows * NCH; i += EW * 64) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < EW; ++w) v += red[w * 64 * NCH + i];
        dF_s[((int64_t)b * T + t0) * NCH + i] = v;
    }
}

static size_t energy_bwd_smem(int A) {
    return sizeof(float) * (size_t)(((A * 12 + 3) & ~3) + ((TILE_FLOATS + 3) & ~3) + EW * 64 * NCH);
}

// ------------------------------------------------- after the loop: dPH etc --
// grid (ceil(T/64), B, ceil(A/P2_A)), block 256 (4 waves; lane = frame).  Each
// wave owns 8 attention columns at a time and walks all L steps with them in
// registers: dPH[t][a] = sum_s de_s[t] w_ee[a] (1 - tanh^2(...)), and the
// per-(tile,b) partial sums of dW_fe[a][c] and dw_ee[a].
#define P2_A 128
#define P2_Q 8
__global__ __launch_bounds__(256) void phase2_kernel(
    const float *__restrict__ PH, const float *__restrict__ sproj /* (L,B,A) */,
    const float *__restrict__ de /* (L,B,T) */, const float *__restrict__ Fc /* (L,B,T,NCH) */,
    const float *__restrict__ w_ee, const float *__restrict__ W_fe, const int32_t *__restrict__ lengths,
    float *__restrict__ dPH, float *__restrict__ part /* [tile*B + b][A][11] */, int L, int B, int T, int A) {
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    const int len = min(lengths[b], T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = t0 + lane;
    const bool tv = t < len;
    const bool trow = t < T;
    float *pout = part + ((int64_t)blockIdx.x * B + b) * A * 11;
    const int a_lo = blockIdx.z * P2_A + wave * (P2_A / 4);
    for (int aq = 0; aq < P2_A / 4; aq += P2_Q) {
        const int a0 = a_lo + aq;
        if (a0 >= A) break;  // wave-uniform
        float ph[P2_Q], wee[P2_Q], wfe[P2_Q][NCH];
        float acc_ph[P2_Q], acc_ee[P2_Q], acc_fe[P2_Q][NCH];
#pragma unroll
        for (int q = 0; q < P2_Q; ++q) {
            const int a = a0 + q;
            const bool av = a < A;
            ph[q] = (tv && av) ? PH[((int64_t)b * T + t) * A + a] : 0.f;
            wee[q] = av ? w_ee[a] : 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                wfe[q][c] = av ? W_fe[a * NCH + c] : 0.f;
                acc_fe[q][c] = 0.f;
            }
            acc_ph[q] = 0.f;
            acc_ee[q] = 0.f;
        }
        if (t0 < len) {  // block-uniform
            for (int s = 0; s < L; ++s) {
                float dev = 0.f, F[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) F[c] = 0.f;
                if (tv) {
                    dev = de[((int64_t)s * B + b) * T + t];
                    const float *fp = Fc + (((int64_t)s * B + b) * T + t) * NCH;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) F[c] = fp[c];
                }
                const float *sp = sproj + ((int64_t)s * B + b) * A;
#pragma unroll
                for (int q = 0; q < P2_Q; ++q) {
                    const int a = a0 + q;
                    float x = ph[q] + (a < A ? sp[a] : 0.f);
#pragma unroll
                    for (int c = 0; c < NCH; ++c) x += F[c] * wfe[q][c];
                    const float th = tanh_f(x);
                    const float dth = dev * (1.f - th * th);
                    const float dpre = dth * wee[q];
                    acc_ph[q] += dpre;
                    acc_ee[q] += dev * th;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) acc_fe[q][c] += dpre * F[c];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < P2_Q; ++q) {
            const int a = a0 + q;
            if (a < A) {  // wave-uniform
                if (trow) dPH[((int64_t)b * T + t) * A + a] = acc_ph[q];
                const float se = wave_sum(acc_ee[q]);
                float sf[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) sf[c] = wave_sum(acc_fe[q][c]);
                if (lane == 0) {
                    pout[a * 11 + 0] = se;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) pout[a * 11 + 1 + c] = sf[c];
                }
            }
        }
    }
}

// The same pass in 16 x 16 tiles (A % 16 == 0).  grid (ceil(T/64), B, ceil(A/32)), block 256: wave w owns frames
// [16w, 16w+16) of the 64-frame tile and the two 16-column tiles of attention units, all L steps with the
// accumulators in registers; lane = (unit a = lane & 15 of each tile, frames 4q .. 4q+3 with q = lane >> 4).  Per
// step, frame and unit:
//   x      = PH + sproj + sum_c F[f][c] W_fe[a][c]
//   dpre   = de[f] w_ee[a] (1 - tanh^2 x);   dPH += dpre;   dw_ee += de[f] tanh x
//   dW_fe[a][c] += F[f][c] dpre              (the lane's own four frames; summed over q and the waves at the end)
// with the two column tiles paired in packed-fp32 operations (v_pk_fma_f32).  The first versions ran the two
// 10-wide products on v_mfma_f32_16x16x4_f32 (K padded to 12: 3 + 4 instructions per tile and step); measured,
// the matrix pipe's 14 x 32 cycles per wave and step did not hide under the other waves' VALU work, and the
// pass is VALU-issue-bound either way (2 transcendentals + ~7 VALU per element): 96 packed FMAs cost less
// (0.87 -> 0.77 ms with every label step live, tools/dec_time.py).
#define P3_NAT 2
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool vecF>
__global__ __launch_bounds__(256) void phase2_tile_kernel(
    const float *__restrict__ PH, const float *__restrict__ sproj /* (L,B,A) */,
    const float *__restrict__ de /* (L,B,T) */, const float *__restrict__ Fc /* (L,B,T,NCH) */,
    const float *__restrict__ w_ee, const float *__restrict__ W_fe, const int32_t *__restrict__ lengths,
    float *__restrict__ dPH, float *__restrict__ part /* [tile*B + b][A][11] */, int L, int B, int T, int A) {
    static_assert(P3_NAT == 2, "the two column tiles are paired in packed fp32 operations");
    __shared__ float fe_s[4][P3_NAT][NCH][17];
    __shared__ float ee_s[4][P3_NAT][16];
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    const int len = min(lengths[b], T);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, q = lane >> 4;
    const int a_base = blockIdx.z * (16 * P3_NAT);
    const int tf = t0 + 16 * wave;
    float *pout = part + ((int64_t)blockIdx.x * B + b) * A * 11;
    if (t0 >= len) {   // block-uniform: nothing valid in this tile
        for (int i = tid; i < 64 * 16 * P3_NAT; i += 256) {
            const int t = t0 + i / (16 * P3_NAT), a = a_base + i % (16 * P3_NAT);
            if (t < T && a < A) dPH[((int64_t)b * T + t) * A + a] = 0.f;
        }
        for (int i = tid; i < 16 * P3_NAT * 11; i += 256) {
            const int a = a_base + i / 11;
            if (a < A) pout[a * 11 + i % 11] = 0.f;
        }
        return;
    }
    // step-invariant operands, {tile 0, tile 1} pairs; columns past A carry zeros
    const int a0 = a_base + col, a1 = a_base + 16 + col;
    const bool av[P3_NAT] = {a0 < A, a1 < A};
    const f32x2 wee2 = {av[0] ? w_ee[a0] : 0.f, av[1] ? w_ee[a1] : 0.f};
    f32x2 wfe2[NCH], afe[NCH], ph2[4], acc_ph2[4], acc_ee2 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        wfe2[c] = f32x2{av[0] ? W_fe[a0 * NCH + c] : 0.f, av[1] ? W_fe[a1 * NCH + c] : 0.f};
        afe[c] = f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = tf + 4 * q + r;
        ph2[r] = f32x2{(av[0] && t < len) ? PH[((int64_t)b * T + t) * A + a0] : 0.f,
                       (av[1] && t < len) ? PH[((int64_t)b * T + t) * A + a1] : 0.f};
        acc_ph2[r] = f32x2{0.f, 0.f};
    }
    // Per-step inputs (the tile's 64 x 10 conv features, 64 energy gradients, 16*P3_NAT state
    // projections) are staged through LDS once per workgroup -- loading them per lane costs ~13
    // scattered vector-memory instructions per wave and step, which bound the first version.
    // Double-buffered in LDS, two register sets in flight, one barrier per step.  Frames at or past
    // the utterance length are staged as zeros.
    // Every thread has one role for the whole pass (threads 0..159: one 16-byte piece of the F block; 160..223: one
    // energy gradient; 224..255: one state projection) and issues the same three range-checked buffer loads each
    // step, the ones outside its role at an out-of-range offset (they return 0 and move no data).  No branch sits
    // between a load and its use, so the loads of step s+2 stay in flight under the math of step s: with the
    // per-lane branches of the earlier version the compiler drained the vector-memory counter at every join,
    // which exposed the full load latency in every step (0.89 -> 0.69 ms at config 2 inside the step; with the
    // math switched off the staging skeleton alone takes 0.17 ms).
    static_assert(16 * P3_NAT <= 32, "state-projection loaders are threads 224..255");
    // vecF (host: T even and Fc 16-byte aligned): the F block's offset ((s*B+b)*T + t0)*NCH floats is a multiple of 4
    constexpr int P3_OOB = 0x7ffffff0;
    const int nfl = (min(len, t0 + 64) - t0) * NCH;   // valid floats of the tile's [64][NCH] block
    int offF[3];
    bool keepF[4];
    if constexpr (vecF) {
        const int e = tid * 4;
        offF[0] = (tid < 160 && e < nfl) ? e * 4 : P3_OOB;
        offF[1] = offF[2] = P3_OOB;
#pragma unroll
        for (int j = 0; j < 4; ++j) keepF[j] = e + j < nfl;   // an utterance may end inside the 16 bytes
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = tid + 256 * j;
            offF[j] = (e < 640 && e < nfl) ? e * 4 : P3_OOB;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) keepF[j] = true;
    }
    const int offD = (tid >= 160 && tid < 224 && t0 + tid - 160 < len) ? (tid - 160) * 4 : P3_OOB;
    const int offS = (tid >= 224 && tid - 224 < 16 * P3_NAT && a_base + tid - 224 < A) ? (tid - 224) * 4 : P3_OOB;
    // one descriptor per array for the whole pass (host: L*B*T*NCH*4 < 2 GiB); a step moves the scalar offset only
    const int row0 = b * T + t0;                                   // step 0's first frame
    __amdgpu_buffer_rsrc_t dF = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Fc), 0, L * B * T * NCH * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t dD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(de), 0, L * B * T * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t dS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sproj), 0, L * B * A * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < 3; ++j) if (offF[j] != P3_OOB) offF[j] += row0 * NCH * 4;
    const int voffD = offD != P3_OOB ? offD + row0 * 4 : P3_OOB;
    const int voffS = offS != P3_OOB ? offS + (b * A + a_base) * 4 : P3_OOB;
    const int stepF = B * T * NCH * 4, stepD = B * T * 4, stepS = B * A * 4;
    auto fetch = [&](int s, pd_u32x4 &rF, unsigned &rF1, unsigned &rF2, unsigned &rD, unsigned &rS) {
        if constexpr (vecF) {
            rF = __builtin_amdgcn_raw_buffer_load_b128(dF, offF[0], s * stepF, 0);
        } else {
            rF[0] = __builtin_amdgcn_raw_buffer_load_b32(dF, offF[0], s * stepF, 0);
            rF1 = __builtin_amdgcn_raw_buffer_load_b32(dF, offF[1], s * stepF, 0);
            rF2 = __builtin_amdgcn_raw_buffer_load_b32(dF, offF[2], s * stepF, 0);
        }
        rD = __builtin_amdgcn_raw_buffer_load_b32(dD, voffD, s * stepD, 0);
        rS = __builtin_amdgcn_raw_buffer_load_b32(dS, voffS, s * stepS, 0);
    };
    // the F block is staged as it lies in memory ([64][NCH], no row padding)
    __shared__ __attribute__((aligned(16))) float Fs[2][64 * NCH];
    __shared__ __attribute__((aligned(16))) float des[2][64];
    __shared__ float sps[2][32];
    auto stage = [&](int buf, const pd_u32x4 &rF, unsigned rF1, unsigned rF2, unsigned rD, unsigned rS) {
        if constexpr (vecF) {
            if (tid < 160) {
                pd_u32x4 v = rF;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = keepF[j] ? v[j] : 0u;
                *reinterpret_cast<pd_u32x4 *>(&Fs[buf][tid * 4]) = v;
            }
        } else {
            Fs[buf][tid] = __uint_as_float(rF[0]);
            Fs[buf][tid + 256] = __uint_as_float(rF1);
            if (tid < 128) Fs[buf][tid + 512] = __uint_as_float(rF2);
        }
        if (tid >= 160 && tid < 224) des[buf][tid - 160] = __uint_as_float(rD);
        if (tid >= 224) sps[buf][tid - 224] = __uint_as_float(rS);
    };
    // one label step: (nF..) holds step s+1's rows (loaded one step ago), (fF..) is free and takes step s+2's
    auto step = [&](int s, pd_u32x4 &nF, unsigned &nF1, unsigned &nF2, unsigned &nD, unsigned &nS,
                    pd_u32x4 &fF, unsigned &fF1, unsigned &fF2, unsigned &fD, unsigned &fS) {
        const int buf = s & 1;
        // unconditional (the last two steps fetch step L-1 again): a branch here would make the compiler count the
        // outstanding loads for its not-taken side, i.e. wait for these new loads too before the staging below
        fetch(min(s + 2, L - 1), fF, fF1, fF2, fD, fS);
        const float4 dv = *reinterpret_cast<const float4 *>(&des[buf][16 * wave + 4 * q]);
        const float dev[4] = {dv.x, dv.y, dv.z, dv.w};
        // every term of this step carries a factor de[f]: a wave whose 16 frames all have de == 0
        // (label steps past the utterance's text, where no gradient arrives; frames past its
        // length) adds exact zeros -- skip its math (wave-uniform), keep the barrier
        const bool live = __any(dv.x != 0.f || dv.y != 0.f || dv.z != 0.f || dv.w != 0.f);
        if (live) {
        // the lane's four frames (rows 4q .. 4q+3 of the wave's 16) are 40 contiguous floats of the staged block;
        // the 16 lanes of a column group read the same addresses (LDS broadcast)
        const float4 *Fr = reinterpret_cast<const float4 *>(&Fs[buf][(16 * wave + 4 * q) * NCH]);
        float Fv[4 * NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float4 t4 = Fr[i];
            Fv[4 * i] = t4.x; Fv[4 * i + 1] = t4.y; Fv[4 * i + 2] = t4.z; Fv[4 * i + 3] = t4.w;
        }
        const f32x2 spv = {sps[buf][col], sps[buf][16 + col]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x2 x = ph2[r] + spv;
#pragma unroll
            for (int c = 0; c < NCH; ++c) x += Fv[r * NCH + c] * wfe2[c];
            const f32x2 th = {tanh_f(x.x), tanh_f(x.y)};
            const f32x2 dp = dev[r] * (1.f - th * th);       // w_ee[a], a factor of every term of both sums, is applied once at the end
            acc_ph2[r] += dp;
            acc_ee2 += dev[r] * th;
#pragma unroll
            for (int c = 0; c < NCH; ++c) afe[c] += Fv[r * NCH + c] * dp;
        }
        }
        stage(buf ^ 1, nF, nF1, nF2, nD, nS);            // after the last step: a copy nobody reads
        __syncthreads();
    };
    pd_u32x4 F0 = {0u, 0u, 0u, 0u}, F1v = {0u, 0u, 0u, 0u};
    unsigned p1 = 0, p2 = 0, pD = 0, pS = 0, r1 = 0, r2 = 0, rD = 0, rS = 0;
    fetch(0, F0, p1, p2, pD, pS);
    stage(0, F0, p1, p2, pD, pS);
    fetch(min(1, L - 1), F1v, r1, r2, rD, rS);
    __syncthreads();
    for (int s = 0; s < L; s += 2) {
        step(s, F1v, r1, r2, rD, rS, F0, p1, p2, pD, pS);
        if (s + 1 < L) step(s + 1, F0, p1, p2, pD, pS, F1v, r1, r2, rD, rS);
    }
#pragma unroll
    for (int n = 0; n < P3_NAT; ++n) {
        const int a = a_base + 16 * n + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = tf + 4 * q + r;
            if (av[n] && t < T) dPH[((int64_t)b * T + t) * A + a] = acc_ph2[r][n] * wee2[n];
        }
        float e = acc_ee2[n];
        e += __shfl_xor(e, 16, 64);
        e += __shfl_xor(e, 32, 64);
        if (q == 0) ee_s[wave][n][col] = e;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {     // the lane's sum over its four frames -> the wave's sixteen
            float v = afe[c][n];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (q == 0) fe_s[wave][n][c][col] = v * wee2[n];
        }
    }
    __syncthreads();
    for (int i = tid; i < 16 * P3_NAT * 11; i += 256) {
        const int al = i / 11, k = i % 11;          // local column, 0 = w_ee slot, 1 + c = W_fe slot
        const int n = al >> 4, cl = al & 15;
        const int a = a_base + al;
        if (a >= A) continue;
        float v;
        if (k == 0) v = (ee_s[0][n][cl] + ee_s[1][n][cl]) + (ee_s[2][n][cl] + ee_s[3][n][cl]);
        else v = (fe_s[0][n][k - 1][cl] + fe_s[1][n][k - 1][cl]) + (fe_s[2][n][k - 1][cl] + fe_s[3][n][k - 1][cl]);
        pout[a * 11 + k] = v;
    }
}

// split [A][11] sums into dw_ee[A] and dW_fe[A][10]
__global__ __launch_bounds__(256) void split_fe_kernel(const float *__restrict__ sum11, int A,
                                                       float *__restrict__ dw_ee, float *__restrict__ dW_fe) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A * 11) return;
    const int a = i / 11, c = i % 11;
    if (c == 0) dw_ee[a] = sum11[i];
    else dW_fe[a * NCH + c - 1] = sum11[i];
}

// conv-filter gradient: dcw[c][k] = sum_{s>=1,b,t<len} dF[s][b][t][c] alpha[s-1][b][t+k-PADW]
// grid ((L-1)*B), block 128: thread k < KW owns tap k for all channels.
__global__ __launch_bounds__(128) void dconv_kernel(const float *__restrict__ dF, const float *__restrict__ alpha,
                                                    const int32_t *__restrict__ lengths,
                                                    float *__restrict__ part /* [(L-1)*B][NCH*KW] */, int B,
                                                    int T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s = 1 + blockIdx.x / B, b = blockIdx.x % B;
    const int len = min(lengths[b], T);
    float *aw = smem;                   // [T + KW] : alpha_{s-1}[t - PADW], zero padded
    float *dFs = aw + ((T + KW + 3) & ~3);   // [len][NCH] (12-float rows)
    const int tid = threadIdx.x;
    const float *ap = alpha + ((int64_t)(s - 1) * B + b) * T;
    for (int i = tid; i < T + KW; i += 128) {
        const int t = i - PADW;
        aw[i] = (t >= 0 && t < T) ? ap[t] : 0.f;
    }
    const float *fp = dF + ((int64_t)s * B + b) * T * NCH;
    for (int i = tid; i < len * NCH; i += 128) dFs[(i / NCH) * 12 + (i % NCH)] = fp[i];
    __syncthreads();
    if (tid < KW) {
        float acc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = 0.f;
        for (int t = 0; t < len; ++t) {
            const float av = aw[t + tid];
            const float *f = dFs + t * 12;
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] += f[c] * av;
        }
        float *po = part + (int64_t)blockIdx.x * NCH * KW;
#pragma unroll
        for (int c = 0; c < NCH; ++c) po[c * KW + tid] = acc[c];
    }
}

// ==================================================== persistent backward ==
// The L-step BPTT loop (dscell, dg, dalpha, energy_bwd above) in ONE launch, same skeleton as the persistent
// forward: workgroup (b, k) of the grid B x ceil(T/64), 512 threads, one per compute unit, holds in REGISTERS the
// 64-frame tile k of PH[b] (energy backward, C/D layout of the MFMAs) and of hbatch[b] (dalpha = h . dg: lane
// (frame, 1/8 of the columns)), so a step streams nothing from HBM but its own saved alpha / conv-feature tile.
// The M = 32 products are 16 x 16 tile jobs: cell-backward jobs X (16 rows x 16 hidden units) on workgroups
// [0, NX), context-gradient jobs G (16 rows x 16 columns of E, the K = 4 Hd reduction split in two halves whose
// partial results the consumer adds) on the next NG.  Per step s = L-1 .. 0 the chain is
//   dsp_{s+1}            -> X: dS = dsp W_se + [drec_{s+1} W_ss + dyp_{s+1} W_sy], cell backward -> drec_s   (edge R)
//   drec_s               -> G: dg_s = drec_s W_gs + dyp_s W_gy (two partials)                              (edge G)
//   dg_s[b]              -> (b, k): dalpha tile = h . dg + conv^T(dF_{s+1}); <alpha, dalpha> partial         (edge D)
//   8 dot partials       -> (b, k): de, dpre = de w sech^2, dF tile, dsp partial over the tile's frames     (edge P1)
//   dsp partials of b    -> (b, k): dsp_s[b, chunk k]                                                        (edge P2) -> X
// Off the chain: the bracketed term of dS (X gathers drec_s right after publishing it and keeps the product in a
// register until the next step; dyp W_sy and dyp W_gy are hoisted GEMMs Q1, Q2), and conv^T(dF_{s+1}) (computed
// from the neighbours' published dF tiles while the X and G jobs run; a Toeplitz product on the matrix cores:
// out[16 a + n] = sum_v x[192 a + v] kf[v - 12 n] over the flattened (frame, channel) window).
// Hand-offs exactly as in the forward kernel (tagged 16-byte sc1 granules, two parity slots, bounded spins).
struct PBParams {
    asr_attdec_bwd_args a;
    const float *tab;      // padded W_fe / w_ee tables (e2_prep_kernel)
    const float *Q1;       // (L,B,Hd) dyp W_sy
    const float *Q2;       // (L,B,E)  dyp W_gy
    float *xch;            // exchange area: X_dsp | X_drec | X_dgp | X_dot | X_dspp | X_dF
    unsigned *err;
    int NTILE, MTB, NJX, NJG, ECG, ECH, A16, Tp;
    int o_drec, o_dgp, o_dot, o_dspp, o_dF, xch_bytes;
    unsigned long long *stamps;
    int pace[PD_NPACE];           // as PDParams.pace: pauses before the waits, in source order (asr_attdec_pace 8..15)
};
#define PB_XW (163 * 12)          // flattened dF window: frames t0 - 49 .. t0 + 113, 12 floats each
#define PB_KF (1200 + 2 * 180)    // flattened reversed taps, 180 zeros either side
#define PB_NKS 345                // k-steps of the Toeplitz product: (180 + 1200) / 4

static size_t attdec_bwd_persist_smem() {
    return sizeof(float) * (size_t)(PD_AL * 25 + TT * 12 + PD_AL + 4 * TT + (PB_XW + 12) + PB_KF + PD_WAVES * TT +
                                    4 * PD_AL + 2 * TT * 16 + 2 * PD_WAVES * 16 * 17 + 2 * PD_AL + 64);
}

template <bool STAMPS>
__global__ __launch_bounds__(PD_THREADS) void attdec_bwd_persist_kernel(PBParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const asr_attdec_bwd_args &a = p.a;
    const asr_attdec_fwd_args &f = p.a.f;
    const int B = f.B, T = f.T, E = f.E, Hd = f.Hd, L = f.L, A = 2 * Hd;
    const int NTILE = p.NTILE, MTB = p.MTB, ECG = p.ECG, ECH = p.ECH, A16 = p.A16, Tp = p.Tp;
    const int w = blockIdx.x;
    const int b = w / NTILE, k = w - b * NTILE, t0 = k * TT;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int len = min(f.lengths[b], T);
    constexpr float K2L = 2.88539008177792681472f;
    E2Smem m;
    m.sp = smem;                          // [PD_AL][12] w_ee[a] * W_fe[a][c] (columns 10, 11 zero): A operand of the dF product
    m.wfe = m.sp + PD_AL * 12;            // [PD_AL][12] K2L * {W_fe[a][0..9], sproj_s[b][a], 0}: the state projection rides in k-slot 10
    m.wee = m.wfe + PD_AL * 12;           // [PD_AL]
    m.Fs = m.wee + PD_AL;                 // [TT][12] saved conv features of this step
    float *dgs = m.Fs + TT * 12;          // [8 ECG <= PD_AL] dg_s[b]
    float *altile = dgs + PD_AL;          // [TT] alpha_s tile
    float *dcv = altile + TT;             // [TT] conv^T term of dalpha
    float *detile = dcv + TT;             // [TT]
    float *pdot = detile + TT;            // [TT]
    float *xw = pdot + TT;                // [PB_XW + 12]
    float *kfp = xw + PB_XW + 12;         // [PB_KF]
    float *dcvw = kfp + PB_KF;            // [PD_WAVES][TT]
    m.dspl = dcvw + PD_WAVES * TT;        // [4][PD_AL]
    m.dFl = m.dspl + 4 * PD_AL;           // [2][TT][16]
    float (*red16)[16][17] = reinterpret_cast<float (*)[16][17]>(m.dFl + 2 * TT * 16);                    // X, G
    float (*red16b)[16][17] = reinterpret_cast<float (*)[16][17]>(m.dFl + 2 * TT * 16 + PD_WAVES * 16 * 17);  // X2
    float *rbuf = m.dFl + 2 * TT * 16 + 2 * PD_WAVES * 16 * 17;     // [NTILE][ECH]: NTILE * ECH <= 2 PD_AL
    float *sred = rbuf + 2 * PD_AL;       // [32]
    int *s_abort = reinterpret_cast<int *>(sred + 32);
    float *sdot = sred + 34;

    const int64_t BH = (int64_t)B * Hd, BT = (int64_t)B * T, BA = (int64_t)B * A, BE = (int64_t)B * E;
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(p.xch, 0, p.xch_bytes, 0x00020000);

    int *rank_s = reinterpret_cast<int *>(sred + 36);
    {   // rank of this workgroup among all of them by (frames in the tile, index)
        int *keys = reinterpret_cast<int *>(xw);
        const int nwg = gridDim.x;      // <= 256 compute units <= PD_THREADS
        if (tid < nwg) {
            const int bb = tid / NTILE, kk = tid - bb * NTILE;
            keys[tid] = max(0, min(TT, min(f.lengths[bb], T) - kk * TT)) * 4096 + tid;
        }
        __syncthreads();
        const float c = wave_sum((tid < nwg && keys[tid] < keys[w]) ? 1.0f : 0.0f);
        if ((tid & 63) == 0) sred[wave] = c;
        __syncthreads();
        if (tid == 0) {
            float r = 0.f;
            for (int i = 0; i < PD_WAVES; ++i) r += sred[i];
            *rank_s = (int)r;
        }
        __syncthreads();
    }
    for (int i = tid; i < PD_AL * 12; i += PD_THREADS) {
        const int a_ = i / 12;
        const float wf = i < A16 * 12 ? p.tab[i] : 0.f;
        m.wfe[i] = wf * K2L;
        m.sp[i] = a_ < A16 ? wf * p.tab[A16 * 12 + a_] : 0.f;
    }
    for (int i = tid; i < PD_AL; i += PD_THREADS) {
        m.wee[i] = i < A16 ? p.tab[A16 * 12 + i] : 0.f;
        dgs[i] = 0.f;
    }
    for (int i = tid; i < PB_KF; i += PD_THREADS) {
        const int v = i - 180, j = v / 12, c = v - 12 * j;
        kfp[i] = (v >= 0 && v < 1200 && c < NCH) ? f.conv_w[c * KW + (KW - 1 - j)] : 0.f;
    }
    for (int i = tid; i < PB_XW + 12; i += PD_THREADS) xw[i] = 0.f;
    for (int i = tid; i < TT; i += PD_THREADS) dcv[i] = 0.f;
    if (tid == 0) *s_abort = 0;

    const int nrows = max(0, min(TT, len - t0));
    const int nact = (len + TT - 1) / TT;          // tiles of this utterance that hold frames
    const int MT = A16 / 16;
    // ---- energy role: PH tile, pre-scaled
    float4 phr[PD_MTP];
    {
        const int lane = tid & 63, q = lane >> 4, col = lane & 15, tt = wave & 3, part = wave >> 2;
        const int tl = 16 * tt + col;
        const int mt0 = part * PD_MTP, mt1 = min(MT, mt0 + PD_MTP);
        const float *PHrow = f.PH + ((int64_t)b * T + t0 + tl) * A;
#pragma unroll
        for (int j = 0; j < PD_MTP; ++j) {
            const float4 v = e2_load_ph(PHrow, tl < nrows && mt0 + j < mt1, 16 * (mt0 + j) + 4 * q, A, f.PH);
            phr[j] = make_float4(v.x * K2L, v.y * K2L, v.z * K2L, v.w * K2L);
        }
    }
    // ---- dalpha role: hbatch tile, lane (frame tid >> 3, column group tid & 7)
    float4 hreg[PD_HF];
    {
        const int tf = tid >> 3, cgp = tid & 7;
        const float *hb = f.hbatch + ((int64_t)b * T + t0 + tf) * E + cgp * ECG;
#pragma unroll
        for (int i = 0; i < PD_HF; ++i) {
            const bool ok = tf < nrows && 4 * i < ECG && cgp * ECG + 4 * i < E;
            const float4 v = ld4(ok ? hb + 4 * i : f.hbatch);
            const unsigned mk = ok ? 0xffffffffu : 0u;
            hreg[i] = make_float4(__uint_as_float(__float_as_uint(v.x) & mk), __uint_as_float(__float_as_uint(v.y) & mk),
                                  __uint_as_float(__float_as_uint(v.z) & mk), __uint_as_float(__float_as_uint(v.w) & mk));
        }
    }
    // ---- job roles, dealt by load: the workgroups are ranked by the frames their tile holds (ties by index), the X
    // jobs go to the lightest ones (a tile past its utterance's end has no attention work at all), the G jobs to the
    // next, the fullest tiles stay job-free.  Every workgroup computes the same ranking from the lengths.
    const int NX = p.NJX * MTB, NG = p.NJG * 2 * MTB;
    const int jrank = *rank_s;
    const bool is_x = jrank < NX, is_g = !is_x && jrank < NX + NG;
    const int jq = is_x ? jrank : jrank - NX;
    const int gh = is_g ? (jq & 1) : 0;                    // G: half of the reduction
    const int jr = is_g ? (jq >> 1) : jq;
    const int jt = jr / MTB, jb0 = 16 * (jr - jt * MTB);   // 16-wide tile along units (X) / columns of E (G), first row
    const int KH = 2 * Hd;                                  // reduction length of every pass (= A = half of 4 Hd)
    int kqK = (KH + 31) / 32;
    kqK = (kqK + 3) & ~3;
    const int NGR_A = A / 4, NGR_R = Hd;                    // granules per row: dsp (A / 4), drec (4 Hd / 4)
    float dC_reg = 0.f, accW_reg = 0.f;                     // X owner threads (tid < 256): row tid >> 4, unit tid & 15
    __syncthreads();

    // ---- X2 (off the chain). drec_s W_ss + dyp_s W_sy for the next step's dS, kept in a register.  A workgroup
    // whose tile holds no frames runs it right behind its X job, one with attention work at the end of the step:
    // either way it is done long before the next dsp arrives.
    auto do_x2 = [&](int n, int s, int slot, unsigned ep) {
        (void)n;
            const int tid = pd_opaque((int)threadIdx.x);
            const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, ji = lane & 15;
            const int jrow = jb0 + ji, ju = 16 * jt + ji;
            const int kb = (wave * 4 + q) * kqK, ke = min(KH, kb + kqK);
            const int obi = tid >> 4, oej = tid & 15;
            const int ob = jb0 + obi, oj = 16 * jt + oej;
            const bool owner = tid < 256 && ob < B && oj < Hd;
            Tile16Acc acc;
            tile16_init(acc);
            const float *wrow = ju < Hd ? a.W_ssT + (int64_t)ju * 4 * Hd : nullptr;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float4 wv[PD_KF];
#pragma unroll
                for (int fq = 0; fq < PD_KF; ++fq)
                    wv[fq] = t16_load_fast(wrow ? wrow + c * KH : nullptr, kb + 4 * fq, ke, a.W_ssT);
                int off[PD_KF];
#pragma unroll
                for (int fq = 0; fq < PD_KF; ++fq)
                    off[fq] = (jrow < B && kb + 4 * fq < ke)
                                  ? (p.o_drec + slot * NGR_R * 4 * B + (((c * KH + kb) >> 2) + fq) * 4 * B + 4 * jrow) * 4
                                  : PD_OOB;
                float4 av[PD_KF];
                pd_pause(p.pace[0]);
                pd_gather<PD_KF, false>(av, rX, off, ep, PD_OOB, s_abort, p.err);   // published by now or within a spin or two: no probe
#pragma unroll
                for (int fq = 0; fq < PD_KF; ++fq) {
                    acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].x, wv[fq].x, acc.a0, 0, 0, 0);
                    acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].y, wv[fq].y, acc.a1, 0, 0, 0);
                    acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].z, wv[fq].z, acc.a0, 0, 0, 0);
                    acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].w, wv[fq].w, acc.a1, 0, 0, 0);
                }
            }
            tile16_reduce(acc, red16b);
            if (*s_abort) return;   // leaves the lambda; the caller's next barrier-side check leaves the kernel
            if (tid < 256)
                accW_reg = tile16_get(red16b, obi, oej) + (owner ? p.Q1[(int64_t)s * BH + (int64_t)ob * Hd + oj] : 0.f);
    };
    for (int n = 0; n < L; ++n) {
        const int s = L - 1 - n;
        const int slot = n & 1, pslot = slot ^ 1;
        const unsigned ep = (unsigned)((n >> 1) & 1), pep = (unsigned)(((n - 1) >> 1) & 1);
        PD_STAMP(0);
        // ---- X. cell backward of step s for 16 rows x 16 units
        if (is_x) {
            const int tid = pd_opaque((int)threadIdx.x);
            const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, ji = lane & 15;
            const int jrow = jb0 + ji, ju = 16 * jt + ji;
            const int kb = (wave * 4 + q) * kqK, ke = min(KH, kb + kqK);
            const int obi = tid >> 4, oej = tid & 15;
            const int ob = jb0 + obi, oj = 16 * jt + oej;
            const bool owner = tid < 256 && ob < B && oj < Hd;
            float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cn = 0.f, cp = 0.f;
            if (owner) {
                const float *gd = f.gates + (int64_t)s * 4 * BH + (int64_t)ob * 4 * Hd;
                ig = gd[oj]; fg = gd[Hd + oj]; gg = gd[2 * Hd + oj]; og = gd[3 * Hd + oj];
                cn = f.C[(int64_t)(s + 1) * BH + (int64_t)ob * Hd + oj];
                cp = f.C[(int64_t)s * BH + (int64_t)ob * Hd + oj];
            }
            float dS = 0.f;
            if (n > 0) {
                const float *wrow = ju < Hd ? a.W_seT + (int64_t)ju * A : nullptr;
                float4 wv[PD_KF];
#pragma unroll
                for (int fq = 0; fq < PD_KF; ++fq) wv[fq] = t16_load_fast(wrow, kb + 4 * fq, ke, a.W_seT);
                int off[PD_KF];
#pragma unroll
                for (int fq = 0; fq < PD_KF; ++fq)
                    off[fq] = (jrow < B && kb + 4 * fq < ke)
                                  ? (pslot * NGR_A * 4 * B + ((kb >> 2) + fq) * 4 * B + 4 * jrow) * 4 : PD_OOB;
                float4 av[PD_KF];
                pd_pause(p.pace[1]);
                pd_gather<PD_KF>(av, rX, off, pep, off[0], s_abort, p.err);
                PD_STAMP(1);
                Tile16Acc acc;
                tile16_init(acc);
#pragma unroll
                for (int fq = 0; fq < PD_KF; ++fq) {
                    acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].x, wv[fq].x, acc.a0, 0, 0, 0);
                    acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].y, wv[fq].y, acc.a1, 0, 0, 0);
                    acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].z, wv[fq].z, acc.a0, 0, 0, 0);
                    acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].w, wv[fq].w, acc.a1, 0, 0, 0);
                }
                tile16_reduce(acc, red16);
                if (*s_abort) return;
                if (tid < 256) dS = tile16_get(red16, obi, oej) + accW_reg;
            }
            if (tid < 256) {
                float dr[4] = {0.f, 0.f, 0.f, 0.f};
                if (owner) {
                    const float tc = tanh_f(cn);
                    const float dO = dS * tc;
                    const float dc = dS * og * (1.f - tc * tc) + dC_reg;
                    dr[0] = dc * gg * ig * (1.f - ig);
                    dr[1] = dc * cp * fg * (1.f - fg);
                    dr[2] = dc * ig * (1.f - gg * gg);
                    dr[3] = dO * og * (1.f - og);
                    dC_reg = dc * fg;
                    float *dp = a.drec + (int64_t)s * 4 * BH + (int64_t)ob * 4 * Hd;
                    dp[oj] = dr[0]; dp[Hd + oj] = dr[1]; dp[2 * Hd + oj] = dr[2]; dp[3 * Hd + oj] = dr[3];
                }
                // 4 units of a row sit on 4 adjacent lanes: one tagged 16-byte granule per (row, gate, unit quad)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float v0 = dr[g];
                    const float v1 = ASR_DPP(v0, 0x55), v2 = ASR_DPP(v0, 0xAA), v3 = ASR_DPP(v0, 0xFF);
                    if ((oej & 3) == 0 && ob < B && oj < Hd)
                        pd_st4_tag(rX, (p.o_drec + slot * NGR_R * 4 * B + ((g * Hd + oj) >> 2) * 4 * B + 4 * ob) * 4,
                                   make_float4(v0, v1, v2, v3), ep);
                }
            }
            PD_STAMP(2);
        }
        if (is_x && nrows == 0 && n + 1 < L) {
            // not while the G jobs gather the same drec_s: the workgroup first waits (one wave, a granule per tile)
            // until its utterance's <alpha, dalpha> partials exist -- the chain is two hand-offs further by then
            if ((pd_opaque((int)threadIdx.x) >> 6) == 0) {
                const int lane = pd_opaque((int)threadIdx.x) & 63;
                const int off[1] = {lane < nact ? (p.o_dot + ((slot * B + b) * NTILE + lane) * 4) * 4 : PD_OOB};
                float4 dv[1];
                pd_pause(p.pace[2]);
                pd_gather<1, false>(dv, rX, off, ep, PD_OOB, s_abort, p.err);
            }
            __syncthreads();
            if (*s_abort) return;
            do_x2(n, s, slot, ep);
        }
        if (nrows > 0) {
            // ---- CV (off the chain). conv^T of the neighbours' dF_{s+1}; this step's alpha / conv features / sproj
            {
                const int tid = pd_opaque((int)threadIdx.x);
                // all of this step's saved-tensor loads first (one memory round trip), the dF window gather behind them
                const float alv = tid < nrows ? f.alpha[(int64_t)s * BT + (int64_t)b * T + t0 + tid] : 0.f;
                float fsv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = tid + u * PD_THREADS, r = i / 12, c = i - 12 * r;
                    fsv[u] = (i < TT * 12 && r < nrows && c < NCH) ? f.Fc[(((int64_t)s * B + b) * T + t0 + r) * NCH + c] : (c == NCH ? 1.0f : 0.f);
                }
                float4 spv = zero4();
                if (4 * tid < A) spv = ld4(f.sproj + (int64_t)s * BA + (int64_t)b * A + 4 * tid);
                float4 xv[1] = {zero4()};
                if (n > 0 && (tid >> 6) * 64 < 163 * 3) {      // wave-uniform
                    const int r = tid / 3, gq = tid - 3 * r;
                    const int t = t0 - (KW - 1 - PADW) + r;
                    const bool ok = tid < 163 * 3 && t >= 0 && t < len;
                    const int off[1] = {ok ? (p.o_dF + ((pslot * B + b) * Tp + t) * 12 + 4 * gq) * 4 : PD_OOB};
                    pd_pause(p.pace[3]);
                    pd_gather<1, false>(xv, rX, off, pep, PD_OOB, s_abort, p.err);
                }
                if (tid < TT) altile[tid] = alv;
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (tid + u * PD_THREADS < TT * 12) m.Fs[tid + u * PD_THREADS] = fsv[u];
                if (4 * tid < A) {
                    float *d = m.wfe + 4 * tid * 12 + 10;
                    d[0] = spv.x * K2L; d[12] = spv.y * K2L; d[24] = spv.z * K2L; d[36] = spv.w * K2L;
                }
                if (n > 0 && tid < 163 * 3) *reinterpret_cast<float4 *>(xw + 4 * tid) = xv[0];
            }
            __syncthreads();
            if (*s_abort) return;
            if (n > 0) {
                const int tid = pd_opaque((int)threadIdx.x);
                const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, col = lane & 15;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
                const float *xa = xw + 192 * (col & 3) + q;
                const float *kb_ = kfp + 180 - 12 * col + q;
                const int ks0 = wave * 44;
#pragma unroll
                for (int u = 0; u < 44; u += 4) {
                    float av[4], bv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ks = ks0 + u + e;
                        av[e] = ks < PB_NKS ? xa[4 * ks] : 0.f;
                        bv[e] = ks < PB_NKS ? kb_[4 * ks] : 0.f;
                    }
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc3, 0, 0, 0);
                }
                acc0 += acc2;
                acc1 += acc3;
                if (lane < 16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dcvw[wave * TT + 16 * r + lane] = acc0[r] + acc1[r];
                }
                __syncthreads();
                if (tid < TT) {
                    float v = 0.f;
#pragma unroll
                    for (int wv = 0; wv < PD_WAVES; ++wv) v += dcvw[wv * TT + tid];
                    dcv[tid] = v;
                }
            }
            PD_STAMP(5);
        }
        // ---- G. one half of dg_s = drec_s W_gs + dyp_s W_gy for 16 rows x 16 columns
        if (is_g) {
            const int tid = pd_opaque((int)threadIdx.x);
            const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, ji = lane & 15;
            const int jrow = jb0 + ji, je = 16 * jt + ji;
            const int kb = (wave * 4 + q) * kqK, ke = min(KH, kb + kqK);
            const float *wrow = je < E ? a.W_gsT + (int64_t)je * 4 * Hd + gh * KH : nullptr;
            float4 wv[PD_KF];
#pragma unroll
            for (int fq = 0; fq < PD_KF; ++fq) wv[fq] = t16_load_fast(wrow, kb + 4 * fq, ke, a.W_gsT);
            int off[PD_KF];
#pragma unroll
            for (int fq = 0; fq < PD_KF; ++fq)
                off[fq] = (jrow < B && kb + 4 * fq < ke)
                              ? (p.o_drec + slot * NGR_R * 4 * B + (((gh * KH + kb) >> 2) + fq) * 4 * B + 4 * jrow) * 4 : PD_OOB;
            float4 av[PD_KF];
            pd_pause(p.pace[4]);
            pd_gather<PD_KF>(av, rX, off, ep, off[0], s_abort, p.err);
            PD_STAMP(3);
            Tile16Acc acc;
            tile16_init(acc);
#pragma unroll
            for (int fq = 0; fq < PD_KF; ++fq) {
                acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].x, wv[fq].x, acc.a0, 0, 0, 0);
                acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].y, wv[fq].y, acc.a1, 0, 0, 0);
                acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].z, wv[fq].z, acc.a0, 0, 0, 0);
                acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fq].w, wv[fq].w, acc.a1, 0, 0, 0);
            }
            tile16_reduce(acc, red16);
            if (*s_abort) return;
            if (tid < 64) {
                const int bi = tid >> 2, n4 = (tid & 3) * 4;
                const int bb = jb0 + bi, e0 = 16 * jt + n4;
                if (bb < B && e0 < E) {
                    float4 v = make_float4(tile16_get(red16, bi, n4), tile16_get(red16, bi, n4 + 1),
                                           tile16_get(red16, bi, n4 + 2), tile16_get(red16, bi, n4 + 3));
                    if (gh == 0) {
                        const float4 q2 = ld4(p.Q2 + (int64_t)s * BE + (int64_t)bb * E + e0);
                        v.x += q2.x; v.y += q2.y; v.z += q2.z; v.w += q2.w;
                    }
                    pd_st4_tag(rX, (p.o_dgp + ((slot * 2 + gh) * B + bb) * E + e0) * 4, v, ep);
                }
            }
            PD_STAMP(4);
        }
        if (nrows > 0) {
            // ---- DA. dg_s[b] (two partials), dalpha tile, <alpha, dalpha>, de
            {
                const int tid = pd_opaque((int)threadIdx.x);
                const int lane = tid & 63, wave = tid >> 6;
                if (wave * 256 < E) {      // wave-uniform
                    const bool act = 4 * tid < E;
                    const int off[2] = {act ? (p.o_dgp + ((slot * 2 + 0) * B + b) * E + 4 * tid) * 4 : PD_OOB,
                                        act ? (p.o_dgp + ((slot * 2 + 1) * B + b) * E + 4 * tid) * 4 : PD_OOB};
                    float4 v[2];
                    pd_pause(p.pace[5]);
                    pd_gather<2, false>(v, rX, off, ep, PD_OOB, s_abort, p.err);
                    if (act) {
                        const float4 d4 = make_float4(v[0].x + v[1].x, v[0].y + v[1].y, v[0].z + v[1].z, v[0].w + v[1].w);
                        *reinterpret_cast<float4 *>(dgs + 4 * tid) = d4;
                        if (k == 0) *reinterpret_cast<float4 *>(a.dg + (int64_t)s * BE + (int64_t)b * E + 4 * tid) = d4;
                    }
                }
                __syncthreads();
                if (*s_abort) return;
                PD_STAMP(6);
                const int tf = tid >> 3, cgp = tid & 7;
                float acc = 0.f;
                const float *dgp = dgs + cgp * ECG;
#pragma unroll
                for (int i = 0; i < PD_HF; ++i) {
                    const float4 d4 = *reinterpret_cast<const float4 *>(dgp + 4 * i);
                    acc += hreg[i].x * d4.x + hreg[i].y * d4.y + hreg[i].z * d4.z + hreg[i].w * d4.w;
                }
                acc += ASR_DPP(acc, 0xB1);     // quad_perm [1,0,3,2]
                acc += ASR_DPP(acc, 0x4E);     // quad_perm [2,3,0,1]
                acc += ASR_DPP(acc, 0x141);    // row_half_mirror: the other quad of the 8 lanes
                const float da = acc + dcv[tf];
                const float al_ = altile[tf];
                if (cgp == 0) pdot[tf] = tf < nrows ? al_ * da : 0.f;
                __syncthreads();
                if (wave == 0) {
                    const float v = wave_sum(pdot[lane]);
                    if (lane == 0) pd_st4_tag(rX, (p.o_dot + ((slot * B + b) * NTILE + k) * 4) * 4, make_float4(v, 0.f, 0.f, 0.f), ep);
                    const int off[1] = {lane < nact ? (p.o_dot + ((slot * B + b) * NTILE + lane) * 4) * 4 : PD_OOB};
                    float4 dv[1];
                    pd_pause(p.pace[6]);
                    pd_gather<1, false>(dv, rX, off, ep, PD_OOB, s_abort, p.err);
                    const float d = wave_sum(lane < nact ? dv[0].x : 0.f);
                    if (lane == 0) *sdot = d;
                }
                __syncthreads();
                if (*s_abort) return;
                const float de = al_ * (da - *sdot);
                if (cgp == 0) {
                    detile[tf] = tf < nrows ? de : 0.f;
                    if (tf < nrows) a.de[(int64_t)s * BT + (int64_t)b * T + t0 + tf] = de;
                }
                __syncthreads();
            }
            PD_STAMP(7);
            // ---- EB. dpre = de w sech^2(pre): dsp partial over this tile's frames, dF tile
            {
                const int tid = pd_opaque((int)threadIdx.x);
                const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, col = lane & 15, tt = wave & 3, part = wave >> 2;
                const int tl = 16 * tt + col;
                const int mt0 = part * PD_MTP;
                const float de4 = 4.0f * detile[tl];
                float fB[3];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) fB[ks] = m.Fs[tl * 12 + 4 * ks + q];
                struct PBTile { float a0, a1, a2; };
                auto ldt = [&](int mt) {
                    PBTile t;
                    const float *wr = m.wfe + (16 * mt + col) * 12 + q;
                    t.a0 = wr[0]; t.a1 = wr[4]; t.a2 = wr[8];
                    return t;
                };
                // dF^T[c = 4q + r][t = col] / de-less form, partial over a: four independent accumulation chains
                f32x4 dF0 = {0.f, 0.f, 0.f, 0.f}, dF1 = {0.f, 0.f, 0.f, 0.f}, dF2 = {0.f, 0.f, 0.f, 0.f}, dF3 = {0.f, 0.f, 0.f, 0.f};
                PBTile c0 = ldt(mt0), c1 = ldt(mt0 + 1);
                // x = K2L (PH + sproj + F W_fe^T): PH is the accumulator input, sproj k-slot 10 (F[t][10] = 1).
                // Software pipeline: pair j + 1's chains are issued before pair j's transcendental work (see the forward)
                auto xmm = [&](const PBTile &t0_, const PBTile &t1_, int j, f32x4 &x0, f32x4 &x1) {
                    x0 = f32x4{phr[j].x, phr[j].y, phr[j].z, phr[j].w};
                    x1 = f32x4{phr[j + 1].x, phr[j + 1].y, phr[j + 1].z, phr[j + 1].w};
                    x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t0_.a0, fB[0], x0, 0, 0, 0);
                    x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t1_.a0, fB[0], x1, 0, 0, 0);
                    x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t0_.a1, fB[1], x0, 0, 0, 0);
                    x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t1_.a1, fB[1], x1, 0, 0, 0);
                    x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t0_.a2, fB[2], x0, 0, 0, 0);
                    x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t1_.a2, fB[2], x1, 0, 0, 0);
                };
                f32x4 x0, x1;
                xmm(c0, c1, 0, x0, x1);
#pragma unroll
                for (int j = 0; j < PD_MTP; j += 2) {
                    PBTile n0 = c0, n1 = c1;
                    f32x4 y0 = x0, y1 = x1;
                    if (j + 2 < PD_MTP) {
                        n0 = ldt(mt0 + j + 2); n1 = ldt(mt0 + j + 3);
                        xmm(n0, n1, j + 2, y0, y1);
                    }
                    // sech^2 / 4 = E / (1 + E)^2 = r (1 - r), r = 1 / (1 + E), E = 2^(-|x|); the 4 rides in de4 and w_ee[a]
                    // in the dF operand table and in the final dsp sum: two transcendentals + three VALU per element
                    float dp0[4], dp1[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float r0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-fabsf(x0[r])));
                        const float r1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-fabsf(x1[r])));
                        dp0[r] = de4 * __builtin_fmaf(-r0, r0, r0);
                        dp1[r] = de4 * __builtin_fmaf(-r1, r1, r1);
                    }
                    // dF^T[c][t] += sum_a (w_a W_fe[a][c]) dp[a][t]: k-step r pairs a = 16 mt + 4 q + r on both operands
                    const float *wc0 = m.sp + (16 * (mt0 + j) + 4 * q) * 12 + col;
                    const float *wc1 = wc0 + 16 * 12;
                    dF0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc0[0], dp0[0], dF0, 0, 0, 0);
                    dF1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc1[0], dp1[0], dF1, 0, 0, 0);
                    dF2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc0[12], dp0[1], dF2, 0, 0, 0);
                    dF3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc1[12], dp1[1], dF3, 0, 0, 0);
                    dF0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc0[24], dp0[2], dF0, 0, 0, 0);
                    dF1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc1[24], dp1[2], dF1, 0, 0, 0);
                    dF2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc0[36], dp0[3], dF2, 0, 0, 0);
                    dF3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc1[36], dp1[3], dF3, 0, 0, 0);
                    const float t0_ = row16_sum4(dp0[0], dp0[1], dp0[2], dp0[3]);
                    const float t1_ = row16_sum4(dp1[0], dp1[1], dp1[2], dp1[3]);
                    if ((col & 3) == 0) {
                        m.dspl[tt * PD_AL + 16 * (mt0 + j) + 4 * q + (col >> 2)] = t0_;
                        m.dspl[tt * PD_AL + 16 * (mt0 + j + 1) + 4 * q + (col >> 2)] = t1_;
                    }
                    c0 = n0; c1 = n1;
                    x0 = y0; x1 = y1;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) m.dFl[(part * TT + tl) * 16 + 4 * q + r] = (dF0[r] + dF1[r]) + (dF2[r] + dF3[r]);
                __syncthreads();
                // dsp partial of this tile (sum over its 4 x 16 frames), 16-byte tagged granules
                if (4 * tid < A) {
                    const float4 u0 = *reinterpret_cast<const float4 *>(m.dspl + 4 * tid);
                    const float4 u1 = *reinterpret_cast<const float4 *>(m.dspl + PD_AL + 4 * tid);
                    const float4 u2 = *reinterpret_cast<const float4 *>(m.dspl + 2 * PD_AL + 4 * tid);
                    const float4 u3 = *reinterpret_cast<const float4 *>(m.dspl + 3 * PD_AL + 4 * tid);
                    const float4 w4 = *reinterpret_cast<const float4 *>(m.wee + 4 * tid);
                    pd_st4_tag(rX, (p.o_dspp + ((slot * B + b) * NTILE + k) * A16 + 4 * tid) * 4,
                               make_float4(w4.x * ((u0.x + u1.x) + (u2.x + u3.x)), w4.y * ((u0.y + u1.y) + (u2.y + u3.y)),
                                           w4.z * ((u0.z + u1.z) + (u2.z + u3.z)), w4.w * ((u0.w + u1.w) + (u2.w + u3.w))), ep);
                }
                // dF tile: published for the neighbours' conv^T of the next step (12 floats a frame) and saved
                if (tid < 3 * TT) {
                    const int r = tid / 3, gq = tid - 3 * r;
                    if (r < nrows) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int c = 4 * gq + e;
                            v[e] = c < NCH ? m.dFl[r * 16 + c] + m.dFl[(TT + r) * 16 + c] : 0.f;
                        }
                        pd_st4_tag(rX, (p.o_dF + ((slot * B + b) * Tp + t0 + r) * 12 + 4 * gq) * 4,
                                   make_float4(v[0], v[1], v[2], v[3]), ep);
                        float *dFo = a.dF + (((int64_t)s * B + b) * T + t0 + r) * NCH + 4 * gq;
                        dFo[0] = v[0]; dFo[1] = v[1];
                        if (gq < 2) { dFo[2] = v[2]; dFo[3] = v[3]; }
                    }
                }
            }
            PD_STAMP(8);
        }
        // ---- RD. dsp_s[b, chunk k] = sum of the tiles' partials  (dsp_0 multiplies S_0 = 0: not needed)
        if (s > 0) {
            const int tid = pd_opaque((int)threadIdx.x);
            const int ncg = ECH / 4;
            const int kk = tid / ncg, g4 = tid - kk * ncg;
            const int colA = k * ECH + 4 * g4;
            if ((tid >> 6) * 64 < nact * ncg) {     // wave-uniform
                const bool act = kk < nact && colA < A;
                const int off[1] = {act ? (p.o_dspp + ((slot * B + b) * NTILE + kk) * A16 + colA) * 4 : PD_OOB};
                float4 v[1];
                pd_pause(p.pace[7]);
                pd_gather<1, false>(v, rX, off, ep, PD_OOB, s_abort, p.err);
                if (kk < nact) *reinterpret_cast<float4 *>(rbuf + kk * ECH + 4 * g4) = act ? v[0] : zero4();
            }
            __syncthreads();
            if (*s_abort) return;
            if (tid < ncg && k * ECH + 4 * tid < A) {
                float4 d4 = zero4();
                for (int t = 0; t < nact; ++t) {
                    const float4 v = *reinterpret_cast<const float4 *>(rbuf + t * ECH + 4 * tid);
                    d4.x += v.x; d4.y += v.y; d4.z += v.z; d4.w += v.w;
                }
                const int ca = k * ECH + 4 * tid;
                pd_st4_tag(rX, (slot * NGR_A * 4 * B + (ca >> 2) * 4 * B + 4 * b) * 4, d4, ep);
                *reinterpret_cast<float4 *>(a.dsp + (int64_t)s * BA + (int64_t)b * A + ca) = d4;
            }
        }
        if (is_x && nrows > 0 && n + 1 < L) do_x2(n, s, slot, ep);
        PD_STAMP(9);
    }
}

static int64_t bwd_ntile(int T) { return cdiv(T, TT); }

static int64_t attdec_bwd_ws_base_floats(int B, int T, int E, int Hd, int L) {
    (void)E;
    const int64_t A = 2 * Hd, nt = bwd_ntile(T);
    int64_t n = 0;
    n += (int64_t)B * T + 64;                    // dalpha
    n += (int64_t)B * Hd + 64;                   // dC carry
    n += nt * B * A + 64;                        // dsp partials
    n += ((A + 15) & ~15) * 13 + 64;             // padded W_fe / w_ee tables
    int64_t post = nt * B * A * 11 + A * 11 + 64 + asr_colsum_ws_floats((int)(nt * B), (int)(A * 11)) + 64;
    int64_t conv = (int64_t)(L > 1 ? L - 1 : 1) * B * NCH * KW + 64 +
                   asr_colsum_ws_floats((L > 1 ? L - 1 : 1) * B, NCH * KW) + 64;
    n += post > conv ? post : conv;
    return (n + 63) & ~(int64_t)63;
}
// persistent loop kernel: Q1 (L,B,Hd) | Q2 (L,B,E) | exchange area | 64 | error word | 63
static int64_t attdec_bwd_xch_floats(int B, int T, int E, int Hd) {
    const int64_t A = 2 * Hd, A16 = (A + 15) & ~15, nt = bwd_ntile(T);
    return 2 * ((int64_t)A * B + 4 * (int64_t)Hd * B + 2 * (int64_t)B * E + (int64_t)B * nt * 4 + (int64_t)B * nt * A16 +
                (int64_t)B * nt * TT * 12);
}
extern "C" int64_t asr_attdec_bwd_ws_floats(int B, int T, int E, int Hd, int L) {
    return attdec_bwd_ws_base_floats(B, T, E, Hd, L) + (((int64_t)L * B * (Hd + E) + 63) & ~(int64_t)63) +
           attdec_bwd_xch_floats(B, T, E, Hd) + 64 + 64;
}
// Index (in floats) inside asr_attdec_bwd's workspace of the persistent backward kernel's sticky error word
// (non-zero after the call iff an in-launch wait timed out: the gradients are then invalid).
extern "C" int64_t asr_attdec_bwd_err_offset(int B, int T, int E, int Hd, int L) {
    return asr_attdec_bwd_ws_floats(B, T, E, Hd, L) - 64;
}

// Shape test of the one-launch backward (everything but pointer alignment), as attdec_fwd_persist_hf.
static bool attdec_bwd_persist_ok(int B, int T, int E, int Hd, int n_cu) {
    const int A = 2 * Hd, A16p = (A + 15) & ~15, MTB = cdiv(B, 16);
    const int64_t nt = bwd_ntile(T), nwg = (int64_t)B * nt;
    const int NJX = cdiv(Hd, 16), NJG = cdiv(E, 16), ECG = ((cdiv(E, 8) + 3) / 4) * 4, ECH = ((cdiv(A, (int)nt) + 3) / 4) * 4;
    return A % 4 == 0 && Hd % 4 == 0 && energy_bwd2_smem(A) <= 150 * 1024 && A % 16 == 0 && E % 4 == 0 && nwg <= n_cu &&
           (int64_t)(NJX + 2 * NJG) * MTB <= nwg && nwg <= PD_THREADS && A16p <= PD_AL && ECG <= 4 * PD_HF &&
           2 * Hd <= 128 * PD_KF && nt * ECH <= 2 * PD_AL && nt * (ECH / 4) <= PD_THREADS && nt <= 64 &&
           attdec_bwd_xch_floats(B, T, E, Hd) * 4 < (int64_t)0x7fff0000 && attdec_bwd_persist_smem() <= 150 * 1024;
}

// Which form asr_attdec_fwd / asr_attdec_bwd take for a shape on the current device, given 16-byte aligned tensors
// (torch allocations are): bit 0 = the forward runs its L-step loop as one persistent launch, bit 1 = the backward
// does.  0 with asr_attdec_persist(0).  The Python mirror splits a batch that is too large for one launch
// (B x ceil(T/64) workgroups > compute units: BASELINE config 4, T = 800) into equal parts that each fit.
extern "C" int asr_attdec_plan(int B, int T, int E, int Hd, int L) {
    (void)L;
    if (B <= 0 || T <= 0 || E <= 0 || Hd <= 0 || !attdec_persist_on()) return 0;
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return 0;
    return (attdec_fwd_persist_hf(B, T, E, Hd, n_cu) > 0 ? 1 : 0) | (attdec_bwd_persist_ok(B, T, E, Hd, n_cu) ? 2 : 0);
}

extern "C" int asr_attdec_bwd_tail(const asr_attdec_bwd_args *a, void *stream);

extern "C" int asr_attdec_bwd(const asr_attdec_bwd_args *a, void *stream) {
    ASR_CHECK_ARG(a != nullptr, "attdec_bwd: null args");
    const asr_attdec_fwd_args *f = &a->f;
    const int B = f->B, T = f->T, E = f->E, Hd = f->Hd, L = f->L, A = 2 * Hd;
    ASR_CHECK_ARG(B > 0 && T > 0 && E > 0 && Hd > 0 && L > 0, "attdec_bwd: bad shape");
    ASR_CHECK_ARG(f->hbatch && f->PH && f->lengths && f->w_ee && f->W_fe && f->conv_w && f->C && f->alpha &&
                      f->sproj && f->gates && f->Fc,
                  "attdec_bwd: null forward tensor");
    ASR_CHECK_ARG(a->W_seT && a->W_ssT && a->W_syT && a->W_gsT && a->W_gyT && a->dyp && a->drec && a->dsp &&
                      a->dg && a->de && a->dF && a->dPH && a->dw_ee && a->dW_fe && a->dconv_w && a->ws,
                  "attdec_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t esm = energy_bwd_smem(A), dsm = dalpha_smem(E);
    const size_t csm = sizeof(float) * (size_t)(((T + KW + 3) & ~3) + (size_t)T * 12);
    ASR_CHECK_ARG(esm <= 159 * 1024 && dsm <= 159 * 1024 && csm <= 159 * 1024,
                  "attdec_bwd: T=%d / A=%d / E=%d exceed the LDS budget", T, A, E);
    static thread_local bool attr_set = false;
    if (!attr_set) {
        ASR_CHECK_HIP(hipFuncSetAttribute((const void *)energy_bwd_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        ASR_CHECK_HIP(hipFuncSetAttribute((const void *)dalpha_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        ASR_CHECK_HIP(hipFuncSetAttribute((const void *)dconv_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        attr_set = true;
    }
    const int64_t BH = (int64_t)B * Hd, BT = (int64_t)B * T, BA = (int64_t)B * A, BE = (int64_t)B * E;
    const int64_t nt = bwd_ntile(T);
    float *w = a->ws;
    float *dalpha = w;            w += BT + 64;
    float *dC = w;                w += BH + 64;
    float *dsp_part = w;          w += nt * BA + 64;
    const int A16p = (A + 15) & ~15;
    float *tab = w;               w += (int64_t)A16p * 13 + 64;
    float *post = w;
    const int vecPH = al16(f->PH) && A % 4 == 0;
    const size_t esm2 = energy_bwd2_smem(A);
    const bool use_v2 = vecPH && esm2 <= 150 * 1024;
    if (use_v2) {
        static thread_local bool attr2 = false;
        if (!attr2) {
            ASR_CHECK_HIP(hipFuncSetAttribute((const void *)energy_bwd2_kernel,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr2 = true;
        }
    }
    const int vecH = al16(f->hbatch) && E % 4 == 0;
    const int vecT = Hd % 4 == 0 && al16(a->drec) && al16(a->dyp) && al16(dsp_part) && al16(a->W_ssT) &&
                     al16(a->W_syT) && al16(a->W_seT) && al16(a->W_gsT) && al16(a->W_gyT);

    if (use_v2)
        hipLaunchKernelGGL(e2_prep_kernel, dim3(cdiv((int64_t)A16p * 12, 256)), dim3(256), 0, st, f->w_ee, f->W_fe,
                           A, A16p, tab);
    // dscell sums the per-tile dsp partials into LDS (16 rows x A) when that fits beside its tile buffer
    const size_t ds_smem = sizeof(float) * 16 * (size_t)A;
    const int ds_lds = (vecT && A % 4 == 0 && ds_smem <= 44 * 1024) ? 1 : 0;
    bool persistent = false;
    unsigned *err_word = reinterpret_cast<unsigned *>(a->ws + asr_attdec_bwd_err_offset(B, T, E, Hd, L));
    bool inited = false;        // dC, dsp_0, the error word (and the exchange area): one launch, below
    {   // one launch for the whole loop when every workgroup of the (B x frame tile) grid gets a compute unit of its own
        static int n_cu = -1;
        if (n_cu < 0) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
                n_cu = 0;
        }
        PBParams p;
        p.NTILE = (int)nt;
        p.MTB = cdiv(B, 16);
        p.NJX = cdiv(Hd, 16);
        p.NJG = cdiv(E, 16);
        p.ECG = ((cdiv(E, 8) + 3) / 4) * 4;
        p.ECH = ((cdiv(A, (int)nt) + 3) / 4) * 4;
        p.A16 = A16p;
        p.Tp = (int)nt * TT;
        const int64_t nwg = (int64_t)B * nt;
        const int64_t xfl = attdec_bwd_xch_floats(B, T, E, Hd);
        const size_t psm = attdec_bwd_persist_smem();
        if (attdec_persist_on() && use_v2 && vecH && vecT && attdec_bwd_persist_ok(B, T, E, Hd, n_cu) &&
            al16(a->dg) && al16(a->dsp) && al16(a->ws) && al16(f->sproj)) {
            static thread_local bool attr3 = false;
            if (!attr3) {
                ASR_CHECK_HIP(hipFuncSetAttribute((const void *)attdec_bwd_persist_kernel<false>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                ASR_CHECK_HIP(hipFuncSetAttribute((const void *)attdec_bwd_persist_kernel<true>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                attr3 = true;
            }
            float *q1 = a->ws + attdec_bwd_ws_base_floats(B, T, E, Hd, L);
            float *q2 = q1 + (int64_t)L * B * Hd;
            // hoisted: Q1 = dyp W_sy, Q2 = dyp W_gy for all steps (the per-step kernels redo them row tile by row tile)
            asr_gemm_desc d;
            memset(&d, 0, sizeof(d));
            d.A = a->dyp; d.M = L * B; d.K = Hd; d.a_s0 = Hd;
            d.alpha = 1.0f; d.splitk = 1; d.nbatch = 1;
            d.B = a->W_syT; d.N = Hd; d.b_s0 = Hd; d.C = q1; d.c_s0 = Hd;
            ASR_CHECK(asr_sgemm(&d, stream));
            d.B = a->W_gyT; d.N = E; d.b_s0 = Hd; d.C = q2; d.c_s0 = E;
            ASR_CHECK(asr_sgemm(&d, stream));
            p.a = *a;
            p.tab = tab;
            p.Q1 = q1;
            p.Q2 = q2;
            p.xch = q1 + (((int64_t)L * B * (Hd + E) + 63) & ~(int64_t)63);
            p.o_drec = 2 * A * B;
            p.o_dgp = p.o_drec + 8 * Hd * B;
            p.o_dot = p.o_dgp + 4 * B * E;
            p.o_dspp = p.o_dot + 8 * B * (int)nt;
            p.o_dF = p.o_dspp + 2 * B * (int)nt * A16p;
            p.xch_bytes = (int)(xfl * 4);
            p.err = reinterpret_cast<unsigned *>(a->ws + asr_attdec_bwd_err_offset(B, T, E, Hd, L));
            p.stamps = g_attdec_stamps;
            for (int i = 0; i < PD_NPACE; ++i) p.pace[i] = g_pd_pace[PD_NPACE + i];
            attdec_init(dC, BH, a->dsp, BA, nullptr, 0, err_word, p.xch, xfl, st);
            inited = true;
            if (p.stamps) hipLaunchKernelGGL(attdec_bwd_persist_kernel<true>, dim3((unsigned)nwg), dim3(PD_THREADS), psm, st, p);
            else hipLaunchKernelGGL(attdec_bwd_persist_kernel<false>, dim3((unsigned)nwg), dim3(PD_THREADS), psm, st, p);
            ASR_CHECK_LAUNCH("attdec_bwd_persist_kernel");
            persistent = true;
        }
    }
    if (!inited) attdec_init(dC, BH, a->dsp, BA, nullptr, 0, err_word, nullptr, 0, st);
    for (int s = persistent ? -1 : L - 1; s >= 0; --s) {
        const bool has_next = s + 1 < L;
        hipLaunchKernelGGL(dscell_kernel, dim3(cdiv(Hd, 16), cdiv(B, 16)), dim3(1024), ds_lds ? ds_smem : 0, st, s, L,
                           has_next ? a->drec + (s + 1) * 4 * BH : (const float *)nullptr,
                           has_next ? a->dyp + (s + 1) * BH : (const float *)nullptr,
                           (const float *)dsp_part, (int64_t)BA,
                           has_next ? a->dsp + (s + 1) * BA : (float *)nullptr, a->W_ssT, a->W_syT, a->W_seT,
                           f->gates + s * 4 * BH, f->C + s * BH, f->C + (s + 1) * BH, f->lengths, dC,
                           a->drec + s * 4 * BH, B, Hd, T, vecT, ds_lds);
        hipLaunchKernelGGL(dg_kernel, dim3(cdiv(E, 16), cdiv(B, 16)), dim3(1024), 0, st,
                           (const float *)(a->drec + s * 4 * BH), a->dyp + s * BH, a->W_gsT, a->W_gyT,
                           a->dg + s * BE, B, Hd, E, vecT);
        hipLaunchKernelGGL(dalpha_kernel, dim3(cdiv(T, DA_TT), B), dim3(256), dsm, st, f->hbatch,
                           (const float *)(a->dg + s * BE),
                           has_next ? (const float *)(a->dF + (s + 1) * BT * NCH) : (const float *)nullptr,
                           f->conv_w, f->lengths, dalpha, T, E, vecH);
        if (use_v2) {
            hipLaunchKernelGGL(energy_bwd2_kernel, dim3(cdiv(T, TT), B), dim3(E2W * 64), esm2, st, f->PH,
                               (const float *)(f->alpha + s * BT), (const float *)dalpha,
                               (const float *)(f->sproj + s * BA), (const float *)tab,
                               (const float *)(f->Fc + s * BT * NCH), f->lengths, a->de + s * BT,
                               a->dF + s * BT * NCH, dsp_part, B, T, A);
        } else {
            hipLaunchKernelGGL(energy_bwd_kernel, dim3(cdiv(T, TT), B), dim3(EW * 64), esm, st, f->PH,
                               (const float *)(f->alpha + s * BT), (const float *)dalpha,
                               (const float *)(f->sproj + s * BA), f->w_ee, f->W_fe,
                               (const float *)(f->Fc + s * BT * NCH), f->lengths, a->de + s * BT,
                               a->dF + s * BT * NCH, dsp_part, B, T, A, vecPH);
        }
    }
    // dsp_0 (it multiplies S_0 = 0 in dW_se = sum_s dsp_s^T S_s, irrelevant to every gradient) is defined as zero:
    // cleared by attdec_init above; neither loop writes step 0's slab.
    // dPH (the critical path: d hbatch needs it) and the per-(tile, utterance) partial sums of dW_fe / dw_ee
    {
        float *part = post;
        const bool p3 = A % 16 == 0 && (int64_t)L * B * T * NCH * 4 < (1ll << 31) && (int64_t)L * B * A * 4 < (1ll << 31);
        if (p3 && (T & 1) == 0 && (((uintptr_t)f->Fc) & 15) == 0)
            hipLaunchKernelGGL(phase2_tile_kernel<true>, dim3(nt, B, cdiv(A, 16 * P3_NAT)), dim3(256), 0, st, f->PH,
                               (const float *)f->sproj, (const float *)a->de, (const float *)f->Fc, f->w_ee,
                               f->W_fe, f->lengths, a->dPH, part, L, B, T, A);
        else if (p3)
            hipLaunchKernelGGL(phase2_tile_kernel<false>, dim3(nt, B, cdiv(A, 16 * P3_NAT)), dim3(256), 0, st, f->PH,
                               (const float *)f->sproj, (const float *)a->de, (const float *)f->Fc, f->w_ee,
                               f->W_fe, f->lengths, a->dPH, part, L, B, T, A);
        else
            hipLaunchKernelGGL(phase2_kernel, dim3(nt, B, cdiv(A, P2_A)), dim3(256), 0, st, f->PH,
                               (const float *)f->sproj, (const float *)a->de, (const float *)f->Fc, f->w_ee,
                               f->W_fe, f->lengths, a->dPH, part, L, B, T, A);
    }
    ASR_CHECK_LAUNCH("attdec_bwd kernels");
    if (a->defer_tail) return ASR_OK;      // the caller runs asr_attdec_bwd_tail (e.g. on a second stream)
    return asr_attdec_bwd_tail(a, stream);
}

// The rest of asr_attdec_bwd: gradients of the attention's own small weights -- dW_fe, dw_ee (column sums over the
// partials phase 2 left in the workspace) and dconv_w (from the saved dF and alpha).  Nothing in the backward pass
// reads them, so a caller that set defer_tail may run this on another stream once the main call's work is complete
// there (an event), overlapping the encoder's backward; it reads a->ws, a->dF, f.alpha, f.lengths.
extern "C" int asr_attdec_bwd_tail(const asr_attdec_bwd_args *a, void *stream) {
    ASR_CHECK_ARG(a != nullptr, "attdec_bwd_tail: null args");
    const asr_attdec_fwd_args *f = &a->f;
    const int B = f->B, T = f->T, E = f->E, Hd = f->Hd, L = f->L, A = 2 * Hd;
    ASR_CHECK_ARG(B > 0 && T > 0 && E > 0 && Hd > 0 && L > 0, "attdec_bwd_tail: bad shape");
    ASR_CHECK_ARG(f->alpha && f->lengths && a->dF && a->dw_ee && a->dW_fe && a->dconv_w && a->ws, "attdec_bwd_tail: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t csm = sizeof(float) * (size_t)(((T + KW + 3) & ~3) + (size_t)T * 12);
    ASR_CHECK_ARG(csm <= 159 * 1024, "attdec_bwd_tail: T=%d exceeds the LDS budget", T);
    static thread_local bool attr_set = false;
    if (!attr_set) {
        ASR_CHECK_HIP(hipFuncSetAttribute((const void *)dconv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        attr_set = true;
    }
    const int64_t BH = (int64_t)B * Hd, BT = (int64_t)B * T, BA = (int64_t)B * A;
    const int64_t nt = bwd_ntile(T);
    const int A16p = (A + 15) & ~15;
    float *post = a->ws + (BT + 64) + (BH + 64) + (nt * BA + 64) + ((int64_t)A16p * 13 + 64);   // as laid out by asr_attdec_bwd
    // dW_fe, dw_ee
    {
        float *part = post;
        float *sum11 = post + nt * BA * 11;
        // deterministic two-stage column sum over the (tile, b) partials
        ASR_CHECK(asr_colsum(part, (int)(nt * B), A * 11, (int64_t)A * 11, 0, 0, sum11, 0,
                             sum11 + A * 11 + 64, stream));
        hipLaunchKernelGGL(split_fe_kernel, dim3(cdiv((int64_t)A * 11, 256)), dim3(256), 0, st,
                           (const float *)sum11, A, a->dw_ee, a->dW_fe);
    }
    // conv filter gradient (stream order: runs after the column sums have finished with `post`)
    if (L > 1) {
        float *part = post;
        hipLaunchKernelGGL(dconv_kernel, dim3((L - 1) * B), dim3(128), csm, st, (const float *)a->dF,
                           (const float *)f->alpha, f->lengths, part, B, T);
        ASR_CHECK(asr_colsum(part, (L - 1) * B, NCH * KW, (int64_t)NCH * KW, 0, 0, a->dconv_w, 0,
                             part + (int64_t)(L - 1) * B * NCH * KW + 64, stream));
    } else {
        ASR_CHECK_HIP(hipMemsetAsync(a->dconv_w, 0, sizeof(float) * NCH * KW, st));
    }
    ASR_CHECK_LAUNCH("attdec_bwd_tail kernels");
    return ASR_OK;
}
