// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// products, fp32 accumulate -- bit-for-bit an fmaf chain, so the fp32 parity
// budget of the path is untouched).
//
// Used for every hoisted matrix product of the AttModel step: the LSTM input
// projections and their gradients (torch.nn.LSTM inside Models/encoder.py:25-33),
// L_he(hbatch) (Models/attention.py:30, hoisted out of the decoder loop), L_yy
// (Models/decoder.py:55) and every weight gradient.
//
// Tile: BM x BN x 16 per 256-thread block (4 waves as 2x2), each wave owns
// (BM/2)x(BN/2) as 32x32 MFMA tiles.  Both operands are staged k-major in LDS
// (As[k][m], Bs[k][n]) so one ds_read_b32 per lane feeds an MFMA operand
// (lane l holds A[m = l&31][k = l>>5]) without bank conflicts; global loads are
// float4 along whichever index is contiguous in memory, prefetched into
// registers one k-tile ahead of the MFMAs.
#include <stdlib.h>
#include <string.h>

#include "common.h"

#define ASR_GEMM_PRECISION_DEFAULT 1

// LDS row padding: the k-contiguous source (mode 0) is scattered into the k-major tile as
// 4-byte writes, lane -> (row = lane/4, k-chunk = lane%4); a row stride = 2 mod 8 floats puts the
// four k-chunks of a half-wave 8 banks apart (conflict-free); the row-contiguous source (mode 1)
// writes 16-byte vectors and needs a stride that is a multiple of 4 floats.
#define GPAD(MODE) ((MODE) == 0 ? 2 : 4)

struct OperandView {
    const float *p;
    int mode;          // 0: k contiguous, rows indexed by m/n; 1: m/n contiguous, rows indexed by k
    int64_t s0, s1;
    int d;
};

// 32-bit index arithmetic: the two-level map only divides when it is really two-level
__device__ __forceinline__ int64_t view_rowoff(const OperandView &v, int i) {
    if (v.d == 0x7fffffff) return (int64_t)i * v.s0;
    const int hi = i / v.d, lo = i - hi * v.d;
    return (int64_t)hi * v.s1 + (int64_t)lo * v.s0;
}

// Per-thread slots of a (R x GK) operand tile, R/64 float4s each.
//   mode 0: element (r,k) at p[rowoff(r0+r) + k];   slot f -> r = f>>2, k4 = f&3   (row fixed per slot)
//   mode 1: element (r,k) at p[rowoff(k) + r0+r];   slot f -> k = f/(R/4), r4 = f%(R/4)
template <int R, int GK, int MODE, bool FAST, bool KLIST = false>
struct TileLoader {
    static constexpr int NS = R * GK / 1024;   // float4 slots per thread
    static constexpr int KQ = GK / 4;          // float4s per row (mode 0)
    const float *base[NS];   // mode 0: row pointer (or null); mode 1: p + column
    int rk[NS];              // mode 0: k offset inside the tile; mode 1: k row inside the tile
    int rcol[NS];            // mode 1: first column of the float4
    unsigned okmask[NS];     // FAST: all-ones if the slot of the tile in flight is in range
    int kx[NS];              // KLIST: the row-list entry of this slot for the tile loaded next

    __device__ __forceinline__ void apply_mask(float4 (&regs)[NS]) const {
        if (!FAST) return;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const unsigned m = okmask[i];
            regs[i] = make_float4(__uint_as_float(__float_as_uint(regs[i].x) & m),
                                  __uint_as_float(__float_as_uint(regs[i].y) & m),
                                  __uint_as_float(__float_as_uint(regs[i].z) & m),
                                  __uint_as_float(__float_as_uint(regs[i].w) & m));
        }
    }

    // midx (mode 0): list entry r names the operand's row
    __device__ __forceinline__ void init(const OperandView &v, int r0, int rmax, const int32_t *midx) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int f = tid + 256 * i;
            if (MODE == 0) {
                const int r = r0 + f / KQ;
                base[i] = r < rmax ? v.p + view_rowoff(v, midx ? midx[r] : r) : nullptr;
                rk[i] = (f % KQ) * 4;
                rcol[i] = 0;
            } else {
                const int r = r0 + (f % (R / 4)) * 4;
                base[i] = r < rmax ? v.p + r : nullptr;
                rk[i] = f / (R / 4);
                rcol[i] = r;
            }
        }
    }

    // KLIST: the list entries are fetched one k-tile ahead of the rows they name, so the row
    // loads never wait behind an index load
    __device__ __forceinline__ void init_klist(const int32_t *kidx, int k0, int klast) {
        if (!KLIST) return;
#pragma unroll
        for (int i = 0; i < NS; ++i) kx[i] = kidx[min(k0 + rk[i], klast)];
    }

    // FAST: every float4 that starts in bounds lies fully in bounds and is 16-byte
    // aligned (host-checked), so the loads are unconditional (out-of-range slots read a
    // safe address and are masked to zero): no branches, all loads of a tile in flight.
    __device__ __forceinline__ void load(const OperandView &v, int rmax, int k0, int kend, bool vec,
                                         float4 (&regs)[NS], const int32_t *kidx = nullptr, int klast = 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int k = k0 + rk[i];
            const int krow = KLIST ? kx[i] : k;      // mode 1: the operand row that holds index k
            if (FAST) {
                const bool ok = base[i] != nullptr && k < kend;
                const float *src = MODE == 0 ? base[i] + k : base[i] + view_rowoff(v, krow);
                src = ok ? src : v.p;
                regs[i] = ld4(src);      // masked to zero later, in apply_mask (after the MFMAs)
                okmask[i] = ok ? 0xffffffffu : 0u;
                continue;
            }
            float4 out = zero4();
            if (base[i] != nullptr && k < kend) {
                if (MODE == 0) {
                    const float *src = base[i] + k;
                    if (vec && k + 3 < kend) {
                        out = ld4(src);
                    } else {
                        out.x = src[0];
                        if (k + 1 < kend) out.y = src[1];
                        if (k + 2 < kend) out.z = src[2];
                        if (k + 3 < kend) out.w = src[3];
                    }
                } else {
                    const float *src = base[i] + view_rowoff(v, krow);
                    const int r = rcol[i];
                    if (vec && r + 3 < rmax) {
                        out = ld4(src);
                    } else {
                        out.x = src[0];
                        if (r + 1 < rmax) out.y = src[1];
                        if (r + 2 < rmax) out.z = src[2];
                        if (r + 3 < rmax) out.w = src[3];
                    }
                }
            }
            regs[i] = out;
        }
        if (KLIST) {
#pragma unroll
            for (int i = 0; i < NS; ++i) kx[i] = kidx[min(k0 + GK + rk[i], klast)];
        }
    }
};

template <int R, int GK, int MODE>
__device__ __forceinline__ void store_tile(float (*lds)[R + GPAD(MODE)], const float4 (&regs)[R * GK / 1024]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R * GK / 1024; ++i) {
        const int f = tid + 256 * i;
        if (MODE == 0) {
            const int r = f / (GK / 4), k = (f % (GK / 4)) * 4;
            lds[k + 0][r] = regs[i].x;
            lds[k + 1][r] = regs[i].y;
            lds[k + 2][r] = regs[i].z;
            lds[k + 3][r] = regs[i].w;
        } else {
            const int k = f / (R / 4), r = (f % (R / 4)) * 4;
            *reinterpret_cast<float4 *>(&lds[k][r]) = regs[i];
        }
    }
}

// Epilogue shared by the fp32 and the bf16 kernels (the C/D layout of a 32x32 MFMA does not depend on the
// input type): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const asr_gemm_desc &d, float *C, f32x16 (&acc)[TM][TN], int m0, int n0,
                                              int wm, int wn, int WM, int WN, int split) {
    const int lane = threadIdx.x & 63;
    const bool first = (split == 0);
    OperandView vc{nullptr, 0, d.c_s0, d.c_s1, d.c_d};
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int64_t coff[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            coff[r] = row < d.M ? view_rowoff(vc, d.m_idx ? d.m_idx[row] : row) : -1;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + (lane & 31);
            if (col >= d.N) continue;
            float bias = 0.f;
            if (first) {
                const int bh = d.bias_d > 0 ? col / d.bias_d : 0, bl = col - bh * (d.bias_d > 0 ? d.bias_d : 0);
                if (d.bias1) bias += d.bias1[bh * d.bias1_s1 + bl];
                if (d.bias2) bias += d.bias2[bh * d.bias2_s1 + bl];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (coff[r] < 0) continue;
                float *dst = C + coff[r] + col;
                float v = d.alpha * acc[i][j][r] + bias;
                if (d.splitk > 1) {
                    atomicAdd(dst, v);
                } else {
                    if (d.accumulate) v += *dst;
                    if (d.act == 1) v = tanh_f(v);
                    else if (d.act == 2) v = fmaxf(v, 0.f);
                    if (d.c_keep) v = d.c_keep[(dst - d.C)] ? v * d.c_keep_scale : 0.f;   // nbatch == 1
                    *dst = v;
                }
            }
        }
    }
}

template <int BM, int BN, int GK, int AMODE, int BMODE, bool FAST, bool KLIST = false>
__global__ __launch_bounds__(256) void sgemm_kernel(asr_gemm_desc d, int vecA, int vecB, int ktiles_per_split) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) float As[2][GK][BM + GPAD(AMODE)];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][BN + GPAD(BMODE)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int batch = blockIdx.z / d.splitk, split = blockIdx.z % d.splitk;

    // two-level batch index (batch_d > 0): (batch / batch_d, batch % batch_d) -- e.g. (utterance, head)
    const int bhi = d.batch_d > 0 ? batch / d.batch_d : 0, blo = batch - bhi * (d.batch_d > 0 ? d.batch_d : 0);
    OperandView va{d.A + (int64_t)blo * d.batch_a + (int64_t)bhi * d.batch_a1, d.a_mode, d.a_s0, d.a_s1, d.a_d};
    OperandView vb{d.B + (int64_t)blo * d.batch_b + (int64_t)bhi * d.batch_b1, d.b_mode, d.b_s0, d.b_s1, d.b_d};
    float *C = d.C + (int64_t)blo * d.batch_c + (int64_t)bhi * d.batch_c1;

    const int kbeg = split * ktiles_per_split * GK;
    const int kend = min(d.K, kbeg + ktiles_per_split * GK);
    const int nk = kend > kbeg ? (kend - kbeg + GK - 1) / GK : 0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[BM * GK / 1024], rb[BN * GK / 1024];
    static_assert(!KLIST || (AMODE == 1 && BMODE == 1), "a k row list indexes the rows of both operands");
    TileLoader<BM, GK, AMODE, FAST, KLIST> la;
    TileLoader<BN, GK, BMODE, FAST, KLIST> lb;
    const int32_t *kidx = d.k_idx;
    const int klast = d.K - 1;
    la.init(va, m0, d.M, AMODE == 0 ? d.m_idx : nullptr);
    lb.init(vb, n0, d.N, nullptr);
    if (nk > 0) {
        la.init_klist(kidx, kbeg, klast);
        lb.init_klist(kidx, kbeg, klast);
        la.load(va, d.M, kbeg, kend, vecA, ra, kidx, klast);
        lb.load(vb, d.N, kbeg, kend, vecB, rb, kidx, klast);
        la.apply_mask(ra);
        lb.apply_mask(rb);
        store_tile<BM, GK, AMODE>(As[0], ra);
        store_tile<BN, GK, BMODE>(Bs[0], rb);
    }
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            la.load(va, d.M, kbeg + (kt + 1) * GK, kend, vecA, ra, kidx, klast);
            lb.load(vb, d.N, kbeg + (kt + 1) * GK, kend, vecB, rb, kidx, klast);
        }
        const int kh = lane >> 5, l31 = lane & 31;
        // all LDS operand reads of the k-tile first, then the MFMAs: the reads stream in under the
        // matrix pipe (progressive lgkmcnt waits) instead of one exposed LDS round trip per k-pair
        float a[GK / 2][TM], b[GK / 2][TN];
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[kk][i] = As[cur][2 * kk + kh][wm * WM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[kk][j] = Bs[cur][2 * kk + kh][wn * WN + j * 32 + l31];
        }
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            __builtin_amdgcn_sched_barrier(0);   // keep the wait for the prefetched tile behind the MFMAs
            la.apply_mask(ra);
            lb.apply_mask(rb);
            store_tile<BM, GK, AMODE>(As[cur ^ 1], ra);
            store_tile<BN, GK, BMODE>(Bs[cur ^ 1], rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    gemm_epilogue<TM, TN>(d, C, acc, m0, n0, wm, wn, WM, WN, split);
}


// ===================================================================== bf16 ==
// The same products on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16: 16x the fp32 MFMA rate), fp32
// accumulate.  Every fp32 operand element x is split while it is staged into LDS,
//     hi = bf16(x),  lo = bf16(x - hi)            (round to nearest even both times),
// and NPROD = 3 products are accumulated per k-step:  a_lo b_hi + a_hi b_lo + a_hi b_hi  (the bf16x3 split:
// what is dropped -- a_lo b_lo and the residuals past 16 mantissa bits -- is below 2^-16 of |a||b| per
// product; measured against the fp32 MFMA kernel in tests/test_gpu_parity.py::test_sgemm_bf16x3).  NPROD = 1 keeps
// a_hi b_hi only: plain bf16 inputs (8 mantissa bits), an opt-in mode with a tolerance of its own.
// GK = 32 (two k-steps of 16 per tile).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));
#define BF_GK 32

__device__ __forceinline__ unsigned bf_pack(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};      // v_cvt_pk_bf16_f32 (RNE; NaN stays NaN)
    return __builtin_bit_cast(unsigned, v);
}
// hi word and lo word of a pair of fp32 values
__device__ __forceinline__ void bf_split(float a, float b, unsigned &hi, unsigned &lo) {
    hi = bf_pack(a, b);
    lo = bf_pack(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

// LDS images, laid out for the 256 B/clk read forms (MI355X_MICROARCH.md, LDS: ds_read_b32 gets a quarter of that at
// one or two waves per SIMD; a first version that fetched every fragment with four of them spent 2/3 of its time there):
//   mode 0 operand (k contiguous in memory): image [row][40 bf16] (32 k + 16 B of padding: a 16-lane group of a
//     ds_read_b128 then covers all 16 bank quads); a lane's fragment of a 32x32x16 MFMA is ONE ds_read_b128;
//     staged with ds_write_b64 (4 consecutive k of a row per float4).
//   mode 1 operand (rows contiguous in memory): image [k][R + 32 bf16] (row pitch = 64 B mod 256 B: the 4 k rows of a
//     transposed block sit on disjoint banks); staged with ds_write_b64 (4 consecutive rows at one k), fragment =
//     two ds_read_b64_tr_b16 (4 k x 16 rows per 16-lane group, delivered k-major).
typedef short s16x4g __attribute__((ext_vector_type(4)));
typedef short s16x8g __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2g __attribute__((ext_vector_type(2)));
#define BF2_ROW0 80                                   // bytes per row of a mode-0 image
#define BF2_PITCH1(R) (2 * (R) + 64)                  // bytes per k row of a mode-1 image
#define BF2_IMG(R, MODE) ((MODE) == 0 ? (R) * BF2_ROW0 : BF_GK * BF2_PITCH1(R))

// fragment (row = r0 + lane&31, k = 16 ks + 8 (lane>>5) .. +7) of one image
template <int R, int MODE>
__device__ __forceinline__ bf16x8 frag_bf2(const unsigned char *img, int r0, int ks, int lane) {
    if (MODE == 0) {
        const u32x4g v = *reinterpret_cast<const u32x4g *>(img + (r0 + (lane & 31)) * BF2_ROW0 + (ks * 2 + (lane >> 5)) * 16);
        return __builtin_bit_cast(bf16x8, v);
    } else {
        const int q = (lane & 15) >> 2, p = lane & 3;
        const unsigned char *a = img + (16 * ks + 8 * (lane >> 5) + q) * BF2_PITCH1(R) + (r0 + 16 * ((lane >> 4) & 1) + 4 * p) * 2;
        typedef __attribute__((address_space(3))) s16x4g *lds_p;
        const s16x4g v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
        const s16x4g v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * BF2_PITCH1(R)));
        const s16x8g v = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        return __builtin_bit_cast(bf16x8, v);
    }
}

// ------------------------------------------------- software-pipelined main loop --
// The main loop keeps TWO k-tiles of global loads in flight and does the split + LDS staging of tile kt+1 in the shadow of tile kt's MFMAs, in the SAME wave: after every (i, j) group of
// MFMAs comes one slot's worth of split work (an MFMA holds the vector issue port for 8 of its 32 cycles; the rest is
// free for independent instructions).  With the phases back to back instead (loads, MFMAs, wait, split, barrier) the
// two workgroups of a CU fell into step and every phase ran alone: 27 % of the kernel's time had the matrix pipe busy.
// Epilogue with 16-byte stores: a 32x32 accumulator tile (lane = column, 4 consecutive rows per register quad) goes
// through a wave-private 4 KB LDS image [row][32] (conflict-free both ways: ds_write_b32 by column, ds_read_b128 by
// row quads) and leaves as float4 rows, 8 rows x 128 B per store instruction instead of 2 x 128 B, a quarter of the
// store instructions and address arithmetic.  Needs split-K off, N % 4 == 0 and 16-byte aligned C rows (host-checked).
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_v4(const asr_gemm_desc &d, float *C, f32x16 (&acc)[TM][TN], int m0, int n0,
                                                 int wm, int wn, int WM, int WN, float *scratch) {
    const int lane = threadIdx.x & 63;
    float *S = scratch + (threadIdx.x >> 6) * 1024;
    OperandView vc{nullptr, 0, d.c_s0, d.c_s1, d.c_d};
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int64_t coff[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = m0 + wm * WM + i * 32 + p * 8 + rr;
            coff[p] = row < d.M ? view_rowoff(vc, d.m_idx ? d.m_idx[row] : row) : -1;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + c4;
            float bias[4] = {0.f, 0.f, 0.f, 0.f};
            if (col < d.N && (d.bias1 || d.bias2)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cc = col + e;
                    const int bh = d.bias_d > 0 ? cc / d.bias_d : 0, bl = cc - bh * (d.bias_d > 0 ? d.bias_d : 0);
                    if (d.bias1) bias[e] += d.bias1[bh * d.bias1_s1 + bl];
                    if (d.bias2) bias[e] += d.bias2[bh * d.bias2_s1 + bl];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the previous tile's row reads are done
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                S[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 v = *reinterpret_cast<const float4 *>(&S[(p * 8 + rr) * 32 + c4]);
                if (coff[p] < 0 || col >= d.N) continue;
                float *dst = C + coff[p] + col;
                v = make_float4(d.alpha * v.x + bias[0], d.alpha * v.y + bias[1], d.alpha * v.z + bias[2], d.alpha * v.w + bias[3]);
                if (d.accumulate) {
                    const float4 o = *reinterpret_cast<const float4 *>(dst);
                    v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
                }
                if (d.act == 1) v = make_float4(tanh_f(v.x), tanh_f(v.y), tanh_f(v.z), tanh_f(v.w));
                else if (d.act == 2) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                if (d.c_keep) {
                    const unsigned km = *reinterpret_cast<const unsigned *>(d.c_keep + (dst - d.C));
                    const float ks = d.c_keep_scale;
                    v = make_float4((km & 0xffu) ? v.x * ks : 0.f, (km & 0xff00u) ? v.y * ks : 0.f,
                                    (km & 0xff0000u) ? v.z * ks : 0.f, (km & 0xff000000u) ? v.w * ks : 0.f);
                }
                *reinterpret_cast<float4 *>(dst) = v;
            }
        }
    }
}

template <int N>
struct IntTag { static constexpr int value = N; };

// PACKED: the operand arrives pre-split, one 32-bit word {bf16 hi : bf16 lo} per element (asr_pack_hilo; the same
// split as bf_split, done once by a producer instead of by every tile that reads the element): two byte permutes
// per pair instead of two conversions, two shifts and two subtractions.
template <int R, int MODE, int NPROD, bool MASK, bool PACKED = false>
__device__ __forceinline__ void store_slot_bf2(unsigned char *hi, unsigned char *lo, float4 v, unsigned okm, int i) {
    const int f = threadIdx.x + 256 * i;
    if (MASK) v = make_float4(__uint_as_float(__float_as_uint(v.x) & okm), __uint_as_float(__float_as_uint(v.y) & okm),
                    __uint_as_float(__float_as_uint(v.z) & okm), __uint_as_float(__float_as_uint(v.w) & okm));
    unsigned h0, l0, h1, l1;
    if (PACKED) {
        const unsigned wx = __float_as_uint(v.x), wy = __float_as_uint(v.y), wz = __float_as_uint(v.z), ww = __float_as_uint(v.w);
        h0 = __builtin_amdgcn_perm(wy, wx, 0x07060302u); l0 = __builtin_amdgcn_perm(wy, wx, 0x05040100u);
        h1 = __builtin_amdgcn_perm(ww, wz, 0x07060302u); l1 = __builtin_amdgcn_perm(ww, wz, 0x05040100u);
    } else {
        bf_split(v.x, v.y, h0, l0);
        bf_split(v.z, v.w, h1, l1);
    }
    const int off = MODE == 0 ? (f >> 3) * BF2_ROW0 + (f & 7) * 8 : (f / (R / 4)) * BF2_PITCH1(R) + (f % (R / 4)) * 8;
    *reinterpret_cast<u32x2g *>(hi + off) = u32x2g{h0, h1};
    if (NPROD > 1) *reinterpret_cast<u32x2g *>(lo + off) = u32x2g{l0, l1};
}

// Operand tile fetched with range-checked buffer loads: per slot ONE 32-bit byte offset fixed at set-up (row / column
// part) plus a wave-uniform offset per k-tile; a slot outside the matrix gets an offset past the buffer and reads
// zeros, so the main loop carries no pointer arithmetic, no selects and no masks.  Needs the operand's readable
// span (asr_gemm_desc.a_span / b_span, < 2 GiB); mode 1 additionally a single-level k stride and no row list.
#define BUF_OOB 0xfffffff0u
template <int R, int GK, int MODE>
struct BufLoader {
    static constexpr int NS = R * GK / 1024;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[NS];
    int rk[NS];
    int kstep;              // bytes per unit of k

    __device__ __forceinline__ void init(const OperandView &v, int64_t span, int r0, int rmax, const int32_t *midx) {
        const int tid = threadIdx.x;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(v.p), 0, (int)span, 0x00020000);
        kstep = MODE == 0 ? 4 : (int)(v.s0 * 4);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int f = tid + 256 * i;
            if (MODE == 0) {
                const int r = r0 + f / (GK / 4);
                rk[i] = (f % (GK / 4)) * 4;
                voff[i] = r < rmax ? (unsigned)((view_rowoff(v, midx ? midx[r] : r) + rk[i]) * 4) : BUF_OOB;
            } else {
                const int c = r0 + (f % (R / 4)) * 4;
                rk[i] = f / (R / 4);
                voff[i] = c < rmax ? (unsigned)(((int64_t)rk[i] * v.s0 + c) * 4) : BUF_OOB;
            }
        }
    }
    __device__ __forceinline__ void load(int k0, int kend, float4 (&regs)[NS]) const {
        const int soff = k0 * kstep;
        if (k0 + GK <= kend) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const u32x4g v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff[i], soff, 0);
                regs[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        } else {            // the k-tile that holds kend: slots past it read zeros
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const unsigned o = k0 + rk[i] < kend ? voff[i] : BUF_OOB;
                const u32x4g v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, soff, 0);
                regs[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        }
    }
};

// PK: bit 0 = the A operand, bit 1 = the B operand arrives as {hi:lo} words (asr_gemm_desc.a_packed / b_packed)
template <int BM, int BN, int AMODE, int BMODE, int NPROD, bool KLIST = false, bool BUFA = false, bool BUFB = false, int PK = 0>
__global__ __launch_bounds__(256, 2) void sgemm_bf3_kernel(asr_gemm_desc d, int ktiles_per_split, int flags) {
    constexpr int GK = BF_GK;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int LO = NPROD > 1 ? 2 : 1;
    constexpr int IA = BF2_IMG(BM, AMODE), IB = BF2_IMG(BN, BMODE);
    constexpr int NSA = BM * GK / 1024, NSB = BN * GK / 1024, NSL = NSA + NSB;
    constexpr int NCH = 2 * TM * TN;                       // MFMA groups per k-tile: (ks, i, j)
    // one pool: the operand images, reused by the epilogue's transposition image (4 KB per wave) after the main loop
    constexpr int IAL = NPROD > 1 ? IA : 16, IBL = NPROD > 1 ? IB : 16;
    constexpr int POOL0 = 2 * IA + 2 * IB + LO * IAL + LO * IBL, POOL = POOL0 > 16384 ? POOL0 : 16384;
    __shared__ __attribute__((aligned(16))) unsigned char pool[POOL];
    unsigned char(*Ah)[IA] = reinterpret_cast<unsigned char(*)[IA]>(pool);
    unsigned char(*Bh)[IB] = reinterpret_cast<unsigned char(*)[IB]>(pool + 2 * IA);
    unsigned char(*Al)[IAL] = reinterpret_cast<unsigned char(*)[IAL]>(pool + 2 * IA + 2 * IB);
    unsigned char(*Bl)[IBL] = reinterpret_cast<unsigned char(*)[IBL]>(pool + 2 * IA + 2 * IB + LO * IAL);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y;
    if (flags & 1) {      // workgroups go to the 8 XCDs round-robin by linear id: XCD x works through one contiguous run of the tile order
        const int nt = gridDim.x * gridDim.y, id = by * gridDim.x + bx;
        const int q = nt >> 3, r = nt & 7, x = id & 7, loc = id >> 3;
        const int tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
        by = tile / gridDim.x;
        bx = tile - by * gridDim.x;
    }
    const int n0 = bx * BN, m0 = by * BM;
    const int batch = blockIdx.z / d.splitk, split = blockIdx.z % d.splitk;
    const int bhi = d.batch_d > 0 ? batch / d.batch_d : 0, blo = batch - bhi * (d.batch_d > 0 ? d.batch_d : 0);
    OperandView va{d.A + (int64_t)blo * d.batch_a + (int64_t)bhi * d.batch_a1, d.a_mode, d.a_s0, d.a_s1, d.a_d};
    OperandView vb{d.B + (int64_t)blo * d.batch_b + (int64_t)bhi * d.batch_b1, d.b_mode, d.b_s0, d.b_s1, d.b_d};
    float *C = d.C + (int64_t)blo * d.batch_c + (int64_t)bhi * d.batch_c1;

    const int kbeg = split * ktiles_per_split * GK;
    const int kend = min(d.K, kbeg + ktiles_per_split * GK);
    const int nk = kend > kbeg ? (kend - kbeg + GK - 1) / GK : 0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[2][NSA], rb[2][NSB];
    TileLoader<BM, GK, AMODE, true, KLIST> la;
    TileLoader<BN, GK, BMODE, true, KLIST> lb;
    BufLoader<BM, GK, AMODE> ba;
    BufLoader<BN, GK, BMODE> bb;
    const int32_t *kidx = d.k_idx;
    const int klast = d.K - 1;
    if (BUFA) ba.init(va, d.a_span - (va.p - d.A) * 4, m0, d.M, AMODE == 0 ? d.m_idx : nullptr);
    else la.init(va, m0, d.M, AMODE == 0 ? d.m_idx : nullptr);
    if (BUFB) bb.init(vb, d.b_span - (vb.p - d.B) * 4, n0, d.N, nullptr);
    else lb.init(vb, n0, d.N, nullptr);
    auto load_tile = [&](auto S, int kt) {     // k-tile kt -> register set S
        constexpr int s = decltype(S)::value;
        if (BUFA) ba.load(kbeg + kt * GK, kend, ra[s]);
        else la.load(va, d.M, kbeg + kt * GK, kend, 1, ra[s], kidx, klast);
        if (BUFB) bb.load(kbeg + kt * GK, kend, rb[s]);
        else lb.load(vb, d.N, kbeg + kt * GK, kend, 1, rb[s], kidx, klast);
    };
    // slot sl of register set S (k-tile kt) -> LDS buffer buf; the in-range mask is recomputed here rather than kept
    // in registers for two tiles in flight
    auto stage_slot = [&](auto S, int buf, int sl, int kt) {
        constexpr int s = decltype(S)::value;
        const int lb_ = NPROD > 1 ? buf : 0;
        const int k0 = kbeg + kt * GK;
        if (sl < NSA) {
            const unsigned ok = (BUFA || (la.base[sl] != nullptr && k0 + la.rk[sl] < kend)) ? 0xffffffffu : 0u;
            store_slot_bf2<BM, AMODE, NPROD, !BUFA, (PK & 1) != 0>(Ah[buf], Al[lb_], ra[s][sl], ok, sl);
        } else {
            const int t = sl - NSA;
            const unsigned ok = (BUFB || (lb.base[t] != nullptr && k0 + lb.rk[t] < kend)) ? 0xffffffffu : 0u;
            store_slot_bf2<BN, BMODE, NPROD, !BUFB, (PK & 2) != 0>(Bh[buf], Bl[lb_], rb[s][t], ok, t);
        }
    };
    if (nk > 0) {
        if (!BUFA) la.init_klist(kidx, kbeg, klast);
        if (!BUFB) lb.init_klist(kidx, kbeg, klast);
        load_tile(IntTag<0>{}, 0);
        if (nk > 1) load_tile(IntTag<1>{}, 1);
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) stage_slot(IntTag<0>{}, 0, sl, 0);
    }
    __syncthreads();

    // one k-tile: LDS buffer P (= kt & 1) feeds the MFMAs, register set P^1 (tile kt+1) is staged into LDS buffer P^1,
    // register set P (free since the previous iteration staged it) receives tile kt+2
    auto body = [&](auto Ptag, int kt) {
        constexpr int P = decltype(Ptag)::value;
        const bool more = kt + 1 < nk;
        if (kt + 2 < nk) load_tile(IntTag<P>{}, kt + 2);
        constexpr int LP = NPROD > 1 ? P : 0;
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ks = c / (TM * TN), i = (c % (TM * TN)) / TN, j = c % TN;
            if (c % (TM * TN) == 0) {
#pragma unroll
                for (int ii = 0; ii < TM; ++ii) {
                    ah[ii] = frag_bf2<BM, AMODE>(Ah[P], wm * WM + ii * 32, ks, lane);
                    if (NPROD > 1) al[ii] = frag_bf2<BM, AMODE>(Al[LP], wm * WM + ii * 32, ks, lane);
                }
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    bh[jj] = frag_bf2<BN, BMODE>(Bh[P], wn * WN + jj * 32, ks, lane);
                    if (NPROD > 1) bl[jj] = frag_bf2<BN, BMODE>(Bl[LP], wn * WN + jj * 32, ks, lane);
                }
            }
            if (NPROD > 1) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            if (more) {
#pragma unroll
                for (int sl = c * NSL / NCH; sl < (c + 1) * NSL / NCH; ++sl) stage_slot(IntTag<P ^ 1>{}, P ^ 1, sl, kt + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        body(IntTag<0>{}, kt);
        if (kt + 1 < nk) body(IntTag<1>{}, kt + 1);
    }
    if (flags & 32) gemm_epilogue_v4<TM, TN>(d, C, acc, m0, n0, wm, wn, WM, WN, reinterpret_cast<float *>(pool));
    else gemm_epilogue<TM, TN>(d, C, acc, m0, n0, wm, wn, WM, WN, split);
}

// 0: exact fp32 products (v_mfma_f32_32x32x2_f32); 1: bf16x3 split products; 2: plain bf16 products.
static int g_gemm_precision = -1;
static int gemm_precision() {
    if (g_gemm_precision < 0) {
        const char *e = getenv("ASR_GEMM_PRECISION");
        int m = ASR_GEMM_PRECISION_DEFAULT;
        if (e) {
            if (!strcmp(e, "f32") || !strcmp(e, "0")) m = 0;
            else if (!strcmp(e, "bf16x3") || !strcmp(e, "1")) m = 1;
            else if (!strcmp(e, "bf16") || !strcmp(e, "2")) m = 2;
        }
        g_gemm_precision = m;
    }
    return g_gemm_precision;
}
extern "C" int asr_gemm_precision(int mode) {
    const int old = gemm_precision();
    if (mode >= 0 && mode <= 2) g_gemm_precision = mode;
    return old;
}

static bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// {hi:lo} words of one or two fp32 arrays (bf_split's arithmetic, element by element)
__global__ __launch_bounds__(256) void pack_hilo_kernel(const float *__restrict__ s0, const float *__restrict__ s1, int64_t n,
                                                        unsigned *__restrict__ dst) {
    const int64_t tot = s1 ? 2 * n : n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
        const float x = i < n ? s0[i] : s1[i - n];
        unsigned hi, lo;
        bf_split(x, 0.f, hi, lo);
        dst[i] = (hi << 16) | (lo & 0xffffu);
    }
}
extern "C" int asr_pack_hilo(const float *src0, const float *src1, int64_t n, float *dst, void *stream) {
    ASR_CHECK_ARG(src0 && dst && n >= 0, "pack_hilo: bad argument");
    if (n == 0) return ASR_OK;
    const int64_t tot = src1 ? 2 * n : n;
    const unsigned blocks = (unsigned)std::min<int64_t>((tot + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_hilo_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src0, src1, n,
                       reinterpret_cast<unsigned *>(dst));
    ASR_CHECK_LAUNCH("pack_hilo_kernel");
    return ASR_OK;
}

// asr_dropout_fwd (misc.hip: same keep decision per element index, same scaled value) whose output leaves as {hi:lo}
// words: the inter-layer dropout of the encoder feeds only matrix products, which then read it pre-split
__global__ __launch_bounds__(256) void dropout_pack_kernel(const float *__restrict__ x, unsigned *__restrict__ y,
                                                           uint8_t *__restrict__ mask, int64_t n, uint32_t thresh,
                                                           float scale, uint64_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const bool keep = asr_keep(seed, i, thresh);
        mask[i] = keep;
        unsigned hi, lo;
        bf_split(keep ? x[i] * scale : 0.f, 0.f, hi, lo);
        y[i] = (hi << 16) | (lo & 0xffffu);
    }
}
extern "C" int asr_dropout_fwd_packed(const float *x, float *y_packed, uint8_t *mask, int64_t n, float p, uint64_t seed,
                                      void *stream) {
    ASR_CHECK_ARG(x && y_packed && mask && n >= 0 && p >= 0.f && p < 1.f, "dropout_fwd_packed: bad argument");
    if (n == 0) return ASR_OK;
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(dropout_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<unsigned *>(y_packed), mask, n, thresh, 1.0f / (1.0f - p), seed);
    ASR_CHECK_LAUNCH("dropout_pack_kernel");
    return ASR_OK;
}

extern "C" int asr_sgemm(const asr_gemm_desc *din, void *stream) {
    ASR_CHECK_ARG(din != nullptr, "sgemm: null descriptor");
    asr_gemm_desc d = *din;
    ASR_CHECK_ARG(d.A && d.B && d.C, "sgemm: null operand");
    ASR_CHECK_ARG(d.M >= 0 && d.N >= 0 && d.K >= 0, "sgemm: negative dimension");
    ASR_CHECK_ARG(d.a_mode == 0 || d.a_mode == 1, "sgemm: bad a_mode %d", d.a_mode);
    ASR_CHECK_ARG(d.b_mode == 0 || d.b_mode == 1, "sgemm: bad b_mode %d", d.b_mode);
    if (d.nbatch <= 0) d.nbatch = 1;
    if (d.splitk <= 0) d.splitk = 1;
    if (d.a_d <= 0) d.a_d = 0x7fffffff;
    if (d.b_d <= 0) d.b_d = 0x7fffffff;
    if (d.c_d <= 0) d.c_d = 0x7fffffff;
    ASR_CHECK_ARG(!(d.splitk > 1 && d.act != 0), "sgemm: activation needs splitk == 1");
    ASR_CHECK_ARG(d.act >= 0 && d.act <= 2, "sgemm: bad act %d", d.act);
    ASR_CHECK_ARG(d.batch_d >= 0, "sgemm: negative batch_d");
    ASR_CHECK_ARG(!d.m_idx || (d.a_mode == 0 && d.nbatch == 1), "sgemm: m_idx needs a_mode 0 and nbatch 1");
    ASR_CHECK_ARG(!d.k_idx || (d.a_mode == 1 && d.b_mode == 1 && d.nbatch == 1),
                  "sgemm: k_idx needs a_mode 1, b_mode 1 and nbatch 1");
    ASR_CHECK_ARG(!d.c_keep || (d.splitk == 1 && d.nbatch == 1), "sgemm: c_keep needs splitk 1 and nbatch 1");
    ASR_CHECK_ARG(d.bias_d >= 0, "sgemm: negative bias_d");
    if (d.M == 0 || d.N == 0) return ASR_OK;
    if (d.K == 0) d.k_idx = nullptr;

    auto strides_ok = [&](int64_t s0, int64_t s1, int64_t sb) {
        return (s0 % 4 == 0) && (s1 % 4 == 0 || s1 == 0) && (sb % 4 == 0);
    };
    const bool two = d.nbatch > 1 && d.batch_d > 0;
    const int vecA = aligned16(d.A) && strides_ok(d.a_s0, d.a_s1, d.nbatch > 1 ? d.batch_a : 0) &&
                     (!two || d.batch_a1 % 4 == 0);
    const int vecB = aligned16(d.B) && strides_ok(d.b_s0, d.b_s1, d.nbatch > 1 ? d.batch_b : 0) &&
                     (!two || d.batch_b1 % 4 == 0);

    const int64_t tiles128 = (int64_t)cdiv(d.M, 128) * cdiv(d.N, 128) * d.nbatch * (d.splitk > 0 ? d.splitk : 1);
    const bool big = d.M >= 128 && d.N >= 128 && tiles128 >= 128;
    static int gk_big = 0;
    if (gk_big == 0) {
        const char *e = getenv("ASR_GEMM_GK");
        gk_big = (e && atoi(e) == 16) ? 16 : 32;   // 32 measured 2-5 % faster on the config-2 shapes
    }
    // fast path: 16-byte aligned operands whose contiguous extent is a multiple of 4
    // (mode 1: a float4 that starts below M may run past it when the caller vouches that the row is
    // readable up to the next multiple of 4 -- those accumulator rows are never stored)
    const bool fastA = vecA && (d.a_mode == 0 ? (d.K % 4 == 0 || (d.pad_ok & 4)) : (d.M % 4 == 0 || (d.pad_ok & 1)));
    const bool fastB = vecB && (d.b_mode == 0 ? d.K % 4 == 0 : (d.N % 4 == 0 || (d.pad_ok & 2)));
    const bool fast = fastA && fastB;
    const int GK = (big && fast) ? gk_big : 16;
    const int ktiles = cdiv(d.K, GK);
    int splitk = d.splitk;
    if (splitk > ktiles) splitk = ktiles > 0 ? ktiles : 1;
    d.splitk = splitk;
    const int per = ktiles > 0 ? cdiv(ktiles, splitk) : 0;

    hipStream_t st = (hipStream_t)stream;
    const int sel = d.a_mode * 2 + d.b_mode;
    // bf16 matrix pipe (precision modes 1 and 2): fast-path shapes only, K worth a tile
    const int prec = gemm_precision();
    if (prec != 0 && fast && d.K >= 32 && d.M >= 32 && d.N >= 32) {
        const int kt = cdiv(d.K, BF_GK);
        int sk = din->splitk > 0 ? din->splitk : 1;
        if (sk > kt) sk = kt;
        d.splitk = sk;
        const int perb = cdiv(kt, sk);
        const int64_t zb = (int64_t)d.nbatch * sk;
        const int64_t t128 = (int64_t)cdiv(d.M, 128) * cdiv(d.N, 128) * zb, t12864 = (int64_t)cdiv(d.M, 128) * cdiv(d.N, 64) * zb;
        static int bf_tile = -1, xcd_map = 1;
        if (bf_tile < 0) {
            const char *e = getenv("ASR_GEMM_BF_TILE");
            bf_tile = e ? atoi(e) : 0;
            const char *x = getenv("ASR_GEMM_XCD_MAP");
            xcd_map = x ? (atoi(x) & 1) : 1;
        }
        // buffer-load operand fetch: spans known and below 2 GiB, k offsets below 2 GiB, mode-1 operands single-level
        const int64_t lim = 0x7ffffff0;
        static int buf_env = -1;
        if (buf_env < 0) { const char *e = getenv("ASR_GEMM_BUF"); buf_env = e ? atoi(e) : 1; }
        const bool buf_a = buf_env && !d.k_idx && d.a_span > 0 && d.a_span <= lim &&
                           (d.a_mode == 0 || (d.a_d == 0x7fffffff && (int64_t)d.K * d.a_s0 * 4 <= lim));
        const bool buf_b = buf_env && !d.k_idx && d.b_span > 0 && d.b_span <= lim &&
                           (d.b_mode == 0 || (d.b_d == 0x7fffffff && (int64_t)d.K * d.b_s0 * 4 <= lim));
        // 16-byte epilogue stores: no split-K atomics, N % 4 == 0, C rows (and the keep mask) aligned
        static int epi_env = -1;
        if (epi_env < 0) { const char *e = getenv("ASR_GEMM_EPI4"); epi_env = e ? atoi(e) : 1; }
        const bool vec_c = epi_env && sk == 1 && d.N % 4 == 0 && aligned16(d.C) && d.c_s0 % 4 == 0 &&
                           (d.c_s1 % 4 == 0) && (d.nbatch == 1 || (d.batch_c % 4 == 0 && (!two || d.batch_c1 % 4 == 0))) &&
                           (!d.c_keep || (((uintptr_t)d.c_keep) & 3) == 0);
        const int kflags = xcd_map | (vec_c ? 32 : 0);
        // pre-split operands: only where a kernel is instantiated for them (128 x 128 tiles, buffer loads on both
        // operands, bf16x3, A k-contiguous); the caller (asr_amd/ops.py) asks asr_sgemm_packed_ok first
        const int pk = (din->a_packed ? 1 : 0) | (din->b_packed ? 2 : 0);
        const bool pk_tiles = prec == 1 && d.M >= 128 && d.N >= 128 && (t128 >= 320 || (sk > 1 && t128 >= 256)) && bf_tile == 0;
        // (i) A k-contiguous, buffer loads on both operands: B packed, or both (then B k-contiguous too);
        // (ii) a weight-gradient product (both operands row-contiguous) whose B operand -- the layer's input -- is
        //      packed: with a k list (pointer loaders) or with buffer loads on both
        ASR_CHECK_ARG(pk == 0 ||
                      (pk_tiles && buf_a && buf_b && d.a_mode == 0 && !d.k_idx && pk != 1 && (pk == 2 || d.b_mode == 0)) ||
                      (pk_tiles && pk == 2 && d.a_mode == 1 && d.b_mode == 1 && (d.k_idx || (buf_a && buf_b))),
                      "sgemm: a_packed / b_packed set for a shape or mode without a pre-split kernel");
        // 128 x 128 tiles from 320 of them on (measured down to 375: tools/gemm_tile_pick.py; at 188 the 64 x 64 tile wins);
        // split-K launches sized for one resident round (asr_amd/ops.py pick_splitk) keep them from 256 on
        int shape = bf_tile ? bf_tile
                            : (d.M >= 128 && d.N >= 128 && (t128 >= 320 || (sk > 1 && t128 >= 256)) ? 1
                                                                                                   : (d.M >= 128 && t12864 >= 384 ? 2 : 3));
#define BF_LAUNCH(BM_, BN_, AM, BM2, NP, KL)                                                                          \
    if (pk == 2 && BM_ == 128 && BN_ == 128 && AM == 0 && NP == 3 && !KL)                                             \
        hipLaunchKernelGGL((sgemm_bf3_kernel<128, 128, 0, BM2, 3, false, true, true, 2>), grid, dim3(256), 0, st, d, perb, kflags); \
    else if (pk == 3 && BM_ == 128 && BN_ == 128 && AM == 0 && BM2 == 0 && NP == 3 && !KL)                            \
        hipLaunchKernelGGL((sgemm_bf3_kernel<128, 128, 0, 0, 3, false, true, true, 3>), grid, dim3(256), 0, st, d, perb, kflags); \
    else if (pk == 2 && BM_ == 128 && BN_ == 128 && AM == 1 && BM2 == 1 && NP == 3 && KL)                             \
        hipLaunchKernelGGL((sgemm_bf3_kernel<128, 128, 1, 1, 3, true, false, false, 2>), grid, dim3(256), 0, st, d, perb, kflags); \
    else if (pk == 2 && BM_ == 128 && BN_ == 128 && AM == 1 && BM2 == 1 && NP == 3 && !KL)                            \
        hipLaunchKernelGGL((sgemm_bf3_kernel<128, 128, 1, 1, 3, false, true, true, 2>), grid, dim3(256), 0, st, d, perb, kflags); \
    else if (buf_a && buf_b && !KL && NP == 3)                                                                        \
        hipLaunchKernelGGL((sgemm_bf3_kernel<BM_, BN_, AM, BM2, NP, false, true, true>), grid, dim3(256), 0, st, d, perb, kflags); \
    else if (buf_a && !KL && NP == 3)                                                                                 \
        hipLaunchKernelGGL((sgemm_bf3_kernel<BM_, BN_, AM, BM2, NP, false, true, false>), grid, dim3(256), 0, st, d, perb, kflags); \
    else hipLaunchKernelGGL((sgemm_bf3_kernel<BM_, BN_, AM, BM2, NP, KL>), grid, dim3(256), 0, st, d, perb, kflags)
#define BF_MODES(BM_, BN_, NP)                                   \
    switch (sel) {                                               \
        case 0: { BF_LAUNCH(BM_, BN_, 0, 0, NP, false); } break;     \
        case 1: { BF_LAUNCH(BM_, BN_, 0, 1, NP, false); } break;     \
        case 2: { BF_LAUNCH(BM_, BN_, 1, 0, NP, false); } break;     \
        default:                                                     \
            if (d.k_idx) { BF_LAUNCH(BM_, BN_, 1, 1, NP, true); }    \
            else { BF_LAUNCH(BM_, BN_, 1, 1, NP, false); }           \
            break;                                                   \
    }
#define BF_SHAPE(BM_, BN_)                                                                          \
    {                                                                                               \
        dim3 grid(cdiv(d.N, BN_), cdiv(d.M, BM_), (unsigned)zb);                                    \
        ASR_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "sgemm: grid too large");                 \
        if (prec == 1) { BF_MODES(BM_, BN_, 3) } else { BF_MODES(BM_, BN_, 1) }                     \
    }
        if (shape == 1) BF_SHAPE(128, 128)
        else if (shape == 2) BF_SHAPE(128, 64)
        else BF_SHAPE(64, 64)
#undef BF_SHAPE
#undef BF_MODES
#undef BF_LAUNCH
        ASR_CHECK_LAUNCH("sgemm_bf3_kernel");
        return ASR_OK;
    }
    ASR_CHECK_ARG(!din->a_packed && !din->b_packed, "sgemm: pre-split operands need the bf16x3 fast path (precision mode, alignment, K >= 32)");
#define LAUNCH(BM_, BN_, GK_, AM, BM2, F)                                                              \
    hipLaunchKernelGGL((sgemm_kernel<BM_, BN_, GK_, AM, BM2, F>), grid, dim3(256), 0, st, d, vecA, vecB, per)
#define LAUNCH_KLIST(BM_, BN_, GK_, F)                                                                 \
    hipLaunchKernelGGL((sgemm_kernel<BM_, BN_, GK_, 1, 1, F, true>), grid, dim3(256), 0, st, d, vecA, vecB, per)
#define LAUNCH_MODES(BM_, BN_, GK_, F)                    \
    switch (sel) {                                        \
        case 0: LAUNCH(BM_, BN_, GK_, 0, 0, F); break;    \
        case 1: LAUNCH(BM_, BN_, GK_, 0, 1, F); break;    \
        case 2: LAUNCH(BM_, BN_, GK_, 1, 0, F); break;    \
        default:                                          \
            if (d.k_idx) LAUNCH_KLIST(BM_, BN_, GK_, F);  \
            else LAUNCH(BM_, BN_, GK_, 1, 1, F);          \
            break;                                        \
    }
    static int tile_cfg = -1;
    if (tile_cfg < 0) {
        const char *e = getenv("ASR_GEMM_TILE");
        tile_cfg = e ? atoi(e) : 2;   // 128 x 64 measured best on the config-2 shapes (tools/gemm_bench.py)
    }
    // few tiles (e.g. N = 256: the Conformer's d_model) leave 128 x 64 workgroups unevenly spread over
    // the 256 CUs; 64 x 64 tiles double the count (65 -> 78 TFLOP/s at 11968 x 256 x 2048,
    // tools/conformer_gemms.py).  Split-K launches keep their tuned shape.
    int cfg = tile_cfg;
    if (cfg == 2 && big && fast && splitk == 1 &&
        (int64_t)cdiv(d.M, 128) * cdiv(d.N, 64) * d.nbatch < 1024)
        cfg = 3;
    if (big && fast && cfg == 1) {          // experiment: 64 x 128 tiles (no gain once measured on correct results)
        dim3 grid(cdiv(d.N, 128), cdiv(d.M, 64), d.nbatch * splitk);
        ASR_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "sgemm: grid too large");
        if (GK == 32) { LAUNCH_MODES(64, 128, 32, true) } else { LAUNCH_MODES(64, 128, 16, true) }
    } else if (big && fast && cfg == 2) {   // default: 128 x 64 tiles (more, smaller tiles: shorter tail)
        dim3 grid(cdiv(d.N, 64), cdiv(d.M, 128), d.nbatch * splitk);
        ASR_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "sgemm: grid too large");
        if (GK == 32) { LAUNCH_MODES(128, 64, 32, true) } else { LAUNCH_MODES(128, 64, 16, true) }
    } else if (big && fast && cfg == 3) {   // experiment: 64 x 64 tiles (short workgroups)
        dim3 grid(cdiv(d.N, 64), cdiv(d.M, 64), d.nbatch * splitk);
        ASR_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "sgemm: grid too large");
        if (GK == 32) { LAUNCH_MODES(64, 64, 32, true) } else { LAUNCH_MODES(64, 64, 16, true) }
    } else if (big) {
        dim3 grid(cdiv(d.N, 128), cdiv(d.M, 128), d.nbatch * splitk);
        ASR_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "sgemm: grid too large");
        if (fast && GK == 32) { LAUNCH_MODES(128, 128, 32, true) }
        else if (fast) { LAUNCH_MODES(128, 128, 16, true) }
        else { LAUNCH_MODES(128, 128, 16, false) }
    } else {
        dim3 grid(cdiv(d.N, 64), cdiv(d.M, 64), d.nbatch * splitk);
        ASR_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "sgemm: grid too large");
        if (fast) { LAUNCH_MODES(64, 64, 16, true) } else { LAUNCH_MODES(64, 64, 16, false) }
    }
#undef LAUNCH_MODES
#undef LAUNCH_KLIST
#undef LAUNCH
    ASR_CHECK_LAUNCH("sgemm_kernel");
    return ASR_OK;
}
