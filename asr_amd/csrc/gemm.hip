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
#include "common.cuh"

#define GK 16
#define GPAD 4

struct OperandView {
    const float *p;
    int mode;          // 0: k contiguous, rows indexed by m/n; 1: m/n contiguous, rows indexed by k
    int64_t s0, s1;
    int d;
};

__device__ __forceinline__ int64_t view_rowoff(const OperandView &v, int64_t i) {
    return (i / v.d) * v.s1 + (i % v.d) * v.s0;
}

// Load this thread's share of a (R x GK) operand tile into regs[R/64] float4s.
//   mode 0: element (r,k) at p[rowoff(r0+r) + k];   thread f -> r = f>>2, k4 = f&3
//   mode 1: element (r,k) at p[rowoff(k) + r0+r];   thread f -> k = f/(R/4), r4 = f%(R/4)
template <int R>
__device__ __forceinline__ void load_tile(const OperandView &v, int r0, int rmax, int k0, int kend,
                                          bool vec, float4 (&regs)[R / 64]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R / 64; ++i) {
        const int f = tid + 256 * i;
        float4 out = zero4();
        if (v.mode == 0) {
            const int r = r0 + (f >> 2);
            const int k = k0 + (f & 3) * 4;
            if (r < rmax && k < kend) {
                const float *src = v.p + view_rowoff(v, r) + k;
                if (vec && k + 3 < kend) {
                    out = ld4(src);
                } else {
                    out.x = src[0];
                    if (k + 1 < kend) out.y = src[1];
                    if (k + 2 < kend) out.z = src[2];
                    if (k + 3 < kend) out.w = src[3];
                }
            }
        } else {
            const int k = k0 + f / (R / 4);
            const int r = r0 + (f % (R / 4)) * 4;
            if (k < kend && r < rmax) {
                const float *src = v.p + view_rowoff(v, k) + r;
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
}

template <int R>
__device__ __forceinline__ void store_tile(int mode, float (*lds)[R + GPAD], const float4 (&regs)[R / 64]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < R / 64; ++i) {
        const int f = tid + 256 * i;
        if (mode == 0) {
            const int r = f >> 2, k = (f & 3) * 4;
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

template <int BM, int BN>
__global__ __launch_bounds__(256) void sgemm_kernel(asr_gemm_desc d, int vecA, int vecB, int ktiles_per_split) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) float As[2][GK][BM + GPAD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][BN + GPAD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int batch = blockIdx.z / d.splitk, split = blockIdx.z % d.splitk;

    OperandView va{d.A + (int64_t)batch * d.batch_a, d.a_mode, d.a_s0, d.a_s1, d.a_d};
    OperandView vb{d.B + (int64_t)batch * d.batch_b, d.b_mode, d.b_s0, d.b_s1, d.b_d};
    float *C = d.C + (int64_t)batch * d.batch_c;

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

    float4 ra[BM / 64], rb[BN / 64];
    if (nk > 0) {
        load_tile<BM>(va, m0, d.M, kbeg, kend, vecA, ra);
        load_tile<BN>(vb, n0, d.N, kbeg, kend, vecB, rb);
        store_tile<BM>(va.mode, As[0], ra);
        store_tile<BN>(vb.mode, Bs[0], rb);
    }
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            load_tile<BM>(va, m0, d.M, kbeg + (kt + 1) * GK, kend, vecA, ra);
            load_tile<BN>(vb, n0, d.N, kbeg + (kt + 1) * GK, kend, vecB, rb);
        }
        const int kh = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[cur][2 * kk + kh][wm * WM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[cur][2 * kk + kh][wn * WN + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            store_tile<BM>(va.mode, As[cur ^ 1], ra);
            store_tile<BN>(vb.mode, Bs[cur ^ 1], rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool first = (split == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + (lane & 31);
            if (col >= d.N) continue;
            float bias = 0.f;
            if (first) {
                if (d.bias1) bias += d.bias1[col];
                if (d.bias2) bias += d.bias2[col];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= d.M) continue;
                float *dst = C + rowoff2(row, d.c_d, d.c_s0, d.c_s1) + col;
                float v = d.alpha * acc[i][j][r] + bias;
                if (d.splitk > 1) {
                    atomicAdd(dst, v);
                } else {
                    if (d.accumulate) v += *dst;
                    if (d.act == 1) v = tanh_f(v);
                    *dst = v;
                }
            }
        }
}

static bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

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
    ASR_CHECK_ARG(d.act == 0 || d.act == 1, "sgemm: bad act %d", d.act);
    if (d.M == 0 || d.N == 0) return ASR_OK;

    auto strides_ok = [&](int64_t s0, int64_t s1, int64_t sb) {
        return (s0 % 4 == 0) && (s1 % 4 == 0 || s1 == 0) && (sb % 4 == 0);
    };
    const int vecA = aligned16(d.A) && strides_ok(d.a_s0, d.a_s1, d.nbatch > 1 ? d.batch_a : 0);
    const int vecB = aligned16(d.B) && strides_ok(d.b_s0, d.b_s1, d.nbatch > 1 ? d.batch_b : 0);

    const int ktiles = cdiv(d.K, GK);
    int splitk = d.splitk;
    if (splitk > ktiles) splitk = ktiles > 0 ? ktiles : 1;
    d.splitk = splitk;
    const int per = ktiles > 0 ? cdiv(ktiles, splitk) : 0;

    hipStream_t st = (hipStream_t)stream;
    const int64_t big_tiles = (int64_t)cdiv(d.M, 128) * cdiv(d.N, 128) * d.nbatch * splitk;
    const bool big = d.M >= 128 && d.N >= 128 && big_tiles >= 192;
    if (big) {
        dim3 grid(cdiv(d.N, 128), cdiv(d.M, 128), d.nbatch * splitk);
        ASR_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "sgemm: grid too large");
        hipLaunchKernelGGL((sgemm_kernel<128, 128>), grid, dim3(256), 0, st, d, vecA, vecB, per);
    } else {
        dim3 grid(cdiv(d.N, 64), cdiv(d.M, 64), d.nbatch * splitk);
        ASR_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "sgemm: grid too large");
        hipLaunchKernelGGL((sgemm_kernel<64, 64>), grid, dim3(256), 0, st, d, vecA, vecB, per);
    }
    ASR_CHECK_LAUNCH("sgemm_kernel");
    return ASR_OK;
}
