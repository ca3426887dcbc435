// Error reporting and the small data-movement helpers of the backward passes.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void asr_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int asr_version(void) { return 100; }
extern "C" const char *asr_last_error(void) { return g_err; }

// ---------------------------------------------------------------- colsum --
// out[n] (+)= sum_m A[rowoff(m) + n].  Two stages so the result is
// deterministic: each block sums a strip of rows into part[strip][n] (or
// straight into out when one strip suffices).
// Strip height: enough strips for about 1024 workgroups (a 12 MB matrix of 256 columns would otherwise
// be summed by 96 workgroups), at least 32 rows each.
static int colsum_strips(int M, int N) {
    const int colblocks = cdiv(N, 64);
    int strips = cdiv(1024, colblocks);
    const int max_strips = cdiv(M > 0 ? M : 1, 32);
    if (strips > max_strips) strips = max_strips;
    return strips < 1 ? 1 : strips;
}
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ A, int M, int N,
                                                     int64_t s0, int64_t s1, int d,
                                                     float *__restrict__ out, int accumulate,
                                                     int rows_per_block) {
    __shared__ float sh[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int q = threadIdx.x >> 6;
    const int m_begin = blockIdx.y * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    float acc = 0.f;
    if (n < N) {
        // eight independent loads in flight per thread (a serial chain of dependent adds would leave
        // the pass latency-bound on tall matrices)
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int m = m_begin + q;
        for (; m + 28 < m_end; m += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += A[rowoff2(m + 4 * u, d, s0, s1) + n];
        }
        for (; m < m_end; m += 4) a8[0] += A[rowoff2(m, d, s0, s1) + n];
        acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    }
    sh[q][threadIdx.x & 63] = acc;
    __syncthreads();
    if (q == 0 && n < N) {
        float v = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
        float *dst = out + (int64_t)blockIdx.y * N + n;
        if (accumulate) v += *dst;
        *dst = v;
    }
}

extern "C" int asr_colsum(const float *A, int M, int N, int64_t s0, int64_t s1, int d, float *out,
                          int accumulate, float *part, void *stream) {
    ASR_CHECK_ARG(A && out, "colsum: null pointer");
    ASR_CHECK_ARG(M >= 0 && N > 0, "colsum: bad shape");
    if (d <= 0) d = 0x7fffffff;
    hipStream_t st = (hipStream_t)stream;
    const int strips = colsum_strips(M, N);
    if (strips == 1 || part == nullptr) {
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64), 1), dim3(256), 0, st, A, M, N, s0, s1, d, out,
                           accumulate, M > 0 ? M : 1);
    } else {
        const int rows = cdiv(M, strips);
        const int used = cdiv(M, rows);
        ASR_CHECK_ARG(used <= 65535, "colsum: too many strips");
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64), used), dim3(256), 0, st, A, M, N, s0, s1, d,
                           part, 0, rows);
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64), 1), dim3(256), 0, st, part, used, N,
                           (int64_t)N, (int64_t)0, 0x7fffffff, out, accumulate, used);
    }
    ASR_CHECK_LAUNCH("colsum_kernel");
    return ASR_OK;
}

extern "C" int64_t asr_colsum_ws_floats(int M, int N) { return (int64_t)colsum_strips(M, N) * N; }

// Several column sums over matrices of the same height in two launches (all the bias gradients of
// one block of the Conformer backward: their producers finish at different times, their sums do not
// depend on each other).  Row-major contiguous sources (row stride = n[i]).
__global__ __launch_bounds__(256) void colsum_multi_kernel(asr_colsum_list l, int M, int rows_per_block,
                                                           float *__restrict__ part, int stage) {
    __shared__ float sh[4][64];
    const int z = blockIdx.z, N = l.n[z];
    if ((int)blockIdx.x * 64 >= N) return;
    const float *A = stage == 0 ? l.src[z] : part + l.off[z];
    float *out = stage == 0 ? part + l.off[z] + (int64_t)blockIdx.y * N : l.out[z];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int q = threadIdx.x >> 6;
    const int m_begin = blockIdx.y * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    float acc = 0.f;
    if (n < N) {
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int m = m_begin + q;
        for (; m + 28 < m_end; m += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += A[(int64_t)(m + 4 * u) * N + n];
        }
        for (; m < m_end; m += 4) a8[0] += A[(int64_t)m * N + n];
        acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    }
    sh[q][threadIdx.x & 63] = acc;
    __syncthreads();
    if (q == 0 && n < N)
        out[n] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

static int colsum_multi_strips(const asr_colsum_list *l, int M) {
    int colblocks = 0;
    for (int i = 0; i < l->count; ++i) colblocks += cdiv(l->n[i], 64);
    int strips = cdiv(1024, colblocks > 0 ? colblocks : 1);
    const int max_strips = cdiv(M > 0 ? M : 1, 32);
    if (strips > max_strips) strips = max_strips;
    return strips < 1 ? 1 : strips;
}
extern "C" int64_t asr_colsum_multi_ws_floats(const asr_colsum_list *l, int M) {
    if (!l || l->count <= 0 || l->count > ASR_COLSUM_LIST_MAX) return 0;
    int64_t cols = 0;
    for (int i = 0; i < l->count; ++i) cols += l->n[i];
    return (int64_t)colsum_multi_strips(l, M) * cols;
}
extern "C" int asr_colsum_multi(const asr_colsum_list *lin, int M, float *part, void *stream) {
    ASR_CHECK_ARG(lin && part, "colsum_multi: null pointer");
    ASR_CHECK_ARG(lin->count > 0 && lin->count <= ASR_COLSUM_LIST_MAX && M > 0, "colsum_multi: bad list");
    asr_colsum_list l = *lin;
    const int strips = colsum_multi_strips(&l, M);
    const int rows = cdiv(M, strips), used = cdiv(M, rows);
    int maxn = 0;
    int64_t off = 0;
    for (int i = 0; i < l.count; ++i) {
        ASR_CHECK_ARG(l.src[i] && l.out[i] && l.n[i] > 0, "colsum_multi: bad entry %d", i);
        l.off[i] = off;
        off += (int64_t)used * l.n[i];
        if (l.n[i] > maxn) maxn = l.n[i];
    }
    ASR_CHECK_ARG(used <= 65535, "colsum_multi: too many strips");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(cdiv(maxn, 64), used, l.count), dim3(256), 0, st, l, M, rows, part, 0);
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(cdiv(maxn, 64), 1, l.count), dim3(256), 0, st, l, used, used, part, 1);
    ASR_CHECK_LAUNCH("colsum_multi_kernel");
    return ASR_OK;
}

// ------------------------------------------------------------- transpose --
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in,
                                                        float *__restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const float *src = in + (int64_t)blockIdx.z * R * C;
    float *dst = out + (int64_t)blockIdx.z * R * C;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8) {
        int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(int64_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) dst[(int64_t)c * R + r] = tile[tx][i];
    }
}

extern "C" int asr_transpose(const float *in, float *out, int R, int C, int nbatch, void *stream) {
    ASR_CHECK_ARG(in && out && R > 0 && C > 0 && nbatch > 0, "transpose: bad argument");
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(R, 32), nbatch), dim3(256), 0,
                       (hipStream_t)stream, in, out, R, C);
    ASR_CHECK_LAUNCH("transpose_kernel");
    return ASR_OK;
}

// ---------------------------------------------- embedding-gradient gather --
// table[v][:] = sum over rows i with idx[i] == v of src[i][:]   (Models/decoder.py:57,
// gradient of the L_ys lookup).  One block per table row scans the index list in
// order, so the sum is deterministic and needs no atomics.
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float *__restrict__ src,
                                                           const int64_t *__restrict__ idx, int n_rows,
                                                           int width, int64_t src_stride,
                                                           float *__restrict__ table, int vec) {
    // one pass over the index list collects this row's hits, a parallel rank sort
    // restores ascending order, then the block sums the hit rows.
    constexpr int MAXH = 1024;
    __shared__ int hits[MAXH];
    __shared__ int sorted[MAXH];
    __shared__ int nhit;
    const int v = blockIdx.x;
    float *dst = table + (int64_t)v * width;
    if (threadIdx.x == 0) nhit = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_rows; i += 256)
        if (idx[i] == v) {
            const int slot = atomicAdd(&nhit, 1);
            if (slot < MAXH) hits[slot] = i;
        }
    __syncthreads();
    const int total = nhit;
    const int n = total < MAXH ? total : MAXH;
    // restore ascending row order (deterministic sum): rank = number of smaller hits
    for (int a = threadIdx.x; a < n; a += 256) {
        const int key = hits[a];
        int rank = 0;
        for (int o = 0; o < n; ++o) rank += hits[o] < key;
        sorted[rank] = key;
    }
    __syncthreads();
    if (total <= MAXH && vec) {
        // 16-byte columns, 8 hit rows in flight per column; the additions stay in ascending row
        // order (a frequent label -- the padding id -- has hundreds of hits)
        for (int c4 = threadIdx.x; c4 < width / 4; c4 += 256) {
            float4 acc = zero4();
            for (int h0 = 0; h0 < n; h0 += 8) {
                float4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool ok = h0 + u < n;
                    t[u] = ld4(src + (int64_t)sorted[ok ? h0 + u : 0] * src_stride + c4 * 4);
                    if (!ok) t[u] = zero4();
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
            }
            *reinterpret_cast<float4 *>(dst + c4 * 4) = acc;
        }
        return;
    }
    for (int c = threadIdx.x; c < width; c += 256) {
        float acc = 0.f;
        if (total <= MAXH) {
            for (int h = 0; h < n; ++h) acc += src[(int64_t)sorted[h] * src_stride + c];
        } else {   // a label repeated more than MAXH times: plain ordered scan
            for (int i = 0; i < n_rows; ++i)
                if (idx[i] == v) acc += src[(int64_t)i * src_stride + c];
        }
        dst[c] = acc;
    }
}

extern "C" int asr_scatter_rows(const float *src, const int64_t *idx, int n_rows, int width,
                                int64_t src_stride, float *table, int n_table_rows, void *stream) {
    ASR_CHECK_ARG(src && idx && table, "scatter_rows: null pointer");
    ASR_CHECK_ARG(n_rows >= 0 && width > 0 && n_table_rows > 0, "scatter_rows: bad shape");
    const int vec = width % 4 == 0 && src_stride % 4 == 0 && (((uintptr_t)src) & 15) == 0 &&
                    (((uintptr_t)table) & 15) == 0;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(n_table_rows), dim3(256), 0, (hipStream_t)stream, src,
                       idx, n_rows, width, src_stride, table, vec);
    ASR_CHECK_LAUNCH("scatter_rows_kernel");
    return ASR_OK;
}

// ---------------------------------------------------------------- dropout --
__device__ __forceinline__ uint32_t mix32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return (uint32_t)x;
}

__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                          uint8_t *__restrict__ mask, int64_t n,
                                                          uint32_t thresh, float scale, uint64_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint8_t keep = mix32(seed + (uint64_t)i * 0x9E3779B97F4A7C15ULL) >= thresh;
        mask[i] = keep;
        y[i] = keep ? x[i] * scale : 0.f;
    }
}
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float *__restrict__ dy,
                                                          const uint8_t *__restrict__ mask,
                                                          float *__restrict__ dx, int64_t n, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = mask[i] ? dy[i] * scale : 0.f;
}

extern "C" int asr_dropout_fwd(const float *x, float *y, uint8_t *mask, int64_t n, float p, uint64_t seed,
                               void *stream) {
    ASR_CHECK_ARG(x && y && mask && n >= 0 && p >= 0.f && p < 1.f, "dropout_fwd: bad argument");
    if (n == 0) return ASR_OK;
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, mask, n,
                       thresh, 1.0f / (1.0f - p), seed);
    ASR_CHECK_LAUNCH("dropout_fwd_kernel");
    return ASR_OK;
}
__global__ __launch_bounds__(256) void dropout_fwd_block_kernel(const float *__restrict__ x,
                                                                float *__restrict__ y,
                                                                uint8_t *__restrict__ mask, int64_t rows,
                                                                int64_t cols, int64_t ld, int64_t first,
                                                                uint32_t thresh, float scale, uint64_t seed) {
    const int64_t n = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t o = (e / cols) * ld + e % cols;
        const uint8_t keep = mix32(seed + (uint64_t)(first + o) * 0x9E3779B97F4A7C15ULL) >= thresh;
        mask[o] = keep;
        y[o] = keep ? x[o] * scale : 0.f;
    }
}
extern "C" int asr_dropout_fwd_block(const float *x, float *y, uint8_t *mask, int64_t rows, int64_t cols,
                                     int64_t ld, int64_t first, float p, uint64_t seed, void *stream) {
    ASR_CHECK_ARG(x && y && mask && rows >= 0 && cols >= 0 && ld >= cols && first >= 0 && p >= 0.f && p < 1.f,
                  "dropout_fwd_block: bad argument");
    const int64_t n = rows * cols;
    if (n == 0) return ASR_OK;
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(dropout_fwd_block_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, mask,
                       rows, cols, ld, first, thresh, 1.0f / (1.0f - p), seed);
    ASR_CHECK_LAUNCH("dropout_fwd_block_kernel");
    return ASR_OK;
}
extern "C" int asr_dropout_bwd(const float *dy, const uint8_t *mask, float *dx, int64_t n, float p,
                               void *stream) {
    ASR_CHECK_ARG(dy && dx && mask && n >= 0 && p >= 0.f && p < 1.f, "dropout_bwd: bad argument");
    if (n == 0) return ASR_OK;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, mask, dx, n,
                       1.0f / (1.0f - p));
    ASR_CHECK_LAUNCH("dropout_bwd_kernel");
    return ASR_OK;
}

// ------------------------------------------------------------ elementwise --
// out = dU * (1 - U^2): gradient through u = tanh(.) of Models/decoder.py:55
__global__ __launch_bounds__(256) void mul_dtanh_kernel(const float *__restrict__ dU,
                                                        const float *__restrict__ U,
                                                        float *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float u = U[i];
        out[i] = dU[i] * (1.f - u * u);
    }
}
extern "C" int asr_mul_dtanh(const float *dU, const float *U, float *out, int64_t n, void *stream) {
    ASR_CHECK_ARG(dU && U && out && n >= 0, "mul_dtanh: bad argument");
    if (n == 0) return ASR_OK;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(mul_dtanh_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dU, U, out, n);
    ASR_CHECK_LAUNCH("mul_dtanh_kernel");
    return ASR_OK;
}

// out[i] = a[i] + b[i] (bias fusion b_ih + b_hh for both directions etc.)
__global__ __launch_bounds__(256) void scale_kernel(float *__restrict__ x, int64_t n, float s) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= s;
}
extern "C" int asr_scale(float *x, int64_t n, float s, void *stream) {
    ASR_CHECK_ARG(x && n >= 0, "scale: bad argument");
    if (n == 0) return ASR_OK;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, s);
    ASR_CHECK_LAUNCH("scale_kernel");
    return ASR_OK;
}

// ---------------------------------------------------------------------------
// *value = NaN if *flag_word != 0.  The persistent recurrence kernels set a sticky error word in
// their workspace when an in-launch wait times out (blstm.hip); the loss wrapper folds it into the
// one value the training loop reads every step, so a bad step ends the job as a NaN loss does in
// the reference (train.py:66-68) without a device->host copy of its own.
__global__ void poison_if_flag_kernel(const unsigned *__restrict__ flag_word, float *__restrict__ value) {
    if (*flag_word != 0u) *value = __uint_as_float(0x7fc00000u);
}

extern "C" int asr_poison_if_flag(const void *flag_word, float *value, void *stream) {
    ASR_CHECK_ARG(flag_word && value, "poison_if_flag: null pointer");
    hipLaunchKernelGGL(poison_if_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned *>(flag_word), value);
    ASR_CHECK_LAUNCH("poison_if_flag_kernel");
    return ASR_OK;
}

// ---------------------------------------------------------------------------
// out[r] = index of the first maximum of row r (torch.argmax's tie rule for the frame-wise
// labels of CTCModel.decode, Models/CTCModel.py:33): one wave per row.
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float *__restrict__ x, int64_t rows, int cols,
                                                          int64_t ld, int64_t *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float *xr = x + r * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < cols; c += 64) {
        const float v = xr[c];
        if (v > best || (v == best && c < bi) || (v != v && !(best != best))) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const bool onan = ob != ob, bnan = best != best;
        // torch treats NaN as the maximum; among equals the lower index wins
        const bool take = bnan ? (onan && oi < bi) : (onan || ob > best || (ob == best && oi < bi));
        if (take) { best = ob; bi = oi; }
    }
    if (lane == 0) out[r] = bi == 0x7fffffff ? 0 : bi;
}

extern "C" int asr_argmax_rows(const float *x, int64_t rows, int cols, int64_t ld, int64_t *out, void *stream) {
    ASR_CHECK_ARG(x && out, "argmax_rows: null pointer");
    ASR_CHECK_ARG(rows >= 0 && cols > 0 && ld >= cols, "argmax_rows: bad shape");
    if (rows == 0) return ASR_OK;
    ASR_CHECK_ARG(cdiv(rows, 4) <= 0x7fffffff, "argmax_rows: too many rows");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, cols,
                       ld, out);
    ASR_CHECK_LAUNCH("argmax_rows_kernel");
    return ASR_OK;
}

// ---------------------------------------------------------------------------
// A HIP stream with a hardware queue of its own.  HIP multiplexes a process's streams onto
// GPU_MAX_HW_QUEUES (4) hardware queues PER PRIORITY; the one exception is a stream created with a
// compute-unit mask, which always gets a new queue.  asr_amd/streams.py can run the off-critical-path
// work (weight-gradient GEMMs, bucket packing) on such a stream (ASR_SIDE_STREAM_KIND=cumask); it is
// NOT its default: measured, a stream created this way overlaps worse than a plain normal-priority
// stream created before RCCL's, which is what streams.py uses.  (Round 1 used
// a high-priority stream for the same purpose; a high-priority stream that waits on an event of the
// normal-priority main stream can starve it -- tools/stall_probe.py.)
// cu_mask: n_words x 32 bits, bit i = compute unit i enabled (NULL / 0 words: every CU).
extern "C" int asr_stream_create(void **stream, const uint32_t *cu_mask, int n_words) {
    ASR_CHECK_ARG(stream != nullptr && n_words >= 0 && n_words <= 32, "stream_create: bad argument");
    uint32_t all[32];
    if (cu_mask == nullptr || n_words == 0) {
        int dev = 0, cus = 0;
        ASR_CHECK_HIP(hipGetDevice(&dev));
        ASR_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        n_words = (cus + 31) / 32;
        ASR_CHECK_ARG(n_words >= 1 && n_words <= 32, "stream_create: %d compute units", cus);
        for (int w = 0; w < n_words; ++w) {
            const int left = cus - 32 * w;
            all[w] = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
        }
        cu_mask = all;
    }
    hipStream_t st = nullptr;
    ASR_CHECK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, cu_mask));
    *stream = (void *)st;
    return ASR_OK;
}

extern "C" int asr_stream_destroy(void *stream) {
    ASR_CHECK_ARG(stream != nullptr, "stream_destroy: null stream");
    ASR_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    return ASR_OK;
}

// ---------------------------------------------------------------------------
// A stand-in for a collective's kernel, for tests and timing experiments only (tests/test_gpu_parity.py::
// test_step_under_competing_kernels; never on the product path): n_wg workgroups of `threads` lanes that each hold
// lds_bytes of LDS and a compute unit's wave slots for `usec` microseconds of wall time (the 100 MHz counter), doing
// nothing.  RCCL's kernels look like this to the rest of the GPU: one workgroup per channel, resident for the length
// of the collective, on whichever compute units the dispatcher picks.  The persistent kernels of this library need
// all their workgroups resident at once; this lets a test check that they only WAIT while such a kernel is in the way.
__global__ void occupy_kernel(int usec, int lds_floats, float *sink) {
    extern __shared__ float occ_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) occ_lds[i] = (float)i;
    __syncthreads();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)usec * 100ull) {
        if (lds_floats > 0) acc += occ_lds[(threadIdx.x * 7) % lds_floats];
        __builtin_amdgcn_s_sleep(8);
    }
    if (sink != nullptr && acc == -1.0f) sink[0] = acc;     // keeps the LDS reads alive; never true
}

extern "C" int asr_debug_occupy(int n_wg, int threads, int lds_bytes, int usec, void *stream) {
    ASR_CHECK_ARG(n_wg > 0 && n_wg <= 4096 && threads >= 64 && threads <= 1024 && threads % 64 == 0 && lds_bytes >= 0 &&
                      lds_bytes <= 64 * 1024 && usec >= 0 && usec <= 100000,
                  "debug_occupy: bad argument (n_wg %d, threads %d, lds %d B, %d us)", n_wg, threads, lds_bytes, usec);
    hipLaunchKernelGGL(occupy_kernel, dim3(n_wg), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, usec,
                       lds_bytes / 4, (float *)nullptr);
    ASR_CHECK_LAUNCH("occupy_kernel");
    return ASR_OK;
}
