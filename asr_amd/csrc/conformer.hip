// Conformer encoder (Models/Transformer_encoder.py) -- everything that is not a matrix product:
// LayerNorm, swish / GLU / ReLU gradients, the depthwise convolution, BatchNorm1d over all frames,
// the rel_shift + masked softmax of the relative-position attention, the 3x3 stride-2 patch
// gather of the Conv2d subsampling.  The matrix products run on asr_sgemm.
//
// Layout: activations stay (B, T, C) row-major ("channels last") from the subsampling to the
// encoder output: every kernel here is coalesced along C, and the reference's transposes to
// (B, C, T) around its Conv1d / BatchNorm1d calls (:230,237) disappear.  All of these kernels
// are HBM-bound passes (bytes per launch in DESIGN.md 3.6).
#include "common.h"

#define GRID1D(n) ((int)(((n) + 255) / 256 > 8192 ? 8192 : ((n) + 255) / 256))
#define FOR_EACH(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (int64_t)gridDim.x * 256)

__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ------------------------------------------------------------- LayerNorm --
// nn.LayerNorm(d) (:161,194,218,250): one wave per row, the row in registers (d <= 1024).
#define LN_MAXV 16
// NV = registers per lane (64 NV >= d): instantiated for 1, 2, 4, 8, 16 so a 256-wide row costs 4, not 16
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                     const float *__restrict__ b, float *__restrict__ y,
                                                     float *__restrict__ mean, float *__restrict__ rstd,
                                                     int64_t rows, int d, float eps) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float *xr = x + r * d;
    float v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        v[i] = c < d ? xr[c] : 0.f;
        s += v[i];
    }
    const float mu = wave_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        const float t = c < d ? v[i] - mu : 0.f;
        q += t * t;
    }
    const float rs = rsqrtf(wave_sum(q) / d + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < d) y[r * d + c] = (v[i] - mu) * rs * g[c] + b[c];
    }
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}

// dx = rstd (gy - mean(gy) - xhat mean(gy xhat)), gy = dy gamma; per-block partial sums of
// dy xhat (d gamma) and dy (d beta) go to part[block][2][d] for a deterministic second stage.
#define LN_BWD_ROWS 16
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                     const float *__restrict__ g, const float *__restrict__ mean,
                                                     const float *__restrict__ rstd, float *__restrict__ dx,
                                                     float *__restrict__ part, int64_t rows, int d, const float *__restrict__ res) {
    __shared__ float red[4][2][256];   // 256 columns at a time
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * LN_BWD_ROWS;
    float gg[NV], dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        gg[i] = c < d ? g[c] : 0.f;
        dg[i] = 0.f;
        db[i] = 0.f;
    }
    // narrow rows (NV <= 4): the loads of all four rows of the wave are issued before the first reduction, so
    // the row-to-row chain of load -> two wave reductions -> store does not expose one memory round trip per row
    constexpr int RPW = NV <= 4 ? 4 : 1;
    const int64_t rend = min(rows, r0 + LN_BWD_ROWS);
    for (int64_t rb = r0 + w; rb < rend; rb += 4 * RPW) {
        float dyv[RPW][NV], xv[RPW][NV], mus[RPW], rss[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int64_t r = rb + 4 * q;
            const bool ok = r < rend;
            mus[q] = ok ? mean[r] : 0.f;
            rss[q] = ok ? rstd[r] : 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + i * 64;
                dyv[q][i] = (ok && c < d) ? dy[r * d + c] : 0.f;
                xv[q][i] = (ok && c < d) ? x[r * d + c] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int64_t r = rb + 4 * q;
            if (r >= rend) break;
            const float mu = mus[q], rs = rss[q];
            float xh[NV], gy[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + i * 64;
                xh[i] = c < d ? (xv[q][i] - mu) * rs : 0.f;
                gy[i] = dyv[q][i] * gg[i];
                dg[i] += dyv[q][i] * xh[i];
                db[i] += dyv[q][i];
                s1 += gy[i];
                s2 += gy[i] * xh[i];
            }
            s1 = wave_sum(s1) / d;
            s2 = wave_sum(s2) / d;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + i * 64;
                if (c < d) {
                    float v = rs * (gy[i] - s1 - xh[i] * s2);
                    if (res) v += res[r * d + c];
                    dx[r * d + c] = v;
                }
            }
        }
    }
    // cross-wave sum of the column partials, 256 columns at a time
    float *out = part + (int64_t)blockIdx.x * 2 * d;
#pragma unroll
    for (int i0 = 0; i0 < NV; i0 += 4) {
        if (i0 * 64 >= d) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[w][0][i * 64 + lane] = i0 + i < NV ? dg[i0 + i < NV ? i0 + i : 0] : 0.f;
            red[w][1][i * 64 + lane] = i0 + i < NV ? db[i0 + i < NV ? i0 + i : 0] : 0.f;
        }
        __syncthreads();
        const int c = i0 * 64 + threadIdx.x;
        if (c < d) {
            out[c] = (red[0][0][threadIdx.x] + red[1][0][threadIdx.x]) + (red[2][0][threadIdx.x] + red[3][0][threadIdx.x]);
            out[d + c] = (red[0][1][threadIdx.x] + red[1][1][threadIdx.x]) + (red[2][1][threadIdx.x] + red[3][1][threadIdx.x]);
        }
    }
}

extern "C" int asr_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                                 float *rstd, int64_t rows, int d, float eps, void *stream) {
    ASR_CHECK_ARG(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
    ASR_CHECK_ARG(rows >= 0 && d > 0 && d <= 64 * LN_MAXV, "layernorm_fwd: d must be in [1, %d]", 64 * LN_MAXV);
    if (rows == 0) return ASR_OK;
#define LN_FWD(NV_)                                                                                              \
    hipLaunchKernelGGL((ln_fwd_kernel<NV_>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, \
                       y, mean, rstd, rows, d, eps)
    if (d <= 64) LN_FWD(1); else if (d <= 128) LN_FWD(2); else if (d <= 256) LN_FWD(4);
    else if (d <= 512) LN_FWD(8); else LN_FWD(16);
#undef LN_FWD
    ASR_CHECK_LAUNCH("ln_fwd_kernel");
    return ASR_OK;
}
extern "C" int64_t asr_layernorm_bwd_blocks(int64_t rows) { return rows > 0 ? (rows + LN_BWD_ROWS - 1) / LN_BWD_ROWS : 1; }
extern "C" int asr_layernorm_bwd(const float *dy, const float *x, const float *gamma, const float *mean,
                                 const float *rstd, float *dx, float *part, int64_t rows, int d, const float *res,
                                 void *stream) {
    ASR_CHECK_ARG(dy && x && gamma && mean && rstd && dx && part, "layernorm_bwd: null pointer");
    ASR_CHECK_ARG(rows > 0 && d > 0 && d <= 64 * LN_MAXV, "layernorm_bwd: bad shape");
#define LN_BWD(NV_)                                                                                          \
    hipLaunchKernelGGL((ln_bwd_kernel<NV_>), dim3((unsigned)asr_layernorm_bwd_blocks(rows)), dim3(256), 0,       \
                       (hipStream_t)stream, dy, x, gamma, mean, rstd, dx, part, rows, d, res)
    if (d <= 64) LN_BWD(1); else if (d <= 128) LN_BWD(2); else if (d <= 256) LN_BWD(4);
    else if (d <= 512) LN_BWD(8); else LN_BWD(16);
#undef LN_BWD
    ASR_CHECK_LAUNCH("ln_bwd_kernel");
    return ASR_OK;
}

// ------------------------------------------------------------ elementwise --
// swish (:175-177) x sigmoid(x); GLU (:232-233) out * sigmoid(gate) over the two halves of a row;
// ReLU gradient for the subsampling convolutions (:391-394); out = a x + b y for residuals
// (:259,263-265) and running statistics.
// mask != NULL: the inverted dropout that follows the activation (:205) in the same pass
__global__ __launch_bounds__(256) void swish_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                        uint8_t *__restrict__ mask, int64_t n, uint32_t thresh,
                                                        float scale, uint64_t seed) {
    FOR_EACH(i, n) {
        const float v = x[i];
        float s = v * sigmoid_exact(v);
        if (mask) {
            const bool keep = asr_keep(seed, i, thresh);
            mask[i] = keep;
            s = keep ? s * scale : 0.f;
        }
        y[i] = s;
    }
}
__global__ __launch_bounds__(256) void swish_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                        const uint8_t *__restrict__ mask, float *__restrict__ dx,
                                                        int64_t n, float scale) {
    FOR_EACH(i, n) {
        const float v = x[i], s = sigmoid_exact(v);
        float g = dy[i];
        if (mask) g = mask[i] ? g * scale : 0.f;
        dx[i] = g * s * (1.f + v * (1.f - s));
    }
}
// y = x + factor * keep * o with keep = the AND of ndrop dropout draws (seeds seed, seed + 1, ...): the output
// dropouts of a block and its residual connection (:207,259,263-265) in one pass; mask == NULL: y = x + factor * o
__global__ __launch_bounds__(256) void residual_dropout_kernel(const float *__restrict__ x, const float *__restrict__ o,
                                                               float *__restrict__ y, uint8_t *__restrict__ mask, int64_t n,
                                                               float factor, uint32_t thresh, uint64_t seed, int ndrop) {
    FOR_EACH(i, n) {
        float v = o[i] * factor;
        if (mask) {
            bool keep = true;
            for (int k = 0; k < ndrop; ++k) keep = keep && asr_keep(seed + k, i, thresh);
            mask[i] = keep;
            v = keep ? v : 0.f;
        }
        y[i] = x[i] + v;
    }
}
__global__ __launch_bounds__(256) void scale_mask_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ mask,
                                                         float *__restrict__ out, int64_t n, float factor) {
    FOR_EACH(i, n) out[i] = (!mask || mask[i]) ? dy[i] * factor : 0.f;
}
// the same with 16-byte accesses (n % 4 == 0, aligned pointers): four values and their four mask bytes per thread
__global__ __launch_bounds__(256) void scale_mask_v4_kernel(const float4 *__restrict__ dy, const uint32_t *__restrict__ mask,
                                                            float4 *__restrict__ out, int64_t n4, float factor) {
    FOR_EACH(i, n4) {
        const float4 v = dy[i];
        const uint32_t m = mask ? mask[i] : 0xffffffffu;
        out[i] = make_float4((m & 0xffu) ? v.x * factor : 0.f, (m & 0xff00u) ? v.y * factor : 0.f,
                             (m & 0xff0000u) ? v.z * factor : 0.f, (m & 0xff000000u) ? v.w * factor : 0.f);
    }
}
__global__ __launch_bounds__(256) void glu_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t rows, int d) {
    FOR_EACH(i, rows * d) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        y[i] = x[r * 2 * d + c] * sigmoid_exact(x[r * 2 * d + d + c]);
    }
}
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                      float *__restrict__ dx, int64_t rows, int d) {
    FOR_EACH(i, rows * d) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        const float a = x[r * 2 * d + c], s = sigmoid_exact(x[r * 2 * d + d + c]), g = dy[i];
        dx[r * 2 * d + c] = g * s;
        dx[r * 2 * d + d + c] = g * a * s * (1.f - s);
    }
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                       float *__restrict__ dx, int64_t n) {
    FOR_EACH(i, n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
__global__ __launch_bounds__(256) void axpby_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                    float *__restrict__ out, int64_t n, float a, float b) {
    FOR_EACH(i, n) out[i] = a * x[i] + b * y[i];
}
// qu = q + u, qv = q + v with u, v (d) broadcast over the rows (:94-95)
__global__ __launch_bounds__(256) void add_two_bias_kernel(const float *__restrict__ q, const float *__restrict__ u,
                                                           const float *__restrict__ v, float *__restrict__ qu,
                                                           float *__restrict__ qv, int64_t rows, int d) {
    FOR_EACH(i, rows * d) {
        const int c = (int)(i % d);
        const float t = q[i];
        qu[i] = t + u[c];
        qv[i] = t + v[c];
    }
}

#define EW_ENTRY(name, check, launch)                                                   \
    {                                                                                   \
        ASR_CHECK_ARG(check, name ": bad argument");                                    \
        if (n == 0) return ASR_OK;                                                      \
        launch;                                                                         \
        ASR_CHECK_LAUNCH(name);                                                         \
        return ASR_OK;                                                                  \
    }
static uint32_t drop_thresh(float p) { return (uint32_t)((double)p * 4294967296.0); }
extern "C" int asr_swish_fwd(const float *x, float *y, uint8_t *mask, int64_t n, float p, uint64_t seed, void *stream)
    EW_ENTRY("swish_fwd", x && y && n >= 0 && p >= 0.f && p < 1.f,
             hipLaunchKernelGGL(swish_fwd_kernel, dim3(GRID1D(n)), dim3(256), 0, (hipStream_t)stream, x, y, mask, n,
                                drop_thresh(p), 1.0f / (1.0f - p), seed))
extern "C" int asr_swish_bwd(const float *dy, const float *x, const uint8_t *mask, float *dx, int64_t n, float p,
                             void *stream)
    EW_ENTRY("swish_bwd", dy && x && dx && n >= 0 && p >= 0.f && p < 1.f,
             hipLaunchKernelGGL(swish_bwd_kernel, dim3(GRID1D(n)), dim3(256), 0, (hipStream_t)stream, dy, x, mask, dx, n,
                                1.0f / (1.0f - p)))
extern "C" int asr_residual_dropout(const float *x, const float *o, float *y, uint8_t *mask, int64_t n, float factor,
                                    float p, uint64_t seed, int ndrop, void *stream)
    EW_ENTRY("residual_dropout", x && o && y && n >= 0 && p >= 0.f && p < 1.f && ndrop >= 0 && ndrop <= 4,
             hipLaunchKernelGGL(residual_dropout_kernel, dim3(GRID1D(n)), dim3(256), 0, (hipStream_t)stream, x, o, y, mask,
                                n, factor, drop_thresh(p), seed, ndrop))
extern "C" int asr_scale_mask(const float *dy, const uint8_t *mask, float *out, int64_t n, float factor, void *stream) {
    if (n > 0 && n % 4 == 0 && dy && out && (((uintptr_t)dy | (uintptr_t)out) & 15) == 0 && ((uintptr_t)mask & 3) == 0) {
        hipLaunchKernelGGL(scale_mask_v4_kernel, dim3(GRID1D(n / 4)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float4 *>(dy), reinterpret_cast<const uint32_t *>(mask),
                           reinterpret_cast<float4 *>(out), n / 4, factor);
        ASR_CHECK_LAUNCH("scale_mask_v4_kernel");
        return ASR_OK;
    }
    EW_ENTRY("scale_mask", dy && out && n >= 0,
             hipLaunchKernelGGL(scale_mask_kernel, dim3(GRID1D(n)), dim3(256), 0, (hipStream_t)stream, dy, mask, out, n,
                                factor))
}
extern "C" int asr_relu_bwd(const float *dy, const float *y, float *dx, int64_t n, void *stream)
    EW_ENTRY("relu_bwd", dy && y && dx && n >= 0,
             hipLaunchKernelGGL(relu_bwd_kernel, dim3(GRID1D(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n))
extern "C" int asr_axpby(const float *x, const float *y, float *out, int64_t n, float a, float b, void *stream)
    EW_ENTRY("axpby", x && y && out && n >= 0,
             hipLaunchKernelGGL(axpby_kernel, dim3(GRID1D(n)), dim3(256), 0, (hipStream_t)stream, x, y, out, n, a, b))
extern "C" int asr_glu_fwd(const float *x, float *y, int64_t rows, int d, void *stream) {
    const int64_t n = rows * d;
    EW_ENTRY("glu_fwd", x && y && rows >= 0 && d > 0,
             hipLaunchKernelGGL(glu_fwd_kernel, dim3(GRID1D(n)), dim3(256), 0, (hipStream_t)stream, x, y, rows, d))
}
extern "C" int asr_glu_bwd(const float *dy, const float *x, float *dx, int64_t rows, int d, void *stream) {
    const int64_t n = rows * d;
    EW_ENTRY("glu_bwd", dy && x && dx && rows >= 0 && d > 0,
             hipLaunchKernelGGL(glu_bwd_kernel, dim3(GRID1D(n)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, rows, d))
}
extern "C" int asr_add_two_bias(const float *q, const float *u, const float *v, float *qu, float *qv, int64_t rows,
                                int d, void *stream) {
    const int64_t n = rows * d;
    EW_ENTRY("add_two_bias", q && u && v && qu && qv && rows >= 0 && d > 0,
             hipLaunchKernelGGL(add_two_bias_kernel, dim3(GRID1D(n)), dim3(256), 0, (hipStream_t)stream, q, u, v, qu,
                                qv, rows, d))
}

// ---------------------------------------------------- depthwise convolution --
// DepthwiseConv.conv (:183,187-188): y[b,t,c] = bias[c] + sum_k w[c,k] x[b, t + k - pad, c], zero
// padding, K odd and pad = (K-1)/2 (K = 31, :215-216,242-244).  A thread owns one channel and DW_TF
// consecutive frames: its K taps live in registers, every input frame is loaded once and feeds all
// the outputs it touches.  flip != 0 reverses the taps: the same kernel is then the data gradient
// dx[t] = sum_k w[c,k] dy[t - k + pad].
#define DW_K 31
#define DW_TF 8
__global__ __launch_bounds__(256) void dwconv_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                     const float *__restrict__ bias, float *__restrict__ y, int T,
                                                     int C, int flip) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int t0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * DW_TF;
    const int b = blockIdx.z;
    if (c >= C || t0 >= T) return;
    float wk[DW_K];
#pragma unroll
    for (int k = 0; k < DW_K; ++k) wk[k] = w[c * DW_K + (flip ? DW_K - 1 - k : k)];
    float acc[DW_TF];
    const float bv = bias ? bias[c] : 0.f;
#pragma unroll
    for (int o = 0; o < DW_TF; ++o) acc[o] = bv;
    const float *xb = x + (int64_t)b * T * C + c;
#pragma unroll
    for (int i = 0; i < DW_TF + DW_K - 1; ++i) {
        const int t = t0 - DW_K / 2 + i;
        const float xv = (t >= 0 && t < T) ? xb[(int64_t)t * C] : 0.f;
#pragma unroll
        for (int o = 0; o < DW_TF; ++o) {
            const int k = i - o;
            if (k >= 0 && k < DW_K) acc[o] += wk[k] * xv;
        }
    }
#pragma unroll
    for (int o = 0; o < DW_TF; ++o)
        if (t0 + o < T) y[((int64_t)b * T + t0 + o) * C + c] = acc[o];
}

// dw[c,k] = sum_{b,t} dy[b,t,c] x[b, t + k - pad, c]: a block takes 64 channels x DWW_FRAMES frames of
// one utterance, each wave a quarter of the frames; partials part[chunk][k][c] are summed by asr_colsum.
#define DWW_FRAMES 64
__global__ __launch_bounds__(256) void dwconv_bwd_w_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                           float *__restrict__ part, int T, int C, int chunks_per_b) {
    __shared__ float red[4][DW_K][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int b = blockIdx.y / chunks_per_b, ch = blockIdx.y % chunks_per_b;
    float acc[DW_K];
#pragma unroll
    for (int k = 0; k < DW_K; ++k) acc[k] = 0.f;
    if (c < C) {
        // the wave's 16 gradient frames and the 46 input frames they touch, loaded once into registers
        constexpr int NF = DWW_FRAMES / 4;
        const float *xb = x + (int64_t)b * T * C + c;
        const float *db = dy + (int64_t)b * T * C + c;
        const int tb = ch * DWW_FRAMES + w * NF;
        float g[NF], xw[NF + DW_K - 1];
#pragma unroll
        for (int o = 0; o < NF; ++o) g[o] = tb + o < T ? db[(int64_t)(tb + o) * C] : 0.f;
#pragma unroll
        for (int i = 0; i < NF + DW_K - 1; ++i) {
            const int ti = tb - DW_K / 2 + i;
            xw[i] = (ti >= 0 && ti < T) ? xb[(int64_t)ti * C] : 0.f;
        }
#pragma unroll
        for (int o = 0; o < NF; ++o)
#pragma unroll
            for (int k = 0; k < DW_K; ++k) acc[k] += g[o] * xw[o + k];
    }
#pragma unroll
    for (int k = 0; k < DW_K; ++k) red[w][k][lane] = acc[k];
    __syncthreads();
    if (w == 0 && c < C) {
        float *out = part + (int64_t)blockIdx.y * DW_K * C + c;
#pragma unroll
        for (int k = 0; k < DW_K; ++k)
            out[(int64_t)k * C] = (red[0][k][lane] + red[1][k][lane]) + (red[2][k][lane] + red[3][k][lane]);
    }
}

extern "C" int asr_dwconv(const float *x, const float *w, const float *bias, float *y, int B, int T, int C, int K,
                          int flip, void *stream) {
    ASR_CHECK_ARG(x && w && y, "dwconv: null pointer");
    ASR_CHECK_ARG(K == DW_K, "dwconv: kernel size must be %d (Transformer_encoder.py:215)", DW_K);
    ASR_CHECK_ARG(B > 0 && T > 0 && C > 0 && B <= 65535, "dwconv: bad shape");
    hipLaunchKernelGGL(dwconv_kernel, dim3(cdiv(C, 64), cdiv(T, 4 * DW_TF), B), dim3(256), 0, (hipStream_t)stream, x, w,
                       bias, y, T, C, flip);
    ASR_CHECK_LAUNCH("dwconv_kernel");
    return ASR_OK;
}
extern "C" int64_t asr_dwconv_bwd_w_chunks(int B, int T) { return (int64_t)B * cdiv(T, DWW_FRAMES); }
extern "C" int asr_dwconv_bwd_w(const float *dy, const float *x, float *part, int B, int T, int C, int K, void *stream) {
    ASR_CHECK_ARG(dy && x && part, "dwconv_bwd_w: null pointer");
    ASR_CHECK_ARG(K == DW_K, "dwconv_bwd_w: kernel size must be %d", DW_K);
    ASR_CHECK_ARG(B > 0 && T > 0 && C > 0, "dwconv_bwd_w: bad shape");
    const int cpb = cdiv(T, DWW_FRAMES);
    ASR_CHECK_ARG((int64_t)B * cpb <= 65535, "dwconv_bwd_w: too many chunks");
    hipLaunchKernelGGL(dwconv_bwd_w_kernel, dim3(cdiv(C, 64), B * cpb), dim3(256), 0, (hipStream_t)stream, dy, x, part,
                       T, C, cpb);
    ASR_CHECK_LAUNCH("dwconv_bwd_w_kernel");
    return ASR_OK;
}

// ---------------------------------------------------- BatchNorm1d + swish --
// nn.BatchNorm1d(d) over (B, d, T) (:223,235): per-channel statistics over ALL B*T frames (padded
// ones included, as in the reference), followed by swish (:236), fused.  The caller supplies the
// channel sums: msum (sum of x, from asr_colsum) and vsum (sum of (x - mean)^2, from
// asr_bn_var_part + asr_colsum) with multipliers ms, vs (1/rows in training; 1 with the running
// statistics in eval mode), so no statistic ever visits the host.
#define BN_ROWS 32       // rows per block of the partial-sum kernels: 8 per wave, all their loads in flight at once
__global__ __launch_bounds__(256) void bn_var_part_kernel(const float *__restrict__ x, const float *__restrict__ msum,
                                                          float ms, float *__restrict__ part, int64_t rows, int C) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * BN_ROWS;
    float acc = 0.f;
    if (c < C) {
        const float mu = msum[c] * ms;
        float xv[BN_ROWS / 4];
#pragma unroll
        for (int k = 0; k < BN_ROWS / 4; ++k) {       // branch-free: rows past the end re-read row r0 and count as zero
            const int64_t r = r0 + w + 4 * k;
            xv[k] = x[(r < rows ? r : r0) * C + c];
        }
#pragma unroll
        for (int k = 0; k < BN_ROWS / 4; ++k) {
            const float t = r0 + w + 4 * k < rows ? xv[k] - mu : 0.f;
            acc += t * t;
        }
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < C) part[(int64_t)blockIdx.y * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
__global__ __launch_bounds__(256) void bn_swish_fwd_kernel(const float *__restrict__ x, const float *__restrict__ msum,
                                                           const float *__restrict__ vsum, float ms, float vs,
                                                           const float *__restrict__ g, const float *__restrict__ b,
                                                           float *__restrict__ y, int64_t rows, int C, float eps) {
    FOR_EACH(i, rows * C) {
        const int c = (int)(i % C);
        const float u = (x[i] - msum[c] * ms) * rsqrtf(vsum[c] * vs + eps) * g[c] + b[c];
        y[i] = u * sigmoid_exact(u);
    }
}
// part[block][0][c] = sum du, part[block][1][c] = sum du xhat with du = dy swish'(u)
__global__ __launch_bounds__(256) void bn_swish_bwd_part_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                const float *__restrict__ msum,
                                                                const float *__restrict__ vsum, float ms, float vs,
                                                                const float *__restrict__ g, const float *__restrict__ b,
                                                                float *__restrict__ part, int64_t rows, int C, float eps) {
    __shared__ float red[4][2][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * BN_ROWS;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        const float mu = msum[c] * ms, rs = rsqrtf(vsum[c] * vs + eps), gc = g[c], bc = b[c];
        float xv[BN_ROWS / 4], dv[BN_ROWS / 4];
#pragma unroll
        for (int k = 0; k < BN_ROWS / 4; ++k) {       // branch-free, as above: 16 loads in flight per lane
            const int64_t r = r0 + w + 4 * k, rr = r < rows ? r : r0;
            xv[k] = x[rr * C + c];
            dv[k] = dy[rr * C + c];
        }
#pragma unroll
        for (int k = 0; k < BN_ROWS / 4; ++k) {
            const float xh = (xv[k] - mu) * rs, u = xh * gc + bc, s = sigmoid_exact(u);
            const float du = r0 + w + 4 * k < rows ? dv[k] * s * (1.f + u * (1.f - s)) : 0.f;
            a0 += du;
            a1 += du * xh;
        }
    }
    red[w][0][lane] = a0;
    red[w][1][lane] = a1;
    __syncthreads();
    if (w == 0 && c < C) {
        float *out = part + (int64_t)blockIdx.y * 2 * C;
        out[c] = (red[0][0][lane] + red[1][0][lane]) + (red[2][0][lane] + red[3][0][lane]);
        out[C + c] = (red[0][1][lane] + red[1][1][lane]) + (red[2][1][lane] + red[3][1][lane]);
    }
}
// training: dx = gamma rstd (du - sum(du)/M - xhat sum(du xhat)/M); eval (sums == NULL): dx = gamma rstd du
__global__ __launch_bounds__(256) void bn_swish_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                           const float *__restrict__ msum, const float *__restrict__ vsum,
                                                           float ms, float vs, const float *__restrict__ g,
                                                           const float *__restrict__ b, const float *__restrict__ sums,
                                                           float *__restrict__ dx, int64_t rows, int C, float eps) {
    const float inv = 1.0f / (float)rows;
    FOR_EACH(i, rows * C) {
        const int c = (int)(i % C);
        const float rs = rsqrtf(vsum[c] * vs + eps), gc = g[c];
        const float xh = (x[i] - msum[c] * ms) * rs, u = xh * gc + b[c], s = sigmoid_exact(u);
        float du = dy[i] * s * (1.f + u * (1.f - s));
        if (sums) du -= (sums[c] + xh * sums[C + c]) * inv;
        dx[i] = gc * rs * du;
    }
}

extern "C" int64_t asr_bn_blocks(int64_t rows) { return rows > 0 ? (rows + BN_ROWS - 1) / BN_ROWS : 1; }
extern "C" int asr_bn_var_part(const float *x, const float *msum, float ms, float *part, int64_t rows, int C, void *stream) {
    ASR_CHECK_ARG(x && msum && part && rows > 0 && C > 0, "bn_var_part: bad argument");
    ASR_CHECK_ARG(asr_bn_blocks(rows) <= 65535, "bn_var_part: too many rows");
    hipLaunchKernelGGL(bn_var_part_kernel, dim3(cdiv(C, 64), (unsigned)asr_bn_blocks(rows)), dim3(256), 0,
                       (hipStream_t)stream, x, msum, ms, part, rows, C);
    ASR_CHECK_LAUNCH("bn_var_part_kernel");
    return ASR_OK;
}
extern "C" int asr_bn_swish_fwd(const float *x, const float *msum, const float *vsum, float ms, float vs,
                                const float *gamma, const float *beta, float *y, int64_t rows, int C, float eps,
                                void *stream) {
    ASR_CHECK_ARG(x && msum && vsum && gamma && beta && y && rows > 0 && C > 0, "bn_swish_fwd: bad argument");
    hipLaunchKernelGGL(bn_swish_fwd_kernel, dim3(GRID1D(rows * C)), dim3(256), 0, (hipStream_t)stream, x, msum, vsum, ms,
                       vs, gamma, beta, y, rows, C, eps);
    ASR_CHECK_LAUNCH("bn_swish_fwd_kernel");
    return ASR_OK;
}
extern "C" int asr_bn_swish_bwd_part(const float *dy, const float *x, const float *msum, const float *vsum, float ms,
                                     float vs, const float *gamma, const float *beta, float *part, int64_t rows, int C,
                                     float eps, void *stream) {
    ASR_CHECK_ARG(dy && x && msum && vsum && gamma && beta && part && rows > 0 && C > 0, "bn_swish_bwd_part: bad argument");
    ASR_CHECK_ARG(asr_bn_blocks(rows) <= 65535, "bn_swish_bwd_part: too many rows");
    hipLaunchKernelGGL(bn_swish_bwd_part_kernel, dim3(cdiv(C, 64), (unsigned)asr_bn_blocks(rows)), dim3(256), 0,
                       (hipStream_t)stream, dy, x, msum, vsum, ms, vs, gamma, beta, part, rows, C, eps);
    ASR_CHECK_LAUNCH("bn_swish_bwd_part_kernel");
    return ASR_OK;
}
extern "C" int asr_bn_swish_bwd(const float *dy, const float *x, const float *msum, const float *vsum, float ms, float vs,
                                const float *gamma, const float *beta, const float *sums, float *dx, int64_t rows,
                                int C, float eps, void *stream) {
    ASR_CHECK_ARG(dy && x && msum && vsum && gamma && beta && dx && rows > 0 && C > 0, "bn_swish_bwd: bad argument");
    hipLaunchKernelGGL(bn_swish_bwd_kernel, dim3(GRID1D(rows * C)), dim3(256), 0, (hipStream_t)stream, dy, x, msum, vsum,
                       ms, vs, gamma, beta, sums, dx, rows, C, eps);
    ASR_CHECK_LAUNCH("bn_swish_bwd_kernel");
    return ASR_OK;
}

// ------------------------------------- rel_shift + masked softmax (:99-102,110-137) --
// attn[b,h,i,:] = softmax_j( keep[b,j] ? (ac[b,h,i,j] + shift(bd)[b,h,i,j]) * scale : -1e4 ), in place
// over ac.  shift(bd)[i,j] = bd[r, c-1] (0 when c == 0) with (r, c) = divmod((i+1) T + j, T+1): the
// reference's pad / view / slice trick read as an index map.  One wave per query row.
#define SM_MAXV 16
__global__ __launch_bounds__(256) void relattn_softmax_fwd_kernel(float *__restrict__ ac, const float *__restrict__ bd,
                                                                  const uint8_t *__restrict__ keep, float scale, int H,
                                                                  int T, int ld, int64_t nrows, float *__restrict__ ad,
                                                                  uint8_t *__restrict__ dmask, uint32_t thresh, float dscale,
                                                                  uint64_t seed) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= nrows) return;
    const int i = (int)(row % T);
    const int64_t bh = row / T;
    const int b = (int)(bh / H);
    float *a = ac + row * ld;
    const float *m = bd + bh * (int64_t)T * ld;
    float v[SM_MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < SM_MAXV; ++u) {
        const int j = lane + u * 64;
        v[u] = -INFINITY;
        if (j < T) {
            const int flat = (i + 1) * T + j;
            const int r = flat / (T + 1), c = flat - r * (T + 1);
            const float sh = c ? m[(int64_t)r * ld + c - 1] : 0.f;
            v[u] = keep[(int64_t)b * T + j] ? (a[j] + sh) * scale : -1e4f;
            mx = fmaxf(mx, v[u]);
        }
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < SM_MAXV; ++u) {
        const int j = lane + u * 64;
        v[u] = j < T ? __expf(v[u] - mx) : 0.f;
        s += v[u];
    }
    const float inv = 1.0f / wave_sum(s);
#pragma unroll
    for (int u = 0; u < SM_MAXV; ++u) {
        const int j = lane + u * 64;
        if (j >= ld) continue;
        const float p = j < T ? v[u] * inv : 0.f;          // the padding columns of the row stay zero
        a[j] = p;
        if (ad) {                                         // the dropout on the map (:133-134) in the same pass
            const int64_t e = row * ld + j;
            const bool kp = asr_keep(seed, e, thresh);
            dmask[e] = kp;
            ad[e] = kp ? p * dscale : 0.f;
        }
    }
}
// ds = attn (dattn - sum_j attn dattn) * scale, zero at masked keys; in place over dattn (= d ac).  The same pass
// writes d bd through the shift's index map (ds[i, j] -> dbd[r, c - 1] with (r, c) = divmod((i + 1) T + j, T + 1),
// nothing for c = 0: every element of dbd is the image of at most one (i, j), two contiguous runs per query row) and
// zero-fills what no (i, j) maps to -- the first T - 1 entries of buffer row 0 and the padding columns -- instead of a
// second kernel gathering it from ds (72 MB read + 72 MB written per layer at config 5).
__global__ __launch_bounds__(256) void relattn_softmax_bwd_kernel(const float *__restrict__ attn, float *__restrict__ dattn,
                                                                  float *__restrict__ dbd,
                                                                  const uint8_t *__restrict__ keep, float scale, int H,
                                                                  int T, int ld, int64_t nrows,
                                                                  const uint8_t *__restrict__ dmask, float dscale) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= nrows) return;
    const int b = (int)(row / T / H);
    const int i = (int)(row % T);
    float *m = dbd + (row / T) * (int64_t)T * ld;
    const float *p = attn + row * ld;
    float *g = dattn + row * ld;
    float pv[SM_MAXV], gv[SM_MAXV];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < SM_MAXV; ++u) {
        const int j = lane + u * 64;
        pv[u] = j < T ? p[j] : 0.f;
        gv[u] = j < T ? g[j] : 0.f;
        if (dmask && j < T) gv[u] = dmask[row * ld + j] ? gv[u] * dscale : 0.f;   // gradient of the map's dropout
        s += pv[u] * gv[u];
    }
    s = wave_sum(s);
#pragma unroll
    for (int u = 0; u < SM_MAXV; ++u) {
        const int j = lane + u * 64;
        if (j < T) {
            const float ds = keep[(int64_t)b * T + j] ? pv[u] * (gv[u] - s) * scale : 0.f;
            g[j] = ds;
            const int flat = (i + 1) * T + j;
            const int r = flat / (T + 1), c = flat - r * (T + 1);
            if (c) m[(int64_t)r * ld + c - 1] = ds;
        } else if (j < ld) {
            g[j] = 0.f;
            m[(int64_t)i * ld + j] = 0.f;
        }
    }
    if (i == 0)
        for (int c = lane; c < T - 1; c += 64) m[c] = 0.f;
}
extern "C" int asr_relattn_softmax_fwd(float *ac, const float *bd, const uint8_t *keep, float scale, int B, int H, int T,
                                       int ld, float *attn_drop, uint8_t *drop_mask, float p, uint64_t seed, void *stream) {
    ASR_CHECK_ARG((attn_drop == nullptr) == (drop_mask == nullptr) && p >= 0.f && p < 1.f, "relattn_softmax_fwd: bad dropout");
    ASR_CHECK_ARG(ac && bd && keep, "relattn_softmax_fwd: null pointer");
    ASR_CHECK_ARG(ld >= T && ld <= 64 * SM_MAXV, "relattn_softmax_fwd: bad row stride");
    ASR_CHECK_ARG(B > 0 && H > 0 && T > 0 && T <= 64 * SM_MAXV, "relattn_softmax_fwd: T must be in [1, %d]", 64 * SM_MAXV);
    const int64_t nrows = (int64_t)B * H * T;
    hipLaunchKernelGGL(relattn_softmax_fwd_kernel, dim3(cdiv(nrows, 4)), dim3(256), 0, (hipStream_t)stream, ac, bd, keep,
                       scale, H, T, ld, nrows, attn_drop, drop_mask, drop_thresh(p), 1.0f / (1.0f - p), seed);
    ASR_CHECK_LAUNCH("relattn_softmax_fwd_kernel");
    return ASR_OK;
}
extern "C" int asr_relattn_softmax_bwd(const float *attn, float *dattn, float *dbd, const uint8_t *keep, float scale,
                                       int B, int H, int T, int ld, const uint8_t *drop_mask, float p, void *stream) {
    ASR_CHECK_ARG(p >= 0.f && p < 1.f, "relattn_softmax_bwd: bad dropout");
    ASR_CHECK_ARG(attn && dattn && dbd && keep, "relattn_softmax_bwd: null pointer");
    ASR_CHECK_ARG(ld >= T && ld <= 64 * SM_MAXV, "relattn_softmax_bwd: bad row stride");
    ASR_CHECK_ARG(B > 0 && H > 0 && T > 0 && T <= 64 * SM_MAXV, "relattn_softmax_bwd: T must be in [1, %d]", 64 * SM_MAXV);
    const int64_t nrows = (int64_t)B * H * T;
    hipLaunchKernelGGL(relattn_softmax_bwd_kernel, dim3(cdiv(nrows, 4)), dim3(256), 0, (hipStream_t)stream, attn, dattn,
                       dbd, keep, scale, H, T, ld, nrows, drop_mask, 1.0f / (1.0f - p));
    ASR_CHECK_LAUNCH("relattn_softmax_bwd_kernel");
    return ASR_OK;
}

// --------------------------------- 3x3 stride-2 patches (CNN_embedding, :386-407) --
// nn.Conv2d(Cin, Cout, 3, 2) without padding on a channels-last map x (B, Ti, Fi, Cin): row
// (b, to, fo) of col holds its 9 Cin inputs in (ki, kj, cin) order, To = (Ti-3)/2+1, Fo = (Fi-3)/2+1; the
// convolution itself is col @ W'^T on asr_sgemm with W' = the weight permuted to (Cout, 3, 3, Cin).
__global__ __launch_bounds__(256) void im2col_3x3s2_kernel(const float *__restrict__ x, float *__restrict__ col, int Ti,
                                                           int Fi, int Cin, int To, int Fo, int ld, int64_t n) {
    FOR_EACH(e0, n) {
        const int kk = (int)(e0 % ld);
        if (kk >= 9 * Cin) { col[e0] = 0.f; continue; }      // zero columns up to the padded row length
        const int64_t e = (e0 / ld) * 9 * Cin + kk;
        const int cin = (int)(e % Cin);
        int64_t q = e / Cin;
        const int kj = (int)(q % 3); q /= 3;
        const int ki = (int)(q % 3); q /= 3;
        const int fo = (int)(q % Fo); q /= Fo;
        const int to = (int)(q % To);
        const int64_t b = q / To;
        col[e0] = x[((b * Ti + 2 * to + ki) * Fi + 2 * fo + kj) * Cin + cin];
    }
}
// 16-byte version for Cin % 4 == 0 and ld == 9 Cin (the second convolution: 2.1 GB of patches)
__global__ __launch_bounds__(256) void im2col_3x3s2_v4_kernel(const float4 *__restrict__ x, float4 *__restrict__ col,
                                                              int Ti, int Fi, int C4, int To, int Fo, int64_t n4) {
    FOR_EACH(e, n4) {
        const int c4 = (int)(e % C4);
        int64_t q = e / C4;
        const int kj = (int)(q % 3); q /= 3;
        const int ki = (int)(q % 3); q /= 3;
        const int fo = (int)(q % Fo); q /= Fo;
        const int to = (int)(q % To);
        const int64_t b = q / To;
        col[e] = x[((b * Ti + 2 * to + ki) * Fi + 2 * fo + kj) * C4 + c4];
    }
}
__global__ __launch_bounds__(256) void col2im_3x3s2_v4_kernel(const float4 *__restrict__ dcol, const float4 *__restrict__ y,
                                                              float4 *__restrict__ dx, int Ti, int Fi, int C4, int To,
                                                              int Fo, int64_t n4) {
    FOR_EACH(e, n4) {
        const int c = (int)(e % C4);
        int64_t q = e / C4;
        const int fi = (int)(q % Fi); q /= Fi;
        const int ti = (int)(q % Ti);
        const int64_t b = q / Ti;
        float4 acc = zero4();
#pragma unroll
        for (int ki = 0; ki < 3; ++ki) {
            const int t2 = ti - ki;
            if (t2 < 0 || (t2 & 1) || t2 / 2 >= To) continue;
#pragma unroll
            for (int kj = 0; kj < 3; ++kj) {
                const int f2 = fi - kj;
                if (f2 < 0 || (f2 & 1) || f2 / 2 >= Fo) continue;
                const float4 v = dcol[(((b * To + t2 / 2) * Fo + f2 / 2) * 9 + ki * 3 + kj) * C4 + c];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        if (y) {
            const float4 m = y[e];
            acc.x = m.x > 0.f ? acc.x : 0.f; acc.y = m.y > 0.f ? acc.y : 0.f;
            acc.z = m.z > 0.f ? acc.z : 0.f; acc.w = m.w > 0.f ? acc.w : 0.f;
        }
        dx[e] = acc;
    }
}
// The same gather for a map that is the FIRST convolution's output (one input channel, no gradient w.r.t. the
// features): d pre-activation = relu'(y) * col2im(dcol) is not written at all -- the only readers would be the
// first convolution's weight gradient dW1[c][tap] = sum_pos d pre[pos][c] * taps[pos][tap] and its bias gradient,
// each a pass over the 957 MB map at config 5.  A thread keeps the 9 + 1 sums of its 4 channels in registers over
// its grid-stride positions (its channel group is fixed: the stride is a multiple of C4), the block adds the
// threads that share a channel group and writes one row of partials [C][10] (9 taps, then the bias): a column sum
// over the rows finishes the job (asr_colsum).  taps: (B*Ti*Fi, ldt) rows of the first convolution's patches.
#define C2I_MAXB 2048
__global__ __launch_bounds__(256) void col2im_dw_v4_kernel(const float4 *__restrict__ dcol, const float4 *__restrict__ y,
                                                           const float *__restrict__ taps, int ldt,
                                                           float *__restrict__ part, int Ti, int Fi, int C4, int To,
                                                           int Fo, int64_t n4, bool vect) {
    __shared__ float red[256][41];
    float4 sum[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) sum[k] = zero4();
    FOR_EACH(e, n4) {
        int64_t q = e / C4;
        const int64_t pos = q;
        const int c = (int)(e - q * C4);
        const int fi = (int)(q % Fi); q /= Fi;
        const int ti = (int)(q % Ti);
        const int64_t b = q / Ti;
        float4 acc = zero4();
#pragma unroll
        for (int ki = 0; ki < 3; ++ki) {
            const int t2 = ti - ki;
            if (t2 < 0 || (t2 & 1) || t2 / 2 >= To) continue;
#pragma unroll
            for (int kj = 0; kj < 3; ++kj) {
                const int f2 = fi - kj;
                if (f2 < 0 || (f2 & 1) || f2 / 2 >= Fo) continue;
                const float4 v = dcol[(((b * To + t2 / 2) * Fo + f2 / 2) * 9 + ki * 3 + kj) * C4 + c];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        const float4 m = y[e];
        acc.x = m.x > 0.f ? acc.x : 0.f; acc.y = m.y > 0.f ? acc.y : 0.f;
        acc.z = m.z > 0.f ? acc.z : 0.f; acc.w = m.w > 0.f ? acc.w : 0.f;
        const float *tp = taps + pos * ldt;
        float tv[12];
        if (vect) {      // rows of 12 floats, 16-byte aligned (host-checked): three loads instead of nine
            const float4 t0 = ld4(tp), t1 = ld4(tp + 4), t2 = ld4(tp + 8);
            tv[0] = t0.x; tv[1] = t0.y; tv[2] = t0.z; tv[3] = t0.w; tv[4] = t1.x; tv[5] = t1.y; tv[6] = t1.z; tv[7] = t1.w;
            tv[8] = t2.x; tv[9] = t2.y; tv[10] = t2.z; tv[11] = t2.w;
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) tv[k] = tp[k];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float t = tv[k];
            sum[k].x += acc.x * t; sum[k].y += acc.y * t; sum[k].z += acc.z * t; sum[k].w += acc.w * t;
        }
        sum[9].x += acc.x; sum[9].y += acc.y; sum[9].z += acc.z; sum[9].w += acc.w;
    }
    // threads tid, tid + C4, ... share a channel group (256 % C4 == 0, host-checked)
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        red[tid][4 * k] = sum[k].x; red[tid][4 * k + 1] = sum[k].y; red[tid][4 * k + 2] = sum[k].z; red[tid][4 * k + 3] = sum[k].w;
    }
    __syncthreads();
    const int C = 4 * C4;
    for (int i = tid; i < C * 10; i += 256) {
        const int ch = i / 10, k = i - ch * 10;          // output element [ch][k]
        const int g = ch >> 2, l = ch & 3;
        float v = 0.f;
        for (int t = g; t < 256; t += C4) v += red[t][4 * k + l];
        part[(int64_t)blockIdx.x * C * 10 + i] = v;
    }
}
// the data gradient as a gather: dx[b,ti,fi,c] = sum over the (ki, kj) whose output position exists
__global__ __launch_bounds__(256) void col2im_3x3s2_kernel(const float *__restrict__ dcol, const float *__restrict__ y,
                                                           float *__restrict__ dx, int Ti, int Fi, int Cin, int To, int Fo,
                                                           int64_t n) {
    FOR_EACH(e, n) {
        const int c = (int)(e % Cin);
        int64_t q = e / Cin;
        const int fi = (int)(q % Fi); q /= Fi;
        const int ti = (int)(q % Ti);
        const int64_t b = q / Ti;
        float acc = 0.f;
        if (y && !(y[e] > 0.f)) { dx[e] = 0.f; continue; }   // gradient of the ReLU that produced this map
#pragma unroll
        for (int ki = 0; ki < 3; ++ki) {
            const int t2 = ti - ki;
            if (t2 < 0 || (t2 & 1) || t2 / 2 >= To) continue;
#pragma unroll
            for (int kj = 0; kj < 3; ++kj) {
                const int f2 = fi - kj;
                if (f2 < 0 || (f2 & 1) || f2 / 2 >= Fo) continue;
                acc += dcol[(((b * To + t2 / 2) * Fo + f2 / 2) * 9 + ki * 3 + kj) * Cin + c];
            }
        }
        dx[e] = acc;
    }
}
extern "C" int asr_im2col_3x3s2(const float *x, float *col, int B, int Ti, int Fi, int Cin, int ld, void *stream) {
    ASR_CHECK_ARG(x && col && B > 0 && Ti >= 3 && Fi >= 3 && Cin > 0 && ld >= 9 * Cin, "im2col_3x3s2: bad argument");
    const int To = (Ti - 3) / 2 + 1, Fo = (Fi - 3) / 2 + 1;
    const int64_t n = (int64_t)B * To * Fo * ld;
    if (Cin % 4 == 0 && ld == 9 * Cin && (((uintptr_t)x | (uintptr_t)col) & 15) == 0) {
        hipLaunchKernelGGL(im2col_3x3s2_v4_kernel, dim3(GRID1D(n / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)x, (float4 *)col, Ti, Fi, Cin / 4, To, Fo, n / 4);
        ASR_CHECK_LAUNCH("im2col_3x3s2_v4_kernel");
        return ASR_OK;
    }
    hipLaunchKernelGGL(im2col_3x3s2_kernel, dim3(GRID1D(n)), dim3(256), 0, (hipStream_t)stream, x, col, Ti, Fi, Cin, To,
                       Fo, ld, n);
    ASR_CHECK_LAUNCH("im2col_3x3s2_kernel");
    return ASR_OK;
}
extern "C" int asr_col2im_dw_blocks(int B, int Ti, int Fi, int Cin) {
    const int64_t n4 = (int64_t)B * Ti * Fi * Cin / 4;
    return (int)std::min<int64_t>((n4 + 255) / 256, C2I_MAXB);
}
extern "C" int asr_col2im_3x3s2_dw(const float *dcol, const float *y, const float *taps, int ldt, float *part, int B,
                                   int Ti, int Fi, int Cin, void *stream) {
    ASR_CHECK_ARG(dcol && y && taps && part && B > 0 && Ti >= 3 && Fi >= 3 && ldt >= 9, "col2im_3x3s2_dw: bad argument");
    ASR_CHECK_ARG(Cin % 4 == 0 && Cin <= 1024 && 256 % (Cin / 4) == 0 && (((uintptr_t)dcol | (uintptr_t)y) & 15) == 0,
                  "col2im_3x3s2_dw: Cin must be a multiple of 4 whose quarter divides 256, 16-byte aligned maps");
    const int To = (Ti - 3) / 2 + 1, Fo = (Fi - 3) / 2 + 1;
    const int64_t n4 = (int64_t)B * Ti * Fi * Cin / 4;
    hipLaunchKernelGGL(col2im_dw_v4_kernel, dim3(asr_col2im_dw_blocks(B, Ti, Fi, Cin)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)dcol, (const float4 *)y, taps, ldt, part, Ti, Fi, Cin / 4, To, Fo, n4,
                       ldt >= 12 && ldt % 4 == 0 && (((uintptr_t)taps) & 15) == 0);
    ASR_CHECK_LAUNCH("col2im_dw_v4_kernel");
    return ASR_OK;
}
extern "C" int asr_col2im_3x3s2(const float *dcol, const float *y, float *dx, int B, int Ti, int Fi, int Cin,
                                void *stream) {
    ASR_CHECK_ARG(dcol && dx && B > 0 && Ti >= 3 && Fi >= 3 && Cin > 0, "col2im_3x3s2: bad argument");
    const int To = (Ti - 3) / 2 + 1, Fo = (Fi - 3) / 2 + 1;
    const int64_t n = (int64_t)B * Ti * Fi * Cin;
    if (Cin % 4 == 0 && (((uintptr_t)dcol | (uintptr_t)dx | (uintptr_t)y) & 15) == 0) {
        hipLaunchKernelGGL(col2im_3x3s2_v4_kernel, dim3(GRID1D(n / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)dcol, (const float4 *)y, (float4 *)dx, Ti, Fi, Cin / 4, To, Fo, n / 4);
        ASR_CHECK_LAUNCH("col2im_3x3s2_v4_kernel");
        return ASR_OK;
    }
    hipLaunchKernelGGL(col2im_3x3s2_kernel, dim3(GRID1D(n)), dim3(256), 0, (hipStream_t)stream, dcol, y, dx, Ti, Fi, Cin,
                       To, Fo, n);
    ASR_CHECK_LAUNCH("col2im_3x3s2_kernel");
    return ASR_OK;
}
