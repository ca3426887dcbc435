// Fused log-softmax + label-smoothed cross-entropy, forward and gradient in one
// pass over the logits (Loss/label_smoothing.py:14-28).
//
// HBM-bound: per live row the algorithmic traffic is V*4 bytes read (logits) +
// V*4 bytes written (gradient); padded rows (t >= text_length) write zeros only.
// One 256-thread block per (b,l) row keeps the whole row in registers between
// the reduction and the gradient write, so the logits are read exactly once.
//
//   loss_row = -(1-eps) lp_y - eps/(V-1) (sum_v z_v - V lse - lp_y)
//   dz_v     = w (softmax(z)_v - q_v),  q_y = 1-eps, q_v = eps/(V-1) otherwise
//   w        = 1/t_i (T_norm) [/B (B_norm)] for l < t_i, else 0
#include "common.h"

#define LS_THREADS 256

__device__ __forceinline__ float block_reduce_sum(float v, float *sh) {
    v = wave_sum(v);
    int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < LS_THREADS / 64; ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float block_reduce_max(float v, float *sh) {
    v = wave_max(v);
    int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int i = 1; i < LS_THREADS / 64; ++i) r = fmaxf(r, sh[i]);
    return r;
}

// NV4 = float4 chunks per thread held in registers (row must satisfy V%4==0,
// V <= 4*256*NV4).  NV4 == 0: generic scalar path that re-reads the row (L2).
template <int NV4>
__global__ __launch_bounds__(LS_THREADS) void ls_ce_kernel(
    const float *__restrict__ z, const int64_t *__restrict__ tgt,
    const int64_t *__restrict__ tlen, float *__restrict__ row_loss, float *__restrict__ dz,
    int L, int V, float eps, int T_norm, float bscale) {
    __shared__ float sh[LS_THREADS / 64];
    const int row = blockIdx.x;
    const int b = row / L, l = row % L;
    const int tid = threadIdx.x;
    const int64_t tl = tlen[b];
    const float *zr = z + (int64_t)row * V;
    float *dr = dz ? dz + (int64_t)row * V : nullptr;
    if (l >= tl) {  // padded label position: no loss, zero gradient
        if (tid == 0) row_loss[row] = 0.f;
        if (dr) {
            if (NV4 > 0) {
                for (int i = tid; i < V / 4; i += LS_THREADS) ((float4 *)dr)[i] = make_float4(0, 0, 0, 0);
            } else {
                for (int i = tid; i < V; i += LS_THREADS) dr[i] = 0.f;
            }
        }
        return;
    }
    const int y = (int)tgt[row];
    const float off = eps / (float)(V - 1);
    const float w = bscale * (T_norm ? 1.0f / (float)tl : 1.0f);

    if (NV4 > 0) {
        const int n4 = V / 4;
        float4 v[NV4 > 0 ? NV4 : 1];
        float mx = -INFINITY, sz = 0.f;
#pragma unroll
        for (int k = 0; k < NV4; ++k) {
            int i = tid + k * LS_THREADS;
            if (i < n4) {
                v[k] = ((const float4 *)zr)[i];
                mx = fmaxf(mx, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
                sz += (v[k].x + v[k].y) + (v[k].z + v[k].w);
            }
        }
        mx = block_reduce_max(mx, sh);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < NV4; ++k) {
            int i = tid + k * LS_THREADS;
            if (i < n4) {
                v[k].x = expf(v[k].x - mx); v[k].y = expf(v[k].y - mx);
                v[k].z = expf(v[k].z - mx); v[k].w = expf(v[k].w - mx);
                se += (v[k].x + v[k].y) + (v[k].z + v[k].w);
            }
        }
        se = block_reduce_sum(se, sh);
        sz = block_reduce_sum(sz, sh);
        const float lse = mx + logf(se);
        if (tid == 0) {
            float lpy = zr[y] - lse;
            row_loss[row] = w * (-(1.0f - eps) * lpy - off * (sz - (float)V * lse - lpy));
        }
        if (dr) {
            const float inv = w / se;
            const float woff = w * off;
#pragma unroll
            for (int k = 0; k < NV4; ++k) {
                int i = tid + k * LS_THREADS;
                if (i < n4) {
                    float4 o;
                    o.x = v[k].x * inv - woff; o.y = v[k].y * inv - woff;
                    o.z = v[k].z * inv - woff; o.w = v[k].w * inv - woff;
                    int c = y - 4 * i;
                    const float fix = w * (off - (1.0f - eps));  // q_y instead of off
                    if (c == 0) o.x += fix; else if (c == 1) o.y += fix;
                    else if (c == 2) o.z += fix; else if (c == 3) o.w += fix;
                    ((float4 *)dr)[i] = o;
                }
            }
        }
    } else {
        float mx = -INFINITY, sz = 0.f;
        for (int i = tid; i < V; i += LS_THREADS) { float a = zr[i]; mx = fmaxf(mx, a); sz += a; }
        mx = block_reduce_max(mx, sh);
        float se = 0.f;
        for (int i = tid; i < V; i += LS_THREADS) se += expf(zr[i] - mx);
        se = block_reduce_sum(se, sh);
        sz = block_reduce_sum(sz, sh);
        const float lse = mx + logf(se);
        if (tid == 0) {
            float lpy = zr[y] - lse;
            row_loss[row] = w * (-(1.0f - eps) * lpy - off * (sz - (float)V * lse - lpy));
        }
        if (dr) {
            const float inv = w / se;
            for (int i = tid; i < V; i += LS_THREADS) {
                float q = (i == y) ? (1.0f - eps) : off;
                dr[i] = expf(zr[i] - mx) * inv - w * q;
            }
        }
    }
}

// deterministic single-block sum of the per-row losses
__global__ __launch_bounds__(LS_THREADS) void ls_sum_kernel(const float *__restrict__ rows, int n,
                                                             float *__restrict__ out) {
    __shared__ double sh[LS_THREADS];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += LS_THREADS) acc += (double)rows[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = LS_THREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sh[0];
}

extern "C" int asr_ls_ce_fwd_bwd(const float *logits, const int64_t *targets,
                                 const int64_t *text_lengths, int B, int L, int V, float eps,
                                 int T_norm, int B_norm, float *row_loss, float *loss,
                                 float *dlogits, void *stream) {
    ASR_CHECK_ARG(logits && targets && text_lengths && row_loss && loss, "ls_ce: null pointer");
    ASR_CHECK_ARG(B > 0 && L > 0 && V > 1, "ls_ce: bad shape B=%d L=%d V=%d", B, L, V);
    hipStream_t st = (hipStream_t)stream;
    const int rows = B * L;
    const float bscale = B_norm ? 1.0f / (float)B : 1.0f;
    const bool vec = (V % 4 == 0) && (((uintptr_t)logits & 15) == 0) &&
                     (!dlogits || ((uintptr_t)dlogits & 15) == 0);
    const int n4 = V / 4;
#define LS_LAUNCH(NV4) \
    hipLaunchKernelGGL((ls_ce_kernel<NV4>), dim3(rows), dim3(LS_THREADS), 0, st, logits, targets, \
                       text_lengths, row_loss, dlogits, L, V, eps, T_norm, bscale)
    if (vec && n4 <= LS_THREADS * 1) LS_LAUNCH(1);
    else if (vec && n4 <= LS_THREADS * 2) LS_LAUNCH(2);
    else if (vec && n4 <= LS_THREADS * 4) LS_LAUNCH(4);
    else if (vec && n4 <= LS_THREADS * 8) LS_LAUNCH(8);
    else LS_LAUNCH(0);
#undef LS_LAUNCH
    ASR_CHECK_LAUNCH("ls_ce_kernel");
    hipLaunchKernelGGL(ls_sum_kernel, dim3(1), dim3(LS_THREADS), 0, st, row_loss, rows, loss);
    ASR_CHECK_LAUNCH("ls_sum_kernel");
    return ASR_OK;
}
