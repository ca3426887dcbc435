// Global-norm gradient clipping + Adam (coupled L2 weight decay) over a list of tensors:
// the caller-side tail of the training step, train.py:60-63 with the optimizer of
// train.py:146 (torch.nn.utils.clip_grad_norm_(model.parameters(), hp.clip) followed by
// torch.optim.Adam(lr, weight_decay=1e-5).step()).  Two launches for the norm (deterministic
// two-stage sum), one for the update of all tensors; no host synchronisation anywhere --
// the clip coefficient stays on the device.  HBM-bound: 8 floats of traffic per parameter.
#include "common.h"

#define OPT_BLOCKS 32   // workgroups per tensor

// stage 1: part[t * OPT_BLOCKS + bx] = sum of squares of this block's share of tensor t
__global__ __launch_bounds__(256) void sqnorm_part_kernel(asr_tensor_list g, float *__restrict__ part) {
    __shared__ float red[4];
    const int t = blockIdx.y;
    const float *p = g.p[t];
    const int64_t n = g.n[t];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)OPT_BLOCKS * 256) {
        const float v = p[i];
        acc += v * v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[t * OPT_BLOCKS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// stage 2 (one workgroup): out[0] (+)= sum of the partials, in a fixed order
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float *__restrict__ part, int nparts,
                                                           float *__restrict__ out, int accumulate) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = (red[0] + red[1]) + (red[2] + red[3]);
        out[0] = accumulate ? out[0] + s : s;
    }
}

// One pass over every tensor.  Per element, in torch's order of operations:
//   g  = grad * clip_coef            clip_coef = min(1, max_norm / (sqrt(norm2) + 1e-6))   [clip_grad_norm_]
//   g += weight_decay * p                                                                 [Adam, coupled L2]
//   m  = beta1 m + (1 - beta1) g ;  v = beta2 v + (1 - beta2) g^2
//   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// The clipped gradient is written back (clip_grad_norm_ scales .grad in place).
__global__ __launch_bounds__(256) void clip_adam_kernel(asr_tensor_list P, asr_tensor_list G, asr_tensor_list M,
                                                        asr_tensor_list V, const float *__restrict__ norm2,
                                                        float max_norm, float step_size, float inv_bc2_sqrt,
                                                        float beta1, float omb1, float beta2, float omb2,
                                                        float eps, float weight_decay) {
    const int t = blockIdx.y;
    float *__restrict__ p = P.p[t];
    float *__restrict__ g = G.p[t];
    float *__restrict__ m = M.p[t];
    float *__restrict__ v = V.p[t];
    const int64_t n = P.n[t];
    float coef = 1.f;
    if (max_norm > 0.f) {
        const float tn = sqrtf(norm2[0]);
        // torch.clamp(clip_coef, max=1.0): a NaN norm (a poisoned gradient, asr_poison_if_flag) gives a NaN
        // coefficient and with it NaN gradients and parameters, as clip_grad_norm_ does; fminf would drop it
        const float c = max_norm / (tn + 1e-6f);
        coef = c < 1.0f ? c : (c == c ? 1.0f : c);
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)OPT_BLOCKS * 256) {
        const float pi = p[i];
        const float gc = g[i] * coef;
        const float gi = gc + weight_decay * pi;
        const float mi = beta1 * m[i] + omb1 * gi;
        const float vi = beta2 * v[i] + omb2 * gi * gi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        g[i] = gc;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / denom);
    }
}

static int check_list(const asr_tensor_list *l, const char *what) {
    ASR_CHECK_ARG(l != nullptr, "%s: null tensor list", what);
    ASR_CHECK_ARG(l->count >= 0 && l->count <= ASR_TENSOR_LIST_MAX, "%s: %d tensors (max %d per call)", what,
                  l->count, ASR_TENSOR_LIST_MAX);
    for (int i = 0; i < l->count; ++i)
        ASR_CHECK_ARG(l->p[i] != nullptr && l->n[i] >= 0, "%s: tensor %d is null or has a negative size", what, i);
    return ASR_OK;
}

extern "C" int64_t asr_sqnorm_ws_floats(void) { return (int64_t)ASR_TENSOR_LIST_MAX * OPT_BLOCKS + 64; }

extern "C" int asr_grad_sqnorm(const asr_tensor_list *grads, float *norm2, int accumulate, float *ws,
                               void *stream) {
    ASR_CHECK(check_list(grads, "grad_sqnorm"));
    ASR_CHECK_ARG(norm2 && ws, "grad_sqnorm: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (grads->count == 0) {
        if (!accumulate) ASR_CHECK_HIP(hipMemsetAsync(norm2, 0, sizeof(float), st));
        return ASR_OK;
    }
    hipLaunchKernelGGL(sqnorm_part_kernel, dim3(OPT_BLOCKS, grads->count), dim3(256), 0, st, *grads, ws);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, (const float *)ws,
                       grads->count * OPT_BLOCKS, norm2, accumulate);
    ASR_CHECK_LAUNCH("sqnorm kernels");
    return ASR_OK;
}

extern "C" int asr_clip_adam_step(const asr_tensor_list *params, const asr_tensor_list *grads,
                                  const asr_tensor_list *exp_avg, const asr_tensor_list *exp_avg_sq,
                                  const float *norm2, double max_norm, double lr, double beta1, double beta2,
                                  double eps, double weight_decay, int64_t step, void *stream) {
    ASR_CHECK(check_list(params, "clip_adam_step"));
    ASR_CHECK(check_list(grads, "clip_adam_step"));
    ASR_CHECK(check_list(exp_avg, "clip_adam_step"));
    ASR_CHECK(check_list(exp_avg_sq, "clip_adam_step"));
    ASR_CHECK_ARG(grads->count == params->count && exp_avg->count == params->count &&
                      exp_avg_sq->count == params->count,
                  "clip_adam_step: the four lists must have the same length");
    for (int i = 0; i < params->count; ++i)
        ASR_CHECK_ARG(grads->n[i] == params->n[i] && exp_avg->n[i] == params->n[i] && exp_avg_sq->n[i] == params->n[i],
                      "clip_adam_step: tensor %d has different sizes in the four lists", i);
    ASR_CHECK_ARG(step >= 1, "clip_adam_step: step counts from 1 (got %lld)", (long long)step);
    ASR_CHECK_ARG(max_norm <= 0.0 || norm2 != nullptr, "clip_adam_step: clipping needs the squared norm");
    ASR_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && lr >= 0.0,
                  "clip_adam_step: bad hyper-parameter");
    if (params->count == 0) return ASR_OK;
    // hyper-parameters arrive as doubles (Python floats in torch.optim.Adam); every derived
    // constant is formed in double and rounded once, as torch's scalar path does
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(clip_adam_kernel, dim3(OPT_BLOCKS, params->count), dim3(256), 0, (hipStream_t)stream,
                       *params, *grads, *exp_avg, *exp_avg_sq, norm2, (float)max_norm, step_size, inv_bc2_sqrt,
                       (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)weight_decay);
    ASR_CHECK_LAUNCH("clip_adam_kernel");
    return ASR_OK;
}
