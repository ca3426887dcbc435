// Device-side input pipeline of the training step: what the
// reference does per utterance on the host in datasets.py:110-120 (mean/variance normalisation,
// SpecAugment: frequency masks filled with the utterance mean, then time masks filled with 0),
// in collate_fn (datasets.py:225-247: zero padding to the longest utterance) and in
// utils/utils.py:47-65 (frame stacking), as three launches over the raw padded batch:
//   feat_colsum_kernel  per (utterance, 64-frame chunk): column sums of the normalised frames
//   feat_plan_kernel    per utterance: the fill value of each frequency mask -- the mean of the
//                       utterance AS MASKED SO FAR (datasets.py:62 calls cloned.mean() inside the
//                       mask loop), from the column sums, in double, in a fixed order
//   feat_apply_kernel   elementwise: normalise, apply the masks in the reference's order, zero the
//                       padding, write the stacked layout (B, T/stack, F*stack) the encoder reads
// HBM-bound: the raw batch is read twice (sums, apply) and the stacked batch written once.
// The mask positions come from the host (the reference draws them with Python's `random`;
// asr_amd/utils/features.py replays the same sequence of randrange calls).
#include "common.h"

#define FEAT_CH 64   // frames per partial-sum chunk

__global__ __launch_bounds__(256) void feat_colsum_kernel(const float *__restrict__ raw,
                                                          const int32_t *__restrict__ mel_len,
                                                          const float *__restrict__ mean,
                                                          const float *__restrict__ var, int Tm, int F, int nchunk,
                                                          float *__restrict__ part /* [B][nchunk][F] */) {
    const int b = blockIdx.y, ch = blockIdx.x;
    const int len = min(max(mel_len[b], 0), Tm);
    const int t0 = ch * FEAT_CH, t1 = min(len, t0 + FEAT_CH);
    for (int f = threadIdx.x; f < F; f += 256) {
        const float mu = mean ? mean[f] : 0.f, sd = var ? sqrtf(var[f]) : 1.f;
        float acc = 0.f;
        for (int t = t0; t < t1; ++t) {
            float v = raw[((int64_t)b * Tm + t) * F + f];
            if (mean) v = (v - mu) / sd;     // the division the reference does (datasets.py:111-112)
            acc += v;
        }
        part[((int64_t)b * nchunk + ch) * F + f] = acc;
    }
}

__global__ __launch_bounds__(256) void feat_plan_kernel(const float *__restrict__ part,
                                                        const int32_t *__restrict__ mel_len,
                                                        const int32_t *__restrict__ fmask /* [B][nF][2] */, int nF,
                                                        int Tm, int F, int nchunk, int fill_zero,
                                                        float *__restrict__ fill /* [B][nF] */) {
    extern __shared__ double cs[];   // [F]
    const int b = blockIdx.x;
    const int len = min(max(mel_len[b], 0), Tm);
    for (int f = threadIdx.x; f < F; f += 256) {
        double acc = 0.0;
        for (int c = 0; c < nchunk; ++c) acc += (double)part[((int64_t)b * nchunk + c) * F + f];
        cs[f] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int f = 0; f < F; ++f) total += cs[f];
        const double cnt = (double)len * (double)F;
        for (int i = 0; i < nF; ++i) {
            const int f0 = fmask[((int64_t)b * nF + i) * 2], f1 = fmask[((int64_t)b * nF + i) * 2 + 1];
            const float m = (fill_zero || cnt == 0.0) ? 0.f : (float)(total / cnt);
            fill[(int64_t)b * nF + i] = m;
            for (int f = max(f0, 0); f < min(f1, F); ++f) {
                const double nv = (double)len * (double)m;
                total += nv - cs[f];
                cs[f] = nv;
            }
        }
    }
}

__global__ __launch_bounds__(256) void feat_apply_kernel(
    const float *__restrict__ raw, const int32_t *__restrict__ mel_len, const float *__restrict__ mean,
    const float *__restrict__ var, const int32_t *__restrict__ fmask, int nF, const float *__restrict__ fill,
    const int32_t *__restrict__ tmask /* [B][nT][2] */, int nT, int Tm, int F, int stack, int Ts,
    float *__restrict__ out /* [B][Ts][F*stack] */) {
    const int b = blockIdx.y;
    const int len = min(max(mel_len[b], 0), Tm);
    const int64_t n = (int64_t)Ts * stack * F;       // the tail frames (Tm % stack) are dropped
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int t = (int)(e / F), f = (int)(e % F);
        float v = 0.f;
        if (t < len) {
            v = raw[((int64_t)b * Tm + t) * F + f];
            if (mean) v = (v - mean[f]) / sqrtf(var[f]);
            for (int i = 0; i < nF; ++i) {       // later masks overwrite earlier ones
                const int f0 = fmask[((int64_t)b * nF + i) * 2], f1 = fmask[((int64_t)b * nF + i) * 2 + 1];
                if (f >= f0 && f < f1) v = fill[(int64_t)b * nF + i];
            }
            for (int i = 0; i < nT; ++i) {
                const int a0 = tmask[((int64_t)b * nT + i) * 2], a1 = tmask[((int64_t)b * nT + i) * 2 + 1];
                if (t >= a0 && t < a1) v = 0.f;
            }
        }
        out[(int64_t)b * n + e] = v;              // (t, f) -> (t / stack, (t % stack) * F + f): the same offset
    }
}

extern "C" int64_t asr_featurize_ws_floats(int B, int Tm, int F, int nF) {
    return (int64_t)B * cdiv(Tm > 0 ? Tm : 1, FEAT_CH) * F + (int64_t)B * (nF > 0 ? nF : 1) + 16;
}

extern "C" int asr_featurize(const float *raw, const int32_t *mel_lengths, const float *mean, const float *var,
                             const int32_t *fmask, int nF, int fill_zero, const int32_t *tmask, int nT, int stack,
                             int B, int Tm, int F, float *ws, float *out, void *stream) {
    ASR_CHECK_ARG(raw && mel_lengths && out && ws, "featurize: null pointer");
    ASR_CHECK_ARG((mean == nullptr) == (var == nullptr), "featurize: mean and var come together");
    ASR_CHECK_ARG(B > 0 && Tm > 0 && F > 0 && stack > 0, "featurize: bad shape B=%d T=%d F=%d stack=%d", B, Tm, F, stack);
    ASR_CHECK_ARG(nF >= 0 && nT >= 0 && (nF == 0 || fmask) && (nT == 0 || tmask), "featurize: bad mask lists");
    ASR_CHECK_ARG(F <= 4096, "featurize: at most 4096 channels");
    hipStream_t st = (hipStream_t)stream;
    const int Ts = Tm / stack;
    if (Ts == 0) return ASR_OK;
    const int nchunk = cdiv(Tm, FEAT_CH);
    float *part = ws, *fill = ws + (int64_t)B * nchunk * F;
    if (nF > 0) {
        hipLaunchKernelGGL(feat_colsum_kernel, dim3(nchunk, B), dim3(256), 0, st, raw, mel_lengths, mean, var, Tm, F,
                           nchunk, part);
        ASR_CHECK_LAUNCH("feat_colsum_kernel");
        hipLaunchKernelGGL(feat_plan_kernel, dim3(B), dim3(256), sizeof(double) * (size_t)F, st, part, mel_lengths,
                           fmask, nF, Tm, F, nchunk, fill_zero, fill);
        ASR_CHECK_LAUNCH("feat_plan_kernel");
    }
    const int64_t n = (int64_t)Ts * stack * F;
    const int gx = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(feat_apply_kernel, dim3(gx, B), dim3(256), 0, st, raw, mel_lengths, mean, var, fmask, nF, fill,
                       tmask, nT, Tm, F, stack, Ts, out);
    ASR_CHECK_LAUNCH("feat_apply_kernel");
    return ASR_OK;
}
