// CTC head of the BLSTM encoder: log-softmax + CTC loss + gradient w.r.t. the logits, for
// train.py:56-57 of the reference (F.log_softmax(predict_ts, 2).transpose(0, 1) followed by
// F.ctc_loss(.., blank=0), default reduction 'mean', zero_infinity False).  The arithmetic is
// torch's documented CTC (third-party to the reference): log-space alpha/beta recursions over
// the blank-augmented label path l' (|l'| = 2 L_b + 1),
//     alpha_t(s) = lp_t(l'_s) + logsumexp(alpha_{t-1}(s), alpha_{t-1}(s-1), [l'_s != l'_{s-2}] alpha_{t-1}(s-2))
//     nll_b      = -logsumexp(alpha_{T_b-1}(S_b-1), alpha_{T_b-1}(S_b-2))
//     loss       = mean_b nll_b / max(L_b, 1)
//     dz[b,t,v]  = w_b (softmax(z[b,t])_v - sum_{s: l'_s = v} exp(alpha_t(s) + beta_t(s) - lp_t(v) + nll_b)),  t < T_b
// with w_b = 1 / (B max(L_b, 1)); frames t >= T_b get a zero gradient.
//
// Three passes, all on the caller's stream:
//   ctc_lse_kernel        one workgroup per live frame: lse over V (HBM-bound, logits read once) and
//                         the gathered path scores lpg[b,t,s] = z[b,t,l'_s] - lse, so the recursions
//                         read contiguous rows instead of gathering from the logits;
//   ctc_alpha_beta_kernel one workgroup per (utterance, direction): T_b dependent steps, the path
//                         state in LDS (double-buffered, one barrier per step), rows prefetched
//                         four steps ahead (latency-bound, like the encoder recurrence);
//   ctc_grad_kernel       one workgroup per frame: dense w_b softmax row, then the path occupancies
//                         subtracted.  Repeated labels are summed by ONE thread per distinct label along a
//                         chain built once per utterance, blanks by a fixed-order tree: no float
//                         atomics, bitwise reproducible.
#include "common.h"

#define CTC_T 256

// The recursions run in double: alpha + beta + nll cancels values of magnitude ~T * |log p|
// (thousands for an untrained model), which fp32 resolves to ~1e-3 only; the passes are
// latency-bound, so the fp64 rate of the chip (half its fp32 vector rate) costs nothing.
__device__ __forceinline__ double ctc_lse3(double a, double b, double c) {
    double m = fmax(a, fmax(b, c));
    if (m == -INFINITY) m = 0.0;
    return log(exp(a - m) + exp(b - m) + exp(c - m)) + m;
}

__device__ __forceinline__ int ctc_label(const int64_t *__restrict__ tg, int s, int blank, int V) {
    if (!(s & 1)) return blank;
    const int64_t l = tg[s >> 1];
    return (int)(l < 0 ? 0 : (l >= V ? V - 1 : l));     // keeps the gather in bounds; torch does not check either
}

__global__ __launch_bounds__(CTC_T) void ctc_lse_kernel(const float *__restrict__ z,
                                                        const int64_t *__restrict__ targets,
                                                        const int32_t *__restrict__ in_len,
                                                        const int32_t *__restrict__ tgt_len, int T, int V, int L,
                                                        int blank, float *__restrict__ lse,
                                                        float *__restrict__ lpg, int vec) {
    __shared__ float sh[CTC_T / 64];
    const int row = blockIdx.x, b = row / T, t = row % T;
    const int Tb = min(max(in_len[b], 0), T);
    if (t >= Tb) return;
    const int tid = threadIdx.x;
    const float *zr = z + (int64_t)row * V;
    float mx = -INFINITY;
    if (vec) {
        for (int i = tid; i < V / 4; i += CTC_T) {
            const float4 q = ld4(zr + 4 * i);
            mx = fmaxf(mx, fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
        }
    } else {
        for (int i = tid; i < V; i += CTC_T) mx = fmaxf(mx, zr[i]);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) sh[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    float se = 0.f;
    if (vec) {
        for (int i = tid; i < V / 4; i += CTC_T) {     // second read of the row: L1/L2
            const float4 q = ld4(zr + 4 * i);
            se += (expf(q.x - mx) + expf(q.y - mx)) + (expf(q.z - mx) + expf(q.w - mx));
        }
    } else {
        for (int i = tid; i < V; i += CTC_T) se += expf(zr[i] - mx);
    }
    se = wave_sum(se);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = se;
    __syncthreads();
    se = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    const float l = mx + logf(se);
    if (tid == 0) lse[row] = l;
    const int Lb = min(max(tgt_len[b], 0), L), Sb = 2 * Lb + 1, S = 2 * L + 1;
    const int64_t *tg = targets + (int64_t)b * L;
    for (int s = tid; s < Sb; s += CTC_T) lpg[(int64_t)row * S + s] = zr[ctc_label(tg, s, blank, V)] - l;
}

// grid (B, 2): y = 0 alpha (also the label chains and nll), y = 1 beta.  blockDim >= 2L+1.
__global__ __launch_bounds__(1024) void ctc_alpha_beta_kernel(
    const float *__restrict__ lpg, const int64_t *__restrict__ targets, const int32_t *__restrict__ in_len,
    const int32_t *__restrict__ tgt_len, int T, int V, int L, int blank, double *__restrict__ alpha,
    double *__restrict__ beta, double *__restrict__ nlld, float *__restrict__ nll,
    int32_t *__restrict__ chain_prev, int32_t *__restrict__ chain_last) {
    extern __shared__ double smem[];
    const int S = 2 * L + 1;
    double *buf0 = smem, *buf1 = smem + (S + 4);         // state index s lives at [s + 2]
    int *lab = reinterpret_cast<int *>(smem + 2 * (S + 4));
    const int b = blockIdx.x, dir = blockIdx.y, s = threadIdx.x;
    const int Tb = min(max(in_len[b], 0), T);
    const int Lb = min(max(tgt_len[b], 0), L), Sb = 2 * Lb + 1;
    const int64_t *tg = targets + (int64_t)b * L;
    const bool live = s < Sb;
    const int ls = live ? ctc_label(tg, s, blank, V) : blank;
    if (s < S) lab[s] = ls;
    for (int i = s; i < S + 4; i += blockDim.x) { buf0[i] = -INFINITY; buf1[i] = -INFINITY; }
    __syncthreads();
    if (dir == 0 && live && (s & 1)) {   // chains of equal labels (one thread per distinct label sums them)
        int prev = -1, last = 1;
        for (int q = s - 2; q >= 1; q -= 2)
            if (lab[q] == ls) { prev = q; break; }
        for (int q = s + 2; q < Sb; q += 2)
            if (lab[q] == ls) { last = 0; break; }
        chain_prev[(int64_t)b * S + s] = prev;
        chain_last[(int64_t)b * S + s] = last;
    }
    if (Tb == 0) {
        if (dir == 0 && s == 0) {
            nll[b] = Lb == 0 ? 0.f : INFINITY;
            nlld[b] = Lb == 0 ? 0.0 : (double)INFINITY;
        }
        return;
    }
    const float *lrow = lpg + (int64_t)b * T * S;
    double *out = (dir == 0 ? alpha : beta) + (int64_t)b * T * S;
    // neighbours: alpha looks at s-1, s-2 of the previous frame, beta at s+1, s+2 of the next one
    const int o1 = dir == 0 ? -1 : 1, o2 = 2 * o1;
    const int s2 = s + o2;
    const bool skip = live && s2 >= 0 && s2 < Sb && lab[s2] != ls;
    const int tstart = dir == 0 ? 0 : Tb - 1, tstep = dir == 0 ? 1 : -1;
    auto row_lp = [&](int k) -> float {   // path score of step k (k-th processed frame); -inf past the end
        const int t = tstart + k * tstep;
        return (live && k < Tb) ? lrow[(int64_t)t * S + s] : -INFINITY;
    };
    double a;
    {
        const bool init = dir == 0 ? (s == 0 || s == 1) : (s == Sb - 1 || s == Sb - 2);
        a = (live && init) ? row_lp(0) : -INFINITY;
        if (live) out[(int64_t)tstart * S + s] = a;
        if (s < S) buf0[s + 2] = a;
    }
    float q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = row_lp(1 + u);
    __syncthreads();
    for (int k0 = 1; k0 < Tb; k0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u;
            if (k < Tb) {   // uniform
                const double *rd = (k & 1) ? buf0 : buf1;
                double *wr = (k & 1) ? buf1 : buf0;
                a = -INFINITY;
                if (live) {
                    const double a1 = rd[s + 2], a2 = rd[s + 2 + o1], a3 = skip ? rd[s + 2 + o2] : -INFINITY;
                    a = ctc_lse3(a1, a2, a3) + (double)q[u];
                    out[(int64_t)(tstart + k * tstep) * S + s] = a;
                }
                q[u] = row_lp(k + 4);
                if (s < S) wr[s + 2] = a;
                __syncthreads();
            }
        }
    }
    if (dir == 0 && s == 0) {
        const double *fin = ((Tb - 1) & 1) ? buf1 : buf0;
        const double l1 = fin[Sb - 1 + 2], l2 = Sb >= 2 ? fin[Sb - 2 + 2] : -INFINITY;
        const double v = -ctc_lse3(l1, l2, -INFINITY);
        nlld[b] = v;
        nll[b] = (float)v;
    }
}

__global__ __launch_bounds__(CTC_T) void ctc_grad_kernel(
    const float *__restrict__ z, const int64_t *__restrict__ targets, const int32_t *__restrict__ in_len,
    const int32_t *__restrict__ tgt_len, int B, int T, int V, int L, int blank,
    const float *__restrict__ lse, const float *__restrict__ lpg, const double *__restrict__ alpha,
    const double *__restrict__ beta, const double *__restrict__ nlld, const int32_t *__restrict__ chain_prev,
    const int32_t *__restrict__ chain_last, float *__restrict__ dz, int vec) {
    extern __shared__ double smem[];
    __shared__ float sh[CTC_T / 64];
    const int S = 2 * L + 1;
    float *gam = reinterpret_cast<float *>(smem);
    const int row = blockIdx.x, b = row / T, t = row % T, tid = threadIdx.x;
    const int Tb = min(max(in_len[b], 0), T);
    float *dr = dz + (int64_t)row * V;
    if (t >= Tb) {
        if (vec) for (int i = tid; i < V / 4; i += CTC_T) *reinterpret_cast<float4 *>(dr + 4 * i) = zero4();
        else for (int i = tid; i < V; i += CTC_T) dr[i] = 0.f;
        return;
    }
    const int Lb = min(max(tgt_len[b], 0), L), Sb = 2 * Lb + 1;
    const int64_t *tg = targets + (int64_t)b * L;
    const float w = 1.0f / ((float)B * (float)max(Lb, 1));
    const float l = lse[row];
    const double nl = nlld[b];
    const float *zr = z + (int64_t)row * V;
    const int64_t ro = (int64_t)row * S;
    // occupancies; the blank's (and those of labels equal to the blank) summed in a fixed order
    float bsum = 0.f;
    for (int s = tid; s < Sb; s += CTC_T) {
        const float g = (float)exp(alpha[ro + s] + beta[ro + s] - (double)lpg[ro + s] + nl);
        gam[s] = g;
        if (ctc_label(tg, s, blank, V) == blank) bsum += g;
    }
    // dense part: w * softmax
    if (vec) {
        for (int i = tid; i < V / 4; i += CTC_T) {
            const float4 q = ld4(zr + 4 * i);
            *reinterpret_cast<float4 *>(dr + 4 * i) =
                make_float4(w * expf(q.x - l), w * expf(q.y - l), w * expf(q.z - l), w * expf(q.w - l));
        }
    } else {
        for (int i = tid; i < V; i += CTC_T) dr[i] = w * expf(zr[i] - l);
    }
    bsum = wave_sum(bsum);
    if ((tid & 63) == 0) sh[tid >> 6] = bsum;
    __syncthreads();   // also orders the dense stores before the overwrites below (same workgroup)
    if (tid == 0) {
        bsum = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        dr[blank] = w * (expf(zr[blank] - l) - bsum);
    }
    const int32_t *cp = chain_prev + (int64_t)b * S, *cl = chain_last + (int64_t)b * S;
    for (int s = 2 * tid + 1; s < Sb; s += 2 * CTC_T) {
        const int v = ctc_label(tg, s, blank, V);
        if (v == blank || !cl[s]) continue;
        float sum = 0.f;
        for (int q = s; q >= 0; q = cp[q]) sum += gam[q];
        dr[v] = w * (expf(zr[v] - l) - sum);
    }
}

__global__ __launch_bounds__(CTC_T) void ctc_mean_kernel(const float *__restrict__ nll,
                                                         const int32_t *__restrict__ tgt_len, int B, int L,
                                                         float *__restrict__ loss) {
    __shared__ double sh[CTC_T];
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += CTC_T) acc += (double)nll[b] / (double)max(min(max(tgt_len[b], 0), L), 1);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = CTC_T / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sh[0] / (double)B);
}

extern "C" int64_t asr_ctc_ws_floats(int B, int T, int L) {
    const int64_t S = 2 * (int64_t)L + 1, BT = (int64_t)B * T;
    // alpha, beta, nll (double) | lse | lpg | chain_prev | chain_last
    return 2 * (2 * BT * S + B) + BT + BT * S + 2 * (int64_t)B * S + 16;
}

extern "C" int asr_ctc_loss_fwd_bwd(const float *logits, const int64_t *targets, const int32_t *input_lengths,
                                    const int32_t *target_lengths, int B, int T, int V, int L, int blank,
                                    float *ws, float *nll, float *loss, float *dlogits, void *stream) {
    ASR_CHECK_ARG(logits && (targets || L == 0) && input_lengths && target_lengths && ws && nll && loss && dlogits,
                  "ctc: null pointer");
    ASR_CHECK_ARG(B > 0 && T > 0 && V > 0 && L >= 0, "ctc: bad shape B=%d T=%d V=%d L=%d", B, T, V, L);
    ASR_CHECK_ARG(blank >= 0 && blank < V, "ctc: blank %d outside the vocabulary", blank);
    const int S = 2 * L + 1;
    ASR_CHECK_ARG(S <= 1024, "ctc: at most 511 labels per utterance (got L=%d)", L);
    ASR_CHECK_ARG((int64_t)B * T <= 0x7fffffff / 4, "ctc: too many frames");
    hipStream_t st = (hipStream_t)stream;
    const int64_t BT = (int64_t)B * T;
    ASR_CHECK_ARG(((uintptr_t)ws & 7) == 0, "ctc: workspace must be 8-byte aligned");
    double *alpha = reinterpret_cast<double *>(ws), *beta = alpha + BT * S, *nlld = beta + BT * S;
    float *lse = reinterpret_cast<float *>(nlld + B), *lpg = lse + BT;
    int32_t *cprev = reinterpret_cast<int32_t *>(lpg + BT * S), *clast = cprev + (int64_t)B * S;
    const int vec = (V % 4 == 0) && (((uintptr_t)logits & 15) == 0) && (((uintptr_t)dlogits & 15) == 0);
    hipLaunchKernelGGL(ctc_lse_kernel, dim3((unsigned)BT), dim3(CTC_T), 0, st, logits, targets, input_lengths,
                       target_lengths, T, V, L, blank, lse, lpg, vec);
    ASR_CHECK_LAUNCH("ctc_lse_kernel");
    const int nthr = ((S + 63) / 64) * 64;
    const size_t smem_ab = sizeof(double) * (size_t)(2 * (S + 4)) + sizeof(int) * (size_t)S;
    hipLaunchKernelGGL(ctc_alpha_beta_kernel, dim3(B, 2), dim3(nthr), smem_ab, st, lpg, targets, input_lengths,
                       target_lengths, T, V, L, blank, alpha, beta, nlld, nll, cprev, clast);
    ASR_CHECK_LAUNCH("ctc_alpha_beta_kernel");
    hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)BT), dim3(CTC_T), sizeof(float) * (size_t)S, st, logits,
                       targets, input_lengths, target_lengths, B, T, V, L, blank, lse, lpg, alpha, beta, nlld, cprev,
                       clast, dlogits, vec);
    ASR_CHECK_LAUNCH("ctc_grad_kernel");
    hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(CTC_T), 0, st, nll, target_lengths, B, L, loss);
    ASR_CHECK_LAUNCH("ctc_mean_kernel");
    return ASR_OK;
}
