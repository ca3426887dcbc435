// Bidirectional LSTM recurrence over a ragged batch, without packing.
//
// Replaces the recurrence inside torch.nn.LSTM as Models/encoder.py:32-34 calls
// it (pack_padded_sequence(enforce_sorted=False) -> LSTM -> pad_packed_sequence):
// gate order i,f,g,o; sequence b runs over lengths[b] frames; the reverse
// direction starts at frame lengths[b]-1 from a zero state; rows at
// t >= lengths[b] are exactly 0.
//
// The input projections x_t W_ih^T + b_ih + b_hh for all t and both directions
// are one MFMA GEMM (asr_sgemm) into G; this file only runs the sequential
// part.  One launch per time step advances BOTH directions (forward at t =
// step, reverse at t = T-1-step); a workgroup owns 4 hidden units (all four
// gates = one 16-column MFMA tile) of 16 batch rows, reads h_{t-1} from the
// neighbouring time row of the layer output itself (padded rows are zeros, so
// the reverse direction needs no per-sequence start offset) and writes h_t, c_t
// and the activated gates.  HBM/L2 traffic per step is W_hh once (2*4H*H*4 B)
// plus h_{t-1}; the step is latency-bound, not bandwidth-bound.
#include "tile16.cuh"

__global__ __launch_bounds__(256) void blstm_fwd_step_kernel(
    const float *__restrict__ W_hh, float *__restrict__ G, float *__restrict__ Cs,
    float *__restrict__ Hout, int64_t ho_t, int64_t ho_b, const int32_t *__restrict__ lengths,
    int T, int B, int H, int step, int vec) {
    __shared__ float red[4][16][17];
    const int d = blockIdx.y;
    const int t = d == 0 ? step : T - 1 - step;
    const int tp = d == 0 ? t - 1 : t + 1;
    const bool has_prev = tp >= 0 && tp < T;
    const int j0 = blockIdx.x * 4, b0 = blockIdx.z * 16;
    const int lane = threadIdx.x & 63;

    if (has_prev) {  // uniform
        const int i = lane & 15;
        const int b = b0 + i;
        const float *xrow = b < B ? Hout + tp * ho_t + b * ho_b + d * H : nullptr;
        const int gate = i >> 2, j = j0 + (i & 3);
        const float *wrow = j < H ? W_hh + ((int64_t)d * 4 * H + gate * H + j) * H : nullptr;
        Tile16Acc acc;
        tile16_init(acc);
        tile16_seg(acc, xrow, wrow, H, vec != 0);
        tile16_reduce(acc, red);
    }
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int bi = tid >> 2, ju = tid & 3;
        const int b = b0 + bi, j = j0 + ju;
        if (b < B && j < H) {
            const int64_t grow = (((int64_t)t * B + b) * 2 + d) * 4 * H;
            const int64_t crow = (((int64_t)t * B + b) * 2 + d) * H;
            float *hdst = Hout + t * ho_t + b * ho_b + d * H + j;
            if (t < lengths[b]) {
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v = G[grow + g * H + j];
                    if (has_prev) v += tile16_get(red, bi, g * 4 + ju);
                    pre[g] = v;
                }
                const float ig = sigmoid_f(pre[0]), fg = sigmoid_f(pre[1]);
                const float gg = tanh_f(pre[2]), og = sigmoid_f(pre[3]);
                const float cp = has_prev ? Cs[(((int64_t)tp * B + b) * 2 + d) * H + j] : 0.f;
                const float c = fg * cp + ig * gg;
                G[grow + 0 * H + j] = ig;
                G[grow + 1 * H + j] = fg;
                G[grow + 2 * H + j] = gg;
                G[grow + 3 * H + j] = og;
                Cs[crow + j] = c;
                *hdst = og * tanh_f(c);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) G[grow + g * H + j] = 0.f;
                Cs[crow + j] = 0.f;
                *hdst = 0.f;
            }
        }
    }
}

extern "C" int asr_blstm_fwd(const float *W_hh, float *G, float *Cs, float *Hout, int64_t ho_t,
                             int64_t ho_b, const int32_t *lengths, int T, int B, int H, void *stream) {
    ASR_CHECK_ARG(W_hh && G && Cs && Hout && lengths, "blstm_fwd: null pointer");
    ASR_CHECK_ARG(T > 0 && B > 0 && H > 0, "blstm_fwd: bad shape T=%d B=%d H=%d", T, B, H);
    hipStream_t st = (hipStream_t)stream;
    const int vec = (H % 4 == 0) && (ho_t % 4 == 0) && (ho_b % 4 == 0) && (((uintptr_t)Hout & 15) == 0) &&
                    (((uintptr_t)W_hh & 15) == 0);
    dim3 grid(cdiv(H, 4), 2, cdiv(B, 16));
    for (int step = 0; step < T; ++step) {
        hipLaunchKernelGGL(blstm_fwd_step_kernel, grid, dim3(256), 0, st, W_hh, G, Cs, Hout, ho_t, ho_b,
                           lengths, T, B, H, step, vec);
    }
    ASR_CHECK_LAUNCH("blstm_fwd_step_kernel");
    return ASR_OK;
}

// ------------------------------------------------------------------ backward --
// Step order is the reverse of the forward's: direction 0 walks t = T-1..0,
// direction 1 walks t = 0..T-1.  dh_t = dHout_t + dG_{t'} W_hh (t' = the step
// processed just before, whose pre-activation gradients already sit in G), with
// W_hh^T passed pre-transposed so both MFMA operands are k-contiguous.
__global__ __launch_bounds__(1024) void blstm_bwd_step_kernel(
    const float *__restrict__ W_hhT, float *__restrict__ G, const float *__restrict__ Cs,
    const float *__restrict__ dHout, int64_t dho_t, int64_t dho_b,
    const int32_t *__restrict__ lengths, float *__restrict__ dc_ws, int T, int B, int H, int step,
    int vec) {
    __shared__ float red[T16_MAXW][16][17];
    const int d = blockIdx.y;
    const int t = d == 0 ? T - 1 - step : step;
    const int tn = d == 0 ? t + 1 : t - 1;  // processed in the previous launch
    const int tp = d == 0 ? t - 1 : t + 1;  // the forward pass's previous time (c_{prev})
    const bool has_next = tn >= 0 && tn < T;
    const bool has_prev = tp >= 0 && tp < T;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.z * 16;
    const int lane = threadIdx.x & 63;

    if (has_next) {  // uniform
        const int i = lane & 15;
        const int b = b0 + i, j = j0 + i;
        const float *xrow = b < B ? G + (((int64_t)tn * B + b) * 2 + d) * 4 * H : nullptr;
        const float *wrow = j < H ? W_hhT + ((int64_t)d * H + j) * 4 * H : nullptr;
        Tile16Acc acc;
        tile16_init(acc);
        tile16_seg(acc, xrow, wrow, 4 * H, vec != 0);
        tile16_reduce(acc, red);
    }
    const int tid = threadIdx.x;
    if (tid < 256) {
        const int bi = tid >> 4, ji = tid & 15;
        const int b = b0 + bi, j = j0 + ji;
        if (b < B && j < H) {
            const int64_t grow = (((int64_t)t * B + b) * 2 + d) * 4 * H;
            if (t < lengths[b]) {
                float dh = dHout[t * dho_t + b * dho_b + d * H + j];
                if (has_next) dh += tile16_get(red, bi, ji);
                const float ig = G[grow + 0 * H + j], fg = G[grow + 1 * H + j];
                const float gg = G[grow + 2 * H + j], og = G[grow + 3 * H + j];
                const float c = Cs[(((int64_t)t * B + b) * 2 + d) * H + j];
                const float cp = has_prev ? Cs[(((int64_t)tp * B + b) * 2 + d) * H + j] : 0.f;
                const float tc = tanh_f(c);
                float *dcp = dc_ws + ((int64_t)d * B + b) * H + j;
                const float dc = *dcp + dh * og * (1.f - tc * tc);
                G[grow + 0 * H + j] = dc * gg * ig * (1.f - ig);
                G[grow + 1 * H + j] = dc * cp * fg * (1.f - fg);
                G[grow + 2 * H + j] = dc * ig * (1.f - gg * gg);
                G[grow + 3 * H + j] = dh * tc * og * (1.f - og);
                *dcp = dc * fg;
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) G[grow + g * H + j] = 0.f;
            }
        }
    }
}

extern "C" int asr_blstm_bwd(const float *W_hhT, float *G, const float *Cs, const float *dHout,
                             int64_t dho_t, int64_t dho_b, const int32_t *lengths, float *dc_ws, int T,
                             int B, int H, void *stream) {
    ASR_CHECK_ARG(W_hhT && G && Cs && dHout && lengths && dc_ws, "blstm_bwd: null pointer");
    ASR_CHECK_ARG(T > 0 && B > 0 && H > 0, "blstm_bwd: bad shape T=%d B=%d H=%d", T, B, H);
    hipStream_t st = (hipStream_t)stream;
    ASR_CHECK_HIP(hipMemsetAsync(dc_ws, 0, sizeof(float) * 2 * (size_t)B * H, st));
    const int vec = (H % 4 == 0) && (((uintptr_t)G & 15) == 0) && (((uintptr_t)W_hhT & 15) == 0);
    dim3 grid(cdiv(H, 16), 2, cdiv(B, 16));
    for (int step = 0; step < T; ++step) {
        hipLaunchKernelGGL(blstm_bwd_step_kernel, grid, dim3(1024), 0, st, W_hhT, G, Cs, dHout, dho_t,
                           dho_b, lengths, dc_ws, T, B, H, step, vec);
    }
    ASR_CHECK_LAUNCH("blstm_bwd_step_kernel");
    return ASR_OK;
}
