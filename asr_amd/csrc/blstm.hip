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
#include <string.h>

#include "tile16.h"

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
        tile16_seg(acc, xrow, wrow, H, vec != 0, W_hh);
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

static int blstm_fwd_steps(const float *W_hh, float *G, float *Cs, float *Hout, int64_t ho_t,
                          int64_t ho_b, const int32_t *lengths, int T, int B, int H, int s0, int s1,
                          void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int vec = (H % 4 == 0) && (ho_t % 4 == 0) && (ho_b % 4 == 0) && (((uintptr_t)Hout & 15) == 0) &&
                    (((uintptr_t)W_hh & 15) == 0);
    dim3 grid(cdiv(H, 4), 2, cdiv(B, 16));
    for (int step = s0; step < s1; ++step) {
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
        tile16_seg(acc, xrow, wrow, 4 * H, vec != 0, W_hhT);
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

static int blstm_bwd_steps(const float *W_hhT, float *G, const float *Cs, const float *dHout,
                          int64_t dho_t, int64_t dho_b, const int32_t *lengths, float *dc_ws, int T,
                          int B, int H, int s0, int s1, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (s0 == 0) ASR_CHECK_HIP(hipMemsetAsync(dc_ws, 0, sizeof(float) * 2 * (size_t)B * H, st));
    const int vec = (H % 4 == 0) && (((uintptr_t)G & 15) == 0) && (((uintptr_t)W_hhT & 15) == 0);
    dim3 grid(cdiv(H, 16), 2, cdiv(B, 16));
    for (int step = s0; step < s1; ++step) {
        hipLaunchKernelGGL(blstm_bwd_step_kernel, grid, dim3(1024), 0, st, W_hhT, G, Cs, dHout, dho_t,
                           dho_b, lengths, dc_ws, T, B, H, step, vec);
    }
    ASR_CHECK_LAUNCH("blstm_bwd_step_kernel");
    return ASR_OK;
}

// ===================================================== persistent kernels ==
// One launch per layer instead of one per time step.  W_hh slices live in
// registers for the whole sequence, the cell state c in the owner thread's
// registers; the only data exchanged between workgroups per step is h_t
// (forward) / dG_t (backward), handed over inside the launch:
//   producer: write-through (sc1) stores -> every storing wave s_waitcnt
//             vmcnt(0) -> workgroup barrier -> ONE lane stores the
//             workgroup's flag = step+1 (sc1)
//   consumer: wave 0 polls the group's flags with sc1 loads (relaxed, bounded
//             spin) -> workgroup barrier -> every wave reads the payload with
//             sc1 loads straight into the MFMA operand registers.
// (cdna_hip_programming.md guideline 16 / MI355X_MICROARCH.md "valid forms",
// flag row.)  The protocol does not depend on dispatch order or placement; it
// needs all workgroups resident, which the grid guarantees by size (<= 256
// workgroups of 256 threads, a few KB of LDS).  A poll that exceeds its spin
// budget sets *err and makes the workgroup leave; the others then time out the
// same way, so the grid always drains.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// in-launch waits are bounded by wall time: asr_wait_expired (common.h), 4 s on the 100 MHz counter
#define BLSTM_FLAG_WORDS_ 1024   // = BLSTM_FLAG_WORDS (defined with the host dispatch)
#define BLSTM_PACE_START 8     // initial wait before a step's first exchange gather, units of s_sleep 2
#define BLSTM_PACE_MAX 40
#define OOB_OFF 0x7ffffff0   // beyond every buffer here: the range check returns zeros

__device__ __forceinline__ float4 ld4_sc1(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16 /* sc1 */);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st_sc1(float *p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// epoch tag carried in the mantissa LSB of every exchanged float (tag mode)
__device__ __forceinline__ unsigned tag_bits(float v, unsigned e) { return (__float_as_uint(v) & ~1u) | e; }
__device__ __forceinline__ bool tags_ok(const float4 &v, unsigned e) {
    return (((__float_as_uint(v.x) ^ e) | (__float_as_uint(v.y) ^ e) | (__float_as_uint(v.z) ^ e) |
             (__float_as_uint(v.w) ^ e)) & 1u) == 0u;
}
// bf16x3 split of the recurrent products (asr_blstm_precision): a value travels / sits in a register as ONE
// 32-bit word {bf16 hi : bf16 lo}, hi = bf16(x), lo = bf16(x - hi) (round to nearest even); two such words give one
// word of the hi operand and one of the lo operand of v_mfma_f32_16x16x32_bf16 with a byte permute each.
typedef __bf16 bl_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bl_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bl_hilo(float x) {
    bl_bf16x2 h = {(__bf16)x, (__bf16)0.f};
    const unsigned hb = __builtin_bit_cast(unsigned, h) & 0xffffu;
    bl_bf16x2 l = {(__bf16)(x - __uint_as_float(hb << 16)), (__bf16)0.f};
    return (hb << 16) | (__builtin_bit_cast(unsigned, l) & 0xffffu);
}
// eight {hi:lo} words (k = 0..7) -> the hi fragment and the lo fragment (8 bf16 each, k order kept)
__device__ __forceinline__ void bl_frags(const unsigned (&w)[8], u32x4 &hi, u32x4 &lo) {
    hi = u32x4{__builtin_amdgcn_perm(w[1], w[0], 0x07060302u), __builtin_amdgcn_perm(w[3], w[2], 0x07060302u),
               __builtin_amdgcn_perm(w[5], w[4], 0x07060302u), __builtin_amdgcn_perm(w[7], w[6], 0x07060302u)};
    lo = u32x4{__builtin_amdgcn_perm(w[1], w[0], 0x05040100u), __builtin_amdgcn_perm(w[3], w[2], 0x05040100u),
               __builtin_amdgcn_perm(w[5], w[4], 0x05040100u), __builtin_amdgcn_perm(w[7], w[6], 0x05040100u)};
}
__device__ __forceinline__ f32x4 bl_mfma3(const u32x4 &ah, const u32x4 &al, const u32x4 &bh, const u32x4 &bl, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bl_bf16x8, al), __builtin_bit_cast(bl_bf16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bl_bf16x8, ah), __builtin_bit_cast(bl_bf16x8, bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bl_bf16x8, ah), __builtin_bit_cast(bl_bf16x8, bh), acc, 0, 0, 0);
    return acc;
}

// ------------------------------------------------- saved-activation prefetch --
// In a training step the backward recurrence of a layer finds its saved activations COLD: the layer's gates (164 MB at
// the config-2 size), cell states and upstream gradient were written a whole forward + decoder pass ago, and the
// side stream's weight-gradient GEMMs stream hundreds of MB through the 256 MiB Infinity Cache meanwhile.  Alone on a
// warm working set a launch takes 815-840 us, after a 1 GiB fill 970, in the step 1030; READING the tensors once
// after the fill brings 970 back to 831 (tools/lstm_cold.py, LSTM_TOUCH=1) -- a line fetched by anybody sits in the
// memory-side cache for everybody.  The XCD-grouped grid has workgroups to spare (labels 2 NBT .. 7 used to leave
// at once); they now walk the time steps AHEAD of the recurrence and touch one dword of every 128-byte line the step
// will load -- G, Cs, dHout rows of the frames inside their sequences -- staying `dist` steps ahead of the slowest
// group's published progress: far enough to cover an HBM round trip several times over, near enough that the
// GEMMs' traffic does not push the lines out again before use.  Nothing depends on it: the loads are range-checked
// buffer loads whose results are dropped, a prefetcher that loses sight of the progress words (4 s) just leaves.
// Progress words: ws word BLSTM_PROG_WORD + group = (launch stamp & 0xffff) << 16 | step, written by thread 0 of
// the group's workgroup 0 every step (a branch-free buffer store; every other lane's offset is out of range);
// 0xffff = the group has left its loop.  A word with another launch's stamp counts as "not started".
#define BLSTM_PROG_WORD 1032      // 8 words: backward groups (words 1025..2047 of the workspace are otherwise unused)
#define BLSTM_PROG_WORD_FWD 1040  // 8 words: forward groups
__device__ __noinline__ void blstm_prefetch_steps(const float *G, int64_t g_bytes, int g_row, const float *Cs, int64_t c_bytes,
                                                  const float *X, int64_t x_bytes, int64_t x_t, int64_t x_b,
                                                  const int32_t *lengths, int T, int B, int H, int s0, int s1,
                                                  bool dir0_descends, const unsigned *prog, int ngroups, unsigned stamp16,
                                                  int dist, int pw, int npw) {
    // G rows are g_row floats per (frame, row, direction); Cs / X (= dHout) may be nullptr: the forward reads only G.
    // Unit of work of a WAVE: one (step, direction, quarter of the batch rows) -- all its touches are issued before
    // the first is waited for, so a unit costs one memory round trip and `dist` can stay small.
    const int lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G), 0, (int)g_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Cs ? Cs : G), 0, (int)(Cs ? c_bytes : 0), 0x00020000);
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X ? X : G), 0, (int)(X ? x_bytes : 0), 0x00020000);
    constexpr int NPART = 8;                               // 2 directions x 4 row chunks
    const int RB = (B + 3) / 4;
    const int nlg = (g_row * 4 + 127) / 128 + 1, nlc = (H * 4 + 127) / 128 + 1;   // touches per row segment: every 128 B + its last dword
    const int nG = RB * nlg, nC = Cs ? RB * nlc : 0, nX = X ? RB * nlc : 0, ntot = nG + nC + nX;
    const int nunits = (s1 - s0) * NPART;
    for (int u = pw; u < nunits; u += npw) {
        const int s = s0 + u / NPART, part = u % NPART, d = part & 1, rb0 = (part >> 1) * RB;
        const int t = (d == 0) == dir0_descends ? T - 1 - s : s;
        unsigned long long t0 = 0;
        for (int spins = 0;; ++spins) {      // stay within `dist` steps of the slowest group
            int p = 0x7fffffff;
            for (int g = 0; g < ngroups; ++g) {
                const unsigned w = __hip_atomic_load(prog + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int ps = (w >> 16) != stamp16 ? s0 : ((w & 0xffffu) == 0xffffu ? 0x7fffffff : (int)(w & 0xffffu));
                p = min(p, ps);
            }
            if (p == 0x7fffffff) return;     // every group has left its loop
            if (s <= p + dist) break;
            if (asr_wait_expired(spins, t0)) return;
            __builtin_amdgcn_s_sleep(32);
        }
        for (int base = 0; base < ntot; base += 64 * 16) {
            unsigned v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                int idx = base + 64 * k + lane;
                const bool in = idx < ntot;
                // which tensor, row of the chunk, touch within the row segment
                const int which = idx < nG ? 0 : (idx < nG + nC ? 1 : 2);
                idx -= which == 0 ? 0 : (which == 1 ? nG : nG + nC);
                const int nl = which == 0 ? nlg : nlc;
                const int r = idx / nl, l = idx - r * nl, b = rb0 + r;
                const bool ok = in && b < B && t < lengths[b < B ? b : 0];
                const int seg = (which == 0 ? g_row : H) * 4;
                const int64_t row0 = which == 0 ? ((((int64_t)t * B + b) * 2 + d) * g_row) * 4
                                   : (which == 1 ? ((((int64_t)t * B + b) * 2 + d) * H) * 4
                                                 : (t * x_t + b * x_b + (int64_t)d * H) * 4);
                const int off = ok ? (int)(row0 + min(l * 128, seg - 4)) : OOB_OFF;
                v[k] = which == 0 ? __builtin_amdgcn_raw_buffer_load_b32(rG, off, 0, 0)
                     : (which == 1 ? __builtin_amdgcn_raw_buffer_load_b32(rC, off, 0, 0)
                                   : __builtin_amdgcn_raw_buffer_load_b32(rX, off, 0, 0));
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
        }
    }
}

// wave-wide: wait until flags[0..n) >= want (n <= 128).  Returns false on timeout.
__device__ __forceinline__ bool poll_flags(unsigned *flags, int n, unsigned want) {
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = 0;
    for (int spins = 0; !asr_wait_expired(spins, t0); ++spins) {
        unsigned v0 = lane < n ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : 0xffffffffu;
        unsigned v1 = lane + 64 < n ? __hip_atomic_load(flags + lane + 64, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT)
                                    : 0xffffffffu;
        if (__all(v0 >= want && v1 >= want)) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// grid (NJ = H/(4 NT), 2, NBT), block 256.  Workgroup = 4*NT hidden units (NT 16-column gate
// tiles) of one direction for MT*16 batch rows; hx[dir][bt][parity][NJ][MT*16][4 NT]: the h
// exchange.  (MT, NT) = (1, 2) for 17..32 rows: two batch tiles on separate workgroups, each
// loading only ITS 16 rows of h_{t-1} per step (20 KB instead of 40 KB at H = 320 -- the price of
// an in-launch hand-off sits in the consumer CU's memory queue); (2, 1) when H % 8 != 0; (1, 1)
// for up to 16 rows.
// BF: the products on the bf16 matrix pipe (bf16x3 split, MT == 1, H % 32 == 0): the exchanged h travels as
// {hi:lo} words, W_hh sits in registers pre-split, k-step ks = wave + 4 f (32 wide) of the KQ4 = ceil(H/128) steps
// of a wave; everything else -- tags, issue order, cell update -- is the fp32 kernel's.
// PROD: the product path's settings (no timing-experiment bits, in-band tags) are compile-time constants: every
// run-time switch inside the step loop is a scalar branch between the loads and their waits.
template <int MT, int NT, int KQ4, bool BF = false, bool PROD = false>
__global__ __launch_bounds__(256) void blstm_fwd_persist_kernel(
    const float *__restrict__ W_hh, float *__restrict__ G, float *__restrict__ Cs,
    float *__restrict__ Hout, int64_t ho_t, int64_t ho_b, int64_t hout_bytes,
    const int32_t *__restrict__ lengths, int T, int B, int H, int s0, int s1, float *hx, unsigned *flags,
    unsigned *err, int dbg_arg, int tagmode_arg, int xcd_nbt_arg = 0, unsigned xcd_epoch = 0) {
    const int dbg = PROD ? 0 : dbg_arg, tagmode = PROD ? 1 : tagmode_arg;
    const int pf_dist = PROD ? (xcd_nbt_arg >> 8) : 0;      // prefetch distance in steps (0: off), see blstm_prefetch_steps
    const int xcd_nbt = xcd_nbt_arg & 0xff;
    static_assert(MT * NT == 2 || (MT == 1 && NT == 1), "128 (64) owner threads");
    constexpr int UN = 4 * NT;                               // hidden units per workgroup
    __shared__ float red[2][4][MT][NT][16][17];
    __shared__ int s_abort, s_local;
    // Classic grid: (NJ, 2, NBT).  XCD-grouped grid (xcd_nbt > 0): 8 * NJ workgroups in one dimension; workgroups are
    // dealt round-robin over the 8 XCDs, so id & 7 labels the XCD and each (direction, batch tile) group takes the
    // workgroups of ONE label (labels 2 * xcd_nbt .. 7 leave at once).  Whether the group really shares an XCD is
    // checked below from HW_REG_XCC_ID before anything relies on it.
    int jt = blockIdx.x, d = blockIdx.y, bt = blockIdx.z, NJ = gridDim.x, NBT = gridDim.z;
    if (xcd_nbt > 0) {
        const int grp = blockIdx.x & 7;
        if (grp >= 2 * xcd_nbt) {       // spare workgroups: touch the input-projection gates ahead of the recurrence
            if (pf_dist > 0)
                blstm_prefetch_steps(G, (int64_t)T * B * 8 * H * 4, 4 * H, nullptr, 0, nullptr, 0, 0, 0, lengths, T, B, H,
                                     s0 == 0 ? 1 : s0, s1, false, flags + BLSTM_PROG_WORD_FWD, 2 * xcd_nbt,
                                     xcd_epoch & 0xffffu, pf_dist,
                                     ((grp - 2 * xcd_nbt) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3)) * 4 + (int)(threadIdx.x >> 6),
                                     (8 - 2 * xcd_nbt) * (int)(gridDim.x >> 3) * 4);
            return;
        }
        jt = blockIdx.x >> 3; d = grp & 1; bt = grp >> 1; NJ = gridDim.x >> 3; NBT = xcd_nbt;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    constexpr int KQ = KQ4 * 4;
    const int kb = (wave * 4 + q) * KQ;
    const int b0 = bt * MT * 16;

    static_assert(!BF || MT == 1, "the bf16 form is built for one batch tile per workgroup");
    float4 wreg[NT][BF ? 1 : KQ4];
    u32x4 wbh[BF ? NT : 1][KQ4], wbl[BF ? NT : 1][KQ4];    // BF: W_hh fragments, hi and lo
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int gate = i >> 2, j = jt * UN + 4 * nt + (i & 3);
        const float *wrow = W_hh + ((int64_t)d * 4 * H + gate * H + j) * H;
        if (BF) {
#pragma unroll
            for (int f = 0; f < KQ4; ++f) {
                const int k0 = 32 * (wave + 4 * f) + 8 * q;
                unsigned ww[8];
                const bool ok = j < H && k0 < H;      // H % 32 == 0: the eight floats are in range together
                const float4 w0 = ok ? ld4(wrow + k0) : zero4(), w1 = ok ? ld4(wrow + k0 + 4) : zero4();
                ww[0] = bl_hilo(w0.x); ww[1] = bl_hilo(w0.y); ww[2] = bl_hilo(w0.z); ww[3] = bl_hilo(w0.w);
                ww[4] = bl_hilo(w1.x); ww[5] = bl_hilo(w1.y); ww[6] = bl_hilo(w1.z); ww[7] = bl_hilo(w1.w);
                bl_frags(ww, wbh[nt][f], wbl[nt][f]);
            }
        } else {
#pragma unroll
            for (int f = 0; f < KQ4; ++f) wreg[nt][f] = (j < H && kb + 4 * f < H) ? ld4(wrow + kb + 4 * f) : zero4();
        }
    }
    const bool owner = tid < MT * NT * 64;
    const int bl = tid / UN, u = tid % UN;                   // local batch row, local unit
    const int bi = b0 + bl;
    const int jo = jt * UN + u;
    const bool valid = owner && bi < B && jo < H;
    const int len = valid ? lengths[bi] : 0;
    float c = 0.f;
    if (s0 > 0 && valid) {   // continuation launch: the cell state of step s0-1 is in Cs, h_{s0-1} still in hx
        const int tprev = d == 0 ? s0 - 1 : T - s0;
        c = Cs[(((int64_t)tprev * B + bi) * 2 + d) * H + jo];
    }
    const int chunk = MT * 16 * UN;                          // floats per workgroup per step: [rows][UN]
    float *hxd = hx + (int64_t)(d * NBT + bt) * 2 * NJ * chunk;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(hxd, 0, 2 * NJ * chunk * 4, 0x00020000);
    unsigned *fl = flags + (d * NBT + bt) * NJ;
    if (tid == 0) { s_abort = 0; s_local = 0; }
    __syncthreads();
    // XCD-grouped grid: every workgroup publishes the XCD it runs on (sc1 store, stamped with this launch's epoch);
    // if all NJ members of the group report the same XCD, the h exchange uses PLAIN stores -- the line stays in that
    // XCD's L2, where the consumers' sc1 loads (L1 bypass, L2-served) find it without the trip through the fabric.
    // Any other outcome (mixed placement, a member that does not show up within the bounded poll) keeps the
    // write-through stores: the choice only changes latency, every member of a group reads the same words and so
    // makes the same choice.
    if (xcd_nbt > 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xffu;
        unsigned *xid = flags + 512 + (d * NBT + bt) * 64;
        if (tid == 0) __hip_atomic_store(xid + jt, (xcd_epoch << 8) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == 0) {
            unsigned v = 0;
            bool ok = false;
            for (int spins = 0; spins < (1 << 16); ++spins) {
                v = lane < NJ ? __hip_atomic_load(xid + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((xcd_epoch << 8) | xcc);
                ok = (v >> 8) == xcd_epoch;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            const bool same = __all(ok && (v & 0xffu) == xcc);
            if (lane == 0) s_local = same && !(dbg & 16);
        }
        __syncthreads();
    }
    const bool xlocal = s_local != 0;
    if (!(dbg & 32)) __builtin_amdgcn_s_setprio(3);   // issue priority over co-resident waves of other kernels (side-stream GEMMs)

    // Order of the vector-memory instructions inside a step (s_waitcnt vmcnt counts loads AND
    // stores, in issue order, so whatever is issued before the exchange loads is waited for with
    // them):  exchange loads of h_{s-1}  ->  this step's input-projection gates (HBM latency,
    // needed only after the MFMAs)  ->  the PREVIOUS step's saved activations (stores nobody in
    // the launch waits for)  ->  wait for the exchange loads only.
    // All of it branch-free (buffer accesses; invalid lanes get an out-of-range offset: loads
    // return 0, stores are dropped), so the compiler's wait counts are exact on the one path.
    // The set-up loads are consumed here: a load still pending at the loop entry would cost a
    // conservative queue drain in EVERY iteration.
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int f = 0; f < KQ4; ++f) {
            if (BF) {
                asm volatile("" ::"v"(wbh[nt][f].x), "v"(wbh[nt][f].y), "v"(wbh[nt][f].z), "v"(wbh[nt][f].w));
                asm volatile("" ::"v"(wbl[nt][f].x), "v"(wbl[nt][f].y), "v"(wbl[nt][f].z), "v"(wbl[nt][f].w));
            } else {
                asm volatile("" ::"v"(wreg[nt][f].x), "v"(wreg[nt][f].y), "v"(wreg[nt][f].z), "v"(wreg[nt][f].w));
            }
        }
    asm volatile("" ::"v"(c), "v"(len));
    float sv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // ig, fg, gg, og, c, h of the previous step
    int sv_g = OOB_OFF, sv_c = OOB_OFF, sv_h = OOB_OFF;   // their byte offsets
    __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(G, 0, (int)((int64_t)T * B * 8 * H * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(Cs, 0, (int)((int64_t)T * B * 2 * H * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(Hout, 0, (int)hout_bytes, 0x00020000);
    const int gstep = valid ? H * 4 : 0;
    // progress word of this (direction, batch tile) group, for the prefetching workgroups (blstm_prefetch_steps)
    __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(flags + BLSTM_PROG_WORD_FWD, 0, 8 * 4, 0x00020000);
    const int prog_off = (xcd_nbt > 0 && pf_dist > 0 && jt == 0 && tid == 0) ? (d * NBT + bt) * 4 : OOB_OFF;
    const unsigned prog_stamp = (xcd_epoch & 0xffffu) << 16;
    // Shadow lanes: the threads right behind the owners (no cell of their own) load the NEXT step's input-projection
    // gates of the owner MT*NT*64 places before them and drop the values.  In the training step G is cold (164 MB per
    // layer); the touch brings the lines into this CU's L1 / the XCD's L2 a step before the owner needs them.
    const int lead = ((dbg >> 24) & 15) ? ((dbg >> 24) & 15) : 1;      // steps ahead (timing experiments: bits 24..27)
    const int sh = tid - MT * NT * 64;
    const bool shadow = sh >= 0 && sh < MT * NT * 64 && !(dbg & 64);
    const int sh_bi = b0 + sh / UN, sh_jo = jt * UN + sh % UN;
    const bool sh_valid = shadow && sh_bi < B && sh_jo < H;
    auto shadow_offset = [&](int t) -> int {
        return (sh_valid && t >= 0 && t < T) ? (int)(((((int64_t)t * B + sh_bi) * 2 + d) * 4 * H + sh_jo) * 4) : OOB_OFF;
    };

    auto gate_offset = [&](int t) -> int {
        return valid ? (int)(((((int64_t)t * B + bi) * 2 + d) * 4 * H + jo) * 4) : OOB_OFF;
    };
    auto store_saved = [&]() {
        const int gs = sv_g == OOB_OFF ? 0 : H * 4;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[0]), rG, sv_g, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[1]), rG, sv_g + gs, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[2]), rG, sv_g + 2 * gs, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[3]), rG, sv_g + 3 * gs, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[4]), rC, sv_c, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[5]), rH, sv_h, 0, 0);
    };
    // cell update of the owner lanes, hand-off store, activations kept for the deferred stores
    auto cell = [&](int s, int t, int growb, float (&pre)[4], bool recur) {
        if (owner) {
            float h = 0.f;
            float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cn = 0.f;
            if (valid && t < len) {
                if (recur) {
                    const int m = bl >> 4, r = bl & 15, nt = u >> 2, uu = u & 3;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        pre[g] += (red[s & 1][0][m][nt][r][g * 4 + uu] + red[s & 1][1][m][nt][r][g * 4 + uu]) +
                                  (red[s & 1][2][m][nt][r][g * 4 + uu] + red[s & 1][3][m][nt][r][g * 4 + uu]);
                }
                ig = sigmoid_f(pre[0]); fg = sigmoid_f(pre[1]);
                gg = tanh_f(pre[2]);    og = sigmoid_f(pre[3]);
                cn = fg * c + ig * gg;
                h = og * tanh_f(cn);
            }
            c = cn;   // padded rows keep a zero state (the reverse direction starts from it)
            // the 4 units of a batch row sit on 4 adjacent lanes: one 16-byte sc1 store per row
            // quad broadcasts on the VALU (DPP), not three trips through the LDS crossbar: the storing lane
            // is lane 0 of its quad (UN is a multiple of 4)
            const float hx0 = BF ? __uint_as_float(bl_hilo(h)) : h;     // BF: the exchanged copy is the {hi:lo} word
            const float h1 = ASR_DPP(hx0, 0x55), h2 = ASR_DPP(hx0, 0xAA), h3 = ASR_DPP(hx0, 0xFF);
            if ((u & 3) == 0 && !(dbg & 8)) {
                const float h = hx0;
                u32x4 pk = {__float_as_uint(h), __float_as_uint(h1), __float_as_uint(h2), __float_as_uint(h3)};
                if (tagmode) {
                    const unsigned e = (unsigned)((s >> 1) & 1);
                    pk = u32x4{tag_bits(h, e), tag_bits(h1, e), tag_bits(h2, e), tag_bits(h3, e)};
                }
                if (xlocal) __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc, (((s & 1) * NJ + jt) * chunk + bl * UN + u) * 4, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc, (((s & 1) * NJ + jt) * chunk + bl * UN + u) * 4, 0, 16);
            }
            sv[0] = ig; sv[1] = fg; sv[2] = gg; sv[3] = og; sv[4] = cn; sv[5] = h;
            sv_g = growb;
            sv_c = valid ? (int)(((((int64_t)t * B + bi) * 2 + d) * H + jo) * 4) : OOB_OFF;
            sv_h = valid ? (int)((t * ho_t + bi * ho_b + d * H + jo) * 4) : OOB_OFF;
        }
        if (!tagmode) {
            if (!(dbg & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(fl + jt, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // timing experiments (asr_debug_set bits 12..27): thresholds of the pace rule, or a fixed pace
    const int pace_up = ((dbg >> 12) & 15) ? ((dbg >> 12) & 15) : 3, pace_dn = (dbg >> 16) & 15, pace_fix = (dbg >> 20) & 0xf;
    int pace = pace_fix ? pace_fix - 1 : BLSTM_PACE_START, calm = 0;     // paced first attempt of the exchange gather, see the loop
    int s = s0;
    if (s == 0) {   // first step of the layer: no recurrent term, nothing pending
        const int t = d == 0 ? 0 : T - 1;
        const int growb = gate_offset(t);
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            pre[g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rG, growb + g * gstep, 0, 0));
        asm volatile("" ::"v"(pre[0]), "v"(pre[1]), "v"(pre[2]), "v"(pre[3]));
        cell(0, t, growb, pre, false);
        s = 1;
    }
    for (; s < s1; ++s) {
        const int t = d == 0 ? s : T - 1 - s;
        const int growb = gate_offset(t);
        if (!tagmode) {
            if (wave == 0 && !(dbg & 1)) {
                if (!poll_flags(fl, NJ, (unsigned)s)) {
                    if (lane == 0) {
                        s_abort = 1;
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            __syncthreads();
            if (s_abort) break;
        }
        const int pbase = ((s - 1) & 1) * NJ * chunk;
        const unsigned want = (unsigned)(((s - 1) >> 1) & 1);
        // exchange loads: 16-byte pieces a[m][f] of this lane's row(s) of h_{s-1}, first unit k = kof(f)
        constexpr int NA = BF ? 2 * KQ4 : KQ4;
        auto kof = [&](int f) -> int {
            return BF ? 32 * (wave + 4 * (f >> 1)) + 8 * q + 4 * (f & 1) : kb + 4 * f;
        };
        float4 a[MT][NA];
        float pre[4];
        bool ok = true;
        // Paced first attempt.  A gather issued the moment the step starts goes out ~0.6 us before the last producer's
        // h is visible, and a failed attempt costs a whole extra round trip (0.7 us); waiting first and reading once is
        // cheaper (forward 2.39 -> 2.05, backward 2.25 -> 2.07 us per step at H = 320 with the best fixed wait: 12 and 4
        // units).  The wait adapts per wave: every 16 steps it grows by one unit if 5 or more first attempts failed and
        // shrinks by one if at most 1 did (unit = s_sleep 2 = 128 cycles).  Only timing
        // depends on it: an early attempt still falls into the retry loop below.
        if (!(dbg & 4))
            for (int w_ = 0; w_ < pace; ++w_) __builtin_amdgcn_s_sleep(2);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int f = 0; f < NA; ++f) {
                const int k = kof(f);                 // first of 4 consecutive units
                a[m][f] = ld4_sc1(rsrc, k < H ? (pbase + (k / UN) * chunk + (m * 16 + i) * UN + k % UN) * 4 : OOB_OFF);
            }
        __builtin_amdgcn_sched_barrier(0);   // keep the issue order: loads return in order
        {
            const int so = shadow_offset(d == 0 ? s + lead : T - 1 - lead - s);
            const int goff = owner ? growb : so, gst = owner ? gstep : (so == OOB_OFF ? 0 : H * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pre[g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rG, goff + g * gst, 0, 0));
        }
        store_saved();
        if (PROD)   // progress word for the prefetchers: one lane of the group in range, write-through
            __builtin_amdgcn_raw_buffer_store_b32(prog_stamp | (unsigned)s, rP, prog_off, 0, 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int f = 0; f < NA; ++f) ok = ok & (tags_ok(a[m][f], want) | (kof(f) >= H));
        if (tagmode && !(dbg & 1)) {
            {
                calm += __all(ok) ? 0 : 0x100;                 // high byte: failed first attempts of this window
                if ((++calm & 0xff) == 16) {                   // every 16 steps: aim at 2..4 failures per window
                    const int fails = calm >> 8;
                    pace = fails >= pace_up ? min(pace + 1, BLSTM_PACE_MAX) : (fails <= pace_dn ? max(pace - 1, 0) : pace);
                    if (pace_fix) pace = pace_fix - 1;
                    calm = 0;
                }
            }
            unsigned long long t0 = 0;
            for (int spins = 0; !__all(ok); ++spins) {   // wave-uniform
                if (asr_wait_expired(spins, t0)) {
                    if (lane == 0) {
                        s_abort = 1;
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                ok = true;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int f = 0; f < NA; ++f) {
                        const int k = kof(f);
                        a[m][f] = ld4_sc1(rsrc, k < H ? (pbase + (k / UN) * chunk + (m * 16 + i) * UN + k % UN) * 4 : OOB_OFF);
                    }
                __builtin_amdgcn_sched_barrier(0);   // all loads in flight before the first check
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int f = 0; f < NA; ++f) ok = ok & (tags_ok(a[m][f], want) | (kof(f) >= H));
            }
        }
        if (BF) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int f = 0; f < KQ4; ++f) {
                    const float4 lo4 = a[0][2 * f], hi4 = a[0][2 * f + 1];
                    const unsigned ww[8] = {__float_as_uint(lo4.x), __float_as_uint(lo4.y), __float_as_uint(lo4.z),
                                            __float_as_uint(lo4.w), __float_as_uint(hi4.x), __float_as_uint(hi4.y),
                                            __float_as_uint(hi4.z), __float_as_uint(hi4.w)};
                    u32x4 ah, al;
                    bl_frags(ww, ah, al);
                    acc = bl_mfma3(ah, al, wbh[nt][f], wbl[nt][f], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) red[s & 1][wave][0][nt][q * 4 + r][i] = acc[r];
            }
        } else {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int f = 0; f < KQ4; ++f) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][f].x, wreg[nt][f].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][f].y, wreg[nt][f].y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][f].z, wreg[nt][f].z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][f].w, wreg[nt][f].w, acc1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) red[s & 1][wave][m][nt][q * 4 + r][i] = acc0[r] + acc1[r];
            }
        }
        // every wave consumes its gate loads here (not only the owner lanes): no load is left
        // pending across the loop's back edge, where the compiler would have to drain the
        // whole queue before reusing the registers
        asm volatile("" ::"v"(pre[0]), "v"(pre[1]), "v"(pre[2]), "v"(pre[3]));
        __syncthreads();
        if (s_abort) break;
        cell(s, t, growb, pre, true);
    }
    store_saved();   // the last processed step's saved activations
    if (PROD) __builtin_amdgcn_raw_buffer_store_b32(prog_stamp | 0xffffu, rP, prog_off, 0, 16);   // this group is done
}

// grid (NJT = H/16, 2, NBT = ceil(B/16)), block 256: workgroup = 16 hidden units x 16 batch
// rows of one direction.  The exchanged data is dG itself (G is overwritten in place).
template <int KQ4>
__global__ __launch_bounds__(256) void blstm_bwd_persist_kernel(
    const float *__restrict__ W_hhT, float *G, const float *__restrict__ Cs,
    const float *__restrict__ dHout, int64_t dho_t, int64_t dho_b, const int32_t *__restrict__ lengths,
    int T, int B, int H, unsigned *flags, unsigned *err, int dbg) {
    __shared__ float red[4][16][17];
    __shared__ __attribute__((aligned(16))) float dgl[16][4][16];
    __shared__ int s_abort;
    const bool vec16 = (H % 4) == 0;
    const int jt = blockIdx.x, d = blockIdx.y, bt = blockIdx.z, NJT = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    constexpr int KQ = KQ4 * 4;
    const int K = 4 * H;
    const int kb = (wave * 4 + q) * KQ;
    const int j0 = jt * 16, b0 = bt * 16;

    float4 wreg[KQ4];
    {
        const int j = j0 + i;
        const float *wrow = W_hhT + ((int64_t)d * H + j) * K;
#pragma unroll
        for (int f = 0; f < KQ4; ++f) wreg[f] = (j < H && kb + 4 * f < K) ? ld4(wrow + kb + 4 * f) : zero4();
    }
    const int bi = tid >> 4, ji = tid & 15;
    const int b = b0 + bi, j = j0 + ji;
    const bool valid = b < B && j < H;
    const int len = valid ? lengths[b] : 0;
    float dcarry = 0.f;
    const bool arow = (b0 + i) < B;
    __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(G, 0, (int)((int64_t)T * B * 8 * H * 4), 0x00020000);
    unsigned *fl = flags + (d * gridDim.z + bt) * NJT;
    if (tid == 0) s_abort = 0;
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? T - 1 - s : s;
        const int tn = d == 0 ? t + 1 : t - 1;   // processed in the previous step
        const int tp = d == 0 ? t - 1 : t + 1;   // the forward pass's previous time
        const int64_t grow = (((int64_t)t * B + b) * 2 + d) * 4 * H;
        float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cc = 0.f, cp = 0.f, dh = 0.f;
        const bool act = valid && t < len;
        if (act) {
            ig = G[grow + j]; fg = G[grow + H + j]; gg = G[grow + 2 * H + j]; og = G[grow + 3 * H + j];
            cc = Cs[(((int64_t)t * B + b) * 2 + d) * H + j];
            if (tp >= 0 && tp < T) cp = Cs[(((int64_t)tp * B + b) * 2 + d) * H + j];
            dh = dHout[t * dho_t + b * dho_b + d * H + j];
        }
        if (s > 0) {
            if (wave == 0 && !(dbg & 1)) {
                if (!poll_flags(fl, NJT, (unsigned)s)) {
                    if (lane == 0) {
                        s_abort = 1;
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            __syncthreads();
            if (s_abort) break;
            const int rowbase = (int)(((((int64_t)tn * B + b0 + i) * 2 + d) * 4 * H) * 4);   // bytes
            // issue every load of this step's dG rows before the first MFMA: one memory
            // round trip per step instead of one per chunk
            float4 a[KQ4];
#pragma unroll
            for (int f = 0; f < KQ4; ++f)
                a[f] = ld4_sc1(rsrc, (arow && kb + 4 * f < K) ? rowbase + (kb + 4 * f) * 4 : OOB_OFF);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < KQ4; ++f) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].x, wreg[f].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].y, wreg[f].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].z, wreg[f].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].w, wreg[f].w, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][q * 4 + r][i] = acc0[r] + acc1[r];
            __syncthreads();
        }
        if (valid) {
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
            if (act) {
                if (s > 0) dh += (red[0][bi][ji] + red[1][bi][ji]) + (red[2][bi][ji] + red[3][bi][ji]);
                const float tc = tanh_f(cc);
                const float dc = dcarry + dh * og * (1.f - tc * tc);
                d0 = dc * gg * ig * (1.f - ig);
                d1 = dc * cp * fg * (1.f - fg);
                d2 = dc * ig * (1.f - gg * gg);
                d3 = dh * tc * og * (1.f - og);
                dcarry = dc * fg;
            }
            dgl[bi][0][ji] = d0; dgl[bi][1][ji] = d1; dgl[bi][2][ji] = d2; dgl[bi][3][ji] = d3;
        }
        __syncthreads();
        {   // 16-byte write-through stores: thread -> (row, gate, 4 consecutive units)
            const int g = (tid >> 2) & 3, part = tid & 3;
            const int jj = j0 + part * 4;
            if (b < B && jj < H && !(dbg & 8)) {
                const float *src = &dgl[bi][g][part * 4];
                const int off = (int)((grow + g * H + jj) * 4);
                if (vec16 && jj + 3 < H) {
                    u32x4 pk = {__float_as_uint(src[0]), __float_as_uint(src[1]), __float_as_uint(src[2]),
                                __float_as_uint(src[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc, off, 0, 16);
                } else {
                    for (int e = 0; e < 4 && jj + e < H; ++e) st_sc1(G + grow + g * H + jj + e, src[e]);
                }
            }
        }
        if (!(dbg & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(fl + jt, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Backward recurrence, reduce-scatter form.  A workgroup owns UW hidden units (all four
// gates: KL = 4*UW rows of W_hh) of 16 batch rows.  After its cell backward it holds
// dG_t for those rows locally (LDS) and multiplies it with ITS rows of W_hh:
//     P[b][j] = sum_{n in own rows} dG_t[b][n] W_hh[n][j]      for ALL j   (M = j, N = b, K = KL)
// with W_hh's rows resident in registers for the whole sequence.  The partial products
// are handed to the other workgroups of the (direction, batch tile) group through
// xP[dir][bt][parity][src][H/16][16 rows][16 units] (every store instruction of a wave
// writes whole 128-byte lines); the owner of unit j sums the NJT partials in a
// fixed order (deterministic).  Compared with exchanging dG itself this moves
// 16*H*4 B out and in per workgroup and step instead of 16*4H*4 B in, and the
// MFMA operands never come from another workgroup.
// BF: the products on the bf16 matrix pipe (bf16x3 split): W_hh's rows sit in registers as hi / lo fragments, the
// workgroup's own dG goes through LDS as {hi:lo} words; the exchanged partial products stay fp32.
template <int UW, int NMT, bool BF = false, bool PROD = false>
__global__ __launch_bounds__(256) void blstm_bwd_rs_kernel(
    const float *__restrict__ W_hh, float *__restrict__ G, const float *__restrict__ Cs,
    const float *__restrict__ dHout, int64_t dho_t, int64_t dho_b, const int32_t *__restrict__ lengths,
    int64_t dh_bytes, int T, int B, int H, int s0, int s1, float *__restrict__ dc_ws, float *xP, unsigned *flags,
    unsigned *err, int dbg_arg, int tagmode_arg, int xcd_nbt_arg = 0, unsigned xcd_epoch = 0) {
    const int dbg = PROD ? 0 : dbg_arg, tagmode = PROD ? 1 : tagmode_arg;
    constexpr int KL = 4 * UW, KS = KL / 4, DSTR = KL + 4;
    constexpr int NI = 16 * UW / 4;     // 16-byte items per source workgroup (16 rows x UW/4)
    constexpr int NG = 256 / NI;        // thread groups that split the sources
    constexpr int KSB = KL / 32;        // BF: k-steps of 32 own gate rows
    __shared__ __attribute__((aligned(16))) float dgl[16][DSTR];
    __shared__ __attribute__((aligned(16))) float part[NG][NI][4];
    __shared__ int s_abort, s_local;
    // classic grid (NJT, 2, NBT), or the XCD-grouped one of the forward kernel (8 * NJT workgroups, label = id & 7)
    int jt = blockIdx.x, d = blockIdx.y, bt = blockIdx.z, NJT = gridDim.x, NBT = gridDim.z;
    const int pf_dist = PROD ? (xcd_nbt_arg >> 8) : 0;      // prefetch distance in steps (0: off), see blstm_prefetch_steps
    const int xcd_nbt = xcd_nbt_arg & 0xff;
    if (xcd_nbt > 0) {
        const int grp = blockIdx.x & 7;
        if (grp >= 2 * xcd_nbt) {
            if (pf_dist > 0)
                blstm_prefetch_steps(G, (int64_t)T * B * 8 * H * 4, 4 * H, Cs, (int64_t)T * B * 2 * H * 4, dHout, dh_bytes,
                                     dho_t, dho_b, lengths, T, B, H, s0 == 0 ? 1 : s0, s1, true, flags + BLSTM_PROG_WORD,
                                     2 * xcd_nbt, xcd_epoch & 0xffffu, pf_dist,
                                     ((grp - 2 * xcd_nbt) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3)) * 4 + (int)(threadIdx.x >> 6),
                                     (8 - 2 * xcd_nbt) * (int)(gridDim.x >> 3) * 4);
            return;
        }
        jt = blockIdx.x >> 3; d = grp & 1; bt = grp >> 1; NJT = gridDim.x >> 3; NBT = xcd_nbt;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int j0 = jt * UW, b0 = bt * 16;
    const int MTILES = (H + 15) / 16, Hp = MTILES * 16;

    // A operand: W_hh[own row k][j = 16*mt + i], k = 4*ks + q  (resident for all T steps)
    float areg[NMT][BF ? 1 : KS];
    u32x4 abh[BF ? NMT : 1][KSB], abl[BF ? NMT : 1][KSB];
#pragma unroll
    for (int mtl = 0; mtl < NMT; ++mtl) {
        const int jc = 16 * (wave + 4 * mtl) + i;
        if (BF) {
#pragma unroll
            for (int kb2 = 0; kb2 < KSB; ++kb2) {
                unsigned ww[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 32 * kb2 + 8 * q + e;
                    const int g = k / UW, ju = k % UW;
                    const bool ok = jc < H && j0 + ju < H;
                    ww[e] = bl_hilo(ok ? W_hh[((int64_t)d * 4 * H + g * H + j0 + ju) * H + jc] : 0.f);
                }
                bl_frags(ww, abh[mtl][kb2], abl[mtl][kb2]);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = 4 * ks + q;
                const int g = k / UW, ju = k % UW;
                const bool ok = jc < H && j0 + ju < H;
                areg[mtl][ks] = ok ? W_hh[((int64_t)d * 4 * H + g * H + j0 + ju) * H + jc] : 0.f;
            }
        }
    }
    const bool owner = tid < 16 * UW;
    const int bi = tid / UW, ju = tid % UW;
    const int b = b0 + bi, j = j0 + ju;
    const bool valid = owner && b < B && j < H;
    const int len = valid ? lengths[b] : 0;
    float dcarry = 0.f;
    if (s0 > 0 && valid) dcarry = dc_ws[((int64_t)d * B + b) * H + j];   // continuation launch
    const int64_t slotf = (int64_t)NJT * 16 * Hp;                       // floats per parity slot
    float *xg = xP + ((int64_t)(d * NBT + bt) * 2) * slotf;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(xg, 0, (int)(2 * slotf * 4), 0x00020000);
    unsigned *fl = flags + (d * NBT + bt) * NJT;
    if (tid == 0) { s_abort = 0; s_local = 0; }
    __syncthreads();
    // XCD-grouped grid: plain exchange stores when the whole group reports one XCD (see the forward kernel)
    if (xcd_nbt > 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xffu;
        unsigned *xid = flags + 512 + (d * NBT + bt) * 64;
        if (tid == 0) __hip_atomic_store(xid + jt, (xcd_epoch << 8) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == 0) {
            unsigned v = 0;
            bool ok = false;
            for (int spins = 0; spins < (1 << 16); ++spins) {
                v = lane < NJT ? __hip_atomic_load(xid + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((xcd_epoch << 8) | xcc);
                ok = (v >> 8) == xcd_epoch;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            const bool same = __all(ok && (v & 0xffu) == xcc);
            if (lane == 0) s_local = same && !(dbg & 16);
        }
        __syncthreads();
    }
    const bool xlocal = s_local != 0;
    if (!(dbg & 32)) __builtin_amdgcn_s_setprio(3);   // issue priority over co-resident waves of other kernels (side-stream GEMMs)

    // Vector-memory issue order inside a step, as in the forward kernel (s_waitcnt vmcnt counts
    // loads and stores in issue order): exchange loads of the partial products  ->  this step's
    // saved activations (HBM latency, needed after the reduction)  ->  the PREVIOUS step's dG
    // stores (nobody in the launch reads them)  ->  wait for the exchange loads only.  All of it
    // branch-free through buffer accesses (inactive lanes: out-of-range offset).
    __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(G, 0, (int)((int64_t)T * B * 8 * H * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Cs), 0,
                                                                  (int)((int64_t)T * B * 2 * H * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dHout), 0, (int)dh_bytes,
                                                                  0x00020000);
    float sv[4] = {0.f, 0.f, 0.f, 0.f};   // dG of the previous step, stored one step late
    int sv_g = OOB_OFF;
    // progress word of this (direction, batch tile) group, for the prefetching workgroups (blstm_prefetch_steps)
    __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(flags + BLSTM_PROG_WORD, 0, 8 * 4, 0x00020000);
    const int prog_off = (xcd_nbt > 0 && pf_dist > 0 && jt == 0 && tid == 0) ? (d * NBT + bt) * 4 : OOB_OFF;
    const unsigned prog_stamp = (xcd_epoch & 0xffffu) << 16;
    // the set-up loads are consumed here: a load pending at the loop entry would cost a
    // conservative queue drain in every iteration
#pragma unroll
    for (int mtl = 0; mtl < NMT; ++mtl) {
        if (BF) {
#pragma unroll
            for (int kb2 = 0; kb2 < KSB; ++kb2) {
                asm volatile("" ::"v"(abh[mtl][kb2].x), "v"(abh[mtl][kb2].y), "v"(abh[mtl][kb2].z), "v"(abh[mtl][kb2].w));
                asm volatile("" ::"v"(abl[mtl][kb2].x), "v"(abl[mtl][kb2].y), "v"(abl[mtl][kb2].z), "v"(abl[mtl][kb2].w));
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(areg[mtl][ks]));
        }
    }
    asm volatile("" ::"v"(dcarry), "v"(len));

    struct Act { float ig, fg, gg, og, cc, cp, dh; };
    auto load_act = [&](int t, bool act, int growb) -> Act {
        const int tp = d == 0 ? t - 1 : t + 1;   // the forward pass's previous time
        const int hs = act ? H * 4 : 0;
        const int cb = act ? (int)(((((int64_t)t * B + b) * 2 + d) * H + j) * 4) : OOB_OFF;
        const int cpb = (act && tp >= 0 && tp < T) ? (int)(((((int64_t)tp * B + b) * 2 + d) * H + j) * 4) : OOB_OFF;
        const int db = act ? (int)((t * dho_t + b * dho_b + d * H + j) * 4) : OOB_OFF;
        Act a;
        a.ig = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rG, growb, 0, 0));
        a.fg = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rG, growb + hs, 0, 0));
        a.gg = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rG, growb + 2 * hs, 0, 0));
        a.og = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rG, growb + 3 * hs, 0, 0));
        a.cc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rC, cb, 0, 0));
        a.cp = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rC, cpb, 0, 0));
        a.dh = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rD, db, 0, 0));
        return a;
    };
    auto store_saved = [&]() {
        const int hs = sv_g == OOB_OFF ? 0 : H * 4;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[0]), rG, sv_g, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[1]), rG, sv_g + hs, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[2]), rG, sv_g + 2 * hs, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[3]), rG, sv_g + 3 * hs, 0, 0);
    };
    // cell backward of the owner lanes (dh already complete), then the partial products of
    // this workgroup's dG rows and their hand-off
    auto cell = [&](int s, bool act, int growb, const Act &a, float dh) {
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        if (act) {
            const float tc = tanh_f(a.cc);
            const float dc = dcarry + dh * a.og * (1.f - tc * tc);
            d0 = dc * a.gg * a.ig * (1.f - a.ig);
            d1 = dc * a.cp * a.fg * (1.f - a.fg);
            d2 = dc * a.ig * (1.f - a.gg * a.gg);
            d3 = dh * tc * a.og * (1.f - a.og);
            dcarry = dc * a.fg;
        }
        sv[0] = d0; sv[1] = d1; sv[2] = d2; sv[3] = d3;
        sv_g = valid ? (int)((((((int64_t)(d == 0 ? T - 1 - s : s)) * B + b) * 2 + d) * 4 * H + j) * 4) : OOB_OFF;
        (void)growb;
        if (owner) {
            if (BF) {   // {hi:lo} words: split once by the owner, not by every reader
                dgl[bi][ju] = __uint_as_float(bl_hilo(d0)); dgl[bi][UW + ju] = __uint_as_float(bl_hilo(d1));
                dgl[bi][2 * UW + ju] = __uint_as_float(bl_hilo(d2)); dgl[bi][3 * UW + ju] = __uint_as_float(bl_hilo(d3));
            } else {
                dgl[bi][ju] = d0; dgl[bi][UW + ju] = d1; dgl[bi][2 * UW + ju] = d2; dgl[bi][3 * UW + ju] = d3;
            }
        }
        __syncthreads();
        if (s + 1 < T) {   // uniform
            f32x4 acc[NMT];
#pragma unroll
            for (int mtl = 0; mtl < NMT; ++mtl) acc[mtl] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (BF) {
#pragma unroll
                for (int kb2 = 0; kb2 < KSB; ++kb2) {
                    const float4 x0 = *reinterpret_cast<const float4 *>(&dgl[i][32 * kb2 + 8 * q]);
                    const float4 x1 = *reinterpret_cast<const float4 *>(&dgl[i][32 * kb2 + 8 * q + 4]);
                    const unsigned ww[8] = {__float_as_uint(x0.x), __float_as_uint(x0.y), __float_as_uint(x0.z),
                                            __float_as_uint(x0.w), __float_as_uint(x1.x), __float_as_uint(x1.y),
                                            __float_as_uint(x1.z), __float_as_uint(x1.w)};
                    u32x4 bh, bl;
                    bl_frags(ww, bh, bl);
#pragma unroll
                    for (int mtl = 0; mtl < NMT; ++mtl) acc[mtl] = bl_mfma3(abh[mtl][kb2], abl[mtl][kb2], bh, bl, acc[mtl]);
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float bv = dgl[i][4 * ks + q];
#pragma unroll
                for (int mtl = 0; mtl < NMT; ++mtl)
                    acc[mtl] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[mtl][ks], bv, acc[mtl], 0, 0, 0);
            }
            }
            if (!(dbg & 8)) {
#pragma unroll
                for (int mtl = 0; mtl < NMT; ++mtl) {
                    const int mt = wave + 4 * mtl;
                    u32x4 pk = {__float_as_uint(acc[mtl][0]), __float_as_uint(acc[mtl][1]),
                                __float_as_uint(acc[mtl][2]), __float_as_uint(acc[mtl][3])};
                    if (tagmode) {
                        const unsigned e = (unsigned)((s >> 1) & 1);
                        pk = u32x4{tag_bits(acc[mtl][0], e), tag_bits(acc[mtl][1], e), tag_bits(acc[mtl][2], e),
                                   tag_bits(acc[mtl][3], e)};
                    }
                    const int64_t off = (s & 1) * slotf + (((int64_t)jt * MTILES + mt) * 16 + i) * 16 + 4 * q;
                    if (xlocal) __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc, mt < MTILES ? (int)(off * 4) : OOB_OFF, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc, mt < MTILES ? (int)(off * 4) : OOB_OFF, 0, 16);
                }
            }
        }
        if (!tagmode) {
            if (!(dbg & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(fl + jt, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto gate_offset = [&](int t) -> int {
        return (valid && t < len) ? (int)(((((int64_t)t * B + b) * 2 + d) * 4 * H + j) * 4) : OOB_OFF;
    };

    // Through the XCD's L2 a failed first attempt costs a short retry, not a trip through the fabric, and waiting first
    // only loses time: no pacing then (alone 926 -> 857 us per launch, with cold data 1120 -> 983; in the step
    // 1222 -> 1047 us; tools/lstm_cold.py).  The write-through form keeps the adaptive rule.
    const int pace_up = ((dbg >> 12) & 15) ? ((dbg >> 12) & 15) : 3, pace_dn = (dbg >> 16) & 15;
    const int pace_fix = ((dbg >> 20) & 0xf) ? ((dbg >> 20) & 0xf) : (xlocal ? 1 : 0);
    int pace = pace_fix ? pace_fix - 1 : BLSTM_PACE_START, calm = 0;     // paced first attempt of the exchange gather
    int s = s0;
    if (s == 0) {   // first step: nothing to receive, nothing pending
        const int t = d == 0 ? T - 1 : 0;
        const bool act = valid && t < len;
        const int growb = gate_offset(t);
        const Act a = load_act(t, act, growb);
        asm volatile("" ::"v"(a.ig), "v"(a.fg), "v"(a.gg), "v"(a.og), "v"(a.cc), "v"(a.cp), "v"(a.dh));
        cell(0, act, growb, a, a.dh);
        s = 1;
    }
    // reduction roles: thread -> (16-byte item of this workgroup's 16 x UW block, group of sources)
    const int item = tid % NI, grp = tid / NI;
    const int ib = item / (UW / 4), ih = item % (UW / 4);
    const int jj = j0 + 4 * ih;                  // first unit of this item
    // sources per thread group: NJT <= 4*NMT*(16/UW) and NG = 64/UW*... = 256/NI groups -> NMT loads each
    constexpr int CH = NMT;
    static_assert(CH <= 8, "exchange loads per lane");
    // saved activations one step ahead, as in the forward kernel (cold in the training step)
    Act a_next, a_next2;
    {
        const int tn = d == 0 ? T - 1 - s : s;
        const bool actn = valid && s < s1 && tn < len;
        a_next = load_act(tn, actn, actn ? gate_offset(tn) : OOB_OFF);
        const int s2 = s + 1, tn2 = d == 0 ? T - 1 - s2 : s2;
        const bool actn2 = valid && s2 < s1 && tn2 < len;
        a_next2 = load_act(tn2, actn2, actn2 ? gate_offset(tn2) : OOB_OFF);
    }
    for (; s < s1; ++s) {
        const int t = d == 0 ? T - 1 - s : s;
        const bool act = valid && t < len;
        const int growb = gate_offset(t);
        if (!tagmode) {
            if (wave == 0 && !(dbg & 1)) {
                if (!poll_flags(fl, NJT, (unsigned)s)) {
                    if (lane == 0) {
                        s_abort = 1;
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            __syncthreads();
            if (s_abort) break;
        }
        // sum the NJT partial products of the previous step: every thread issues its share of
        // the 16-byte loads at once (one memory round trip), then a fixed-order reduction
        // through LDS.  CH*NG >= NJT by construction; the sweep loop after the fast path is a
        // safety net only.
        const int off0 = (int)((((s - 1) & 1) * slotf + ((int64_t)(jj >> 4) * 16 + ib) * 16 + (jj & 15)) * 4);
        const unsigned want = (unsigned)(((s - 1) >> 1) & 1);
        float4 v[CH];
        if (!(dbg & 4))      // paced first attempt, as in the forward kernel
            for (int w_ = 0; w_ < pace; ++w_) __builtin_amdgcn_s_sleep(2);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int w = grp + c * NG;
            v[c] = ld4_sc1(rsrc, w < NJT ? off0 + (int)((int64_t)w * MTILES * 256 * 4) : OOB_OFF);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the issue order: loads return in order
        const Act a = a_next;
        a_next = a_next2;
        {
            const int sn = s + 2, tn = d == 0 ? T - 1 - sn : sn;
            const bool actn = valid && sn < s1 && tn < len;
            a_next2 = load_act(tn, actn, actn ? gate_offset(tn) : OOB_OFF);
        }
        store_saved();
        if (PROD)   // progress word for the prefetchers: one lane of the group in range, write-through
            __builtin_amdgcn_raw_buffer_store_b32(prog_stamp | (unsigned)s, rP, prog_off, 0, 16);
        __builtin_amdgcn_sched_barrier(0);
        bool ok = true;
#pragma unroll
        for (int c = 0; c < CH; ++c) ok = ok & (tags_ok(v[c], want) | (grp + c * NG >= NJT));
        if (tagmode && !(dbg & 1)) {
            {
                calm += __all(ok) ? 0 : 0x100;                 // high byte: failed first attempts of this window
                if ((++calm & 0xff) == 16) {                   // every 16 steps: aim at 2..4 failures per window
                    const int fails = calm >> 8;
                    pace = fails >= pace_up ? min(pace + 1, BLSTM_PACE_MAX) : (fails <= pace_dn ? max(pace - 1, 0) : pace);
                    if (pace_fix) pace = pace_fix - 1;
                    calm = 0;
                }
            }
            unsigned long long t0 = 0;
            for (int spins = 0; !__all(ok); ++spins) {   // wave-uniform
                if (asr_wait_expired(spins, t0)) {
                    if (lane == 0) {
                        s_abort = 1;
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int w = grp + c * NG;
                    v[c] = ld4_sc1(rsrc, w < NJT ? off0 + (int)((int64_t)w * MTILES * 256 * 4) : OOB_OFF);
                }
                __builtin_amdgcn_sched_barrier(0);   // all loads in flight before the first check
                ok = true;
#pragma unroll
                for (int c = 0; c < CH; ++c) ok = ok & (tags_ok(v[c], want) | (grp + c * NG >= NJT));
            }
        }
        float4 sum = zero4();
#pragma unroll
        for (int c = 0; c < CH; ++c) { sum.x += v[c].x; sum.y += v[c].y; sum.z += v[c].z; sum.w += v[c].w; }
        for (int w0 = grp + CH * NG; w0 < NJT; w0 += NG) {   // rare: more than CH*NG sources
            float4 x = zero4();
            unsigned long long t1 = 0;
            for (int spins = 0;; ++spins) {
                x = ld4_sc1(rsrc, off0 + (int)((int64_t)w0 * MTILES * 256 * 4));
                if (!tagmode || (dbg & 1) || __all(tags_ok(x, want))) break;
                if (asr_wait_expired(spins, t1)) {
                    if (lane == 0) {
                        s_abort = 1;
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            sum.x += x.x; sum.y += x.y; sum.z += x.z; sum.w += x.w;
        }
        *reinterpret_cast<float4 *>(&part[grp][item][0]) = sum;
        // every wave consumes its activation loads here (not only the owner lanes): no load
        // is left pending across the loop's back edge
        asm volatile("" ::"v"(a.ig), "v"(a.fg), "v"(a.gg), "v"(a.og), "v"(a.cc), "v"(a.cp), "v"(a.dh));
        __syncthreads();
        if (s_abort) break;
        float dh = a.dh;
        if (act) {
            const int it2 = bi * (UW / 4) + (ju >> 2);
            float ps = 0.f;
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) ps += part[gq][it2][ju & 3];
            dh += ps;
        }
        cell(s, act, growb, a, dh);
    }
    store_saved();   // the last processed step's dG
    if (PROD) __builtin_amdgcn_raw_buffer_store_b32(prog_stamp | 0xffffu, rP, prog_off, 0, 16);   // this group is done
    if (s1 < T && valid) dc_ws[((int64_t)d * B + b) * H + j] = dcarry;   // picked up by the next launch
}

// One launch instead of two memsets before a persistent launch: zero the flag words, fill the exchange buffer with
// 0xFF (every exchanged float starts with the "wrong" epoch tag, LSB = 1).  n16 = 16-byte pieces of the exchange.
__global__ __launch_bounds__(256) void blstm_init_ws_kernel(unsigned *flags, int nflags, uint4 *xch, int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    if (i < nflags) flags[i] = 0u;
    const uint4 ff = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    for (int64_t k = i; k < n16; k += nth) xch[k] = ff;      // grid-stride: a few 16-byte stores per thread
}
static void blstm_init_ws(unsigned *flags, void *xch, int64_t xch_bytes, hipStream_t st) {
    // 64 bytes per thread (the backward's 3.3 MB exchange took 27 us at one 16-byte store per thread, 800 workgroups)
    const int64_t n16 = xch_bytes / 16, want = (n16 + 3) / 4, n = want > BLSTM_FLAG_WORDS_ ? want : BLSTM_FLAG_WORDS_;
    hipLaunchKernelGGL(blstm_init_ws_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, flags, BLSTM_FLAG_WORDS_,
                       static_cast<uint4 *>(xch), n16);
}

// ------------------------------------------------------------ host dispatch --
#define BLSTM_FLAG_WORDS 1024   // zeroed before every launch (multiple of 16 bytes, start of ws)
#define BLSTM_ERR_WORD 1024     // sticky: set by a kernel whose poll timed out
#define BLSTM_HX_OFF 2048

// Hidden units per backward workgroup.  16 where the kernel is instantiated for it (H > 192): half as many,
// fatter workgroups -- every workgroup publishes partial products for ALL units each step, so the exchange
// traffic of a (direction, batch tile) group scales with its workgroup count, and the 40 extra MFMAs per step
// cost less than that (config-2 layer: 3.41 -> 3.14 us per time step alone, 1867 -> 1757 us per launch inside
// the training step).  8 below that, while the grid still fits one workgroup per CU.  ASR_BWD_UW=8 / 16 forces.
static int bwd_unit_width(int B, int H) {
    static int force = -1;
    if (force < 0) {
        const char *e = getenv("ASR_BWD_UW");
        force = e ? atoi(e) : 0;
    }
    const bool fits8 = cdiv(H, 8) * 2 * cdiv(B, 16) <= 256 && cdiv(H, 8) <= 128;
    if (force == 16 || !fits8) return 16;
    if (force == 8) return 8;
    return cdiv(cdiv(H, 16), 4) >= 4 ? 16 : 8;
}

extern "C" int64_t asr_blstm_ws_floats(int B, int H) {
    const int64_t fwd = (int64_t)4 * cdiv(H, 4) * cdiv(B, 16) * 64;
    const int uw = bwd_unit_width(B, H);
    const int64_t bwd = (int64_t)2 * cdiv(B, 16) * 2 * cdiv(H, uw) * 16 * (cdiv(H, 16) * 16);
    return BLSTM_HX_OFF + (fwd > bwd ? fwd : bwd) + 64;
}

// timing experiments only (tools/lstm_bench.py): bit 0 skips the in-launch waits, bit 1 the
// store drain -- results are then WRONG; never set by the product path.
static int g_dbg_init() {
    const char *e = getenv("ASR_BLSTM_DEBUG");          // timing experiments only, see asr_debug_set
    const int f = e ? (int)strtol(e, nullptr, 0) : 0;
    return (f & 0xff) | (f & 0x0ffff000);
}
static int g_dbg = g_dbg_init();
static int g_bwd_form = 0;   // bit 8 of the debug flags selects the older dG-exchange kernel (A/B timing)
static int g_fwd_form = 0;   // bit 10: forward kernel with both batch tiles in one workgroup (A/B timing)
static int g_tagmode = 1;    // bit 9 selects the flag protocol instead of the in-band epoch tags
extern "C" void asr_debug_set(int flags) {
    g_dbg = (flags & 0xff) | (flags & 0x0ffff000);
    g_bwd_form = (flags >> 8) & 1;
    g_tagmode = ((flags >> 9) & 1) ? 0 : 1;
    g_fwd_form = (flags >> 10) & 1;
}

static bool al16p(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// Arithmetic of the recurrent products h W_hh^T / dG W_hh inside the persistent kernels: 0 = exact fp32
// (v_mfma_f32_16x16x4_f32), 1 = bf16x3 split on v_mfma_f32_16x16x32_bf16 (shapes with H % 32 == 0, one batch
// tile per workgroup; others stay fp32).  ASR_BLSTM_PRECISION = f32 | bf16x3 sets the start-up value.
#define BLSTM_PRECISION_DEFAULT 1
static int g_blstm_precision = -1;
static int blstm_precision() {
    if (g_blstm_precision < 0) {
        const char *e = getenv("ASR_BLSTM_PRECISION");
        g_blstm_precision = BLSTM_PRECISION_DEFAULT;
        if (e && (!strcmp(e, "f32") || !strcmp(e, "0"))) g_blstm_precision = 0;
        if (e && (!strcmp(e, "bf16x3") || !strcmp(e, "1"))) g_blstm_precision = 1;
    }
    return g_blstm_precision;
}
extern "C" int asr_blstm_precision(int mode) {
    const int old = blstm_precision();
    if (mode == 0 || mode == 1) g_blstm_precision = mode;
    return old;
}

// Grid of the persistent kernels at 17..32 batch rows: 1 = one (direction, batch tile) exchange group per XCD label
// with the verified same-XCD hand-off (default), 0 = the three-dimensional grid with write-through stores only.
// ASR_BLSTM_XCD = 0 | 1 sets the start-up value.  Results are bit-identical either way.
static int g_blstm_xcd = -1;
static int blstm_xcd_groups() {
    if (g_blstm_xcd < 0) {
        const char *e = getenv("ASR_BLSTM_XCD");
        g_blstm_xcd = (e && atoi(e) == 0) ? 0 : 1;
    }
    return g_blstm_xcd;
}
extern "C" int asr_blstm_xcd_groups(int mode) {
    const int old = blstm_xcd_groups();
    if (mode == 0 || mode == 1) g_blstm_xcd = mode;
    return old;
}
// Steps the prefetching workgroups of the XCD-grouped grids stay ahead of the recurrence (blstm_prefetch_steps);
// ASR_BLSTM_PREFETCH=0 switches them off, asr_blstm_prefetch(n) at run time (returns the previous value; n < 0 queries).
static int g_blstm_pf = -1;
static int blstm_prefetch_dist() {
    if (g_blstm_pf < 0) {
        const char *e = getenv("ASR_BLSTM_PREFETCH");
        g_blstm_pf = e ? atoi(e) : 16;
        if (g_blstm_pf < 0) g_blstm_pf = 0;
        if (g_blstm_pf > 255) g_blstm_pf = 255;
    }
    return g_blstm_pf;
}
static int blstm_prefetch_dist_fwd() {      // timing experiments: ASR_BLSTM_PREFETCH_FWD overrides the distance for the forward kernel
    static int v = -2;
    if (v == -2) {
        const char *e = getenv("ASR_BLSTM_PREFETCH_FWD");
        v = e ? atoi(e) : -1;
        if (v > 255) v = 255;
    }
    return v >= 0 ? v : blstm_prefetch_dist();
}
extern "C" int asr_blstm_prefetch(int dist) {
    const int old = blstm_prefetch_dist();
    if (dist >= 0) g_blstm_pf = dist > 255 ? 255 : dist;
    return old;
}
// stamp of a launch's XCD reports (24 bits, never 0): words left by an earlier launch do not count
static unsigned blstm_xcd_epoch() {
    static unsigned epoch = 0;
    epoch = (epoch + 1) & 0xffffffu;
    if (epoch == 0) epoch = 1;
    return epoch;
}

template <int NT>
static bool launch_fwd_persist_bf(int nks, dim3 grid, hipStream_t st, const float *W_hh, float *G, float *Cs,
                                  float *Hout, int64_t ho_t, int64_t ho_b, const int32_t *lengths, int T, int B,
                                  int H, int s0, int s1, float *hx, unsigned *flags, unsigned *err, int xcd_nbt = 0,
                                  unsigned xcd_epoch = 0) {
    const int64_t hout_bytes = ((int64_t)(T - 1) * ho_t + (int64_t)(B - 1) * ho_b + 2 * H) * 4;
#define FWD_BF_CASE(K)                                                                                           \
    case K:                                                                                                      \
        if (g_dbg == 0 && g_tagmode == 1)                                                                        \
            hipLaunchKernelGGL((blstm_fwd_persist_kernel<1, NT, K, true, true>), grid, dim3(256), 0, st, W_hh, G, \
                               Cs, Hout, ho_t, ho_b, hout_bytes, lengths, T, B, H, s0, s1, hx, flags, err, 0, 1, \
                               xcd_nbt, xcd_epoch);                                                              \
        else                                                                                                     \
            hipLaunchKernelGGL((blstm_fwd_persist_kernel<1, NT, K, true>), grid, dim3(256), 0, st, W_hh, G, Cs,  \
                               Hout, ho_t, ho_b, hout_bytes, lengths, T, B, H, s0, s1, hx, flags, err, g_dbg,    \
                               g_tagmode, xcd_nbt, xcd_epoch);                                                   \
        return true;
    switch (nks) { FWD_BF_CASE(1) FWD_BF_CASE(2) FWD_BF_CASE(3) FWD_BF_CASE(4) }
#undef FWD_BF_CASE
    return false;
}

template <int MT, int NT>
static bool launch_fwd_persist(int kq4, dim3 grid, hipStream_t st, const float *W_hh, float *G, float *Cs,
                               float *Hout, int64_t ho_t, int64_t ho_b, const int32_t *lengths, int T, int B,
                               int H, int s0, int s1, float *hx, unsigned *flags, unsigned *err) {
    // byte extent of Hout as the kernel addresses it (largest element offset + 1)
    const int64_t hout_bytes = ((int64_t)(T - 1) * ho_t + (int64_t)(B - 1) * ho_b + 2 * H) * 4;
#define FWD_CASE(K)                                                                                       \
    case K:                                                                                               \
        hipLaunchKernelGGL((blstm_fwd_persist_kernel<MT, NT, K>), grid, dim3(256), 0, st, W_hh, G, Cs,    \
                           Hout, ho_t, ho_b, hout_bytes, lengths, T, B, H, s0, s1, hx, flags, err, g_dbg, \
                           g_tagmode);                                                                    \
        return true;
    switch (kq4) {
        FWD_CASE(1) FWD_CASE(2) FWD_CASE(3) FWD_CASE(4) FWD_CASE(5) FWD_CASE(6) FWD_CASE(7) FWD_CASE(8)
    }
#undef FWD_CASE
    return false;
}

// Steps [s0, s1) of the layer (step s = frame s of the forward direction and frame T-1-s of the
// reverse direction).  A launch with s0 > 0 continues the previous one on the same ws: the
// exchange buffer still holds h_{s0-1} with the right epoch tags and c_{s0-1} is read from Cs.
extern "C" int asr_blstm_fwd_range(const float *W_hh, float *G, float *Cs, float *Hout, int64_t ho_t,
                                   int64_t ho_b, const int32_t *lengths, int T, int B, int H, int s0, int s1,
                                   float *ws, void *stream) {
    ASR_CHECK_ARG(W_hh && G && Cs && Hout && lengths, "blstm_fwd: null pointer");
    ASR_CHECK_ARG(T > 0 && B > 0 && H > 0, "blstm_fwd: bad shape T=%d B=%d H=%d", T, B, H);
    ASR_CHECK_ARG(0 <= s0 && s0 < s1 && s1 <= T, "blstm_fwd: bad step range [%d, %d) of %d", s0, s1, T);
    hipStream_t st = (hipStream_t)stream;
    const int NJ = cdiv(H, 4), MT = cdiv(B, 16);
    const int kq4 = (cdiv(H, 16) + 3) / 4;
    // the persistent kernel addresses G / Cs / Hout through 32-bit buffer offsets
    const int64_t big = (int64_t)0x7fff0000;
    const bool persist = ws != nullptr && H % 4 == 0 && NJ <= 128 && MT <= 2 && kq4 <= 8 && al16p(W_hh) &&
                         al16p(ws) && (int64_t)T * B * 8 * H * 4 < big && ho_t >= 0 && ho_b >= 0 &&
                         ((int64_t)(T - 1) * ho_t + (int64_t)(B - 1) * ho_b + 2 * H) * 4 < big;
    if (!persist) return blstm_fwd_steps(W_hh, G, Cs, Hout, ho_t, ho_b, lengths, T, B, H, s0, s1, stream);
    unsigned *flags = reinterpret_cast<unsigned *>(ws);
    unsigned *err = flags + BLSTM_ERR_WORD;
    float *hx = ws + BLSTM_HX_OFF;
    if (s0 == 0) {
        if (g_tagmode) blstm_init_ws(flags, hx, (int64_t)sizeof(float) * 4 * NJ * MT * 64, st);
        else ASR_CHECK_HIP(hipMemsetAsync(flags, 0, sizeof(unsigned) * BLSTM_FLAG_WORDS, st));
    }
    bool ok;
    const bool bf = blstm_precision() == 1 && H % 32 == 0 && H <= 512 && g_tagmode;
    if (bf && MT == 1) {
        ok = launch_fwd_persist_bf<1>(cdiv(H / 32, 4), dim3(NJ, 2, 1), st, W_hh, G, Cs, Hout, ho_t, ho_b, lengths, T, B,
                                      H, s0, s1, hx, flags, err);
    } else if (bf && g_fwd_form == 0 && blstm_xcd_groups() && H / 8 <= 64) {
        // one (direction, batch tile) group of H/8 workgroups per XCD label (8 hidden units per workgroup)
        ok = launch_fwd_persist_bf<2>(cdiv(H / 32, 4), dim3(8 * (H / 8), 1, 1), st, W_hh, G, Cs, Hout, ho_t, ho_b, lengths,
                                      T, B, H, s0, s1, hx, flags, err,
                                      (2 * MT < 8 && T < 0xffff) ? (MT | (blstm_prefetch_dist_fwd() << 8)) : MT, blstm_xcd_epoch());
    } else if (bf && g_fwd_form == 0) {
        ok = launch_fwd_persist_bf<2>(cdiv(H / 32, 4), dim3(H / 8, 2, 2), st, W_hh, G, Cs, Hout, ho_t, ho_b, lengths, T,
                                      B, H, s0, s1, hx, flags, err);
    } else if (MT == 1) {
        ok = launch_fwd_persist<1, 1>(kq4, dim3(NJ, 2, 1), st, W_hh, G, Cs, Hout, ho_t, ho_b, lengths, T, B, H, s0,
                                      s1, hx, flags, err);
    } else if (H % 8 == 0 && g_fwd_form == 0) {   // two batch tiles on separate workgroups of 8 units
        ok = launch_fwd_persist<1, 2>(kq4, dim3(H / 8, 2, 2), st, W_hh, G, Cs, Hout, ho_t, ho_b, lengths, T, B, H,
                                      s0, s1, hx, flags, err);
    } else {
        ok = launch_fwd_persist<2, 1>(kq4, dim3(NJ, 2, 1), st, W_hh, G, Cs, Hout, ho_t, ho_b, lengths, T, B, H, s0,
                                      s1, hx, flags, err);
    }
    ASR_CHECK_ARG(ok, "blstm_fwd: no persistent instantiation for H=%d", H);
    ASR_CHECK_LAUNCH("blstm_fwd_persist_kernel");
    return ASR_OK;
}

extern "C" int asr_blstm_fwd(const float *W_hh, float *G, float *Cs, float *Hout, int64_t ho_t,
                             int64_t ho_b, const int32_t *lengths, int T, int B, int H, float *ws,
                             void *stream) {
    ASR_CHECK_ARG(T > 0, "blstm_fwd: bad shape T=%d B=%d H=%d", T, B, H);
    return asr_blstm_fwd_range(W_hh, G, Cs, Hout, ho_t, ho_b, lengths, T, B, H, 0, T, ws, stream);
}

// Steps [s0, s1) of the backward recurrence (step s = frame T-1-s of the forward direction and
// frame s of the reverse direction); a launch with s0 > 0 continues the previous one on the
// same ws and dc_ws.
// 1 if asr_blstm_bwd / asr_blstm_bwd_range will read W_hhT for this shape (the step-wise kernels and the dG-exchange
// form do), 0 if the reduce-scatter form runs, which only reads W_hh: the caller may then pass W_hh in both places
// and skip the transposition.  Mirrors the dispatch below; a call that does need W_hhT rejects W_hhT == W_hh.
extern "C" int asr_blstm_bwd_needs_transposed(int T, int B, int H, int64_t dho_t, int64_t dho_b, int have_ws) {
    const int NJT = cdiv(H, 16), NBT = cdiv(B, 16);
    const int64_t gbytes = (int64_t)T * B * 8 * H * 4;
    const bool persist = have_ws && NJT <= 128 && NJT * 2 * NBT <= 256 && 2 * NBT * NJT <= BLSTM_FLAG_WORDS &&
                         gbytes < (int64_t)0x7fffffff;
    if (!persist || g_bwd_form != 0) return 1;
    const int uw = bwd_unit_width(B, H);
    const int njt = cdiv(H, uw), nmt = cdiv(cdiv(H, 16), 4);
    const int64_t xbytes = (int64_t)2 * njt * 16 * (cdiv(H, 16) * 16) * 4;
    const int64_t dh_bytes = ((int64_t)(T - 1) * dho_t + (int64_t)(B - 1) * dho_b + 2 * H) * 4;
    const bool rs = njt <= 128 && njt * 2 * NBT <= 256 && 2 * NBT * njt <= BLSTM_FLAG_WORDS && xbytes < 0x7fffffff &&
                    gbytes < (int64_t)0x7fff0000 && dho_t >= 0 && dho_b >= 0 && dh_bytes < (int64_t)0x7fff0000 &&
                    (uw == 8 ? (nmt >= 1 && nmt <= 6) : (nmt >= 4 && nmt <= 8));
    return rs ? 0 : 1;
}

extern "C" int asr_blstm_bwd_range(const float *W_hh, const float *W_hhT, float *G, const float *Cs,
                                   const float *dHout, int64_t dho_t, int64_t dho_b, const int32_t *lengths,
                                   float *dc_ws, int T, int B, int H, int s0, int s1, float *ws, void *stream) {
    ASR_CHECK_ARG(W_hhT && G && Cs && dHout && lengths && dc_ws, "blstm_bwd: null pointer");
    ASR_CHECK_ARG(T > 0 && B > 0 && H > 0, "blstm_bwd: bad shape T=%d B=%d H=%d", T, B, H);
    ASR_CHECK_ARG(0 <= s0 && s0 < s1 && s1 <= T, "blstm_bwd: bad step range [%d, %d) of %d", s0, s1, T);
    const bool full = s0 == 0 && s1 == T;
    hipStream_t st = (hipStream_t)stream;
    const int NJT = cdiv(H, 16), NBT = cdiv(B, 16);
    const int kq4 = (cdiv(4 * H, 16) + 3) / 4;
    const int64_t gbytes = (int64_t)T * B * 8 * H * 4;
    bool persist = ws != nullptr && NJT <= 128 && NJT * 2 * NBT <= 256 && 2 * NBT * NJT <= BLSTM_FLAG_WORDS &&
                   gbytes < (int64_t)0x7fffffff && al16p(W_hhT) && al16p(G) && al16p(ws);
    unsigned *flags = reinterpret_cast<unsigned *>(ws);
    unsigned *err = flags ? flags + BLSTM_ERR_WORD : nullptr;
    // preferred: reduce-scatter form (needs the untransposed W_hh)
    if (persist && W_hh != nullptr && g_bwd_form == 0) {
        const int uw = bwd_unit_width(B, H);
        const int njt = cdiv(H, uw), nmt = cdiv(cdiv(H, 16), 4);
        const int64_t xbytes = (int64_t)2 * njt * 16 * (cdiv(H, 16) * 16) * 4;
        // G / Cs / dHout are addressed through 32-bit buffer offsets
        const int64_t dh_bytes = ((int64_t)(T - 1) * dho_t + (int64_t)(B - 1) * dho_b + 2 * H) * 4;
        if (njt <= 128 && njt * 2 * NBT <= 256 && 2 * NBT * njt <= BLSTM_FLAG_WORDS && xbytes < 0x7fffffff &&
            gbytes < (int64_t)0x7fff0000 && dho_t >= 0 && dho_b >= 0 && dh_bytes < (int64_t)0x7fff0000) {
            dim3 g2(njt, 2, NBT);
            int xcd_nbt = 0;
            unsigned xcd_epoch = 0;
            if (blstm_xcd_groups() && g_tagmode && 2 * NBT <= 8 && NBT > 1 && njt <= 64) {   // one (direction, batch tile) group per XCD label
                g2 = dim3(8 * njt, 1, 1);
                xcd_nbt = NBT;
                xcd_epoch = blstm_xcd_epoch();
            }
            // idle XCD labels prefetch the saved activations (blstm_prefetch_steps): product instantiation only
            const int xcd_pf = (xcd_nbt > 0 && 2 * NBT < 8 && T < 0xffff) ? (xcd_nbt | (blstm_prefetch_dist() << 8)) : xcd_nbt;
            float *xP = ws + BLSTM_HX_OFF;
            bool launched = true;
            if (s0 == 0) {
                if (g_tagmode) blstm_init_ws(flags, xP, (int64_t)xbytes * 2 * NBT, st);
                else ASR_CHECK_HIP(hipMemsetAsync(flags, 0, sizeof(unsigned) * BLSTM_FLAG_WORDS, st));
            }
            const bool bfb = blstm_precision() == 1;
#define RS_CASE(U, N)                                                                                        \
    do {                                                                                                     \
        if (bfb && g_dbg == 0 && g_tagmode == 1)                                                             \
            hipLaunchKernelGGL((blstm_bwd_rs_kernel<U, N, true, true>), g2, dim3(256), 0, st, W_hh, G, Cs,   \
                               dHout, dho_t, dho_b, lengths, dh_bytes, T, B, H, s0, s1, dc_ws, xP, flags,    \
                               err, 0, 1, xcd_pf, xcd_epoch);                                                \
        else if (bfb)                                                                                        \
            hipLaunchKernelGGL((blstm_bwd_rs_kernel<U, N, true>), g2, dim3(256), 0, st, W_hh, G, Cs, dHout,  \
                               dho_t, dho_b, lengths, dh_bytes, T, B, H, s0, s1, dc_ws, xP, flags, err,      \
                               g_dbg, g_tagmode, xcd_nbt, xcd_epoch);                                        \
        else                                                                                                 \
            hipLaunchKernelGGL((blstm_bwd_rs_kernel<U, N>), g2, dim3(256), 0, st, W_hh, G, Cs, dHout, dho_t, \
                               dho_b, lengths, dh_bytes, T, B, H, s0, s1, dc_ws, xP, flags, err, g_dbg,      \
                               g_tagmode, xcd_nbt, xcd_epoch);                                               \
    } while (0)
            if (uw == 8) {
                switch (nmt) {
                    case 1: RS_CASE(8, 1); break;
                    case 2: RS_CASE(8, 2); break;
                    case 3: RS_CASE(8, 3); break;
                    case 4: RS_CASE(8, 4); break;
                    case 5: RS_CASE(8, 5); break;
                    case 6: RS_CASE(8, 6); break;
                    default: launched = false;
                }
            } else {
                switch (nmt) {
                    case 4: RS_CASE(16, 4); break;
                    case 5: RS_CASE(16, 5); break;
                    case 6: RS_CASE(16, 6); break;
                    case 7: RS_CASE(16, 7); break;
                    case 8: RS_CASE(16, 8); break;
                    default: launched = false;
                }
            }
#undef RS_CASE
            if (launched) {
                ASR_CHECK_LAUNCH("blstm_bwd_rs_kernel");
                return ASR_OK;
            }
        }
    }
    ASR_CHECK_ARG(W_hhT != W_hh, "blstm_bwd: this shape needs the transposed W_hh (asr_blstm_bwd_needs_transposed)");
    dim3 grid(NJT, 2, NBT);
    if (!full) persist = false;   // the dG-exchange kernel (A/B timing only) runs whole layers
    if (persist) {
        ASR_CHECK_HIP(hipMemsetAsync(flags, 0, sizeof(unsigned) * BLSTM_FLAG_WORDS, st));
#define BWD_CASE(K)                                                                                        \
    case K:                                                                                                \
        hipLaunchKernelGGL((blstm_bwd_persist_kernel<K>), grid, dim3(256), 0, st, W_hhT, G, Cs, dHout,     \
                           dho_t, dho_b, lengths, T, B, H, flags, err, g_dbg);                             \
        break;
        switch (kq4) {
            BWD_CASE(1) BWD_CASE(2) BWD_CASE(3) BWD_CASE(4) BWD_CASE(5) BWD_CASE(6) BWD_CASE(8) BWD_CASE(10)
            BWD_CASE(12) BWD_CASE(16) BWD_CASE(20)
            default: persist = false;
        }
#undef BWD_CASE
    }
    if (!persist)
        return blstm_bwd_steps(W_hhT, G, Cs, dHout, dho_t, dho_b, lengths, dc_ws, T, B, H, s0, s1, stream);
    ASR_CHECK_LAUNCH("blstm_bwd_persist_kernel");
    return ASR_OK;
}

extern "C" int asr_blstm_bwd(const float *W_hh, const float *W_hhT, float *G, const float *Cs,
                             const float *dHout, int64_t dho_t, int64_t dho_b, const int32_t *lengths,
                             float *dc_ws, int T, int B, int H, float *ws, void *stream) {
    ASR_CHECK_ARG(T > 0, "blstm_bwd: bad shape T=%d B=%d H=%d", T, B, H);
    return asr_blstm_bwd_range(W_hh, W_hhT, G, Cs, dHout, dho_t, dho_b, lengths, dc_ws, T, B, H, 0, T, ws, stream);
}

// ------------------------------------------------ allocation-free entry point --
// For callers that cannot allocate device memory (code under HIP graph capture, tight serving
// loops, harnesses that forbid hipMalloc): the forward recurrence of asr_blstm_fwd with the flag
// words and the h exchange of the persistent kernel in a file-scope device array instead of a
// caller-provided workspace.  Sized for the persistent kernel's own limits (B <= 32, H <= 512,
// H % 4 == 0): BLSTM_HX_OFF + 4 * (H/4) * ceil(B/16) * 64 + 64 floats = 264.25 KiB per GPU, held
// until the process ends.  In host code `blstm_scratch` is only a handle: its device address is
// taken once per device with hipGetSymbolAddress and every use below goes through that pointer.
// Calls on one GPU share the array (it keeps the last call's contents; what must start clean --
// the flag words and the epoch tags of the exchange -- is reset by asr_blstm_fwd_range at s0 = 0),
// so calls that overlap on one GPU (two streams) race.
#define BLSTM_STATIC_MAX_B 32
#define BLSTM_STATIC_MAX_H 512
#define BLSTM_STATIC_FLOATS (BLSTM_HX_OFF + 4 * (BLSTM_STATIC_MAX_H / 4) * 2 * 64 + 64)
#define BLSTM_STATIC_MAX_DEVICES 64
static __device__ __align__(256) float blstm_scratch[BLSTM_STATIC_FLOATS];
static float *g_scratch_dev[BLSTM_STATIC_MAX_DEVICES];   // device address of blstm_scratch, per device

extern "C" int64_t asr_blstm_fwd_noalloc_max_floats(void) { return BLSTM_STATIC_FLOATS; }

// Device address of blstm_scratch on the device `stream` belongs to (the current device for the null
// stream); taken once per device.  A stream of another device than the current one is rejected: the
// symbol lookup and the launch below both act on the current device.
static int blstm_noalloc_scratch(void *stream, float **out) {
    int dev = -1;
    ASR_CHECK_HIP(hipGetDevice(&dev));
    if (stream != nullptr) {
        hipDevice_t sdev = -1;
        ASR_CHECK_HIP(hipStreamGetDevice(static_cast<hipStream_t>(stream), &sdev));
        ASR_CHECK_ARG((int)sdev == dev, "blstm_fwd_noalloc: the stream belongs to device %d, the current device is %d",
                      (int)sdev, dev);
    }
    ASR_CHECK_ARG(dev >= 0 && dev < BLSTM_STATIC_MAX_DEVICES, "blstm_fwd_noalloc: device %d out of range", dev);
    float *ws = __atomic_load_n(&g_scratch_dev[dev], __ATOMIC_ACQUIRE);
    if (ws == nullptr) {   // first call on this device (idempotent: racing threads store the same address)
        void *p = nullptr;
        ASR_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(blstm_scratch)));
        ASR_CHECK_ARG(p != nullptr, "blstm_fwd_noalloc: no device address for the scratch array");
        ws = static_cast<float *>(p);
        __atomic_store_n(&g_scratch_dev[dev], ws, __ATOMIC_RELEASE);
    }
    *out = ws;
    return ASR_OK;
}

// The sticky error word of asr_blstm_fwd_noalloc on the current device (a device address inside the
// static scratch; NULL on failure, see asr_last_error): non-zero after a launch whose in-kernel wait
// timed out, in which case Hout / Cs of that call are garbage although the call returned ASR_OK (the
// launch is asynchronous).  Hand it to asr_poison_if_flag on the same stream, as the workspace entry
// points' callers do with word BLSTM_ERR_WORD of their workspace, or copy the 4 bytes back.  Sticky:
// the library never clears it (the set-up of a call resets words 0 .. BLSTM_FLAG_WORDS-1 only).
extern "C" const void *asr_blstm_fwd_noalloc_err_word(void) {
    float *ws = nullptr;
    if (blstm_noalloc_scratch(nullptr, &ws) != ASR_OK) return nullptr;
    return reinterpret_cast<unsigned *>(ws) + BLSTM_ERR_WORD;
}

extern "C" int asr_blstm_fwd_noalloc(const float *W_hh, float *G, float *Cs, float *Hout, int64_t ho_t,
                                     int64_t ho_b, const int32_t *lengths, int T, int B, int H, void *stream) {
    ASR_CHECK_ARG(T > 0 && B > 0 && H > 0, "blstm_fwd_noalloc: bad shape T=%d B=%d H=%d", T, B, H);
    ASR_CHECK_ARG(B <= BLSTM_STATIC_MAX_B && H <= BLSTM_STATIC_MAX_H && H % 4 == 0,
                  "blstm_fwd_noalloc: B=%d H=%d is outside the static scratch (B <= %d, H <= %d, H %% 4 == 0); "
                  "use asr_blstm_fwd with a workspace",
                  B, H, BLSTM_STATIC_MAX_B, BLSTM_STATIC_MAX_H);
    ASR_CHECK_ARG(asr_blstm_ws_floats(B, H) >= 0 &&
                      BLSTM_HX_OFF + (int64_t)4 * cdiv(H, 4) * cdiv(B, 16) * 64 + 64 <= BLSTM_STATIC_FLOATS,
                  "blstm_fwd_noalloc: scratch too small for B=%d H=%d", B, H);
    float *ws = nullptr;
    if (int rc = blstm_noalloc_scratch(stream, &ws)) return rc;
    // the persistent kernel only: its step-wise fallback would not need the scratch, but a caller of
    // this entry point asked for the one-launch path and must hear when it cannot have it
    const int kq4 = (cdiv(H, 16) + 3) / 4;
    const int64_t big = (int64_t)0x7fff0000;
    ASR_CHECK_ARG(kq4 <= 8 && al16p(W_hh) && (int64_t)T * B * 8 * H * 4 < big && ho_t >= 0 && ho_b >= 0 &&
                      ((int64_t)(T - 1) * ho_t + (int64_t)(B - 1) * ho_b + 2 * H) * 4 < big,
                  "blstm_fwd_noalloc: shape or alignment outside the persistent kernel's limits");
    return asr_blstm_fwd_range(W_hh, G, Cs, Hout, ho_t, ho_b, lengths, T, B, H, 0, T, ws, stream);
}
