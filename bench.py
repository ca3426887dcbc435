"""Benchmark of the AttModel training hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

metric  : train frames/sec (fwd+bwd) -- sum of mel (10 ms fbank) frames in the
          batch / wall seconds of {forward + label-smoothed CE + backward
          (+ gradient all-reduce complete, N > 1)}; whole-job aggregate over ranks.
workload: BASELINE config 2: B=32 per GPU, 80-dim fbank,
          T_mel = linspace(1500, 900, 32), frame stacking 3 (T=500, D=240),
          5-layer BLSTM H=320, V=1000, L=100, fp32, synthetic N(0,1) features,
          random-init weights (seed 1234), inputs resident in HBM.
timing  : HIP events on the launch stream at every step boundary inside the timed region (one event per step);
          `ms_per_step` = the MEDIAN step (max over ranks for N > 1), `value` = frames / that;
          `ms_per_step_mean` = wall clock of the K steps / K (barrier + synchronize on both sides) beside it.
dtype   : fp32 tensors and fp32 accumulation everywhere; the matrix products run as the bf16x3 split on the bf16
          matrix pipe (three bf16 products per fp32 product, ~16 mantissa bits; include/asr_amd.h).  The same K steps
          with exact fp32 products (asr_gemm_precision(0) + asr_blstm_precision(0)) are `ms_per_step_exact_f32`.
Beside it, not part of `value`: `config5_conformer` = the same step for BASELINE config 5 (Conformer encoder),
`config4_csj` = BASELINE config 4 (T = 800, L = 150, V = 7000) with the decoder path it took, `config1_speech_commands`
(GPU step and the CPU leg on config 1, >= 5 steps), `ms_per_step_with_loss_item`.
A "step" is one such pass over one batch.  For N > 1 there is one rank per GPU, each rank its
own batch (weak scaling), gradients all-reduced in buckets on RCCL (asr_amd/parallel.py).  Under a
launcher (WORLD_SIZE set: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`)
this process IS a rank; without one, `python bench.py --gpus N` starts the N ranks itself as a child
torch.distributed.run (the reference's mp.spawn, train.py:180-186; rendezvous port from MASTER_PORT or
a free one, not the reference's '600010') before anything touches the GPU.  `--rehearsal` runs the
same N-rank code path with all ranks sharing GPU 0 over gloo (a 1-GPU box): the line it prints is
marked and is not a throughput claim.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = vector peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (the headline figure with 2:1 sparsity is not used)
XGMI_LINK_GBS = 153.0          # one xGMI link (7 per GPU)
# HBM-side bytes per launch from the rocprofv3 PMC passes kept under profiles/ (FETCH_SIZE x 2: the gfx950
# correction for 16-byte streams of MI355X_MICROARCH.md, + WRITE_SIZE), same kernels and shapes as here
PMC_TRAFFIC = {                                      # profiles/r3_pmc_mfma.txt (recurrences, GEMM), r2_pmc_fetch_write.txt (rest)
    # the recurrences' figures include the prefetching workgroups, which read every saved line once more on purpose
    # (round 2, without them: 2 * 83329e3 + 240165e3 and 2 * 100787e3 + 165567e3 = the algorithmic bytes)
    "blstm_fwd": 2 * 147587e3 + 240388e3,
    "blstm_bwd": 2 * 197002e3 + 166125e3,
    "sgemm_proj": 2 * 61809e3 + 80000e3,
    "attdec_fwd": 2 * 180997e3 + 166028e3,
    "attdec_bwd": 2 * 394285e3 + 249210e3,
    "ls_ce_config4": 2 * 66465e3 + 131470e3,
}
# Matrix-pipe utilisation from the SQ counter passes of the same kernels and shapes (profiles/r3_pmc_mfma.txt):
# SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), and against 1024 SIMDs x duration x 2.4 GHz
PMC_MFMA_BUSY = {
    "sgemm_proj": {"mfma_busy": 0.378, "mfma_busy_wall": 0.281, "clock_GHz_under_load": 1.78},
    "blstm_fwd": {"mfma_busy": 0.044, "mfma_busy_wall": 0.040},
    "blstm_bwd": {"mfma_busy": 0.063, "mfma_busy_wall": 0.037},
}


class KernelTimer:
    """HIP events on the launch stream round every call of the named C-ABI entry points while it is
    installed (the calls enqueue one kernel plus its few-KB set-up memsets): the in-step duration of
    the recurrence launches, side-stream GEMMs sharing the CUs included."""

    def __init__(self, names):
        from asr_amd import _lib
        self.lib = _lib.load()
        self.names = names
        self.orig = {n: getattr(self.lib, n) for n in names}
        self.pairs = {n: [] for n in names}
        for n in names:
            setattr(self.lib, n, self._wrap(n))

    def _wrap(self, n):
        orig, pairs = self.orig[n], self.pairs[n]

        def call(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = orig(*a)
            e1.record()
            pairs.append((e0, e1))
            return rc
        return call

    def finish(self):
        """Uninstall; average seconds per call (the caller has synchronised)."""
        for n in self.names:
            setattr(self.lib, n, self.orig[n])
        return {n: (sum(a.elapsed_time(b) for a, b in ps) * 1e-3 / len(ps) if ps else None)
                for n, ps in self.pairs.items()}


def make_hp(**kw):
    """hparams_template defaults for the attributes the hot path reads."""
    import types
    d = dict(lmfb_dim=80, frame_stacking=3, num_hidden_nodes=320, num_encoder_layer=5, encoder_dropout=0.2,
             encoder_type=None, decoder_type='Attention', num_classes=1000, eos_id=1, T_norm=True, B_norm=False,
             beam_width=1, max_decoder_seq_len=200, clip=5.0, lr=1e-3, batch_size=32)
    d.update(kw)
    d.setdefault('num_hidden_nodes_encoder', d['num_hidden_nodes'])
    d.setdefault('num_hidden_nodes_decoder', d['num_hidden_nodes'])
    return types.SimpleNamespace(**d)


def make_batch(hp, dev, seed=1234, B=32, t_hi=1500, t_lo=900, L=100, l_lo=50):
    """Synthetic batch of BASELINE config 2, already frame-stacked (a view in the
    reference, utils/utils.py:47-65).  Returns x (B,T,D) on dev, lengths (B,) CPU
    int64 (stacked), targets (B,L) dev, text_lengths (B,) dev."""
    g = torch.Generator().manual_seed(seed)
    mel_len = torch.linspace(t_hi, t_lo, B).long()
    text_len = torch.linspace(L, l_lo, B).long()
    T_mel = int(mel_len.max())
    st = hp.frame_stacking
    mel = torch.randn(B, T_mel, hp.lmfb_dim, generator=g)
    tgt = torch.randint(0, hp.num_classes, (B, L), generator=g)
    for b in range(B):
        mel[b, mel_len[b]:] = 0
        tgt[b, text_len[b]:] = 0
    n = T_mel // st
    x = mel[:, : n * st].reshape(B, n, -1).contiguous()
    lengths = mel_len // st
    make_batch.mel_frames = int(mel_len.sum())
    return x.to(dev), lengths, tgt.to(dev), text_len.to(dev)


def cpu_baseline(hp):
    """The CPU restatement of the reference (oracle/ref_model.py: stock PyTorch ops
    in the reference's op order -- nn.LSTM over a packed batch, per-step Linear /
    Conv1d attention, dense one-hot loss) timed on this host: ONE fwd + loss + bwd
    step of the same config-2 batch (B=32), i.e. the same workload, 1 step."""
    from oracle import ref_model as R
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
    host_cores = os.cpu_count()
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    m = R.AttModel(hp).train()
    cpu = torch.device("cpu")
    x, lengths, tgt, tl = make_batch(hp, cpu)
    frames = make_batch.mel_frames
    t0 = time.time()
    y = m(x, lengths, tgt)
    loss = R.label_smoothing_loss(y, tgt, tl, hp.T_norm, hp.B_norm)
    loss.backward()
    dt = time.time() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": threads, "host_cores": host_cores, "kind": "port",
            "sample": f"1 step fwd+loss+bwd of the full config-2 batch (B=32, {frames} mel frames, L=100) "
                      f"in {dt:.1f} s with torch CPU ops on {threads} threads"}


def time_kernel_events(fn, iters):
    """Average duration of one enqueue of fn() from HIP events on the current stream."""
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def roofline_probe(hp, dev, in_step=None):
    """Prices the step's main kernels against the gfx950 peaks of MI355X_MICROARCH.md.  The two
    recurrence kernels carry their IN-STEP duration (KernelTimer over the timed region: `avg_us`,
    what `achieved` is computed from) beside the duration of the same launch timed alone
    (`alone_us`); the others are timed alone with HIP events on the launch stream, same shapes as
    the timed step.  Algorithmic work counts the frames inside their sequences only (config 2:
    12 795 of 16 000 stacked frames), not the padded ones the kernels mask.  The dominant kernel of
    the step by total duration is blstm_bwd_rs_kernel (profiles/: rocprofv3 kernel stats)."""
    in_step = in_step or {}
    import ctypes as C
    from asr_amd import _lib, ops
    lib = _lib.load()
    H = hp.num_hidden_nodes_encoder
    B, T = 32, 500
    out = {}
    g = torch.Generator(device="cpu").manual_seed(1)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # (1) BLSTM recurrence, one layer = one launch: T steps x 2 dirs x (B x H) @ (H x 4H)
    W = (torch.randn(2, 4 * H, H, generator=g) * 0.05).to(dev)
    G0 = torch.randn(T, B, 2, 4 * H, generator=g).to(dev)
    Gw = G0.clone()
    Cs = torch.empty(T, B, 2, H, device=dev)
    Hout = torch.empty(T, B, 2 * H, device=dev)
    lengths = torch.linspace(500, 300, B).to(torch.int32).to(dev)
    ws = torch.zeros(int(lib.asr_blstm_ws_floats(B, H)), device=dev)

    def lstm_fwd():
        lib.asr_blstm_fwd(W.data_ptr(), Gw.data_ptr(), Cs.data_ptr(), Hout.data_ptr(), B * 2 * H, 2 * H,
                          lengths.data_ptr(), T, B, H, ws.data_ptr(), st)
    alone = time_kernel_events(lstm_fwd, 5)
    dt = in_step.get("asr_blstm_fwd") or alone
    # algorithmic work: the frames inside their sequences (the kernel masks the padded rows, it does not skip them)
    valid = float(lengths.sum())
    flops = valid * 2 * (2.0 * H * 4 * H)
    out["blstm_fwd"] = {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS,
                        "traffic": PMC_TRAFFIC.get("blstm_fwd"),
                        "kernel": "blstm_fwd_persist_kernel", "shape": [T, B, H], "valid_frames": valid,
                        "avg_us": dt * 1e6, "alone_us": alone * 1e6, "timed": "in the step" if in_step else "alone",
                        "us_per_time_step": dt * 1e6 / T,
                        "note": "latency-bound recurrence: 500 dependent steps per launch"}
    # (1b) its backward: dG[t] = f(dH[t] + dG[t+1] W_hh, saved gates), same T x 2 x (B x 4H) @ (4H x H) chain
    WT = W.transpose(1, 2).contiguous()
    dH = torch.randn(T, B, 2 * H, generator=g).to(dev)
    dc = torch.empty(2 * B * H, device=dev)

    def lstm_bwd():
        lib.asr_blstm_bwd(W.data_ptr(), WT.data_ptr(), Gw.data_ptr(), Cs.data_ptr(), dH.data_ptr(), B * 2 * H, 2 * H,
                          lengths.data_ptr(), dc.data_ptr(), T, B, H, ws.data_ptr(), st)
    alone = time_kernel_events(lstm_bwd, 5)
    dt = in_step.get("asr_blstm_bwd") or alone
    out["blstm_bwd"] = {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS,
                        "traffic": PMC_TRAFFIC.get("blstm_bwd"),
                        "kernel": "blstm_bwd_rs_kernel", "shape": [T, B, H], "valid_frames": valid,
                        "avg_us": dt * 1e6, "alone_us": alone * 1e6, "timed": "in the step" if in_step else "alone",
                        "us_per_time_step": dt * 1e6 / T,
                        "note": "dominant kernel of the step by total duration; latency-bound recurrence (500 dependent "
                                "steps); in the step it runs ~13 % longer than alone: memory-system latency under the "
                                "side stream's weight-gradient GEMMs (profiles/r3_pmc_mfma.txt)"}
    # (2) input-projection GEMM of an upper encoder layer: (T*B x 2H) @ (2H x 4H), fp32 MFMA
    M, N, K = T * B, 4 * H, 2 * H
    A = torch.randn(M, K, generator=g).to(dev)
    Wp = torch.randn(N, K, generator=g).to(dev)
    Cm = torch.empty(M, N, device=dev)
    dt = time_kernel_events(lambda: ops.sgemm(A, Wp, Cm, M, N, K, a_stride=K, b_stride=K, c_stride=N), 20)
    flops = 2.0 * M * N * K
    # the default arithmetic of the BLSTM path's hoisted products is the bf16x3 split (three bf16 MFMA products per
    # fp32 product, fp32 accumulate: asr_gemm_precision): priced against the dense bf16 MFMA peak with the three
    # products counted; `fp32_equivalent_tflops` is 2 M N K / time
    out["sgemm_proj"] = {"bound": "mfma", "achieved": 3.0 * flops / dt / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": 3.0 * flops / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                         "traffic": PMC_TRAFFIC.get("sgemm_proj"),
                         "kernel": "sgemm_bf3_kernel<128,128,0,0,3> (bf16x3 split, buffer loads)", "shape": [M, N, K],
                         "avg_us": dt * 1e6, "fp32_equivalent_tflops": flops / dt / 1e12,
                         "fp32_equivalent_frac_of_fp32_mfma_peak": flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS}
    # (3) fused label-smoothed CE: logits read once + gradient written once
    Bq, L, V = 32, 100, hp.num_classes
    z = torch.randn(Bq, L, V, generator=g).to(dev)
    tg = torch.randint(0, V, (Bq, L), generator=g).to(dev)
    tl = torch.full((Bq,), L, dtype=torch.int64, device=dev)
    row = torch.empty(Bq * L, device=dev)
    ls = torch.empty((), device=dev)
    dz = torch.empty_like(z)

    def loss_k():
        lib.asr_ls_ce_fwd_bwd(z.data_ptr(), tg.data_ptr(), tl.data_ptr(), Bq, L, V, C.c_float(0.1), 1, 0,
                              row.data_ptr(), ls.data_ptr(), dz.data_ptr(), st)
    dt = time_kernel_events(loss_k, 50)
    nbytes = 2.0 * Bq * L * V * 4
    out["ls_ce"] = {"bound": "hbm", "achieved": nbytes / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": nbytes / dt / 1e9 / HBM_PEAK_GBS, "traffic": None,
                    "kernel": "ls_ce_kernel<1> + ls_sum_kernel", "shape": [Bq, L, V], "avg_us": dt * 1e6}
    # (3b) the same at the config-4 (CSJ, 7k BPE) shape, where the loss is a real HBM stream
    Bq, L, V = 32, 150, 7000
    z = torch.randn(Bq, L, V, generator=g).to(dev)
    tg = torch.randint(0, V, (Bq, L), generator=g).to(dev)
    tl = torch.full((Bq,), L, dtype=torch.int64, device=dev)
    row = torch.empty(Bq * L, device=dev)
    dz = torch.empty_like(z)
    dt = time_kernel_events(loss_k, 20)
    nbytes = 2.0 * Bq * L * V * 4
    out["ls_ce_config4"] = {"bound": "hbm", "achieved": nbytes / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": nbytes / dt / 1e9 / HBM_PEAK_GBS, "traffic": PMC_TRAFFIC.get("ls_ce_config4"),
                            "kernel": "ls_ce_kernel<8> + ls_sum_kernel", "shape": [Bq, L, V], "avg_us": dt * 1e6}
    del z, dz
    # (4) the attention decoder loop: one persistent launch per pass (csrc/attdec.hip).  PH and hbatch are read ONCE
    # per launch and stay in registers for all L steps, so the algorithmic HBM bytes are the saved activations; the
    # kernels are bound by the hand-off latency of the 4-5 dependent edges per label step and by transcendental
    # issue (tanh / sech^2 of B T A elements per step), not by HBM -- the fraction says so
    from asr_amd.Models.decoder import Decoder
    Bq, T, L, V, Hd, E = 32, 500, 100, hp.num_classes, hp.num_hidden_nodes_decoder, 2 * hp.num_hidden_nodes_encoder
    A = 2 * Hd
    torch.manual_seed(0)
    dec = Decoder(hp).to(dev).train()
    lens = torch.linspace(500, 300, Bq).long()
    hb = torch.randn(Bq, T, E, generator=g).to(dev)
    for b in range(Bq):
        hb[b, lens[b]:] = 0
    tgt = torch.randint(0, V, (Bq, L), generator=g).to(dev)
    wy = torch.randn(Bq, L, V, generator=g).to(dev)
    kt = KernelTimer(("asr_attdec_fwd", "asr_attdec_bwd"))
    for _ in range(4):
        dec.zero_grad(set_to_none=True)
        hd = hb.clone().requires_grad_(True)
        (dec(hd, lens, tgt) * wy).sum().backward()
    torch.cuda.synchronize()
    tdec = kt.finish()
    vf = float(lens.sum())
    fwd_bytes = 4.0 * (2 * vf * A + L * Bq * T + L * vf * 10 + L * Bq * (E + 2 * Hd + 4 * Hd + A))
    bwd_bytes = 4.0 * (2 * vf * A + L * Bq * T + L * vf * 10 + L * Bq * (A + 4 * Hd + Hd + Hd + E)
                       + L * Bq * (4 * Hd + A + E) + L * vf * 11)
    for key, name, nb in (("attdec_fwd", "asr_attdec_fwd", fwd_bytes), ("attdec_bwd", "asr_attdec_bwd", bwd_bytes)):
        dt = tdec[name]
        out[key] = {"bound": "hbm", "achieved": nb / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": nb / dt / 1e9 / HBM_PEAK_GBS, "traffic": PMC_TRAFFIC.get(key),
                    "kernel": key + "_persist_kernel (whole L-step loop" + (", hoisted dyp GEMMs and phase 2 included)"
                                                                            if key == "attdec_bwd" else ")"),
                    "shape": [Bq, T, L, A], "avg_us": dt * 1e6, "us_per_label_step": dt * 1e6 / L,
                    "note": "latency / transcendental-issue bound: PH and hbatch are register-resident, "
                            "HBM carries only the saved activations"}
    for k, v in PMC_MFMA_BUSY.items():
        if k in out:
            out[k].update(v)
    return out


def config5_probe(dev, steps=5, warmup=2):
    """BASELINE config 5 beside the headline (not part of `value`): AttModel(encoder_type='Conformer') --
    TransformerEncoder(80, 256, 16, 4, 0.1), Models/AttModel.py:17 -- B=32, T_mel linspace(1500, 900) -> 374
    subsampled frames, V=1000, L=100, dropout as the reference trains; forward + label-smoothed CE + backward."""
    import types
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    B, Tm, V, L = 32, 1500, 1000, 100
    hp = types.SimpleNamespace(lmfb_dim=80, frame_stacking=1, num_hidden_nodes=320, num_hidden_nodes_encoder=256,
                               num_hidden_nodes_decoder=320, num_encoder_layer=5, encoder_dropout=0.2,
                               encoder_type="Conformer", decoder_type="Attention", num_classes=V, eos_id=1,
                               T_norm=True, B_norm=False, beam_width=1, max_decoder_seq_len=200,
                               score_func="log_softmax")
    torch.manual_seed(1234)
    m = AttModel(hp).to(dev).train()
    g = torch.Generator().manual_seed(1234)
    lens = torch.linspace(1500, 900, B).long()
    x = torch.randn(B, Tm, 80, generator=g)
    src_pos = torch.zeros(B, Tm, dtype=torch.int64)
    for b in range(B):
        x[b, lens[b]:] = 0
        src_pos[b, :lens[b]] = torch.arange(1, int(lens[b]) + 1)
    tl = torch.linspace(100, 50, B).long()
    tgt = torch.randint(0, V, (B, L), generator=g)
    for b in range(B):
        tgt[b, tl[b]:] = 0
    x, src_pos, tgt, tl_d = x.to(dev), src_pos.to(dev), tgt.to(dev), tl.to(dev)

    def step():
        m.zero_grad(set_to_none=True)
        loss = label_smoothing_loss(m(x, lens, tgt, src_pos), tgt, tl_d, hp.T_norm, hp.B_norm)
        loss.backward()
        return loss

    # side stream on: since round 3 the Conformer blocks put their weight / bias gradients there too, beside the next
    # block's data-gradient chain (56.4 -> 53.9 ms; with only the decoder's gradients there it gained nothing)
    from asr_amd import streams
    was = streams.ENABLED
    streams.enable(True)
    try:
        per, _, loss = timed_steps(step, steps, warmup)
        ms = median(per)
    finally:
        streams.enable(was)
    n_par = sum(p.numel() for p in m.parameters())
    return {"workload": "BASELINE config 5: Conformer x16 d256 h4 + attention decoder 320, B=32, T_mel linspace(1500,900) "
                        "-> T'=374, V=1000, L=100, fp32, dropout 0.1",
            "ms_per_step": ms, "frames_per_s": float(lens.sum()) / (ms * 1e-3), "steps": steps, "warmup": warmup,
            "parameters": n_par, "loss": float(loss.detach())}


def timed_steps(step, steps, warmup, sync_loss=False, ahead=2):
    """warmup untimed calls of step(), then `steps` timed ones: per-step durations (ms) from HIP events recorded on
    the current stream at every step boundary, and the wall clock of the whole region.  sync_loss: float(loss)
    after every step, the reference's `print(loss.item())` (train.py:69-73)."""
    wevs = []
    for i in range(warmup):
        if ahead and i >= ahead:
            wevs[i - ahead].synchronize()
        step()
        e = torch.cuda.Event()
        e.record()
        wevs.append(e)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    loss = None
    for i in range(steps):
        if ahead and i >= ahead:
            evs[i + 1 - ahead].synchronize()      # at most `ahead` steps in flight (see main)
        loss = step()
        if sync_loss:
            float(loss)
        evs[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    return per, wall, loss


def median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def config1_probe(dev, steps=5):
    """BASELINE config 1 (speech_commands shape): lmfb 40, 98 mel frames, B=8, V=37, L=3,
    stack 3 -> T=32, H=320 x 5.  The CPU leg (the oracle = stock-PyTorch restatement, >= 5 steps after one warm-up,
    BASELINE.md section 3) beside the same step on the GPU."""
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from oracle import ref_model as R
    hp = make_hp(lmfb_dim=40, num_classes=37, encoder_dropout=0.0)
    B, Tm, L = 8, 98, 3
    g = torch.Generator().manual_seed(1234)
    mel = torch.randn(B, Tm, 40, generator=g)
    x = mel[:, : (Tm // 3) * 3].reshape(B, Tm // 3, -1).contiguous()
    lengths = torch.full((B,), Tm // 3, dtype=torch.int64)
    tgt = torch.randint(0, 37, (B, L), generator=g)
    tl = torch.full((B,), L, dtype=torch.int64)
    frames = B * Tm
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    ref = R.AttModel(hp).train()

    def cpu_step():
        ref.zero_grad(set_to_none=True)
        loss = R.label_smoothing_loss(ref(x, lengths, tgt), tgt, tl, hp.T_norm, hp.B_norm)
        loss.backward()
        return loss
    cpu_step()
    cpu_ms = []
    for _ in range(steps):
        t0 = time.perf_counter()
        lc = cpu_step()
        cpu_ms.append((time.perf_counter() - t0) * 1e3)
    m = AttModel(hp)
    m.load_state_dict(ref.state_dict())
    m = m.to(dev).train()
    xd, td, tld = x.to(dev), tgt.to(dev), tl.to(dev)

    def gpu_step():
        m.zero_grad(set_to_none=True)
        loss = label_smoothing_loss(m(xd, lengths, td), td, tld, hp.T_norm, hp.B_norm)
        loss.backward()
        return loss
    per, _, lg = timed_steps(gpu_step, 20, 5)
    return {"workload": "BASELINE config 1: B=8, 98 mel frames x 40, stack 3 -> T=32, 5xBLSTM 320, V=37, L=3, same weights",
            "mel_frames": frames,
            "cpu": {"kind": "port", "cores": threads, "steps": steps, "ms_per_step_median": median(cpu_ms),
                    "ms_per_step": cpu_ms, "frames_per_s": frames / (median(cpu_ms) * 1e-3), "loss": float(lc.detach())},
            "gpu": {"steps": 20, "ms_per_step_median": median(per), "frames_per_s": frames / (median(per) * 1e-3),
                    "loss": float(lg.detach())},
            "loss_rel_diff": abs(float(lg.detach()) - float(lc.detach())) / abs(float(lc.detach()))}


def config4_probe(dev, steps=6, warmup=3):
    """BASELINE config 4 (CSJ shape): lmfb 40 -> D_in 120, V=7000, B=32, T_mel linspace(2400,
    1200) -> T=800, L=150; forward + label-smoothed CE + backward, dropout 0.2.  Reports which decoder path ran:
    416 (b, 64-frame tile) workgroups do not fit 256 compute units, so the decoder loop runs as two persistent
    launches of 16 utterances each (asr_amd/Models/decoder.py::batch_parts); `per_step_decoder_ms_per_step` is the
    same step on the four-launches-per-label-step loop (asr_attdec_persist(0)), the path this shape took before."""
    from asr_amd import _lib, streams
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from asr_amd.Models.decoder import batch_parts
    lib = _lib.load()
    hp = make_hp(lmfb_dim=40, num_classes=7000)
    torch.manual_seed(1234)
    m = AttModel(hp).to(dev).train()
    x, lengths, tgt, tl = make_batch(hp, dev, seed=1234, B=32, t_hi=2400, t_lo=1200, L=150, l_lo=75)
    frames = int(torch.linspace(2400, 1200, 32).long().sum())
    T, E, Hd, L = x.shape[1], 2 * hp.num_hidden_nodes_encoder, hp.num_hidden_nodes_decoder, 150
    parts = batch_parts(lib, 32, T, E, Hd, L)
    plan = [int(lib.asr_attdec_plan(b1 - b0, T, E, Hd, L)) for b0, b1 in parts]

    def step():
        for p in m.parameters():
            p.grad = None
        loss = label_smoothing_loss(m(x, lengths, tgt), tgt, tl, hp.T_norm, hp.B_norm)
        loss.backward()
        return loss

    was = streams.ENABLED
    streams.enable(True)
    try:
        per, _, loss = timed_steps(step, steps, warmup)
        old = lib.asr_attdec_persist(0)
        try:
            per0, _, _ = timed_steps(step, max(3, steps // 2), 2)
        finally:
            lib.asr_attdec_persist(old)
    finally:
        streams.enable(was)
    ms = median(per)
    return {"workload": "BASELINE config 4 (CSJ): B=32, T_mel linspace(2400,1200) x 40, stack 3 -> T=800, D=120, "
                        "5xBLSTM 320, V=7000, L=150, dropout 0.2",
            "ms_per_step": ms, "frames_per_s": frames / (ms * 1e-3), "steps": steps, "warmup": warmup,
            "decoder_path": ("persistent loop kernels, %d launch(es) per pass over batch parts %s" % (len(parts), parts)
                             if all(v == 3 for v in plan) else "four launches per label step (fallback)"),
            "attdec_plan": plan,
            "per_step_decoder_ms_per_step": median(per0), "loss": float(loss.detach())}


def start_watchdog(rank, dev):
    """Rehearsal diagnostics: every 20 s one line per rank saying whether the device still retires work
    (stream queries never block), so a stalled run names the rank and the stream that sit still."""
    import threading
    main = torch.cuda.current_stream(dev)

    def loop():
        from asr_amd import streams
        n = 0
        while True:
            time.sleep(20)
            n += 1
            side = streams._streams.get(dev.index)
            print(f"[bench rank {rank}] watchdog {20 * n:4d} s: main stream idle={main.query()} "
                  f"side stream idle={side.query() if side is not None else None}", file=sys.stderr, flush=True)
    threading.Thread(target=loop, daemon=True).start()


N1_FILE = os.path.join(os.environ.get("TMPDIR", "/tmp"), "asr_amd_bench_n1.json")


def n1_store(ms):
    """An N = 1 run leaves its ms_per_step where an N > 1 run on the same host can find it."""
    try:
        with open(N1_FILE, "w") as f:
            json.dump({"ms_per_step": ms, "time": time.time(), "host": os.uname().nodename}, f)
    except OSError:
        pass


def n1_reference():
    """(ms_per_step at N = 1, where it came from) for the N > 1 line's `scaling_efficiency_vs`: environment variable
    ASR_BENCH_N1_MS, else what an N = 1 run of this bench left on this host within the last two hours."""
    v = os.environ.get("ASR_BENCH_N1_MS")
    if v:
        return float(v), "ASR_BENCH_N1_MS"
    try:
        with open(N1_FILE) as f:
            d = json.load(f)
        if d.get("host") == os.uname().nodename and time.time() - d["time"] < 7200:
            return float(d["ms_per_step"]), f"{N1_FILE} ({int(time.time() - d['time'])} s old)"
    except (OSError, ValueError, KeyError):
        pass
    return None


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as ONE child process tree
    (`python -m torch.distributed.run --nproc-per-node N bench.py ...`, the launch line of the task's
    contract) and return its exit code.  The parent never initialises the GPU -- `device_count()` does
    not -- so nothing that has touched the device is ever replaced or forked."""
    import subprocess
    n = args.gpus
    rehearsal = args.rehearsal or bool(os.environ.get("ASR_BENCH_REHEARSAL"))
    stub = bool(os.environ.get("ASR_BENCH_STUB"))
    have = torch.cuda.device_count()
    if not (rehearsal or stub) and have < n:
        print(f"bench.py: --gpus {n} needs {n} GPUs, this host shows {have}; `--rehearsal` runs the {n}-rank code "
              "path on one GPU over gloo", file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.run(cmd, env=env).returncode


def stub_rank(args):
    """ASR_BENCH_STUB=1 (CPU tests of the launcher, tests/test_bench_spawn.py): this rank joins a gloo
    group, runs the real bucketed reducer over hand-set gradients of a small CPU-resident AttModel in
    place of the HIP step, and rank 0 prints a line of the usual shape.  Never a measurement."""
    import torch.distributed as dist
    from asr_amd.Models.AttModel import AttModel
    from asr_amd.parallel import BucketedGradReducer
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1234 + rank)
    model = AttModel(make_hp(lmfb_dim=4, frame_stacking=1, num_hidden_nodes=8, num_encoder_layer=2, num_classes=7))
    with BucketedGradReducer(model) as reducer:
        reducer.broadcast_parameters()

        def step():
            for p in model.parameters():
                p.grad = torch.full_like(p, float(rank + 1))
            reducer.finish()

        for _ in range(args.warmup):
            step()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        dist.barrier()
        dt = time.perf_counter() - t0
        per_rank = [None] * world
        dist.all_gather_object(per_rank, dt / args.steps * 1e3)
        mean_ok = all(bool(torch.allclose(p.grad, torch.full_like(p, (world + 1) / 2.0))) for p in model.parameters())
        ar = reducer.allreduce_probe(iters=2)            # collective: every rank
        if rank == 0:
            res = {"metric": "stub", "value": 0.0, "unit": "frames/s", "n_gpus": world,
                   "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": max(per_rank), "per_rank_ms": per_rank, "data": "stub",
                   "allreduce_bytes_per_step": reducer.bytes_per_step(), "mean_ok": mean_ok, "gpus_arg": args.gpus,
                   "allreduce_buckets": ar}
            tot_b, tot_s = sum(a["bytes"] for a in ar), sum(a["us"] for a in ar) * 1e-6
            res["allreduce_bus_GBs"] = tot_b / tot_s / 1e9 * 2.0 * (world - 1) / world
            n1 = n1_reference()
            if n1 is not None:
                res["scaling_efficiency_vs"] = {"n1_ms_per_step": n1[0], "source": n1[1],
                                                "efficiency": n1[0] / max(per_rank)}
            print(json.dumps(res), flush=True)
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--max-ahead", type=int, default=2,
                    help="steps the host may run ahead of the device (0 = unbounded); see the comment at the warm-up loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true", help="skip the Conformer (config 5) side measurement")
    ap.add_argument("--no-config4", action="store_true", help="skip the CSJ-shape (config 4) side measurement")
    ap.add_argument("--dropout", type=float, default=0.2, help="encoder dropout (template default 0.2)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="N ranks share GPU 0 over gloo: the N-rank code path on a 1-GPU box (not a throughput claim)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks; "
              "n_gpus reports the ranks that ran", file=sys.stderr)
    if os.environ.get("ASR_BENCH_STUB"):
        raise SystemExit(stub_rank(args))
    if args.rehearsal or os.environ.get("ASR_BENCH_REHEARSAL"):
        os.environ["ASR_BENCH_ONE_DEVICE"] = "1"
        os.environ["ASR_BENCH_BACKEND"] = "gloo"
    hang_s = int(os.environ.get("ASR_BENCH_HANG_DUMP_S", "0"))
    if hang_s > 0:      # diagnostics: where does every thread of this rank sit if the run stalls?
        import faulthandler
        faulthandler.dump_traceback_later(hang_s, repeat=True, file=sys.stderr)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("ASR_BENCH_ONE_DEVICE"):      # rehearsal: every rank on GPU 0 (gloo only)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dp = bool(os.environ.get("ASR_BENCH_FORCE_DP"))   # rehearsal: the N-GPU code path in a group of one
    from asr_amd import streams as _streams
    _streams.prepare(dev)      # helper streams own their hardware queues before RCCL creates its streams
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("ASR_BENCH_BACKEND", "nccl")     # "gloo": rehearsal on a 1-GPU box
        # every stream at normal priority (TORCH_NCCL_HIGH_PRIORITY unset): a high-priority RCCL stream
        # that waits on a normal-priority stream's event starved it on this platform -- asr_amd/parallel.py
        # RCCL prints a version banner on stdout when its first communicator comes up; stdout is reserved for the
        # one JSON line, so file descriptor 1 points at stderr until the parameters have been broadcast
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from asr_amd.parallel import BucketedGradReducer

    verbose = bool(os.environ.get("ASR_BENCH_VERBOSE"))

    def log(msg):
        if verbose:
            print(f"[bench rank {rank}] {time.strftime('%H:%M:%S')} {msg}", file=sys.stderr, flush=True)

    hp = make_hp(encoder_dropout=args.dropout)
    torch.manual_seed(1234)                      # identical weights on every rank
    model = AttModel(hp).to(dev).train()
    log('model built')
    reducer = BucketedGradReducer(model, force=force_dp)
    if world > 1 and os.environ.get("ASR_BENCH_ONE_DEVICE"):
        start_watchdog(rank, dev)
    reducer.broadcast_parameters()
    if world > 1 or force_dp:
        torch.cuda.synchronize()
        log('broadcast synchronised')
        if world > 1 and not os.environ.get("ASR_BENCH_NO_BARRIER"):
            dist.barrier()      # no rank starts its steps while another still drains its parameter copies
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    log('parameters broadcast')
    x, lengths, tgt, tl = make_batch(hp, dev, seed=1234 + rank)
    frames = make_batch.mel_frames
    from asr_amd.optim import ClipAdam
    opt = ClipAdam(model.parameters(), lr=hp.lr, weight_decay=1e-5)          # fused clip + Adam
    opt_stock = torch.optim.Adam(model.parameters(), lr=hp.lr, weight_decay=1e-5)

    def step():
        for p in model.parameters():
            p.grad = None
        y = model(x, lengths, tgt)
        loss = label_smoothing_loss(y, tgt, tl, hp.T_norm, hp.B_norm)
        loss.backward()
        reducer.finish()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # The host enqueues a step in 3-5 ms and would run up to K steps ahead of the device; every step in flight holds
    # its own ~2 GB of activations in torch's caching allocator (a block is reusable only once the device has passed
    # its last use), so an unbounded run-ahead keeps growing the pool with hipMalloc calls INSIDE the timed region
    # (seen as 20-30 ms steps scattered over the first dozen).  As a training loop that looks at its loss does, the
    # host stays at most `ahead` steps in front of the device: it waits for the end-of-step event of step i - ahead
    # before it enqueues step i -- the device never idles (two steps are 34 ms of queued work).
    ahead = max(0, args.max_ahead)
    wevs = []
    for i in range(args.warmup):
        if ahead and i >= ahead:
            wevs[i - ahead].synchronize()
        step()
        e = torch.cuda.Event()
        e.record()
        wevs.append(e)
        log(f'warm-up step {i} enqueued')
    barrier()
    log('warm-up done')
    # the timed region: EXACTLY args.steps steps between two barrier + synchronize pairs; one HIP event per step
    # boundary on the launch stream (a step ends with the side stream joined and the reducer finished on it)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    host_t = []
    for i in range(args.steps):
        if ahead and i >= ahead:
            evs[i + 1 - ahead].synchronize()      # the end of step i - ahead
        loss = step()
        evs[i + 1].record()
        host_t.append(time.perf_counter())
    barrier()
    dt = time.perf_counter() - t0
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    med_ms = median(step_ms)
    # the recurrence kernels are timed IN the step, in a pass of their own behind the timed region: HIP events on the
    # launch stream round every asr_blstm_fwd / asr_blstm_bwd call (one launch per layer).  Inside the timed region the
    # 20 event records per step cost 0.3-0.6 ms of it (each one is a barrier packet between two dependent launches).
    in_step = {}
    if rank == 0 and world == 1:       # N > 1: every step is a collective, and the roofline probes are N = 1 only
        ktimer = KernelTimer(("asr_blstm_fwd", "asr_blstm_bwd"))
        for _ in range(max(3, args.steps // 4)):
            step()
        torch.cuda.synchronize()
        in_step = ktimer.finish()
    barrier()
    from asr_amd.Models import encoder as _enc
    if _enc.persistent_error_flags():
        raise SystemExit("bench.py: a persistent recurrence kernel reported a timed-out in-launch wait; "
                         "the run is invalid")
    per_rank_ms = [med_ms]
    if world > 1:
        import torch.distributed as dist
        per_rank_ms = [None] * world
        dist.all_gather_object(per_rank_ms, med_ms)
        tt = torch.tensor([dt, med_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, med_ms = float(tt[0]), float(tt[1])
        ft = torch.tensor([frames], device=dev, dtype=torch.float64)
        dist.all_reduce(ft, op=dist.ReduceOp.SUM)
        total_frames = float(ft)
    else:
        total_frames = float(frames)
    mean_ms = dt / args.steps * 1e3
    ms = med_ms

    # the same K steps with exact fp32 products everywhere (the reference's arithmetic, Models/encoder.py:25-26),
    # and with the reference's per-step `loss.item()` host sync (train.py:69-73)
    exact_ms = item_ms = None
    if world == 1:
        from asr_amd import _lib as _l
        lib_ = _l.load()
        og, ob = lib_.asr_gemm_precision(0), lib_.asr_blstm_precision(0)
        try:
            exact_ms = median(timed_steps(step, args.steps, 3)[0])
        finally:
            lib_.asr_gemm_precision(og)
            lib_.asr_blstm_precision(ob)
        item_ms = median(timed_steps(step, args.steps, 2, sync_loss=True)[0])

    # full optimiser step (clip + Adam, train.py:60-63), reported beside the metric: with the fused
    # asr_amd.optim.ClipAdam, and with the reference's stock clip_grad_norm_ + torch.optim.Adam
    def time_full(update):
        step()                                   # untimed: optimiser state + first-use kernel loads
        update()
        barrier()
        n = max(2, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(n):
            step()
            update()
        barrier()
        return (time.perf_counter() - t0) / n * 1e3

    def stock_update():
        torch.nn.utils.clip_grad_norm_(model.parameters(), hp.clip)
        opt_stock.step()

    full_ms = full_stock_ms = None
    if rank == 0 or world > 1:
        full_ms = time_full(lambda: opt.step(max_norm=hp.clip))
        full_stock_ms = time_full(stock_update)

    # the gradient exchange alone (N > 1): each bucket's all-reduce timed by itself, bus GB/s against one xGMI link
    ar = reducer.allreduce_probe() if (world > 1 or force_dp) else []

    if rank == 0:
        res = {
            "metric": "train frames/sec (fwd+bwd), AttModel 1k-BPE LibriSpeech shape",
            "value": total_frames / (ms * 1e-3),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 storage + accumulate, bf16x3 products (3 bf16 MFMA products per fp32 product)",
            "data": "synthetic",
            "ms_per_step_is": "median of the per-step HIP-event durations of the timed region (max over ranks)",
            "ms_per_step_mean": mean_ms, "value_from_mean": total_frames / (mean_ms * 1e-3),
            "ms_per_step_min_max": [min(step_ms), max(step_ms)],
            "ms_per_step_all": [round(v, 3) for v in step_ms],
            "max_steps_in_flight": ahead,
            "host_enqueue_ms_all": [round((b - a) * 1e3, 3) for a, b in zip([t0] + host_t[:-1], host_t)],
            "ms_per_step_exact_f32": exact_ms,
            "value_exact_f32": (total_frames / (exact_ms * 1e-3)) if exact_ms else None,
            "ms_per_step_with_loss_item": item_ms,
            "config": {"workload": "BASELINE config 2: B=32/GPU, T_mel linspace(1500,900), stack 3 -> T=500, "
                                   "D=240, 5xBLSTM H=320, V=1000, L=100",
                       "global_batch": 32 * world, "parallelism": f"dp{world}",
                       "encoder_dropout": args.dropout, "loss": float(loss.detach())},
            "stacked_frames_per_s": total_frames / 3.0 / (ms * 1e-3),
            "full_step_ms_with_clip_adam": full_ms,
            "full_step_ms_with_stock_clip_adam": full_stock_ms,
        }
        if world > 1 or force_dp:
            import torch.distributed as dist
            res["rccl_ranks"] = dist.get_world_size()
            res["backend"] = dist.get_backend()
            res["per_rank_ms"] = per_rank_ms
            res["allreduce_bytes_per_step"] = reducer.bytes_per_step()
            res["allreduce_buckets"] = ar
            if ar:
                tot_b, tot_s = sum(a["bytes"] for a in ar), sum(a["us"] for a in ar) * 1e-6
                res["allreduce_bus_GBs"] = tot_b / tot_s / 1e9 * 2.0 * (world - 1) / world
                res["xgmi_link_peak_GBs"] = XGMI_LINK_GBS
            n1 = n1_reference()
            if n1 is not None:       # weak scaling: N ranks each do the N = 1 work; efficiency = t(1) / t(N)
                res["scaling_efficiency_vs"] = {"n1_ms_per_step": n1[0], "source": n1[1],
                                                "efficiency": n1[0] / ms}
            if os.environ.get("ASR_BENCH_ONE_DEVICE"):
                res["rehearsal"] = (f"{world} ranks share GPU 0 over {dist.get_backend()}: the N-rank code path "
                                    "(hooks, side-stream packing, collective, mean), NOT a throughput claim")
        probes = roofline_probe(hp, dev, in_step) if world == 1 else {}
        if probes:
            res["roofline"] = dict(probes["blstm_bwd"])
            res["roofline_all"] = probes
        if world == 1 and not force_dp:
            n1_store(ms)
        if world == 1 and not args.no_config4:
            try:
                res["config4_csj"] = config4_probe(dev)
            except Exception as e:      # a side measurement must never cost the headline line
                res["config4_csj"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_config5:
            try:
                res["config5_conformer"] = config5_probe(dev)
            except Exception as e:      # a side measurement must never cost the headline line
                res["config5_conformer"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["config1_speech_commands"] = config1_probe(dev)
            except Exception as e:
                res["config1_speech_commands"] = {"error": repr(e)[:300]}
            res["cpu_baseline"] = cpu_baseline(make_hp(encoder_dropout=0.0))
        print(json.dumps(res), flush=True)
    reducer.close()
    if world > 1 or force_dp:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
