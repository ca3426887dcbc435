"""CPU: `python bench.py --gpus 2` without a launcher starts two ranks itself (the reference's
mp.spawn, train.py:180-186) and rank 0 prints ONE line with n_gpus = 2.  The HIP step is stubbed
(ASR_BENCH_STUB=1: the real bucketed reducer over gloo on hand-set gradients), everything else --
argument handling, the child torch.distributed.run, the rendezvous on 127.0.0.1, the JSON line -- is
the code the GPU job runs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("MASTER_PORT", None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus2_spawns_two_ranks_stubbed_step():
    r = _run({"ASR_BENCH_STUB": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["gpus_arg"] == 2
    assert len(res["per_rank_ms"]) == 2 and res["steps"] == 3 and res["warmup"] == 1
    assert res["mean_ok"] is True                       # DDP's mean over the two ranks' gradients
    assert res["allreduce_bytes_per_step"] > 0


def test_gpus8_spawns_eight_ranks_stubbed_step():
    """The launch the driver's 8-GPU node sees (`--gpus 8`), rehearsed with eight gloo ranks on the CPU: one line from
    rank 0 with the fields the scaling run is read from (rccl_ranks, per_rank_ms, allreduce_bus_GBs) and
    `scaling_efficiency_vs` against the N = 1 step time handed over in ASR_BENCH_N1_MS."""
    r = _run({"ASR_BENCH_STUB": "1", "ASR_BENCH_N1_MS": "5.0", "OMP_NUM_THREADS": "1"},
             "--gpus", "8", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["rccl_ranks"] == 8 and res["gpus_arg"] == 8
    assert len(res["per_rank_ms"]) == 8 and all(v > 0 for v in res["per_rank_ms"])
    assert res["allreduce_bus_GBs"] > 0 and len(res["allreduce_buckets"]) == 3
    assert res["mean_ok"] is True                       # mean of 1..8 = 4.5 on every parameter
    eff = res["scaling_efficiency_vs"]
    assert eff["n1_ms_per_step"] == 5.0 and eff["source"] == "ASR_BENCH_N1_MS"
    assert abs(eff["efficiency"] - 5.0 / res["ms_per_step"]) < 1e-9


def test_gpus2_without_gpus_refuses_loudly():
    """No stub, no rehearsal, fewer than N GPUs: the parent must say so and fail, not run one rank."""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode == 2
    assert "needs 2 GPUs" in r.stderr and not r.stdout.strip()
