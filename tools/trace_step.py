"""Per-queue timeline summary of one training step from a rocprofv3 kernel trace (csv).
usage: python tools/trace_step.py <kernel_trace.csv>   (trace of: bench.py --steps 3 --warmup 1 --no-cpu-baseline)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
names = [r["Kernel_Name"] for r in rows]
# steps are delimited by the loss kernel (one per step)
marks = [i for i, n in enumerate(names) if "ls_ce_kernel" in n]
print("loss kernels:", len(marks), "queues:", sorted({r["Queue_Id"] for r in rows}))
if len(marks) < 4:
    sys.exit(0)
a, b = marks[2], marks[3]            # one steady-state step: loss kernel to loss kernel
seg = rows[a:b]
t0, t1 = seg[0]["s"], rows[b]["s"]
print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(seg)} kernels")
byq = collections.defaultdict(list)
for r in seg:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    gaps = sum(max(0, rs[i + 1]["s"] - rs[i]["e"]) for i in range(len(rs) - 1))
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e6:.3f} ms, gaps between consecutive {gaps / 1e6:.3f} ms")
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][0] += 1; agg[k][1] += r["e"] - r["s"]
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"    {k:60s} {c:5d} {t / 1e6:8.3f} ms")
    # largest gaps
    gl = sorted(((rs[i + 1]["s"] - rs[i]["e"], rs[i]["Kernel_Name"][:40], rs[i + 1]["Kernel_Name"][:40]) for i in range(len(rs) - 1)), reverse=True)[:8]
    for g, x, y in gl:
        print(f"    gap {g / 1e3:8.1f} us after {x} before {y}")
