"""Summarise the counter passes of tools/pmc_r3.sh: per pass directory, the named kernels' counter values per launch
(mean over the launches after the first) and their durations from the same run's rows."""
import collections, csv, glob, os, sys
root = sys.argv[1]
KEYS = ("sgemm_bf3_kernel", "blstm_fwd_persist_kernel", "blstm_bwd_rs_kernel")
for d in sorted(glob.glob(os.path.join(root, "*", ""))):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = next((k for k in KEYS if k in r["Kernel_Name"]), None)
        if k:
            acc[(k, r["Counter_Name"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Dispatch_Id"]))
    for (k, c), v in sorted(acc.items()):
        v = v[1:] if len(v) > 1 else v
        print(f"{os.path.basename(d.rstrip('/')):34s} {k:26s} {c:30s} {sum(x[0] for x in v) / len(v):16.1f}   {sum(x[1] for x in v) / len(v) / 1e3:9.1f} us  (n={len(v)})")
