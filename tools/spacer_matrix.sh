set -o pipefail
O=gpurun_out/r2g; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5"
for k in 1 2 3 4 5 7; do
ASR_SIDE_SPACERS=$k timeout -k 10 200 $B > $O/bench_sp$k.json 2> $O/bench_sp$k.err; rc=$?; echo "bench spacers=$k rc=$rc"; [ $rc -eq 0 ] || exit 0
done
ASR_SIDE_STREAM_KIND=normal timeout -k 10 200 $B > $O/bench_normal.json 2> $O/bench_normal.err; rc=$?; echo "bench normal rc=$rc"; [ $rc -eq 0 ] || exit 0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2g/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        r=d.get('roofline_all',{})
        print(f.split('/')[-1], 'ms/step', round(d['ms_per_step'],3), 'full', round(d['full_step_ms_with_clip_adam'],3),
              'bwd in-step us', round(r.get('blstm_bwd',{}).get('avg_us',0),1), 'alone', round(r.get('blstm_bwd',{}).get('alone_us',0),1))
    except Exception as e: print(f, 'ERR', e)
PY
