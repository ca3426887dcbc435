mkdir -p gpurun_out/r3a
for cfg in "normal all" "cumask all" "cumask hi:22" "cumask hi:20" "cumask hi:16"; do
  set -- $cfg
  ASR_SIDE_STREAM_KIND=$1 ASR_SIDE_CU_MASK=$2 timeout -k 10 200 python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-config5 > gpurun_out/r3a/b.json 2>gpurun_out/r3a/b.err || { echo "FAILED $cfg"; tail -3 gpurun_out/r3a/b.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3a/b.json')); r=d['roofline_all']
print('$cfg', 'ms/step %.2f' % d['ms_per_step'], 'bwd in-step %.0f us alone %.0f' % (r['blstm_bwd']['avg_us'], r['blstm_bwd']['alone_us']), 'fwd %.0f' % r['blstm_fwd']['avg_us'])"
done
