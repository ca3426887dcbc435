# Round-3 counter passes (one rocprofv3 run per counter or per small same-block group, --kernel-trace beside it):
#   the projection GEMM at the config-2 shape (tools/gemm_one.py), both recurrence kernels warm (tools/lstm_one.py)
#   and cold with / without the prefetching workgroups (tools/lstm_one_cold.py).
# usage (GPU box, repo root): bash tools/pmc_r3.sh ; python tools/pmc_summary.py gpurun_out/r3pmc > gpurun_out/r3pmc/summary.txt
R=$PWD
OUT=$R/gpurun_out/r3pmc
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() { # tag counters script
  timeout -s KILL 120 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -o p -- python3 $R/$3 > $OUT/$1.log 2>&1 || { echo "pass $1 failed"; tail -3 $OUT/$1.log; return 1; }
}
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES FETCH_SIZE WRITE_SIZE; do
  run gemm_$c $c tools/gemm_one.py || exit 1
  run lstm_$c $c tools/lstm_one.py || exit 1
done
run gemm_TCC "TCC_HIT_sum TCC_MISS_sum" tools/gemm_one.py || exit 1
run lstm_TCC "TCC_HIT_sum TCC_MISS_sum" tools/lstm_one.py || exit 1
for pf in 0 16; do
  export ASR_BLSTM_PREFETCH=$pf
  run cold${pf}_FETCH_SIZE FETCH_SIZE tools/lstm_one_cold.py || exit 1
  run cold${pf}_TCC "TCC_HIT_sum TCC_MISS_sum" tools/lstm_one_cold.py || exit 1
  run cold${pf}_EA "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" tools/lstm_one_cold.py || echo "(EA pass unavailable)"
done
unset ASR_BLSTM_PREFETCH
echo done
