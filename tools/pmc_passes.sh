set -e
R=$PWD
mkdir -p gpurun_out/r2x
cd /tmp; export TMPDIR=/tmp; cd $R
run() { # name counter script [env]
  timeout -s KILL 120 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/gpurun_out/r2x/$1_$2 -o p -- python3 $3 > gpurun_out/r2x/$1_$2.log 2>&1
}
run lstm FETCH_SIZE tools/lstm_one.py && run lstm WRITE_SIZE tools/lstm_one.py && \
DEC_L=100 run dec FETCH_SIZE tools/dec_one.py && DEC_L=100 run dec WRITE_SIZE tools/dec_one.py && \
run loss FETCH_SIZE tools/loss_one.py && run loss WRITE_SIZE tools/loss_one.py
tail -1 gpurun_out/r2x/loss_WRITE_SIZE.log
ls gpurun_out/r2x
