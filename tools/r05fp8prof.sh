set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp8prof -o p -- python3 $R/bench.py --mode stress --trans-dim 1536 --trans-heads 24 --fp8 --steps 4 --warmup 2 > $R/gpurun_out/r05fp8_bench.log 2>&1
f=$(find /tmp/fp8prof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r05fp8_kernel_stats.csv
tail -3 $R/gpurun_out/r05fp8_bench.log
