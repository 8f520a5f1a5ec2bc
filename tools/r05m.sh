set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05m_stats -o p -- python3 $R/bench.py --mode train --steps 4 --warmup 3 > $R/gpurun_out/r05m_train_under_rocprof.json 2>/dev/null
cp $(find $R/gpurun_out/r05m_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05m_kernel_stats_train_k2048_b8.csv
rm -rf $R/gpurun_out/r05m_stats
head -45 $R/gpurun_out/r05m_kernel_stats_train_k2048_b8.csv | cut -c1-200
