set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1 2; do
  PATHS_FUSE_QKV=$m rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05c_stats_f$m -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 --sustain 0 --breakdown-steps 0 > $R/gpurun_out/r05c_f$m.json 2>/dev/null
  python3 $R/tools/kstats.py $R/gpurun_out/r05c_stats_f$m > $R/gpurun_out/r05c_kstats_f$m.txt
  head -28 $R/gpurun_out/r05c_kstats_f$m.txt
  rm -rf $R/gpurun_out/r05c_stats_f$m
done
cd $R && python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "standalone or fp8_stress or fused_importance" > gpurun_out/r05c_tests.log 2>&1; tail -5 gpurun_out/r05c_tests.log
