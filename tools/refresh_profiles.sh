#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: bench + kernel stats + three --pmc passes, summarised ON THE BOX (the raw
# counter CSVs are > 64 MiB: gpurun does not copy that much back); what comes back under gpurun_out/ is ready for profiles/.
# usage: tools/refresh_profiles.sh TAG
set -e
TAG=${1:-r05a}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
RAW=/tmp/paths_prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_1gpu.json 2> $OUT/${TAG}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 > $OUT/${TAG}_bench_1gpu_under_rocprof.json 2>/dev/null
cp $(find $RAW/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_bench_k2048_b8.csv
echo "kernel stats done"
PMC_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 --rotate 1 --sustain 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $RAW/pmc_sq -o p -- python3 $R/bench.py $PMC_ARGS > /dev/null 2>&1
echo "pmc sq done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/pmc_fetch -o p -- python3 $R/bench.py $PMC_ARGS > /dev/null 2>&1
echo "pmc fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/pmc_write -o p -- python3 $R/bench.py $PMC_ARGS > /dev/null 2>&1
echo "pmc write done"
python3 $R/tools/pmc_profile_summary.py $RAW/pmc_sq $RAW/pmc_fetch $RAW/pmc_write $OUT/${TAG}_pmc_traffic.json
rm -rf $RAW
ls -la $OUT | grep $TAG
