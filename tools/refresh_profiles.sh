#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: regenerates the raw profile data under gpurun_out/<tag>_*.
# usage: tools/refresh_profiles.sh TAG        then copy / summarise into profiles/ (tools/pmc_profile_summary.py)
set -e
TAG=${1:-r03a}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_1gpu.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 > $OUT/${TAG}_bench_1gpu_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-steps 0 --rotate 1 --sustain 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-steps 0 --rotate 1 --sustain 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-steps 0 --rotate 1 --sustain 0 > /dev/null 2>&1
echo done; ls $OUT | grep $TAG
