set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/lvl_ser
PATHS_OVERLAP_AGGREGATOR=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/lvl_ser -o p -- python3 $R/bench.py --eager --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 --sustain 0 --breakdown-steps 0 --rotate 1 > /dev/null 2>&1
python3 $R/tools/agg_by_level.py /tmp/lvl_ser
