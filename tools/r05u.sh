set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PATHS_OVERLAP_AGGREGATOR=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_ser -o p -- python3 $R/bench.py --eager --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 --sustain 0 --breakdown-steps 0 --rotate 1 > /dev/null 2>&1
echo "== one stream (PATHS_OVERLAP_AGGREGATOR=0, --eager): kernel durations and gaps as rocprofv3 sees them"; python3 $R/tools/agg_gaps.py /tmp/gaps_ser
rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_live -o p -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 --sustain 0 --breakdown-steps 0 --rotate 1 > /dev/null 2>&1
echo "== live (two streams)"; python3 $R/tools/agg_gaps.py /tmp/gaps_live
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 --sustain 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d[\"roofline\"][\"attn_ffn\"]))"
