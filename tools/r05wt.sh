set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "" $R/tools/_bin/v/wt0.so; do
  export PATHS_HIP_LIB=$lib
  echo "=========== PATHS_HIP_LIB=$lib"
  rm -rf /tmp/gaps_ser
  PATHS_OVERLAP_AGGREGATOR=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_ser -o p -- python3 $R/bench.py --eager --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 --sustain 0 --breakdown-steps 0 --rotate 1 > /dev/null 2>&1
  python3 $R/tools/agg_gaps.py /tmp/gaps_ser
  python3 $R/tools/seq_gaps.py /tmp/gaps_ser 100
done
for rep in 1 2; do for lib in "" $R/tools/_bin/v/wt0.so; do
  PATHS_HIP_LIB=$lib python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 --sustain 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['roofline']['attn_ffn']; print('lib=$lib', d['value'], 'sustained', d['sustained']['slides_per_s'], 'span', a['serialized_span_us'], a['serialized_span_replayed_us'], a['serialized_frac_replayed'])"
done; done
