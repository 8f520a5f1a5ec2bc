# usage: r05var.sh name1 name2 ...   ("base" = the product library): one-stream kernel durations + bench spans, A/B/A/B
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
lib_of() { if [ "$1" = "base" ]; then echo ""; else echo $R/tools/_bin/v/$1.so; fi; }
for v in "$@"; do
  export PATHS_HIP_LIB=$(lib_of $v)
  echo "=========== $v"
  rm -rf /tmp/gaps_ser
  PATHS_OVERLAP_AGGREGATOR=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_ser -o p -- python3 $R/bench.py --eager --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 --sustain 0 --breakdown-steps 0 --rotate 1 > /dev/null 2>&1
  python3 $R/tools/agg_gaps.py /tmp/gaps_ser | grep -v "^gap"
done
for rep in 1 2; do for v in "$@"; do
  PATHS_HIP_LIB=$(lib_of $v) python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 --sustain 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['roofline']['attn_ffn']; print('$v', d['value'], 'sustained', d['sustained']['slides_per_s'], 'span', a['serialized_span_us'], a['serialized_span_replayed_us'], a['serialized_frac_replayed'])"
done; done
