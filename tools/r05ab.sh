# usage: r05ab.sh ENVVAR val0 val1 ...   (default bench A/B/A/B on one box)
V=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $V=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 --breakdown-steps 0 --sustain 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$v rep=$rep', d['value'], 'sustained', d['sustained']['slides_per_s'], 'two lanes', d['host']['launch_modes']['replay_two_lanes_slides_per_s'])"
done; done
