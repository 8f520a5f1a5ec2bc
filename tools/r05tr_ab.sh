for rep in 1 2; do for v in 0 1; do
PATHS_ZERO_GRAD_ALL=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline --stress-steps 0 --k1024-steps 0 --td192-steps 0 --sustain 0 --breakdown-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PATHS_ZERO_GRAD_ALL=$v', 'train', d['train']['ms_per_step'])"
done; done
