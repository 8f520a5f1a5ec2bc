for v in 1 0 1 0; do PATHS_PARENT_SMALL_TILES=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 --stress-steps 0 --td192-steps 0 --sustain 0 --breakdown-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PARENT_SMALL=$v', 'headline', d['value'], 'k1024', d['k1024']['at_headline_batch']['slides_per_s'], d['k1024']['slides_per_s'])"; done
