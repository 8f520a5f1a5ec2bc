set -o pipefail
python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "three_adamw or recursion_gradients_vs or selection_chain_backward" > gpurun_out/r05l_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05l_tests.log; tail -3 gpurun_out/r05l_tests.log
for rep in 1 2; do
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
  PATHS_H_TRAIN_SMALL=$1 PATHS_TRAIN_SPLITK_IMPORTANCE=$2 python bench.py --mode train --steps 12 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('h_small=$1 splitk=$2 rep=$rep', d['ms_per_step'], 'ms')"
done; done
