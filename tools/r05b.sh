set -o pipefail
python -m pytest tests -m gpu -q -x > gpurun_out/r05b_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r05b_tests.log
tail -3 gpurun_out/r05b_tests.log
for rep in 1 2; do for m in 0 1 2; do
  PATHS_FUSE_QKV=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --stress-steps 0 --k1024-steps 0 --td192-steps 0 > gpurun_out/r05b_bench_f${m}_${rep}.json 2> gpurun_out/r05b_bench_f${m}_${rep}.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05b_bench_f${m}_${rep}.json").read().strip().splitlines()[-1])
a=d["roofline"]["attn_ffn"]
print("fuse=${m} rep=${rep}", d["value"], d["sustained"]["slides_per_s"], "attn_ffn ser", a["serialized_span_us"], a["serialized_frac"], "replayed", a.get("serialized_span_replayed_us"), a.get("serialized_frac_replayed"), "two lanes", d["host"]["launch_modes"]["replay_two_lanes_slides_per_s"])
print({k: v for k, v in d["serialized_breakdown"]["us_per_launch"].items()})
PY
done; done
