set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/lvl192
GEO_ONLY=1 GEO_NO_TRAIN=1 PATHS_OVERLAP_AGGREGATOR=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/lvl192 -o p -- python3 $R/tools/geometry_time.py 192 4 128 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, statistics as st
f = glob.glob('/tmp/lvl192/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
kinds = {'attn_h3_any_kernel<48>': 'attn', 'tlayer_ws_kernel<192, true, false': 'chain', 'token0_dist_kernel<192>': 'tail', 'tlayer_ws_kernel<192, false, true, false, false': 'in_proj', 'importance_tokens_rows': 'rows', 'EpiBias': 'gemm'}
seq = {}
for s, e, n in rows:
    for k, v in kinds.items():
        if k in n: seq.setdefault(v, []).append((e - s) / 1e3)
for k, v in seq.items():
    n = len(v)
    for per in (5, 9, 4, 10):
        if n % per == 0: break
    lv = [[v[i] for i in range(j, n, per)][3:] for j in range(per)]
    print(f"{k:8s} launches {n} = {per} per step: " + "  ".join(f"{st.median(x):6.1f}" for x in lv if x))
PY
