set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dp in 0 0.05; do
rm -rf /tmp/trp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -o p -- python3 $R/bench.py --mode train --dropout $dp --steps 5 --warmup 2 > $R/gpurun_out/r05trd_$dp.json 2>/dev/null
f=$(find /tmp/trp -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r05trd_${dp}_kernel_stats.csv
tail -c 400 $R/gpurun_out/r05trd_$dp.json; echo
done
