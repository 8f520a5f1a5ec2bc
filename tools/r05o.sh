set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GEO_NO_TRAIN=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05o_stats -o p -- python3 $R/tools/geometry_time.py 192 4 128 > $R/gpurun_out/r05o_geo.txt 2>/dev/null
cat $R/gpurun_out/r05o_geo.txt
python3 $R/tools/kstats.py $R/gpurun_out/r05o_stats > $R/gpurun_out/r05o_kstats.txt
rm -rf $R/gpurun_out/r05o_stats
head -40 $R/gpurun_out/r05o_kstats.txt | cut -c1-200
