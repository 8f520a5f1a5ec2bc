set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/geo_ser
GEO_ONLY=1 GEO_NO_TRAIN=1 PATHS_OVERLAP_AGGREGATOR=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/geo_ser -o p -- python3 $R/tools/geometry_time.py 192 4 128 > /dev/null 2>&1
python3 $R/tools/seq_gaps.py /tmp/geo_ser 100
