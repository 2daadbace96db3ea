#!/bin/bash
# The two HBM-traffic counter passes of tools/profile_round.sh for ONE workload (run on a GPU
# box): bash tools/pmc_traffic_one.sh c4 r03   -> gpurun_out/prof_<tag>/pmc_<workload>_*;
# tools/merge_traffic.py then replaces that workload's entry in profiles/<tag>_hbm_traffic.json.
set -u
WL=${1:-c4}
TAG=${2:-r00}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$OUT/pmc_${WL}_$ctr"
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_${WL}_$ctr" -o $WL \
        -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs --no-sequence --sustain 0 --one-stream \
        --workload $WL --steps 2 --warmup 1 > "$OUT/pmc_${WL}_$ctr.log" 2>&1 || { tail -3 "$OUT/pmc_${WL}_$ctr.log"; exit 1; }
    echo "$ctr done"
done
