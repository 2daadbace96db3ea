#!/bin/bash
# rocprofv3 kernel statistics of one bench workload (run on a GPU box): bash tools/prof_one.sh c4 [steps warmup]
set -u
WL=${1:-c4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_one
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$WL" -o $WL \
    -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs --no-sequence --sustain 0 --one-stream --workload $WL --steps ${2:-100} --warmup ${3:-50} > "$OUT/$WL.log" 2>&1 || { tail -5 "$OUT/$WL.log"; exit 1; }
f=$(find "$OUT/$WL" -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4,6,7 "$f" | cut -c1-160
