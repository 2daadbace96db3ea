#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box:
#   1. --kernel-trace --stats of the default bench and of the c3 / c4 workloads
#   2. separate --pmc passes (FETCH_SIZE, then WRITE_SIZE; they do not fit one pass)
# Usage (from the repo root, through gpurun):  bash tools/profile_round.sh r01
# Results land in gpurun_out/prof_<tag>/; tools/summarize_prof.py condenses them.
set -u
TAG=${1:-r00}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

run() { # name, limit, args...
    local name=$1 limit=$2
    shift 2
    echo "== $name"
    timeout -s KILL "$limit" "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "   rc=$rc"
    if [ $rc -ne 0 ]; then tail -5 "$OUT/$name.log"; exit $rc; fi
}

BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs --no-sequence --sustain 0 --one-stream"
# (a chip that has been idle raises its clocks over the first twenty or so launches of a big
# kernel -- the tracker reads 960 us at first and 845 after 20 steps in one and the same trace --,
# so the runs are long enough for the averages to be those of the settled clocks)
for wl in framepair c3 c4 f1 f2 f3 vo; do
    run stats_$wl 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$wl" -o $wl \
        -- $BENCH --workload $wl --steps 100 --warmup 50
done
run stats_c5 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c5" -o c5 \
    -- $BENCH --workload c5 --frames 96 --steps 3 --warmup 1
for wl in framepair c3 c4; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
        run pmc_${wl}_$ctr 120 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_${wl}_$ctr" -o $wl \
            -- $BENCH --workload $wl --steps 2 --warmup 1
    done
done
ls -R "$OUT" | head -60
