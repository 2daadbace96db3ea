#!/bin/bash
# A longer random-input campaign than tools/fuzz_all.sh for the kernels that changed last
# (run on a GPU box); prints one summary line per generator and keeps going after a failure.
# Usage: bash tools/fuzz_long.sh [part]   part = a | b (each fits one 20-minute GPU call)
set -u
part=${1:-a}
run() { echo "== $*"; timeout -k 10 900 python "$@" 2>&1 | tail -1; echo "   rc=${PIPESTATUS[0]}"; }
if [ "$part" = a ]; then
  run tools/lk_fuzz.py 6000 21
  run tools/lk_fuzz.py 1200 21 batch
  run tools/lk_fuzz.py 3000
  run tools/flow_chain_fuzz.py 500
else
  run tools/l2_fuzz.py 8000
  run tools/match_pairs_fuzz.py 3000
  run tools/triangulate_fuzz.py 4000
  run tools/corners_fuzz.py 5000
  run tools/corners_batch_fuzz.py 2000
fi
