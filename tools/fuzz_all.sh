#!/bin/bash
# Long runs of every random-input tool on a GPU box; the summary lines go to one log
# (gpurun_out/fuzz_all.log) that is committed under profiles/ per round.
# Usage: bash tools/fuzz_all.sh
set -u
mkdir -p gpurun_out
LOG=gpurun_out/fuzz_all.log
: > $LOG
run() { # tool args...
  echo "== $*" | tee -a $LOG
  timeout -k 10 600 python -u tools/"$1".py "${@:2}" > gpurun_out/fuzz_last.txt 2>&1
  rc=$?
  grep -E "MISMATCH|cases agree|masks identical" gpurun_out/fuzz_last.txt | tail -8 | tee -a $LOG
  echo "   rc=$rc" | tee -a $LOG
  if [ $rc -ge 124 ]; then echo "stopping: time limit or fault" | tee -a $LOG; exit $rc; fi
}
run hamming_fuzz 3000
run hamming_fuzz 60 big
run l2_fuzz 3000
run l2_fuzz 30 big
run lk_fuzz 2000
run lk_fuzz 1500 21
run lk_fuzz 400 21 batch
run lk_fuzz 30 21 big
run corners_fuzz 2500
run corners_fuzz 40 big
run match_pairs_fuzz 1500
run triangulate_fuzz 1500
run geometry_fuzz 150
run pose_fuzz 600
run epipolar_pairs_fuzz 1500
run corners_batch_fuzz 1000
run flow_chain_fuzz 200
