#!/bin/bash
# One stream against two for the frame-pair step, on one chip (run on a GPU box).
set -u
mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for v in "" "--one-stream"; do
    timeout -k 10 120 python bench.py $v --sustain 0 --no-other-configs --no-cpu-baseline --no-sequence --steps 40 --warmup 10 > gpurun_out/ab/str.json 2> gpurun_out/ab/str.err || { echo failed; tail -3 gpurun_out/ab/str.err; exit 1; }
    python - "$v" <<'PY'
import json, sys
a = json.loads(open("gpurun_out/ab/str.json").read().strip().splitlines()[-1])
print(sys.argv[1] or "two streams", "step ms", a["ms_per_step"], flush=True)
PY
  done
done
