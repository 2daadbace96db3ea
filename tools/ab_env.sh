#!/bin/bash
# A/B of environment settings on one chip (run on a GPU box): every argument is one setting
# ("NAME=value" or "-" for none); prints the frame-pair step and its kernels per setting.
# Usage: bash tools/ab_env.sh - SFM_LK_DERIV_IMAGE=1 -
set -u
mkdir -p gpurun_out/ab
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e timeout -k 10 120 python bench.py --one-stream --sustain 0 --no-other-configs --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/ab/env_$i.json 2> gpurun_out/ab/env_$i.err || { echo "run $i ($v) failed"; tail -3 gpurun_out/ab/env_$i.err; break; }
  python - "$v" gpurun_out/ab/env_$i.json <<'PY'
import json, sys
a = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "step ms", a["ms_per_step"], {k: round(v["avg_us"], 1) for k, v in a["kernels"].items()}, flush=True)
PY
done
