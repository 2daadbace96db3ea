#!/bin/bash
# A/B of environment settings for the float-matching workload on one chip (run on a GPU box):
# every argument is one setting ("NAME=value" or "-" for none).
set -u
mkdir -p gpurun_out/ab
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e timeout -k 10 120 python bench.py --workload c4 --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/ab/c4_$i.json 2> gpurun_out/ab/c4_$i.err || { echo "run $i ($v) failed"; tail -3 gpurun_out/ab/c4_$i.err; break; }
  python - "$v" gpurun_out/ab/c4_$i.json <<'PY'
import json, sys
a = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = a["variants"]["unit_norm_real"]["kernels"]
print(sys.argv[1], "step ms", a["ms_per_step"], {n: round(v["avg_us"], 1) for n, v in k.items()}, "integer", a["variants"]["sift_like_integer"]["ms_per_step"], flush=True)
PY
done
