#!/bin/bash
# A/B of library builds for the float-matching workload on one chip (run on a GPU box):
# ab/libsfm_<variant>.so takes the library's place in this scratch copy in turn; prints the step and
# the kernels of the call per variant.  Usage: bash tools/ab_libs_c4.sh base A B base
set -u
mkdir -p gpurun_out/ab
cp sfm_amd/lib/libsfm_hip.so /tmp/libsfm_keep.so
for v in "$@"; do
  cp ab/libsfm_$v.so sfm_amd/lib/libsfm_hip.so
  timeout -k 10 120 python bench.py --workload c4 --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/ab/c4_$v.json 2> gpurun_out/ab/c4_$v.err || { echo "c4 $v failed"; tail -3 gpurun_out/ab/c4_$v.err; break; }
  python - "$v" gpurun_out/ab/c4_$v.json <<'PY'
import json, sys
a = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = a["variants"]["unit_norm_real"]["kernels"]
print(sys.argv[1], "step ms", a["ms_per_step"], {n: round(v["avg_us"], 1) for n, v in k.items()}, "integer", a["variants"]["sift_like_integer"]["ms_per_step"], flush=True)
PY
done
cp /tmp/libsfm_keep.so sfm_amd/lib/libsfm_hip.so
