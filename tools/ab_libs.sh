#!/bin/bash
# A/B of library builds on one chip (run on a GPU box): ab/libsfm_<variant>.so, linked beforehand from
# variant objects, takes the library's place in this scratch copy in turn; prints the frame-pair step, the
# tracker and the triangulation kernel per variant.  Usage: bash tools/ab_libs.sh base A B base
set -u
mkdir -p gpurun_out/ab
cp sfm_amd/lib/libsfm_hip.so /tmp/libsfm_keep.so
for v in "$@"; do
  cp ab/libsfm_$v.so sfm_amd/lib/libsfm_hip.so
  timeout -k 10 120 python bench.py --one-stream --sustain 0 --no-other-configs --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/ab/fp_$v.json 2> gpurun_out/ab/fp_$v.err || { echo "framepair $v failed"; break; }
  timeout -k 10 120 python bench.py --workload c3 --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/ab/c3_$v.json 2> gpurun_out/ab/c3_$v.err || { echo "c3 $v failed"; break; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
a = json.loads(open(f"gpurun_out/ab/fp_{v}.json").read().strip().splitlines()[-1])
b = json.loads(open(f"gpurun_out/ab/c3_{v}.json").read().strip().splitlines()[-1])
print(v, "step ms", a["ms_per_step"], "track us", a["kernels"]["lk_track_kernel"]["avg_us"],
      "c3 ms", b["ms_per_step"], "c3 kern us", b["roofline"]["avg_launch_us"], flush=True)
PY
done
cp /tmp/libsfm_keep.so sfm_amd/lib/libsfm_hip.so
