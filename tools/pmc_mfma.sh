#!/bin/bash
# Matrix-pipe occupancy counters of one bench workload (developer tool).
# Usage: bash tools/pmc_mfma.sh <workload> [extra bench args]
set -u
WL=${1:-c4}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_mfma_$WL
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT" -o $WL -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs --sustain 0 --one-stream --workload $WL --steps 2 --warmup 1 "$@" > "$OUT/run.log" 2>&1
echo rc=$?
python3 - "$OUT" <<'PY'
import csv, glob, sys, re, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
seen = set()
for r in csv.DictReader(open(f)):
    m = re.search(r"([A-Za-z0-9_]+_kernel)", r["Kernel_Name"])
    k = m.group(1) if m else r["Kernel_Name"][:40]
    a = acc[k][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        d = dur[k]; d[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[1] += 1
for k, c in acc.items():
    if "rocclr" in k or "at::" in k or "elementwise" in k or "reduce" in k: continue
    v = {n: a[0] / a[1] for n, a in c.items()}
    w = v.get("SQ_WAVES", 0) or 1
    us = dur[k][0] / dur[k][1] / 1e3
    print(f"{k}: dur={us:.1f}us waves={w:.0f} valu/wave={v.get('SQ_INSTS_VALU',0)/w:.0f} wave_qcycles/wave={v.get('SQ_WAVE_CYCLES',0)/w:.0f} "
          f"active/wave={v.get('SQ_ACTIVE_INST_ANY',0)/w:.0f} wait_any/wave={v.get('SQ_WAIT_ANY',0)/w:.0f} wait_inst/wave={v.get('SQ_WAIT_INST_ANY',0)/w:.0f} "
          f"mfma_busy_cycles={v.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.0f} busy_cycles={v.get('SQ_BUSY_CYCLES',0):.0f} gui_active={v.get('GRBM_GUI_ACTIVE',0):.0f} clock_GHz~{v.get('GRBM_GUI_ACTIVE',0)/8/(us*1e3) if us else 0:.2f}")
PY
