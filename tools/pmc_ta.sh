#!/bin/bash
# Texture-address / L1 counters of one bench workload (developer tool): how busy the vector
# memory path of a CU is beside its vector ALU.
# Usage: bash tools/pmc_ta.sh <workload> [extra bench args]
set -u
WL=${1:-framepair}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_ta_$WL
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
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
    us = dur[k][0] / dur[k][1] / 1e3
    print(k, f"dur={us:.1f}us", " ".join(f"{n}={x:.4g}" for n, x in sorted(v.items())))
PY
