#!/bin/bash
# LDS instruction / conflict counters of one bench workload (developer tool).
# Usage: bash tools/pmc_lds.sh <workload> [extra bench args]
set -u
WL=${1:-framepair}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_lds_$WL
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT \
    --output-format csv -d "$OUT" -o $WL -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs --sustain 0 --one-stream --workload $WL --steps 2 --warmup 1 "$@" > "$OUT/run.log" 2>&1
echo rc=$?
python3 - "$OUT" <<'PY'
import csv, glob, sys, re, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print(open(sys.argv[1] + "/run.log").read()[-1500:]); sys.exit(1)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(fs[0])):
    m = re.search(r"([A-Za-z0-9_]+_kernel)", r["Kernel_Name"])
    k = m.group(1) if m else r["Kernel_Name"][:40]
    a = acc[k][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, c in acc.items():
    if "rocclr" in k or "at::" in k: continue
    v = {n: a[0] / a[1] for n, a in c.items()}
    w = v.get("SQ_WAVES", 0) or 1
    print(k, {n: round(x / w, 1) for n, x in v.items() if n != "SQ_WAVES"}, "waves", w)
PY
