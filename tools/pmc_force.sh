#!/bin/bash
# PMC passes for one kernel, each in its own rocprofv3 run (per MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do
# not fit one pass; never combine --pmc with trace domains other than --kernel-trace). Every run is wrapped in
# `timeout` (a bad counter set makes rocprofv3 hang after aborting).
# usage: tools/pmc_force.sh <outdir> <kernel-regex> <python script + args...>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=$1; KRE=$2; shift 2
mkdir -p "$OUT"
run() { name=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$KRE" --output-format csv -d "$OUT/$name" -o p -- python "${CMD[@]}" > "$OUT/$name.log" 2>&1 || echo "pass $name failed/timeout"; }
CMD=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc  TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run sq   SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY
run sq2  SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
lines = []
for d in sorted(os.listdir(out)):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])
            per[k][0] += float(r["Counter_Value"]); per[k][1] += 1
        for (kn, cn), (s, n) in per.items():
            lines.append("%-50s %-28s avg/dispatch %.6g  (dispatches=%d)" % (kn, cn, s / n, n))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
