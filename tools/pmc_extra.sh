#!/bin/bash
# extra stall/latency counters for one kernel (same conventions as pmc_force.sh)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=$1; KRE=$2; shift 2
mkdir -p "$OUT"
CMD=("$@")
run() { name=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$KRE" --output-format csv -d "$OUT/$name" -o p -- python "${CMD[@]}" > "$OUT/$name.log" 2>&1 || echo "pass $name failed/timeout"; }
run lds  SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
run vmem SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES
run misc SQ_IFETCH SQ_INSTS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SMEM
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(os.listdir(out)):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            per[k][0] += float(r["Counter_Value"]); per[k][1] += 1
        for (kn, cn), (s, n) in per.items():
            if n > 50: print("%-42s %-28s avg/dispatch %.6g" % (kn, cn, s / n))
PY
