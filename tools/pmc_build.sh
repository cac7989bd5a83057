#!/bin/bash
# SQ / LDS / VMEM counters of one kernel (default k_build_rows) from a few neighbor builds at -s 80
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/pmc_build}; KRE=${2:-k_build_rows}
mkdir -p "$OUT"
CMD=(tools/prof_force.py --steps 20 --kernels 1)
run() { name=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$KRE" --output-format csv -d "$OUT/$name" -o p -- python "${CMD[@]}" > "$OUT/$name.log" 2>&1 || echo "pass $name failed/timeout"; }
run sq   SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY
run sq2  SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE
run lds  SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
run vmem SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CU_CYCLES
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
            print("%-42s %-28s avg/dispatch %.6g (n=%d)" % (kn, cn, s / n, n))
PY
