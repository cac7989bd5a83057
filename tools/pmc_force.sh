#!/bin/bash
# PMC passes for one kernel (separate rocprofv3 runs: SQ / TA+TCP / TCC / FETCH / WRITE), per MI355X_MICROARCH.md
# usage: tools/pmc_force.sh <outdir> <kernel-regex> <bench args...>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=$1; KRE=$2; shift 2
mkdir -p "$OUT"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$KRE" --output-format csv -d "$OUT/$name" -o p -- python bench.py "${BENCH_ARGS[@]}" > "$OUT/$name.log" 2>&1; }
BENCH_ARGS=("$@")
run sq   SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY
run sq2  SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64
run ta   TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run tcp  TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum
run tcp2 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum
run tcc  TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for d in sorted(os.listdir(out)):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            per[k][0] += float(r["Counter_Value"]); per[k][1] += 1
        ndisp = {}
        for (kn, cn), (s, n) in per.items():
            agg[(kn, cn)] = (s / n, n)
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for (kn, cn), (avg, n) in agg.items():
        line = "%-62s %-36s avg/dispatch %.6g  (n=%d)" % (kn, cn, avg, n)
        print(line); fo.write(line + "\n")
PY
