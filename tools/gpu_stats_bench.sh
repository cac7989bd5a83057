#!/bin/bash
# tools/gpu_stats_bench.sh [outdir]: rocprofv3 --kernel-trace --stats of the driver's bench command; prints the line's roofline block next to the kernel averages
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-stats_bench}
mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o t -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-loopback --no-cold > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err)
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) $O/kernel_stats_bench.md > /dev/null
head -14 $O/kernel_stats_bench.md | cut -c1-160
python - $O/bench.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
print("value", d["value"], {k: r.get(k) for k in ("frac", "frac_sampled", "kernel_ms", "kernel_span_ms_all_launches", "dispatch_and_completion_overhead_ms", "frac_span_only", "launches_all", "kernel_ms_sampled", "launches_sampled")})
PY
python tools/rocpd_timed_region.py $(find $O/kt -name "*.db" | head -1) $O/bench.json | tee $O/frac_crosscheck.txt
find $O -name "*.db" -delete
