#!/bin/bash
# tools/gpu_bench_driver.sh <outdir> [bench args]: the driver's bench command on the GPU box, line kept under gpurun_out/<outdir>/ and summarised
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-bench}; shift
mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $O/bench_driver.json 2> $O/bench_driver.err
python - $O/bench_driver.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("value", d["value"], d["value_windows"], "ms/step", d["ms_per_step"])
print({k: r.get(k) for k in ("frac", "frac_sampled", "kernel_ms", "kernel_span_ms_all_launches", "dispatch_and_completion_overhead_ms", "frac_span_only", "launches_all", "kernel_ms_sampled", "launches_sampled", "frac_kernel_only", "kernel_only_ms")})
print("loopback", d.get("rank_path_loopback"))
print("cold", (d.get("perf_summary_cold") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d["phases_s"], d["force_launched_behind_build"])
PY
tail -3 $O/bench_driver.err
