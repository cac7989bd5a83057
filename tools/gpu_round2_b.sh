#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
BUILDKERNELS=1 timeout 600 python tools/prof_force.py --size 80 --steps 100 > gpurun_out/r2b/prof_build.log 2>&1
timeout 900 python tools/run_configs.py > gpurun_out/r2b/configs.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2b/pytest_gpu.log 2>&1
tail -15 gpurun_out/r2b/pytest_gpu.log
cat gpurun_out/r2b/prof_build.log
cat gpurun_out/r2b/configs.log
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2b/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2b/bench_prof.log 2>&1)
tail -3 gpurun_out/r2b/bench_prof.log
python tools/rocpd_stats.py $(find gpurun_out/r2b/prof -name "*.db" | head -1) > gpurun_out/r2b/kernel_stats.md 2>&1
head -40 gpurun_out/r2b/kernel_stats.md
