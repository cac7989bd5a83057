#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
BUILDKERNELS=1 timeout 600 python tools/prof_force.py --size 80 --steps 100 > gpurun_out/r2a/prof_build.log 2>&1
timeout 900 python tools/run_configs.py > gpurun_out/r2a/configs.log 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest_gpu.log 2>&1
tail -5 gpurun_out/r2a/pytest_gpu.log
cat gpurun_out/r2a/prof_build.log
cat gpurun_out/r2a/configs.log
