#!/bin/bash
# usage: tools/gpu_tests.sh <outdir> [pytest args]   (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift
mkdir -p $out
timeout 2700 python -m pytest tests -m gpu -q "$@" > $out/pytest_gpu.log 2>&1
tail -25 $out/pytest_gpu.log | cut -c1-400
