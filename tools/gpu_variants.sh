#!/bin/bash
# tools/gpu_variants.sh "<config> <config> ..." <variant> <variant> ...: tools/run_one.py <config> with the product library ("base") and
# with every named library under variants/, interleaved, twice (run on the GPU box)
cd $GRAFT_REPO_ROOT
cfgs=$1; shift
for rep in 1 2; do
  for c in $cfgs; do
    for v in base "$@"; do
      if [ $v = base ]; then out=$(python tools/run_one.py $c 2>&1 | tail -1); else out=$(MMD_LIB_DIR=variants/$v python tools/run_one.py $c 2>&1 | tail -1); fi
      printf "%-12s %s\n" $v "$out"
    done
  done
done
