#!/bin/bash
# tools/gpu_ab_nb.sh [variant dir names...] — same-box A/B of the neighbor build (+ binning) on resident data at -s 80 DP, EAM -s 64, half lists: product library against variants/<name>
cd $GRAFT_REPO_ROOT
for rnd in 1 2 3; do for v in base ${@:-nbbase}; do
  if [ $v = base ]; then unset MMD_LIB_DIR; else export MMD_LIB_DIR=variants/$v; fi
  timeout 120 python tools/prof_force.py --steps 40 --kernels 1 2>&1 | grep "neighbor_build" | sed "s/^/$v LJ s80 full: /"
done; done
for v in base ${@:-nbbase}; do
  if [ $v = base ]; then unset MMD_LIB_DIR; else export MMD_LIB_DIR=variants/$v; fi
  timeout 120 python tools/prof_force.py --steps 40 --kernels 1 --half 1 2>&1 | grep "neighbor_build" | sed "s/^/$v LJ s80 half: /"
  timeout 120 python tools/prof_force.py --steps 40 --kernels 1 --deck in.eam.miniMD --size 64 2>&1 | grep "neighbor_build" | sed "s/^/$v EAM s64 full: /"
done
