#!/bin/bash
# tools/gpu_ab_force.sh [variant dir names...] — same-box A/B of Force::compute (LJ full lists, -s 80 DP: force only and inside the step) : product library against variants/<name>
cd $GRAFT_REPO_ROOT
for rnd in 1 2 3; do for v in base ${@:-prev}; do
  if [ $v = base ]; then unset MMD_LIB_DIR; else export MMD_LIB_DIR=variants/$v; fi
  timeout 120 python tools/prof_force.py --steps 100 --kernels 0,1 2>&1 | grep "^force\|neighbor_build\|tile stats" | tr '\n' ' ' | sed "s/^/$v: /"; echo
  timeout 120 python bench.py --no-cpu-baseline --no-loopback --no-cold --steps 100 --warmup 20 --windows 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$v: bench %.1f %s  kernel_ms %.5f frac %.4f kernel_only %.5f' % (d['value'], [round(x) for x in d['value_windows']], r['kernel_ms'], r['frac'], r['kernel_only_ms']))"
done; done
