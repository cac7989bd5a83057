#!/bin/bash
# tools/gpu_ab_lib.sh <size> <variant> ...: bench.py at --size with the product library ("base") and with every named library under variants/, interleaved, three rounds
cd $GRAFT_REPO_ROOT
S=$1; shift
for rnd in 1 2 3; do for v in base "$@"; do
  if [ $v = base ]; then unset MMD_LIB_DIR; else export MMD_LIB_DIR=variants/$v; fi
  python bench.py --size $S --no-cpu-baseline --no-cold --steps 400 --warmup 40 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%-10s' % '$v', [round(v) for v in d['value_windows']], 'ms/step %.4f' % d['ms_per_step'])"
done; done
