#!/bin/bash
# tools/gpu_ab_driver.sh <variant> ...: the driver's bench command with the product library ("base") and variant libraries, interleaved, three rounds
cd $GRAFT_REPO_ROOT
for rnd in 1 2 3; do for v in base "$@"; do
  if [ $v = base ]; then unset MMD_LIB_DIR; else export MMD_LIB_DIR=variants/$v; fi
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cold --no-loopback 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('%-10s' % '$v', [round(v) for v in d['value_windows']], 'ms/step %.4f' % d['ms_per_step'], 'kernel %.4f span %.4f overhead %.4f' % (r['kernel_ms'], r['kernel_span_ms_all_launches'] or 0, r['dispatch_and_completion_overhead_ms'] or 0))"
done; done
