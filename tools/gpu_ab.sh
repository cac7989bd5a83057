#!/bin/bash
# A/B of bench.py under option sets: tools/gpu_ab.sh "opt=val,..." "opt=val,..." ...   (prints value, ms/step, in-run kernel ms, kernel-only ms)
cd $GRAFT_REPO_ROOT
for rnd in 1 2; do
for o in "$@"; do
  MMD_BENCH_OPTIONS="$o" python bench.py --no-cpu-baseline --no-loopback --steps 100 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
r = d['roofline']
print('%-28s value %.1f  ms/step %.4f  kernel_ms %.4f (frac %.3f)  kernel_only_ms %.4f (frac %.3f)' % ('$o', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel_only_ms'], r['frac_kernel_only']))"
done; done
