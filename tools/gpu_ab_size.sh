#!/bin/bash
# tools/gpu_ab_size.sh <size> "opt=val,..." "opt=val,..." ...: bench.py at --size under option sets, interleaved, three rounds (value windows, ms/step)
cd $GRAFT_REPO_ROOT
S=$1; shift
for rnd in 1 2 3; do for o in "$@"; do
  MMD_BENCH_OPTIONS="$o" python bench.py --size $S --no-cpu-baseline --no-cold --steps 400 --warmup 40 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('-s $S %-24s' % '$o', [round(v) for v in d['value_windows']], 'ms/step %.4f' % d['ms_per_step'], 'neigh+comm per rebuild us %.1f' % (1e6 * (d['phases_s']['neigh'] + d['phases_s']['comm']) / max(1, d['steps'] // 20)))"
done; done
