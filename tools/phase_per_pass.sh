#!/bin/bash
# mean wavefront time per pass of the persistent loop, per phase (development builds with -DM4Q_DEV_PHASE_CLOCK):
#   tools/phase_per_pass.sh <lib> [ENV=VAL ...]     -> ns per pass for backward / forward and the launch time
lib=$1; shift
for kv in "$@"; do export "$kv"; done
M4Q_LIB=$lib M4Q_PHASE_TRACE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $M4Q_BENCH_ARGS 2>&1 | python3 -c "
import sys, json, re
t = {}; ms = None
for line in sys.stdin:
    m = re.match(r'm4q phase (.*?)\s+(\d+)( ticks)?', line)
    if m: t[m.group(1).strip()] = int(m.group(2))
    if line.startswith('{'): ms = json.loads(line)['roofline']['avg_launch_ms']
p = t.get('passes', 1)
out = ['%s %.0f' % (k.split('|')[0].strip(), 10.0 * v / p) for k, v in t.items() if v > 0 and 'passes' not in k and 'exact' not in k.split('|')[0]]
print('launch %.2f ms | ns per pass: %s' % (ms or -1, '; '.join(out)))
"
