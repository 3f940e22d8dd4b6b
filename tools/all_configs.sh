#!/bin/bash
# launch time of every BASELINE configuration (and the exact mode) on the current library, one line each:  tools/all_configs.sh [lib]
[ -n "$1" ] && export M4Q_LIB=$1
one() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %8.3f ms/launch   %.3e h-steps/s' % (' '.join(sys.argv[1:]) or 'config 3', d['roofline']['avg_launch_ms'], d['value']))" "$@"; }
one; one --config 2; one --config 4; one --config 5 --batch 131072; one --exact-qp; one --exact-qp --config 2; one --exact-qp --config 4
M4Q_FORCE_COMPLEX=1 one
echo '(config 4 on per-member models, M4Q_NO_SG=1:)'; M4Q_NO_SG=1 one --config 4
