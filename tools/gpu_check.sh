#!/bin/bash
# usage (inside gpurun): tools/gpu_check.sh [tests|notests]   -- GPU parity suite then the bench on both paths
set -o pipefail
if [ "$1" != "notests" ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/tests.log 2>&1
  echo "tests exit $?"; tail -3 gpurun_out/tests.log
fi
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('real   ', '%.4g' % d['value'], 'ms', '%.2f' % d['ms_per_step'], 'frac', '%.3f' % d['roofline']['frac'])"
M4Q_FORCE_COMPLEX=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('complex', '%.4g' % d['value'], 'ms', '%.2f' % d['ms_per_step'], 'frac', '%.3f' % d['roofline']['frac'])"
