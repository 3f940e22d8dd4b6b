#!/bin/bash
# usage (inside gpurun): tools/power_probe.sh "<bench args>" [seconds] [ENV=VAL ...]
# Socket power (rocm-smi) sampled every 0.5 s while bench.py keeps the GPU busy with back-to-back launches; prints min / mean / max
# of the samples taken under load, the power cap, and the launch time of the run.
args=$1; secs=${2:-8}; shift 2
for kv in "$@"; do export "$kv"; done
python bench.py --steps 600 --warmup 2 --no-cpu-baseline $args > /tmp/power_bench.json 2>/dev/null &
pid=$!
sleep 4
: > /tmp/power_samples.txt
for i in $(seq 1 $((secs*2))); do
  rocm-smi --showpower 2>/dev/null | grep -E "Power \(W\)" | sed 's/.*: //' >> /tmp/power_samples.txt
  sleep 0.5
done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
cap=$(rocm-smi --showmaxpower 2>/dev/null | grep -E "Power \(W\)" | sed 's/.*: //')
python3 - "$args $*" "$cap" <<'PY'
import sys
v = [float(x) for x in open('/tmp/power_samples.txt').read().split() if float(x) > 400]
print("%-46s socket power under load: min %.0f mean %.0f max %.0f W over %d samples (cap %s W)" % (sys.argv[1].strip() or "config 3", min(v), sum(v) / len(v), max(v), len(v), sys.argv[2]))
PY
