#!/bin/bash
# Register / spill / LDS figures of every kernel in one built shape object:  tools/kernel_regs.sh 9_2_2 [filter]
set -e
obj=/root/repo/mpc4quantum_amd/csrc/build/kernels_$1.o
tmp=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $obj
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/k.co --unbundle
$B/llvm-readelf --notes $tmp/k.co | python3 -c "
import sys, re
cur = {}
rows = []
for line in sys.stdin:
    m = re.match(r'\s*-?\s*\.?(\w+):\s*(.*)', line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == 'name' and ('kernel' in v) and not v.endswith('.kd'):
        cur['name'] = v
    if k in ('vgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size', 'agpr_count'):
        cur[k] = v
    if k == 'symbol':
        rows.append(cur); cur = {}
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for r in rows:
    n = r.get('name', '?')
    if flt in n:
        print('%-110s vgpr %4s agpr %4s spill %4s scratch %5s' % (n[:110], r.get('vgpr_count'), r.get('agpr_count'), r.get('vgpr_spill_count'), r.get('private_segment_fixed_size')))
" "$2"
rm -rf $tmp
