#!/bin/bash
# Register / spill / LDS figures of every kernel in one built shape object:  tools/kernel_regs.sh 9_2_2 [filter] [object file]
set -e
obj=${3:-/root/repo/mpc4quantum_amd/csrc/build/kernels_$1.o}
tmp=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $obj
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/k.co --unbundle
$B/llvm-readelf --notes $tmp/k.co | python3 -c "
import sys, re
# amdhsa.kernels is a YAML list: one item per kernel, opened by '  - .<first key>' (keys sorted: .agpr_count first)
rows, cur = [], None
for line in sys.stdin:
    if re.match(r'\s+- \.\w+:', line) and not re.match(r'\s+- \.(offset|address_space|actual_access|name|size|value_kind):', line):
        cur = {}
        rows.append(cur)
    m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)', line)
    if m and cur is not None and m.group(1) in ('name', 'vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count',
                                                'private_segment_fixed_size', 'agpr_count', 'group_segment_fixed_size'):
        if m.group(1) != 'name' or 'kernel' in m.group(2):
            cur[m.group(1)] = m.group(2).strip()
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for r in rows:
    n = r.get('name', '')
    if n and flt in n:
        print('%-92s vgpr %4s agpr %4s sgpr %4s | spilled vgpr %4s sgpr %4s scratch %5s B' % (n[:92], r.get('vgpr_count'), r.get('agpr_count'), r.get('sgpr_count'), r.get('vgpr_spill_count'), r.get('sgpr_spill_count'), r.get('private_segment_fixed_size')))
" "$2"
rm -rf $tmp
