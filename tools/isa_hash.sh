#!/bin/bash
# sha256 of the gfx950 disassembly of every kernel in a shape object (symbol by symbol): "did this source change alter any code?"
#   tools/isa_hash.sh <object.o> > hashes.txt
B=/opt/rocm/lib/llvm/bin; tmp=$(mktemp -d)
$B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $1 && $B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/k.co --unbundle
$B/llvm-objdump -d $tmp/k.co | python3 -c "
import sys, re, hashlib
cur=None; h={}
for line in sys.stdin:
    m=re.match(r'^[0-9a-f]+ <(\S+)>:', line)
    if m: cur=m.group(1); h[cur]=hashlib.sha256(); continue
    if cur and re.match(r'\s+\S', line):
        # drop addresses / encodings / resolved branch targets: the instruction text only
        h[cur].update(re.sub(r'\s*//.*', '', line).strip().encode()+b'\n')
for k in sorted(h): print(h[k].hexdigest()[:16], k)
"
rm -rf $tmp
