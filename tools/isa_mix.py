"""Static instruction mix of the loops of one kernel in a built shape object:
   python tools/isa_mix.py 9_2_1 'mpc_kernelIdLi1ELb0E'      (development tool)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

B = "/opt/rocm/lib/llvm/bin/"
obj = "/root/repo/mpc4quantum_amd/csrc/build/kernels_%s.o" % sys.argv[1]
pat = sys.argv[2]
tmp = tempfile.mkdtemp()
subprocess.check_call([B + "llvm-objcopy", "--dump-section", ".hip_fatbin=%s/fat.bin" % tmp, obj])
subprocess.check_call([B + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                       "--input=%s/fat.bin" % tmp, "--output=%s/k.co" % tmp, "--unbundle"])
dis = subprocess.check_output([B + "llvm-objdump", "-d", "--no-show-raw-insn", "%s/k.co" % tmp]).decode()
# split by symbol
blocks = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", dis)
body = None
for b in blocks:
    m = re.match(r"[0-9a-f]+ <([^>]+)>:", b)
    if m and pat in m.group(1) and not m.group(1).endswith(".kd"):
        body = b
        print("kernel", m.group(1))
        break
if body is None:
    sys.exit("kernel not found")
ins = []          # (addr, mnemonic, text)
for line in body.splitlines():
    m = re.match(r"\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)", line)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2) + " " + m.group(4)))
addr_index = {a: i for i, (a, _, _) in enumerate(ins)}


def cls(mn, text):
    if mn.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_pk_fma", "v_pk_mul", "v_pk_add")):
        return "fp64" + ("_dpp" if "dpp" in mn or "row_newbcast" in text else "")
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem" if not mn.startswith("scratch_") else "scratch"
    if mn.startswith("s_waitcnt"):
        return "waitcnt"
    if mn.startswith("s_nop"):
        return "nop"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith(("v_mov", "v_accvgpr")):
        return "vmov"
    if mn.startswith("v_"):
        return "valu_other"
    return "other"


loops = []
for i, (a, mn, text) in enumerate(ins):
    if mn.startswith("s_cbranch") or mn == "s_branch":
        m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", text)
        if m:
            # objdump prints symbol+offset; turn into absolute using the kernel's first address
            tgt = ins[0][0] + int(m.group(1), 16)
            if tgt in addr_index and addr_index[tgt] < i:
                loops.append((addr_index[tgt], i))
tot = collections.Counter(cls(mn, t) for _, mn, t in ins)
print("whole kernel: %d instructions  %s" % (len(ins), dict(tot)))
for lo, hi in sorted(loops, key=lambda r: r[0] - r[1])[:12]:
    c = collections.Counter(cls(mn, t) for _, mn, t in ins[lo:hi + 1])
    n = hi - lo + 1
    print("loop @%06x..%06x  %5d instr: " % (ins[lo][0], ins[hi][0], n) + "  ".join("%s %d (%.0f%%)" % (k, v, 100.0 * v / n)
                                                                                  for k, v in c.most_common()))
