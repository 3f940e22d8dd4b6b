#!/usr/bin/env python3
"""Where the spills are: for every loop of a compiled kernel whose body holds > 100 DPP FMAs or >= 40 fp64 MFMAs (the horizon loops), count the
scratch (spill) instructions, LDS reads, waits and register moves inside it.

    python3 tools/hot_loops.py 9_2_1 'mpc_kernelIdLi1ELb0'        # shape, substring of the mangled kernel name
Reads mpc4quantum_amd/csrc/build/kernels_<shape>.o, or the object file given as fourth argument (third: DPP-FMA threshold, default 100) (runs on the CPU box:
disassembly only)."""
import re
import subprocess
import sys
import tempfile

B = "/opt/rocm/lib/llvm/bin/"


def main():
    shape, flt = sys.argv[1], sys.argv[2]
    obj = sys.argv[4] if len(sys.argv) > 4 else "mpc4quantum_amd/csrc/build/kernels_%s.o" % shape
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([B + "llvm-objcopy", "--dump-section", ".hip_fatbin=%s/fat.bin" % tmp, obj])
        subprocess.check_call([B + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=%s/fat.bin" % tmp, "--output=%s/k.co" % tmp, "--unbundle"])
        dis = subprocess.check_output([B + "llvm-objdump", "-d", "%s/k.co" % tmp], text=True)
    cur, funcs = None, {}
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur:
            funcs[cur].append(line)
    for name, L in funcs.items():
        if flt not in name or not L:
            continue
        def addr(i):
            m = re.search(r"//\s*([0-9A-F]{12}):", L[i])
            return int(m.group(1), 16) if m else None
        amap = {addr(i): i for i in range(len(L)) if addr(i) is not None}
        tot_scr = sum("scratch_" in x for x in L)
        print("%s: %d instructions, %d scratch instructions in all" % (name, len(amap), tot_scr))
        for i, l in enumerate(L):
            m = re.match(r"\s+(s_cbranch_\w+|s_branch)\s+(\d+)", l)
            if not m:
                continue
            off = int(m.group(2))
            off = off - 65536 if off > 32767 else off
            if off >= 0:
                continue
            t = addr(i) + 4 + 4 * off
            if t not in amap:
                continue
            body = L[amap[t]:i + 1]
            dpp = sum("v_fmac_f64_dpp" in x for x in body)
            mfma = sum("v_mfma_f64" in x for x in body)
            if (dpp <= int(sys.argv[3] if len(sys.argv) > 3 else 100) and mfma < 40) or len(body) > 4000:
                continue
            cnt = lambda p: sum(re.match(r"\s+" + p, x) is not None for x in body)   # noqa: E731
            print("   loop of %4d instructions: %3d fp64 MFMA, %4d v_fmac_f64_dpp, %3d other fp64 FMA, scratch loads %d stores %d, LDS reads %d, "
                  "s_waitcnt %d, s_nop %d, v_mov/accvgpr %d" % (len(body), mfma, dpp, cnt("v_fma_f64|v_fmac_f64_e"), cnt("scratch_load"),
                                                                cnt("scratch_store"), cnt("ds_read"), cnt("s_waitcnt"), cnt("s_nop"),
                                                                cnt("v_mov|v_accvgpr")))


if __name__ == "__main__":
    main()
