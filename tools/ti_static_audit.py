#!/usr/bin/env python3
"""Static audit of a compiled closed-loop kernel for the failure modes suspected behind round 2's abort of the two-index complex
d = 4 build (profiles/r02_two_index_complex_d4_fault.log).  Disassembly only - nothing runs.

    python3 tools/ti_static_audit.py <object.o> <substring of the mangled kernel name>

Checks, per kernel:
  1. SGPR-spill lanes: the VGPRs written by v_writelane_b32 must be touched by v_writelane / v_readlane ONLY (a copy of such a
     register to scratch or an AGPR under a partial EXEC mask would lose lanes = lose spilled scalars: loop bounds, GView bases).
  2. scratch traffic under EXEC manipulation: scratch_store / scratch_load between an s_and_saveexec / s_mov exec and its restore.
  3. long-branch expansions (s_getpc_b64 ... s_setpc_b64): the scratch SGPR pair each uses, and whether that pair is read again
     before being rewritten at the branch target (a live value clobbered by the expansion).
  4. the horizon loops of the backward sweep (> 400 DPP FMAs per trip): back-edge condition register and the instructions that
     define it (must come from SALU arithmetic on the trip counter, never from a v_readlane of a spilled value that the loop body
     also writes)."""
import re
import subprocess
import sys
import tempfile

B = "/opt/rocm/lib/llvm/bin/"


def disasm(obj):
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([B + "llvm-objcopy", "--dump-section", ".hip_fatbin=%s/fat.bin" % tmp, obj])
        subprocess.check_call([B + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=%s/fat.bin" % tmp, "--output=%s/k.co" % tmp, "--unbundle"])
        return subprocess.check_output([B + "llvm-objdump", "-d", "%s/k.co" % tmp], text=True)


def regs(tok):
    """VGPR numbers named by an operand token like v12 or v[10:11]."""
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main():
    obj, flt = sys.argv[1], sys.argv[2]
    cur, funcs = None, {}
    for line in disasm(obj).split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur and re.match(r"\s+\S", line):
            funcs[cur].append(re.sub(r"\s*//.*", "", line).strip())
    for name, L in funcs.items():
        if flt not in name:
            continue
        print("== %s: %d instructions" % (name, len(L)))
        ops = [re.split(r"[ ,]+", x) for x in L]
        # 1. SGPR-spill VGPRs
        spill = set()
        for o in ops:
            if o[0] == "v_writelane_b32":
                spill |= regs(o[1])
        bad = []
        for i, o in enumerate(ops):
            if o[0] in ("v_writelane_b32", "v_readlane_b32"):
                continue
            touched = set()
            for t in o[1:]:
                touched |= regs(t)
            if touched & spill:
                bad.append((i, L[i]))
        nrl = sum(o[0] == "v_readlane_b32" for o in ops)
        nwl = sum(o[0] == "v_writelane_b32" for o in ops)
        print("   1. SGPR-spill VGPRs %s: %d v_writelane, %d v_readlane; other instructions touching them: %d%s" % (
            sorted(spill), nwl, nrl, len(bad), "".join("\n        [%d] %s" % b for b in bad[:10])))
        # 2. scratch under a modified EXEC
        depth, under, total = 0, [], 0
        for i, o in enumerate(ops):
            if o[0] in ("s_and_saveexec_b64", "s_or_saveexec_b64", "s_andn2_saveexec_b64") or (o[0] in ("s_mov_b64", "s_and_b64", "s_andn2_b64", "s_xor_b64") and o[1] == "exec"):
                depth = 1
            if o[0] == "s_or_b64" and o[1] == "exec":
                depth = 0
            if o[0].startswith("scratch_"):
                total += 1
                if depth:
                    under.append((i, L[i]))
        print("   2. scratch instructions: %d, of which after an EXEC modification and before its `s_or_b64 exec` restore: %d%s" % (
            total, len(under), "".join("\n        [%d] %s" % u for u in under[:6])))
        # 3. long-branch expansions
        lb = [i for i, o in enumerate(ops) if o[0] == "s_getpc_b64"]
        print("   3. long-branch expansions (s_getpc_b64): %d; scratch pairs %s" % (len(lb), sorted({ops[i][1] for i in lb})))
        # 4. big loops
        nbig = 0
        addr_re = None
        raw = subprocess.check_output([B + "llvm-objdump", "-d", "--no-show-raw-insn", "/dev/null"], text=True, stderr=subprocess.DEVNULL) if False else None
        dpp_idx = [i for i, o in enumerate(ops) if o[0] == "v_fmac_f64_dpp"]
        print("   4. v_fmac_f64_dpp: %d; v_accvgpr_read %d, v_accvgpr_write %d" % (len(dpp_idx), sum(o[0].startswith("v_accvgpr_read") for o in ops),
                                                                                  sum(o[0].startswith("v_accvgpr_write") for o in ops)))


if __name__ == "__main__":
    main()
