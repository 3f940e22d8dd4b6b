"""Build libm4q_hip.so for gfx950: one object per problem shape (m4q_kernels.hip with
-DM4Q_NX/-DM4Q_NU/-DM4Q_ORDER), one for the C ABI, linked in-tree next to the Python package.

    python mpc4quantum_amd/csrc/build.py [--force] [--jobs N] [--shfl]
"""
import argparse
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(PKG, "libm4q_hip.so")
LIB_GEN = os.path.join(PKG, "libm4q_hip_gen.so")   # the closed-loop kernels with the generator plant (m4q_kernels.hip: M4Q_VARIANT_GEN)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-command-line-argument"]
HEADERS = ["m4q_device.h", "m4q_dpp_gen.h", "m4q_mpc.h", "m4q_tile.h", "m4q_tile3.h", "m4q_args.h", "m4q_shapes.inc",
           os.path.join("..", "..", "include", "m4q.h")]
STAMP = os.path.join(OBJ, "flags.stamp")       # the extra flags the objects in OBJ were built with


def shapes():
    """[(nx, nu, order, plant_only)] from m4q_shapes.inc."""
    txt = open(os.path.join(HERE, "m4q_shapes.inc")).read()
    return [(int(a), int(b), int(c), "plant-only" in rest)
            for a, b, c, rest in re.findall(r"^M4Q_SHAPE\((\d+),\s*(\d+),\s*(\d+)\)(.*)$", txt, re.M)]


def stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def run(cmd):
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + res.stdout)
        raise RuntimeError("hipcc failed")
    return res.stdout


def build(force=False, jobs=None, extra=()):
    """Objects built with other extra flags than the ones requested now (experiment / ablation / debug builds write to the
    same files) are rebuilt: a plain build() never ships the leftovers of an experiment."""
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    gen = os.path.join(HERE, "m4q_dpp_gen.h")
    if stale(gen, [os.path.join(HERE, "gen_dpp.py")]):          # the generated DPP statements follow their generator
        text = run([sys.executable, os.path.join(HERE, "gen_dpp.py")])
        if not os.path.exists(gen) or open(gen).read() != text:  # (rewritten - and everything rebuilt - only if they differ)
            with open(gen, "w") as f:
                f.write(text)
    want = " ".join(extra)
    have = open(STAMP).read() if os.path.exists(STAMP) else None
    if have != want:
        force = True
    jobs_list = []
    objs = []
    gen_objs = []
    for nx, nu, order, plant_only in shapes():
        obj = os.path.join(OBJ, "kernels_%d_%d_%d.o" % (nx, nu, order))
        objs.append(obj)
        src = os.path.join(HERE, "m4q_kernels.hip")
        dims = ["-DM4Q_NX=%d" % nx, "-DM4Q_NU=%d" % nu, "-DM4Q_ORDER=%d" % order]
        if force or stale(obj, [src] + hdrs):
            jobs_list.append([HIPCC] + COMMON + list(extra) + dims + (["-DM4Q_PLANT_ONLY"] if plant_only else []) + ["-c", src, "-o", obj])
        d = {4: 2, 9: 3, 16: 4}.get(nx)
        if d and not plant_only:                       # shapes with a device plant: their generator-plant closed-loop kernels
            gobj = os.path.join(OBJ, "kernelsg_%d_%d_%d.o" % (nx, nu, order))
            gen_objs.append(gobj)
            if force or stale(gobj, [src] + hdrs):
                jobs_list.append([HIPCC] + COMMON + list(extra) + dims + ["-DM4Q_VARIANT_GEN", "-c", src, "-o", gobj])
    capi = os.path.join(OBJ, "capi.o")
    objs.append(capi)
    src = os.path.join(HERE, "m4q_capi.hip")
    if force or stale(capi, [src] + hdrs):
        jobs_list.append([HIPCC] + COMMON + ["-c", src, "-o", capi])
    if jobs_list:
        with ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 2)) as ex:
            list(ex.map(run, jobs_list))
    if force or jobs_list or stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    if force or jobs_list or stale(LIB_GEN, gen_objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_GEN] + gen_objs)
    with open(STAMP, "w") as f:
        f.write(want)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--dev", action="store_true", help="development build (-DM4Q_DEV): required for --shfl and for any --define of a "
                                                       "development switch; the sources refuse those switches without it")
    ap.add_argument("--shfl", action="store_true", help="debug build (needs --dev): row broadcasts through ds_bpermute instead of DPP")
    ap.add_argument("--define", action="append", default=[], help="extra -D for the kernel objects (tuning experiments; forces a rebuild)")
    ap.add_argument("--flag", action="append", default=[], help="extra compiler flag for the kernel objects, e.g. "
                                                                "--flag=-mllvm --flag=-amdgpu-sched-strategy=max-ilp (forces a rebuild)")
    a = ap.parse_args()
    dev_names = ("M4Q_BCAST_SHFL", "M4Q_NOP", "M4Q_DEV_PHASE_CLOCK", "M4Q_TWO_INDEX_COMPLEX", "M4Q_EXP")
    if not a.dev and (a.shfl or any(d.split("=")[0] in dev_names for d in a.define)):
        sys.exit("development switches need --dev (the library they produce is not the product)")
    extra = (["-DM4Q_DEV"] if a.dev else []) + (["-DM4Q_BCAST_SHFL"] if a.shfl else []) + ["-D" + d for d in a.define] + list(a.flag)
    print(build(a.force or bool(extra), a.jobs, extra))
