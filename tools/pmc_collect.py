#!/usr/bin/env python3
"""Counter evidence for one bench configuration (run INSIDE gpurun, from the repo root):

    python3 tools/pmc_collect.py --tag r02 --config 3 [--batch B] [--exact-qp] [--complex]

Runs `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline ...` under rocprofv3 once per counter group (separate --pmc
passes, never combined with tracing; the program itself follows `--`), then once more under `--kernel-trace --stats`,
sums every counter over the dispatches of the closed-loop kernel, and writes

    gpurun_out/<tag>/pmc_<key>.txt      counter table + derived figures (copy to profiles/)
    gpurun_out/<tag>/pmc_<key>.json     {key: {...}}  (merge into profiles/<tag>_pmc.json: read by bench.py)
    gpurun_out/<tag>/stats_<key>.csv    rocprofv3 kernel stats of the traced run

key = config<N>_B<batch>_<real|complex>_<clip|exact>.  FETCH_SIZE / WRITE_SIZE are reported raw (KB, as rocprofv3 prints them);
the factors measured by tools/ubench_fetch.hip for this kernel's access widths are applied by the reader, not here."""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

GROUPS = [
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY",
    "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY",
    "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM",
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64",
    "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES",
    "FETCH_SIZE",
    "WRITE_SIZE",
    "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum",
]


def run(cmd, log):
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        return subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, env=env, timeout=600).returncode


def bench_line(log):
    for line in reversed(open(log).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r05")
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--exact-qp", action="store_true")
    ap.add_argument("--complex", action="store_true")
    a = ap.parse_args()
    out = os.path.join("gpurun_out", a.tag)
    os.makedirs(out, exist_ok=True)
    bench = ["python3", "bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--config", str(a.config)]
    if a.batch:
        bench += ["--batch", str(a.batch)]
    if a.exact_qp:
        bench += ["--exact-qp"]
    if a.complex:
        os.environ["M4Q_FORCE_COMPLEX"] = "1"
    tot, ms = {}, []
    line = None
    for i, group in enumerate(GROUPS):
        d = os.path.join(out, "pmc_tmp_%d" % i)
        log = os.path.join(out, "pmc_tmp_%d.log" % i)
        shutil.rmtree(d, ignore_errors=True)               # (a reused directory would add an earlier configuration's dispatches)
        rc = run(["rocprofv3", "--pmc"] + group.split() + ["--output-format", "csv", "-d", d, "--"] + bench, log)
        line = bench_line(log) or line
        if rc != 0 or bench_line(log) is None:
            print("pass %d (%s) failed, rc %d: see %s" % (i, group, rc, log))
            continue
        ms.append(bench_line(log)["roofline"]["avg_launch_ms"])
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "mpc_kernel" in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # traced run: kernel time without counters
    d = os.path.join(out, "trace_tmp")
    log = os.path.join(out, "trace_tmp.log")
    shutil.rmtree(d, ignore_errors=True)
    rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + bench[:3] + ["4", "--warmup", "1"] + bench[6:], log)
    traced = bench_line(log)
    b = traced or line
    key = "config%d_B%d_%s_%s" % (a.config, b["config"]["batch_per_gpu"], "real" if b["dtype"] == "f64" else "complex",
                                  "exact" if a.exact_qp else "clip")
    # (the same suffixes bench.py appends: which real path ran)
    wl = b["config"]["workload"]
    if wl.endswith("(real)"):
        key += "_real9"
    if wl.endswith("(traceless-tile)"):
        key += "_tile"
    if wl.endswith("(traceless-sg)"):
        key += "_sg"
    stats_ms = None
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(out, "stats_%s.csv" % key), "w") as g:
            g.write(open(f).read())
        for r in rows:
            if "mpc_kernel" in r["Name"]:
                stats_ms = float(r["AverageNs"]) / 1e6
    rec = {"kernel": b["roofline"]["kernel"], "counters": tot, "pmc_pass_launch_ms": ms,
           "traced_avg_launch_ms": stats_ms, "hip_event_launch_ms_same_run": traced["roofline"]["avg_launch_ms"] if traced else None,
           "horizon_steps_per_launch": b["config"]["qp_solves_per_step"] * b["config"]["horizon"],
           "value_traced_run": traced["value"] if traced else None}
    lines = ["# %s   kernel %s" % (key, rec["kernel"]),
             "# launch: %.3f ms traced (rocprofv3 --kernel-trace --stats, 4 launches), %.3f ms HIP events in the same run; PMC passes %s ms"
             % (stats_ms or -1, rec["hip_event_launch_ms_same_run"] or -1, ", ".join("%.2f" % v for v in ms)),
             "# horizon-steps per launch: %d" % rec["horizon_steps_per_launch"]]
    for k in sorted(tot):
        lines.append("%-30s %.6g" % (k, tot[k]))
    c = tot
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        lines.append("# wave time: waiting (SQ_WAIT_ANY) %.1f %%, issue stall (SQ_WAIT_INST_ANY) %.1f %%, issuing (SQ_ACTIVE_INST_ANY) %.1f %%"
                     % (100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
                        100 * c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]))
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_VALU_FMA_F64" in c:
        lines.append("# VALU instructions: %.1f %% are fp64 FMAs; fp64 MFMA ops: %g" % (100 * c["SQ_INSTS_VALU_FMA_F64"] / c["SQ_INSTS_VALU"],
                                                                                       c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)))
    open(os.path.join(out, "pmc_%s.txt" % key), "w").write("\n".join(lines) + "\n")
    json.dump({key: rec}, open(os.path.join(out, "pmc_%s.json" % key), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    sys.exit(main())
