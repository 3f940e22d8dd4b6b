#!/usr/bin/env python3
"""Merge the records tools/pmc_collect.py left under gpurun_out/<tag>/ into profiles/<tag>_pmc.json (read by bench.py), copy the
counter tables and kernel-trace stats next to it, and print DESIGN.md's table of configurations from the merged file.
    python3 tools/pmc_merge.py [--tag r02]
Fields a record had before and the new one lacks (the FETCH_SIZE calibration factors) are kept."""
import argparse
import glob
import json
import os
import shutil

ap = argparse.ArgumentParser()
ap.add_argument("--tag", default="r05")
a = ap.parse_args()
dst = "profiles/%s_pmc.json" % a.tag
P = json.load(open(dst)) if os.path.exists(dst) else {}
for f in sorted(glob.glob("gpurun_out/%s/pmc_config*.json" % a.tag)):
    for key, rec in json.load(open(f)).items():
        for k, v in P.get(key, {}).items():
            rec.setdefault(k, v)
        P[key] = rec
        for src, name in (("pmc_%s.txt", "%s_pmc_%s.txt"), ("stats_%s.csv", "%s_kernel_stats_%s.csv")):
            s = "gpurun_out/%s/%s" % (a.tag, src % key)
            if os.path.exists(s):
                shutil.copy(s, "profiles/" + name % (a.tag, key))
json.dump(P, open(dst, "w"), indent=1)
print("| key | launch | horizon-steps/s | FMA issue slots filled | VALU busy | waves waiting | clock | L2-miss traffic (fetch + write) |")
print("|---|---|---|---|---|---|---|---|")
for key, r in P.items():
    c = r["counters"]
    ms = r["traced_avg_launch_ms"]
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    slots = 1024 * cycles / 4.0
    pmc_ms = sum(r["pmc_pass_launch_ms"]) / len(r["pmc_pass_launch_ms"])
    print("| %s | %.1f ms | %.3g | %.1f %% | %.1f %% | %.1f %% | %.2f GHz | %.0f + %.0f GB |" % (
        key, ms, r["horizon_steps_per_launch"] / (ms * 1e-3), 100 * c["SQ_INSTS_VALU_FMA_F64"] / slots,
        100 * c["SQ_ACTIVE_INST_VALU"] / slots, 100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], cycles / (pmc_ms * 1e-3) / 1e9,
        r.get("fetch_factor", 2.0) * c["FETCH_SIZE"] * 1024 / 1e9, r.get("write_factor", 1.0) * c["WRITE_SIZE"] * 1024 / 1e9))
