#!/usr/bin/env python3
"""Turn the admissions a GPU run recorded into the committed expectation the parity tests check against.

    M4Q_RECORD_ADMISSIONS=1 python -m pytest tests -m gpu -q          # on the GPU box: records, asserts nothing about the list
    python tools/parity_admissions.py [--merge] [--force]             # gpurun_out/parity_admissions_measured.json -> profiles/r04_parity_admissions.json

Without --merge the new record REPLACES the committed one, and is refused (exit 2) when it would lose something: fewer
admissions than the committed "measured" block holds, or a committed case the recording run did not execute ("cases_run" in the
measured file) - a partial run (pytest -k ...) must be folded in with --merge, which touches only the cases it executed.  --force
overrides (a kernel change that really removed admissions: say so in the commit).

A teacher-forced MPC step that misses the fixed parity bounds may pass on `tol + 10 x (what the ORACLE itself moves by under a
1e-15 perturbation of the guess the step starts from)` only if this file lists that step for that case
(tests/test_gpu_parity.py: _admit).  "allowed" is what the tests read; "measured" keeps the errors and the oracle's sensitivity of
the recording run (order: us[k], xs[k+1], X guess left behind, U guess left behind; exact cases: us[k] / sat, xs[k+1])."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = [a for a in sys.argv[1:] if a not in ("--merge", "--force")]
    src = args[0] if args else os.path.join(ROOT, "gpurun_out", "parity_admissions_measured.json")
    meas = json.load(open(src))
    recs = meas["admissions"]
    ran = set(meas.get("cases_run", [])) | {r["case"] for r in recs}
    dst = os.path.join(ROOT, "profiles", "r04_parity_admissions.json")
    old = json.load(open(dst))["measured"] if os.path.exists(dst) else []
    if "--merge" in sys.argv:
        # a partial recording run (python -m pytest -k ...): cases it EXECUTED replace their records, the others keep theirs
        recs = [r for r in old if r["case"] not in ran] + recs
    elif "--force" not in sys.argv:
        lost = sorted({r["case"] for r in old} - ran)
        if lost or len(recs) < len(old):
            sys.exit("refused: the new record (%d admissions, %d cases run) would replace a fuller one (%d admissions)%s - use "
                     "--merge for a partial run, --force if admissions really went away"
                     % (len(recs), len(ran), len(old), "; committed cases it did not execute: %s" % lost if lost else ""))
    allowed = {}
    for r in recs:
        allowed.setdefault(r["case"], [])
        if r["step"] not in allowed[r["case"]]:
            allowed[r["case"]].append(r["step"])
    out = {"note": "steps of the teacher-forced GPU parity tests that pass on the oracle-sensitivity clause instead of the fixed bounds; "
                   "every case not listed admits nothing (config 3 order 1 - the headline -, configs 1 and 2, every exact-mode case)",
           "allowed": {k: sorted(v) for k, v in sorted(allowed.items())}, "measured": recs}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("%d admissions in %d cases -> %s" % (len(recs), len(allowed), dst))


if __name__ == "__main__":
    main()
