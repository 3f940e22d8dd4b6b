"""Host-side logic of bench.py that needs no GPU: the roofline inputs, the staleness rule for counter records, the CPU share."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("m4q_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_executed_flops_and_compulsory_bytes():
    b = _bench()
    # ~3.4 k complex MACs = 27 kflop per horizon-step for the complex recursion at d=3 (order 1: P = 2); the real
    # recursion executes a quarter of that, the traceless one (8 coordinates) 5.2 kflop
    assert b.mac_per_hstep(9, 2, 2) == 486 + 2800 + 119
    assert abs(b.executed_flop_per_hstep(9, 2, 2, "complex") - 27e3) < 0.3e3
    assert b.executed_flop_per_hstep(9, 2, 2, "real") == b.executed_flop_per_hstep(9, 2, 2, "complex") / 4
    assert b.executed_flop_per_hstep(9, 2, 2, "traceless") == 2.0 * b.mac_per_hstep(8, 2, 2) == 5176.0
    assert b.executed_flop_per_hstep(9, 2, 2, "traceless-tile") == 5176.0
    # constant target: the clipped real sweeps of a recursion of dimension >= 8 skip the row form of A_t and A_t xbar
    assert b.executed_flop_per_hstep(9, 2, 2, "traceless", targ_const=True) == 5176.0 - 2 * 3 * 64
    assert b.executed_flop_per_hstep(4, 1, 1, "traceless", targ_const=True) == b.executed_flop_per_hstep(4, 1, 1, "traceless")
    assert abs(b.executed_flop_per_hstep(16, 3, 3, "complex") - 126e3) < 2e3
    # config 3, real path: model 8*9*27, x0 (16+8)*9, xs 16*9*21, us 8*2*20, guess 16*9*41 + 8*2*40, codes and counts 8 + 80
    per = 8 * 9 * 27 + 24 * 9 + 16 * 9 * 21 + 8 * 2 * 20 + 16 * 9 * 41 + 8 * 2 * 40 + 8 + 80
    assert b.compulsory_bytes(65536, 9, 2, 2, 40, 20, "real") == 65536 * per
    assert b.compulsory_bytes(1, 9, 2, 2, 40, 20, "complex") - b.compulsory_bytes(1, 9, 2, 2, 40, 20, "real") == 8 * 9 * 27 + 8 * 9


def test_counter_records_are_refused_when_stale(tmp_path, monkeypatch):
    b = _bench()
    rec = {"config3_B65536_real_clip": {"traced_avg_launch_ms": 50.6, "hip_event_launch_ms_same_run": 50.0,
                                        "counters": {"FETCH_SIZE": 1.0}}}
    path = tmp_path / "pmc.json"
    path.write_text(json.dumps(rec))
    monkeypatch.setattr(b, "PMC_JSON", str(path))
    got, why = b.pmc_record("config3_B65536_real_clip", 50.9)          # within 3 %
    assert got is not None and why is None
    got, why = b.pmc_record("config3_B65536_real_clip", 52.0)          # 4 % off: the record describes another binary
    assert got is None and "stale" in why
    got, why = b.pmc_record("config4_B65536_real_clip", 85.0)
    assert got is None and "no counter record" in why


def test_committed_counter_records_cover_every_baseline_config():
    recs = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc.json")))
    # (d = 2, 3 run their backward sweep on matrix-core tiles: key suffix _tile, MFMA instructions counted; d = 4, the complex path,
    #  and the DPP sweeps kept for comparison execute none; the exact mode at d = 3 runs its pinned sweep on tiles)
    for key, tile in (("config2_B8192_real_clip_tile", True), ("config3_B65536_real_clip_tile", True), ("config3_B65536_real_clip", False),
                      ("config3_B65536_complex_clip", False), ("config3_B65536_real_exact_tile", True), ("config4_B65536_real_clip", False),
                      ("config4_B65536_real_clip_sg", False), ("config5_B131072_real_clip_tile", True)):
        r = recs[key]
        assert r["traced_avg_launch_ms"] > 0 and r["counters"]["SQ_INSTS_VALU_FMA_F64"] > 0
        assert (r["counters"]["SQ_INSTS_VALU_MFMA_MOPS_F64"] > 0) == tile
        # the HIP-event time of bench.py and rocprofv3's kernel-trace average of the same run agree
        assert abs(r["hip_event_launch_ms_same_run"] - r["traced_avg_launch_ms"]) <= 0.02 * r["traced_avg_launch_ms"]
    # the headline's record: v_mfma_f64_4x4x4_4b_f64 at 44 per wavefront and horizon index (4 members), and fewer vector instructions than
    # the DPP sweeps need for the same work
    t, d = recs["config3_B65536_real_clip_tile"], recs["config3_B65536_real_clip"]
    per_index = t["counters"]["SQ_INSTS_VALU_MFMA_MOPS_F64"] / (t["horizon_steps_per_launch"] / 4.0)
    assert 40 <= per_index <= 48
    assert t["counters"]["SQ_INSTS_VALU"] < 0.65 * d["counters"]["SQ_INSTS_VALU"]
    assert t["traced_avg_launch_ms"] < 0.95 * d["traced_avg_launch_ms"]
    # d = 4: the shared-generator kernel (two wavefronts per SIMD) against the per-member-model kernel of the same configuration
    g, m = recs["config4_B65536_real_clip_sg"], recs["config4_B65536_real_clip"]
    assert g["counters"]["SQ_WAVES"] == 2 * m["counters"]["SQ_WAVES"] and g["traced_avg_launch_ms"] < 0.95 * m["traced_avg_launch_ms"]


def test_usable_cores_is_positive_and_bounded():
    b = _bench()
    n = b.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))


# ---- the launcher: `bench.py --gpus N` starts N ranks itself ---------------------------------------------
def _run_bench(*argv, env=None, timeout=120):
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "M4Q_UID_FILE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, env=e, timeout=timeout, cwd=ROOT)


def test_gpus_flag_starts_that_many_ranks_and_prints_one_line():
    """No device: every rank reports its environment, the unique-id file travels from rank 0 to the others, rank 0 counts the
    ranks that joined.  One JSON line on stdout, n_gpus = ranks that joined."""
    res = _run_bench("--gpus", "2", "--launch-check")
    assert res.returncode == 0, res.stderr.decode()
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert [r["RANK"] for r in out["ranks"]] == ["0", "1"] and [r["LOCAL_RANK"] for r in out["ranks"]] == ["0", "1"]
    assert all(r["WORLD_SIZE"] == "2" and r["MASTER_ADDR"] == "127.0.0.1" for r in out["ranks"])
    assert len({r["MASTER_PORT"] for r in out["ranks"]}) == 1 and len({r["M4Q_UID_FILE"] for r in out["ranks"]}) == 1
    assert len(set(out["pids"])) == 2                      # two processes, neither of them the launcher
    assert not os.path.exists(out["uid_file"])             # the launcher cleans up after the ranks
    res = _run_bench("--gpus", "4", "--launch-check")
    assert json.loads(res.stdout.decode().strip())["n_gpus"] == 4


def test_gpus_8_launch_rehearsal():
    """The driver's own N: eight ranks, one unique-id file with eight readers, one line, LOCAL_RANK 0..7 (= the device each rank
    opens), nothing left behind."""
    res = _run_bench("--gpus", "8", "--launch-check")
    assert res.returncode == 0, res.stderr.decode()
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and len(set(out["pids"])) == 8
    assert [r["LOCAL_RANK"] for r in out["ranks"]] == [str(i) for i in range(8)]
    assert all(r["WORLD_SIZE"] == "8" for r in out["ranks"]) and len({r["M4Q_UID_FILE"] for r in out["ranks"]}) == 1
    assert not os.path.exists(out["uid_file"])


def test_a_rank_that_dies_fails_the_launch_and_stops_its_siblings():
    import time
    t0 = time.time()
    res = _run_bench("--gpus", "3", "--launch-check", "--fail-rank", "1", "--launch-timeout", "60")
    assert res.returncode == 3                             # the dead rank's code, not a hang until rank 0's own deadline
    assert time.time() - t0 < 30
    assert res.stdout.decode().strip() == ""               # no result line from a launch that lost a rank
    assert "rank 1 exited with code 3" in res.stderr.decode()


def test_launcher_deadline_stops_ranks_that_never_finish():
    import io
    import subprocess
    import sys
    from mpc4quantum_amd.distributed import launch_local_ranks
    err = io.StringIO()
    marker = "m4q_launch_deadline_probe_%d" % os.getpid()
    rc = launch_local_ranks([sys.executable, "-c", "import time; time.sleep(120)  # " + marker], 2, timeout=1.0, out=io.StringIO(),
                            err=subprocess.DEVNULL)
    assert rc == 124
    left = subprocess.run(["ps", "-eo", "args"], stdout=subprocess.PIPE).stdout.decode()
    assert marker not in left


def test_gpus_flag_is_refused_when_it_disagrees_with_the_node_or_the_launcher():
    # more ranks than GPUs (this container has none): refused before any rank starts, by a count taken in a child process
    res = _run_bench("--gpus", "2")
    assert res.returncode == 2 and b"usable GPU" in res.stderr and res.stdout.strip() == b""
    # under an external launcher the flag must agree with WORLD_SIZE
    res = _run_bench("--gpus", "4", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and b"WORLD_SIZE=2" in res.stderr


def test_launcher_never_loads_the_hip_library():
    """The launcher process must not touch the GPU: it may not even load libm4q_hip.so (static initialisers of a HIP library run
    at dlopen).  Checked on the module level: bench.launch() and distributed.launch_local_ranks reach no _lib.lib() call."""
    import ast
    src = open(os.path.join(ROOT, "mpc4quantum_amd", "distributed.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "launch_local_ranks"][0]
    names = {n.attr for n in ast.walk(fn) if isinstance(n, ast.Attribute)} | {n.id for n in ast.walk(fn) if isinstance(n, ast.Name)}
    assert "_lib" not in names and "lib" not in names


def test_committed_parity_admissions_never_cover_the_headline():
    """profiles/r04_parity_admissions.json is what the GPU parity tests allow to pass on the oracle-sensitivity clause.  It must hold
    no entry for config 3 order 1 (the headline), config 1 or config 2 on any arithmetic path, and none for an exact-mode case."""
    import re
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_parity_admissions.json")))
    for case, steps in d["allowed"].items():
        m = re.match(r"stepwise\[cfg(\d\w?)-o(\d)-B\d+-T(\d+)-(\w+)\]$", case)
        assert m, case
        cfg, order = m.group(1), int(m.group(2))
        # the headline, configs 1 and 2, and the two strict T = 80 cases (config 5 made well conditioned; config 5 at order 2)
        assert (cfg, order) not in (("1", 1), ("1", 2), ("2", 1), ("3", 1), ("5w", 1), ("5", 2)), case
        assert steps == sorted(set(steps)) and all(0 <= s < 20 for s in steps)
    # every allowed step is backed by a measured record with its errors and the oracle's own sensitivity
    seen = {(r["case"], r["step"]) for r in d["measured"]}
    assert seen == {(c, s) for c, v in d["allowed"].items() for s in v}
    for r in d["measured"]:
        for e, s_k, tol in zip(r["errs"], r["oracle_sensitivity"], r["fixed_bounds"]):
            assert e <= tol + 10 * s_k


def test_parity_admissions_tool_refuses_to_replace_a_fuller_record(tmp_path):
    """tools/parity_admissions.py: a recording that holds fewer admissions than the committed record, or that did not execute a
    committed case, is refused without --merge / --force (round 4 lost a measured record to a later partial run)."""
    src = tmp_path / "measured.json"
    src.write_text(json.dumps({"admissions": [], "cases_run": ["stepwise[cfg1-o1-B1-T10-real]"]}))
    before = open(os.path.join(ROOT, "profiles", "r04_parity_admissions.json")).read()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_admissions.py"), str(src)], capture_output=True, text=True)
    assert res.returncode != 0 and "refused" in res.stderr
    assert open(os.path.join(ROOT, "profiles", "r04_parity_admissions.json")).read() == before
