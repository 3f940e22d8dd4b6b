"""Host-side logic of bench.py that needs no GPU: the roofline inputs, the staleness rule for counter records, the CPU share."""
import importlib.util
import json
import os

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
    recs = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc.json")))
    for key in ("config2_B8192_real_clip", "config3_B65536_real_clip", "config3_B65536_complex_clip", "config3_B65536_real_exact",
                "config4_B65536_real_clip", "config5_B131072_real_clip"):
        r = recs[key]
        assert r["traced_avg_launch_ms"] > 0 and r["counters"]["SQ_INSTS_VALU_FMA_F64"] > 0
        assert r["counters"]["SQ_INSTS_VALU_MFMA_MOPS_F64"] == 0            # no MFMA instruction in any of the default kernels
        # the HIP-event time of bench.py and rocprofv3's kernel-trace average of the same run agree
        assert abs(r["hip_event_launch_ms_same_run"] - r["traced_avg_launch_ms"]) <= 0.02 * r["traced_avg_launch_ms"]


def test_usable_cores_is_positive_and_bounded():
    b = _bench()
    n = b.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))
