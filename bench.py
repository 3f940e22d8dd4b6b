#!/usr/bin/env python3
"""bench.py - MPC horizon-steps/s of the fused closed-loop kernel on BASELINE config 3
(3-level transmon, n=9, m=2, T=40, n_steps=20, 65,536-member model ensemble per GPU).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--batch B]

ONE invocation for any N.  With --gpus N > 1 and no launcher in the environment (WORLD_SIZE unset) this process is only the
launcher: it never touches the GPU, starts N child processes of this same script - one rank per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_PORT (a free port) / M4Q_UID_FILE (where rank 0 publishes the RCCL unique id) set - relays rank 0's single
JSON line, and fails (stopping the siblings) when any rank fails or the deadline passes.  Under an external launcher
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`:
WORLD_SIZE set) the process is one of the ranks and --gpus must agree with WORLD_SIZE.  `n_gpus` in the line is the number of
ranks the RCCL communicator actually joined (an all-reduce of 1), not the flag.

One "step" = one complete receding-horizon run (all n_steps MPC steps, every SQP iteration, plant
propagation) of the rank's whole ensemble, inputs resident in HBM.  Unit of work = one MPC
horizon-step: value = sum over ranks, instances and MPC steps of qp_solves * T,
divided by the max-over-ranks time of the K timed steps.  The ensemble shards with no data-path
collective; one gather of the results closes each step when N > 1 (weak scaling: per-GPU batch fixed).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per horizon-step (SURVEY.md 8d / BASELINE.md 4), order 1: the COMPLEX recursion of lqr.py (complex MAC = 8 flop)
ALG_FLOP = {4: 3.5e3, 9: 27e3, 16: 126e3}
PEAK_F64_TFLOPS = 78.6      # MI355X fp64 dense rate (AMD spec; 256 CU x 4 SIMD x 16 FMA lanes x 2.4 GHz).  v_fma_f64 (used here, with
                            # DPP row broadcasts) and v_mfma_f64 run on the SAME pipe at the same rate: tools/ubench_mfma.hip measures
                            # 76-78 TFLOP/s for either and the SUM when both are issued (profiles/r02_ubench_mfma.txt)
PEAK_HBM_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md
PMC_JSON = os.path.join(ROOT, "profiles", "r05_pmc.json")


def mac_per_hstep(n, m, P):
    """Multiply-accumulates of one horizon index of one QP solve on n coordinates (linearise (2P + 2) n^2,
    Riccati 2 N^3 + 4 m N^2 with N = n + 1, forward n^2 + n m + m N)."""
    N = n + 1
    return (2 * P + 2) * n * n + 2 * N ** 3 + 4 * m * N * N + n * n + n * m + m * N


def executed_flop_per_hstep(n, m, P, path, targ_const=False):
    """Arithmetic the selected path EXECUTES per horizon-step: complex recursion on n = d*d coordinates (8 flop per complex MAC),
    the same recursion on real numbers in the Hermitian basis (2 flop per MAC), or on the n - 1 traceless coordinates.
    targ_const: the target is the same over the horizon window and the clipped real sweeps of a recursion of dimension >= 8 then
    skip the row form of A_t and the product A_t xbar (csrc/m4q_kernels.hip: M4Q_TC_MIN_N): (P + 1) n^2 MACs fewer."""
    if path == "complex":
        return 8.0 * mac_per_hstep(n, m, P)
    nr = n - 1 if path.startswith("traceless") else n
    macs = mac_per_hstep(nr, m, P)
    if targ_const and nr >= 8:
        macs -= (P + 1) * nr * nr
    return 2.0 * macs


def compulsory_bytes(B, n, m, P, T, ns, path):
    """HBM bytes a persistent launch cannot avoid (model and guess terms counted once per run, not per solve):
    per member, the model and initial state in, xs / us / SQP-guess checkpoint / codes / solve counts out."""
    sz = 8 if path == "real" else 16
    per = sz * n * n * (1 + P) + (16 + sz) * n + 16 * n * (ns + 1) + 8 * m * ns + 16 * n * (T + 1) + 8 * m * T + 8 + 4 * ns
    return B * per


def pmc_record(key, avg_launch_ms):
    """Counter record of this exact configuration from profiles/r05_pmc.json (tools/pmc_collect.py), or (None, why).  Refused when
    the launch it was taken on differs from the one measured now by more than 3 %: counters describe a binary, not a config."""
    try:
        rec = json.load(open(PMC_JSON))[key]
    except Exception:
        return None, "no counter record for %s in profiles/r05_pmc.json" % key
    # like for like: the HIP-event launch time bench.py itself measured in the record's traced run (rocprofv3's own kernel-trace
    # average of that run, 1 % higher, is kept beside it and must agree: tests/test_bench_logic.py)
    ref = rec.get("hip_event_launch_ms_same_run") or rec.get("traced_avg_launch_ms") or 0.0
    if not ref or abs(avg_launch_ms - ref) > 0.03 * ref:
        return None, "counter record for %s was taken at %.2f ms per launch, this run measures %.2f ms: stale, not reported" % (
            key, ref, avg_launch_ms)
    return rec, None


def usable_cores():
    """Host cores this job may actually use: the affinity mask, cut to the cgroup CPU quota when there is one (the GPU boxes show
    all 256 hardware threads in the mask but give a one-GPU job cpu.max = 16 cores; 256 workers on that share ran 30x slower each)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, (q + per // 2) // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_worker(config, lo, hi, budget):
    """One host core: the NumPy oracle (a port of the reference arithmetic) on members [lo, hi) of the ensemble, member by
    member, until the time budget is spent.  Runs in its own process (`bench.py --cpu-worker`), which never touches the GPU."""
    from mpc4quantum_amd import configs
    from oracle import m4q_oracle as orc
    q = configs.build(config, batch=hi - lo, offset=lo, total=max(hi, {1: 1, 2: 8192, 3: 65536, 4: 65536, 5: 2 ** 20}[config]),
                      host_models=False)
    units = done = 0
    spent = 0.0
    while done < hi - lo and (spent < budget or done < 1):
        if q["scales"] is not None:      # the member's model, built outside the timed part (setup, not the hot path)
            gens = [q["scales"][done, k] * q["generators"][k] for k in range(q["generators"].shape[0])]
        else:
            gens = list(q["generators"])
        mdl = orc.discretize_homogeneous(gens, q["dt"], q["order"])[None]
        t0 = time.perf_counter()
        _, _, _, solves = orc.mpc_batch(q["x0"][done:done + 1], mdl, q["dim_u"], q["order"], q["X_targ"], q["U_targ"], q["dt"],
                                        q["horizon"], q["n_steps"], q["plant_op0"], list(q["plant_ops"][0]), q["Q"], q["R"],
                                        q["Qf"], q["sat"], q["du"])
        spent += time.perf_counter() - t0
        units += int(solves.sum()) * q["horizon"]
        done += 1
    print(json.dumps({"units": units, "members": done, "seconds": spent}))


def cpu_baseline(config, p, cores, seconds_budget=20.0, members_per_core=64):
    """The oracle on `cores` host cores: one single-threaded child process each (`bench.py --cpu-worker ...`), each on its
    own slice of the first members of the same ensemble for ~seconds_budget.  Rate = all horizon-steps / longest worker."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = []
    for c in range(cores):
        lo, hi = c * members_per_core, min((c + 1) * members_per_core, p["x0"].shape[0])
        if lo >= hi:
            break
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(config), str(lo), str(hi),
                                       str(seconds_budget)], stdout=subprocess.PIPE, env=env, cwd=ROOT))
    results = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=seconds_budget * 6 + 120)
            results.append(json.loads(out.decode().strip().splitlines()[-1]))
        except Exception:                                # a worker that died or hung is left out of the sum
            pr.kill()
    if not results:
        return None
    units = sum(r["units"] for r in results)
    wall = max(r["seconds"] for r in results)
    return {"value": units / wall, "unit": "MPC horizon-steps/s", "cores": len(results), "kind": "port",
            "per_core": units / sum(r["seconds"] for r in results), "cpu": cpu_model(),
            "host_threads_visible": len(os.sched_getaffinity(0)), "host_cores_usable": usable_cores(),
            "sample": "%d of %d ensemble members (%d processes x ~%.0f s, one core each), full closed loop (n_steps=%d, T=%d), "
                      "NumPy oracle" % (sum(r["members"] for r in results), p["x0"].shape[0], len(results), wall, p["n_steps"],
                                        p["horizon"])}


def visible_devices():
    """GPUs a rank of this job could open, counted in a CHILD process (m4q_device_count there): the launcher itself must never
    initialise the GPU (a process that has, and then starts ranks, is what takes these hosts down)."""
    import subprocess
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--device-count"], stdout=subprocess.PIPE, timeout=300,
                             cwd=ROOT)
        return int(res.stdout.decode().strip().splitlines()[-1])
    except Exception:
        return -1


def launch(args):
    """--gpus N > 1 without an external launcher: this process starts the N ranks and never touches the GPU itself."""
    from mpc4quantum_amd.distributed import launch_local_ranks
    if not args.launch_check:
        have = visible_devices()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but this node shows %d usable GPU(s) (m4q_device_count in a child process "
                             "returned %d): refusing to start ranks that would share or miss a device\n" % (args.gpus, max(have, 0), have))
            return 2
    child = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    return launch_local_ranks(child, args.gpus, timeout=args.launch_timeout)


def launch_check(args, rank, world):
    """What a rank does under --launch-check: no device, no library.  The unique-id file exchange is the product's own
    (distributed.exchange_unique_id, with a random token in place of RCCL's id), then every rank leaves a
    `<uid file>.joined.<rank>` note holding the token it read and rank 0 counts the notes that carry ITS token - the
    file-system stand-in for the all-reduce of 1 that gives `n_gpus` in the real run."""
    from mpc4quantum_amd.distributed import exchange_unique_id
    if rank == args.fail_rank:
        sys.exit(3)
    token, path = exchange_unique_id(rank, world, timeout=60.0, make_id=lambda: os.urandom(128))
    env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "M4Q_UID_FILE")}
    note = "%s.joined.%d" % (path, rank)
    with open(note + ".tmp", "w") as f:
        json.dump({"token": token.hex(), "env": env, "pid": os.getpid()}, f)
    os.replace(note + ".tmp", note)
    if rank != 0:
        return 0
    ranks, t0 = {}, time.time()
    while len(ranks) < world and time.time() - t0 < args.launch_timeout:
        for r in range(world):
            if r not in ranks:
                try:
                    ranks[r] = json.load(open("%s.joined.%d" % (path, r)))
                except (OSError, ValueError):
                    pass
        time.sleep(0.01)
    joined = [r for r, v in sorted(ranks.items()) if v["token"] == token.hex()]
    for r in range(world):
        try:
            os.unlink("%s.joined.%d" % (path, r))
        except OSError:
            pass
    print(json.dumps({"launch_check": True, "n_gpus": len(joined), "ranks": [ranks[r]["env"] for r in joined],
                      "pids": [ranks[r]["pid"] for r in joined], "uid_file": path}))
    return 0 if len(joined) == world else 1


def main():
    if len(sys.argv) == 6 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
    if len(sys.argv) == 2 and sys.argv[1] == "--device-count":
        from mpc4quantum_amd import _lib
        print(_lib.device_count())
        return 0
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="ensemble members per GPU (default: the config's own size)")
    ap.add_argument("--order", type=int, default=1, help="order of the Taylor-truncated model (vectorize.py:8-49); 1 = the BASELINE configs'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0, help="host cores for the CPU baseline (0 = all this job may use: affinity "
                                                             "mask cut to the cgroup CPU quota)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="time budget of every CPU-baseline worker")
    ap.add_argument("--exact-qp", action="store_true", help="not the headline: every QP solved to the box-constrained optimum "
                                                            "(M4Q_QP_EXACT_BOX); roofline flops then count pinned sweeps")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: take the N > 1 code path (RCCL communicator, gather "
                                                              "buffers bound to the session, one gather per run) even with one rank")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="deadline (s) of the N ranks started by --gpus N")
    ap.add_argument("--launch-check", action="store_true", help="rehearsal of the launch without any device: every rank reports "
                                                                "its environment, the unique-id file is exchanged, rank 0 counts "
                                                                "the ranks that joined and prints the one JSON line")
    ap.add_argument("--fail-rank", type=int, default=-1, help="(with --launch-check) this rank exits 3 before joining: a test "
                                                              "that a dying rank fails the launch")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d: one of the two is wrong" % (args.gpus, world))
    if args.launch_check:
        return launch_check(args, rank, world)
    multi = world > 1 or args.force_dist
    comm = None
    dev_index = -1
    if multi:
        # one process per GPU; RCCL through the C ABI (m4q_comm_*): no PyTorch in the product path.  The launcher has set
        # RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT; the unique id travels through a file keyed by this launch.
        from mpc4quantum_amd.distributed import RcclComm
        dev_index = local_rank
        comm = RcclComm.from_env(device=dev_index)

    import numpy as np
    from mpc4quantum_amd import _lib, configs
    from mpc4quantum_amd.session import EnsembleSession

    # every rank takes its own slice [rank*B, (rank+1)*B) of ONE world-sized ensemble draw (weak scaling);
    # per-member models are built on the device from the config's generators and scales
    full = args.batch or {1: 1, 2: 8192, 3: 65536, 4: 65536, 5: 2 ** 20}[args.config]
    p = configs.build(args.config, batch=full, order=args.order, offset=rank * full, total=full * world, host_models=False)
    B, n, m, T, ns = p["batch"], p["dim_x"], p["dim_u"], p["horizon"], p["n_steps"]
    # The CPU baseline runs FIRST: its worker processes are started, and have ended, before this process loads the HIP library
    # or makes its first HIP call (they need nothing from the GPU run; a process that holds a GPU context starts no children).
    cpu_base = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(args.config, p, args.cpu_cores or usable_cores(), args.cpu_seconds)
    P = _lib.lib().m4q_library_size(p["order"], m)
    per_model = p["scales"] is not None

    sess = EnsembleSession(B, n, m, p["order"], T, ns, p["dt"], p["sat"], p["du"], model_per_instance=per_model,
                           target_cols=ns + T + 1, device=dev_index if multi else -1, exact_qp=args.exact_qp)
    shard = None
    if multi:
        # the product's multi-GPU path (mpc4quantum_amd/distributed.py): the session's outputs are bound into ONE library-owned
        # device buffer [final states | us | codes | steps done | solve counts | status] and one RCCL gather moves it, enqueued
        # behind the kernel on the communicator's stream.  Two such buffers alternate: the gather of run k travels while the
        # kernel of run k+1 computes (every run's gather is inside the timed region; the last one is waited for at the
        # closing fence).
        from mpc4quantum_amd.distributed import ShardedResults
        shard = [ShardedResults(sess, B, comm, dst=0, final_state_only=True, slot=i) for i in range(2)]
    if per_model:
        sess.build_models(p["dt"], p["generators"], p["scales"])
        models = None
    else:
        models = p["models"] if p["models"] is not None else configs.build(args.config, batch=1)["models"]
    sess.load_problem(models, p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"], p["plant_ops"])
    # (a target that does not move over the run: the clipped sweeps take their constant-target form; not the exact mode's)
    targ_const = (not args.exact_qp) and bool(np.all(p["X_targ"] == p["X_targ"][..., :1]))
    path = sess.path()                  # "real" / "complex": the dtype of the arithmetic
    detail = sess.path_detail()         # "complex" | "real" (d*d Hermitian coordinates) | "traceless" (d*d - 1) | "traceless-tile" | "traceless-sg"

    runs = [0]

    def one_step():
        if multi:
            sh = shard[runs[0] % 2]
            runs[0] += 1
            sh.wait()                                              # its previous gather (two runs ago) has landed
            sh.bind()
            sess.run(0, ns)
            sh.gather(wait=False)                                  # the one collective of the job
        else:
            sess.run(0, ns)

    def fence():
        # barrier + device drained on both sides of the timed region (the contract's barrier + synchronize)
        sess.sync()
        if multi:
            for sh in shard:
                sh.wait()
            comm.wait()
            comm.barrier()
            sess.sync()

    for _ in range(args.warmup):
        one_step()
    fence()
    sess.kernel_ms()                                               # drop warm-up launches from the event log
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = sess.kernel_ms()

    if os.environ.get("M4Q_PHASE_TRACE"):
        sess.qp_stats()              # development builds (-DM4Q_DEV_PHASE_CLOCK): prints the wavefront time per phase of the main loop
    res = sess.results()
    units_per_step = int(res["qp_solves"].astype(np.int64).sum()) * T
    ok = int((res["exit_codes"] == 0).sum())
    code_hist = [int((res["exit_codes"] == c).sum()) for c in range(4)]     # members by exit code (mpc.py:130): 0 done, 1 exit_condition,
                                                                            # 2 the exact solver gave up, 3 non-finite
    info = sess.info()
    if multi:
        elapsed = float(comm.allreduce([elapsed], "max")[0])               # MAX over ranks of the timed region
        tot = comm.allreduce([float(units_per_step), float(ok), 1.0] + [float(c) for c in code_hist], "sum")
        units_total = float(tot[0])
        ok_total = int(tot[1])
        joined = int(round(tot[2]))                                        # ranks the communicator really has
        code_hist = [int(round(v)) for v in tot[3:7]]
    else:
        units_total = float(units_per_step)
        ok_total = ok
        joined = 1

    if rank == 0:
        value = units_total * args.steps / elapsed
        avg_launch_s = kern_ms / max(launches, 1) / 1e3
        qp_stats = sess.qp_stats() if args.exact_qp else None
        # units of arithmetic per launch: horizon-steps; in the exact mode one pinned sweep + policy rollout over the horizon
        # is the arithmetic of one clipped solve
        hsteps = T * qp_stats[1] if qp_stats else units_per_step
        flops_exec = executed_flop_per_hstep(n, m, P, detail, targ_const) * hsteps
        flops_alg = ALG_FLOP[n] * hsteps
        cbytes = compulsory_bytes(B, n, m, P, T, ns, path)
        key = "config%d_B%d_%s_%s" % (args.config, B, path, "exact" if args.exact_qp else "clip")
        if detail == "real":
            key += "_real9"
        if detail == "traceless-tile":
            key += "_tile"
        if detail == "traceless-sg":
            key += "_sg"
        rec, why = pmc_record(key, 1e3 * avg_launch_s)
        traffic = issue = None
        if rec:
            c = rec["counters"]
            # FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for this kernel's 8-16 B per
            # lane row accesses as for wide streaming reads (tools/ubench_fetch.hip, profiles/r02_fetch_calibration.txt)
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                traffic = (rec.get("fetch_factor", 2.0) * c["FETCH_SIZE"] + rec.get("write_factor", 1.0) * c["WRITE_SIZE"]) * 1024.0
            if "SQ_INSTS_VALU_FMA_F64" in c and "GRBM_GUI_ACTIVE" in c:
                cycles = c["GRBM_GUI_ACTIVE"] / 8.0                       # shader cycles of the counted launch (sum over 8 XCDs)
                slots = 1024 * cycles / 4.0                               # fp64 FMA wave-instruction slots: 1024 SIMDs, 4 cycles each
                pmc_ms = sum(rec["pmc_pass_launch_ms"]) / len(rec["pmc_pass_launch_ms"])
                mfma = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)           # one per v_mfma_f64_4x4x4_4b_f64 (16.7 cycles of the same pipe)
                issue = {"fma_f64_wave_insts": c["SQ_INSTS_VALU_FMA_F64"], "frac_of_fma_issue_slots": c["SQ_INSTS_VALU_FMA_F64"] / slots,
                         "mfma_f64_wave_insts": mfma,
                         "frac_of_fp64_pipe_cycles": (4.0 * c["SQ_INSTS_VALU_FMA_F64"] + 16.7 * mfma) / (1024 * cycles),
                         "valu_busy": c["SQ_ACTIVE_INST_VALU"] / slots if "SQ_ACTIVE_INST_VALU" in c else None,
                         "fma_share_of_valu_insts": c["SQ_INSTS_VALU_FMA_F64"] / c["SQ_INSTS_VALU"] if "SQ_INSTS_VALU" in c else None,
                         "wave_time_waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in c else None,
                         "clock_ghz_under_load": cycles / (pmc_ms * 1e-3) / 1e9,
                         "note": "from profiles/r05_pmc.json (separate --pmc passes on this binary); all 64 lanes counted"}
        out = {
            "metric": "MPC horizon-steps/sec across batch (3-level transmon, T=40)" if args.config == 3
                      else "MPC horizon-steps/sec across batch (config %d)" % args.config,
            "value": value, "unit": "MPC horizon-steps/s", "n_gpus": joined, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if path == "real" else "c128", "data": "synthetic",
            "config": {"workload": "BASELINE config %d: d=%d (n=%d, m=%d), order %d, T=%d, n_steps=%d, %d ensemble members per GPU, "
                                   "per-instance models, full closed loop per step; %s arithmetic path (%s)" % (args.config, p["d"], n, m, p["order"], T, ns, B, path, detail),
                       "batch_per_gpu": B, "horizon": T, "n_steps": ns, "qp_solves_per_step": units_per_step // T,
                       "instances_ok": ok_total, "exit_codes": {str(c): code_hist[c] for c in range(4)}, "parallelism": "ensemble-sharded x%d, one gather" % joined,
                       "grid": info["grid"], "lds_bytes": info["lds_bytes"], "hbm_resident_bytes": info["hbm_bytes"],
                       **({"qp_mode": "exact box-constrained (active set on the Riccati factorisation)",
                           "exact_qp_stats": dict(zip(("qp_solves", "pinned_sweeps", "ratio_steps", "end_kkt", "end_precision",
                                                       "end_cap"), qp_stats))} if qp_stats else {})},
            "roofline": {"bound": "valu_f64", "achieved": flops_exec / avg_launch_s / 1e12, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops_exec / avg_launch_s / 1e12 / PEAK_F64_TFLOPS, "traffic": traffic,
                         "kernel": "mpc_kernel<%s, PLANT_HAMILTONIAN, %s%s>" % (
                             "double" if path == "real" else "cplx", "true" if args.exact_qp else "false",
                             ", TL, TILE" if detail == "traceless-tile" and not args.exact_qp else
                             ", TL, false, SG" if detail == "traceless-sg" else
                             ", TL" if detail in ("traceless", "traceless-tile") else ""),
                         "launches": launches, "avg_launch_ms": 1e3 * avg_launch_s,
                         "flop_per_horizon_step": executed_flop_per_hstep(n, m, P, detail, targ_const), "horizon_steps_per_launch": hsteps,
                         "note": "compute-bound kernel: intensity >> the fp64 machine balance, so the binding roof is the fp64 pipe "
                                 "(v_fma_f64 with DPP row broadcasts%s; fp64 MFMA and fp64 VALU share one pipe and one peak).  achieved = "
                                 "flops of the recursion the selected path performs (real path: a quarter of SURVEY 8d's "
                                 "complex-recursion count; products padded to 4 x 4 tiles are not counted twice) / HIP-event launch "
                                 "time" % (", and - the %s sweep - v_mfma_f64_4x4x4_4b_f64 tiles" % ("pinned" if args.exact_qp else "backward")
                                           if detail == "traceless-tile" else ""),
                         "algorithmic_equivalent": {"achieved": flops_alg / avg_launch_s / 1e12, "unit": "TFLOP/s",
                                                    "note": "SURVEY 8d complex-recursion flops / time: a speed-up-adjusted throughput, "
                                                            "NOT a fraction of any roof (exceeds the peak on the real path at d=4)"},
                         "issue": issue,
                         "hbm": {"achieved": cbytes / avg_launch_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": cbytes / avg_launch_s / 1e9 / PEAK_HBM_GBS, "compulsory_bytes": cbytes,
                                 "note": "compulsory bytes of the persistent launch (models, states and guesses once per run)"},
                         "traffic_note": (why + (" (N = %d ranks: the record is keyed by the per-GPU batch and was taken on one GPU "
                                                          "alone; frac above does not depend on it)" % joined if joined > 1 else "")) if why
                                         else "FETCH_SIZE/WRITE_SIZE of profiles/r05_pmc.json, taken on this binary (launch time "
                                                "within 3 %): L2-miss bytes per launch, mostly served by the Infinity Cache"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out))
    if multi:
        for sh in shard:
            sh.close()
    sess.close()
    if multi:
        comm.close()


if __name__ == "__main__":
    sys.exit(main() or 0)
