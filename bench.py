#!/usr/bin/env python3
"""bench.py - MPC horizon-steps/s of the fused closed-loop kernel on BASELINE config 3
(3-level transmon, n=9, m=2, T=40, n_steps=20, 65,536-member model ensemble per GPU).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

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

# algorithmic work per horizon-step (SURVEY.md 8d / BASELINE.md 4), order 1
ALG_FLOP = {4: 3.5e3, 9: 27e3, 16: 126e3}
PEAK_F64_TFLOPS = 78.6      # MI355X fp64 vector == fp64 matrix dense rate (AMD spec; 256 CU x 4 SIMD x 16 FMA lanes x 2.4 GHz;
                            # tools/ubench_dpp.hip measures 78.0 with v_fmac_f64_dpp)
PEAK_HBM_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md


def alg_bytes_per_hstep(n, m, P, T):
    """model + X_guess in + U_guess in + X_opt, U_opt out, per QP solve, divided by T."""
    return (16 * n * n * (1 + P) + 16 * n * (T + 1) + 8 * m * T + 16 * n * (T + 1) + 8 * m * T) / T


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
            "per_core": units / sum(r["seconds"] for r in results),
            "sample": "%d of %d ensemble members (%d processes x ~%.0f s, one core each), full closed loop (n_steps=%d, T=%d), "
                      "NumPy oracle" % (sum(r["members"] for r in results), p["x0"].shape[0], len(results), wall, p["n_steps"],
                                        p["horizon"])}


def main():
    if len(sys.argv) == 6 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="ensemble members per GPU (default: the config's own size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0, help="host cores for the CPU baseline (0 = all this process may use, "
                                                             "at most 16)")
    ap.add_argument("--backend", default="nccl", help="collective backend for N > 1: nccl (= RCCL over xGMI, the real thing) or "
                                                      "gloo (rehearsal on a box with fewer GPUs than ranks: results staged through the host)")
    ap.add_argument("--exact-qp", action="store_true", help="not the headline: every QP solved to the box-constrained optimum "
                                                            "(M4Q_QP_EXACT_BOX); roofline flops then count pinned sweeps")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: take the N > 1 code path (process group, torch-owned "
                                                              "output buffers, gather) even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend="gloo")

    import numpy as np
    from mpc4quantum_amd import _lib, configs
    from mpc4quantum_amd.session import EnsembleSession

    # every rank takes its own slice [rank*B, (rank+1)*B) of ONE world-sized ensemble draw (weak scaling);
    # per-member models are built on the device from the config's generators and scales
    full = args.batch or {1: 1, 2: 8192, 3: 65536, 4: 65536, 5: 2 ** 20}[args.config]
    p = configs.build(args.config, batch=full, offset=rank * full, total=full * world, host_models=False)
    B, n, m, T, ns = p["batch"], p["dim_x"], p["dim_u"], p["horizon"], p["n_steps"]
    P = _lib.lib().m4q_library_size(p["order"], m)
    per_model = p["scales"] is not None

    sess = EnsembleSession(B, n, m, p["order"], T, ns, p["dt"], p["sat"], p["du"], model_per_instance=per_model,
                           target_cols=ns + T + 1, device=dev_index if multi else -1, exact_qp=args.exact_qp)
    gather_bufs = None
    if multi:
        # results live in torch-owned HBM so RCCL can gather them without a copy
        us_t = torch.empty(B * ns * m, dtype=torch.float64, device="cuda")
        xs_t = torch.empty(B * (ns + 1) * n * 2, dtype=torch.float64, device="cuda")
        sess.bind_output(_lib.F_US, us_t.data_ptr(), us_t.numel() * 8)
        sess.bind_output(_lib.F_XS, xs_t.data_ptr(), xs_t.numel() * 8)
        us_all = torch.empty(world * us_t.numel(), dtype=torch.float64, device="cuda") if rank == 0 else None
        gather_bufs = (us_t, us_all)
    if per_model:
        sess.build_models(p["dt"], p["generators"], p["scales"])
        models = None
    else:
        models = p["models"] if p["models"] is not None else configs.build(args.config, batch=1)["models"]
    sess.load_problem(models, p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"], p["plant_ops"])
    path = sess.path()

    def one_step():
        sess.run(0, ns)
        if multi:
            sess.sync()
            us_t, us_all = gather_bufs
            xs_final = xs_t.view(B, ns + 1, n * 2)[:, -1, :].contiguous()
            if args.backend != "nccl":                             # rehearsal: gloo moves host tensors
                xs_final, us_src = xs_final.cpu(), us_t.cpu()
                outs = [torch.empty_like(xs_final) for _ in range(world)] if rank == 0 else None
                dist.gather(xs_final, outs, dst=0)
                chunks = [torch.empty_like(us_src) for _ in range(world)] if rank == 0 else None
                dist.gather(us_src, chunks, dst=0)
                return
            outs = [torch.empty_like(xs_final) for _ in range(world)] if rank == 0 else None
            dist.gather(xs_final, outs, dst=0)                     # the one RCCL collective of the job
            chunks = list(us_all.chunk(world)) if rank == 0 else None
            dist.gather(us_t, chunks, dst=0)

    def fence():
        sess.sync()
        if multi:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

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

    res = sess.results()
    units_per_step = int(res["qp_solves"].astype(np.int64).sum()) * T
    ok = int((res["exit_codes"] == 0).sum())
    info = sess.info()
    if multi:
        t = torch.tensor([elapsed, float(units_per_step), float(ok)], dtype=torch.float64,
                         device="cuda" if args.backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        units_total = float(t[1])
        ok_total = int(t[2])
    else:
        units_total = float(units_per_step)
        ok_total = ok

    if rank == 0:
        value = units_total * args.steps / elapsed
        avg_launch_s = kern_ms / max(launches, 1) / 1e3
        flops = ALG_FLOP[n] * units_per_step
        qp_stats = sess.qp_stats() if args.exact_qp else None
        if qp_stats:
            # one pinned sweep + policy rollout over the horizon is the arithmetic of one clipped solve
            flops = ALG_FLOP[n] * T * qp_stats[1]
        abytes = alg_bytes_per_hstep(n, m, P, T) * units_per_step
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("config%d_B%d_%s" % (args.config, B, path))
            except Exception:
                traffic = None
        out = {
            "metric": "MPC horizon-steps/sec across batch (3-level transmon, T=40)" if args.config == 3
                      else "MPC horizon-steps/sec across batch (config %d)" % args.config,
            "value": value, "unit": "MPC horizon-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if path == "real" else "c128", "data": "synthetic",
            "config": {"workload": "BASELINE config %d: d=%d (n=%d, m=%d), order %d, T=%d, n_steps=%d, %d ensemble members per GPU, "
                                   "per-instance models, full closed loop per step; %s arithmetic path" % (args.config, p["d"], n, m, p["order"], T, ns, B, path),
                       "batch_per_gpu": B, "horizon": T, "n_steps": ns, "qp_solves_per_step": units_per_step // T,
                       "instances_ok": ok_total, "parallelism": "ensemble-sharded x%d, one gather" % world,
                       "grid": info["grid"], "lds_bytes": info["lds_bytes"], "hbm_resident_bytes": info["hbm_bytes"],
                       **({"qp_mode": "exact box-constrained (active set on the Riccati factorisation)",
                           "exact_qp_stats": dict(zip(("qp_solves", "pinned_sweeps", "ratio_steps", "end_kkt", "end_precision",
                                                       "end_cap"), qp_stats))} if qp_stats else {})},
            "roofline": {"bound": "mfma", "achieved": flops / avg_launch_s / 1e12, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / avg_launch_s / 1e12 / PEAK_F64_TFLOPS, "traffic": traffic,
                         "kernel": "mpc_kernel<%s, PLANT_HAMILTONIAN, %s>" % ("double" if path == "real" else "cplx",
                                                                              "true" if args.exact_qp else "false"), "launches": launches, "avg_launch_ms": 1e3 * avg_launch_s,
                         "note": "fp64 compute roof: v_fma_f64 (VALU, used here with DPP row broadcasts) and v_mfma_f64 share the "
                                 "78.6 TFLOP/s dense rate on MI355X; achieved = ALGORITHMIC flops (SURVEY 8d, complex recursion) / launch time. "
                                 "The real path executes a quarter of them: its executed-FMA issue rate is 47% of peak, the complex path's 72% "
                                 "(PMC, DESIGN.md section 5)",
                         "hbm": {"achieved": abytes / avg_launch_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": abytes / avg_launch_s / 1e9 / PEAK_HBM_GBS}},
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = args.cpu_cores or min(16, len(os.sched_getaffinity(0)))
            out["cpu_baseline"] = cpu_baseline(args.config, p, cores)
        print(json.dumps(out))
    sess.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
