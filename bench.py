#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X.

A "step" is ONE GP log_probability evaluation with fresh hyper-parameters, i.e. the
optimiser / MCMC step every tinygp tutorial runs: assemble K = k(X,X) +
diag on the device, blocked Cholesky in place, forward triangular solve, reductions, scalar
back on the host.  X, the noise diagonal and the residual are resident in HBM before the
timed region starts.

  python bench.py --gpus N --steps K --warmup W [--workload c2|c1|c3|n<int>]

N > 1 (launched by torch.distributed.run, one rank per GPU): the path shards as REPLICAS --
each rank evaluates its own hyper-parameter point on its own GPU, no data-path collective
("scaling": "weak"); the only collectives are the timing barrier and the MAX over ranks.

Prints ONE JSON line on rank 0 (contract in the task statement) with the `roofline` of the
dominant kernel (fp64 MFMA trailing update, gemm_nt_kernel<double, 0>) measured live with
HIP events on the launching stream, and a `cpu_baseline` of the NumPy/SciPy oracle.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet: FP64 matrix = FP64 vector = 78.6 TFLOP/s
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table
# HBM bytes per trailing-update launch for workload c2 at the default nb_outer = 1024, from the
# committed PMC passes (profiles/r01_h_pmc_hbm_traffic_nb1024.md: rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE in separate runs, KB units, FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM"
# for wide coalesced reads): (2 x 55.090 GB + 12.634 GB) / 42 launches of gemm_nt_kernel<double,0>.
# PMC collection serialises kernels, so it cannot run inside the timed region; other
# configurations report null.
PMC_TRAFFIC_BYTES_PER_LAUNCH_C2 = (2 * 55.090e9 + 12.634e9) / 42
PMC_TRAFFIC_NB = 1024


def trailing_update_bytes(n_pad: int, nb: int, itemsize: int, first_small_tiles: int = 0,
                          first_split: int = 0):
    """(algorithmic bytes, launches) of one factorisation's 128x128-tile trailing-update launches.

    Mirrors the launch shapes of csrc/chol.hip (look-ahead: next panel's block column, then
    the rest).  Per launch: the lower-trapezoid entries of C are read and written once and the
    panel operand (m x k) is read once.  Block-column updates of at most `first_small_tiles`
    128x128 tiles run on the 64x64-tile kernel and are not part of the profiled kernel; a
    block-column update is issued in two k-ranges (`first_split` blocks early, the rest after
    the panel).
    """
    total, launches = 0, 0
    k0 = 0
    while k0 < n_pad:
        kb = min(nb, n_pad - k0)
        nxt = k0 + kb
        mt = n_pad - nxt
        if mt <= 0:
            break
        kbn = min(nb, mt)
        for which, (m, nn) in enumerate(((mt, kbn), (mt - kbn, mt - kbn))):
            if m <= 0:
                continue
            tiles = (m // 128) * (nn // 128) - (nn // 128) * (nn // 128 - 1) // 2
            if which == 0 and tiles <= first_small_tiles:
                continue
            entries = nn * m - nn * (nn - 1) // 2
            ks = [kb]
            if which == 0 and 0 < first_split < kb // 128:
                ks = [first_split * 128, kb - first_split * 128]
            for k in ks:
                total += itemsize * (2 * entries + m * k)
                launches += 1
        k0 = nxt
    return total, launches


def emit(obj):
    """Print the ONE JSON line last: flush C stdio first (RCCL's banner sits in libc's buffer)."""
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(obj), flush=True)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="c2", help="c1|c2|c3 (BASELINE configs) or n<int>[d<int>]")
    p.add_argument("--nb-outer", type=int, default=0, help="override the outer block (0 = default)")
    p.add_argument("--lookahead", type=int, default=-1)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-profile", action="store_true", help="disable per-launch HIP-event timing")
    p.add_argument("--stages", action="store_true", help="also print per-stage times to stderr")
    p.add_argument("--unfused", action="store_true", help="separate factor and solve passes")
    p.add_argument("--distributed", action="store_true",
                   help="ONE matrix, 1-D block-cyclic columns over the N GPUs with RCCL panel "
                        "broadcast (strong scaling; BASELINE config 4) instead of replicas")
    p.add_argument("--nb-dist", type=int, default=1024, help="block-column width of --distributed")
    return p.parse_args()


def workload_spec(name: str):
    from tinygp_amd import synthetic

    if name in synthetic.CONFIGS:
        c = dict(synthetic.CONFIGS[name])
        c["name"] = name
        return c
    if name.startswith("n"):
        body = name[1:]
        dtype = "float64"
        if body.endswith("f32"):
            body, dtype = body[:-3], "float32"
        d = 1
        if "d" in body:
            body, ds = body.split("d")
            d = int(ds)
        return dict(name=name, n=int(body), d=d, dtype=dtype, diag=0.01 if dtype == "float64" else 0.1,
                    kernel=("expsq" if d == 1 else "matern52") if dtype == "float64" else "sum")
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(spec, rank):
    """The oracle (SciPy/OpenBLAS LAPACK) on this box's host cores, on a bounded sample:
    the same workload at N_s <= 8192, extrapolated stage by stage (assembly and trsv ~ N^2,
    dpotrf ~ N^3) to the workload's N.  Reported, never the target."""
    import scipy.linalg as sla

    from oracle import tinygp_np as o
    from tinygp_amd import synthetic

    n = spec["n"]
    ns = min(n, 8192)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # OpenBLAS with every core of a 2-socket box is far from its best: pick the thread count
    # that factors a 4096 x 4096 probe fastest and use it for the sample (reported in `cores`).
    threads, limiter = cores, None
    try:
        from threadpoolctl import threadpool_limits

        rngp = np.random.default_rng(0)
        Bp = rngp.normal(size=(4096, 512))
        Kp = Bp @ Bp.T + 4096 * np.eye(4096)
        best = None
        for th in sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores}):
            with threadpool_limits(limits=th):
                tq = time.perf_counter()
                sla.cholesky(Kp, lower=True, check_finite=False)
                tq = time.perf_counter() - tq
            if best is None or tq < best[0]:
                best = (tq, th)
        threads = best[1]
        limiter = threadpool_limits(limits=threads)
    except Exception:
        pass
    X, y = synthetic.make_inputs(ns, spec["d"], spec["dtype"])
    kern = synthetic.config_kernel(o, spec["kernel"])
    t0 = time.perf_counter()
    K = kern(X, X)
    K[np.diag_indices(ns)] += spec["diag"]
    t1 = time.perf_counter()
    L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    t2 = time.perf_counter()
    alpha = sla.solve_triangular(L, y, lower=True, check_finite=False)
    ll = -0.5 * float(alpha @ alpha) - float(np.sum(np.log(np.diag(L)))) - 0.5 * ns * np.log(2 * np.pi)
    t3 = time.perf_counter()
    if limiter is not None:
        limiter.restore_original_limits()
    f2, f3 = (n / ns) ** 2, (n / ns) ** 3
    t_full = (t1 - t0) * f2 + (t2 - t1) * f3 + (t3 - t2) * f2
    return {
        "value": 1.0 / t_full, "unit": "evals/s", "cores": threads, "kind": "port",
        "sample": (f"oracle/tinygp_np.py (SciPy dpotrf/dtrtrs, OpenBLAS, {threads} threads = the fastest of "
                   f"8..{cores} on a 4096^2 dpotrf probe; box has {cores} cores) timed at "
                   f"N={ns}: assembly {t1 - t0:.2f}s potrf {t2 - t1:.2f}s solve+reduce {t3 - t2:.3f}s"
                   + ("" if ns == n else f"; extrapolated to N={n} (N^2 / N^3 per stage)")),
        "potrf_gflops": (ns**3 / 3) / (t2 - t1) / 1e9,
        "loglik_sample": ll,
    }


def run_distributed(args, spec, X, y, rank, local_rank, world, dist, torch):
    """Strong scaling: one log_probability of ONE N x N matrix spread over all ranks."""
    import torch.distributed as tdist

    from tinygp_amd import kernels, synthetic
    from tinygp_amd.distributed import BlockCyclicCholesky, HipBlockOps

    if dist is None:  # single rank still goes through RCCL (self-broadcast)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    n = spec["n"]
    dt = np.dtype(spec["dtype"])
    kern = synthetic.config_kernel(kernels, spec["kernel"])
    solver = BlockCyclicCholesky(kern, X, np.full(n, spec["diag"], dtype=dt), nb=args.nb_dist,
                                 ops=HipBlockOps(local_rank), dist=tdist)

    def one_step(step):
        u = ((step * 7) % 11 - 5) / 5.0
        solver.assemble(synthetic.config_kernel(kernels, spec["kernel"], amp=1.5 * (1 + 0.02 * u),
                                                scale=2.5 * (1 + 0.03 * u)))
        solver.factor()
        ll = solver.log_probability(y)
        if not np.isfinite(ll):
            raise SystemExit(f"numerical failure in the distributed bench step (info={solver.info})")
        return ll

    def barrier():
        torch.cuda.synchronize(); tdist.barrier(); torch.cuda.synchronize()

    for s_ in range(args.warmup):
        one_step(s_)
    barrier()
    t0 = time.perf_counter()
    for s_ in range(args.steps):
        one_step(args.warmup + s_)
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
    elapsed = float(t.item())
    result = None
    if rank == 0:
        flops = n**3 / 3.0
        result = ({
            "metric": f"GP log_probability evals/sec + Cholesky TFLOP/s ({'fp64' if dt == np.float64 else 'fp32'}), N={n:,}",
            "value": args.steps / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64" if dt == np.float64 else "f32",
            "data": "synthetic",
            "config": {"workload": f"{spec['name']}: {spec['kernel']} kernel, {spec['d']}-D X, N={n}, "
                                   "1-D block-cyclic column Cholesky, RCCL panel broadcast",
                       "n": n, "nb": args.nb_dist, "parallelism": f"block-cyclic columns x{world}"},
            "aggregate_cholesky_tflops": flops * args.steps / elapsed / 1e12,
            "roofline": None, "cpu_baseline": None})
    tdist.barrier()
    tdist.destroy_process_group()
    if result is not None:
        emit(result)


def main():
    args = parse_args()
    os.environ.setdefault("NCCL_DEBUG", "WARN")  # keep RCCL's version banner off stdout
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False "
                         "(tinygp_amd has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    from tinygp_amd import _ffi, kernels, noise, synthetic
    from tinygp_amd.solvers import DirectSolver

    spec = workload_spec(args.workload)
    n, d = spec["n"], spec["d"]
    dt = np.dtype(spec["dtype"])
    ctx = _ffi.Ctx(device=local_rank)
    if args.nb_outer:
        ctx.set_option("nb_outer", args.nb_outer)
    if args.lookahead >= 0:
        ctx.set_option("lookahead", args.lookahead)
    ctx.set_option("profile", 0 if args.no_profile else 1)
    nb_used = ctx.set_option("nb_outer", 512)
    ctx.set_option("nb_outer", nb_used)
    la_used = ctx.set_option("lookahead", 1)
    ctx.set_option("lookahead", la_used)
    fst_used = ctx.set_option("first_small_tiles", 0)
    ctx.set_option("first_small_tiles", fst_used)
    fsp_used = ctx.set_option("first_split", 0)
    ctx.set_option("first_split", fsp_used)

    X, y = synthetic.make_inputs(n, d, spec["dtype"])

    if args.distributed:
        return run_distributed(args, spec, X, y, rank, local_rank, world, dist, torch)

    def kernel_at(step):
        # a different hyper-parameter point per step and per rank (replicas), like an
        # optimiser trajectory around the config's values (amp 1.5, scale 2.5)
        u = ((step * 7 + rank * 3) % 11 - 5) / 5.0
        return synthetic.config_kernel(kernels, spec["kernel"], amp=1.5 * (1 + 0.02 * u),
                                       scale=2.5 * (1 + 0.03 * u))

    # resident inputs: X + noise diagonal uploaded by the solver, residual uploaded once
    solver = DirectSolver(kernel_at(-1), X, noise.Diagonal(np.full(n, spec["diag"], dtype=dt)), ctx=ctx)
    import ctypes as C

    solver.set_residual(y)

    def one_step(step):
        if args.unfused:  # factor, then a separate triangular-solve pass
            solver.refactor(kernel_at(step))
            out = C.c_double()
            _ffi.check(_ffi.lib().tgp_solver_logprob(solver._handle, None, C.byref(out)), "tgp_solver_logprob")
            ll = out.value
        else:  # one fused device pass (what GaussianProcess.log_probability runs)
            ll = solver.factor_log_probability(None, kernel_at(step))
        if solver.info != 0 or not np.isfinite(ll):
            raise SystemExit(f"numerical failure in the bench step (info={solver.info}, ll={ll})")
        return ll

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        one_step(s)

    acc = {"assembly_ms": 0.0, "potrf_ms": 0.0, "syrk_ms": 0.0, "syrk_launches": 0.0, "trsv_ms": 0.0,
           "syrk_flops": 0.0}
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(args.warmup + s)
        if not args.no_profile:
            ms = (C.c_double * 8)()
            _ffi.lib().tgp_solver_timings(solver._handle, ms, 8)
            acc["assembly_ms"] += ms[0]; acc["potrf_ms"] += ms[1]; acc["syrk_ms"] += ms[2]
            acc["syrk_launches"] += ms[3]; acc["trsv_ms"] += ms[4]; acc["syrk_flops"] += ms[6]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        peak = FP64_MFMA_PEAK_TFLOPS if dt == np.float64 else FP32_MFMA_PEAK_TFLOPS
        roofline = None
        extra = {}
        if not args.no_profile and acc["syrk_ms"] > 0:
            n_pad = -(-n // 128) * 128
            alg_bytes, alg_launches = trailing_update_bytes(n_pad, int(nb_used), np.dtype(dt).itemsize,
                                                             int(fst_used), int(fsp_used))
            achieved = acc["syrk_flops"] / (acc["syrk_ms"] * 1e-3) / 1e12
            launches = max(acc["syrk_launches"], 1.0)
            roofline = {
                "kernel": f"gemm_nt_kernel<{'double' if dt == np.float64 else 'float'}, 0> (Cholesky trailing update)",
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak,
                "traffic": (PMC_TRAFFIC_BYTES_PER_LAUNCH_C2
                            if (args.workload == "c2" and nb_used == PMC_TRAFFIC_NB and world == 1
                                and la_used == 1 and fst_used == 1100 and fsp_used == 5) else None),
                "traffic_unit": "bytes/launch (PMC, profiles/r01_h_pmc_hbm_traffic_nb1024.md)",
                "algorithmic_bytes_per_launch": alg_bytes / max(alg_launches, 1),
                "avg_launch_ms": acc["syrk_ms"] / launches,
                "flops_per_launch": acc["syrk_flops"] / launches,
                "launches_per_step": launches / args.steps,
            }
            potrf_tf = (n**3 / 3.0) / (acc["potrf_ms"] / args.steps * 1e-3) / 1e12
            extra = {"cholesky_tflops": potrf_tf,
                     "stage_ms": {"assembly": acc["assembly_ms"] / args.steps,
                                  "potrf": acc["potrf_ms"] / args.steps,
                                  "trailing_update_kernels": acc["syrk_ms"] / args.steps,
                                  "trsv+reduce": acc["trsv_ms"] / args.steps}}
            if args.stages:
                print(json.dumps(extra, indent=1), file=sys.stderr)
        out = {
            "metric": f"GP log_probability evals/sec + Cholesky TFLOP/s ({'fp64' if dt == np.float64 else 'fp32'}), N={n:,}",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if dt == np.float64 else "f32", "data": "synthetic",
            "config": {"workload": (f"{spec['name']}: {spec['kernel']} kernel, {d}-D X, N={n}, "
                                    f"{'fp64' if dt == np.float64 else 'fp32'}, dense Cholesky + tri-solve"),
                       "n": n, "d": d, "diag": spec["diag"],
                       "parallelism": f"replicas x{world} (one evaluation stream per GPU, no data-path collective)" if world > 1 else "single",
                       "nb_outer": int(nb_used)},
            "roofline": roofline,
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, rank)
        else:
            out["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out)


if __name__ == "__main__":
    main()
