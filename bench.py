#!/usr/bin/env python3
"""bench.py -- the hot path of SubspaceInference.jl on MI355X at BASELINE.json's cfg2.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE posterior sample: propose z' -> W_swa + P z' -> Dense-chain forward over the full data in fp64
-> Gaussian log-likelihood -> Metropolis accept (reference src/space_inference.jl:90-95,111-116), all on the
device with inputs resident in HBM.  Each GPU runs its own independent chain (weak scaling, no data-path
collective); `value` = samples of all ranks / max-over-ranks time.  The subspace-construct wall-clock (the
metric's second half: K=100 SWA/deviation pushes from device-resident snapshots + Gram + eigensolve +
projection) is measured before sampling and reported as `construct_wall_ms`.

One JSON line on stdout (rank 0).  `roofline` is computed live from the library's own hipEvent pairs around
the dominant kernel (the 960x960 Dense layer GEMM on the fp64 matrix cores) inside the timed region;
`cpu_baseline` times the NumPy/OpenBLAS oracle on the same workload on this box's host cores (bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "posterior samples/sec (whole node) + subspace-construct wall-clock, 1M-param MLP M=20"
DIMS, ACTS = [128, 960, 960, 1], [1, 1, 0]  # Chain(Dense(128,960,relu), Dense(960,960,relu), Dense(960,1))
B, M, K_SNAP = 100000, 20, 100
SIGMA_Z, SIGMA_M = 0.1, 1.0
PEAK_F64_TFLOPS = 78.6   # MI355X fp64 matrix peak (datasheet; the guide's table has no f64 row -- DESIGN.md)
PEAK_HBM_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def layer_table():
    table, off = [], 0
    for fin, fout, act in zip(DIMS[:-1], DIMS[1:], ACTS):
        table.append((fin, fout, act, off, off + fin * fout))
        off += fin * fout + fout
    return table, off


def glorot_flat(seed):
    rng = np.random.default_rng(seed)
    parts = []
    for fin, fout in zip(DIMS[:-1], DIMS[1:]):
        w = ((rng.random((fout, fin)) - 0.5) * np.sqrt(24.0 / (fin + fout))).astype(np.float32)
        parts += [w.reshape(-1, order="F"), np.zeros(fout, dtype=np.float32)]
    return np.concatenate(parts)


def cpu_baseline(table, w_swa, p, x, y, z0, budget_s):
    """The NumPy/OpenBLAS restatement (oracle) timed on the host: bounded sample of the same workload."""
    from oracle import subspace_oracle as so
    blas = "unknown"
    try:
        from threadpoolctl import threadpool_info
        info = threadpool_info()
        cores = max([i.get("num_threads", 1) for i in info] or [os.cpu_count() or 1])
        blas = ", ".join(sorted({"%s %s" % (i.get("internal_api", "?"), i.get("version", "?")) for i in info
                                 if i.get("user_api") == "blas"})) or blas
    except Exception:
        cores = os.cpu_count() or 1
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    t0 = time.perf_counter()
    lp0 = so.logdensity(table, w_swa, p, x, y, SIGMA_M, z0)
    t1 = time.perf_counter() - t0
    n = int(max(1, min(50, (budget_s - t1) // max(t1, 1e-3))))
    rng = np.random.default_rng(123)
    t0 = time.perf_counter()
    for _ in range(n):
        so.logdensity(table, w_swa, p, x, y, SIGMA_M, z0 + SIGMA_Z * rng.standard_normal(M))
    dt = time.perf_counter() - t0
    single = None
    try:  # the same evaluation on ONE host thread (BASELINE.md asks for both figures); one evaluation, ~5-10 s
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            t0 = time.perf_counter()
            so.logdensity(table, w_swa, p, x, y, SIGMA_M, z0)
            single = 1.0 / (time.perf_counter() - t0)
    except Exception:
        pass
    return {"value": n / dt, "unit": "samples/s", "cores": int(cores), "kind": "port", "single_core_value": single, "cpu_model": cpu_model,
            "host_cpus": os.cpu_count(), "blas": blas,
            "sample": "%d density evaluations (W_swa+P*z, fp64 forward over X 128x%d, SSE) of the cfg2 workload with "
                      "NumPy+OpenBLAS (oracle/subspace_oracle.py), %.1f s" % (n, B, dt)}, lp0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU-baseline work (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # Only the JSON line may reach stdout: RCCL prints a version banner to fd 1 when the communicator is created.
    # Keep the real stdout aside and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("SI_BENCH_FORCE_DIST") == "1":  # the env knob rehearses the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import subspaceinference_jl_amd as si
    table, n_par = layer_table()
    assert n_par == 1047361

    # ---- synthetic inputs (deterministic), made resident in HBM before any timed region
    rng = np.random.default_rng(0)
    x = np.asfortranarray(rng.standard_normal((DIMS[0], B)))
    y = np.asfortranarray(rng.standard_normal((DIMS[-1], B)))
    w0 = torch.from_numpy(glorot_flat(1).astype(np.float64)).cuda()
    gen = torch.Generator(device="cuda").manual_seed(2)
    steps = torch.randn(K_SNAP, n_par, generator=gen, device="cuda", dtype=torch.float64) * 0.01
    ldw = n_par + (n_par & 1)  # rows padded to an even length: 8-B aligned fp32 pairs for the batched push
    snaps = torch.zeros(K_SNAP, ldw, device="cuda", dtype=torch.float32)
    snaps[:, :n_par] = (w0[None, :] + torch.cumsum(steps, dim=0)).to(torch.float32)  # K x N fp32, random walk
    del steps
    torch.cuda.synchronize()

    ctx = si.Context(local_rank)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- subspace construction (K pushes + Gram + eig + project), results stay on the device
    def construct():
        ctx.construct_begin(n_par, K_SNAP)
        # T=100 epochs, full batch, c=1: n = i.  The snapshots are device-resident, so they are pushed in one pass
        # (bit-identical to K_SNAP single pushes, tests/test_gpu_parity.py::test_push_batch_equals_sequential)
        ctx.construct_push_batch_dev(snaps.data_ptr(), 0, ldw, np.arange(1, K_SNAP + 1, dtype=np.float64))
        return ctx.construct_finish(M, want_swa=False, want_p=False)
    ctx.set_profiling(True)
    construct()  # warm-up (allocations, code-object load)
    ctx.reset_stats()
    construct()  # per-kernel breakdown (hipEvent pairs around every launch: ~5 us of stream time each)
    ctx.synchronize()
    cst = ctx.stats()
    ctx.set_profiling(False)
    barrier()
    t0 = time.perf_counter()
    _, _, svals, _ = construct()  # the wall-clock figure, without the event pairs
    ctx.synchronize()
    construct_ms = (time.perf_counter() - t0) * 1e3

    # ---- sampling.  The timed region carries event pairs around the DOMINANT kernel only (roofline.achieved is its
    # live average launch duration); the per-class breakdown comes from a short untimed pass afterwards.
    ctx.infer_setup(table, n_par, M, None, None, x, y, SIGMA_M)
    ctx.set_profiling(True, classes=["dense_main"])
    ctx.sample_rwmh(max(1, args.warmup), SIGMA_Z, seed=100 + rank, chain_id0=rank, want_z=False)
    barrier()
    ctx.reset_stats()
    t0 = time.perf_counter()
    z, lp, acc = ctx.sample_rwmh(args.steps, SIGMA_Z, seed=100 + rank, chain_id0=rank)
    barrier()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    if dist is not None:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = world * args.steps / dt
    bsteps = 20
    ctx.set_profiling(True)
    ctx.reset_stats()
    ctx.sample_rwmh(bsteps, SIGMA_Z, seed=100 + rank, chain_id0=rank, want_z=False)
    ctx.synchronize()
    bst = ctx.stats()

    # ---- extras (outside every timed region above): the "next" rows at the same workload
    extras = {}
    if rank == 0:
        ctx.reset_stats()
        zz = np.ascontiguousarray(z[:, 0, 0])
        ctx.logdensity_grad(zz)  # warm-up (workspace allocation)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.logdensity_grad(zz)
        extras["logdensity_grad_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        ctx.train_setup(table, n_par, glorot_flat(1), x, y, B, 2, 1e-3, 0.9, 0.999)  # ADAM, full batch
        ids = np.arange(B)
        ctx.train_step(ids)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.train_step(ids, want_loss=False)
        ctx.synchronize()
        extras["train_step_full_batch_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        bs = ctx.stats()["backward"]
        extras["backward_sweep_tflops"] = bs["flops"] / max(bs["ms"], 1e-9) / 1e9

    if rank == 0:
        dm = st["dense_main"]
        avg_ms = dm["ms"] / max(1, dm["launches"])
        fl = dm["flops"] / max(1, dm["launches"])
        achieved = fl / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_dense_main.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        out = {
            "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg2: Chain(Dense(128,960,relu),Dense(960,960,relu),Dense(960,1)) N=1047361, "
                                   "X 128x100000 Y 1x100000 fp64, M=20, K=100 fp32 snapshots, RWMH sigma_z=0.1 sigma_m=1",
                       "chains_per_gpu": 1, "parallelism": "independent chains x%d (one per GPU), no data-path collective" % world},
            "construct_wall_ms": construct_ms,
            "construct_device_ms": {k: round(cst[k]["ms"], 4) for k in ("push", "gram", "gram_reduce", "project")},
            "construct_host_eig_ms": round(cst["eig_host"]["ms"], 4),
            "sample_device_ms_per_step": {k: round(bst[k]["ms"] / bsteps, 4) for k in ("reconstruct", "dense", "sse", "rwmh")},
            "accept_rate": float(acc[0]), "lp_last": float(lp[-1, 0]),
            "roofline": {"kernel": "dense_f64_kernel<96,128> layer 960x960 + fused 960->1 tail (v_mfma_f64_16x16x4_f64)", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F64_TFLOPS, "traffic": traffic,
                         "avg_launch_ms": avg_ms, "flops_per_launch": fl, "launches": dm["launches"]},
            "device": ctx.device_name(),
            "next_rows": extras,
        }
        if world == 1 and not args.no_cpu_baseline:
            # W_swa / P of the construction just timed, brought to the host only for the CPU leg (outside all timers)
            w_swa, p, _, _ = ctx.construct_finish(M)
            cb, lp_cpu = cpu_baseline(table, w_swa, p, x, y, z[:, 0, 0], args.cpu_budget)
            out["cpu_baseline"] = cb
            out["parity_lp_rel_err_vs_oracle"] = abs(lp_cpu - float(lp[0, 0])) / abs(lp_cpu)
        else:
            out["cpu_baseline"] = None
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
