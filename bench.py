#!/usr/bin/env python3
"""bench.py -- the hot path of SubspaceInference.jl on MI355X at BASELINE.json's configurations.

    python bench.py --gpus N --steps K --warmup W [--mode chains|construct-sharded|data-sharded]

With N > 1 and no RANK in the environment the parent process -- BEFORE importing torch or touching HIP -- starts the N
ranks itself (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), relays
rank 0's JSON line and exits non-zero when any rank fails.  Started under torchrun (RANK set) it is one rank.
`n_gpus` in the line is the number of ranks RCCL actually saw (all-reduce of ones); N > visible GPUs is an error.

--mode chains (default; the headline, BASELINE cfg2 / cfg3)
    A "step" is ONE posterior sample: propose z' -> W_swa + P z' -> Dense-chain forward over the full data in fp64 ->
    Gaussian log-likelihood -> Metropolis accept (reference src/space_inference.jl:90-95,111-116), all on the device
    with inputs resident in HBM.  Each GPU runs its own independent chain (weak scaling, no data-path collective);
    `value` = samples of all ranks / max-over-ranks time.  The subspace-construct wall-clock (the metric's second half)
    is measured before sampling and reported as `construct_wall_ms`.
--mode construct-sharded --config cfg4|cfg5
    Row-sharded construction: every rank holds N/world rows of the snapshots / W_swa / A / P; K1, K2, K3 are
    row-local, ONE in-place RCCL all-reduce of the K x K Gram matrix.  A step = one whole construction.
--mode data-sharded
    cfg5 density with the observations split over the ranks (B_total = 131072 fixed: strong scaling), W_swa / P (26 GB)
    replicated and device-resident, ONE 8-byte in-place RCCL all-reduce per transition.  A step = one transition.

One JSON line on stdout (rank 0).  `roofline` is computed live from the library's own hipEvent pairs around the
dominant kernel inside the timed region; `cpu_baseline` (N = 1, chains mode) times the CPU restatement on this box's
host cores (bounded sample): the C/OpenMP port single-core and all-core, and the NumPy/OpenBLAS port.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time


def _cpu_budget():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box shows 256 CPUs and
    grants 16).  NumPy's BLAS starts one spinning thread per VISIBLE CPU; once they have burnt the quota the cgroup throttles
    the whole process ~80 ms at a time -- in the middle of whatever is being timed."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


os.environ.setdefault("OPENBLAS_NUM_THREADS", str(_cpu_budget()))   # before NumPy is imported anywhere below

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "posterior samples/sec (whole node) + subspace-construct wall-clock, 1M-param MLP M=20"
DIMS, ACTS = [128, 960, 960, 1], [1, 1, 0]  # Chain(Dense(128,960,relu), Dense(960,960,relu), Dense(960,1))
B, M, K_SNAP = 100000, 20, 100
SIGMA_Z, SIGMA_M = 0.1, 1.0
PEAK_F64_TFLOPS = 78.6   # MI355X fp64 matrix peak (datasheet; the guide's table has no f64 row -- DESIGN.md section 4)
PEAK_F32_TFLOPS = 157.3  # MI355X fp32 matrix peak (guide: v_mfma_f32_32x32x2_f32, 155 measured)
PMC_PROFILE_F32 = "r05_pmc_dense_f32_main.json"   # the same passes on the fp32 kernel (sha-keyed to kernels_gemm_f32.hip)
PEAK_HBM_GBS = 8000.0
PMC_PROFILE = "r05_pmc_dense_main.json"
CFG = {  # construct-only configurations
    "cfg2": dict(n=1047361, k=100, m=20),
    "cfg4": dict(n=5200266, k=200, m=20),
    "cfg5": dict(n=51138049, k=128, m=64),
}
DIMS5, ACTS5, B5_TOTAL, M5 = [1024, 6656, 6656, 1], [1, 1, 0], 8 * 16384, 64


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------ launcher
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", choices=["chains", "construct-sharded", "data-sharded"], default="chains")
    ap.add_argument("--config", choices=sorted(CFG), default="cfg5", help="construct-sharded: which N / K / M")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU-baseline work (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run-launcher", action="store_true",
                    help="ranks join a gloo group, count themselves and print a line with value null: exercises the "
                         "launch / relay / failure logic on a machine without GPUs (tests/test_bench_launcher.py)")
    ap.add_argument("--fail-rank", type=int, default=-1, help="(dry run) this rank exits non-zero")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    dflt = {"chains": (100, 10), "construct-sharded": (5, 2), "data-sharded": (20, 3)}[args.mode]
    args.steps = dflt[0] if args.steps is None else args.steps
    args.warmup = dflt[1] if args.warmup is None else args.warmup
    return args


# tools/bench_rehearsal.py flips these: the N-rank flow of this file on ONE GPU (ranks share the device over gloo).
# bench.py itself has no such mode -- what the driver runs is always one rank per GPU over the in-library RCCL.
REHEARSAL = False
FORCE_DIST = False   # tools/bench_rehearsal.py --world1-rccl: take the N-rank code path (in-library RCCL) with ONE rank
ENTRY = os.path.abspath(__file__)   # the script torchrun starts in every rank


def launch_ranks(args, argv):
    """Parent of an N-rank run.  Imports neither torch nor the HIP library: the ranks are CHILD processes (a process
    that has initialised the GPU must never exec another program), and a failed child is a failed run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), ENTRY] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    # N processes of the library share this box's CPU quota: each rank's host copy pool gets its share (the library also
    # shrinks the pool by itself once a rank joins a communicator; the export covers the copies made before that)
    env.setdefault("SI_HOST_COPY_THREADS", str(max(1, _cpu_budget() // (2 * args.gpus))))
    log("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and ln.rstrip().endswith("}")]
    if r.returncode != 0:
        log("bench.py: a rank failed (torchrun exit code %d); no result line" % r.returncode)
        return r.returncode if r.returncode > 0 else 1
    if len(lines) != 1:
        log("bench.py: expected exactly one JSON line from rank 0, got %d" % len(lines))
        return 1
    try:
        json.loads(lines[0])
    except ValueError:
        log("bench.py: rank 0's line is not valid JSON")
        return 1
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()
    return 0


# ------------------------------------------------------------------------------------------------ helpers
def layer_table(dims, acts):
    table, off = [], 0
    for fin, fout, act in zip(dims[:-1], dims[1:], acts):
        table.append((fin, fout, act, off, off + fin * fout))
        off += fin * fout + fout
    return table, off


def glorot_flat(seed, dims=DIMS):
    import numpy as np
    rng = np.random.default_rng(seed)
    parts = []
    for fin, fout in zip(dims[:-1], dims[1:]):
        w = ((rng.random((fout, fin)) - 0.5) * np.sqrt(24.0 / (fin + fout))).astype(np.float32)
        parts += [w.reshape(-1, order="F"), np.zeros(fout, dtype=np.float32)]
    return np.concatenate(parts)


def _sha16(path):
    """sha256 of the CODE of a kernel source: `//` comments and blank lines are dropped first, so that a comment edit does
    not orphan a PMC profile of the same machine code (tools/make_pmc_json.py hashes the same way)"""
    import re
    with open(path) as f:
        code = "\n".join(ln for ln in (re.sub(r"//.*", "", raw).rstrip() for raw in f) if ln)
    return hashlib.sha256(code.encode()).hexdigest()[:16]


def pmc_traffic(profile, kernel_sources):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE,
    separate passes, gfx950 correction applied -- MI355X_MICROARCH.md), valid ONLY while the kernel sources are the ones
    the profile was taken on: the profile records their sha256; on a mismatch `traffic` is null rather than stale."""
    path = os.path.join(ROOT, "profiles", profile)
    if not os.path.exists(path):
        return None, "no PMC profile committed"
    rec = json.load(open(path))
    want = rec.get("kernel_code_sha16", {})
    have = {s: _sha16(os.path.join(ROOT, "subspaceinference.jl_amd", "csrc", s)) for s in kernel_sources}
    if not want or any(want.get(s) != have[s] for s in kernel_sources):
        return None, "profiles/%s was taken on other kernel sources (sha mismatch): re-run tools/profile_round.sh + tools/make_pmc_json.py" % profile
    return rec.get("hbm_bytes_per_launch"), "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same kernel sources by sha256)" % profile


def host_info():
    info = {"host_cpus": os.cpu_count(), "cpu_model": "unknown"}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    info["cpu_model"] = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return info


def cpu_baseline(table, w_swa, p, x, y, z0, budget_s):
    """The CPU restatement (oracle) timed on the host, bounded sample of the same workload: (1) the C/OpenMP port
    (oracle/subspace_oracle_c.c: blocked fp64 GEMM, AVX-512 / AVX2 by run-time dispatch) on ONE core and on ALL cores,
    (2) the NumPy/OpenBLAS port.  `value` is the fastest all-core figure -- the baseline most favourable to the CPU."""
    import numpy as np
    from oracle import subspace_oracle as so
    out = dict(host_info(), kind="port", unit="samples/s")
    rng = np.random.default_rng(123)
    legs = {}
    lp0 = None
    try:
        from oracle import c_port
        cores = c_port.max_threads()     # affinity mask capped by the cgroup CPU quota (oracle/c_port.py cpu_budget)
        out["cpu_budget"] = {"usable_cpus": c_port.cpu_budget()[0], "cgroup_quota_cpus": c_port.cpu_budget()[1]}
        lpc = c_port.logdensity(table, w_swa, p, x, y, SIGMA_M, z0, threads=cores)   # warm-up + parity
        t0 = time.perf_counter()
        n = 0
        while n < 200 and time.perf_counter() - t0 < budget_s * 0.4:
            c_port.logdensity(table, w_swa, p, x, y, SIGMA_M, z0 + SIGMA_Z * rng.standard_normal(M), threads=cores)
            n += 1
        dt = time.perf_counter() - t0
        legs["c_openmp_all_cores"] = {"value": n / dt, "threads": cores, "evaluations": n, "seconds": round(dt, 2),
                                      "isa": c_port.isa()}
        t0 = time.perf_counter()
        c_port.logdensity(table, w_swa, p, x, y, SIGMA_M, z0, threads=1)
        dt1 = time.perf_counter() - t0
        legs["c_openmp_single_core"] = {"value": 1.0 / dt1, "threads": 1, "evaluations": 1, "seconds": round(dt1, 2)}
        lp0 = lpc
    except Exception as e:  # the C port is optional test infrastructure; its absence is reported, not hidden
        legs["c_openmp_error"] = repr(e)
    blas, threads = "unknown", os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info
        tinfo = threadpool_info()
        threads = max([i.get("num_threads", 1) for i in tinfo] or [threads])
        blas = ", ".join(sorted({"%s %s" % (i.get("internal_api", "?"), i.get("version", "?")) for i in tinfo
                                 if i.get("user_api") == "blas"})) or blas
    except Exception:
        pass
    limiter = None
    try:  # OpenBLAS would start one thread per logical CPU it sees; keep it within what the container may use
        from threadpoolctl import threadpool_limits
        from oracle import c_port as _cp
        if _cp.cpu_budget()[0] < threads:
            threads = _cp.cpu_budget()[0]
            limiter = threadpool_limits(limits=threads)
    except Exception:
        pass
    t0 = time.perf_counter()
    lpn = so.logdensity(table, w_swa, p, x, y, SIGMA_M, z0)
    t1 = time.perf_counter() - t0
    n = int(max(1, min(30, (budget_s * 0.5 - t1) // max(t1, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        so.logdensity(table, w_swa, p, x, y, SIGMA_M, z0 + SIGMA_Z * rng.standard_normal(M))
    dt = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits()
    legs["numpy_openblas"] = {"value": n / dt, "threads": int(threads), "evaluations": n, "seconds": round(dt, 2), "blas": blas}
    if lp0 is None:
        lp0 = lpn
    else:
        legs["c_vs_numpy_lp_rel_diff"] = abs(lp0 - lpn) / abs(lpn)
    best = max((k for k in legs if isinstance(legs[k], dict) and k != "c_openmp_single_core"), key=lambda k: legs[k]["value"])
    out.update(value=legs[best]["value"], cores=int(legs[best]["threads"]), best_leg=best, legs=legs,
               single_core_value=legs.get("c_openmp_single_core", {}).get("value"),
               sample="%d density evaluations (W_swa+P*z, fp64 forward over X 128x%d, SSE) of the cfg2 workload, %s, %.1f s"
                      % (legs[best]["evaluations"], B, best, legs[best]["seconds"]))
    return out, lpn


class Rank:
    """One rank.  Control plane: a torch.distributed `gloo` group (started by torchrun's env) that ships the 128-byte RCCL
    id and brackets the timed regions with barriers.  Data plane: the RCCL communicator INSIDE the library
    (si_comm_init_rank, attach()) -- every GPU collective of the hot path, the rank count and the max-over-ranks of the
    timings go through it; torch never touches a device buffer of the library."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.args = args
        self.backend = None
        self.share_gpu = REHEARSAL   # tools/bench_rehearsal.py only: the ranks share the visible GPU(s) over gloo
        self.ctx = None   # the si ctx whose in-library communicator carries the data plane (attach)

    def init(self, backend="gloo"):
        self.backend = backend
        if self.world > 1 or FORCE_DIST:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group(backend)
            self.dist = dist

    def attach(self, ctx):
        """join the in-library RCCL communicator (one rank per GPU); in tools/bench_rehearsal.py the ranks share
        a device, which RCCL refuses, so the rehearsal stays on the gloo test double of dist.py.  Should the communicator fail to
        come up on ANY rank (it has never met more than one GPU before the driver's 8-GPU run), every rank drops it and
        the run goes on over the gloo control plane -- the headline mode exchanges nothing per step -- with the error
        reported in the line (`comm_error`) instead of a lost scaling curve."""
        self.comm_error = None
        self.comm_hung = False
        if self.dist is not None and not self.share_gpu:
            import threading
            import torch  # noqa: F401
            import subspaceinference_jl_amd as si
            from subspaceinference_jl_amd import dist as sd
            box = {"err": ""}

            def bring_up():
                try:
                    sd.comm_init(ctx)
                    ctx.comm_barrier()
                except Exception as e:   # noqa: BLE001 -- reported, not hidden
                    box["err"] = repr(e)
            # in a thread with a deadline: a communicator that never comes up (one rank failed before it joined) must not
            # take the run with it.  ctypes releases the GIL during the call.
            limit = float(os.environ.get("SI_BENCH_COMM_TIMEOUT", "120"))
            th = threading.Thread(target=bring_up, daemon=True)
            th.start()
            th.join(limit)
            err = box["err"]
            if th.is_alive():
                err = "si_comm_init_rank did not return within %.0f s" % limit
                self.comm_hung = True
            flags = [None] * self.world
            self.dist.all_gather_object(flags, err)
            if any(flags):
                self.comm_error = "; ".join("rank %d: %s" % (r, f) for r, f in enumerate(flags) if f)
                log("bench.py: in-library RCCL communicator unavailable (%s): continuing over gloo" % self.comm_error)
                if self.comm_hung:
                    # the stuck call still owns that ctx: leave it alone for good and carry on with a fresh one
                    ctx = si.Context(self.local_rank)
                else:
                    try:
                        ctx.comm_destroy()
                    except Exception:   # noqa: BLE001
                        pass
            else:
                self.ctx = ctx
        return ctx

    def count_ranks(self, device=None):
        """ranks the collective backend actually saw: all-reduce of ones (RCCL inside the library when attached)"""
        if self.ctx is not None:
            return int(round(float(self.ctx.comm_allreduce_host([1.0], "sum")[0])))
        if self.dist is None:
            return 1
        import torch
        t = torch.ones(1, dtype=torch.int64)
        self.dist.all_reduce(t)
        return int(t.item())

    def max_over_ranks(self, v, device=None):
        if self.ctx is not None:
            return float(self.ctx.comm_allreduce_host([v], "max")[0])
        if self.dist is None:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, v, device=None):
        if self.ctx is not None:
            return [float(a) for a in self.ctx.comm_allgather_host([v])[:, 0]]
        if self.dist is None:
            return [v]
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [float(o.item()) for o in outs]

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
        if getattr(self, "comm_hung", False):   # a thread is still inside the RCCL bring-up: do not wait for it at exit
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)


def emit(real_stdout, out):
    if REHEARSAL:
        out["rehearsal"] = "tools/bench_rehearsal.py: the ranks share the visible GPU(s) over gloo -- flow check, NOT a measurement"
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


# ------------------------------------------------------------------------------------------------ dry run (no GPU)
def run_dry(args, rk, real_stdout):
    rk.init("gloo")
    if rk.rank == args.fail_rank:
        raise SystemExit(3)
    n = rk.count_ranks("cpu")
    if rk.dist is not None:
        rk.dist.barrier()
    if rk.rank == 0:
        emit(real_stdout, {"metric": METRIC, "value": None, "unit": "samples/s", "n_gpus": n, "steps": args.steps,
                           "warmup": args.warmup, "dry_run": True, "mode": args.mode,
                           "host_copy_threads_env": os.environ.get("SI_HOST_COPY_THREADS"), "cpu_budget": _cpu_budget()})
    rk.finish()


# ------------------------------------------------------------------------------------------------ mode: chains
def run_chains(args, rk, real_stdout):
    import numpy as np
    import torch
    import subspaceinference_jl_amd as si
    from subspaceinference_jl_amd import flux
    rank, world = rk.rank, rk.world
    ctx = rk.attach(si.Context(rk.local_rank))
    n_seen = rk.count_ranks()
    table, n_par = layer_table(DIMS, ACTS)
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
    ns = np.arange(1, K_SNAP + 1, dtype=np.float64)   # T = 100 epochs, full batch, c = 1: n = i

    def phase(what):   # progress on stderr (rank 0): a run under a profiler that dies says where
        if rank == 0:
            log("bench.py: %s" % what)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if rk.dist is not None:
            rk.dist.barrier()

    # ---- subspace construction: K pushes + Gram + eig + project, results stay on the device.  Three ways in:
    #   per_push  100 x si_construct_push_dev -- what api.subspace_construction / the .jl wrapper call once per batch
    #             (si_train_push is this call on the device-resident Float32 weights); THE `construct_wall_ms`
    #   batched   si_construct_push_batch_dev -- all K snapshots in one pass, W_swa in registers (bit-identical); an
    #             OFFLINE entry point (snapshots collected first), reported as construct_wall_ms_batched
    #   host      100 x si_construct_push from a pageable Float32 host vector (the .jl wrapper's default path: Zygote /
    #             Flux step in Julia, extract_params, ccall), pipelined through pinned staging
    base = snaps.data_ptr()

    def construct(kind, a32=False):
        ctx.construct_begin(n_par, K_SNAP)
        if a32:
            ctx.construct_set_storage(0)   # SI_F32: the opt-in fp32 storage of the deviation matrix
        if kind == "batched":
            ctx.construct_push_batch_dev(base, 0, ldw, ns)
        elif kind == "per_push":
            for j in range(K_SNAP):
                ctx.construct_push_dev(base + 4 * ldw * j, 0, ns[j])
        else:
            for j in range(K_SNAP):
                ctx.construct_push(snaps_host[j], ns[j])
        return ctx.construct_finish(M, want_swa=False, want_p=False)

    def wall3(kind, a32=False):
        runs = []
        for _ in range(3):   # median of three back-to-back constructions, no event pairs
            barrier()
            t0 = time.perf_counter()
            construct(kind, a32)
            ctx.synchronize()
            runs.append((time.perf_counter() - t0) * 1e3)
        return sorted(runs)[1], [round(t, 4) for t in runs]
    phase("construct: per-push and batched")
    ctx.set_profiling(True)
    construct("per_push")  # warm-up (allocations, code-object load)
    ctx.reset_stats()
    construct("per_push")  # per-kernel breakdown (hipEvent pairs around every launch: ~5 us of stream time each)
    ctx.synchronize()
    cst = ctx.stats()
    ctx.reset_stats()
    construct("batched")
    ctx.synchronize()
    cst_b = ctx.stats()
    ctx.set_profiling(False)
    construct_ms, construct_runs = wall3("per_push")
    construct_b_ms, construct_b_runs = wall3("batched")
    # the same with the deviation matrix stored in fp32 (opt-in; W_swa, G, the eigen-decomposition and P stay fp64)
    phase("construct: fp32-stored A")
    ctx.set_profiling(True)
    construct("per_push", True)
    ctx.reset_stats()
    construct("per_push", True)
    ctx.synchronize()
    cst32 = ctx.stats()
    ctx.set_profiling(False)
    construct32_ms, construct32_runs = wall3("per_push", True)
    construct32_b_ms, _ = wall3("batched", True)
    phase("construct: host push")
    host_push = None
    if rank == 0:
        snaps_host = snaps[:, :n_par].cpu().numpy()   # 100 pageable Float32 vectors (419 MB)
        construct("host")                             # warm-up: pinned staging, copy pool
        hruns, pruns = [], []
        for _ in range(3):
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.construct_begin(n_par, K_SNAP)
            for j in range(K_SNAP):
                ctx.construct_push(snaps_host[j], ns[j])
            tq = time.perf_counter()                  # every push has RETURNED (the caller could be in its next step)
            ctx.synchronize()
            tp = time.perf_counter()
            ctx.construct_finish(M, want_swa=False, want_p=False)
            ctx.synchronize()
            hruns.append((time.perf_counter() - t0) * 1e3)
            pruns.append(((tp - t0) * 1e3, (tq - t0) * 1e3))
        k = int(np.argsort(hruns)[1])
        push_ms = pruns[k][0]
        # what the PCIe link itself delivers on this box: the same 100 x 4.19 MB from PINNED memory, no staging copy, no kernel
        hp = torch.empty(n_par * 4, dtype=torch.uint8).pin_memory()
        dv = torch.empty(n_par * 4, dtype=torch.uint8, device="cuda")
        dv.copy_(hp, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K_SNAP):
            dv.copy_(hp, non_blocking=True)
        torch.cuda.synchronize()
        link_gbs = K_SNAP * n_par * 4 / (time.perf_counter() - t0) / 1e9
        del hp, dv
        host_push = {"construct_host_push_ms": hruns[k], "runs": [round(t, 3) for t in hruns], "push_phase_ms": push_ms,
                     "calls_returned_after_ms": pruns[k][1],
                     "host_push_GBs": K_SNAP * n_par * 4 / (push_ms * 1e-3) / 1e9, "pcie_peak_GBs": 63.0,
                     "pinned_h2d_GBs_measured": link_gbs,
                     "frac_of_measured_link": K_SNAP * n_par * 4 / (push_ms * 1e-3) / 1e9 / link_gbs,
                     "note": "100 x si_construct_push of a pageable 4.19 MB Float32 vector: host copy pool -> pinned double "
                             "buffer -> async H2D + K1, no synchronisation per push; then Gram + eig + projection"}
        del snaps_host
    phase("construct: end to end through api.subspace_construction")
    # end to end through the drop-in call itself: subspace_construction(model, mse, data, ADAM; T = 100, M = 20) with the
    # training step on the device (src/subspace_construction.jl:37-59 as a whole: 100 x [gradient + update! + push] + psvd)
    e2e = e2e_f32 = None
    if rank == 0:
        wr = np.random.default_rng(1)
        mdl = flux.Chain(*[flux.Dense(i, o, a, rng=wr) for i, o, a in zip(DIMS[:-1], DIMS[1:], ACTS)])
        data = flux.DataLoader(x, y, batchsize=B)
        with si.Context(rk.local_rank) as solo:   # rank 0 alone: its own ctx, no communicator, no data-parallel step
            t0 = time.perf_counter()
            si.subspace_construction(mdl, flux.mse, data, flux.ADAM(1e-3), T=K_SNAP, c=1, M=M, ctx=solo, verbose=False,
                                     device_training=True, keep_on_device=True, data_parallel=False)
            solo.synchronize()
            e2e = (time.perf_counter() - t0) * 1e3
        # the same call on Float32 data: the device step then computes in fp32, as the reference's Zygote pass would
        # (src/subspace_construction.jl:39-43 with a Float32 model and Float32 X, Y)
        wr = np.random.default_rng(1)
        mdl32 = flux.Chain(*[flux.Dense(i, o, a, rng=wr) for i, o, a in zip(DIMS[:-1], DIMS[1:], ACTS)])
        data32 = flux.DataLoader(x.astype(np.float32), y.astype(np.float32), batchsize=B)
        with si.Context(rk.local_rank) as solo:
            t0 = time.perf_counter()
            si.subspace_construction(mdl32, flux.mse, data32, flux.ADAM(1e-3), T=K_SNAP, c=1, M=M, ctx=solo, verbose=False,
                                     device_training=True, keep_on_device=True, data_parallel=False)
            solo.synchronize()
            e2e_f32 = (time.perf_counter() - t0) * 1e3
        del data32, mdl32
    construct("per_push")   # the subspace the chains below sample in (rank 0's is broadcast when there are more ranks)
    if rk.ctx is not None:
        ctx.bcast_subspace(0, n_par, M)   # cfg3: (W_swa, P) device to device over RCCL inside the library, once

    # ---- sampling.  The timed region carries event pairs around the DOMINANT kernel only (roofline.achieved is its
    # live average launch duration); the per-class breakdown comes from a short untimed pass afterwards.
    phase("sampling: the timed region (fp64)")
    ctx.infer_setup(table, n_par, M, None, None, x, y, SIGMA_M)
    # every rank evaluates the density at the same point: identical (W_swa, P) after the broadcast <=> identical bits
    lp0_per_rank = rk.gather_floats(float(ctx.logdensity(np.full((M, 1), 0.25))[0]))
    ctx.set_profiling(True, classes=["dense_main"])
    ctx.sample_rwmh(max(1, args.warmup), SIGMA_Z, seed=100, chain_id0=rank, want_z=False)
    barrier()
    ctx.reset_stats()
    t0 = time.perf_counter()
    z, lp, acc = ctx.sample_rwmh(args.steps, SIGMA_Z, seed=100, chain_id0=rank)
    barrier()
    dt_local = time.perf_counter() - t0
    st = ctx.stats()
    dt = rk.max_over_ranks(dt_local)
    value = n_seen * args.steps / dt
    per_rank_ms = rk.gather_floats(dt_local / args.steps * 1e3)
    lp_per_rank = rk.gather_floats(float(lp[-1, 0]))
    ctx.set_profiling(False)
    # the same K transitions WITH the reference's output map (a13, src/space_inference.jl:125): every weight sample
    # delivered to a fresh pageable host array while the chain runs -- what the drop-in sub_inference call does
    phase("sampling: with the output map")
    ctx.sample_rwmh_weights(5, SIGMA_Z, seed=100, chain_id0=rank)   # warm-up: weight ring, pinned staging, copy pool
    barrier()
    t0 = time.perf_counter()
    zw, lpw, _, wmap = ctx.sample_rwmh_weights(args.steps, SIGMA_Z, seed=100, chain_id0=rank)
    barrier()
    dt_map = rk.max_over_ranks(time.perf_counter() - t0)
    map_ok = bool(np.array_equal(zw, z) and np.array_equal(wmap[:, -1, 0], ctx.reconstruct(z[:, -1:, 0])[:, 0]))
    del wmap
    # one long chain, reported beside `value`: itr = 1000 is what BASELINE's cfg2 names; a short --steps run is corroborated
    phase("sampling: 1000-step chain")
    barrier()
    t0 = time.perf_counter()
    z1k, lp1k, _ = ctx.sample_rwmh(1000, SIGMA_Z, seed=100, chain_id0=rank)
    barrier()
    dt_1000 = rk.max_over_ranks(time.perf_counter() - t0)
    bsteps = 20
    ctx.set_profiling(True)
    ctx.reset_stats()
    ctx.sample_rwmh(bsteps, SIGMA_Z, seed=100, chain_id0=rank, want_z=False)
    ctx.synchronize()
    bst = ctx.stats()
    ctx.set_profiling(False)

    # ---- compute_dtype = SI_F32 (SURVEY section 0 Q6 / 8(b),(d)): the same chain with X, the per-step weights and the activations
    # in fp32 on v_mfma_f32_32x32x2_f32 (head + SSE in fp64), reported BESIDE the fp64 headline, never instead of it
    phase("sampling: compute_dtype = SI_F32")
    from subspaceinference_jl_amd import _capi
    ctx.infer_setup(table, n_par, M, None, None, x, y, SIGMA_M, compute_dtype=_capi.SI_F32)
    ctx.set_profiling(True, classes=["dense_main"])
    ctx.sample_rwmh(max(1, args.warmup), SIGMA_Z, seed=100, chain_id0=rank, want_z=False)
    barrier()
    ctx.reset_stats()
    t0 = time.perf_counter()
    z32, lp32, acc32 = ctx.sample_rwmh(args.steps, SIGMA_Z, seed=100, chain_id0=rank)
    barrier()
    dt32 = rk.max_over_ranks(time.perf_counter() - t0)
    st32 = ctx.stats()
    ctx.set_profiling(False)
    z32k, lp32k, acc32k = ctx.sample_rwmh(1000, SIGMA_Z, seed=100, chain_id0=rank)   # divergence from the fp64 chain, same Philox stream
    same = np.all(z32k[:, :, 0] == z1k[:, :, 0], axis=0)
    f32_first_diff = -1 if bool(same.all()) else int(np.argmin(same))
    nsame = int(same.sum()) if f32_first_diff < 0 else f32_first_diff
    f32_lp_rel = float(np.max(np.abs(lp32k[:nsame, 0] - lp1k[:nsame, 0]) / np.abs(lp1k[:nsame, 0]))) if nsame else None
    ctx.infer_setup(table, n_par, M, None, None, x, y, SIGMA_M)   # back to the reference's arithmetic for what follows

    phase("next rows: gradient, training step")
    extras = {}
    if rank == 0:
        # the "next" rows at the same workload (outside every timed region above)
        ctx.reset_stats()
        ctx.set_profiling(True)
        z1 = np.ascontiguousarray(z[:, 0, 0])
        ctx.logdensity_grad(z1)  # warm-up (workspace allocation)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.logdensity_grad(z1)
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
        # the step in the caller's precision: Float32 data => fp32 forward and reverse sweep (si_train_setup_ex)
        x32, y32 = x.astype(np.float32), y.astype(np.float32)
        ctx.reset_stats()
        ctx.train_setup(table, n_par, glorot_flat(1), x32, y32, B, 2, 1e-3, 0.9, 0.999)
        ctx.train_step(ids)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.train_step(ids, want_loss=False)
        ctx.synchronize()
        extras["train_step_full_batch_ms_f32"] = (time.perf_counter() - t0) / 5 * 1e3
        bs32 = ctx.stats()["backward"]
        extras["backward_sweep_tflops_f32"] = bs32["flops"] / max(bs32["ms"], 1e-9) / 1e9
        extras["train_step_f32_note"] = ("Float32 (X, Y) with the Float32 model: the reference's own arithmetic for such data; fp32 operands on "
                                         "v_mfma_f32_32x32x2_f32 forward and reverse, loss / head partials / batch sums in fp64; the fp64 "
                                         "numbers beside it are what Float64 data get (unchanged)")
        del x32, y32
        ctx.set_profiling(False)
        # the reference's own worked example (docs/src/nn_example.md:112-118,188-194: 2-200-50-50-50-1 on 1000 observations):
        # the persistent grid loop for one chain, the one-launch density stacked over 512 chains
        nd, na, nb_obs = [2, 200, 50, 50, 50, 1], [1, 1, 1, 1, 0], 1000
        ntab, noff = [], 0
        for fin, fout, act in zip(nd[:-1], nd[1:], na):
            ntab.append((fin, fout, act, noff, noff + fin * fout))
            noff += fin * fout + fout
        nrng = np.random.default_rng(0)
        with si.Context(rk.local_rank) as nctx:
            nctx.infer_setup(ntab, noff, 20, 0.3 * nrng.standard_normal(noff), 0.05 * nrng.standard_normal((noff, 20)),
                             nrng.standard_normal((2, nb_obs)), nrng.standard_normal((1, nb_obs)), 1.0)
            nctx.sample_rwmh(50, 0.1, seed=1)
            t0 = time.perf_counter()
            nctx.sample_rwmh(20000, 0.1, seed=1)   # (sample(model, spl, itr) of the docs runs thousands of transitions per call)
            extras["nn_example_us_per_transition"] = (time.perf_counter() - t0) / 20000 * 1e6
            extras["nn_example_kernels_specialised_at_run_time"] = bool(nctx.chain_kernel_info()[1])
            nctx.sample_rwmh(10, 0.1, seed=1, nchains=512)
            t0 = time.perf_counter()
            nctx.sample_rwmh(100, 0.1, seed=1, nchains=512)
            extras["nn_example_512_chains_samples_per_s"] = 100 * 512 / (time.perf_counter() - t0)
            extras["nn_example_kernels_specialised_at_run_time"] = extras["nn_example_kernels_specialised_at_run_time"] and bool(nctx.chain_kernel_info()[0])
        # the same single-chain call from a C process (tools/nn_capi_probe.c through the C ABI, no interpreter in the process): the
        # kernel of this call runs 9 % slower inside a Python process (three clocks) -- what a ccall / cgo caller sees
        probe = os.path.join(ROOT, "tools", "bin", "nn_capi_probe")
        if os.path.exists(probe):
            try:
                import subprocess
                pr = subprocess.run([probe, os.path.join(ROOT, "subspaceinference.jl_amd", "libsubspace_hip.so")], capture_output=True, text=True, timeout=120)
                vals = [float(ln.split(":")[1].split("us")[0]) for ln in pr.stdout.splitlines() if ln.startswith("C ABI without Python")]
                if vals:
                    extras["nn_example_us_per_transition_c_process"] = min(vals)
            except Exception as e:   # (a measurement helper: never fatal)
                extras["nn_example_us_per_transition_c_process_error"] = repr(e)
        extras["nn_example_note"] = ("docs/src/nn_example.md's MLP, N = 15801, B = 1000, M = 20 (the largest M of its sweep): one chain in the "
                                     "persistent grid loop (63 workgroups, ONE grid barrier per transition, 20 000 transitions in one launch); "
                                     "512 chains stacked in the one-launch density with the activations in registers; both kernels compiled for "
                                     "this chain's shapes at run time (csrc/chain_spec.inc through hiprtc); nn_example_us_per_transition is measured in THIS process, which "
                                     "runs PyTorch's bundled HIP runtime (ROCm 7.0 in the wheel: the loop is 5 % slower under it, DESIGN 10.1b) and copies the samples "
                                     "into untouched NumPy pages; _c_process: the same call from a C process on the system runtime (the generic kernels of "
                                     "csrc/kernels_chain_grid.hip give 22 us and 0.65 M samples/s); round 4: 41.0 us and 0.336 M samples/s")

    if rank == 0:
        dm = st["dense_main"]
        avg_ms = dm["ms"] / max(1, dm["launches"])
        fl = dm["flops"] / max(1, dm["launches"])
        achieved = fl / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(PMC_PROFILE, ["kernels_gemm.hip"])
        ms_step = dt / args.steps * 1e3
        step_flops = 2.0 * B * sum(a * b for a, b in zip(DIMS[:-1], DIMS[1:]))
        step_tflops = step_flops / (ms_step * 1e-3) / 1e12

        def frac(stats, cls, bound):
            v = stats[cls]
            if v["ms"] <= 0:
                return None
            if bound == "hbm":
                a = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                return {"bound": "hbm", "achieved_GBs": round(a, 1), "frac": round(a / PEAK_HBM_GBS, 4), "ms": round(v["ms"], 4)}
            a = v["flops"] / (v["ms"] * 1e-3) / 1e12
            return {"bound": "mfma", "achieved_TFLOPs": round(a, 2), "frac": round(a / PEAK_F64_TFLOPS, 4), "ms": round(v["ms"], 4)}
        dm32 = st32["dense_main"]
        avg32 = dm32["ms"] / max(1, dm32["launches"])
        fl32 = dm32["flops"] / max(1, dm32["launches"])
        ach32 = fl32 / (avg32 * 1e-3) / 1e12 if avg32 > 0 else 0.0
        traffic32, traffic32_src = pmc_traffic(PMC_PROFILE_F32, ["kernels_gemm_f32.hip"])
        ms_step32 = dt32 / args.steps * 1e3
        out = {
            "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": n_seen, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg2: Chain(Dense(128,960,relu),Dense(960,960,relu),Dense(960,1)) N=1047361, "
                                   "X 128x100000 Y 1x100000 fp64, M=20, K=100 fp32 snapshots, RWMH sigma_z=0.1 sigma_m=1",
                       "chains_per_gpu": 1, "mode": "chains",
                       "parallelism": "independent chains x%d (one per GPU; chain id = rank), subspace broadcast once over RCCL "
                                      "inside the library, no data-path collective per step" % n_seen},
            "value_with_output_map": n_seen * args.steps / dt_map,
            "output_map_note": "the same %d transitions through si_sample_rwmh_weights (what sub_inference calls): every weight "
                               "sample (8.4 MB) streamed to a fresh pageable host array under the following transitions; "
                               "bit-identical to si_reconstruct: %s" % (args.steps, map_ok),
            "chain_1000_steps_samples_per_s": n_seen * 1000 / dt_1000,
            "comm": ("in-library RCCL (si_comm_*), world %d" % n_seen) if rk.ctx is not None else
                    ("none (one rank)" if rk.dist is None else "gloo control plane only"),
            "comm_error": getattr(rk, "comm_error", None),
            "per_rank_ms_per_step": [round(t, 4) for t in per_rank_ms], "lp_last_per_rank": lp_per_rank,
            "chains_are_independent": bool(len(set(lp_per_rank)) == len(lp_per_rank)),
            "subspace_identical_on_all_ranks": bool(len(set(lp0_per_rank)) == 1),   # lp at a fixed z, bit for bit, after si_bcast_subspace
            "construct_wall_ms": construct_ms, "construct_wall_ms_runs": construct_runs,
            "construct_wall_ms_note": "100 x si_construct_push_dev (one K1 launch per batch, the entry point api.py / the .jl "
                                      "wrapper use) + Gram + host eigensolve + projection",
            "construct_wall_ms_batched": construct_b_ms, "construct_wall_ms_batched_runs": construct_b_runs,
            "construct_a_fp32": {"note": "si_construct_set_storage(SI_F32): the deviation columns formed in fp64 and stored rounded once to fp32 "
                                         "(opt-in, SURVEY section 0 Q6); W_swa bit-exact, s rtol 1e-6, P 1e-5 of its scale vs the fp64 oracle "
                                         "(tests/test_gpu_a32.py).  The register-staged Gram kernels read it (no LDS-DMA variant): the Gram "
                                         "itself does not get faster, the push and the projection do",
                                 "construct_wall_ms": construct32_ms, "runs": construct32_runs, "construct_wall_ms_batched": construct32_b_ms,
                                 "device_ms": {k: round(cst32[k]["ms"], 4) for k in ("push", "gram", "gram_reduce", "project")}},
            "construct_host_push": host_push,
            "construct_end_to_end_ms": e2e,
            "construct_end_to_end_ms_f32": e2e_f32,
            "construct_end_to_end_who_gets_it": "the Python mirror's default (device_training = \"auto\" recognises flux.mse).  The Julia wrapper trains "
                                                "on the HOST unless called with device_training = true (a Julia closure cannot be recognised as "
                                                "mse): INTEGRATION.md shows the one-line opt-in.  _f32: the same call on Float32 (X, Y)",
            "construct_end_to_end_note": "subspace_construction(model, mse, DataLoader(batchsize = B), ADAM; T = 100, M = 20) through "
                                         "api.py with the training step on the device: 100 x [forward + reverse sweep + ADAM + K1] + psvd, "
                                         "incl. the one-off upload of X, Y and the weights",
            "construct_device_ms": {k: round(cst[k]["ms"], 4) for k in ("push", "gram", "gram_reduce", "project")},
            "construct_device_ms_batched": {k: round(cst_b[k]["ms"], 4) for k in ("push", "gram", "gram_reduce", "project")},
            "construct_host_eig_ms": round(cst["eig_host"]["ms"], 4),
            "construct_roofline": {"push": frac(cst, "push", "hbm"), "push_batched": frac(cst_b, "push", "hbm"),
                                   "gram": frac(cst, "gram", "mfma"), "project": frac(cst, "project", "hbm")},
            "sample_device_ms_per_step": {k: round(bst[k]["ms"] / bsteps, 4) for k in ("reconstruct", "dense", "sse", "rwmh")},
            "accept_rate": float(acc[0]), "lp_last": float(lp[-1, 0]),
            "roofline": {"kernel": "dense_f64_kernel<96,128> layer 960x960 + fused 960->1 tail (v_mfma_f64_16x16x4_f64)", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F64_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": avg_ms, "flops_per_launch": fl, "launches": dm["launches"]},
            "step_roofline": {"bound": "mfma", "flops_per_step": step_flops, "achieved": step_tflops, "peak": PEAK_F64_TFLOPS,
                              "unit": "TFLOP/s", "frac": step_tflops / PEAK_F64_TFLOPS,
                              "note": "ALL of a transition (propose, reconstruct, 3 layers, SSE, accept) against the fp64 matrix peak"},
            "value_f32": n_seen * args.steps / dt32, "ms_per_step_f32": ms_step32,
            "roofline_f32": {"kernel": "dense_f32_dma_kernel<192,128> layer 960x960 + fused 960->1 head (v_mfma_f32_32x32x2_f32)",
                             "bound": "mfma", "achieved": ach32, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": ach32 / PEAK_F32_TFLOPS,
                             "traffic": traffic32, "traffic_source": traffic32_src, "avg_launch_ms": avg32, "flops_per_launch": fl32,
                             "launches": dm32["launches"]},
            "f32": {"note": "compute_dtype = SI_F32 (si_infer_setup): X rounded once, W_swa + P z formed in fp64 and rounded once per "
                            "transition, activations fp32, narrow head + SSE in fp64; the SAME %d transitions on the same Philox "
                            "stream as `value`; `value` / `dtype` stay the fp64 chain" % args.steps,
                    "step_frac_of_fp32_peak": step_flops / (ms_step32 * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                    "accept_rate": float(acc32[0]), "accept_rate_1000": float(acc32k[0]),
                    "first_transition_of_1000_where_the_chain_leaves_the_f64_chain": f32_first_diff,
                    "max_lp_rel_diff_vs_f64_chain_while_identical": f32_lp_rel,
                    "tolerance_stated": "lp rtol 1e-5 vs the fp64 oracle in tests/test_gpu_f32.py (north_star: 1e-4)"},
            "device": ctx.device_name(),
            "next_rows": extras,
        }
        if n_seen == 1 and not args.no_cpu_baseline:
            # W_swa / P the chains sampled in, brought to the host only for the CPU leg (outside all timers)
            w_swa, p, _ = ctx.construct_get_result()
            cb, lp_cpu = cpu_baseline(table, w_swa, p, x, y, z[:, 0, 0], args.cpu_budget)
            out["cpu_baseline"] = cb
            out["parity_lp_rel_err_vs_oracle"] = abs(lp_cpu - float(lp[0, 0])) / abs(lp_cpu)
        else:
            out["cpu_baseline"] = None
        emit(real_stdout, out)
    ctx.close()
    rk.finish()


# ------------------------------------------------------------------------------------------------ mode: construct-sharded
def run_construct_sharded(args, rk, real_stdout):
    """Row-sharded construction: rank r holds rows row_shard(N, r, world) of every snapshot, of W_swa, A and P."""
    import numpy as np
    import torch
    import subspaceinference_jl_amd as si
    from subspaceinference_jl_amd import dist as sd
    rank, world = rk.rank, rk.world
    dev = torch.device("cuda", rk.local_rank)
    ctx = rk.attach(si.Context(rk.local_rank))
    n_seen = rk.count_ranks()
    cfg = CFG[args.config]
    n, k, m = cfg["n"], cfg["k"], cfg["m"]
    r0, r1 = sd.row_shard(n, rank, world)
    n_loc = r1 - r0
    ldw = n_loc + (n_loc & 1)
    gen = torch.Generator(device="cuda").manual_seed(1000 + rank)
    snaps = torch.zeros((k, ldw), device="cuda", dtype=torch.float32)
    cur = 0.02 * torch.randn(n_loc, generator=gen, device="cuda", dtype=torch.float32)
    for j in range(k):  # random-walk stream of this rank's rows (full-rank A), built row by row to bound memory
        cur = cur + 0.002 * torch.randn(n_loc, generator=gen, device="cuda", dtype=torch.float32)
        snaps[j, :n_loc] = cur
    torch.cuda.synchronize()
    ns = np.arange(1, k + 1, dtype=np.float64)

    def construct():
        ctx.construct_begin(n_loc, k)
        ctx.construct_push_batch_dev(snaps.data_ptr(), 0, ldw, ns)
        return _finish()

    def _finish():
        # K2 local -> ONE in-place RCCL all-reduce of the K x K Gram inside the library (si_construct_allreduce_gram) ->
        # replicated H1 -> K3 local; results stay on the device
        ctx.construct_gram()
        sd._allreduce_gram(ctx)
        return ctx.construct_finish(m, want_swa=False, want_p=False)[2]

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if rk.dist is not None:
            rk.dist.barrier()
    for _ in range(max(1, args.warmup)):
        construct()
    ctx.set_profiling(True)
    ctx.reset_stats()
    construct()
    ctx.synchronize()
    cst = ctx.stats()
    ctx.set_profiling(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s = construct()
    barrier()
    dt = rk.max_over_ranks(time.perf_counter() - t0, dev)
    ms = dt / args.steps * 1e3
    dev_ms = {kk: rk.gather_floats(cst[kk]["ms"], dev) for kk in ("push", "gram", "gram_reduce", "project", "eig_host")}
    if rank == 0:
        total_bytes = float(n) * (k * (4 + 8) + 16) + float(n) * k * 8 + float(n) * (k + m) * 8   # K1 + K2 + K3, algorithmic
        total_flops = float(n) * k * (k + 1) + 2.0 * n * k * m

        def blk(cls, bound):
            v = cst[cls]
            if v["ms"] <= 0:
                return None
            if bound == "hbm":
                a = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                return {"bound": "hbm", "achieved": a, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": a / PEAK_HBM_GBS, "rank0_ms": v["ms"]}
            a = v["flops"] / (v["ms"] * 1e-3) / 1e12
            return {"bound": "mfma", "achieved": a, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s", "frac": a / PEAK_F64_TFLOPS, "rank0_ms": v["ms"]}
        gram = blk("gram", "mfma")
        emit(real_stdout, {
            "metric": "subspace-construct wall-clock, row-sharded (%s)" % args.config, "value": ms, "unit": "ms",
            "n_gpus": n_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": False,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s construct-only: N=%d rows sharded %d-way (%d on rank 0), K=%d fp32 snapshots (device-resident), M=%d"
                                   % (args.config, n, n_seen, n_loc, k, m), "mode": "construct-sharded",
                       "parallelism": "row shards; K1/K2/K3 row-local, ONE in-place RCCL all-reduce of the %dx%d fp64 Gram (%d B)"
                                      % (k, k, 8 * k * k)},
            "aggregate": {"algorithmic_GBs": total_bytes / (ms * 1e-3) / 1e9, "algorithmic_TFLOPs": total_flops / (ms * 1e-3) / 1e12},
            "per_rank_device_ms": dev_ms,
            "roofline": dict(gram or {}, kernel="gram A'A (v_mfma_f64_16x16x4_f64), rank 0", traffic=None),
            "roofline_push": blk("push", "hbm"), "roofline_project": blk("project", "mfma" if m > 32 else "hbm"),
            "s1": float(s[0]), "device": ctx.device_name(), "cpu_baseline": None,
        })
    ctx.close()
    rk.finish()


# ------------------------------------------------------------------------------------------------ mode: data-sharded
def run_data_sharded(args, rk, real_stdout):
    """cfg5 density, observations split over the ranks; W_swa / P device-resident and used in place."""
    import numpy as np
    import torch
    import subspaceinference_jl_amd as si
    from subspaceinference_jl_amd import dist as sd
    rank, world = rk.rank, rk.world
    dev = torch.device("cuda", rk.local_rank)
    ctx = rk.attach(si.Context(rk.local_rank))
    n_seen = rk.count_ranks()
    table, n = layer_table(DIMS5, ACTS5)
    b0, b1 = sd.col_shard(B5_TOTAL, rank, world)
    b_loc = b1 - b0
    ld = (n + 63) // 64 * 64
    gen = torch.Generator(device="cuda").manual_seed(7)         # the same W_swa / P on every rank
    w_t = torch.zeros(ld, device="cuda", dtype=torch.float64)
    w_t[:n] = torch.from_numpy(glorot_flat(1, DIMS5).astype(np.float64)).cuda()
    p_t = torch.zeros((M5, ld), device="cuda", dtype=torch.float64)
    for j in range(M5):
        p_t[j, :n] = 1e-3 * torch.randn(n, generator=gen, device="cuda", dtype=torch.float64)
    gx = torch.Generator(device="cuda").manual_seed(11)
    x_all = torch.randn((B5_TOTAL, DIMS5[0]), generator=gx, device="cuda", dtype=torch.float64)   # [B, in] = in x B col-major
    y_all = torch.randn((B5_TOTAL, 1), generator=gx, device="cuda", dtype=torch.float64)
    x_t, y_t = x_all[b0:b1].contiguous(), y_all[b0:b1].contiguous()
    del x_all, y_all
    torch.cuda.synchronize()
    ctx.infer_setup_dev(table, n, M5, w_t.data_ptr(), p_t.data_ptr(), ld, x_t.data_ptr(), y_t.data_ptr(), DIMS5[0], 1,
                        b_loc, 1.0, borrow=True)
    d_total = B5_TOTAL

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if rk.dist is not None:
            rk.dist.barrier()
    sd.sample_data_sharded(ctx, max(2, args.warmup), 1e-3, seed=5, d_total=d_total)
    ctx.set_profiling(True, classes=["dense_main"])
    barrier()
    ctx.reset_stats()
    t0 = time.perf_counter()
    z, lp, acc = sd.sample_data_sharded(ctx, args.steps, 1e-3, seed=5, d_total=d_total)
    barrier()
    dt = rk.max_over_ranks(time.perf_counter() - t0, dev)
    st = ctx.stats()
    ctx.set_profiling(True)
    ctx.reset_stats()
    sd.sample_data_sharded(ctx, 5, 1e-3, seed=5, d_total=d_total)
    ctx.synchronize()
    bst = ctx.stats()
    ms = dt / args.steps * 1e3
    lps = rk.gather_floats(float(lp[-1, 0]), dev)
    if rank == 0:
        dm = st["dense_main"]
        avg_ms = dm["ms"] / max(1, dm["launches"])
        fl = dm["flops"] / max(1, dm["launches"])
        achieved = fl / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        rec = bst["reconstruct"]
        rec_gbs = rec["bytes"] / max(rec["ms"], 1e-9) / 1e6
        step_flops = 2.0 * B5_TOTAL * sum(a * b for a, b in zip(DIMS5[:-1], DIMS5[1:]))
        emit(real_stdout, {
            "metric": "posterior samples/sec, data-sharded density (cfg5 wide MLP, M=64)", "value": args.steps / dt,
            "unit": "samples/s", "n_gpus": n_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg5: Chain(Dense(1024,6656,relu),Dense(6656,6656,relu),Dense(6656,1)) N=51138049, M=64, "
                                   "B_total=%d split %d-way (%d on rank 0), W_swa/P (26 GB) replicated, device-resident" % (B5_TOTAL, n_seen, b_loc),
                       "mode": "data-sharded",
                       "parallelism": "observations sharded; every rank runs the same Philox stream; ONE 8-byte in-place RCCL all-reduce per transition"},
            "ranks_agree_on_lp": bool(len(set(lps)) == 1),
            "per_step_device_ms": {k: round(bst[k]["ms"] / 5, 4) for k in ("reconstruct", "dense", "sse", "rwmh")},
            "roofline": {"kernel": "dense_f64_kernel layer 6656x6656 + fused 6656->1 tail, rank 0's share", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F64_TFLOPS,
                         "traffic": None, "avg_launch_ms": avg_ms, "flops_per_launch": fl, "launches": dm["launches"]},
            "roofline_reconstruct": {"kernel": "reconstruct_kernel over the 26 GB P (K4)", "bound": "hbm", "achieved": rec_gbs,
                                     "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": rec_gbs / PEAK_HBM_GBS},
            "step_roofline": {"bound": "mfma", "flops_per_step": step_flops, "achieved": step_flops / (ms * 1e-3) / 1e12 / n_seen,
                              "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s per GPU",
                              "frac": step_flops / (ms * 1e-3) / 1e12 / n_seen / PEAK_F64_TFLOPS},
            "lp_last": lps[0], "device": ctx.device_name(), "cpu_baseline": None,
        })
    ctx.close()
    rk.finish()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        # N ranks were asked for and nobody started them: do it ourselves, before torch / HIP are touched
        raise SystemExit(launch_ranks(args, argv))

    # Only the JSON line may reach stdout: RCCL prints a version banner to fd 1 when the communicator is created.
    # Keep the real stdout aside and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rk = Rank(args)
    if rk.world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d (start it as `python bench.py --gpus N` or under "
                         "torchrun with --nproc-per-node N)" % (rk.world, args.gpus))
    if args.dry_run_launcher:
        return run_dry(args, rk, real_stdout)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    if rk.share_gpu:
        rk.local_rank %= torch.cuda.device_count()
        log("bench.py: rehearsal -- rank %d uses GPU %d over gloo (timings are not a result)" % (rk.rank, rk.local_rank))
    if rk.local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: --gpus %d exceeds the %d visible GPUs" % (args.gpus, torch.cuda.device_count()))
    torch.cuda.set_device(rk.local_rank)
    rk.init("gloo")
    {"chains": run_chains, "construct-sharded": run_construct_sharded, "data-sharded": run_data_sharded}[args.mode](args, rk, real_stdout)


if __name__ == "__main__":
    main()
