"""Random problems through the sharded flows over the in-library RCCL communicator at world 1 (every collective is the
identity, so each flow must give the bits of the single-GPU entry points): row-sharded construction (Gram all-reduce, all-gather
in column chunks, subspace broadcast), data-sharded sampling (library loop and step-wise with the explicit collective),
data-parallel training step.  For the guard-page development library.
usage: SI_PROBE_DEV=1 SI_GUARD_ALLOC=end|begin python3 tools/guard_fuzz_comm.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subspaceinference_jl_amd as si  # noqa: E402
from oracle import subspace_oracle as so  # noqa: E402

if os.environ.get("SI_PROBE_DEV"):
    si._capi.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libsubspace_hip_dev.so")
from subspaceinference_jl_amd import dist as sd  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
plain = si.Context(0)
comm = si.Context(0)
sd.comm_init(comm, rank=0, world=1)
for case in range(cases):
    nl = int(rng.integers(1, 4))
    dims = [int(rng.choice([1, 3, 12, 33]))] + [int(rng.choice([3, 17, 64, 96, 130])) for _ in range(nl - 1)] + [int(rng.choice([1, 2, 5]))]
    acts = [int(rng.integers(0, 4)) for _ in range(nl)]
    b = int(rng.choice([5, 64, 300, 1000]))
    table, n = so.layer_table(dims, acts)
    k = int(rng.integers(2, 30))
    m = int(rng.integers(1, min(k, 6) + 1))
    print("case %d dims %s acts %s B %d N %d K %d M %d" % (case, dims, acts, b, n, k, m), flush=True)
    base = 0.3 * rng.standard_normal(n)
    snaps = []
    for j in range(k):
        base = base + 0.05 * rng.standard_normal(n)
        snaps.append(base.astype(np.float32))
    x = np.asfortranarray(rng.standard_normal((dims[0], b)))
    y = np.asfortranarray(rng.standard_normal((dims[-1], b)))
    for c in (plain, comm):
        c.construct_begin(n, k)
        for j, w in enumerate(snaps):
            c.construct_push(w, float(1 + j))
    try:
        if k > n:   # si_construct_finish alone would take its K > N route (A A'); the sharded flow sums the K x K Gram matrix:
            plain.construct_gram()   # same route on both sides, so that "same bits" stays the assertion
        w0, p0, s0, _ = plain.construct_finish(m)
    except si.BoundsError:
        continue
    s1 = sd.sharded_construct_finish_dev(comm, m, n)          # Gram all-reduced in place, all-gather, full result on the device
    w1, p1, sg = comm.construct_get_result()
    assert np.array_equal(w0, w1) and np.array_equal(p0, p1) and np.array_equal(s0, s1) and np.array_equal(s0, sg)
    sd.replicate_subspace_dev(comm, n, m, src=0)               # broadcast (identity at world 1): still the same buffers
    w2, p2, _ = comm.construct_get_result()
    assert np.array_equal(w0, w2) and np.array_equal(p0, p2)
    plain.infer_setup(table, n, m, w0, p0, x, y, 0.9)
    comm.infer_setup(table, n, m, None, None, x, y, 0.9)
    nch = int(rng.integers(1, 3))
    z0, lp0, a0 = plain.sample_rwmh(8, 0.1, seed=case, nchains=nch)
    z1, lp1, a1 = sd.sample_data_sharded(comm, 8, 0.1, seed=case, d_total=dims[-1] * b, nchains=nch)
    assert np.array_equal(z0, z1) and np.array_equal(lp0, lp1) and np.array_equal(a0, a1)
    comm.rwmh_begin(4, 0.1, case, 0, nch, dims[-1] * b)
    for _ in range(4):
        comm.rwmh_step_eval(on_device=True)
        comm.rwmh_allreduce_sse()
        comm.rwmh_step_accept(None)
    z2, lp2, _ = comm.rwmh_end()
    assert np.array_equal(z2, z0[:, :4]) and np.array_equal(lp2, lp0[:4])
    # data-parallel training step == plain training step
    w32 = w0.astype(np.float32)
    optk = int(rng.integers(0, 3))
    for c in (plain, comm):
        c.train_setup(table, n, w32, x, y, b, optk, 0.01, 0.9, 0.999)
    for _ in range(2):
        nb = int(rng.integers(1, b + 1))
        ids = rng.choice(b, nb, replace=False).astype(np.int64)
        la = plain.train_step(ids)
        lb = sd.train_step_data_parallel(comm, ids, nb)
        assert np.isclose(la, lb, rtol=1e-12)
    assert np.array_equal(plain.train_get_weights(), comm.train_get_weights())
print("guard_fuzz_comm: %d cases done" % cases, flush=True)
plain.close()
comm.close()
