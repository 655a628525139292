"""World 2 (and 4) of the in-library RCCL communicator on REAL devices, as fresh child processes -- one process per GPU, the id
through a file, no torch in the collective path (the rank offsets of si_construct_allgather, the non-root adopt path
of si_bcast_subspace, the sharded chain, si_train_step_dp and the collective error agreement had only ever run at world 1).
The build's GPU box has ONE device and RCCL refuses two ranks on one device: the tests skip there and run on any node that
shows at least two GPUs (the driver's multi-GPU node).  Every rank compares with the single-GPU entry points computed in
its own process: same kernels, same order, only the exchange differs -- bit for bit where the reduction order allows."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SI_ROOT"])
import subspaceinference_jl_amd as si
from subspaceinference_jl_amd import dist as sd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ctx = si.Context(rank)
sd.comm_init(ctx, rank=rank, world=world, id_file=os.environ["SI_ID_FILE"], timeout_s=120)
assert ctx.comm_info()[:2] == (world, rank)
assert np.array_equal(ctx.comm_allreduce_host(np.array([1.0, float(rank)]), "sum"), [world, world * (world - 1) / 2])
assert np.array_equal(ctx.comm_allgather_host(np.array([10.0 + rank])).ravel(), 10.0 + np.arange(world))
solo = si.Context(rank)                      # the single-GPU entry points, same device
rng = np.random.default_rng(0)               # identical inputs on every rank
n, k, m = 20011, 24, 5
w0 = rng.standard_normal(n)
snaps = [(w0 + c).astype(np.float32) for c in np.cumsum(0.01 * rng.standard_normal((k, n)), axis=0)]
ns = [float(1 + i // 3) for i in range(k)]
# ---- row-sharded construction: Gram all-reduce + all-gather == single-GPU finish
r0, r1 = si._capi.row_shard(n, rank, world)
ctx.construct_begin(r1 - r0, k)
solo.construct_begin(n, k)
for w, nn in zip(snaps, ns):
    ctx.construct_push(np.ascontiguousarray(w[r0:r1]), nn)
    solo.construct_push(w, nn)
ctx.construct_gram(); ctx.construct_allreduce_gram()
ws, ps, ss, _ = solo.construct_finish(m)
_, _, s_sh, _ = ctx.construct_finish(m, want_swa=False, want_p=False)
ctx.construct_allgather(n)
wg, pg, sg = ctx.construct_get_result()
assert np.array_equal(wg, ws)                                             # K1 is row-local: same bits
assert np.allclose(sg, ss, rtol=1e-11) and np.allclose(pg, ps, rtol=1e-8, atol=1e-12 * np.abs(ps).max())   # G summed in another order
# ---- bcast of a finished construction from the last rank: receivers adopt, values identical to the root's
root = world - 1
b = si.Context(rank); sd.comm_init(b, rank=rank, world=world, id_file=os.environ["SI_ID_FILE"] + ".b", timeout_s=120)
if rank == root:
    b.construct_begin(n, k)
    for w, nn in zip(snaps, ns):
        b.construct_push(w, nn)
    b.construct_finish(m, want_swa=False, want_p=False)
b.bcast_subspace(root, n, m)
wb, pb, sb = b.construct_get_result()
assert np.array_equal(wb, ws) and np.array_equal(pb, ps) and np.array_equal(sb, ss)
# ---- data-sharded chain == the single-GPU chain on the whole data (same Philox stream)
dims = [16, 64, 32, 1]
table, off = [], 0
for fi, fo, a in zip(dims[:-1], dims[1:], [1, 1, 0]):
    table.append((fi, fo, a, off, off + fi * fo)); off += fi * fo + fo
B = 4096
x, y = rng.standard_normal((16, B)), rng.standard_normal((1, B))
wq, pq = 0.2 * rng.standard_normal(off), 0.05 * rng.standard_normal((off, 4))
c0, c1 = sd.col_shard(B, rank, world)
ctx.infer_setup(table, off, 4, wq, pq, np.asfortranarray(x[:, c0:c1]), np.asfortranarray(y[:, c0:c1]), 1.0)
solo.infer_setup(table, off, 4, wq, pq, x, y, 1.0)
zs, lps, accs = ctx.sample_rwmh_sharded(60, 0.05, seed=5, d_total=B)
z1, lp1, acc1 = solo.sample_rwmh(60, 0.05, seed=5)
assert np.allclose(lps, lp1, rtol=1e-11) and np.allclose(zs, z1, rtol=1e-9, atol=1e-12) and np.array_equal(accs, acc1)
# ---- data-parallel training step == the single-GPU step on the whole batch
wt = (0.3 * rng.standard_normal(off)).astype(np.float32)
ctx.train_setup(table, off, wt, x, y, 512, 2, 1e-3, 0.9, 0.999)
solo.train_setup(table, off, wt, x, y, 512, 2, 1e-3, 0.9, 0.999)
for step in range(4):
    ids = rng.permutation(B)[:512]
    a0, a1 = sd.col_shard(ids.size, rank, world)
    ld = ctx.train_step_dp(ids[a0:a1], ids.size)
    ls = solo.train_step(ids)
    assert abs(ld - ls) <= 1e-9 * abs(ls)
assert np.allclose(ctx.train_get_weights(), solo.train_get_weights(), rtol=0, atol=2e-6)
# ---- collective error agreement: ONE rank fails its local preparation, NOBODY deadlocks, every rank gets an error
c = si.Context(rank); sd.comm_init(c, rank=rank, world=world, id_file=os.environ["SI_ID_FILE"] + ".c", timeout_s=120)
try:
    c.bcast_subspace(0, n, m)             # the root holds no finished construction
    raise SystemExit("bcast from an empty root did not fail")
except si.SubspaceError as e:
    assert e.code == (si._capi.SI_ERR_STATE if rank == 0 else si._capi.SI_ERR_COMM), (rank, e.code, str(e))
c.comm_barrier()                          # the communicator is still usable
for h in (c, b, solo, ctx):
    h.close()
print("WORLD_OK rank %d of %d" % (rank, world))
'''


def _gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.parametrize("world", [1, 2, 4])   # world 1 runs everywhere: it keeps the worker script itself honest
@pytest.mark.timeout(600)
def test_in_library_rccl_on_real_devices(world, tmp_path):
    if _gpus() < world:
        pytest.skip("needs %d GPUs (RCCL refuses two ranks on one device); this box shows %d" % (world, _gpus()))
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(world):
        env = dict(os.environ, SI_ROOT=ROOT, RANK=str(r), WORLD_SIZE=str(world), SI_ID_FILE=str(tmp_path / "id"),
                   SI_COMM_NONCE="t%d" % os.getpid(), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        outs = [p.communicate(timeout=500) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so_, se_)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "WORLD_OK rank %d" % r in so_, "rank %d:\n%s\n%s" % (r, so_[-2000:], se_[-4000:])
