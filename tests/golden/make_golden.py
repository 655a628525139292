"""Generates tests/golden/*.npz from the CPU oracle (oracle/subspace_oracle.py).

The reference holds NO fixtures for this path (test/runtests.jl is empty) and cannot be run here (no Julia),
so these vectors pin the ORACLE's outputs, not the reference's: PARITY UNPINNED.  They guard
against silent drift of the oracle and give the GPU tests committed inputs/outputs at README-toy size.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import subspace_oracle as so  # noqa: E402

TOY_DIMS, TOY_ACTS = [10, 20, 20, 2], [0, 0, 0]  # README.md:62 Chain(Dense(10,20),Dense(20,20),Dense(20,2))
CNN_WHC = (6, 6, 2)
CNN_SPEC = [("conv", (3, 3), 4, so.ACT_RELU, (1, 1), (1, 1)), ("maxpool", (2, 2)), ("conv", (2, 2), 3, so.ACT_TANH, (2, 2)),
            ("flatten",), ("dense", 3, so.ACT_IDENTITY)]


def glorot(rng, dims):
    ws = []
    for fin, fout in zip(dims[:-1], dims[1:]):
        ws.append(((rng.random((fout, fin)) - 0.5) * np.sqrt(24.0 / (fin + fout))).astype(np.float32))
        ws.append(np.zeros(fout, dtype=np.float32))
    return ws


def snapshot_stream(n, k, seed, dtype=np.float32):
    """random-walk weights w_t = w_0 + 0.01*cumsum(randn): full-rank deviation matrix."""
    rng = np.random.default_rng(seed)
    w0 = so.extract_params(glorot(np.random.default_rng(1), TOY_DIMS)).astype(np.float64)
    steps = 0.01 * rng.standard_normal((k, n))
    return [(w0 + c).astype(dtype) for c in np.cumsum(steps, axis=0)]


def main():
    table, n = so.layer_table(TOY_DIMS, TOY_ACTS)
    assert n == 682
    # ---- construction, K = 12 (T=3 epochs x 4 batches, c=1 -> n = 1,1,1,1,2,2,2,2,3,3,3,3), M = 3
    snaps = snapshot_stream(n, 12, seed=2)
    ns = [float(i) for i in (1, 2, 3) for _ in range(4)]
    w_swa, a = so.construct_stream(snaps, ns)
    p, s = so.projection_from_A(a, 3)
    np.savez_compressed(os.path.join(HERE, "toy_construct_k12.npz"), snapshots=np.stack(snaps), ns=np.array(ns),
                        W_swa=w_swa, A=a, P=p, s=s[:3])
    # ---- construction at the README toy's REAL shape (README.md:52-79): DataLoader batchsize 1 over 100
    # observations, T = 10 epochs, c = 1  =>  K = 1000 deviation columns of N = 682 weights (K > N), n = epoch per batch, M = 3
    snaps = snapshot_stream(n, 1000, seed=5)
    ns = [float(i) for i in range(1, 11) for _ in range(100)]
    w_swa_k, a_k = so.construct_stream(snaps, ns)
    p_k, s_k = so.projection_from_A(a_k, 3)
    np.savez_compressed(os.path.join(HERE, "toy_construct_k1000.npz"), snapshots=np.stack(snaps), ns=np.array(ns),
                        W_swa=w_swa_k, P=p_k, s=s_k[:20])
    # ---- density + RWMH at README toy size: X = rand(10,100), Y = rand(2,100), M=3, itr=10, sigma = 1
    rng = np.random.default_rng(0)
    x = rng.random((10, 100))
    y = rng.random((2, 100))
    zs = np.asfortranarray(np.random.default_rng(3).standard_normal((3, 5)))
    lps = np.array([so.logdensity(table, w_swa, p, x, y, 1.0, zs[:, j]) for j in range(5)])
    yhat0 = so.forward(table, so.reconstruct(w_swa, p, zs[:, 0]), x)
    z_chain, lp_chain, w_chain, nacc = so.sub_inference(table, x, y, w_swa, p, 1.0, 1.0, 10, seed=1234, chain=0)
    np.savez_compressed(os.path.join(HERE, "toy_density_rwmh.npz"), X=x, Y=y, W_swa=w_swa, P=p, Z=zs, lp=lps,
                        Yhat0=yhat0, Z_chain=z_chain, lp_chain=lp_chain, W_chain=w_chain, nacc=np.array(nacc),
                        seed=np.array(1234), sigma_z=np.array(1.0), sigma_m=np.array(1.0))
    # ---- a small Conv chain (Flux 0.11.2 / NNlib 0.7.23 semantics as restated in the oracle): log-density,
    # forward, gradient with respect to z and an RWMH chain.  conv 3x3 pad 1 (relu) -> MaxPool 2 -> conv 2x2 stride 2 (tanh)
    # -> flatten -> Dense 3
    spec = CNN_SPEC
    ctab, cn = so.conv_table(spec, CNN_WHC)
    rng = np.random.default_rng(11)
    cw = 0.3 * rng.standard_normal(cn)
    cp = np.asfortranarray(0.1 * rng.standard_normal((cn, 4)))
    cx = rng.standard_normal((CNN_WHC[0] * CNN_WHC[1] * CNN_WHC[2], 9))
    cy = rng.standard_normal((3, 9))
    czs = np.asfortranarray(rng.standard_normal((4, 5)))
    clps = np.array([so.logdensity(ctab, cw, cp, cx, cy, 0.7, czs[:, j]) for j in range(5)])
    cyhat0 = so.forward(ctab, so.reconstruct(cw, cp, czs[:, 0]), cx)
    _, cgrad, _ = so.logdensity_grad(ctab, cw, cp, cx, cy, 0.7, czs[:, 1])
    cz_chain, clp_chain, _, cnacc = so.sub_inference(ctab, cx, cy, cw, cp, 0.05, 0.7, 12, seed=77, chain=0)
    np.savez_compressed(os.path.join(HERE, "cnn_density_rwmh.npz"), X=cx, Y=cy, W_swa=cw, P=cp, Z=czs, lp=clps, Yhat0=cyhat0,
                        grad1=cgrad, Z_chain=cz_chain, lp_chain=clp_chain, nacc=np.array(cnacc), seed=np.array(77),
                        sigma_z=np.array(0.05), sigma_m=np.array(0.7))
    # ---- the training step (src/subspace_construction.jl:39-43 with Flux 0.11.2 Descent / Momentum / ADAM as
    # restated in the oracle): 12 steps of each optimiser on a tanh/relu toy chain, shuffled batches of 25 + a ragged one
    tdims, tacts = [10, 20, 20, 2], [so.ACT_TANH, so.ACT_RELU, so.ACT_IDENTITY]
    ttab, tn = so.layer_table(tdims, tacts)
    rng = np.random.default_rng(21)
    tx, ty = rng.random((10, 100)), rng.random((2, 100))
    tw0 = so.extract_params(glorot(np.random.default_rng(5), tdims))
    batches = [rng.permutation(100)[:25] for _ in range(11)] + [np.array([3, 97, 41])]
    opts = {"descent": ("descent", 0.1), "momentum": ("momentum", 0.01, 0.9), "adam": ("adam", 0.001, 0.9, 0.999)}
    out = dict(X=tx, Y=ty, w0=tw0, batches=np.array([np.pad(b, (0, 25 - b.size), constant_values=-1) for b in batches]))
    for name, opt in opts.items():
        w, st, losses = tw0.copy(), so.optimiser_state(tn, opt), []
        for ids in batches:
            losses.append(so.train_step(ttab, w, st, tx[:, ids], ty[:, ids], opt))
        out.update({name + "_w": w, name + "_loss": np.array(losses), name + "_m": st["m"], name + "_v": st["v"],
                    name + "_bp": np.array(st["bp"] if st["bp"] else [0.0, 0.0])})
    np.savez_compressed(os.path.join(HERE, "toy_train_steps.npz"), **out)
    # ---- the same 12 steps with Float32 DATA: a Float32 model on Float32 (X, Y) is an all-Float32 Zygote pass in the reference
    # (nothing promotes), and Flux's optimisers then round their step into the Float32 gradient array before `x .-= step`
    tx32, ty32 = tx.astype(np.float32), ty.astype(np.float32)
    out32 = dict(X=tx32, Y=ty32, w0=tw0, batches=out["batches"])
    for name, opt in opts.items():
        w, st, losses = tw0.copy(), so.optimiser_state(tn, opt), []
        for ids in batches:
            losses.append(so.train_step(ttab, w, st, tx32[:, ids], ty32[:, ids], opt))
        out32.update({name + "_w": w, name + "_loss": np.array(losses), name + "_m": st["m"], name + "_v": st["v"],
                      name + "_bp": np.array(st["bp"] if st["bp"] else [0.0, 0.0])})
    np.savez_compressed(os.path.join(HERE, "toy_train_steps_f32.npz"), **out32)
    print("wrote golden fixtures to", HERE)


if __name__ == "__main__":
    main()
