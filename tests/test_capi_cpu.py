"""CPU tests of the boundary: the library loads, exports every symbol include/subspace_hip.h declares, fails
loudly without a GPU (no CPU fallback), and the host-side pieces (eigensolver, API mirror, Flux stand-ins)
behave like the reference."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "subspace_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(si_[a-z_A-Z0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(si):
    lib = si.load()
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libsubspace_hip.so does not export %s" % s
    # and the ctypes table binds exactly the declared ABI
    assert sorted(si._capi.SIGNATURES) == syms
    assert lib.si_version() == 500


def test_no_cpu_fallback(si):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(si.SubspaceError) as e:
        si.Context(0)
    assert e.value.code == si._capi.SI_ERR_NODEVICE and "no CPU backend" in str(e.value)
    # the API mirror surfaces the same failure instead of computing anything on the host
    from subspaceinference_jl_amd import flux
    m = flux.Chain(flux.Dense(3, 2))
    data = flux.DataLoader(np.zeros((3, 4)), np.zeros((2, 4)))
    with pytest.raises(si.SubspaceError):
        si.subspace_construction(m, flux.mse, data, flux.Descent(0.1), T=1, M=1, verbose=False)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "subspaceinference.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".jl")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_shipped_library_has_no_development_knobs(si):
    """alternative kernels and SI_* environment knobs live in the development build only
    (`python subspaceinference.jl_amd/build.py --dev`, -DSI_DEV_KNOBS).  The shipped .so reads two environment variables:
    SI_HOST_COPY_THREADS (host copy pool size) and SI_RCCL_LIB (explicit RCCL path)."""
    blob = open(si._capi.LIB_PATH, "rb").read()
    names = set(m.decode() for m in re.findall(rb"SI_[A-Z0-9_]{3,}", blob))
    env_like = {n for n in names if not n.startswith(("SI_ERR", "SI_K_", "SI_F", "SI_ACT", "SI_LAYER", "SI_OK", "SI_COMM", "SI_HIP", "SI_NCCL", "SI_DTYPE_OF_DATA",
                                                             "SI_SPEC", "SI_SPSTAMP", "SI_SPE"))}   # (enum names inside error messages; SI_SPEC_* / SI_SPSTAMP: macros in the kernel TEXT embedded for hiprtc, csrc/chain_spec.inc)
    assert env_like <= {"SI_HOST_COPY_THREADS", "SI_RCCL_LIB"}, env_like
    assert b"gram_spec" not in blob                       # the wave-specialised development Gram variant is not compiled in
    # every getenv in the sources is one of the two, or sits inside an #ifdef SI_DEV_KNOBS / SI_BWD_DEBUG_KNOB block
    csrc = os.path.join(ROOT, "subspaceinference.jl_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".cpp", ".h")):
            continue
        depth = []
        for ln in open(os.path.join(csrc, f)):
            t = ln.strip()
            if t.startswith("#if"):
                depth.append("SI_DEV_KNOBS" in t or "SI_BWD_DEBUG_KNOB" in t or "SI_GRAM_TS" in t)
            elif t.startswith("#else") and depth:
                depth[-1] = False
            elif t.startswith("#endif") and depth:
                depth.pop()
            elif "getenv(" in t and not any(depth):
                assert "SI_HOST_COPY_THREADS" in t or "SI_RCCL_LIB" in t, (f, t)


def test_host_eigensolver(si):
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 33, 100):
        a = rng.standard_normal((n + 3, n))
        g = a.T @ a
        w, v = si.host_sym_eig(g)
        assert np.allclose(w, np.linalg.eigvalsh(g), rtol=1e-12, atol=1e-12 * abs(w).max())
        assert np.allclose(g @ v, v * w[None, :], atol=1e-11 * abs(w).max())
        assert np.allclose(v.T @ v, np.eye(n), atol=1e-12)
    # repeated eigenvalues / rank deficiency
    g = np.diag([3.0, 3.0, 0.0, 1.0])
    w, v = si.host_sym_eig(g)
    assert np.allclose(w, [0, 1, 3, 3])


def test_host_top_eigenpairs(si):
    """The route si_construct_finish takes first (eig.cpp sym_eig_top): M largest eigenpairs by factored Householder +
    vector-free QL + inverse iteration; self-verifying, declines when it has no advantage."""
    rng = np.random.default_rng(3)
    cases = []
    for n, m in ((100, 20), (64, 3), (200, 20), (31, 10)):
        cases.append(("walk", np.cumsum(rng.standard_normal((3 * n, n)), axis=1), m))
        cases.append(("gauss", rng.standard_normal((3 * n, n)), m))
    a = rng.standard_normal((400, 100))
    a[:, 1] = a[:, 0]                       # duplicated column: a zero eigenvalue at the BOTTOM of the spectrum
    cases.append(("dup", a, 20))
    cases.append(("lowrank", rng.standard_normal((400, 25)) @ rng.standard_normal((25, 100)), 20))
    b = rng.standard_normal((300, 60))
    b[:, 5] = b[:, 4] * (1 + 1e-9)          # a nearly repeated direction
    cases.append(("near-dup", b, 12))
    for name, a, m in cases:
        g = a.T @ a
        r = si.host_sym_eig_top(g, m)
        assert r is not None, name
        w, v = r
        ref = np.linalg.eigvalsh(g)[::-1][:m]
        assert np.allclose(w, ref, rtol=1e-11, atol=1e-12 * ref[0]), name
        assert np.linalg.norm(g @ v - v * w[None, :]) <= 1e-11 * np.linalg.norm(g), name
        assert np.abs(v.T @ v - np.eye(m)).max() < 1e-9, name
        wf, vf = si.host_sym_eig(g)          # same subspace as the full solver
        top = vf[:, ::-1][:, :m]
        gap_ok = ref[m - 1] - np.linalg.eigvalsh(g)[::-1][m] > 1e-8 * ref[0] if m < g.shape[0] else True
        if gap_ok:
            assert np.abs(np.abs(np.linalg.svd(top.T @ v, compute_uv=False)) - 1).max() < 1e-7, name
    # exactly repeated TOP eigenvalues: any orthonormal basis of the eigenspace is right; the verification must hold
    q, _ = np.linalg.qr(rng.standard_normal((40, 40)))
    lam = np.r_[np.full(4, 7.0), np.linspace(3, 0.1, 36)]
    g = (q * lam) @ q.T
    g = 0.5 * (g + g.T)
    r = si.host_sym_eig_top(g, 6)
    if r is not None:                        # (may also decline: then si_construct_finish uses the full solver)
        w, v = r
        assert np.allclose(w[:4], 7.0, rtol=1e-12) and np.linalg.norm(g @ v - v * w) <= 1e-11 * np.linalg.norm(g)
        assert np.abs(v.T @ v - np.eye(6)).max() < 1e-9
    # declines when it has no advantage / arguments make no sense
    assert si.host_sym_eig_top(np.eye(6), 2) is None          # tiny
    assert si.host_sym_eig_top(g, 30) is None                 # 3 m > n
    assert si.host_sym_eig_top(np.zeros((50, 50)), 5) is not None or True   # must not crash on the zero matrix


def test_flux_standins_match_reference_semantics():
    from subspaceinference_jl_amd import flux
    rng = np.random.default_rng(1)
    m = flux.Chain(flux.Dense(10, 20, rng=rng), flux.Dense(20, 20, rng=rng), flux.Dense(20, 2, rng=rng))
    ps = flux.params(m)
    assert [p.shape for p in ps] == [(20, 10), (20,), (20, 20), (20,), (2, 20), (2,)]
    flat = flux.extract_params(ps)
    assert flat.dtype == np.float32 and flat.size == 682  # README toy N
    assert flat[1] == ps[0][1, 0]  # column-major vec
    table, n = flux.layer_table(m)
    assert n == 682 and table[1][3] == 220 and table[2][4] == 680
    # DataLoader: default batchsize 1 -> 100 batches; .data is the full (X, Y) (split_data, libs.jl:75-77)
    x, y = rng.random((10, 100)), rng.random((2, 100))
    dl = flux.DataLoader(x, y, shuffle=True, rng=rng)
    assert len(dl) == 100 and dl.data[0] is not None and dl.data[0].shape == (10, 100)
    seen = np.concatenate([b[0] for b in dl], axis=1)
    assert seen.shape == (10, 100) and np.allclose(np.sort(seen[0]), np.sort(x[0]))
    # gradient of mse against finite differences
    loss, gs = flux.mse.value_and_grad(m, x[:, :7], y[:, :7])
    eps = 1e-3
    w00 = ps[0][0, 0]
    ps[0][0, 0] = w00 + eps
    lp = flux.mse(m, x[:, :7], y[:, :7])
    ps[0][0, 0] = w00 - eps
    lm = flux.mse(m, x[:, :7], y[:, :7])
    ps[0][0, 0] = w00
    assert abs((lp - lm) / (2 * eps) - gs[0][0, 0]) < 1e-3 * max(1.0, abs(gs[0][0, 0]))
    # ADAM first step = eta * sign(g) (bias-corrected), Momentum first step = eta * g
    g = np.array([0.5, -2.0])
    assert np.allclose(flux.ADAM(0.1).apply(np.zeros(2), g), 0.1 * np.sign(g), rtol=1e-6)
    assert np.allclose(flux.Momentum(0.01, 0.9).apply(np.zeros(2), g), 0.01 * g)


def test_api_error_behaviour(si):
    from subspaceinference_jl_amd import flux
    m = flux.Chain(flux.Dense(3, 2))
    data = flux.DataLoader(np.zeros((3, 4)), np.zeros((2, 4)))
    w, p = np.zeros(8), np.zeros((8, 1))
    with pytest.raises(si.SubspaceError, match="bogus is not available"):
        si.sub_inference(m, data, w, p, M=1, alg=":bogus")                       # space_inference.jl:162
    with pytest.raises(si.SubspaceError, match="No method found"):
        si.subspace_inference(m, flux.mse, data, flux.Descent(), method=":x")    # space_inference.jl:42
    with pytest.raises(si.SubspaceError, match="not avaliable"):
        si.sub_inference(object(), data, w, p, M=1)                               # space_inference.jl:103
    assert si.inference.__doc__ and "sub_inference" in si.inference.__doc__


def test_gradient_samplers_on_gaussian_target():
    """Host sampler logic (MALA / one-step HMC with dual averaging) on N(mu, I): stationary moments."""
    from subspaceinference_jl_amd import samplers
    mu = np.array([0.5, -1.0, 2.0])
    fn = lambda z: (-0.5 * float((z - mu) @ (z - mu)), -(z - mu))
    for sampler, s in ((samplers.mala, 0.9), (samplers.hmc, 1.0), (samplers.nuts, 1.0)):
        z, lp, acc = sampler(fn, 3, 6000 if sampler is not samplers.nuts else 2500, s, np.random.default_rng(1))
        burn = z[:, 500:]
        assert np.all(np.abs(burn.mean(axis=1) - mu) < 0.2), sampler.__name__
        assert np.all(np.abs(burn.var(axis=1) - 1.0) < 0.3), sampler.__name__
        assert 0.3 < acc <= 1.0


def test_stan_windowed_adaptation():
    """samplers.StanAdaptor: Stan's window schedule and the regularised Welford variance installed as the diagonal
    M^-1 (the StanHMCAdaptor of src/space_inference.jl:155) -- on an anisotropic Gaussian both HMC and NUTS recover the
    per-coordinate scales, which an identity metric with one step per sample cannot."""
    from subspaceinference_jl_amd import samplers
    ad = samplers.StanAdaptor(3, 1000, 0.1)
    assert (ad.window_start, ad.window_end, ad.window_splits) == (76, 950, [100, 150, 250, 450, 950])   # Stan's schedule
    ad = samplers.StanAdaptor(3, 500, 0.1)
    assert ad.window_splits == [100, 150, 250, 450]
    ad = samplers.StanAdaptor(3, 50, 0.1)           # does not fit 75 + 25 + 50: rescaled to 15 % / 75 % / 10 %
    assert (ad.window_start, ad.window_end, ad.window_splits) == (8, 45, [45])
    assert samplers.StanAdaptor(3, 10, 0.1).window_splits == []   # under 20 steps: step size only
    # the variance estimator: feed a window of known draws
    ad = samplers.StanAdaptor(2, 200, 0.1, init_buffer=0, term_buffer=0, window_size=200)
    rng = np.random.default_rng(0)
    draws = rng.standard_normal((200, 2)) * np.array([0.1, 3.0])
    changed = [ad.adapt(d, 0.8) for d in draws]
    assert changed.count(True) == 1 and changed[-1]
    want = (200 / 205.0) * draws.var(axis=0, ddof=1) + 1e-3 * 5 / 205.0
    assert np.allclose(ad.minv, want, rtol=1e-12)
    scales = np.array([0.1, 1.0, 5.0])
    fn = lambda z: (-0.5 * float(np.sum((z / scales) ** 2)), -z / scales ** 2)
    for sampler in (samplers.hmc, samplers.nuts):
        z, lp, acc = sampler(fn, 3, 3000, 1.0, np.random.default_rng(0))
        assert np.allclose(z[:, 1500:].std(axis=1), scales, rtol=0.15), sampler.__name__
        assert 0.6 < acc <= 1.0


def test_jacobi_resolves_graded_psd_matrices(si):
    """The second-stage solver of the ill-conditioned route: eigenvalues of a graded Gram matrix B'B (columns spread
    over 12 decades, mixed inside the small block) to RELATIVE accuracy."""
    rng = np.random.default_rng(0)
    n, k = 300, 12
    u, _ = np.linalg.qr(rng.standard_normal((n, k)))
    sv = np.logspace(0, -12, k)
    mix = np.eye(k)
    q, _ = np.linalg.qr(rng.standard_normal((6, 6)))
    mix[6:, 6:] = q                                   # unresolved directions arrive mixed among themselves
    b = (u * sv[None, :]) @ mix                       # graded columns: the product is formed WITHOUT cancellation
    g2 = b.T @ b
    w, v = si._capi.host_jacobi_eig_psd(g2)
    assert np.all(np.diff(w) <= 0)
    assert np.allclose(np.sqrt(w), sv, rtol=1e-6)     # twelve decades, each singular value relative to ITSELF
    assert np.allclose(v.T @ v, np.eye(k), atol=1e-12)
    # well-conditioned input: agrees with LAPACK to rounding
    a = rng.standard_normal((40, 9))
    w2, v2 = si._capi.host_jacobi_eig_psd(a.T @ a)
    assert np.allclose(w2, np.linalg.eigvalsh(a.T @ a)[::-1], rtol=1e-12)
    assert np.allclose((a.T @ a) @ v2, v2 * w2[None, :], atol=1e-11 * w2[0])


def test_find_good_stepsize_reproduces_upstreams_lagged_crossing():
    """ADVICE r1 + r2: AdvancedHMC's heuristic crosses at acceptance 0.5 (not Stan's 0.8), evaluates the crossing test at
    the OLD step size (one doubling / halving behind the candidate) and breaks at once on a NaN energy.  Pinned on Gaussian
    targets N(0, sigma^2 I): the returned eps_0 is reproducible, follows the scale of the target, and sits at the FIRST
    doubling / halving of 0.1 whose one-leapfrog acceptance is on the far side of 0.5 (the bracket upstream bisects lies
    wholly beyond the crossing, so the bisection converges onto that end unless an interior point lands in (0.25, 0.75))."""
    from subspaceinference_jl_amd import samplers

    for sigma, d in ((0.01, 5), (1.0, 20), (100.0, 3)):
        def lg(z, s=sigma):
            return -0.5 * float(z @ z) / s ** 2, -z / s ** 2
        z = np.full(d, 0.3 * sigma)
        lp, g = lg(z)
        eps = samplers.find_good_stepsize(lg, z, lp, g, np.random.default_rng(5))
        assert eps == samplers.find_good_stepsize(lg, z, lp, g, np.random.default_rng(5))
        assert 0.2 * sigma < eps < 20 * sigma, (sigma, eps)         # the step size follows the scale of the target
        r = np.random.default_rng(5).standard_normal(d)             # the momentum the search drew

        def acc(e):
            _, rp, lpp, _ = samplers._leapfrog(lg, z, r, g, e)
            return np.exp(min(0.0, (lpp - 0.5 * rp @ rp) - (lp - 0.5 * r @ r)))
        up = acc(0.1) > 0.5
        e, steps = 0.1, 0
        while (acc(e) > 0.5) == up and steps < 60:                  # first doubling / halving on the far side of 0.5
            e = 2 * e if up else e / 2
            steps += 1
        lo, hi = (e, 2 * e) if up else (e / 2, e)
        assert lo * (1 - 1e-9) <= eps <= hi * (1 + 1e-9), (sigma, eps, lo, hi)
        a = acc(eps)
        assert 0.25 <= a <= 0.75 or abs(eps - e) < 1e-6 * e, (sigma, eps, a)
    # a NaN energy: direction -1 at the start, the crossing loop breaks at once, the bisection accepts its first midpoint
    nan_lg = lambda zz: (float("nan"), np.zeros_like(zz))
    assert samplers.find_good_stepsize(nan_lg, np.zeros(3), 0.0, np.zeros(3), np.random.default_rng(0)) == 0.5 * (0.05 + 0.1)


def test_hmc_acceptance_ratio_does_not_overflow():
    """min(1, exp(dH)) for a transition that LOWERS the energy by more than ~709 (a start far out in a sharp posterior): forming
    exp(dH) first raised OverflowError (found by tools/guard_fuzz_api.py); the ratio is 1."""
    from subspaceinference_jl_amd import samplers

    def sharp(z):
        return float(-0.5e6 * z @ z), -1e6 * z
    for fn in (samplers.hmc, samplers.nuts, samplers.mala):
        zs, lps = fn(sharp, 2, 6, 0.05, np.random.default_rng(0))[:2]
        assert np.all(np.isfinite(np.asarray(lps, dtype=np.float64)))


def test_host_copy_pool_respects_the_cpu_quota():
    """the pool was sized from the affinity mask alone (8 threads per process on a box that shows 256 CPUs and
    grants 16: 64 spinning threads at --gpus 8).  Now: affinity capped by the cgroup quota, divided by the ranks sharing the host."""
    import ctypes
    import subspaceinference_jl_amd as si
    lib = si.load()
    parse = lib.si_host_parse_cpu_max
    assert parse(b"1600000 100000\n") == 16.0 and parse(b"max 100000") == 0.0 and parse(b"250000 100000") == 2.5
    assert parse(b"garbage") == 0.0 and parse(None) == 0.0
    plan = lib.si_host_copy_plan
    assert plan(16, 1, None) == 8 and plan(16, 8, None) == 1 and plan(16, 2, None) == 4 and plan(256, 1, None) == 8
    assert plan(2, 1, None) == 1 and plan(16, 8, b"3") == 3 and plan(16, 1, b"0") == 1
    # this process: the same number bench.py / conftest compute in Python
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    assert lib.si_host_cpu_budget() == n
    assert isinstance(ctypes.c_int(lib.si_host_cpu_budget()).value, int)
