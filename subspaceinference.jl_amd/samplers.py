"""Gradient-based samplers of `sub_inference` (src/space_inference.jl:117-120 `:mala`, :139-160 `:hmc`) as HOST control
logic over a `logdensity_grad(z) -> (lp, grad)` callable.  In the product that callable is the device reverse sweep
(`Context.logdensity_grad`, kernels_bwd.hip); nothing numerical about the model happens here.

The reference delegates to AdvancedMH 0.6.2 (MALA) and AdvancedHMC 0.2.27 (Hamiltonian + StanHMCAdaptor), neither
vendored.  What is restated [upstream, unverifiable offline]:
  * MALA: `MALA(x -> MvNormal((σ_z^2/2) .* x, σ_z))` with `init_params = rand(MvNormal(zeros(M), σ_z))` -- the proposal
    is z' ~ N(z + (σ_z²/2)·∇lp(z), σ_z² I), accepted with the Metropolis–Hastings ratio including the asymmetric
    proposal densities; `itr` samples INCLUDING the initial state.
  * HMC: `StaticTrajectory(Leapfrog(ε), 1)` -- ONE leapfrog step per sample, `DiagEuclideanMetric(M)`, ε from
    `find_good_stepsize`, adapted for `n_adapts = round(itr/2)` iterations by dual averaging to an acceptance of 0.8.
    AdvancedHMC's windowed mass-matrix adaptation (StanHMCAdaptor) is NOT restated: the metric stays the identity
    (documented deviation).  `:nuts` is not built.
The random stream is NumPy's PCG64 seeded by the caller (the reference uses Julia's global MersenneTwister).
"""
import math

import numpy as np


def mala(logdensity_grad, m, itr, sigma_z, rng):
    z = sigma_z * rng.standard_normal(m)
    lp, g = logdensity_grad(z)
    zs = np.empty((m, itr), order="F")
    lps = np.empty(itr)
    zs[:, 0], lps[0] = z, lp
    nacc = 0
    h = 0.5 * sigma_z * sigma_z
    inv2s2 = 1.0 / (2.0 * sigma_z * sigma_z)
    for t in range(1, itr):
        zp = z + h * g + sigma_z * rng.standard_normal(m)
        lpp, gp = logdensity_grad(zp)
        # log q(z | z') - log q(z' | z)
        fwd = zp - z - h * g
        bwd = z - zp - h * gp
        logq = -inv2s2 * (float(bwd @ bwd) - float(fwd @ fwd))
        if -rng.exponential() < lpp - lp + logq:
            z, lp, g = zp, lpp, gp
            nacc += 1
        zs[:, t], lps[t] = z, lp
    return zs, lps, nacc / max(1, itr - 1)


def _leapfrog(logdensity_grad, z, r, g, eps):
    r = r + 0.5 * eps * g
    z = z + eps * r
    lp, g = logdensity_grad(z)
    r = r + 0.5 * eps * g
    return z, r, lp, g


def find_good_stepsize(logdensity_grad, z, lp, g, rng, eps=0.1, max_iter=100):
    """Hoffman & Gelman (2014) Algorithm 4, the heuristic AdvancedHMC's `find_good_stepsize` implements."""
    r = rng.standard_normal(z.size)
    h0 = lp - 0.5 * float(r @ r)
    _, rp, lpp, _ = _leapfrog(logdensity_grad, z, r, g, eps)
    dh = lpp - 0.5 * float(rp @ rp) - h0
    direction = 1.0 if dh > math.log(0.8) else -1.0
    for _ in range(max_iter):
        eps *= 2.0 ** direction
        _, rp, lpp, _ = _leapfrog(logdensity_grad, z, r, g, eps)
        dh = lpp - 0.5 * float(rp @ rp) - h0
        if not np.isfinite(dh):
            dh = -np.inf
        if (direction > 0 and dh <= math.log(0.8)) or (direction < 0 and dh >= math.log(0.8)):
            break
    return eps


def hmc(logdensity_grad, m, itr, sigma_z, rng, delta=0.8):
    z = sigma_z * rng.standard_normal(m)  # initial_theta = rand(MvNormal(zeros(M), sigma_z)), space_inference.jl:140
    lp, g = logdensity_grad(z)
    eps = find_good_stepsize(logdensity_grad, z, lp, g, rng)
    n_adapts = int(round(itr / 2))
    # dual averaging (Nesterov), Stan's constants
    mu, gamma, t0, kappa = math.log(10.0 * eps), 0.05, 10.0, 0.75
    hbar, log_eps_bar = 0.0, 0.0
    zs = np.empty((m, itr), order="F")
    lps = np.empty(itr)
    acc = np.empty(itr)
    for t in range(itr):
        r = rng.standard_normal(m)
        h0 = lp - 0.5 * float(r @ r)
        zp, rp, lpp, gp = _leapfrog(logdensity_grad, z, r, g, eps)
        h1 = lpp - 0.5 * float(rp @ rp)
        a = min(1.0, math.exp(h1 - h0)) if np.isfinite(h1) else 0.0
        if rng.random() < a:
            z, lp, g = zp, lpp, gp
        zs[:, t], lps[t], acc[t] = z, lp, a
        if t < n_adapts:
            it = t + 1
            hbar = (1.0 - 1.0 / (it + t0)) * hbar + (delta - a) / (it + t0)
            log_eps = mu - math.sqrt(it) / gamma * hbar
            eta = it ** (-kappa)
            log_eps_bar = eta * log_eps + (1.0 - eta) * log_eps_bar
            eps = math.exp(log_eps)
            if it == n_adapts:
                eps = math.exp(log_eps_bar)
    return zs, lps, float(acc.mean())
