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
    (documented deviation).
  * NUTS: `NUTS{MultinomialTS, GeneralisedNoUTurn}(Leapfrog(ε))` -- multinomial trajectory sampling with the
    generalised (momentum-sum) no-U-turn criterion, biased progressive sub-tree sampling, max depth 10, divergence
    threshold ΔH > 1000 (the published algorithm of Betancourt 2017 / Stan that AdvancedHMC implements); same step-size
    search and dual averaging as above, identity metric.
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


def _uturn(rho, r_minus, r_plus):
    return float(rho @ r_minus) <= 0.0 or float(rho @ r_plus) <= 0.0


def nuts(logdensity_grad, m, itr, sigma_z, rng, delta=0.8, max_depth=10, max_dh=1000.0):
    z = sigma_z * rng.standard_normal(m)
    lp, g = logdensity_grad(z)
    eps = find_good_stepsize(logdensity_grad, z, lp, g, rng)
    n_adapts = int(round(itr / 2))
    mu, gamma, t0, kappa = math.log(10.0 * eps), 0.05, 10.0, 0.75
    hbar, log_eps_bar = 0.0, 0.0
    zs = np.empty((m, itr), order="F")
    lps = np.empty(itr)
    acc = np.empty(itr)

    def build(zc, rc, gc, v, depth, h0):
        """2**depth leapfrog steps from (zc, rc) in direction v.  Returns the sub-tree summary."""
        if depth == 0:
            z1, r1, lp1, g1 = _leapfrog(logdensity_grad, zc, rc, gc, v * eps)
            h1 = lp1 - 0.5 * float(r1 @ r1)
            if not np.isfinite(h1):
                h1 = -np.inf
            dh = h1 - h0
            return dict(zm=z1, rm=r1, gm=g1, zp=z1, rp=r1, gp=g1, zprop=z1, lpprop=lp1, gprop=g1, logw=dh, rho=r1.copy(),
                        alpha=min(1.0, math.exp(dh)) if dh < 0 else 1.0, n=1, stop=(-dh) > max_dh)
        a = build(zc, rc, gc, v, depth - 1, h0)
        if a["stop"]:
            return a
        if v > 0:
            b = build(a["zp"], a["rp"], a["gp"], v, depth - 1, h0)
        else:
            b = build(a["zm"], a["rm"], a["gm"], v, depth - 1, h0)
        logw = np.logaddexp(a["logw"], b["logw"])
        out = dict(a)
        if v > 0:
            out.update(zp=b["zp"], rp=b["rp"], gp=b["gp"])
        else:
            out.update(zm=b["zm"], rm=b["rm"], gm=b["gm"])
        if not b["stop"] and math.log(rng.random()) < b["logw"] - logw:   # multinomial sampling inside the sub-tree
            out.update(zprop=b["zprop"], lpprop=b["lpprop"], gprop=b["gprop"])
        out["rho"] = a["rho"] + b["rho"]
        out["logw"] = logw
        out["alpha"] = a["alpha"] + b["alpha"]
        out["n"] = a["n"] + b["n"]
        out["stop"] = b["stop"] or _uturn(out["rho"], out["rm"], out["rp"])
        return out

    for t in range(itr):
        r0 = rng.standard_normal(m)
        h0 = lp - 0.5 * float(r0 @ r0)
        tree = dict(zm=z, rm=r0, gm=g, zp=z, rp=r0, gp=g, rho=r0.copy(), logw=0.0)
        zn, lpn, gn = z, lp, g
        alpha_sum, n_alpha = 0.0, 0
        for depth in range(max_depth):
            v = 1.0 if rng.random() < 0.5 else -1.0
            if v > 0:
                sub = build(tree["zp"], tree["rp"], tree["gp"], v, depth, h0)
            else:
                sub = build(tree["zm"], tree["rm"], tree["gm"], v, depth, h0)
            alpha_sum += sub["alpha"]
            n_alpha += sub["n"]
            if sub["stop"]:
                break
            if math.log(rng.random()) < sub["logw"] - tree["logw"]:          # biased progressive sampling
                zn, lpn, gn = sub["zprop"], sub["lpprop"], sub["gprop"]
            if v > 0:
                tree.update(zp=sub["zp"], rp=sub["rp"], gp=sub["gp"])
            else:
                tree.update(zm=sub["zm"], rm=sub["rm"], gm=sub["gm"])
            tree["rho"] = tree["rho"] + sub["rho"]
            tree["logw"] = np.logaddexp(tree["logw"], sub["logw"])
            if _uturn(tree["rho"], tree["rm"], tree["rp"]):
                break
        z, lp, g = zn, lpn, gn
        a = alpha_sum / max(1, n_alpha)
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
