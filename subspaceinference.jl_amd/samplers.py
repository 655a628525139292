"""Gradient-based samplers of `sub_inference` (src/space_inference.jl:117-120 `:mala`, :139-160 `:hmc`) as HOST control
logic over a `logdensity_grad(z) -> (lp, grad)` callable.  In the product that callable is the device reverse sweep
(`Context.logdensity_grad`, kernels_bwd.hip); nothing numerical about the model happens here.

The reference delegates to AdvancedMH 0.6.2 (MALA) and AdvancedHMC 0.2.27 (Hamiltonian + StanHMCAdaptor), neither
vendored.  What is restated [upstream, unverifiable offline]:
  * MALA: `MALA(x -> MvNormal((σ_z^2/2) .* x, σ_z))` with `init_params = rand(MvNormal(zeros(M), σ_z))` -- the proposal
    is z' ~ N(z + (σ_z²/2)·∇lp(z), σ_z² I), accepted with the Metropolis–Hastings ratio including the asymmetric
    proposal densities; `itr` samples INCLUDING the initial state.
  * HMC: `StaticTrajectory(Leapfrog(ε), 1)` -- ONE leapfrog step per sample, `DiagEuclideanMetric(M)`, ε from
    `find_good_stepsize`, adapted for `n_adapts = round(itr/2)` iterations by `StanHMCAdaptor(MassMatrixAdaptor(metric),
    StepSizeAdaptor(0.8, integrator))`: Nesterov dual averaging of log ε to an acceptance of 0.8 on every adaptation
    step, plus Stan's WINDOWED diagonal mass-matrix adaptation (class StanAdaptor below: init buffer 75, terminal
    buffer 50, doubling windows from 25; Welford variance of the draws inside a window, regularised
    (n/(n+5))·var + 1e-3·5/(n+5), installed as M⁻¹ at the window's end together with a restart of the dual averaging
    at the current ε).  With the reference's default `itr` the windows only open when n_adapts > 125.
  * NUTS: `NUTS{MultinomialTS, GeneralisedNoUTurn}(Leapfrog(ε))` -- multinomial trajectory sampling with the
    generalised (momentum-sum) no-U-turn criterion, biased progressive sub-tree sampling, max depth 10, divergence
    threshold ΔH > 1000 (the published algorithm of Betancourt 2017 / Stan that AdvancedHMC implements); same step-size
    search and adaptor as above.
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


def _leapfrog(logdensity_grad, z, r, g, eps, minv=1.0):
    """One leapfrog step of H(z, r) = -lp(z) + r' M⁻¹ r / 2 with the diagonal M⁻¹ = minv."""
    r = r + 0.5 * eps * g
    z = z + eps * (minv * r)
    lp, g = logdensity_grad(z)
    r = r + 0.5 * eps * g
    return z, r, lp, g


def _kinetic(r, minv):
    return 0.5 * float(np.sum(minv * r * r))


def find_good_stepsize(logdensity_grad, z, lp, g, rng, eps=0.1, max_iter=100):
    """AdvancedHMC 0.2.27 `find_good_stepsize` (src/trajectory.jl; called at src/space_inference.jl:147) restated line by
    line [upstream, from memory, unverifiable offline] -- INCLUDING its two quirks:
      * identity metric (it runs before any adaptation), ONE momentum draw, start at eps = 0.1;
      * direction = +1 when the one-leapfrog acceptance exp(dH) at eps is above a_cross = 0.5 (NOT Stan's 0.8), else -1;
      * crossing loop: `eps' = direction == 1 ? 2 eps : eps / 2;  z', H' = A(h, z, eps)` -- the proposal is evaluated at
        the OLD eps, one step behind the candidate eps' (upstream's code, reproduced as is); break when exp(dH) is no
        longer on the starting side of a_cross, else eps = eps'.  Comparisons with a NaN dH are false, so a NaN energy
        makes direction = -1 at the start and BREAKS the loop at once (no extra handling, as upstream);
      * then (eps, eps') sorted and bisected until a_min = 0.25 <= exp(dH) <= a_max = 0.75 (a NaN lands in that branch and
        is accepted, as upstream); after max_iter bisections the lower end is returned.
    The result seeds the dual averaging (mu = log 10 eps), so it matters most for short runs (n_adapts = itr / 2)."""
    a_min, a_cross, a_max = 0.25, 0.5, 0.75
    r = rng.standard_normal(z.size)
    h0 = lp - 0.5 * float(r @ r)
    log_cross = math.log(a_cross)

    def delta_h(e):
        _, rp, lpp, _ = _leapfrog(logdensity_grad, z, r, g, e)
        return lpp - 0.5 * float(rp @ rp) - h0          # = H - H' : exp(d) is the MH ratio (may be NaN / -inf)

    def ratio(d):
        try:
            return math.exp(d)
        except OverflowError:
            return math.inf
    direction = 1 if delta_h(eps) > log_cross else -1
    eps_next = eps
    for _ in range(max_iter):
        eps_next = 2.0 * eps if direction == 1 else 0.5 * eps
        d = delta_h(eps)                               # upstream evaluates at eps, not eps' (sic)
        if direction == 1 and not d > log_cross:
            break
        if direction == -1 and not d < log_cross:
            break
        eps = eps_next
    lo, hi = (eps, eps_next) if eps < eps_next else (eps_next, eps)
    for _ in range(max_iter):
        mid = 0.5 * (lo + hi)
        a = ratio(delta_h(mid))
        if a > a_max:
            lo = mid
        elif a < a_min:
            hi = mid
        else:
            lo = mid
            break
    return lo


class StanAdaptor:
    """`StanHMCAdaptor(MassMatrixAdaptor(DiagEuclideanMetric), StepSizeAdaptor(δ, integrator))` restated from the
    published algorithm (Stan reference manual, "Automatic Parameter Tuning"; stan/mcmc/windowed_adaptation.hpp and
    var_adaptation.hpp, which AdvancedHMC's stan_adaptor.jl cites) [upstream, unverifiable offline].

    Step size: Nesterov dual averaging (γ = 0.05, t0 = 10, κ = 0.75, μ = log 10ε) on every adaptation step, restarted at
    the current ε whenever the metric changes, frozen to exp(x̄) after the last step.  Metric: the draws of a window feed
    a Welford variance; at the window's end M⁻¹ <- (n/(n+5))·var + 1e-3·5/(n+5) and the estimator is reset.  Windows:
    the first opens after `init_buffer` steps, the last closes `term_buffer` steps before the end, sizes double
    (25, 50, 100, ...) and the last one is stretched to the end of the slow phase.  When the three phases do not fit
    (n_adapts < 150) Stan rescales them to 15 % / 75 % / 10 % if n_adapts >= 20, otherwise there is no metric window.
    """

    def __init__(self, m, n_adapts, eps, delta=0.8, init_buffer=75, term_buffer=50, window_size=25):
        self.m, self.n_adapts, self.delta = m, int(n_adapts), delta
        if init_buffer + window_size + term_buffer > self.n_adapts:
            if self.n_adapts >= 20:
                init_buffer = int(0.15 * self.n_adapts)
                term_buffer = int(0.1 * self.n_adapts)
                window_size = self.n_adapts - init_buffer - term_buffer
            else:
                init_buffer, term_buffer, window_size = self.n_adapts + 1, 0, 1   # never inside a window
        self.window_start = init_buffer + 1
        self.window_end = self.n_adapts - term_buffer
        self.window_splits = []
        nxt = init_buffer + window_size
        while nxt <= self.window_end:
            if nxt + 2 * window_size > self.window_end:
                nxt = self.window_end
            self.window_splits.append(nxt)
            window_size *= 2
            nxt += window_size
        self.i = 0
        self.minv = np.ones(m)
        self.eps = eps
        self._restart_da(eps)
        self._wn, self._wmean, self._wm2 = 0, np.zeros(m), np.zeros(m)

    def _restart_da(self, eps):
        self.mu, self.hbar, self.log_eps_bar, self.t = math.log(10.0 * eps), 0.0, 0.0, 0

    def adapt(self, z, accept):
        """After adaptation step i (1-based): returns True when the metric changed."""
        if self.i >= self.n_adapts:
            return False
        self.i += 1
        gamma, t0, kappa = 0.05, 10.0, 0.75
        self.t += 1
        a = min(1.0, accept)
        self.hbar = (1.0 - 1.0 / (self.t + t0)) * self.hbar + (self.delta - a) / (self.t + t0)
        log_eps = self.mu - math.sqrt(self.t) / gamma * self.hbar
        eta = self.t ** (-kappa)
        self.log_eps_bar = eta * log_eps + (1.0 - eta) * self.log_eps_bar
        self.eps = math.exp(min(log_eps, 700.0))   # (a diverging dual average must not raise)
        changed = False
        if self.window_start <= self.i <= self.window_end:
            self._wn += 1
            dlt = z - self._wmean
            self._wmean = self._wmean + dlt / self._wn
            self._wm2 = self._wm2 + dlt * (z - self._wmean)
            if self.i in self.window_splits:
                n = self._wn
                if n >= 2:
                    var = self._wm2 / (n - 1)
                    self.minv = (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0))
                    changed = True
                self._wn, self._wmean, self._wm2 = 0, np.zeros(self.m), np.zeros(self.m)
                self._restart_da(self.eps)
        if self.i == self.n_adapts:
            if self.t > 0:
                self.eps = math.exp(min(self.log_eps_bar, 700.0))
        return changed


def hmc(logdensity_grad, m, itr, sigma_z, rng, delta=0.8):
    z = sigma_z * rng.standard_normal(m)  # initial_theta = rand(MvNormal(zeros(M), sigma_z)), space_inference.jl:140
    lp, g = logdensity_grad(z)
    ad = StanAdaptor(m, int(round(itr / 2)), find_good_stepsize(logdensity_grad, z, lp, g, rng), delta)
    zs = np.empty((m, itr), order="F")
    lps = np.empty(itr)
    acc = np.empty(itr)
    for t in range(itr):
        minv, eps = ad.minv, ad.eps
        r = rng.standard_normal(m) / np.sqrt(minv)          # r ~ N(0, M)
        h0 = lp - _kinetic(r, minv)
        zp, rp, lpp, gp = _leapfrog(logdensity_grad, z, r, g, eps, minv)
        h1 = lpp - _kinetic(rp, minv)
        # (min(1, exp(dH)) without forming exp of a large positive dH: math.exp raises OverflowError above ~709)
        a = (1.0 if h1 - h0 >= 0.0 else math.exp(h1 - h0)) if np.isfinite(h1) else 0.0
        if rng.random() < a:
            z, lp, g = zp, lpp, gp
        zs[:, t], lps[t], acc[t] = z, lp, a
        ad.adapt(z, a)
    return zs, lps, float(acc.mean())


def _uturn(rho, v_minus, v_plus):
    """Generalised no-U-turn criterion (Betancourt 2017): rho = sum of momenta, v = M⁻¹ r at the two ends."""
    return float(rho @ v_minus) <= 0.0 or float(rho @ v_plus) <= 0.0


def nuts(logdensity_grad, m, itr, sigma_z, rng, delta=0.8, max_depth=10, max_dh=1000.0):
    z = sigma_z * rng.standard_normal(m)
    lp, g = logdensity_grad(z)
    ad = StanAdaptor(m, int(round(itr / 2)), find_good_stepsize(logdensity_grad, z, lp, g, rng), delta)
    zs = np.empty((m, itr), order="F")
    lps = np.empty(itr)
    acc = np.empty(itr)

    def build(zc, rc, gc, v, depth, h0, eps, minv):
        """2**depth leapfrog steps from (zc, rc) in direction v.  Returns the sub-tree summary."""
        if depth == 0:
            z1, r1, lp1, g1 = _leapfrog(logdensity_grad, zc, rc, gc, v * eps, minv)
            h1 = lp1 - _kinetic(r1, minv)
            if not np.isfinite(h1):
                h1 = -np.inf
            dh = h1 - h0
            return dict(zm=z1, rm=r1, gm=g1, zp=z1, rp=r1, gp=g1, zprop=z1, lpprop=lp1, gprop=g1, logw=dh, rho=r1.copy(),
                        alpha=min(1.0, math.exp(dh)) if dh < 0 else 1.0, n=1, stop=(-dh) > max_dh)
        a = build(zc, rc, gc, v, depth - 1, h0, eps, minv)
        if a["stop"]:
            return a
        if v > 0:
            b = build(a["zp"], a["rp"], a["gp"], v, depth - 1, h0, eps, minv)
        else:
            b = build(a["zm"], a["rm"], a["gm"], v, depth - 1, h0, eps, minv)
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
        out["stop"] = b["stop"] or _uturn(out["rho"], minv * out["rm"], minv * out["rp"])
        return out

    for t in range(itr):
        minv, eps = ad.minv, ad.eps
        r0 = rng.standard_normal(m) / np.sqrt(minv)
        h0 = lp - _kinetic(r0, minv)
        tree = dict(zm=z, rm=r0, gm=g, zp=z, rp=r0, gp=g, rho=r0.copy(), logw=0.0)
        zn, lpn, gn = z, lp, g
        alpha_sum, n_alpha = 0.0, 0
        for depth in range(max_depth):
            v = 1.0 if rng.random() < 0.5 else -1.0
            if v > 0:
                sub = build(tree["zp"], tree["rp"], tree["gp"], v, depth, h0, eps, minv)
            else:
                sub = build(tree["zm"], tree["rm"], tree["gm"], v, depth, h0, eps, minv)
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
            if _uturn(tree["rho"], minv * tree["rm"], minv * tree["rp"]):
                break
        z, lp, g = zn, lpn, gn
        a = alpha_sum / max(1, n_alpha)
        zs[:, t], lps[t], acc[t] = z, lp, a
        ad.adapt(z, a)
    return zs, lps, float(acc.mean())
