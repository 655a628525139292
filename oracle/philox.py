"""Philox4x32-10 counter RNG (Salmon et al., SC'11) in NumPy -- TEST INFRASTRUCTURE ONLY.

This is the CPU twin of `subspaceinference.jl_amd/csrc/philox.h`.  The reference draws
its proposals from Julia's global MersenneTwister (`rand(rng, proposal)` /
`randexp(rng)` inside AdvancedMH 0.6.2, called from /root/reference
src/space_inference.jl:113-116), whose stream cannot be reproduced on a GPU.  The build
therefore defines its OWN random stream (below) and the oracle restates it bit for bit so
that oracle and HIP chains can be compared step by step.

Stream definition (shared with the device code):
  key      = (seed & 0xffffffff, seed >> 32)
  counter  = (step & 0xffffffff, step >> 32, chain, (purpose << 24) | block)
  purpose 0: proposal normals; block j yields components 2j, 2j+1 via Box-Muller
  purpose 1: the acceptance draw; block 0 yields one Exp(1) variate
  u53(hi, lo) = (((hi << 32 | lo) >> 11) + 0.5) * 2**-53          in (0, 1)
  normals: r = sqrt(-2 ln u53(x1,x0)); t = 2*pi*u53(x3,x2); (r cos t, r sin t)
  exp:     -ln u53(x1, x0)
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32(ctr, key, rounds=10):
    """ctr: (..., 4) uint32, key: (..., 2) uint32 -> (..., 4) uint32."""
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    c0, c1, c2, c3 = (ctr[..., i].astype(np.uint64) for i in range(4))
    k0 = key[..., 0].astype(np.uint32)
    k1 = key[..., 1].astype(np.uint32)
    for _ in range(rounds):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        n0 = (hi1 ^ c1 ^ k0.astype(np.uint64)) & MASK32
        n1 = lo1
        n2 = (hi0 ^ c3 ^ k1.astype(np.uint64)) & MASK32
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        with np.errstate(over="ignore"):
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def _u53(hi, lo):
    v = (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)
    return ((v >> np.uint64(11)).astype(np.float64) + 0.5) * (2.0 ** -53)


def _ctr(step, chain, purpose, block):
    step = int(step)
    return np.array(
        [step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF, chain & 0xFFFFFFFF,
         ((purpose & 0xFF) << 24) | (block & 0xFFFFFF)], dtype=np.uint32)


def _key(seed):
    seed = int(seed)
    return np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)


def normals(seed, chain, step, m):
    """m standard normals for (chain, step): the proposal noise of one RWMH transition."""
    nblk = (m + 1) // 2
    ctr = np.stack([_ctr(step, chain, 0, j) for j in range(nblk)])
    x = philox4x32(ctr, np.broadcast_to(_key(seed), (nblk, 2)))
    u1 = _u53(x[:, 1], x[:, 0])
    u2 = _u53(x[:, 3], x[:, 2])
    r = np.sqrt(-2.0 * np.log(u1))
    t = (2.0 * np.pi) * u2
    out = np.empty(2 * nblk, dtype=np.float64)
    out[0::2] = r * np.cos(t)
    out[1::2] = r * np.sin(t)
    return out[:m]


def randexp(seed, chain, step):
    """One Exp(1) variate for the acceptance test of (chain, step)."""
    x = philox4x32(_ctr(step, chain, 1, 0)[None, :], _key(seed)[None, :])
    return float(-np.log(_u53(x[:, 1], x[:, 0]))[0])
