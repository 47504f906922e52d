"""
Seeded stand-in for the reference's `noised: True` draws (np.random.normal on an unseeded generator, optimizer.py:611-617 and
348-354): a counter-based generator, Philox4x32-10 (Salmon et al., SC'11) + Box-Muller, keyed by `seed` and the triple
(instance, step, sample index).  This module is the numpy mirror of `loop_normal` / `loop_noise` of csrc/mpc_closed_loop.h --
the device-side closed loop (mpc_closed_loop_batch_ex) and the step-by-step host loop of optimizer.py draw the SAME samples.
"""
from __future__ import annotations

import numpy as np

_M0, _M1, _W0, _W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_TAG = 0x4D5043


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(a, dtype=np.uint32) for a in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32(k0 + _W0)
            k1 = np.uint32(k1 + _W1)
    return c0, c1, c2, c3


def normal(seed: int, b, i, j):
    """standard normal sample j of (instance b, step i); arguments broadcast"""
    j = np.asarray(j, dtype=np.int64)
    x0, x1, x2, x3 = philox4x32_10((j >> 1).astype(np.uint32), np.asarray(i, dtype=np.uint32), np.asarray(b, dtype=np.uint32), np.uint32(_TAG),
                                   int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    a1 = ((x0 >> np.uint32(5)).astype(np.uint64) << np.uint64(26)) | (x1 >> np.uint32(6)).astype(np.uint64)
    a2 = ((x2 >> np.uint32(5)).astype(np.uint64) << np.uint64(26)) | (x3 >> np.uint32(6)).astype(np.uint64)
    u1 = (a1.astype(np.float64) + 1.0) * 1.1102230246251565e-16
    u2 = a2.astype(np.float64) * 1.1102230246251565e-16
    r, th = np.sqrt(-2.0 * np.log(u1)), 6.283185307179586 * u2
    return np.where(j & 1, r * np.sin(th), r * np.cos(th))


def sequence_noise(seed: int, b: int, i: int, N: int, sigma: float):
    """mode 1 (CasadiOptimizer, optimizer.py:611-615): noise of the whole predicted input sequence of step i as a (2, N) array
    -- sample index r * N + k, the row-major reshape of the reference's draw"""
    return sigma * normal(seed, b, i, np.arange(2 * N)).reshape(2, N)


def applied_noise(seed: int, b: int, i: int, sigma: float):
    """mode 2 (ForcesproOptimizer, optimizer.py:348-354): noise of the applied input of step i, shape (2,)"""
    return sigma * normal(seed, b, i, np.arange(2))
