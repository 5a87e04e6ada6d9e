"""TEST INFRASTRUCTURE (oracle): CPU restatement of the in-kernel noise stream of ds_renoise_mix's perf mode.

The reference draws re-noise with torch.randn on the host (pipeline/scheduler.py:98-110).  The product's `rng_mode="device"`
(bench.py) draws it inside the kernel instead -- Philox4x32-10 keyed by the seed, counter = offset + vector index, four normals per
counter by Box-Muller (csrc/tile_ops.hip: philox4x32_10, normal4).  This module restates that stream in numpy so that the
device mode can be checked VALUE BY VALUE (the kernel uses the hardware's fast log / sin / cos, so agreement is to ~1e-6
absolute, not bitwise): tests/test_gpu_kernels.py::test_renoise_philox_matches_cpu_restatement.
"""
import numpy as np

_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
_MASK = 0xFFFFFFFF


def philox4x32_10(ctr, seed):
    """ctr: uint64 array of counters (c0 = low word, c1 = high word, c2 = c3 = 0); seed: python int (k0 = low, k1 = high word).
    Returns uint32 [..., 4]."""
    ctr = np.asarray(ctr, dtype=np.uint64)
    c0 = (ctr & np.uint64(_MASK)).astype(np.uint64)
    c1 = (ctr >> np.uint64(32)).astype(np.uint64)
    c2 = np.zeros_like(c0)
    c3 = np.zeros_like(c0)
    k0, k1 = int(seed) & _MASK, (int(seed) >> 32) & _MASK
    for _ in range(10):
        p0 = np.uint64(_M0) * c0                      # 32 x 32 -> 64 bit products (operands < 2^32, no overflow in uint64)
        p1 = np.uint64(_M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(_MASK)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(_MASK)
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def normal4(ctr, seed):
    """Four N(0,1) per counter, float32 [..., 4]: Box-Muller on 24-bit uniforms, u0 / u2 in (0,1], u1 / u3 in [0,1)."""
    r = philox4x32_10(ctr, seed)
    f = np.float32
    scale = f(1.0 / 16777216.0)
    u0 = ((r[..., 0] >> 8).astype(f) + f(1.0)) * scale
    u1 = (r[..., 1] >> 8).astype(f) * scale
    u2 = ((r[..., 2] >> 8).astype(f) + f(1.0)) * scale
    u3 = (r[..., 3] >> 8).astype(f) * scale
    two_pi = f(6.283185307179586)
    ra = np.sqrt(f(-2.0) * np.log(u0)).astype(f)
    rb = np.sqrt(f(-2.0) * np.log(u2)).astype(f)
    return np.stack([ra * np.cos(two_pi * u1), ra * np.sin(two_pi * u1), rb * np.cos(two_pi * u3), rb * np.sin(two_pi * u3)],
                    axis=-1).astype(f)


def tile_noise(shape, seed, offset):
    """The normals ds_renoise_mix draws for tiles of `shape` = [n, C, tf, th, tw] with tw % 4 == 0: element e (row-major over the
    whole batch) takes component e % 4 of counter offset + e // 4."""
    n = int(np.prod(shape))
    assert shape[-1] % 4 == 0
    ctr = np.uint64(offset) + np.arange(n // 4, dtype=np.uint64)
    return normal4(ctr, seed).reshape(shape)
