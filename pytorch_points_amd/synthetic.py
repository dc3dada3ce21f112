"""Synthetic point clouds for tests and bench.py.

Counter-based: every value is a pure function of (seed, flat element index), so inputs do not
depend on numpy's or torch's RNG version.  ``unit_sphere`` is the area-uniform unit sphere
(normalised 3-D Gaussian); ``polar_sphere`` mirrors the reference's only synthetic generator,
``utils/pc_utils.py:504-516 random_sphere`` (theta, phi uniform -> clustered at the poles).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """splitmix64 finaliser, vectorised over uint64."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def uniform01(seed, shape, stream=0):
    """float64 uniforms in (0,1), one per element."""
    n = int(np.prod(shape))
    ctr = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        bits = _splitmix64(ctr ^ key)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / (1 << 53))


def normal(seed, shape):
    """float32 standard normals (Box-Muller on two counter streams)."""
    u1 = uniform01(seed, shape, 0)
    u2 = uniform01(seed, shape, 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.reshape(shape).astype(np.float32)


def unit_sphere(seed, b, n, c=3):
    """(b,n,c) float32, area-uniform on the unit (c-1)-sphere."""
    g = normal(seed, (b, n, c)).astype(np.float64)
    g /= np.maximum(np.linalg.norm(g, axis=-1, keepdims=True), 1e-30)
    return np.ascontiguousarray(g.astype(np.float32))


def polar_sphere(seed, b, n):
    """(b,n,3) float32, theta/phi uniform (the reference's random_sphere distribution)."""
    theta = 2.0 * np.pi * uniform01(seed, (b, n), 0).reshape(b, n)
    phi = np.pi * uniform01(seed, (b, n), 1).reshape(b, n)
    x = np.cos(theta) * np.sin(phi)
    y = np.sin(theta) * np.sin(phi)
    z = np.cos(phi)
    return np.ascontiguousarray(np.stack([x, y, z], -1).astype(np.float32))
