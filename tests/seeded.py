"""Deterministic, RNG-library-independent test data: a counter-based splitmix64 stream -> U[-1,1).  The same
function seeds the reference (in the build container, for the golden fixtures), the oracle and the MI355X build,
so no weights ever need to be shipped."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def uniform(shape, seed, stream=0):
    """float32 array of `shape`, U[-1,1), fully determined by (seed, stream)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over='ignore'):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64) + base
    u = (_splitmix64(idx) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (u * 2.0 - 1.0).astype(np.float32).reshape(shape)


def _scale_for(key, shape, overrides):
    for pat, s in (overrides or {}).items():
        if pat in key:
            return s
    if key.endswith('bias'):
        return 0.05
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    return float(np.sqrt(3.0 / max(fan_in, 1)))          # U(-s,s) with variance 1/fan_in


def seeded_state_dict(shapes, seed, overrides=None):
    """shapes: ordered mapping key -> shape (a module's state_dict order).  Returns key -> float32 ndarray."""
    out = {}
    for i, (k, shp) in enumerate(shapes.items()):
        out[k] = uniform(tuple(shp), seed, stream=i) * np.float32(_scale_for(k, tuple(shp), overrides))
    return out


def seeded_images(N, C, H, W, seed):
    return uniform((N, C, H, W), seed, 1000), uniform((N, C, H, W), seed, 1001)
