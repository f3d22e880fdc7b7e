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


def _upsample_linear(x, f):
    """separable linear upsampling by integer factor f along the last two axes (align_corners=True style)."""
    def up(a, axis):
        n = a.shape[axis]
        pos = np.linspace(0, n - 1, n * f)
        i0 = np.floor(pos).astype(int)
        i1 = np.minimum(i0 + 1, n - 1)
        t = (pos - i0).astype(np.float32)
        shp = [1] * a.ndim
        shp[axis] = -1
        return np.take(a, i0, axis) * (1 - t).reshape(shp) + np.take(a, i1, axis) * t.reshape(shp)
    return up(up(x, -1), -2)


def seeded_images(N, C, H, W, seed, smooth=8):
    """A/B pairs in [-1,1]: low-frequency structure (1/`smooth`-resolution noise, linearly upsampled) plus 10% white
    noise.  Smooth content keeps d(warp)/d(offset) well conditioned, so gradient parity is a test of the kernels and
    not of floor() decisions on white noise; B is a slightly shifted remix of A so registration has signal."""
    def one(stream):
        base = uniform((N, C, H // smooth, W // smooth), seed, stream)
        img = 0.9 * _upsample_linear(base, smooth) + 0.1 * uniform((N, C, H, W), seed, stream + 50)
        return np.ascontiguousarray(img.astype(np.float32))
    a = one(1000)
    b = 0.7 * np.roll(a, (2, -3), axis=(2, 3)) + 0.3 * one(1001)
    return a, np.ascontiguousarray(b.astype(np.float32))
