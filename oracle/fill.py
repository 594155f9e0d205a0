"""Closed-form, platform-exact pseudo-random fills (test infrastructure).

Golden fixtures cannot carry 18 M parameters, and `torch.manual_seed` streams
are not guaranteed stable across builds, so every tensor used by the parity
tests is produced from integer arithmetic only: splitmix64 of the flat element
index, salted by a tensor id.  numpy uint64 arithmetic wraps identically
everywhere, so the goldens generator (run next to the reference) and the tests
(run on the GPU box) see bit-identical inputs.
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def salt_of(name):
    """Stable 32-bit salt for a tensor name."""
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def uniform01(shape, salt):
    """float64 uniform in [0,1) with 53 random bits, exact everywhere."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64(idx ^ (np.uint64(salt) << np.uint64(32)))
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def uniform(shape, salt, lo=-1.0, hi=1.0, dtype=np.float32):
    return (lo + (hi - lo) * uniform01(shape, salt)).astype(dtype)


def normalish(shape, salt, dtype=np.float32):
    """Sum of 4 uniforms, zero mean / unit variance (Irwin-Hall), exact."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += uniform01(shape, (salt * 4 + k) & 0xFFFFFFFF)
    return ((acc - 2.0) * np.sqrt(3.0)).astype(dtype)


def fill_state_dict_(module, seed=0):
    """Overwrite every parameter / float buffer of a torch module in place.

    weights  ~ U(-b, b), b = 1/sqrt(fan_in)        (ndim >= 2)
    biases   ~ U(-0.1, 0.1)                         (name endswith 'bias')
    BN gamma ~ U(0.8, 1.2)                          (1-D 'weight')
    running_mean = 0, running_var = 1 (torch defaults kept), queue untouched.
    """
    import torch
    with torch.no_grad():
        for name, p in module.state_dict().items():
            if not torch.is_floating_point(p):
                continue
            if name.endswith("running_mean") or name.endswith("running_var"):
                continue
            if name == "queue":
                continue
            salt = (salt_of(name) + 0x9E3779B1 * seed) & 0xFFFFFFFF
            shape = tuple(p.shape)
            if p.ndim >= 2:
                fan_in = int(np.prod(shape[1:]))
                b = 1.0 / np.sqrt(fan_in)
                v = uniform(shape, salt, -b, b)
            elif name.endswith("bias"):
                v = uniform(shape, salt, -0.1, 0.1)
            else:
                v = uniform(shape, salt, 0.8, 1.2)
            p.copy_(torch.from_numpy(v))
    return module


def fill_shapes(shapes, seed=0):
    """`fill_state_dict_` for a plain name -> shape mapping: the same values a module with those parameter names would get."""
    import torch
    out = {}
    for name, shape in shapes.items():
        salt = (salt_of(name) + 0x9E3779B1 * seed) & 0xFFFFFFFF
        shape = tuple(shape)
        if len(shape) >= 2:
            b = 1.0 / np.sqrt(int(np.prod(shape[1:])))
            v = uniform(shape, salt, -b, b)
        elif name.endswith("bias"):
            v = uniform(shape, salt, -0.1, 0.1)
        else:
            v = uniform(shape, salt, 0.8, 1.2)
        out[name] = torch.from_numpy(v)
    return out
