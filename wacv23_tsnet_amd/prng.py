"""Counter-based, integer-only PRNG used for synthetic weights and inputs.

Why our own generator: the parity tests must regenerate *bit-identical* weights
(67 M floats at cfg0 -- far too large to commit) and inputs on three sides:
  * the golden-capture script that imports the reference (authoring container only),
  * the CPU oracle, and
  * the GPU box that never sees the reference.
torch / numpy RNG streams and libm transcendental functions are not guaranteed to
be bit-stable across builds and CPU ISAs, so everything here is integer hashing
(splitmix64) followed by exactly-representable float arithmetic.  The hashing runs
on torch int64 tensors (two's-complement wrap-around == uint64 arithmetic; logical
shifts are emulated with a mask) because torch's integer kernels are threaded;
`_hash64_numpy` is the plain uint64 statement of the same function and
tests/test_prng.py pins the two against each other and against fixed known answers.

Distributions (what the reference uses):
  * images   ~ U[0,1)            (quick_start1.py:18 `torch.rand`)
  * labels   ~ Bernoulli(0.5)    (quick_start1.py:19 `torch.randint(0,2)`)
  * bboxes   ~ Bernoulli(0.5)    (quick_start1.py:20)
  * weights  ~ N(0, 0.02), bias 0 (model/networks.py:82,92 `init.normal_`,`init.constant_`)
The normal is an Irwin-Hall(12) sum of 16-bit uniforms (exact in integers; unit
variance, excess kurtosis -0.1) -- a synthetic stand-in with the reference's
first two moments, not a bit-copy of torch's Box-Muller stream.
"""
from __future__ import annotations

import numpy as np
import torch

_MASK64 = 0xFFFFFFFFFFFFFFFF
_GOLDEN = 0x9E3779B97F4A7C15
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB


def _s64(x: int) -> int:
    """Reinterpret an unsigned 64-bit python int as signed (for torch.int64)."""
    x &= _MASK64
    return x - (1 << 64) if x >= (1 << 63) else x


def fnv1a64(name: str) -> int:
    """64-bit FNV-1a of a tensor name -> stream id."""
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & _MASK64
    return h


def _mix_int(z: int) -> int:
    z &= _MASK64
    z = ((z ^ (z >> 30)) * _C1) & _MASK64
    z = ((z ^ (z >> 27)) * _C2) & _MASK64
    return z ^ (z >> 31)


def _key(seed: int, stream: int, lane: int) -> int:
    return _mix_int(seed * 0xD1342543DE82EF95 + stream * _GOLDEN + lane * 0xA0761D6478BD642F + 1)


def _mix_torch(z: torch.Tensor) -> torch.Tensor:
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * _s64(_C1)
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * _s64(_C2)
    return z ^ ((z >> 31) & ((1 << 33) - 1))


def hash64(seed: int, stream: int, n: int, lane: int = 0) -> torch.Tensor:
    """n 64-bit hashes (as int64 bit patterns) for counters 0..n-1 of (seed, stream, lane)."""
    ctr = torch.arange(1, n + 1, dtype=torch.int64)
    return _mix_torch(ctr * _s64(_GOLDEN) + _s64(_key(seed, stream, lane)))


def _hash64_numpy(seed: int, stream: int, n: int, lane: int = 0) -> np.ndarray:
    """Reference statement of hash64 in numpy uint64 (slow; used by tests only)."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(_GOLDEN) + np.uint64(_key(seed, stream, lane))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, name: str, shape) -> torch.Tensor:
    """float32 U[0,1) with 24 random bits (exactly representable)."""
    n = int(np.prod(shape))
    h = hash64(seed, fnv1a64(name), n)
    top = (h >> 40) & ((1 << 24) - 1)
    return (top.to(torch.float32) * (1.0 / 16777216.0)).reshape(tuple(shape))


def bernoulli(seed: int, name: str, shape) -> torch.Tensor:
    """float32 0/1 with p = 0.5 (top hash bit)."""
    n = int(np.prod(shape))
    h = hash64(seed, fnv1a64(name), n)
    return ((h >> 63) & 1).to(torch.float32).reshape(tuple(shape))


def normal(seed: int, name: str, shape, std: float = 0.02) -> torch.Tensor:
    """float32 ~N(0, std): Irwin-Hall(12) of 16-bit uniforms, integer-exact."""
    n = int(np.prod(shape))
    stream = fnv1a64(name)
    acc = torch.zeros(n, dtype=torch.int64)
    for lane in range(3):
        h = hash64(seed, stream, n, lane=lane)
        for sh in (0, 16, 32, 48):
            acc += (h >> sh) & 0xFFFF
    # sum of 12 U{0..65535}; centre 12*65535/2; /65536 -> unit variance of 12 U[0,1)
    z = (2 * acc - 12 * 65535).to(torch.float64) / 131072.0
    return (z * float(std)).to(torch.float32).reshape(tuple(shape))
