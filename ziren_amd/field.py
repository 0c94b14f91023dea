"""KoalaBear helpers for host-side tooling (numpy, canonical integers).

p = 2^31 - 2^24 + 1; the ABI carries Montgomery words with R = 2^32
(reference: crates/core/machine/include/kb31_t.hpp:458-503). These helpers are used to
build inputs (synthetic traces, constants inside bytecode) — never on the proving path.
"""
import numpy as np

P = 0x7F000001
GENERATOR = 3
TWO_ADICITY = 24
_R = (1 << 32) % P
_RINV = pow(_R, P - 2, P)


def to_monty(x):
    """canonical -> Montgomery (scalar int or numpy array)."""
    if isinstance(x, (int, np.integer)):
        return (int(x) % P << 32) % P
    a = np.asarray(x, dtype=np.uint64)
    return ((a << np.uint64(32)) % np.uint64(P)).astype(np.uint32)


def from_monty(x):
    if isinstance(x, (int, np.integer)):
        return int(x) * _RINV % P
    a = np.asarray(x, dtype=np.uint64)
    return (a % np.uint64(P) * np.uint64(_RINV) % np.uint64(P)).astype(np.uint32)


def mul(a, b):
    return (np.asarray(a, dtype=np.uint64) * np.asarray(b, dtype=np.uint64)) % np.uint64(P)


def add(a, b):
    return (np.asarray(a, dtype=np.uint64) + np.asarray(b, dtype=np.uint64)) % np.uint64(P)


def sub(a, b):
    return (np.asarray(a, dtype=np.uint64) + np.uint64(P) - np.asarray(b, dtype=np.uint64)) % np.uint64(P)


def inv(a: int) -> int:
    return pow(int(a), P - 2, P)


def two_adic_generator(bits: int) -> int:
    assert bits <= TWO_ADICITY
    return pow(GENERATOR, (P - 1) >> bits, P)


class SplitMix64:
    """Seeded generator used for every synthetic input."""

    def __init__(self, seed: int):
        self.state = np.uint64(seed)

    def next_u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + idx * np.uint64(0x9E3779B97F4A7C15)
            self.state = z[-1] if n else self.state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return z

    def uniform_field(self, shape):
        n = int(np.prod(shape))
        return (self.next_u64(n) % np.uint64(P)).reshape(shape)
