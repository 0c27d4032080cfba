"""Seeded inputs shared by the parity tests.  Uses the ORACLE's fixed-base
multiplier to make points with known discrete logs, so that every MSM also has the closed-form
answer (sum s_i k_i mod r) * G."""
import numpy as np

from oracle import bn254_ref as o


def fr_mont_bytes(vals) -> bytes:
    return b"".join(o.fr_to_mont_bytes(v) for v in vals)


def fr_std_bytes(vals) -> bytes:
    return b"".join(o.fr_to_std_bytes(v) for v in vals)


def fr_from_mont(buf: bytes):
    return [o.fr_from_mont_bytes(buf[32 * i:32 * i + 32]) for i in range(len(buf) // 32)]


def uniform_scalars(n, seed):
    rng = o.SplitMix64(seed)
    return [rng.fr() for _ in range(n)]


def circom_like_scalars(n, seed):
    """40 % zero, 30 % one, 10 % < 2^16, 20 % uniform."""
    rng = o.SplitMix64(seed)
    out = []
    for _ in range(n):
        u = rng.next() % 100
        if u < 40:
            out.append(0)
        elif u < 70:
            out.append(1)
        elif u < 80:
            out.append(rng.next() & 0xFFFF)
        else:
            out.append(rng.fr())
    return out


def points_with_logs(orc, group, n, seed):
    ks = uniform_scalars(n, seed)
    return ks, orc.fixed_base(group, fr_mont_bytes(ks))


def expected_from_logs(group, scalars, logs):
    e = sum(s * k for s, k in zip(scalars, logs)) % o.R
    if group == 1:
        return o.g1_to_bytes(o.G1.mul(e, o.GEN1))
    return o.g2_to_bytes(o.G2.mul(e, o.GEN2))
