"""Loader + thin helpers for oracle/liboracle.so (TEST INFRASTRUCTURE; never imported by the product)."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORC = None


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp, i32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        lib.orc_cores.restype = i32
        lib.orc_field_op.argtypes = [i32, i32, vp, vp, vp]
        for g in ("g1", "g2"):
            getattr(lib, f"orc_msm_naive_{g}").argtypes = [vp, i32, vp, sz, vp]
            getattr(lib, f"orc_msm_{g}").argtypes = [i32, vp, i32, vp, sz, vp]
            getattr(lib, f"orc_fixed_base_{g}").argtypes = [i32, vp, i32, sz, vp]
            getattr(lib, f"orc_{g}_add").argtypes = [vp, vp, vp]
            getattr(lib, f"orc_{g}_mul").argtypes = [vp, i32, vp, vp]
        lib.orc_domain_gen.argtypes = [i32, vp]
        lib.orc_ntt_forward.argtypes = [vp, vp, i32]
        lib.orc_ntt_inverse.argtypes = [vp, vp, i32]
        lib.orc_quotient_snarkjs.argtypes = [vp, vp, vp, i32, vp, i32]
        lib.orc_quotient_jensgroth.argtypes = [vp, vp, vp, i32, vp, i32]
        lib.orc_build_abc.argtypes = [vp, sz, vp, i32, vp]

    @staticmethod
    def _psz(group):
        return 64 if group == 1 else 128

    def cores(self):
        return self.lib.orc_cores()

    def msm(self, group, scalars: bytes, points: bytes, mont=True, threads=0) -> bytes:
        n = len(scalars) // 32
        out = ctypes.create_string_buffer(self._psz(group))
        fn = self.lib.orc_msm_g1 if group == 1 else self.lib.orc_msm_g2
        fn(threads, scalars, 1 if mont else 0, points, n, out)
        return out.raw

    def msm_naive(self, group, scalars: bytes, points: bytes, mont=True) -> bytes:
        n = len(scalars) // 32
        out = ctypes.create_string_buffer(self._psz(group))
        fn = self.lib.orc_msm_naive_g1 if group == 1 else self.lib.orc_msm_naive_g2
        fn(scalars, 1 if mont else 0, points, n, out)
        return out.raw

    def fixed_base(self, group, scalars: bytes, mont=True, threads=0) -> bytes:
        n = len(scalars) // 32
        out = ctypes.create_string_buffer(self._psz(group) * max(n, 1))
        fn = self.lib.orc_fixed_base_g1 if group == 1 else self.lib.orc_fixed_base_g2
        fn(threads, scalars, 1 if mont else 0, n, out)
        return out.raw[: self._psz(group) * n]

    def add(self, group, a: bytes, b: bytes) -> bytes:
        out = ctypes.create_string_buffer(self._psz(group))
        (self.lib.orc_g1_add if group == 1 else self.lib.orc_g2_add)(a, b, out)
        return out.raw

    def mul(self, group, scalar: bytes, p: bytes, mont=True) -> bytes:
        out = ctypes.create_string_buffer(self._psz(group))
        (self.lib.orc_g1_mul if group == 1 else self.lib.orc_g2_mul)(scalar, 1 if mont else 0, p, out)
        return out.raw

    def ntt(self, src: bytes, log2n: int, inverse=False) -> bytes:
        out = ctypes.create_string_buffer(len(src))
        rc = (self.lib.orc_ntt_inverse if inverse else self.lib.orc_ntt_forward)(src, out, log2n)
        assert rc == 0
        return out.raw

    def quotient_snarkjs(self, Az: bytes, Bz: bytes, Cz: bytes, log2n: int, parallel=True) -> bytes:
        out = ctypes.create_string_buffer(len(Az))
        rc = self.lib.orc_quotient_snarkjs(Az, Bz, Cz, log2n, out, 1 if parallel else 0)
        assert rc == 0
        return out.raw

    def quotient_jensgroth(self, Az: bytes, Bz: bytes, Cz: bytes, log2n: int, parallel=True) -> bytes:
        """computeQuotientPointwise (prover.nim:118-148)"""
        out = ctypes.create_string_buffer(len(Az))
        rc = self.lib.orc_quotient_jensgroth(Az, Bz, Cz, log2n, out, 1 if parallel else 0)
        assert rc == 0
        return out.raw

    def build_abc(self, packed_coeffs: bytes, witness: bytes, log2n: int):
        n = 1 << log2n
        out = ctypes.create_string_buffer(3 * n * 32)
        rc = self.lib.orc_build_abc(packed_coeffs, len(packed_coeffs) // 48, witness, log2n, out)
        assert rc == 0
        raw = out.raw
        return raw[:32 * n], raw[32 * n:64 * n], raw[64 * n:]

    def field_op(self, field, op, a: bytes, b: bytes = b"\0" * 32) -> bytes:
        out = ctypes.create_string_buffer(32)
        self.lib.orc_field_op(field, op, a, b, out)
        return out.raw


def load_oracle() -> Oracle:
    global _ORC
    if _ORC is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _ORC = Oracle(ctypes.CDLL(path))
    return _ORC
