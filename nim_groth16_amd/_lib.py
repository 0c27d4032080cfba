"""ctypes binding of libg16hip.so (include/g16hip.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes
import json
import os
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))

G16_OK, G16_EINVAL, G16_ENODEV, G16_EHIP, G16_ENOMEM, G16_ESELFTEST = 0, -1, -2, -3, -4, -5
SCALARS_MONT, SCALARS_STD, SCALARS_DEVICE, OUT_PARTIAL, OUT_DEVICE, NO_HOST_SYNC = 1, 0, 2, 4, 8, 32
PARTIALS_BYTES = 768

# every symbol include/g16hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "g16_ctx_create", "g16_ctx_destroy", "g16_last_error", "g16_ctx_set_stream", "g16_ctx_synchronize",
    "g16_selftest", "g16_msm_g1", "g16_msm_g2", "g16_msm_g1_dev", "g16_msm_g2_dev",
    "g16_msm_g1_partial_dev", "g16_msm_g2_partial_dev", "g16_g1_sum_partials", "g16_g2_sum_partials",
    "g16_points_register_g1", "g16_points_register_g2", "g16_points_register_g1_dev",
    "g16_points_register_g2_dev", "g16_points_release", "g16_points_count", "g16_points_inf_count", "g16_points_info", "g16_msm_points",
    "g16_points_check_g1", "g16_points_check_g2", "g16_fixed_base_g1", "g16_fixed_base_g2", "g16_quotient", "g16_quotient_dev", "g16_pkey_create",
    "g16_pkey_create_zkey", "g16_pkey_destroy", "g16_pkey_inf_counts", "g16_prove", "g16_build_abc", "g16_pkey_abc_info", "g16_spmv_fr", "g16_prove_partials", "g16_prove_combine",
    "g16_prove_partials_begin", "g16_prove_partials_end",
    "g16_ntt_fr", "g16_ntt_fr_dev", "g16_profile_enable", "g16_profile_reset", "g16_profile_report", "g16_profile_clock",
    "g16_vkey_create", "g16_vkey_destroy", "g16_verify", "g16_pairing",
    "g16_ctx_cancel", "g16_group_create", "g16_group_destroy", "g16_group_size", "g16_group_last_error",
    "g16_group_pkey_create", "g16_group_pkey_destroy", "g16_group_prove",
]
VERIFY_SUBGROUP = 16
GT_BYTES = 384


class PkeyDesc(ctypes.Structure):
    """g16_pkey_desc (include/g16hip.h)"""
    _fields_ = [("nvars", ctypes.c_uint32), ("npubs", ctypes.c_uint32), ("log2_domain", ctypes.c_uint32),
                ("flavour", ctypes.c_uint32),
                ("pointsA1", ctypes.c_void_p), ("pointsB1", ctypes.c_void_p), ("pointsB2", ctypes.c_void_p),
                ("pointsC1", ctypes.c_void_p), ("pointsH1", ctypes.c_void_p),
                ("coeffs", ctypes.c_void_p), ("ncoeffs", ctypes.c_size_t),
                ("alpha1", ctypes.c_void_p), ("beta1", ctypes.c_void_p), ("delta1", ctypes.c_void_p),
                ("beta2", ctypes.c_void_p), ("delta2", ctypes.c_void_p),
                ("shard_index", ctypes.c_uint32), ("shard_count", ctypes.c_uint32)]


class VkeyDesc(ctypes.Structure):
    """g16_vkey_desc (include/g16hip.h)"""
    _fields_ = [("npubs", ctypes.c_uint32), ("alpha1", ctypes.c_void_p), ("beta2", ctypes.c_void_p),
                ("gamma2", ctypes.c_void_p), ("delta2", ctypes.c_void_p), ("pointsIC", ctypes.c_void_p)]


class G16Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"g16hip error {code}: {msg}")
        self.code = code


def lib_path() -> str:
    return os.environ.get("G16HIP_LIB", os.path.join(_HERE, "csrc", "libg16hip.so"))


def device_code_sha16(path: str = None) -> str:
    """Identity of the KERNELS of a libg16hip.so build: sha256 prefix of its .hip_fatbin section (the gfx950 code
    objects).  Host-side edits of the library leave it unchanged; any kernel change alters it.  bench.py reports
    counter-derived numbers (HBM traffic, VALU instruction counts) only next to files stamped with the same value."""
    import hashlib
    import struct
    b = open(path or lib_path(), "rb").read()
    h = hashlib.sha256()
    try:
        assert b[:4] == b"\x7fELF" and b[4] == 2
        shoff = struct.unpack_from("<Q", b, 0x28)[0]
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)

        def sh(i):
            name, _typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", b, shoff + i * shentsize)
            return name, off, size
        _, stro, _ = sh(shstrndx)
        hit = False
        for i in range(shnum):
            n, off, size = sh(i)
            if b[stro + n: b.index(b"\0", stro + n)] == b".hip_fatbin":
                h.update(b[off: off + size])
                hit = True
        assert hit
    except Exception:
        h = hashlib.sha256(b)             # not an ELF with a fat binary: the whole file
    return h.hexdigest()[:16]


_lib = None


def load_library():
    """Loads libg16hip.so.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise G16Error(G16_ENODEV, f"{path} not found: build it with `make -C nim_groth16_amd/csrc` "
                                   "(or __graft_entry__.build()); there is no CPU fallback")
    # torch bundles its own libamdhip64.so.7; load it first so that this library binds to the SAME HIP
    # runtime (same SONAME) -- two HIP runtimes in one process cannot both own the GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    vp, u32, i32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_size_t
    lib.g16_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    lib.g16_ctx_destroy.argtypes = [vp]
    lib.g16_ctx_destroy.restype = None
    lib.g16_last_error.argtypes = [vp]
    lib.g16_last_error.restype = ctypes.c_char_p
    lib.g16_ctx_set_stream.argtypes = [vp, vp]
    lib.g16_ctx_synchronize.argtypes = [vp]
    lib.g16_selftest.argtypes = [vp]
    for name in ("g16_msm_g1", "g16_msm_g2", "g16_msm_g1_dev", "g16_msm_g2_dev",
                 "g16_msm_g1_partial_dev", "g16_msm_g2_partial_dev"):
        getattr(lib, name).argtypes = [vp, vp, u32, vp, sz, vp]
    for name in ("g16_g1_sum_partials", "g16_g2_sum_partials"):
        getattr(lib, name).argtypes = [vp, vp, sz, vp]
    for name in ("g16_points_register_g1", "g16_points_register_g2", "g16_points_register_g1_dev",
                 "g16_points_register_g2_dev"):
        getattr(lib, name).argtypes = [vp, vp, sz, ctypes.POINTER(vp)]
    lib.g16_points_release.argtypes = [vp]
    lib.g16_points_release.restype = None
    lib.g16_points_count.argtypes = [vp]
    lib.g16_points_count.restype = sz
    lib.g16_points_inf_count.argtypes = [vp]
    lib.g16_points_inf_count.restype = sz
    lib.g16_pkey_inf_counts.argtypes = [vp, ctypes.POINTER(sz)]
    lib.g16_points_info.argtypes = [vp, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    lib.g16_msm_points.argtypes = [vp, vp, vp, u32, vp]
    lib.g16_points_check_g1.argtypes = [vp, vp, sz, ctypes.POINTER(sz)]
    lib.g16_points_check_g2.argtypes = [vp, vp, sz, ctypes.POINTER(sz)]
    lib.g16_fixed_base_g1.argtypes = [vp, vp, u32, sz, vp]
    lib.g16_fixed_base_g2.argtypes = [vp, vp, u32, sz, vp]
    lib.g16_quotient.argtypes = [vp, vp, vp, vp, u32, u32, vp]
    lib.g16_quotient_dev.argtypes = [vp, vp, vp, vp, u32, u32, vp]
    lib.g16_pkey_create.argtypes = [vp, ctypes.POINTER(PkeyDesc), ctypes.POINTER(vp)]
    lib.g16_pkey_create_zkey.argtypes = [vp, ctypes.POINTER(PkeyDesc), vp, sz, ctypes.POINTER(vp)]
    lib.g16_pkey_destroy.argtypes = [vp]
    lib.g16_pkey_destroy.restype = None
    lib.g16_prove.argtypes = [vp, vp, vp, u32, vp, vp, vp]
    lib.g16_build_abc.argtypes = [vp, vp, vp, u32, vp]
    lib.g16_pkey_abc_info.argtypes = [vp, ctypes.POINTER(sz)]
    lib.g16_spmv_fr.argtypes = [vp, vp, vp, vp, sz, vp, sz, sz, vp]
    lib.g16_prove_partials.argtypes = [vp, vp, vp, u32, vp]
    lib.g16_prove_combine.argtypes = [vp, vp, vp, sz, u32, vp, vp, vp]
    lib.g16_prove_partials_begin.argtypes = [vp, vp, vp, u32, u32, vp]
    lib.g16_prove_partials_end.argtypes = [vp, vp, vp, vp, vp, u32, vp]
    lib.g16_ntt_fr.argtypes = [vp, vp, vp, u32, i32]
    lib.g16_ntt_fr_dev.argtypes = [vp, vp, vp, u32, i32]
    lib.g16_vkey_create.argtypes = [vp, ctypes.POINTER(VkeyDesc), ctypes.POINTER(vp)]
    lib.g16_vkey_destroy.argtypes = [vp]
    lib.g16_vkey_destroy.restype = None
    lib.g16_verify.argtypes = [vp, vp, vp, vp, u32, sz, ctypes.POINTER(i32)]
    lib.g16_pairing.argtypes = [vp, vp, vp, sz, vp]
    lib.g16_ctx_cancel.argtypes = [vp]
    lib.g16_group_create.argtypes = [ctypes.POINTER(i32), i32, ctypes.POINTER(vp)]
    lib.g16_group_destroy.argtypes = [vp]
    lib.g16_group_destroy.restype = None
    lib.g16_group_size.argtypes = [vp]
    lib.g16_group_last_error.argtypes = [vp]
    lib.g16_group_last_error.restype = ctypes.c_char_p
    lib.g16_group_pkey_create.argtypes = [vp, ctypes.POINTER(PkeyDesc), ctypes.POINTER(vp)]
    lib.g16_group_pkey_destroy.argtypes = [vp]
    lib.g16_group_pkey_destroy.restype = None
    lib.g16_group_prove.argtypes = [vp, vp, vp, u32, vp, vp, vp]
    lib.g16_profile_enable.argtypes = [vp, i32]
    lib.g16_profile_reset.argtypes = [vp]
    lib.g16_profile_report.argtypes = [vp, ctypes.c_char_p, sz]
    lib.g16_profile_clock.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    for name in SYMBOLS:
        if name not in ("g16_ctx_destroy", "g16_last_error", "g16_points_release", "g16_points_count",
                        "g16_points_inf_count",
                        "g16_pkey_destroy", "g16_vkey_destroy", "g16_group_destroy", "g16_group_last_error",
                        "g16_group_pkey_destroy"):
            getattr(lib, name).restype = i32
    _lib = lib
    return lib


def _buf(b):
    """bytes / bytearray / numpy array / int (device pointer) -> c_void_p-compatible"""
    if isinstance(b, int):
        return ctypes.c_void_p(b)
    if isinstance(b, (bytes, bytearray)):
        return ctypes.cast(ctypes.c_char_p(bytes(b)), ctypes.c_void_p) if isinstance(b, bytes) else \
            ctypes.cast((ctypes.c_char * len(b)).from_buffer(b), ctypes.c_void_p)
    if hasattr(b, "ctypes"):            # numpy
        return ctypes.c_void_p(b.ctypes.data)
    raise TypeError(type(b))


class Context:
    """One g16_ctx = one GPU + one stream; used by one host thread at a time (include/g16hip.h)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        h = ctypes.c_void_p()
        rc = self._lib.g16_ctx_create(device, ctypes.byref(h))
        if rc != G16_OK:
            raise G16Error(rc, "g16_ctx_create failed (no usable HIP device?)")
        self._h = h
        self.device = device
        # point sets / keys created through this context.  They belong to the DEVICE (include/g16hip.h): the C ABI
        # lets them outlive the context or die first, in any order; the set only serves close(release_children=True)
        self._children = weakref.WeakSet()

    def close(self, release_children: bool = False):
        if getattr(self, "_h", None):
            if release_children:
                for child in list(self._children):
                    child._free()
            self._lib.g16_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != G16_OK:
            raise G16Error(rc, self._lib.g16_last_error(self._h).decode())

    def selftest(self):
        self._check(self._lib.g16_selftest(self._h))

    def set_stream(self, stream_ptr):
        self._check(self._lib.g16_ctx_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    def synchronize(self):
        self._check(self._lib.g16_ctx_synchronize(self._h))

    def cancel(self):
        """drain the main stream and every MSM lane, forget a pending prove_partials_begin (g16_ctx_cancel)"""
        self._check(self._lib.g16_ctx_cancel(self._h))

    # ---- MSM -----------------------------------------------------------------------------------
    def msm(self, group: int, scalars, points, n: int, mont: bool = True, device: bool = False,
            partial: bool = False) -> bytes:
        """group 1 -> G1 (64-byte points), 2 -> G2 (128-byte points).  Returns the affine result
        bytes (or the XYZZ partial when partial=True)."""
        psz = 64 if group == 1 else 128
        out = ctypes.create_string_buffer(2 * psz if partial else psz)
        if partial:
            fn = self._lib.g16_msm_g1_partial_dev if group == 1 else self._lib.g16_msm_g2_partial_dev
        elif device:
            fn = self._lib.g16_msm_g1_dev if group == 1 else self._lib.g16_msm_g2_dev
        else:
            fn = self._lib.g16_msm_g1 if group == 1 else self._lib.g16_msm_g2
        s = _buf(scalars) if n else None
        p = _buf(points) if n else None
        self._check(fn(self._h, s, SCALARS_MONT if mont else SCALARS_STD, p, n, out))
        return out.raw

    def register_points(self, group: int, points, n: int, device: bool = False) -> "PointSet":
        h = ctypes.c_void_p()
        name = f"g16_points_register_g{group}" + ("_dev" if device else "")
        self._check(getattr(self._lib, name)(self._h, _buf(points) if n else None, n, ctypes.byref(h)))
        return PointSet(self, h, group, n)

    def msm_points(self, pts: "PointSet", scalars, mont: bool = True, device: bool = False,
                   partial: bool = False) -> bytes:
        psz = 64 if pts.group == 1 else 128
        out = ctypes.create_string_buffer(2 * psz if partial else psz)
        flags = (SCALARS_MONT if mont else 0) | (SCALARS_DEVICE if device else 0) | (OUT_PARTIAL if partial else 0)
        self._check(self._lib.g16_msm_points(self._h, pts._h, _buf(scalars) if pts.n else None, flags, out))
        return out.raw

    def points_check(self, group: int, points: bytes):
        """index of the first point off the curve, or None (curves.nim:95-107 mkG1/mkG2 asserts)"""
        psz = 64 if group == 1 else 128
        n = len(points) // psz
        bad = ctypes.c_size_t()
        fn = self._lib.g16_points_check_g1 if group == 1 else self._lib.g16_points_check_g2
        self._check(fn(self._h, _buf(points) if n else None, n, ctypes.byref(bad)))
        return None if bad.value == ctypes.c_size_t(-1).value else bad.value

    def fixed_base(self, group: int, scalars: bytes, mont: bool = True) -> bytes:
        """scalars[i] * generator -> affine points (fake_setup.nim:258-261 `y ** gen1/gen2`)."""
        n = len(scalars) // 32
        psz = 64 if group == 1 else 128
        out = ctypes.create_string_buffer(max(n, 1) * psz)
        fn = self._lib.g16_fixed_base_g1 if group == 1 else self._lib.g16_fixed_base_g2
        self._check(fn(self._h, _buf(scalars) if n else None, SCALARS_MONT if mont else 0, n, out))
        return out.raw[: n * psz]

    def spmv(self, row, col, val, x: bytes, nrows: int) -> bytes:
        """y = M x over Fr (Montgomery): M as triplets -- row, col: uint32 numpy arrays; val: nnz x 32 bytes (numpy
        uint8 array or bytes); x: ncols Fr.  The sparse column dot products of fake_setup.nim:159-187 on the GPU."""
        import numpy as np
        row = np.ascontiguousarray(row, dtype=np.uint32)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        nnz = len(row)
        if not isinstance(val, (bytes, bytearray)):
            val = np.ascontiguousarray(val, dtype=np.uint8)
        assert len(col) == nnz and (len(val) if isinstance(val, (bytes, bytearray)) else val.size) == 32 * nnz
        out = ctypes.create_string_buffer(max(1, 32 * nrows))
        self._check(self._lib.g16_spmv_fr(self._h, _buf(row) if nnz else None, _buf(col) if nnz else None,
                                          _buf(val) if nnz else None, nnz, _buf(x) if len(x) else None, len(x) // 32,
                                          nrows, out))
        return out.raw[: 32 * nrows]

    def quotient(self, Az, Bz, Cz, log2n: int, flavour: int, out=None, device: bool = False):
        if device:
            self._check(self._lib.g16_quotient_dev(self._h, _buf(Az), _buf(Bz), _buf(Cz), log2n, flavour, _buf(out)))
            return None
        res = ctypes.create_string_buffer(32 << log2n)
        self._check(self._lib.g16_quotient(self._h, _buf(Az), _buf(Bz), _buf(Cz), log2n, flavour, res))
        return res.raw

    def sum_partials(self, group: int, xyzz: bytes, count: int) -> bytes:
        psz = 64 if group == 1 else 128
        out = ctypes.create_string_buffer(psz)
        fn = self._lib.g16_g1_sum_partials if group == 1 else self._lib.g16_g2_sum_partials
        self._check(fn(self._h, _buf(xyzz) if count else None, count, out))
        return out.raw

    # ---- NTT -----------------------------------------------------------------------------------
    def ntt(self, src, log2n: int, inverse: bool, dst=None, device: bool = False):
        n = 1 << log2n
        if device:
            self._check(self._lib.g16_ntt_fr_dev(self._h, _buf(src), _buf(dst), log2n, 1 if inverse else 0))
            return None
        out = ctypes.create_string_buffer(32 * n)
        self._check(self._lib.g16_ntt_fr(self._h, _buf(src), out, log2n, 1 if inverse else 0))
        return out.raw

    # ---- profiling -----------------------------------------------------------------------------
    def pairing(self, g1_points: bytes, g2_points: bytes) -> bytes:
        """e(P_i, Q_i) for n pairs -> n x 384 bytes (6 x Fp2 over w^k, Montgomery; curves.nim:218-221)"""
        n = len(g1_points) // 64
        assert len(g1_points) == 64 * n and len(g2_points) == 128 * n
        out = ctypes.create_string_buffer(max(1, GT_BYTES * n))
        self._check(self._lib.g16_pairing(self._h, _buf(g1_points) if n else None, _buf(g2_points) if n else None,
                                          n, out))
        return out.raw[:GT_BYTES * n]

    def profile(self, on):
        """False/0: off; True/1: HIP events around every kernel; 2: around the bucket-accumulation kernels only"""
        self._check(self._lib.g16_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._check(self._lib.g16_profile_reset(self._h))

    def profile_clock(self) -> float:
        """shader clock (GHz) sustained while the accumulate kernels since the last call ran (profiling on); 0 if none"""
        g = ctypes.c_double()
        self._check(self._lib.g16_profile_clock(self._h, ctypes.byref(g)))
        return g.value

    def profile_report(self) -> dict:
        buf = ctypes.create_string_buffer(1 << 16)
        self._check(self._lib.g16_profile_report(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())


class ProvingKey:
    """Device-resident proving key (g16_pkey): registered ProverPoints + CSR of the A/B matrices."""

    def __init__(self, ctx: Context, desc: PkeyDesc, keepalive, section4: bytes = None):
        """section4: the .zkey file's coefficient section as it lies on disk (g16_pkey_create_zkey); desc.coeffs must
        then be NULL"""
        self.ctx = ctx
        h = ctypes.c_void_p()
        if section4 is not None:
            ctx._check(ctx._lib.g16_pkey_create_zkey(ctx._h, ctypes.byref(desc), _buf(section4), len(section4),
                                                     ctypes.byref(h)))
        else:
            ctx._check(ctx._lib.g16_pkey_create(ctx._h, ctypes.byref(desc), ctypes.byref(h)))
        self._h = h
        self.nvars, self.npubs, self.log2n = desc.nvars, desc.npubs, desc.log2_domain
        ctx._children.add(self)
        del keepalive

    # Every call takes an optional `ctx`: the key is a per-DEVICE constant, and any context of its device may
    # prove against it -- several proofs in flight on one GPU = several contexts (private streams + workspaces)
    # sharing ONE resident key.  Default: the context the key was created through.
    def _check_len(self, witness):
        # the C ABI takes a bare pointer and reads nvars * 32 bytes: refuse a short host buffer here
        # (generateProofWithMask's "wrong witness length", prover.nim:236)
        if isinstance(witness, (bytes, bytearray)) and len(witness) != 32 * self.nvars:
            raise ValueError(f"wrong witness length: {len(witness)} bytes, expected {32 * self.nvars}")

    def prove(self, witness, mont: bool = True, r: bytes = None, s: bytes = None, device: bool = False, ctx=None):
        """-> (pi_a 64 B, pi_b 128 B, pi_c 64 B); r, s: Fr Montgomery bytes or None (trivial mask)."""
        c = ctx or self.ctx
        self._check_len(witness)
        out = ctypes.create_string_buffer(256)
        flags = (SCALARS_MONT if mont else 0) | (SCALARS_DEVICE if device else 0)
        c._check(c._lib.g16_prove(c._h, self._h, _buf(witness), flags, _buf(r) if r else None,
                                  _buf(s) if s else None, out))
        raw = out.raw
        return raw[0:64], raw[64:192], raw[192:256]

    def prove_partials(self, witness, mont: bool = True, device: bool = False, out=None, ctx=None):
        """this rank's five XYZZ MSM partials (768 bytes); out = device pointer to write them into HBM."""
        c = ctx or self.ctx
        self._check_len(witness)
        flags = (SCALARS_MONT if mont else 0) | (SCALARS_DEVICE if device else 0)
        if out is not None:
            c._check(c._lib.g16_prove_partials(c._h, self._h, _buf(witness), flags | OUT_DEVICE, _buf(out)))
            return None
        buf = ctypes.create_string_buffer(PARTIALS_BYTES)
        c._check(c._lib.g16_prove_partials(c._h, self._h, _buf(witness), flags, buf))
        return buf.raw

    def prove_partials_begin(self, witness, task_mask: int, task_out=None, mont: bool = True, device: bool = False,
                             ctx=None, nosync: bool = False):
        """first half of a sharded proof with a task-parallel quotient: launches this key's witness MSMs and writes
        the coset vectors named by task_mask (bit 0: A, 1: B, 2: C) to the device pointer task_out.  nosync: do not
        wait for the coset vectors on the host (G16_NO_HOST_SYNC: the caller orders its exchange behind them through
        the context's stream)"""
        c = ctx or self.ctx
        self._check_len(witness)
        flags = (SCALARS_MONT if mont else 0) | (SCALARS_DEVICE if device else 0) | (NO_HOST_SYNC if nosync else 0)
        c._check(c._lib.g16_prove_partials_begin(c._h, self._h, _buf(witness), flags, task_mask,
                                                 _buf(task_out) if task_out is not None else None))

    def prove_partials_end(self, a1, b1, c1, out=None, ctx=None, nosync: bool = False):
        """second half: a1 / b1 / c1 = device pointers to this key's [h_lo, h_hi) slices of the three coset vectors
        (None for an empty range); -> the 768-byte record (or written to the device pointer `out`)"""
        c = ctx or self.ctx
        ptr = lambda x: _buf(x) if x is not None else None          # noqa: E731
        if out is not None:
            c._check(c._lib.g16_prove_partials_end(c._h, self._h, ptr(a1), ptr(b1), ptr(c1),
                                                   OUT_DEVICE | (NO_HOST_SYNC if nosync else 0), _buf(out)))
            return None
        buf = ctypes.create_string_buffer(PARTIALS_BYTES)
        c._check(c._lib.g16_prove_partials_end(c._h, self._h, ptr(a1), ptr(b1), ptr(c1), 0, buf))
        return buf.raw

    def prove_combine(self, partials, count: int, r: bytes = None, s: bytes = None, device: bool = False, ctx=None):
        c = ctx or self.ctx
        out = ctypes.create_string_buffer(256)
        c._check(c._lib.g16_prove_combine(c._h, self._h, _buf(partials), count, SCALARS_DEVICE if device else 0,
                                          _buf(r) if r else None, _buf(s) if s else None, out))
        raw = out.raw
        return raw[0:64], raw[64:192], raw[192:256]

    def inf_counts(self) -> dict:
        """points at infinity per ProverPoints array (this shard) and whether A1 / B1+B2 use compacted entry lists"""
        out = (ctypes.c_size_t * 8)()
        self.ctx._check(self.ctx._lib.g16_pkey_inf_counts(self._h, out))
        v = list(out)
        return {"A1": v[0], "B1": v[1], "B2": v[2], "C1": v[3], "H1": v[4], "B1_and_B2": v[5],
                "compact_A": bool(v[6]), "compact_B": bool(v[7])}

    def abc_info(self) -> dict:
        """shape of the A / B matrices as buildABC runs them (g16_pkey_abc_info)"""
        out = (ctypes.c_size_t * 11)()
        self.ctx._check(self.ctx._lib.g16_pkey_abc_info(self._h, out))
        v = list(out)
        names = ("L<=1", "L=2", "L<=4", "L<=8", "L<=16", "L<=32", "L<=64", "L<=128", "L>128")
        return {"ncoeffs": v[0], "dict_values": v[1], "rows_by_terms": dict(zip(names, v[2:]))}

    def build_abc(self, witness: bytes, mont: bool = True, ctx=None):
        c = ctx or self.ctx
        self._check_len(witness)
        n = 1 << self.log2n
        out = ctypes.create_string_buffer(3 * n * 32)
        c._check(c._lib.g16_build_abc(c._h, self._h, _buf(witness), SCALARS_MONT if mont else 0, out))
        raw = out.raw
        return raw[: 32 * n], raw[32 * n: 64 * n], raw[64 * n:]

    def _free(self):
        if self._h:                      # valid whether or not the creating context is still alive
            self.ctx._lib.g16_pkey_destroy(self._h)
        self._h = None

    destroy = _free

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


class DeviceGroup:
    """g16_group: one proof sharded over several GPUs from ONE process, one host thread per device inside the library
    (the reference's Taskpool shape, msm.nim:96-122).  devices: HIP ordinals, one per member (repeats allowed)."""

    def __init__(self, devices):
        self._lib = load_library()
        arr = (ctypes.c_int32 * len(devices))(*devices)
        h = ctypes.c_void_p()
        rc = self._lib.g16_group_create(arr, len(devices), ctypes.byref(h))
        if rc != G16_OK:
            raise G16Error(rc, "g16_group_create failed")
        self._h, self.devices = h, list(devices)

    def _check(self, rc):
        if rc != G16_OK:
            raise G16Error(rc, self._lib.g16_group_last_error(self._h).decode())

    def size(self) -> int:
        return self._lib.g16_group_size(self._h)

    def load_key(self, desc: PkeyDesc, keepalive=None) -> "GroupKey":
        h = ctypes.c_void_p()
        self._check(self._lib.g16_group_pkey_create(self._h, ctypes.byref(desc), ctypes.byref(h)))
        return GroupKey(self, h, desc.nvars)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.g16_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupKey:
    def __init__(self, group: DeviceGroup, handle, nvars: int):
        self.group, self._h, self.nvars = group, handle, nvars

    def prove(self, witness: bytes, mont: bool = True, r: bytes = None, s: bytes = None):
        """-> (pi_a, pi_b, pi_c): g16_group_prove, bit-identical to ProvingKey.prove on the unsharded key"""
        if len(witness) != 32 * self.nvars:
            raise ValueError(f"wrong witness length: {len(witness)} bytes, expected {32 * self.nvars}")
        out = ctypes.create_string_buffer(256)
        g = self.group
        g._check(g._lib.g16_group_prove(g._h, self._h, _buf(witness), SCALARS_MONT if mont else 0,
                                        _buf(r) if r else None, _buf(s) if s else None, out))
        raw = out.raw
        return raw[0:64], raw[64:192], raw[192:256]

    def destroy(self):
        if self._h:
            self.group._lib.g16_group_pkey_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class VerifyingKey:
    """Device-resident verification key (g16_vkey): IC points, gamma2, delta2 and the Miller value of
    (alpha1, beta2) -- the reference's VKey (zkey_types.nim:62-73)."""

    def __init__(self, ctx: Context, npubs: int, alpha1: bytes, beta2: bytes, gamma2: bytes, delta2: bytes,
                 pointsIC: bytes):
        assert len(alpha1) == 64 and len(beta2) == len(gamma2) == len(delta2) == 128
        assert len(pointsIC) == 64 * (npubs + 1), "pointsIC must hold npubs + 1 points"
        self.ctx, self.npubs = ctx, npubs
        bufs = [ctypes.create_string_buffer(x, len(x)) for x in (alpha1, beta2, gamma2, delta2, pointsIC)]
        a = [ctypes.cast(b, ctypes.c_void_p) for b in bufs]
        desc = VkeyDesc(npubs, a[0], a[1], a[2], a[3], a[4])
        h = ctypes.c_void_p()
        ctx._check(ctx._lib.g16_vkey_create(ctx._h, ctypes.byref(desc), ctypes.byref(h)))
        self._h = h
        ctx._children.add(self)

    def verify(self, proofs, public_io: bytes, mont: bool = True, subgroup: bool = False):
        """proofs: list of (pi_a, pi_b, pi_c) byte triples; public_io: len(proofs) x (npubs+1) Fr scalars (each row
        starts with the constant 1, like Proof.publicIO).  -> list of status codes (1 ok, 0 fails, <0 malformed)."""
        n = len(proofs)
        assert len(public_io) == 32 * n * (self.npubs + 1)
        raw = b"".join(a + b + c for a, b, c in proofs)
        assert len(raw) == 256 * n
        st = (ctypes.c_int32 * max(n, 1))()
        flags = (SCALARS_MONT if mont else 0) | (VERIFY_SUBGROUP if subgroup else 0)
        self.ctx._check(self.ctx._lib.g16_verify(self.ctx._h, self._h, _buf(raw) if n else None,
                                                 _buf(public_io) if n else None, flags, n, st))
        return list(st)[:n]

    def _free(self):
        if self._h:
            self.ctx._lib.g16_vkey_destroy(self._h)
        self._h = None

    destroy = _free

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


class PointSet:
    """Device-resident point set with precomputed window tables (g16_points)."""

    def __init__(self, ctx: Context, handle, group: int, n: int):
        self.ctx, self._h, self.group, self.n = ctx, handle, group, n
        ctx._children.add(self)

    def inf_count(self) -> int:
        """points at infinity (0,0) in the set"""
        return self.ctx._lib.g16_points_inf_count(self._h)

    def info(self):
        """(window bits c, number of tables)"""
        c, w = ctypes.c_uint32(), ctypes.c_uint32()
        self.ctx._check(self.ctx._lib.g16_points_info(self._h, ctypes.byref(c), ctypes.byref(w)))
        return c.value, w.value

    def _free(self):
        if self._h:
            self.ctx._lib.g16_points_release(self._h)
        self._h = None

    release = _free

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx
