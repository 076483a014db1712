"""ctypes binding of include/hekaton.h (libhekaton.so) — the only door into the HIP kernels.

There is no CPU path: importing works anywhere (so host logic can be tested), but creating a
`Context` without a gfx950 device raises, and a missing libhekaton.so raises at load time.
"""
import ctypes as C
import os

import numpy as np

# hk_prove forks four side streams per lane; let the runtime map them onto more hardware queues than
# its default of 4 (must be set before the HIP runtime initialises; harmless if the user set it).
# Measured with 8 proofs in flight (apps/hk_all_in_one): 8 queues 89 proofs/s, 16: 121-122,
# 20-22: 123-125, 24 and more: 114-116 - hk_ctx_create exports the same value for hosts that do not come through here
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HK_LIB") or os.path.join(_HERE, "lib", "libhekaton.so")    # HK_LIB: experiment builds

HK_BN254, HK_BLS12_381 = 0, 1
HK_OK, HK_ERR_LEN, HK_ERR_DOMAIN_TOO_LARGE, HK_ERR_DEVICE, HK_ERR_ARG, HK_ERR_NOMEM = range(6)
CURVE_IDS = {"bn254": HK_BN254, "bls12_381": HK_BLS12_381}


class HekatonError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__("%s failed: %s" % (what, status_str(status)))


class hk_poseidon_desc(C.Structure):          # include/hekaton.h
    _fields_ = [("t", C.c_uint32), ("alpha", C.c_uint32), ("full_rounds", C.c_uint32), ("partial_rounds", C.c_uint32),
                ("consts_offset", C.c_uint32)]


class hk_csr(C.Structure):
    _fields_ = [("row_ptr", C.c_void_p), ("col", C.c_void_p), ("val_mont", C.c_void_p),
                ("n_rows", C.c_size_t), ("nnz", C.c_size_t)]


class hk_pk_desc(C.Structure):
    _fields_ = [("a_g", C.c_void_p), ("a_len", C.c_size_t),
                ("b_g", C.c_void_p), ("b_g_len", C.c_size_t),
                ("b_h", C.c_void_p), ("b_h_len", C.c_size_t),
                ("h_g", C.c_void_p), ("h_len", C.c_size_t),
                ("ck_stage", C.POINTER(C.c_void_p)), ("ck_len", C.POINTER(C.c_size_t)),
                ("n_stages", C.c_size_t),
                ("deltas_g", C.c_void_p), ("last_delta_h", C.c_void_p),
                ("alpha_g", C.c_void_p), ("beta_g", C.c_void_p), ("beta_h", C.c_void_p),
                ("A", C.POINTER(hk_csr)), ("B", C.POINTER(hk_csr)), ("C", C.POINTER(hk_csr)),
                ("n_inst", C.c_size_t), ("n_constraints", C.c_size_t)]


class hk_timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("total_ms", "digits_ms", "msm_a_ms", "msm_b_g1_ms", "msm_b_g2_ms", "msm_l_ms",
                 "witness_map_ms", "msm_h_ms", "finish_ms", "accum_kernel_ms")] + \
               [("accum_kernel_launches", C.c_uint32), ("accum_h_ms", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/hekaton.h declares (tests check the library exports all of them)
EXPORTS = ["hk_status_str", "hk_version", "hk_ctx_create", "hk_ctx_destroy", "hk_ctx_sync",
           "hk_ctx_set_profiling", "hk_ctx_last_timings", "hk_ctx_sizes", "hk_dev_alloc", "hk_dev_free",
           "hk_dev_upload", "hk_dev_download", "hk_msm_g1", "hk_msm_g2", "hk_ntt", "hk_witness_map",
           "hk_pk_upload", "hk_pk_free", "hk_commit", "hk_prove", "hk_fixed_base_g1", "hk_fixed_base_g2", "hk_scalar_pairing_g1", "hk_scalar_pairing_g2", "hk_field_convert", "hk_bases_upload", "hk_bases_free",
           "hk_msm_bases", "hk_multi_pairing", "hk_pairing_products", "hk_ctx_gt_bytes",
           "hk_points_lincomb_g1", "hk_points_lincomb_g2", "hk_points_fold_g2", "hk_points_fold_g1", "hk_points_fold_many_g1", "hk_points_fold_many_g2", "hk_pairing_pairs", "hk_keccak_f1600", "hk_assignment_from_bits", "hk_wprog_upload", "hk_wprog_free", "hk_wprog_run", "hk_gt_pow", "hk_fq12_pow", "hk_gt_pow_prod", "hk_poseidon_path", "hk_assignment_scatter", "hk_commit_batch"]

_lib = None


def load():
    """Loads libhekaton.so (built by __graft_entry__.build()); fails loudly when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libhekaton.so not built (%s): run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` — there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    lib.hk_status_str.restype = C.c_char_p
    lib.hk_status_str.argtypes = [i]
    lib.hk_version.restype = C.c_char_p
    lib.hk_ctx_create.argtypes = [i, i, C.POINTER(vp)]
    lib.hk_ctx_destroy.argtypes = [vp]
    lib.hk_ctx_destroy.restype = None
    lib.hk_ctx_sync.argtypes = [vp]
    lib.hk_ctx_set_profiling.argtypes = [vp, i]
    lib.hk_ctx_last_timings.argtypes = [vp, C.POINTER(hk_timings)]
    lib.hk_ctx_sizes.argtypes = [vp] + [C.POINTER(sz)] * 4
    lib.hk_dev_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.hk_dev_free.argtypes = [vp, vp]
    lib.hk_dev_upload.argtypes = [vp, vp, vp, sz]
    lib.hk_dev_download.argtypes = [vp, vp, vp, sz]
    for f in (lib.hk_msm_g1, lib.hk_msm_g2):
        f.argtypes = [vp, vp, sz, vp, sz, i, i, vp]
    lib.hk_ntt.argtypes = [vp, vp, C.c_uint, i, i]
    for f in (lib.hk_fixed_base_g1, lib.hk_fixed_base_g2):
        f.argtypes = [vp, vp, vp, sz, i, vp]
    for f in (lib.hk_scalar_pairing_g1, lib.hk_scalar_pairing_g2):
        f.argtypes = [vp, vp, vp, sz, vp]
    lib.hk_field_convert.argtypes = [vp, i, vp, vp, sz, i]
    lib.hk_bases_upload.argtypes = [vp, i, vp, sz, C.POINTER(vp)]
    lib.hk_bases_free.argtypes = [vp]
    lib.hk_bases_free.restype = None
    lib.hk_msm_bases.argtypes = [vp, vp, vp, sz, i, i, vp]
    if hasattr(lib, "hk_multi_pairing"):          # absent only from older experiment builds loaded through HK_LIB
        lib.hk_multi_pairing.argtypes = [vp, vp, vp, sz, vp]
        lib.hk_pairing_products.argtypes = [vp, C.POINTER(vp), sz, C.POINTER(vp), sz, sz, vp]
        lib.hk_pairing_pairs.argtypes = [vp, C.POINTER(vp), sz, C.POINTER(vp), sz, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), sz, sz, vp]
        lib.hk_ctx_gt_bytes.argtypes = [vp, C.POINTER(sz)]
        for f in (lib.hk_points_lincomb_g1, lib.hk_points_lincomb_g2):
            f.argtypes = [vp, C.POINTER(vp), vp, sz, sz, vp]
    if hasattr(lib, "hk_points_fold_g2"):
        lib.hk_points_fold_g2.argtypes = [vp, vp, vp, vp, C.c_uint, sz, vp]
    if hasattr(lib, "hk_points_fold_g1"):
        lib.hk_points_fold_g1.argtypes = [vp, vp, vp, vp, C.c_uint, sz, vp]
    for name in ("hk_points_fold_many_g1", "hk_points_fold_many_g2"):
        getattr(lib, name).argtypes = [vp, sz, C.POINTER(vp), C.POINTER(vp), vp, C.c_uint, sz, C.POINTER(vp)]
    if hasattr(lib, "hk_assignment_from_bits"):
        lib.hk_assignment_from_bits.argtypes = [vp, vp, sz, vp, vp, sz, vp]
        lib.hk_wprog_upload.argtypes = [vp, vp, sz, vp, sz, vp, sz, sz, sz, C.POINTER(vp)]
        lib.hk_wprog_free.argtypes = [vp]
        lib.hk_wprog_free.restype = None
        lib.hk_wprog_run.argtypes = [vp, vp, vp, sz, vp, vp, sz, vp]
        lib.hk_assignment_scatter.argtypes = [vp, vp, vp, sz, sz, sz, vp]
    if hasattr(lib, "hk_gt_pow"):
        lib.hk_gt_pow.argtypes = [vp, vp, vp, sz, vp]
        lib.hk_fq12_pow.argtypes = [vp, vp, vp, sz, vp]
        lib.hk_gt_pow_prod.argtypes = [vp, vp, vp, sz, sz, C.c_int, vp]
    lib.hk_poseidon_path.argtypes = [vp, vp, sz, C.POINTER(hk_poseidon_desc), C.POINTER(hk_poseidon_desc), vp, vp, vp, sz, sz,
                                     sz, sz, vp]
    lib.hk_witness_map.argtypes = [vp, C.POINTER(hk_csr), C.POINTER(hk_csr), C.POINTER(hk_csr), sz, sz,
                                   vp, sz, vp, sz, C.POINTER(sz)]
    lib.hk_pk_upload.argtypes = [vp, C.POINTER(hk_pk_desc), C.POINTER(vp)]
    lib.hk_pk_free.argtypes = [vp]
    lib.hk_pk_free.restype = None
    lib.hk_commit.argtypes = [vp, vp, sz, vp, sz, vp, vp]
    lib.hk_commit_batch.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp]
    lib.hk_prove.argtypes = [vp, vp, vp, sz, vp, vp, vp, sz, vp, vp, vp]
    _lib = lib
    return lib


def status_str(s):
    return load().hk_status_str(int(s)).decode()


def check(status, what):
    if status != HK_OK:
        raise HekatonError(status, what)


def ptr(x):
    """c_void_p of a numpy array / DeviceBuffer / bytes-like / int / None."""
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return C.c_void_p(x.ptr)
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return C.c_void_p(x.ctypes.data)
    if isinstance(x, int):
        return C.c_void_p(x)
    raise TypeError(type(x))


class DeviceBuffer:
    """HBM allocation owned through the C ABI (hk_dev_alloc/free)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(load().hk_dev_alloc(ctx.handle, self.nbytes, C.byref(p)), "hk_dev_alloc")
        self.ptr = p.value

    @classmethod
    def from_host(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(ctx, arr.nbytes)
        check(load().hk_dev_upload(ctx.handle, buf.ptr, arr.ctypes.data, arr.nbytes), "hk_dev_upload")
        return buf

    def to_host(self):
        out = np.empty(self.nbytes, dtype=np.uint8)
        check(load().hk_dev_download(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes),
              "hk_dev_download")
        return out

    def free(self):
        if self.ptr:
            load().hk_dev_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def view(self, offset, nbytes):
        """A window of this allocation, usable wherever a DeviceBuffer is (it owns nothing: free() is a no-op)."""
        assert 0 <= offset and offset + nbytes <= self.nbytes
        return DeviceView(self.ctx, self.ptr + int(offset), int(nbytes))


class DeviceView(DeviceBuffer):
    def __init__(self, ctx, ptr_, nbytes):            # noqa: super().__init__ would allocate
        self.ctx, self.ptr, self.nbytes = ctx, ptr_, nbytes

    def free(self):
        pass


class Context:
    """hk_ctx wrapper: one per (process, device)."""

    def __init__(self, curve="bn254", device=0):
        self.lib = load()
        self.curve = curve
        h = C.c_void_p()
        check(self.lib.hk_ctx_create(CURVE_IDS[curve], int(device), C.byref(h)), "hk_ctx_create")
        self.handle = h
        fr, fq, g1, g2 = (C.c_size_t() for _ in range(4))
        check(self.lib.hk_ctx_sizes(h, C.byref(fr), C.byref(fq), C.byref(g1), C.byref(g2)), "hk_ctx_sizes")
        self.fr_bytes, self.fq_bytes, self.g1_bytes, self.g2_bytes = fr.value, fq.value, g1.value, g2.value
        gt = C.c_size_t(12 * fq.value)
        if hasattr(self.lib, "hk_ctx_gt_bytes"):
            check(self.lib.hk_ctx_gt_bytes(h, C.byref(gt)), "hk_ctx_gt_bytes")
        self.gt_bytes = gt.value

    def close(self):
        if self.handle:
            self.lib.hk_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        check(self.lib.hk_ctx_sync(self.handle), "hk_ctx_sync")

    def set_profiling(self, on=True):
        check(self.lib.hk_ctx_set_profiling(self.handle, int(bool(on))), "hk_ctx_set_profiling")

    def last_timings(self):
        t = hk_timings()
        check(self.lib.hk_ctx_last_timings(self.handle, C.byref(t)), "hk_ctx_last_timings")
        return t.as_dict()

    # ---- primitives ---------------------------------------------------------------------------
    def _msm(self, fn, nbytes, bases, n_bases, scalars, n_scalars, mont, checked):
        out = np.zeros(nbytes, dtype=np.uint8)
        check(fn(self.handle, ptr(bases), n_bases, ptr(scalars), n_scalars, int(mont), int(checked),
                 out.ctypes.data), fn.__name__)
        return out

    def msm_g1(self, bases, scalars, n_bases=None, n_scalars=None, montgomery=True, checked=True):
        """E::G1::msm / msm_bigint / msm_unchecked (prover.rs:88,97,117,129; committer.rs:89,113)."""
        nb = n_bases if n_bases is not None else len(bases) // self.g1_bytes
        ns = n_scalars if n_scalars is not None else len(scalars) // self.fr_bytes
        return self._msm(self.lib.hk_msm_g1, self.g1_bytes, bases, nb, scalars, ns, montgomery, checked)

    def msm_g2(self, bases, scalars, n_bases=None, n_scalars=None, montgomery=True, checked=True):
        """E::G2 MSM (prover.rs:107)."""
        nb = n_bases if n_bases is not None else len(bases) // self.g2_bytes
        ns = n_scalars if n_scalars is not None else len(scalars) // self.fr_bytes
        return self._msm(self.lib.hk_msm_g2, self.g2_bytes, bases, nb, scalars, ns, montgomery, checked)

    def fixed_base(self, group, base, scalars, n=None, montgomery=True, out=None):
        """FixedBase::msm + normalize_batch (generator.rs:134-224): out[i] = scalars[i] * base.
        `out` may be a DeviceBuffer (stays in HBM); otherwise a numpy array is returned."""
        pb = self.g1_bytes if group == 1 else self.g2_bytes
        n = n if n is not None else len(scalars) // self.fr_bytes
        fn = self.lib.hk_fixed_base_g1 if group == 1 else self.lib.hk_fixed_base_g2
        base = np.ascontiguousarray(base, dtype=np.uint8)
        res = out if out is not None else np.zeros(n * pb, dtype=np.uint8)
        check(fn(self.handle, base.ctypes.data, ptr(scalars), n, int(montgomery), ptr(res)), fn.__name__)
        return res

    def scalar_pairing(self, group, points, scalars, n=None, out=None):
        """`scalar_pairing` (distributed-prover/src/pairing_ops.rs:32-39): out[i] = scalars[i] * points[i].  `out` may be a
        DeviceBuffer (the result stays in HBM)."""
        pb = self.g1_bytes if group == 1 else self.g2_bytes
        n = n if n is not None else len(scalars) // self.fr_bytes
        fn = self.lib.hk_scalar_pairing_g1 if group == 1 else self.lib.hk_scalar_pairing_g2
        res = out if out is not None else np.zeros(n * pb, dtype=np.uint8)
        check(fn(self.handle, ptr(points), ptr(scalars), n, ptr(res)), fn.__name__)
        return res

    def gt_pow(self, gts, scalars, in_gt=True):
        """hk_gt_pow: element-wise powers in GT (the exponent split along the Frobenius - the bases must have order r);
        in_gt=False: hk_fq12_pow, the plain chain for values of unknown provenance (a verifier's inputs).
        gts: (n, gt_bytes) uint8; scalars: n Fr Montgomery bytes."""
        gts = np.ascontiguousarray(gts, dtype=np.uint8).reshape(-1, self.gt_bytes)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
        out = np.zeros_like(gts)
        fn = self.lib.hk_gt_pow if in_gt else self.lib.hk_fq12_pow
        check(fn(self.handle, gts.ctypes.data, scalars.ctypes.data, gts.shape[0], out.ctypes.data), fn.__name__)
        return out

    def gt_pow_prod(self, gts, scalars, group_len, in_gt=True):
        """hk_gt_pow_prod: out[g] = prod_j gts[g * group_len + j]^scalars[g * group_len + j] (a verifier's
        multi-exponentiations).  Returns (n / group_len, gt_bytes) uint8."""
        gts = np.ascontiguousarray(gts, dtype=np.uint8).reshape(-1, self.gt_bytes)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
        n = gts.shape[0]
        out = np.zeros((n // max(1, group_len), self.gt_bytes), dtype=np.uint8)
        check(self.lib.hk_gt_pow_prod(self.handle, gts.ctypes.data, scalars.ctypes.data, n, group_len, 1 if in_gt else 0,
                                      out.ctypes.data), "hk_gt_pow_prod")
        return out

    def multi_pairing(self, g1, g2, n=None):
        """`pairing(left, right)` (distributed-prover/src/pairing_ops.rs:25-29): prod_i e(g1[i], g2[i]) as ark's Fp12
        bytes (12 Fq, Montgomery)."""
        n = n if n is not None else len(g1) // self.g1_bytes
        out = np.zeros(self.gt_bytes, dtype=np.uint8)
        check(self.lib.hk_multi_pairing(self.handle, ptr(g1) if n else None, ptr(g2) if n else None, n, out.ctypes.data),
              "hk_multi_pairing")
        return out

    def pairing_products(self, lhs, rhs, n=None):
        """Every G1 vector of `lhs` against every G2 vector of `rhs` in one batched launch (the `cross_terms` of
        aggregation.rs:255-263): returns a (len(lhs), len(rhs), gt_bytes) uint8 array."""
        n = n if n is not None else len(lhs[0]) // self.g1_bytes
        keep = [x if isinstance(x, DeviceBuffer) else np.ascontiguousarray(x, dtype=np.uint8) for x in list(lhs) + list(rhs)]
        addr = lambda x: x.ptr if isinstance(x, DeviceBuffer) else x.ctypes.data
        lp = (C.c_void_p * len(lhs))(*[addr(x) for x in keep[:len(lhs)]])
        rp = (C.c_void_p * len(rhs))(*[addr(x) for x in keep[len(lhs):]])
        out = np.zeros((len(lhs), len(rhs), self.gt_bytes), dtype=np.uint8)
        check(self.lib.hk_pairing_products(self.handle, lp, len(lhs), rp, len(rhs), n, out.ctypes.data),
              "hk_pairing_products")
        return out

    def pairing_pairs(self, lhs, rhs, pairs, n=None):
        """pairing(lhs[a], rhs[b]) for each (a, b) of `pairs` in one batched launch (hk_pairing_pairs: the cross terms of a
        GIPA round): returns a (len(pairs), gt_bytes) uint8 array."""
        n = n if n is not None else len(lhs[0]) // self.g1_bytes
        keep = [x if isinstance(x, DeviceBuffer) else np.ascontiguousarray(x, dtype=np.uint8) for x in list(lhs) + list(rhs)]
        addr = lambda x: x.ptr if isinstance(x, DeviceBuffer) else x.ctypes.data
        lp = (C.c_void_p * len(lhs))(*[addr(x) for x in keep[:len(lhs)]])
        rp = (C.c_void_p * len(rhs))(*[addr(x) for x in keep[len(lhs):]])
        pa = (C.c_uint32 * len(pairs))(*[a for a, _ in pairs])
        pb = (C.c_uint32 * len(pairs))(*[b for _, b in pairs])
        out = np.zeros((len(pairs), self.gt_bytes), dtype=np.uint8)
        check(self.lib.hk_pairing_pairs(self.handle, lp, len(lhs), rp, len(rhs), pa, pb, len(pairs), n, out.ctypes.data),
              "hk_pairing_pairs")
        return out

    def points_lincomb(self, group, vecs, coeffs, n=None):
        """out[i] = sum_j coeffs[j] * vecs[j][i] (aggregation.rs:192-203,293-326); coeffs: k Fr Montgomery bytes."""
        pb = self.g1_bytes if group == 1 else self.g2_bytes
        n = n if n is not None else len(vecs[0]) // pb
        keep = [x if isinstance(x, DeviceBuffer) else np.ascontiguousarray(x, dtype=np.uint8) for x in vecs]
        vp_ = (C.c_void_p * len(keep))(*[x.ptr if isinstance(x, DeviceBuffer) else x.ctypes.data for x in keep])
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint8)
        out = np.zeros(n * pb, dtype=np.uint8)
        fn = self.lib.hk_points_lincomb_g1 if group == 1 else self.lib.hk_points_lincomb_g2
        check(fn(self.handle, vp_, coeffs.ctypes.data, len(keep), n, out.ctypes.data), fn.__name__)
        return out

    def points_fold_g2(self, lo, hi, c, n=None, out=None):
        """out[i] = lo[i] + c * hi[i] in G2 (the fold of a TIPA round) through hk_points_fold_g2: the scalar c (an int mod r)
        is split along the endomorphism psi into four ~64-bit parts (endo.Psi4), so the element-wise double-and-add chain is
        ~66 steps instead of 254."""
        from .cp_groth16 import FrCodec
        from .endo import psi4
        n = n if n is not None else len(hi) // self.g2_bytes
        k = psi4(self.curve).decompose(c)
        neg = sum(1 << j for j, v in enumerate(k) if v < 0)
        coeffs = np.ascontiguousarray(FrCodec(self.curve).enc([abs(v) for v in k]), dtype=np.uint8)
        lo_, hi_ = (x if isinstance(x, DeviceBuffer) else np.ascontiguousarray(x, dtype=np.uint8) for x in (lo, hi))
        ptr = lambda x: x.ptr if isinstance(x, DeviceBuffer) else x.ctypes.data
        res = out if out is not None else np.zeros(n * self.g2_bytes, dtype=np.uint8)          # a DeviceBuffer stays in HBM
        check(self.lib.hk_points_fold_g2(self.handle, ptr(lo_), ptr(hi_), coeffs.ctypes.data, neg, n,
                                         res.ptr if isinstance(res, DeviceBuffer) else res.ctypes.data), "hk_points_fold_g2")
        return res

    def points_fold_g1(self, lo, hi, c, n=None, out=None):
        """out[i] = lo[i] + c * hi[i] in G1 through hk_points_fold_g1: c split along the GLV endomorphism into two ~128-bit
        parts (endo.Phi2)."""
        from .cp_groth16 import FrCodec
        from .endo import phi2
        n = n if n is not None else len(hi) // self.g1_bytes
        k = phi2(self.curve).decompose(c)
        neg = sum(1 << j for j, v in enumerate(k) if v < 0)
        coeffs = np.ascontiguousarray(FrCodec(self.curve).enc([abs(v) for v in k]), dtype=np.uint8)
        lo_, hi_ = (x if isinstance(x, DeviceBuffer) else np.ascontiguousarray(x, dtype=np.uint8) for x in (lo, hi))
        ptr = lambda x: x.ptr if isinstance(x, DeviceBuffer) else x.ctypes.data
        res = out if out is not None else np.zeros(n * self.g1_bytes, dtype=np.uint8)          # a DeviceBuffer stays in HBM
        check(self.lib.hk_points_fold_g1(self.handle, ptr(lo_), ptr(hi_), coeffs.ctypes.data, neg, n,
                                         res.ptr if isinstance(res, DeviceBuffer) else res.ctypes.data), "hk_points_fold_g1")
        return res

    def points_fold_many(self, group, los, his, c, n, outs):
        """outs[y][i] = los[y][i] + c * his[y][i] for up to 4 vector pairs and one scalar c (hk_points_fold_many_g1 / _g2):
        the folds of one TIPA round that share a challenge.  los / his: DeviceBuffer / DeviceView or uint8 arrays; outs:
        DeviceBuffer / DeviceView (stay in HBM) or uint8 arrays of n points each."""
        from .cp_groth16 import FrCodec
        from .endo import phi2, psi4
        k = (phi2 if group == 1 else psi4)(self.curve).decompose(c)
        neg = sum(1 << j for j, v in enumerate(k) if v < 0)
        coeffs = np.ascontiguousarray(FrCodec(self.curve).enc([abs(v) for v in k]), dtype=np.uint8)
        keep = [[x if isinstance(x, DeviceBuffer) else np.ascontiguousarray(x, dtype=np.uint8) for x in xs] for xs in (los, his)]
        ptr = lambda x: x.ptr if isinstance(x, DeviceBuffer) else x.ctypes.data
        arr = lambda xs: (C.c_void_p * len(xs))(*[ptr(x) for x in xs])
        fn = self.lib.hk_points_fold_many_g1 if group == 1 else self.lib.hk_points_fold_many_g2
        check(fn(self.handle, len(outs), arr(keep[0]), arr(keep[1]), coeffs.ctypes.data, neg, n, arr(outs)), fn.__name__)
        return outs

    def assignment_from_bits(self, bits, full_cols, full_vals, out=None):
        """hk_assignment_from_bits: the Montgomery assignment of a bit-valued witness, materialised in HBM.  bits: uint8
        array (one per variable); full_cols: column indices of the full-width values; full_vals: their Montgomery bytes.
        Returns a DeviceBuffer of n_v Fr (or fills `out`)."""
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        cols = np.ascontiguousarray(full_cols, dtype=np.uint32)
        vals = np.ascontiguousarray(full_vals, dtype=np.uint8)
        n_v = bits.size
        buf = out if out is not None else DeviceBuffer(self, n_v * self.fr_bytes)
        check(self.lib.hk_assignment_from_bits(self.handle, bits.ctypes.data, n_v, cols.ctypes.data if cols.size else None,
                                               vals.ctypes.data if cols.size else None, cols.size, buf.ptr),
              "hk_assignment_from_bits")
        return buf

    def poseidon_path(self, params, leaf, siblings, index, n_v, col0, z_out):
        """hk_poseidon_path: the membership block of `batch` assignments, written on the device.  params:
        (consts DeviceBuffer or Montgomery bytes, n_consts, (t, alpha, rf, rp, off) of the leaf hash, same of the node hash)
        - poseidon.device_params(curve); leaf: Montgomery bytes (batch, 4 Fr); siblings: (batch, depth Fr); index: uint32
        (batch); z_out: DeviceBuffer (or raw device address) of batch x n_v Fr."""
        consts, n_consts, ld, nd = params
        leaf = np.ascontiguousarray(leaf, dtype=np.uint8)
        batch = leaf.shape[0]
        siblings = np.ascontiguousarray(siblings, dtype=np.uint8).reshape(batch, -1)
        depth = siblings.shape[1] // self.fr_bytes
        index = np.ascontiguousarray(index, dtype=np.uint32)
        a, b = hk_poseidon_desc(*ld), hk_poseidon_desc(*nd)
        zp = z_out.ptr if isinstance(z_out, DeviceBuffer) else int(z_out)
        check(self.lib.hk_poseidon_path(self.handle, ptr(consts), int(n_consts), C.byref(a), C.byref(b), leaf.ctypes.data,
                                        siblings.ctypes.data if depth else None, index.ctypes.data, depth, batch, int(n_v),
                                        int(col0), zp), "hk_poseidon_path")

    def wprog_upload(self, ops, refs, vmap, n_values, n_inputs):
        """hk_wprog_upload: a class's word program (sha_circuit.Tape.word_program) resident on the device."""
        ops = np.ascontiguousarray(ops, dtype=np.uint32)
        refs = np.ascontiguousarray(refs, dtype=np.uint32)
        vmap = np.ascontiguousarray(vmap, dtype=np.uint32)
        h = C.c_void_p()
        check(self.lib.hk_wprog_upload(self.handle, ops.ctypes.data, ops.shape[0], refs.ctypes.data if refs.size else None,
                                       refs.size, vmap.ctypes.data, vmap.size, int(n_values), int(n_inputs), C.byref(h)),
              "hk_wprog_upload")
        return WordProgram(self, h, vmap.size, int(n_inputs))

    def bases_upload(self, group, bases, n=None):
        """Makes a static base set resident with its shift tables (hk_bases_upload); returns a ResidentBases."""
        pb = self.g1_bytes if group == 1 else self.g2_bytes
        n = n if n is not None else len(bases) // pb
        h = C.c_void_p()
        check(self.lib.hk_bases_upload(self.handle, int(group), ptr(bases), n, C.byref(h)), "hk_bases_upload")
        return ResidentBases(self, h, group, n)

    def field_convert(self, which, data, to_mont):
        """Montgomery <-> canonical for a packed array of Fr (`which` = 0) or Fq (1) elements: what ark-ff
        `from_bigint` / `into_bigint` do under ark-serialize.  Returns a new numpy uint8 array."""
        eb = self.fr_bytes if which == 0 else self.fq_bytes
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        n = data.size // eb
        out = np.empty(n * eb, dtype=np.uint8)
        check(self.lib.hk_field_convert(self.handle, int(which), data.ctypes.data, out.ctypes.data, n, int(to_mont)),
              "hk_field_convert")
        return out

    def ntt(self, data, log_m, inverse=False, coset=False):
        """In-place on `data` (numpy uint8 array of 2^log_m Fr or a DeviceBuffer)."""
        check(self.lib.hk_ntt(self.handle, ptr(data), int(log_m), int(inverse), int(coset)), "hk_ntt")
        return data

    def witness_map(self, A, B, Cm, n_inst, n_constraints, z, n_v=None):
        """R1CSToQAP::witness_map (prover.rs:123).  A/B/Cm: (row_ptr u64, col u32, val bytes) triples.
        Returns (h bytes [m Fr, natural order], m)."""
        keep = []
        csrs = []
        for (rp, col, val) in (A, B, Cm):
            rp = np.ascontiguousarray(rp, dtype=np.uint64)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            val = np.ascontiguousarray(val, dtype=np.uint8)
            keep += [rp, col, val]
            csrs.append(hk_csr(rp.ctypes.data, col.ctypes.data, val.ctypes.data, len(rp) - 1, len(col)))
        nv = n_v if n_v is not None else len(z) // self.fr_bytes
        m = 1
        while m < n_constraints + n_inst:
            m *= 2
        out = np.zeros(m * self.fr_bytes, dtype=np.uint8)
        m_out = C.c_size_t()
        check(self.lib.hk_witness_map(self.handle, C.byref(csrs[0]), C.byref(csrs[1]), C.byref(csrs[2]),
                                      n_inst, n_constraints, ptr(z), nv, out.ctypes.data, m, C.byref(m_out)),
              "hk_witness_map")
        return out, m_out.value

    def pk_upload(self, *, a_g, b_g, b_h, h_g, ck_stages, deltas_g, last_delta_h, alpha_g, beta_g, beta_h,
                  matrices=None, n_inst=0, n_constraints=0):
        """hk_pk_upload: all arguments packed-affine numpy uint8 arrays (or DeviceBuffers with explicit
        lengths via (buf, n) tuples).  matrices = (A, B, C) CSR triples as in witness_map."""
        def arr(x):
            return x if isinstance(x, DeviceBuffer) else np.ascontiguousarray(x, dtype=np.uint8)

        def addr(x):
            return x.ptr if isinstance(x, DeviceBuffer) else x.ctypes.data

        def nbytes(x):
            return x.nbytes
        keep = []
        d = hk_pk_desc()
        a_g, b_g, b_h, h_g = arr(a_g), arr(b_g), arr(b_h), arr(h_g)
        keep += [a_g, b_g, b_h, h_g]
        d.a_g, d.a_len = addr(a_g), nbytes(a_g) // self.g1_bytes
        d.b_g, d.b_g_len = addr(b_g), nbytes(b_g) // self.g1_bytes
        d.b_h, d.b_h_len = addr(b_h), nbytes(b_h) // self.g2_bytes
        d.h_g, d.h_len = addr(h_g), nbytes(h_g) // self.g1_bytes
        cks = [arr(c) for c in ck_stages]
        keep += cks
        ck_ptrs = (C.c_void_p * len(cks))(*[addr(c) for c in cks])
        ck_lens = (C.c_size_t * len(cks))(*[nbytes(c) // self.g1_bytes for c in cks])
        d.ck_stage, d.ck_len, d.n_stages = ck_ptrs, ck_lens, len(cks)
        small = [arr(x) for x in (deltas_g, last_delta_h, alpha_g, beta_g, beta_h)]
        keep += small
        d.deltas_g, d.last_delta_h, d.alpha_g, d.beta_g, d.beta_h = [s.ctypes.data for s in small]
        csrs = []
        if matrices is not None:
            for (rp, col, val) in matrices:
                rp = np.ascontiguousarray(rp, dtype=np.uint64)
                col = np.ascontiguousarray(col, dtype=np.uint32)
                val = arr(val)
                keep += [rp, col, val]
                csrs.append(hk_csr(rp.ctypes.data, col.ctypes.data, val.ctypes.data, len(rp) - 1, len(col)))
            d.A, d.B, d.C = C.pointer(csrs[0]), C.pointer(csrs[1]), C.pointer(csrs[2])
        d.n_inst, d.n_constraints = n_inst, n_constraints
        h = C.c_void_p()
        check(self.lib.hk_pk_upload(self.handle, C.byref(d), C.byref(h)), "hk_pk_upload")
        return DevicePk(self, h)


class WordProgram:
    """hk_wprog: witness generation on the device for one proving-key class."""

    def __init__(self, ctx, handle, n_v, n_inputs):
        self.ctx, self.handle, self.n_v, self.n_inputs = ctx, handle, n_v, n_inputs

    def run(self, inputs, full_cols, full_vals, out=None):
        """inputs: uint32 (batch, n_inputs); full_cols: uint32 (k); full_vals: Montgomery bytes (batch, k * fr_bytes).
        Returns a DeviceBuffer holding batch x n_v Fr (or fills `out`)."""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint32)
        batch = inputs.shape[0]
        cols = np.ascontiguousarray(full_cols, dtype=np.uint32)
        vals = np.ascontiguousarray(full_vals, dtype=np.uint8)
        buf = out if out is not None else DeviceBuffer(self.ctx, batch * self.n_v * self.ctx.fr_bytes)
        check(self.ctx.lib.hk_wprog_run(self.ctx.handle, self.handle, inputs.ctypes.data, batch,
                                        cols.ctypes.data if cols.size else None, vals.ctypes.data if cols.size else None,
                                        cols.size, buf.ptr), "hk_wprog_run")
        return buf

    def scatter(self, full_cols, full_vals, out):
        """The full-width values alone (hk_assignment_scatter), into assignments `run(inputs, [], [], out=...)` produced."""
        cols = np.ascontiguousarray(full_cols, dtype=np.uint32)
        vals = np.ascontiguousarray(full_vals, dtype=np.uint8)
        batch = vals.size // max(1, cols.size * self.ctx.fr_bytes)
        check(self.ctx.lib.hk_assignment_scatter(self.ctx.handle, cols.ctypes.data, vals.ctypes.data, cols.size, batch, self.n_v,
                                                 out.ptr), "hk_assignment_scatter")
        return out

    def free(self):
        if self.handle:
            self.ctx.lib.hk_wprog_free(self.handle)
            self.handle = None


class ResidentBases:
    """hk_bases: a static base set (SRS powers, commitment keys) with its shift tables in HBM."""

    def __init__(self, ctx, handle, group, n):
        self.ctx, self.handle, self.group, self.n = ctx, handle, group, n

    def msm(self, scalars, n_scalars=None, montgomery=True, checked=True):
        """`G::Group::msm(&srs_powers, &coeffs)` (distributed-prover/src/kzg.rs:151-152) over the resident bases."""
        ns = n_scalars if n_scalars is not None else len(scalars) // self.ctx.fr_bytes
        out = np.zeros(self.ctx.g1_bytes if self.group == 1 else self.ctx.g2_bytes, dtype=np.uint8)
        check(self.ctx.lib.hk_msm_bases(self.ctx.handle, self.handle, ptr(scalars), ns, int(montgomery), int(checked),
                                        out.ctypes.data), "hk_msm_bases")
        return out

    def free(self):
        if self.handle:
            self.ctx.lib.hk_bases_free(self.handle)
            self.handle = None


class DevicePk:
    """hk_pk wrapper: a proving-key class resident in HBM (with its shift tables and matrices)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.handle = handle

    def free(self):
        if self.handle:
            self.ctx.lib.hk_pk_free(self.handle)
            self.handle = None

    def commit(self, stage, w_stage, kappa, n=None):
        """committer.rs:87-91 — msm(ck[stage], w) + kappa * last_delta_g; returns packed G1 bytes."""
        ctx = self.ctx
        n = n if n is not None else len(w_stage) // ctx.fr_bytes
        kappa = np.ascontiguousarray(kappa, dtype=np.uint8)
        out = np.zeros(ctx.g1_bytes, dtype=np.uint8)
        check(ctx.lib.hk_commit(ctx.handle, self.handle, stage, ptr(w_stage) if n else None, n,
                                kappa.ctypes.data, out.ctypes.data), "hk_commit")
        return out

    def commit_batch(self, stage, w_rows, kappas, n, batch):
        """hk_commit_batch: the stage commitments of `batch` subcircuits of this key's class in one call.  w_rows: their
        stage witnesses row after row (batch x n Fr Montgomery; uint8 array or DeviceBuffer); kappas: batch Fr Montgomery
        (uint8 array).  Returns (batch, g1_bytes) uint8."""
        ctx = self.ctx
        kap = np.ascontiguousarray(kappas, dtype=np.uint8)
        out = np.zeros((batch, ctx.g1_bytes), dtype=np.uint8)
        check(ctx.lib.hk_commit_batch(ctx.handle, self.handle, stage, ptr(w_rows) if n else None, n, kap.ctypes.data, batch,
                                      out.ctypes.data), "hk_commit_batch")
        return out

    def prove(self, z, r, s, kappas, n_v=None):
        """prover.rs:78-155 + committer.rs:112-114; returns (a, b, c) packed affine bytes."""
        ctx = self.ctx
        n_v = n_v if n_v is not None else len(z) // ctx.fr_bytes
        r = np.ascontiguousarray(r, dtype=np.uint8)
        s = np.ascontiguousarray(s, dtype=np.uint8)
        kap = np.ascontiguousarray(kappas, dtype=np.uint8)
        nk = len(kap) // ctx.fr_bytes
        a = np.zeros(ctx.g1_bytes, dtype=np.uint8)
        b = np.zeros(ctx.g2_bytes, dtype=np.uint8)
        c = np.zeros(ctx.g1_bytes, dtype=np.uint8)
        check(ctx.lib.hk_prove(ctx.handle, self.handle, ptr(z), n_v, r.ctypes.data, s.ctypes.data,
                               kap.ctypes.data if nk else None, nk, a.ctypes.data, b.ctypes.data,
                               c.ctypes.data), "hk_prove")
        return a, b, c
