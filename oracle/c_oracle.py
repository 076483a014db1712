"""ctypes wrapper of oracle/c/hk_oracle.cpp (the multi-threaded CPU restatement).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "c", "hk_oracle.cpp")
BUILD_DIR = os.path.join(_HERE, "_build")
LIB = os.path.join(BUILD_DIR, "libhk_oracle.so")


def build(native=False, out=None):
    """g++ build of the oracle; `native` adds -march=native (use only on the machine that runs it)."""
    out = out or LIB
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread"]
    if native:
        cmd.append("-march=native")
    cmd += ["-o", out, SRC]
    subprocess.check_call(cmd)
    return out


class Csr(C.Structure):
    _fields_ = [("row_ptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p),
                ("n_rows", C.c_size_t), ("nnz", C.c_size_t)]


class PkView(C.Structure):
    _fields_ = [("a_g", C.c_void_p), ("b_g", C.c_void_p), ("b_h", C.c_void_p), ("h_g", C.c_void_p),
                ("n_v", C.c_size_t), ("h_len", C.c_size_t),
                ("ck", C.POINTER(C.c_void_p)), ("ck_len", C.POINTER(C.c_size_t)), ("n_stages", C.c_size_t),
                ("deltas_g", C.c_void_p), ("last_delta_h", C.c_void_p), ("alpha_g", C.c_void_p),
                ("beta_g", C.c_void_p), ("beta_h", C.c_void_p)]


_SIZES = {0: (32, 32), 1: (32, 48)}      # curve id -> (Fr bytes, Fq bytes)


class COracle:
    def __init__(self, curve="bn254", lib_path=None):
        path = lib_path or LIB
        if not os.path.exists(path):
            build(out=path)
        self.lib = C.CDLL(path)
        self.cid = {"bn254": 0, "bls12_381": 1}[curve]
        self.fr_bytes, self.fq_bytes = _SIZES[self.cid]
        self.g1_bytes, self.g2_bytes = 2 * self.fq_bytes, 4 * self.fq_bytes
        self.lib.hko_threads.restype = C.c_int

    def set_threads(self, t):
        self.lib.hko_set_threads(int(t))

    def threads(self):
        return self.lib.hko_threads()

    @staticmethod
    def _p(a):
        return C.c_void_p(a.ctypes.data) if a is not None else None

    def msm(self, group, bases, scalars, montgomery=True):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
        pb = self.g1_bytes if group == 1 else self.g2_bytes
        n = min(len(bases) // pb, len(scalars) // self.fr_bytes)
        out = np.zeros(pb, dtype=np.uint8)
        self.lib.hko_msm(self.cid, group, self._p(bases), C.c_size_t(n), self._p(scalars), int(montgomery),
                         self._p(out))
        return out

    def ntt(self, data, log_m, inverse=False, coset=False):
        rc = self.lib.hko_ntt(self.cid, self._p(data), C.c_uint(log_m), int(inverse), int(coset))
        if rc:
            raise ValueError("PolynomialDegreeTooLarge")
        return data

    def _csr(self, t, keep):
        rp = np.ascontiguousarray(t[0], dtype=np.uint64)
        col = np.ascontiguousarray(t[1], dtype=np.uint32)
        val = np.ascontiguousarray(t[2], dtype=np.uint8)
        keep += [rp, col, val]
        return Csr(rp.ctypes.data, col.ctypes.data, val.ctypes.data, len(rp) - 1, len(col))

    def witness_map(self, A, B, Cm, n_inst, n_c, z):
        keep = []
        a, b, c = (self._csr(t, keep) for t in (A, B, Cm))
        m = 1
        while m < n_c + n_inst:
            m *= 2
        z = np.ascontiguousarray(z, dtype=np.uint8)
        out = np.zeros(m * self.fr_bytes, dtype=np.uint8)
        mo = C.c_size_t()
        rc = self.lib.hko_witness_map(self.cid, C.byref(a), C.byref(b), C.byref(c), C.c_size_t(n_inst),
                                      C.c_size_t(n_c), self._p(z), self._p(out), C.byref(mo))
        if rc:
            raise ValueError("witness_map rc=%d" % rc)
        return out, mo.value

    def pk_view(self, *, a_g, b_g, b_h, h_g, ck_stages, deltas_g, last_delta_h, alpha_g, beta_g, beta_h):
        keep = [np.ascontiguousarray(x, dtype=np.uint8) for x in
                (a_g, b_g, b_h, h_g, deltas_g, last_delta_h, alpha_g, beta_g, beta_h)]
        cks = [np.ascontiguousarray(c, dtype=np.uint8) for c in ck_stages]
        v = PkView()
        v.a_g, v.b_g, v.b_h, v.h_g = (k.ctypes.data for k in keep[:4])
        v.n_v = len(keep[0]) // self.g1_bytes
        v.h_len = len(keep[3]) // self.g1_bytes
        v._ptrs = (C.c_void_p * len(cks))(*[c.ctypes.data for c in cks])
        v._lens = (C.c_size_t * len(cks))(*[len(c) // self.g1_bytes for c in cks])
        v.ck, v.ck_len, v.n_stages = v._ptrs, v._lens, len(cks)
        v.deltas_g, v.last_delta_h, v.alpha_g, v.beta_g, v.beta_h = (k.ctypes.data for k in keep[4:])
        v._keep = keep + cks
        return v

    def commit(self, pk, stage, w, kappa):
        w = np.ascontiguousarray(w, dtype=np.uint8)
        kappa = np.ascontiguousarray(kappa, dtype=np.uint8)
        out = np.zeros(self.g1_bytes, dtype=np.uint8)
        rc = self.lib.hko_commit(self.cid, C.byref(pk), C.c_size_t(stage), self._p(w),
                                 C.c_size_t(len(w) // self.fr_bytes), self._p(kappa), self._p(out))
        if rc:
            raise ValueError("commit rc=%d" % rc)
        return out

    def prove(self, pk, A, B, Cm, n_inst, n_c, z, r, s, kappas):
        keep = []
        a, b, c = (self._csr(t, keep) for t in (A, B, Cm))
        z, r, s, kap = (np.ascontiguousarray(x, dtype=np.uint8) for x in (z, r, s, kappas))
        oa = np.zeros(self.g1_bytes, np.uint8)
        ob = np.zeros(self.g2_bytes, np.uint8)
        oc = np.zeros(self.g1_bytes, np.uint8)
        rc = self.lib.hko_prove(self.cid, C.byref(pk), C.byref(a), C.byref(b), C.byref(c), C.c_size_t(n_inst),
                                C.c_size_t(n_c), self._p(z), self._p(r), self._p(s),
                                self._p(kap) if len(kap) else None, C.c_size_t(len(kap) // self.fr_bytes),
                                self._p(oa), self._p(ob), self._p(oc))
        if rc:
            raise ValueError("prove rc=%d" % rc)
        return oa, ob, oc

    def multi_pairing(self, g1, g2, n=None):
        """`pairing(left, right)` (distributed-prover/src/pairing_ops.rs:25-29), parallel over chunks of 4 pairs like
        ark's multi_miller_loop; returns the 12 Fq of ark's Fp12 (Montgomery bytes)."""
        g1 = np.ascontiguousarray(g1, dtype=np.uint8)
        g2 = np.ascontiguousarray(g2, dtype=np.uint8)
        n = n if n is not None else len(g1) // self.g1_bytes
        out = np.zeros(12 * self.fq_bytes, dtype=np.uint8)
        self.lib.hko_multi_pairing(self.cid, self._p(g1), self._p(g2), C.c_size_t(n), self._p(out))
        return out

    def scalar_mul_each(self, group, points, scalars):
        """`scalar_pairing` (distributed-prover/src/pairing_ops.rs:32-39): out[i] = scalars[i] * points[i]."""
        points = np.ascontiguousarray(points, dtype=np.uint8)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
        n = len(scalars) // self.fr_bytes
        out = np.zeros(len(points), dtype=np.uint8)
        self.lib.hko_scalar_mul_each(self.cid, group, self._p(points), self._p(scalars), C.c_size_t(n), self._p(out))
        return out

    def running_bases(self, group, gen_affine, s0, n):
        gen = np.ascontiguousarray(gen_affine, dtype=np.uint8)
        pb = self.g1_bytes if group == 1 else self.g2_bytes
        out = np.zeros(n * pb, dtype=np.uint8)
        self.lib.hko_running_bases(self.cid, group, self._p(gen), C.c_uint64(s0), C.c_size_t(n), self._p(out))
        return out
