"""ctypes loaders for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

* ``port()``  -> oracle/libenoki_oracle.so   (plain-C restatement, oracle/enoki_oracle.c)
* ``ref()``   -> oracle/_ref/libenoki_ref.so (the unmodified reference headers, oracle/ref_driver.cpp)

Both expose the same calling convention (op names as C strings, type codes as in
include/enoki_hip.h), with the prefix ``orc_`` resp. ``ref_``; :class:`Checker` hides the prefix.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

T_BOOL, T_I32, T_U32, T_I64, T_U64, T_F32, T_F64 = range(7)
NP2T = {np.dtype(np.int32): T_I32, np.dtype(np.uint32): T_U32, np.dtype(np.int64): T_I64,
        np.dtype(np.uint64): T_U64, np.dtype(np.float32): T_F32, np.dtype(np.float64): T_F64,
        np.dtype(np.uint8): T_BOOL, np.dtype(np.bool_): T_BOOL}
T2NP = {T_BOOL: np.uint8, T_I32: np.int32, T_U32: np.uint32, T_I64: np.int64, T_U64: np.uint64,
        T_F32: np.float32, T_F64: np.float64}


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Checker:
    def __init__(self, lib, prefix, kind):
        self.lib, self.prefix, self.kind = lib, prefix, kind
        for name in ("cfg1", "cfg2", "cfg3a", "cfg3b"):
            if hasattr(lib, prefix + name):          # the scalar flavour (ref_scalar_driver.cpp) has the elementwise entry points only
                getattr(lib, prefix + name).restype = ctypes.c_float

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def _chk(self, rc, what):
        if rc != 0:
            raise NotImplementedError(f"{self.kind}: {what} -> rc={rc}")

    def unary(self, op, a):
        a = np.ascontiguousarray(a); out = np.empty_like(a)
        self._chk(self._f("unary")(NP2T[a.dtype], op.encode(), _p(a), _p(out), ctypes.c_size_t(a.size)), op)
        return out

    def sincos(self, a):
        a = np.ascontiguousarray(a); s = np.empty_like(a); c = np.empty_like(a)
        self._chk(self._f("sincos")(NP2T[a.dtype], _p(a), _p(s), _p(c), ctypes.c_size_t(a.size)), "sincos")
        return s, c

    def binary(self, op, a, b):
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b, dtype=a.dtype); out = np.empty_like(a)
        self._chk(self._f("binary")(NP2T[a.dtype], op.encode(), _p(a), _p(b), _p(out), ctypes.c_size_t(a.size)), op)
        return out

    def ternary(self, op, a, b, c):
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b, dtype=a.dtype)
        c = np.ascontiguousarray(c, dtype=a.dtype); out = np.empty_like(a)
        self._chk(self._f("ternary")(NP2T[a.dtype], op.encode(), _p(a), _p(b), _p(c), _p(out),
                                     ctypes.c_size_t(a.size)), op)
        return out

    def compare(self, op, a, b):
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b, dtype=a.dtype)
        out = np.empty(a.size, np.uint8)
        self._chk(self._f("compare")(NP2T[a.dtype], op.encode(), _p(a), _p(b), _p(out), ctypes.c_size_t(a.size)), op)
        return out

    def select(self, m, t, f):
        m = np.ascontiguousarray(m, dtype=np.uint8); t = np.ascontiguousarray(t)
        f = np.ascontiguousarray(f, dtype=t.dtype); out = np.empty_like(t)
        self._chk(self._f("select")(NP2T[t.dtype], _p(m), _p(t), _p(f), _p(out), ctypes.c_size_t(t.size)), "select")
        return out

    def cast(self, a, dst_dtype):
        a = np.ascontiguousarray(a); out = np.empty(a.size, dst_dtype)
        self._chk(self._f("cast")(NP2T[a.dtype], NP2T[np.dtype(dst_dtype)], _p(a), _p(out),
                                  ctypes.c_size_t(a.size)), "cast")
        return out

    def gather(self, src, idx, mask):
        src = np.ascontiguousarray(src); idx = np.ascontiguousarray(idx)
        mask = np.ascontiguousarray(mask, dtype=np.uint8); out = np.empty(idx.size, src.dtype)
        self._chk(self._f("gather")(NP2T[src.dtype], NP2T[idx.dtype], _p(src), ctypes.c_size_t(src.size),
                                    _p(idx), _p(mask), _p(out), ctypes.c_size_t(idx.size)), "gather")
        return out

    def scatter(self, target, val, idx, mask, add=False):
        """returns the modified copy of ``target``"""
        target = np.array(target, copy=True); val = np.ascontiguousarray(val, dtype=target.dtype)
        idx = np.ascontiguousarray(idx); mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._chk(self._f("scatter")(NP2T[target.dtype], NP2T[idx.dtype], int(add), _p(target), _p(val),
                                     _p(idx), _p(mask), ctypes.c_size_t(idx.size)), "scatter")
        return target

    def reduce(self, op, a):
        a = np.ascontiguousarray(a); out = np.empty(1, a.dtype)
        self._chk(self._f("reduce")(NP2T[a.dtype], op.encode(), _p(a), _p(out), ctypes.c_size_t(a.size)), op)
        return out[0]

    def mask_reduce(self, op, m):
        m = np.ascontiguousarray(m, dtype=np.uint8); out = ctypes.c_uint64()
        self._chk(self._f("mask_reduce")(op.encode(), _p(m), ctypes.byref(out), ctypes.c_size_t(m.size)), op)
        return out.value

    def linspace(self, lo, hi, n):
        out = np.empty(n, np.float32)
        self._f("linspace_f32")(ctypes.c_float(lo), ctypes.c_float(hi), _p(out), ctypes.c_size_t(n))
        return out

    def psum(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32); out = np.empty_like(a)
        self._f("psum_f32")(_p(a), _p(out), ctypes.c_size_t(a.size))
        return out

    # ---- BASELINE.json configs -------------------------------------------------------------
    def cfg1(self, a, x, b):
        sec = ctypes.c_double()
        y = self._f("cfg1")(_p(a), _p(x), _p(b), ctypes.c_size_t(a.size), ctypes.byref(sec))
        return y, sec.value

    def cfg2(self, a, x, b):
        sec = ctypes.c_double()
        y = self._f("cfg2")(_p(a), _p(x), _p(b), ctypes.c_size_t(a.size), ctypes.byref(sec))
        return y, sec.value

    def cfg3a(self, a, x, b):
        ga = np.empty_like(a); gb = np.empty_like(a); sec = ctypes.c_double()
        y = self._f("cfg3a")(_p(a), _p(x), _p(b), ctypes.c_size_t(a.size), _p(ga), _p(gb), ctypes.byref(sec))
        return y, ga, gb, sec.value

    def cfg3b(self, A, B, x, idx):
        gA = np.empty_like(A); gB = np.empty_like(B); sec = ctypes.c_double()
        y = self._f("cfg3b")(_p(A), _p(B), ctypes.c_size_t(A.size), _p(x), _p(idx), ctypes.c_size_t(x.size),
                             _p(gA), _p(gB), ctypes.byref(sec))
        return y, gA, gB, sec.value


    SPELLINGS = {"fmadd": 0, "a*x+b": 1, "a*x-b": 2, "b-a*x": 3, "b+a*x": 4, "a*x": 5}

    def cfg3b_variant(self, A, B, x, idx, mask=None, func="sin", seed=1.0, spelling="fmadd"):
        """reference build only (oracle/ref_driver.cpp:ref_cfg3b_variant): the neighbours of cfg3b -- f = sin | cos | exp | log | sqrt,
        32- or 64-bit indices, optional mask, backward(seed * y)"""
        gA = np.empty_like(A); gB = np.empty_like(B); sec = ctypes.c_double()
        f = self._f("cfg3b_variant")
        f.restype = ctypes.c_float
        m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
        y = f(_p(A), _p(B), ctypes.c_size_t(A.size), _p(x), _p(idx), ctypes.c_int(int(idx.dtype.itemsize == 8)),
              _p(m) if m is not None else None, ctypes.c_size_t(x.size),
              ctypes.c_int({"sin": 0, "cos": 1, "exp": 2, "log": 3, "sqrt": 4, "rcp": 5, "rsqrt": 6}[func] + 16 * self.SPELLINGS[spelling]),
              ctypes.c_float(seed), _p(gA), _p(gB), ctypes.byref(sec))
        return y, gA, gB, sec.value

    def cfg5(self, tex, n, seed, first_lane=0, bounces=3, width=1024):
        """reference build only: the templated path tracer of examples/path_trace.h on the reference's arrays
        (oracle/ref_driver.cpp:ref_cfg5) -> loss, grad_tex, seconds"""
        tex = np.ascontiguousarray(tex, np.float32)
        g = np.empty_like(tex); sec = ctypes.c_double()
        f = self._f("cfg5")
        f.restype = ctypes.c_float
        y = f(_p(tex), ctypes.c_size_t(tex.size), ctypes.c_size_t(n), ctypes.c_uint64(seed), ctypes.c_uint64(first_lane),
              ctypes.c_int(bounces), ctypes.c_uint32(width), _p(g), ctypes.byref(sec))
        return y, g, sec.value

    def pcg32(self, initstate, initseq, steps, mask, bound, delta):
        """the PCG32 draw script of oracle/ref_driver.cpp:ref_pcg32 -> dict of outputs"""
        initseq = np.ascontiguousarray(initseq, np.uint64); mask = np.ascontiguousarray(mask, np.uint8); n = initseq.size
        o = {"u32": np.empty((steps, n), np.uint32), "f32": np.empty(n, np.float32), "u64": np.empty(n, np.uint64),
             "f64": np.empty(n, np.float64), "bounded": np.empty(n, np.uint32), "after": np.empty(n, np.uint32),
             "state": np.empty(n, np.uint64)}
        self._chk(self._f("pcg32")(ctypes.c_uint64(initstate), _p(initseq), ctypes.c_size_t(n), ctypes.c_int(steps), _p(mask),
                                   _p(o["u32"]), _p(o["f32"]), _p(o["u64"]), _p(o["f64"]), ctypes.c_uint32(bound),
                                   _p(o["bounded"]), ctypes.c_int64(delta), _p(o["after"]), _p(o["state"])), "pcg32")
        return o

    def matrix(self, size, a, b, v):
        """Matrix<FloatX, size> script of oracle/ref_driver.cpp:ref_matrix; a, b: (size*size, n) row-major entries, v: (size, n)"""
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32); v = np.ascontiguousarray(v, np.float32)
        n = a.shape[1]
        o = {"mm": np.empty_like(a), "mv": np.empty_like(v), "trace": np.empty(n, np.float32), "frob": np.empty(n, np.float32),
             "det": np.zeros(n, np.float32), "inv": np.zeros_like(a)}
        self._chk(self._f("matrix")(ctypes.c_int(size), _p(a), _p(b), _p(v), ctypes.c_size_t(n), _p(o["mm"]), _p(o["mv"]),
                                    _p(o["trace"]), _p(o["frob"]), _p(o["det"]), _p(o["inv"])), "matrix")
        return o

    def quaternion(self, a, b, t):
        """Quaternion<FloatX> script of oracle/ref_driver.cpp:ref_quaternion; a, b: (4, n), t: (n) -> (8, 4, n), (9, n)"""
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32); t = np.ascontiguousarray(t, np.float32)
        n = a.shape[1]
        out = np.empty((8, 4, n), np.float32); mat = np.empty((9, n), np.float32)
        self._chk(self._f("quaternion")(_p(a), _p(b), _p(t), ctypes.c_size_t(n), _p(out), _p(mat)), "quaternion")
        return out, mat

    def complex_more(self, a):
        """sinh, cosh, tanh, asin, acos, atan, asinh, acosh, atanh of Complex<FloatX> (ref_complex_more); a: (2, n) -> (9, 2, n)"""
        a = np.ascontiguousarray(a, np.float32)
        out = np.empty((9, 2, a.shape[1]), np.float32)
        self._chk(self._f("complex_more")(_p(a), ctypes.c_size_t(a.shape[1]), _p(out)), "complex_more")
        return out

    def sh(self, d, order):
        """sh_eval of the reference (ref_sh); d: (3, n) unit vectors -> ((order + 1)^2, n)"""
        d = np.ascontiguousarray(d, np.float32)
        out = np.empty(((order + 1) ** 2, d.shape[1]), np.float32)
        self._chk(self._f("sh")(_p(d), ctypes.c_size_t(d.shape[1]), ctypes.c_size_t(order), _p(out)), "sh")
        return out

    def transform(self, v, p):
        """include/enoki/transform.h of the reference (ref_transform); v: (3, n), p: (6, n) -> (8, 16, n) row-major entries"""
        v = np.ascontiguousarray(v, np.float32); p = np.ascontiguousarray(p, np.float32)
        out = np.empty((8, 16, v.shape[1]), np.float32)
        self._chk(self._f("transform")(_p(v), _p(p), ctypes.c_size_t(v.shape[1]), _p(out)), "transform")
        return out

    def ellint(self, phi, k, nu):
        """elliptic integrals of the reference (oracle/ref_driver.cpp:ref_ellint_*); phi, k, nu: (n) -> (10, n), rows
        comp_1, comp_2, comp_3, ellint_1, ellint_2, ellint_3, rf, rd, rc, rj"""
        dt = np.asarray(phi).dtype
        phi, k, nu = (np.ascontiguousarray(v, dt) for v in (phi, k, nu))
        out = np.empty((10, phi.shape[0]), dt)
        name = "ellint_f32" if dt == np.float32 else "ellint_f64"
        self._chk(self._f(name)(_p(phi), _p(k), _p(nu), ctypes.c_size_t(phi.shape[0]), _p(out)), name)
        return out

    def complex(self, a, b):
        """Complex<FloatX> script of oracle/ref_driver.cpp:ref_complex; a, b: (2, n) -> (10, 2, n)"""
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        out = np.empty((10, 2, a.shape[1]), np.float32)
        self._chk(self._f("complex")(_p(a), _p(b), ctypes.c_size_t(a.shape[1]), _p(out)), "complex")
        return out


def _build(target):
    subprocess.run(["make", "-C", ORACLE_DIR, target], check=True, stdout=subprocess.DEVNULL)


_cache = {}


def port(build=True):
    """our C restatement; always buildable (gcc only)"""
    if "port" not in _cache:
        path = os.path.join(ORACLE_DIR, "libenoki_oracle.so")
        if build and (not os.path.exists(path) or
                      os.path.getmtime(path) < os.path.getmtime(os.path.join(ORACLE_DIR, "enoki_oracle.c"))):
            _build("port")
        _cache["port"] = Checker(ctypes.CDLL(path), "orc_", "port")
    return _cache["port"]


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libenoki_ref.so")) or os.path.isdir("/root/reference")


def ref():
    """the real reference build; exists where /root/reference was available at build time"""
    if "ref" not in _cache:
        path = os.path.join(ORACLE_DIR, "_ref", "libenoki_ref.so")
        if not os.path.exists(path):
            if not os.path.isdir("/root/reference"):
                raise FileNotFoundError(path)
            _build("ref")
        _cache["ref"] = Checker(ctypes.CDLL(path), "ref_", "reference")
    return _cache["ref"]


def ref_scalar():
    """the reference's SCALAR path (its `none` row, oracle/Makefile refscalar): generic packets, rcp() = 1 / a, rsqrt() =
    1 / sqrt(a), fmadd() = a * b + c in two roundings -- what rcp / rsqrt / division of the device are pinned to bit for bit,
    and the second anchor (beside the AVX2 row) of the functions built on them"""
    if "refscalar" not in _cache:
        path = os.path.join(ORACLE_DIR, "_ref", "libenoki_refscalar.so")
        if not os.path.exists(path):
            if not os.path.isdir("/root/reference"):
                raise FileNotFoundError(path)
            _build("refscalar")
        _cache["refscalar"] = Checker(ctypes.CDLL(path), "ref_", "refscalar")
    return _cache["refscalar"]
