"""ctypes binding of the C ABI declared in include/enoki_hip.h (libenoki-hip.so).

This is the thinnest possible host-side view of the library: device buffers are raw pointers
obtained from ``ek_hip_malloc`` and every function maps 1:1 onto an exported symbol.  It exists
for the parity tests (which must call *through the C ABI*), for tools/ and for bench.py's
low-level probes; user code goes through the HIPArray / DiffArray classes instead.

There is deliberately no CPU fallback: if the shared library is missing the import fails.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libenoki-hip.so")

# ek_type
BOOL, I32, U32, I64, U64, F32, F64 = range(7)
NP2EK = {np.dtype(np.uint8): BOOL, np.dtype(np.bool_): BOOL, np.dtype(np.int32): I32, np.dtype(np.uint32): U32,
         np.dtype(np.int64): I64, np.dtype(np.uint64): U64, np.dtype(np.float32): F32, np.dtype(np.float64): F64}
EK2NP = {BOOL: np.uint8, I32: np.int32, U32: np.uint32, I64: np.int64, U64: np.uint64, F32: np.float32, F64: np.float64}

UNARY = {n: i for i, n in enumerate(
    ["neg", "abs", "not", "sqrt", "rcp", "rsqrt", "floor", "ceil", "round", "trunc", "sin", "cos", "exp", "log",
     "popcnt", "lzcnt", "tzcnt", "sign", "copy", "tan", "cot", "asin", "acos", "atan", "sinh", "cosh", "tanh", "asinh",
     "acosh", "atanh", "cbrt", "erf", "erfc", "erfinv", "i0e", "dawson", "erfi", "lgamma", "tgamma", "rcp_sqr", "rsqrt_sqr",
     "rsqrt_cube", "sec_sqr", "sech_sqr", "rcp_1p_sqr"])}
BINARY = {n: i for i, n in enumerate(
    ["add", "sub", "mul", "div", "mod", "min", "max", "mulhi", "and", "or", "xor", "sl", "sr", "safe_mul",
     "atan2", "pow", "fmod", "ldexp"])}
TERNARY = {n: i for i, n in enumerate(["fmadd", "fmsub", "fnmadd", "fnmsub", "safe_fmadd", "muladd", "mulsub", "nmuladd"])}
COMPARE = {n: i for i, n in enumerate(["eq", "neq", "lt", "le", "gt", "ge"])}
REDUCE = {n: i for i, n in enumerate(["hsum", "hprod", "hmin", "hmax"])}
MASK_REDUCE = {n: i for i, n in enumerate(["all", "any", "count"])}

EXPORTS = [
    "ek_hip_init", "ek_hip_device", "ek_hip_device_count", "ek_hip_stream", "ek_hip_set_stream", "ek_hip_sync",
    "ek_hip_last_error", "ek_hip_malloc", "ek_hip_free", "ek_hip_malloc_trim", "ek_hip_host_malloc",
    "ek_hip_host_free", "ek_hip_mem_get_info", "ek_hip_memcpy_to_device", "ek_hip_memcpy_to_host",
    "ek_hip_memcpy_device", "ek_hip_memset", "ek_hip_whos", "ek_hip_set_log_level", "ek_hip_log_level",
    "ek_hip_launch_count", "ek_hip_set_tuning", "ek_hip_profile_begin", "ek_hip_profile_end", "ek_hip_unary", "ek_hip_binary", "ek_hip_ternary", "ek_hip_sincos", "ek_hip_sincosh", "ek_hip_pcg32_next", "ek_hip_gather_multi", "ek_hip_gather_multi_sized", "ek_hip_gather_multi_plan", "ek_hip_scatter_add_multi", "ek_hip_concat", "ek_hip_concat_rows",
    "ek_hip_compare", "ek_hip_select", "ek_hip_cast", "ek_hip_fill", "ek_hip_arange", "ek_hip_linspace",
    "ek_hip_reverse", "ek_hip_gather", "ek_hip_scatter", "ek_hip_scatter_add", "ek_hip_reduce",
    "ek_hip_hsum_safe_mul", "ek_hip_mask_reduce", "ek_hip_psum", "ek_hip_map_gathered", "ek_hip_note_launch", "ek_hip_graph_begin", "ek_hip_graph_end", "ek_hip_graph_launch",
    "ek_hip_graph_launch_count", "ek_hip_graph_destroy", "ek_hip_sort_pairs", "ek_hip_reduce_map", "ek_hip_reduce_chain", "ek_hip_map_chain", "ek_hip_map_chain_product", "ek_hip_partition_class_state", "ek_hip_scatter_add_multi_map", "ek_hip_binding_slot",
    "ek_hip_dist_unique_id", "ek_hip_dist_init", "ek_hip_dist_world", "ek_hip_dist_shard_range", "ek_hip_dist_all_reduce",
    "ek_hip_dist_reduce_scatter", "ek_hip_dist_all_gather", "ek_hip_dist_finalize", "ek_hip_dist_rccl_path",
    "ek_hip_bucketed_applicable", "ek_hip_bucketed_pair_create", "ek_hip_bucketed_pair_create_hinted", "ek_hip_bucketed_pair_create_masked", "ek_hip_bucketed_reduce", "ek_hip_bucketed_scatter_add", "ek_hip_bucketed_scatter_add_scaled", "ek_hip_bucketed_early_pair",
    "ek_hip_bucketed_destroy", "ek_hip_index_partition_create", "ek_hip_index_partition_get", "ek_hip_index_partition_destroy", "ek_hip_gather_address",
]


class Operand(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("imm", ctypes.c_uint64), ("size", ctypes.c_size_t)]


class Gathered(ctypes.Structure):
    """ek_gathered: an operand read through an index array (consumed in place by ek_hip_map_gathered)"""
    _fields_ = [("table", ctypes.c_void_p), ("table_size", ctypes.c_size_t), ("index", Operand), ("index_type", ctypes.c_int),
                ("mask", Operand)]


class EnokiHipError(RuntimeError):
    pass


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing -- run `python -m enoki_amd._build` (there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    lib.ek_hip_last_error.restype = ctypes.c_char_p
    lib.ek_hip_stream.restype = ctypes.c_void_p
    lib.ek_hip_whos.restype = ctypes.c_void_p
    lib.ek_hip_profile_end.restype = ctypes.c_void_p
    lib.ek_hip_launch_count.restype = ctypes.c_uint64
    lib.ek_hip_log_level.restype = ctypes.c_uint32
    return lib


lib = load()

_probe = None


def probe_lib():
    """libenoki-hip-probe.so: the measurement kernels of csrc/probe.hip (tools/probe_*.py only; the product never loads it)"""
    global _probe
    if _probe is None:
        path = os.path.join(HERE, "libenoki-hip-probe.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing -- run `python -m enoki_amd._build`")
        _probe = ctypes.CDLL(path)
        _probe.ek_hip_probe_last_error.restype = ctypes.c_char_p
    return _probe


def check(rc):
    if rc != 0:
        raise EnokiHipError(f"[{rc}] " + lib.ek_hip_last_error().decode())


def init(device=-1):
    check(lib.ek_hip_init(device))


def sync():
    check(lib.ek_hip_sync())


def stream():
    return lib.ek_hip_stream()


def set_tuning(key, value):
    check(lib.ek_hip_set_tuning(key.encode(), int(value)))


def whos():
    p = lib.ek_hip_whos()
    s = ctypes.string_at(p).decode()
    ctypes.CDLL(None).free(ctypes.c_void_p(p))
    return s


def profile_begin():
    check(lib.ek_hip_profile_begin())


def profile_end():
    """list of {"kernel", "launches", "total_ms", "bytes", "elements"} since profile_begin()"""
    import json
    p = lib.ek_hip_profile_end()
    s = ctypes.string_at(p).decode()
    ctypes.CDLL(None).free(ctypes.c_void_p(p))
    return json.loads(s)


class Buf:
    """A device array: raw pointer + numpy dtype + element count (owned unless ``own=False``)."""

    def __init__(self, dtype, n, own=True, ptr=None):
        self.dtype = np.dtype(dtype)
        if self.dtype == np.bool_:
            self.dtype = np.dtype(np.uint8)
        self.n = int(n)
        self.own = own
        if ptr is None:
            p = ctypes.c_void_p()
            check(lib.ek_hip_malloc(ctypes.c_size_t(max(self.n, 1) * self.dtype.itemsize), ctypes.byref(p)))
            self.ptr = p.value
        else:
            self.ptr = ptr

    @property
    def ek(self):
        return NP2EK[self.dtype]

    @staticmethod
    def from_numpy(a):
        a = np.ascontiguousarray(a)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        b = Buf(a.dtype, a.size)
        if a.size:
            check(lib.ek_hip_memcpy_to_device(ctypes.c_void_p(b.ptr), a.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.c_size_t(a.nbytes)))
        return b

    def numpy(self):
        out = np.empty(self.n, self.dtype)
        if self.n:
            check(lib.ek_hip_memcpy_to_host(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(self.ptr),
                                            ctypes.c_size_t(out.nbytes)))
        return out

    def view(self, offset, n):
        """non-owning window (element offset), e.g. to exercise misaligned pointers"""
        return Buf(self.dtype, n, own=False, ptr=self.ptr + offset * self.dtype.itemsize)

    def free(self):
        if self.own and self.ptr:
            lib.ek_hip_free(ctypes.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _imm_bits(value, dtype):
    return int(np.array([value], dtype=dtype).view({1: np.uint8, 4: np.uint32, 8: np.uint64}[np.dtype(dtype).itemsize])[0])


def operand(x, dtype=None):
    """Buf -> array operand; python/numpy scalar -> immediate operand of ``dtype``"""
    if isinstance(x, Buf):
        return Operand(x.ptr, 0, x.n)
    return Operand(None, _imm_bits(x, dtype), 1)


def _n(*xs):
    n = 1
    for x in xs:
        if isinstance(x, Buf) and x.n != 1:
            n = x.n
    return n


def _dtype(*xs):
    for x in xs:
        if isinstance(x, Buf):
            return x.dtype
    raise TypeError("at least one operand must be a device buffer")


def unary(op, a, n=None):
    dt = _dtype(a); n = _n(a) if n is None else n
    out = Buf(dt, n); oa = operand(a, dt)
    check(lib.ek_hip_unary(UNARY[op], NP2EK[dt], ctypes.c_void_p(out.ptr), ctypes.byref(oa), ctypes.c_size_t(n)))
    return out


def binary(op, a, b, n=None):
    dt = _dtype(a, b); n = _n(a, b) if n is None else n
    out = Buf(dt, n); oa, ob = operand(a, dt), operand(b, dt)
    check(lib.ek_hip_binary(BINARY[op], NP2EK[dt], ctypes.c_void_p(out.ptr), ctypes.byref(oa), ctypes.byref(ob),
                            ctypes.c_size_t(n)))
    return out


def ternary(op, a, b, c, n=None):
    dt = _dtype(a, b, c); n = _n(a, b, c) if n is None else n
    out = Buf(dt, n); oa, ob, oc = operand(a, dt), operand(b, dt), operand(c, dt)
    check(lib.ek_hip_ternary(TERNARY[op], NP2EK[dt], ctypes.c_void_p(out.ptr), ctypes.byref(oa), ctypes.byref(ob),
                             ctypes.byref(oc), ctypes.c_size_t(n)))
    return out


def sincosh(a):
    dt = _dtype(a); n = _n(a)
    s = Buf(dt, n); c = Buf(dt, n); oa = operand(a, dt)
    check(lib.ek_hip_sincosh(NP2EK[dt], ctypes.c_void_p(s.ptr), ctypes.c_void_p(c.ptr), ctypes.byref(oa),
                             ctypes.c_size_t(n)))
    return s, c


def sincos(a):
    dt = _dtype(a); n = _n(a)
    s, c = Buf(dt, n), Buf(dt, n); oa = operand(a, dt)
    check(lib.ek_hip_sincos(NP2EK[dt], ctypes.c_void_p(s.ptr), ctypes.c_void_p(c.ptr), ctypes.byref(oa),
                            ctypes.c_size_t(n)))
    return s, c


def compare(op, a, b, n=None):
    dt = _dtype(a, b); n = _n(a, b) if n is None else n
    out = Buf(np.uint8, n); oa, ob = operand(a, dt), operand(b, dt)
    check(lib.ek_hip_compare(COMPARE[op], NP2EK[dt], ctypes.c_void_p(out.ptr), ctypes.byref(oa), ctypes.byref(ob),
                             ctypes.c_size_t(n)))
    return out


def select(m, t, f, n=None):
    dt = _dtype(t, f); n = _n(m, t, f) if n is None else n
    out = Buf(dt, n); om, ot, of = operand(m, np.uint8), operand(t, dt), operand(f, dt)
    check(lib.ek_hip_select(NP2EK[dt], ctypes.c_void_p(out.ptr), ctypes.byref(om), ctypes.byref(ot), ctypes.byref(of),
                            ctypes.c_size_t(n)))
    return out


def cast(a, dst_dtype):
    n = a.n
    out = Buf(dst_dtype, n); oa = operand(a)
    check(lib.ek_hip_cast(a.ek, NP2EK[np.dtype(dst_dtype)], ctypes.c_void_p(out.ptr), ctypes.byref(oa),
                          ctypes.c_size_t(n)))
    return out


def fill(dtype, value, n):
    out = Buf(dtype, n)
    check(lib.ek_hip_fill(out.ek, ctypes.c_void_p(out.ptr), ctypes.c_uint64(_imm_bits(value, dtype)), ctypes.c_size_t(n)))
    return out


def arange(dtype, n, start=0, step=1):
    out = Buf(dtype, n)
    check(lib.ek_hip_arange(out.ek, ctypes.c_void_p(out.ptr), ctypes.c_int64(start), ctypes.c_int64(step),
                            ctypes.c_size_t(n)))
    return out


def linspace(dtype, lo, hi, n):
    out = Buf(dtype, n)
    check(lib.ek_hip_linspace(out.ek, ctypes.c_void_p(out.ptr), ctypes.c_double(lo), ctypes.c_double(hi),
                              ctypes.c_size_t(n)))
    return out


def reverse(a):
    out = Buf(a.dtype, a.n)
    check(lib.ek_hip_reverse(a.ek, ctypes.c_void_p(out.ptr), ctypes.c_void_p(a.ptr), ctypes.c_size_t(a.n)))
    return out


def gather(src, index, mask=True, n=None):
    n = _n(index, mask) if n is None else n
    out = Buf(src.dtype, n)
    oi = operand(index); om = operand(mask, np.uint8)
    check(lib.ek_hip_gather(src.ek, index.ek, ctypes.c_void_p(out.ptr), ctypes.c_void_p(src.ptr), ctypes.byref(oi),
                            ctypes.byref(om), ctypes.c_size_t(n)))
    return out


def gather_multi(tables, index, mask=True, n=None, sized=True):
    """outs[c][i] = mask[i] ? tables[c][index[i]] : 0 for 2..4 tables of one length sharing ONE index / mask array
    (ek_hip_gather_multi_sized: may stage {x, y, ..} records; ``sized=False``: plain ek_hip_gather_multi)"""
    n = _n(index, mask) if n is None else n
    count = len(tables)
    outs = [Buf(tables[0].dtype, n) for _ in range(count)]
    oi = operand(index); om = operand(mask, np.uint8)
    po = (ctypes.c_void_p * count)(*[o.ptr for o in outs])
    pb = (ctypes.c_void_p * count)(*[t.ptr for t in tables])
    if sized:
        check(lib.ek_hip_gather_multi_sized(tables[0].ek, index.ek, count, po, pb, ctypes.c_size_t(tables[0].n), ctypes.byref(oi),
                                            ctypes.byref(om), ctypes.c_size_t(n)))
    else:
        check(lib.ek_hip_gather_multi(tables[0].ek, index.ek, count, po, pb, ctypes.byref(oi), ctypes.byref(om), ctypes.c_size_t(n)))
    return outs


class G:
    """gathered operand for map_gathered(): table[index] where mask"""

    def __init__(self, table, index, mask=True):
        self.table, self.index, self.mask = table, index, mask


def map_gathered(op, *xs, n=None):
    """out = op(x0, x1 (, x2)); operands that are `G` instances are gathered in place (ek_hip_map_gathered)"""
    arity = len(xs)
    tables = [x.table if isinstance(x, G) else x for x in xs]
    dt = _dtype(*tables)
    if n is None:
        n = max([x.index.n for x in xs if isinstance(x, G)] + [_n(*[x for x in xs if not isinstance(x, G)])])
    out = Buf(dt, n)
    ops, gs = [], []
    for x in xs:
        if isinstance(x, G):
            gs.append(Gathered(x.table.ptr, x.table.n, operand(x.index), x.index.ek, operand(x.mask, np.uint8)))
            ops.append(None)
        else:
            gs.append(None)
            ops.append(operand(x, dt))
    OpPtr, GPtr = ctypes.POINTER(Operand), ctypes.POINTER(Gathered)
    po = (OpPtr * arity)(*[ctypes.pointer(o) if o is not None else OpPtr() for o in ops])
    pg = (GPtr * arity)(*[ctypes.pointer(g) if g is not None else GPtr() for g in gs])
    code = (BINARY if arity == 2 else TERNARY)[op]
    check(lib.ek_hip_map_gathered(arity, code, NP2EK[dt], ctypes.c_void_p(out.ptr), po, pg, ctypes.c_size_t(n)))
    return out


def scatter(target, value, index, mask=True, n=None):
    n = _n(value, index, mask) if n is None else n
    ov, oi, om = operand(value, target.dtype), operand(index), operand(mask, np.uint8)
    check(lib.ek_hip_scatter(target.ek, index.ek, ctypes.c_void_p(target.ptr), ctypes.byref(ov), ctypes.byref(oi),
                             ctypes.byref(om), ctypes.c_size_t(n)))


def scatter_add(target, value, index, mask=True, n=None, mode=0):
    n = _n(value, index, mask) if n is None else n
    ov, oi, om = operand(value, target.dtype), operand(index), operand(mask, np.uint8)
    check(lib.ek_hip_scatter_add(target.ek, index.ek, ctypes.c_void_p(target.ptr), ctypes.c_size_t(target.n),
                                 ctypes.byref(ov), ctypes.byref(oi), ctypes.byref(om), ctypes.c_size_t(n), mode))


def scatter_add_multi(targets, values, index, mask=True, weights=None, n=None, mode=0):
    """targets[c][index[i]] += (weights[c] * values[c])[i] for all c: ONE pass over the indices (ek_hip_scatter_add_multi)"""
    count = len(targets)
    dt = targets[0].dtype
    n = _n(index, mask, *values) if n is None else n
    ovs = [operand(v, dt) for v in values]
    ows = [None if (weights is None or w is None) else operand(w, dt) for w in (weights or [None] * count)]
    OpPtr = ctypes.POINTER(Operand)
    bases = (ctypes.c_void_p * count)(*[t.ptr for t in targets])
    vals = (OpPtr * count)(*[ctypes.pointer(o) for o in ovs])
    wts = (OpPtr * count)(*[ctypes.pointer(o) if o is not None else OpPtr() for o in ows])
    oi, om = operand(index), operand(mask, np.uint8)
    check(lib.ek_hip_scatter_add_multi(targets[0].ek, index.ek, count, bases, ctypes.c_size_t(targets[0].n), vals,
                                       wts if weights is not None else None, ctypes.byref(oi), ctypes.byref(om),
                                       ctypes.c_size_t(n), mode))


class Chain(ctypes.Structure):
    """ek_chain: base(src[0 .. arity)) under n_maps unary ops, evaluated in one pass (ek_hip_reduce_chain / ek_hip_map_chain)"""
    _fields_ = [("arity", ctypes.c_int), ("base_op", ctypes.c_int), ("src", Operand * 3), ("n_maps", ctypes.c_int),
                ("map_ops", ctypes.c_int * 3)]


def _chain(base, srcs, maps):
    dt = _dtype(*srcs); n = _n(*srcs)
    ch = Chain()
    ch.arity = len(srcs)
    ch.base_op = 0 if len(srcs) == 1 else (BINARY if len(srcs) == 2 else TERNARY)[base]
    for k, x in enumerate(srcs):
        ch.src[k] = operand(x, dt)
    ch.n_maps = len(maps)
    for k, m in enumerate(maps):
        ch.map_ops[k] = UNARY[m]
    return ch, dt, n


def reduce_chain(op, base, srcs, maps):
    """op over maps[-1](.. maps[0](base(*srcs))) in ONE pass over the operands (ek_hip_reduce_chain); base: None for one source,
    "add" | "sub" | "mul" for two, the fma family / "muladd" | "mulsub" | "nmuladd" for three"""
    ch, dt, n = _chain(base, srcs, maps)
    out = Buf(dt, 1)
    check(lib.ek_hip_reduce_chain(REDUCE[op], NP2EK[dt], ctypes.c_void_p(out.ptr), ctypes.byref(ch), ctypes.c_size_t(n)))
    return out


def map_chain(base, srcs, maps):
    """the same chain written out (ek_hip_map_chain)"""
    ch, dt, n = _chain(base, srcs, maps)
    out = Buf(dt, n)
    check(lib.ek_hip_map_chain(NP2EK[dt], ctypes.c_void_p(out.ptr), ctypes.byref(ch), ctypes.c_size_t(n)))
    return out


def map_chain_product(base, srcs, maps, scale=None, w=None, op2="mul", first=True):
    """(chain * scale, op2(w, chain * scale)) in one pass (ek_hip_map_chain_product); first=False: only the product is written"""
    ch, dt, n = _chain(base, srcs, maps)
    out = Buf(dt, n) if first else None
    out2 = Buf(dt, n) if w is not None else None
    sc = operand(dt.type(scale), dt) if scale is not None else None
    wo = operand(w, dt) if w is not None else None
    check(lib.ek_hip_map_chain_product(NP2EK[dt], ctypes.c_void_p(out.ptr if out else None), ctypes.c_void_p(out2.ptr if out2 else None),
                                       ctypes.byref(ch), ctypes.byref(sc) if sc is not None else None, BINARY[op2],
                                       ctypes.byref(wo) if wo is not None else None, ctypes.c_size_t(n)))
    return out, out2


def reduce_map(op, map_op, a):
    """op(map_op(a)) in one pass: the unary op is applied on load (ek_hip_reduce_map)"""
    out = Buf(a.dtype, 1)
    check(lib.ek_hip_reduce_map(REDUCE[op], UNARY[map_op], a.ek, ctypes.c_void_p(out.ptr), ctypes.c_void_p(a.ptr),
                                ctypes.c_size_t(a.n)))
    return out


def scatter_add_multi_map(targets, values, ops, index, mask=True, weights=None, n=None, mode=0):
    """scatter_add_multi with unary ops[c] (name or None) applied to values[c] on load (ek_hip_scatter_add_multi_map)"""
    count = len(targets)
    dt = targets[0].dtype
    n = _n(index, mask, *values) if n is None else n
    ovs = [operand(v, dt) for v in values]
    ows = [None if (weights is None or w is None) else operand(w, dt) for w in (weights or [None] * count)]
    OpPtr = ctypes.POINTER(Operand)
    bases = (ctypes.c_void_p * count)(*[t.ptr for t in targets])
    vals = (OpPtr * count)(*[ctypes.pointer(o) for o in ovs])
    wts = (OpPtr * count)(*[ctypes.pointer(o) if o is not None else OpPtr() for o in ows])
    codes = (ctypes.c_int * count)(*[UNARY["copy"] if o is None else UNARY[o] for o in ops])
    oi, om = operand(index), operand(mask, np.uint8)
    check(lib.ek_hip_scatter_add_multi_map(targets[0].ek, index.ek, count, bases, ctypes.c_size_t(targets[0].n), vals, codes,
                                           wts if weights is not None else None, ctypes.byref(oi), ctypes.byref(om),
                                           ctypes.c_size_t(n), mode))


def reduce(op, a):
    out = Buf(a.dtype, 1)
    check(lib.ek_hip_reduce(REDUCE[op], a.ek, ctypes.c_void_p(out.ptr), ctypes.c_void_p(a.ptr if a.n else None),
                            ctypes.c_size_t(a.n)))
    return out


def hsum_safe_mul(w, g, n=None):
    dt = _dtype(w, g); n = _n(w, g) if n is None else n
    out = Buf(dt, 1); ow, og = operand(w, dt), operand(g, dt)
    check(lib.ek_hip_hsum_safe_mul(NP2EK[dt], ctypes.c_void_p(out.ptr), ctypes.byref(ow), ctypes.byref(og),
                                   ctypes.c_size_t(n)))
    return out


def mask_reduce(op, m):
    res = ctypes.c_uint64()
    check(lib.ek_hip_mask_reduce(MASK_REDUCE[op], ctypes.c_void_p(m.ptr if m.n else None), ctypes.c_size_t(m.n),
                                 ctypes.byref(res)))
    return res.value


def psum(a):
    out = Buf(a.dtype, a.n)
    check(lib.ek_hip_psum(a.ek, ctypes.c_void_p(out.ptr), ctypes.c_void_p(a.ptr), ctypes.c_size_t(a.n)))
    return out


class Bucketed:
    """u = op(A[index], x, C[index]) kept in bucket order (ek_hip_bucketed_*): reductions over map(u) and the adjoint
    scatter_add of the two gathers without a lookup that leaves the CU.  Keeps A, C alive; x and index may be dropped."""

    HINT_ADJOINT = 1
    HINT_BOUNDED = 2

    def __init__(self, op, A, x, C, index, hints=0, mask=None):
        self.A, self.C, self.dtype, self.K = A, C, A.dtype, A.n
        h = ctypes.c_void_p()
        check(lib.ek_hip_bucketed_pair_create_masked(A.ek, index.ek, TERNARY[op], ctypes.c_void_p(A.ptr), ctypes.c_void_p(C.ptr),
                                                     ctypes.c_size_t(A.n), ctypes.c_void_p(x.ptr), ctypes.c_void_p(index.ptr),
                                                     ctypes.c_void_p(mask.ptr if mask is not None else None),
                                                     ctypes.c_size_t(index.n), ctypes.c_uint(hints), ctypes.byref(h)))
        self.handle = h

    @staticmethod
    def applicable(dtype, index_dtype, table_size, n):
        return bool(lib.ek_hip_bucketed_applicable(NP2EK[np.dtype(dtype)], NP2EK[np.dtype(index_dtype)], ctypes.c_size_t(table_size),
                                                   ctypes.c_size_t(n)))

    def reduce(self, op, map_op=None, keep=True, keep_op=None):
        out = Buf(self.dtype, 1)
        check(lib.ek_hip_bucketed_reduce(self.handle, REDUCE[op], UNARY[map_op or "copy"], ctypes.c_void_p(out.ptr), int(keep),
                                         UNARY[keep_op or "copy"]))
        return out

    def scatter_add(self, targets, streams, fresh=None, scales=None):
        """streams[c] = (map_op name | None for a constant, constant value, weighted by x?); fresh[c]: table c holds no data
        yet, its sums are written instead of added; scales[c]: host scalar factor on stream c"""
        count = len(targets)
        bases = (ctypes.c_void_p * count)(*[t.ptr for t in targets])
        from_u = (ctypes.c_int * count)(*[0 if s[0] is None else 1 for s in streams])
        ops = (ctypes.c_int * count)(*[UNARY[s[0] or "copy"] for s in streams])
        imm = (ctypes.c_uint64 * count)(*[_imm_bits(s[1], self.dtype) for s in streams])
        wt = (ctypes.c_int * count)(*[int(bool(s[2])) for s in streams])
        fr = (ctypes.c_int * count)(*[int(bool(f)) for f in fresh]) if fresh is not None else None
        if scales is not None:
            sc = (ctypes.c_uint64 * count)(*[_imm_bits(v, self.dtype) for v in scales])
            check(lib.ek_hip_bucketed_scatter_add_scaled(self.handle, count, bases, from_u, ops, imm, wt, fr, sc))
        else:
            check(lib.ek_hip_bucketed_scatter_add(self.handle, count, bases, from_u, ops, imm, wt, fr))

    def destroy(self):
        if self.handle:
            lib.ek_hip_bucketed_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
