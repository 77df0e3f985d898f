"""Register-machine programs over DiffArray values and runners for the three tape implementations
(reference build, product tape over the CPU oracle array, product tape over HIPArray).  See
tests/cpp/tape_program.h for the encoding."""
import ctypes
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

OPS = ["add", "sub", "mul", "div", "fmadd", "neg", "abs", "sqrt", "rcp", "rsqrt", "sin", "cos", "exp", "log", "hsum",
       "hprod", "min", "max", "gather", "scatter_add", "scatter", "select_gt0", "mulc", "addc", "tanh", "tan",
       "atan2", "fmsub", "fnmadd", "fnmsub", "sinh", "cosh", "asin", "acos", "atan", "psum", "reverse", "asinh", "acosh",
       "atanh", "cbrt", "pow", "cot"]
OPCODE = {n: i for i, n in enumerate(OPS)}


def _f2i(x):
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


class Program:
    """inputs: list of (np.float32 array, is_leaf); index_inputs: list of np.uint32 arrays;
    ops: list of tuples (name, a, b, c) -- register numbers; for mulc/addc b is a float constant;
    for gather b is an index-input number; for scatter(_add) c is an index-input number."""

    def __init__(self, inputs, ops, index_inputs=(), mode="backward", fwd_leaf=0, simplify=False):
        self.inputs = [(np.ascontiguousarray(a, np.float32), bool(l)) for a, l in inputs]
        self.index_inputs = [np.ascontiguousarray(i, np.uint32) for i in index_inputs]
        self.ops, self.mode, self.fwd_leaf, self.simplify = list(ops), mode, fwd_leaf, simplify

    def encode(self):
        prog = []
        for op in self.ops:
            name, args = op[0], list(op[1:]) + [0] * (4 - len(op))
            if name in ("mulc", "addc"):
                args[1] = _f2i(args[1])
            prog += [OPCODE[name]] + [int(a) for a in args[:3]]
        return np.array(prog, np.int32)

    def out_size_bound(self):
        return max([a.size for a, _ in self.inputs] + [i.size for i in self.index_inputs] + [1])


def run(fn, program):
    """fn: the C entry point (ref_tape_program / host_tape_program / hip_tape_program).
    Returns (value array, [grad per input or None])"""
    p = program
    prog = p.encode()
    n_in, n_idx = len(p.inputs), len(p.index_inputs)
    in_ptrs = (ctypes.c_void_p * max(n_in, 1))(*[a.ctypes.data for a, _ in p.inputs])
    sizes = np.array([a.size for a, _ in p.inputs], np.uint64)
    leaf = np.array([1 if l else 0 for _, l in p.inputs], np.uint8)
    idx_ptrs = (ctypes.c_void_p * max(n_idx, 1))(*[i.ctypes.data for i in p.index_inputs])
    idx_sizes = np.array([i.size for i in p.index_inputs] + [0], np.uint64)
    out = np.zeros(p.out_size_bound(), np.float32)
    out_size = ctypes.c_uint64()
    if p.mode == "backward":
        grads = [np.zeros(a.size, np.float32) for a, _ in p.inputs]
    else:
        grads = [np.zeros(p.out_size_bound(), np.float32)]
    g_ptrs = (ctypes.c_void_p * max(len(grads), 1))(*[g.ctypes.data for g in grads])
    rc = fn(prog.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(p.ops)), in_ptrs,
            sizes.ctypes.data_as(ctypes.c_void_p), leaf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n_in),
            idx_ptrs, idx_sizes.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n_idx),
            0 if p.mode == "backward" else 1, int(p.fwd_leaf), int(p.simplify),
            out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(out_size), g_ptrs)
    if rc != 0:
        raise RuntimeError(f"tape program failed with rc={rc}")
    n = out_size.value
    if p.mode == "backward":
        return out[:n].copy(), [g if l else None for g, (_, l) in zip(grads, p.inputs)]
    return out[:n].copy(), [grads[0][:n].copy()]


_libs = {}


def ref_fn():
    if "ref" not in _libs:
        _libs["ref"] = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libenoki_ref.so"))
    return _libs["ref"].ref_tape_program


def ref512_available():
    """the AVX-512 flavour of the reference build (oracle/Makefile ref512) and a host that can run it"""
    path = os.path.join(ROOT, "oracle", "_ref", "libenoki_ref512.so")
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return os.path.exists(path) and all(f in flags for f in ("avx512f", "avx512dq", "avx512bw", "avx512vl", "avx512cd"))


def ref512_fn():
    if "ref512" not in _libs:
        _libs["ref512"] = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libenoki_ref512.so"))
    return _libs["ref512"].ref_tape_program


def refscalar_available():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libenoki_refscalar.so"))


def refscalar_fn():
    """ref_tape_program of the reference's SCALAR row (oracle/Makefile refscalar): rcp / rsqrt / division exact"""
    if "refscalar" not in _libs:
        _libs["refscalar"] = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libenoki_refscalar.so"))
    return _libs["refscalar"].ref_tape_program


def host_lib():
    if "host" not in _libs:
        _libs["host"] = ctypes.CDLL(os.path.join(HERE, "cpp", "libtape_host.so"))
        _libs["host"].host_tape_live_nodes.restype = ctypes.c_size_t
    return _libs["host"]


def hip_lib():
    if "hip" not in _libs:
        _libs["hip"] = ctypes.CDLL(os.path.join(HERE, "cpp", "libtape_hip.so"))
        _libs["hip"].hip_tape_live_nodes.restype = ctypes.c_size_t
    return _libs["hip"]


# ---------------------------------------------------------------------------------------------------
#  Program suite: every differentiable primitive of the first wave, broadcasting, duplicates,
#  gather / scatter / scatter_add specials, forward mode, graph simplification.
# ---------------------------------------------------------------------------------------------------
def suite(n=1000, k=37, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, n).astype(np.float32); x = rng.uniform(-1, 1, n).astype(np.float32)
    b = rng.uniform(-1, 1, n).astype(np.float32)
    pos = rng.uniform(0.5, 2, n).astype(np.float32)
    A = rng.uniform(-1, 1, k).astype(np.float32); B = rng.uniform(-1, 1, k).astype(np.float32)
    idx = rng.integers(0, k, n).astype(np.uint32)
    perm = rng.permutation(n).astype(np.uint32)
    s = np.array([0.75], np.float32)
    P = {}
    # BASELINE config 3a / 3b
    P["cfg3a"] = Program([(a, 1), (x, 0), (b, 1)], [("fmadd", 0, 1, 2), ("sin", 3), ("hsum", 4)])
    P["cfg3b"] = Program([(A, 1), (B, 1), (x, 0)], [("gather", 0, 0), ("gather", 1, 0), ("fmadd", 3, 2, 4), ("sin", 5),
                                                     ("hsum", 6)], index_inputs=[idx])
    P["cfg2_grad"] = Program([(a, 1), (x, 1), (b, 1)], [("fmadd", 0, 1, 2), ("exp", 3), ("sin", 4), ("hsum", 5)])
    # arithmetic
    P["arith"] = Program([(a, 1), (x, 1), (pos, 1)], [("mul", 0, 1), ("add", 3, 2), ("sub", 4, 0), ("neg", 5), ("abs", 6),
                                                      ("mulc", 7, 1.5), ("addc", 8, -0.25), ("hsum", 9)])
    P["square_dup_edge"] = Program([(a, 1)], [("mul", 0, 0), ("hsum", 1)])          # x*x: merged edge weights
    P["sqrt_log"] = Program([(pos, 1)], [("sqrt", 0), ("log", 1), ("hsum", 2)])
    P["cos_exp"] = Program([(a, 1)], [("cos", 0), ("exp", 1), ("hsum", 2)])
    P["minmax_select"] = Program([(a, 1), (x, 1)], [("min", 0, 1), ("max", 0, 1), ("select_gt0", 0, 2, 3), ("hsum", 4)])
    P["fm_family"] = Program([(a, 1), (x, 1), (b, 1)], [("fmsub", 0, 1, 2), ("fnmadd", 0, 1, 3), ("fnmsub", 4, 1, 2),
                                                        ("hsum", 5)])
    P["hprod"] = Program([((1 + 0.001 * a[:64]).astype(np.float32), 1)], [("hprod", 0)])
    # broadcasting: scalar leaf combined with a vector (gradient reduces with hsum)
    P["scalar_leaf"] = Program([(s, 1), (a, 1)], [("mul", 0, 1), ("sin", 2), ("hsum", 3)])
    P["scalar_chain"] = Program([(s, 1), (a, 0)], [("fmadd", 1, 0, 0), ("mul", 2, 2), ("hsum", 3)])
    P["vector_output"] = Program([(a, 1), (x, 1)], [("mul", 0, 1), ("sin", 2)])      # seed = ones over a vector
    # specials
    P["gather_only"] = Program([(A, 1)], [("gather", 0, 0), ("mulc", 1, 2.0), ("hsum", 2)], index_inputs=[idx])
    P["permute_gather"] = Program([(a, 1)], [("gather", 0, 0), ("mul", 1, 1), ("hsum", 2)], index_inputs=[perm])
    P["scatter_add"] = Program([(np.zeros(k, np.float32), 0), (a, 1)],
                               [("scatter_add", 0, 1, 0), ("mul", 2, 2), ("hsum", 3)], index_inputs=[idx])
    P["scatter_add_leaf_target"] = Program([(B, 1), (a, 1)], [("scatter_add", 0, 1, 0), ("sin", 2), ("hsum", 3)],
                                           index_inputs=[idx])
    # (a scatter into a LEAF target segfaults inside the reference build itself -- scatter_combine weight path,
    #  autodiff.cpp:588-592 -- so the plain scatter is exercised with a non-differentiable target)
    P["scatter_perm"] = Program([(np.zeros(n, np.float32), 0), (a, 1)], [("scatter", 0, 1, 0), ("mul", 2, 2), ("hsum", 3)],
                                index_inputs=[perm])
    P["reverse_psum"] = Program([(np.round(a * 8).astype(np.float32), 1), (x, 0)],
                                [("psum", 0), ("reverse", 2), ("mul", 3, 1), ("hsum", 4)])
    # forward mode
    P["fwd_cfg3a"] = Program([(a, 1), (x, 0), (b, 1)], [("fmadd", 0, 1, 2), ("sin", 3), ("hsum", 4)], mode="forward",
                             fwd_leaf=0)
    P["fwd_vector"] = Program([(a, 1), (x, 1)], [("mul", 0, 1), ("exp", 2), ("add", 3, 0)], mode="forward", fwd_leaf=0)
    P["fwd_gather"] = Program([(A, 1)], [("gather", 0, 0), ("sin", 1)], index_inputs=[idx], mode="forward", fwd_leaf=0)
    # graph simplification (explicit, like tests/autodiff.cpp:39-47)
    P["simplify_chain"] = Program([(a, 1), (x, 1)], [("mul", 0, 1), ("sin", 2), ("mulc", 3, 3.0), ("exp", 4), ("hsum", 5)],
                                  simplify=True)
    # class C inside weights: rcp / div / rsqrt use IEEE division on the GPU and in the oracle, rcpps+NR in the
    # AVX2 reference -> compared with a tolerance only
    P["div_rcp_rsqrt"] = Program([(a, 1), (pos, 1)], [("div", 0, 1), ("rcp", 1), ("rsqrt", 1), ("add", 2, 3),
                                                      ("add", 5, 4), ("hsum", 6)])
    # broadcast gradients reaching a scalar node (reference tests test05_hsum_1_fwd, test33_bcast): the sum over the
    # vector node's entries must be formed even when weight and gradient are both size-1 broadcasts
    P["bcast_scalar_leaf"] = Program([(s, 1), (a, 0)], [("add", 1, 0), ("hsum", 2)])
    P["bcast_scalar_leaf_mul"] = Program([(s, 1), (a, 1)], [("add", 1, 0), ("mul", 2, 1), ("addc", 3, 1.0), ("hsum", 4)])
    P["fwd_hsum_of_hsum"] = Program([(a, 1)], [("hsum", 0), ("mul", 1, 0), ("hsum", 2)], mode="forward", fwd_leaf=0)
    P["fwd_scalar_to_vector"] = Program([(s, 1), (a, 0)], [("mul", 0, 1), ("hsum", 2)], mode="forward", fwd_leaf=0)
    # second wave (array_math.h tan .. cbrt).  Vector outputs (seed = ones): the GPU tape must agree bit for bit with
    # the product tape over the CPU oracle; against the AVX2 reference only cbrt / pow are bit-comparable, every
    # other derivative contains rcp() or rsqrt() (class C)
    u = (0.9 * a).astype(np.float32); w = (0.9 * x).astype(np.float32)
    P["sw_trig"] = Program([(u, 1), (w, 1)], [("tan", 0), ("asin", 1), ("acos", 0), ("atan", 1), ("atan2", 0, 1),
                                              ("cot", 1), ("add", 2, 3), ("add", 8, 4), ("add", 9, 5), ("add", 10, 6),
                                              ("add", 11, 7)])
    P["sw_hyp"] = Program([(u, 1), (w, 1), (pos, 1)], [("sinh", 0), ("cosh", 1), ("tanh", 0), ("asinh", 1), ("addc", 2, 1.0),
                                                       ("acosh", 7), ("atanh", 0), ("add", 3, 4), ("add", 10, 5),
                                                       ("add", 11, 6), ("add", 12, 8), ("add", 13, 9)])
    P["sw_cbrt_pow"] = Program([(a, 1), (pos, 1), (x, 1)], [("cbrt", 0), ("pow", 1, 2), ("mul", 3, 4)])
    P["sw_sum"] = Program([(u, 1), (w, 1)], [("tanh", 0), ("atan2", 2, 1), ("asinh", 3), ("hsum", 4)])
    return P


def scatter_twice_suite(seed=11):
    """Two scatters into ONE buffer, then backward: the shape of the reference's tests/autodiff.cpp:433-466
    (test30_scatter), which segfaults inside the reference under the pinned AVX2 row and is therefore compared against the
    AVX-512 flavour (oracle/Makefile ref512).  Only exact arithmetic (small integers, products, sums of a few terms), so
    the packet width cannot change a bit."""
    rng = np.random.default_rng(seed)
    ints = lambda lo, hi, size: rng.integers(lo, hi + 1, size).astype(np.float32)
    P = {}
    # the literal test30 program (values scaled to integers): x -> buf[0..5), y -> buf[3..7), s = dot(buf, buf)
    x5, y4 = np.arange(5, dtype=np.float32), np.arange(4, dtype=np.float32) + 4
    P["test30"] = Program([(x5, 1), (y4, 1), (np.zeros(10, np.float32), 0)],
                          [("scatter", 2, 0, 0), ("scatter", 3, 1, 1), ("mul", 4, 4), ("hsum", 5)],
                          index_inputs=[np.arange(5, dtype=np.uint32), np.arange(4, dtype=np.uint32) + 3])
    for n, m in ((1000, 1500), (70001, 100003)):
        i1 = rng.permutation(m)[:n].astype(np.uint32)
        i2 = rng.permutation(m)[:(2 * n) // 3].astype(np.uint32)          # overlaps i1: those entries lose x's gradient
        x, y = ints(-4, 4, n), ints(-4, 4, i2.size)
        P[f"overwrite_{n}"] = Program([(x, 1), (y, 1), (np.zeros(m, np.float32), 0)],
                                      [("scatter", 2, 0, 0), ("scatter", 3, 1, 1), ("mul", 4, 4), ("hsum", 5)], index_inputs=[i1, i2])
        # a differentiable buffer underneath, and a scatter_add on top of the two scatters
        base, z = ints(-2, 2, m), ints(-2, 2, n)
        P[f"leafless_mix_{n}"] = Program([(x, 1), (y, 1), (base, 0), (z, 1)],
                                         [("mulc", 2, 2.0), ("scatter", 4, 0, 0), ("scatter", 5, 1, 1), ("scatter_add", 6, 3, 0),
                                          ("mul", 7, 7), ("hsum", 8)], index_inputs=[i1, i2])
    return P


def gather_suite(n=1000, k=37, seed=5):
    """Gather adjoints: the backward sweep batches the scatter_adds of gathers that share their index array and fuses the
    pending edge product into them (autodiff_impl.h: PendingScatter).  All data are small integers, so every product
    and every sum is exact in float32 -> the results do not depend on the accumulation order and all implementations
    (reference build, host tape, GPU tape on any of its scatter_add paths) must agree BIT FOR BIT."""
    rng = np.random.default_rng(seed)
    ints = lambda lo, hi, size: rng.integers(lo, hi + 1, size).astype(np.float32)
    A, B, C = ints(-8, 8, k), ints(-8, 8, k), ints(-8, 8, k)
    D = ints(-8, 8, k + 13)
    x, w, v = ints(-4, 4, n), ints(-4, 4, n), ints(-4, 4, n)
    i0 = rng.integers(0, k, n).astype(np.uint32); i1 = rng.integers(0, k, n).astype(np.uint32)
    m = max(n // 3, 1)
    j1 = rng.integers(0, k, m).astype(np.uint32); j2 = rng.integers(0, m, n).astype(np.uint32)
    P = {}
    # a = A[i0], b = B[i0]: u = a x + b, out = u u -> g_b = 2u (shared buffer), g_a = x * 2u (pending product)
    P["shared_idx_weighted"] = Program([(A, 1), (B, 1), (x, 0)], [("gather", 0, 0), ("gather", 1, 0), ("fmadd", 3, 2, 4), ("mul", 5, 5)],
                                       index_inputs=[i0])
    # three tables, all weighted
    P["three_tables"] = Program([(A, 1), (B, 1), (C, 1), (x, 0), (w, 0), (v, 0)],
                                [("gather", 0, 0), ("gather", 1, 0), ("gather", 2, 0), ("mul", 6, 3), ("mul", 7, 4), ("mul", 8, 5),
                                 ("add", 9, 10), ("add", 12, 11), ("mul", 13, 13)], index_inputs=[i0])
    # the gather node is used twice -> the pending product is materialised by the second contribution
    P["two_consumers"] = Program([(A, 1), (x, 0), (w, 0)], [("gather", 0, 0), ("mul", 3, 1), ("mul", 3, 2), ("add", 4, 5), ("mul", 6, 6)],
                                 index_inputs=[i0])
    # gather from a gathered array: the inner adjoint must have run before the outer node's gradient is read
    P["gather_of_gather"] = Program([(A, 1), (x, 0)], [("gather", 0, 0), ("gather", 2, 1), ("mul", 3, 1), ("mul", 4, 4)],
                                    index_inputs=[j1, j2])
    # the same table through the same index array twice (two streams must not target one table)
    P["same_table_twice"] = Program([(A, 1), (x, 0), (w, 0)], [("gather", 0, 0), ("gather", 0, 0), ("mul", 3, 1), ("mul", 4, 2), ("add", 5, 6)],
                                    index_inputs=[i0])
    # different index arrays / different table sizes: separate scatter_adds
    P["different_indices"] = Program([(A, 1), (B, 1), (x, 0), (w, 0)], [("gather", 0, 0), ("gather", 1, 1), ("mul", 4, 2), ("mul", 5, 3), ("add", 6, 7)],
                                     index_inputs=[i0, i1])
    P["different_sizes"] = Program([(A, 1), (D, 1), (x, 0), (w, 0)], [("gather", 0, 0), ("gather", 1, 0), ("mul", 4, 2), ("mul", 5, 3), ("add", 6, 7)],
                                   index_inputs=[i0])
    # broadcast gradients (hsum seeds) through a shared index array
    P["broadcast_grads"] = Program([(A, 1), (B, 1)], [("gather", 0, 0), ("gather", 1, 0), ("add", 2, 3), ("hsum", 4)], index_inputs=[i0])
    # A gather node whose gradient is still a pending product w * g (its multiplication was recorded LAST, so the sweep
    # reaches it first) and that also feeds an earlier-recorded reverse / psum / scatter_add / gather: those adjoints
    # accumulate straight into the node's gradient and must see w * g, not g.
    wm = ints(-4, 4, m)
    for name in ("reverse", "psum"):
        P[f"pending_then_{name}"] = Program([(A, 1), (w, 0), (v, 0)],
                                            [("gather", 0, 0), (name, 3), ("mul", 3, 1), ("mul", 5, 2), ("mul", 4, 2), ("add", 6, 7)],
                                            index_inputs=[i0])
    P["pending_then_scatter_add"] = Program([(A, 1), (w, 0), (v, 0), (np.zeros(k, np.float32), 0)],
                                            [("gather", 0, 0), ("scatter_add", 3, 4, 1), ("mul", 4, 1), ("mul", 6, 2), ("hsum", 7),
                                             ("mul", 5, 5), ("hsum", 9), ("add", 8, 10)], index_inputs=[i0, i1])
    P["pending_then_gather"] = Program([(A, 1), (wm, 0), (v, 0)],
                                       [("gather", 0, 0), ("gather", 3, 1), ("mul", 3, 1), ("mul", 4, 2), ("mul", 5, 5), ("hsum", 7),
                                        ("hsum", 6), ("add", 8, 9)], index_inputs=[j1, j2])
    # scalar weight times vector gradient
    P["scalar_weight"] = Program([(A, 1), (B, 1), (x, 0)], [("gather", 0, 0), ("gather", 1, 0), ("mulc", 3, 3.0), ("mul", 4, 2), ("add", 5, 6), ("mul", 7, 7)],
                                 index_inputs=[i0])
    return P


def random_gather_program(seed, n=1000, k=37, n_ops=18):
    """Random programs mixing table arithmetic (size k), gathers through a few shared index arrays and vector arithmetic
    (size n), vector output.  Small integer data and a bounded number of multiplications keep every value, product and
    sum exactly representable, so all tape implementations must agree bit for bit whatever the accumulation order;
    interior (computed) tables are gathered from as well, which exercises the ordering of the deferred adjoints."""
    rng = np.random.default_rng(5000 + seed)
    ints = lambda lo, hi, size: rng.integers(lo, hi + 1, size).astype(np.float32)
    n_tables, n_vecs, n_idx = 3, 2, 2
    ins = [(ints(-3, 3, k), 1) for _ in range(n_tables)] + [(ints(-2, 2, n), int(rng.integers(0, 2))) for _ in range(n_vecs)]
    idx = [rng.integers(0, k, n).astype(np.uint32) for _ in range(n_idx)]
    kind = ["t"] * n_tables + ["v"] * n_vecs            # register kinds
    depth = [0] * len(kind)                             # multiplications on the path: bounds the magnitudes
    ops = []

    def pick(kd, max_depth=10):
        c = [r for r in range(len(kind)) if kind[r] == kd and depth[r] <= max_depth]
        return int(c[int(rng.integers(0, len(c)))]) if c else None

    def push(op, kd, d):
        ops.append(op); kind.append(kd); depth.append(d)

    for _ in range(n_ops):
        r = rng.random()
        if r < 0.35:                                    # gather from any table register through one of the index arrays
            t = pick("t")
            push(("gather", t, int(rng.integers(0, n_idx))), "v", depth[t])
        elif r < 0.5:                                   # table arithmetic -> interior tables
            a, b = pick("t", 0), pick("t", 0)
            name = ["add", "sub", "mul"][int(rng.integers(0, 3))]
            push((name, a, b), "t", max(depth[a], depth[b]) + (1 if name == "mul" else 0))
        else:
            name = ["add", "sub", "mul", "neg", "mulc", "fmadd"][int(rng.integers(0, 6))]
            if name == "neg":
                a = pick("v"); push(("neg", a), "v", depth[a])
            elif name == "mulc":
                a = pick("v"); push(("mulc", a, float(rng.integers(-2, 3))), "v", depth[a])
            elif name in ("mul", "fmadd"):
                a, b = pick("v", 0), pick("v", 0)
                if name == "mul":
                    push(("mul", a, b), "v", max(depth[a], depth[b]) + 1)
                else:
                    c = pick("v", 1); push(("fmadd", a, b, c), "v", max(depth[a], depth[b], depth[c]) + 1)
            else:
                a, b = pick("v"), pick("v"); push((name, a, b), "v", max(depth[a], depth[b]))
    # make sure every table is used, and tie the last vector registers together (vector output, seed = ones)
    for t in range(n_tables):
        push(("gather", t, 0), "v", 0)
    vec = [r for r in range(len(kind)) if kind[r] == "v"][-6:]
    acc = vec[0]
    for r in vec[1:]:
        push(("add", acc, r), "v", max(depth[acc], depth[r])); acc = len(kind) - 1
    return Program(ins, ops, index_inputs=idx)


TOLERANT = {"div_rcp_rsqrt", "sw_trig", "sw_hyp", "sw_sum", "sw_cbrt_pow"}   # sw_cbrt_pow: d pow / d base goes through log's rcp()
CLASS_C_VALUES = {"sw_trig", "sw_hyp", "sw_sum"}   # the primal itself contains rcp() (tan, cot, sinh, cosh, tanh)          # not bit-comparable against the AVX2 reference build (class C)
ORDER_DEPENDENT_ON_GPU = {            # contain hsum / hprod / fp scatter_add: GPU summation order differs (class D)
    "cfg3a", "cfg3b", "cfg2_grad", "arith", "square_dup_edge", "sqrt_log", "cos_exp", "minmax_select", "fm_family",
    "hprod", "scalar_leaf", "scalar_chain", "gather_only", "permute_gather", "scatter_add", "scatter_add_leaf_target",
    "scatter_perm", "reverse_psum", "fwd_cfg3a", "simplify_chain", "div_rcp_rsqrt", "sw_sum", "bcast_scalar_leaf",
    "bcast_scalar_leaf_mul", "fwd_hsum_of_hsum", "fwd_scalar_to_vector",
}


# ---------------------------------------------------------------------------------------------------
#  Random purely-vertical programs (vector output, seed = ones): every op is class A, so the product tape must agree
#  BIT FOR BIT with the reference build -- on the CPU oracle arrays and on the GPU.
# ---------------------------------------------------------------------------------------------------
def random_program(seed, n=257, n_ops=14, mode="backward"):
    rng = np.random.default_rng(1000 + seed)
    ins = [(rng.uniform(-1, 1, n).astype(np.float32), 1), (rng.uniform(-1, 1, n).astype(np.float32), 1),
           (rng.uniform(-1, 1, n).astype(np.float32), int(rng.integers(0, 2))), (np.array([rng.uniform(0.5, 1.5)], np.float32), 1)]
    n_in = len(ins)
    ops = []
    unary = ["neg", "abs", "sin", "cos", "exp", "mulc", "addc", "sqrt_safe"]
    binary = ["add", "sub", "mul", "min", "max"]
    ternary = ["fmadd", "fmsub", "fnmadd", "fnmsub", "select_gt0"]
    for _ in range(n_ops):
        n_reg = n_in + len(ops)
        pick = lambda: int(rng.integers(0, n_reg))
        kind = rng.integers(0, 3)
        if kind == 0:
            name = unary[int(rng.integers(0, len(unary)))]
            if name == "mulc":
                ops.append(("mulc", pick(), float(np.float32(rng.uniform(-2, 2)))))
            elif name == "addc":
                ops.append(("addc", pick(), float(np.float32(rng.uniform(-1, 1)))))
            elif name == "sqrt_safe":                      # sqrt(|r| + 0.5): two helper ops + sqrt
                ops.append(("abs", pick())); ops.append(("addc", n_in + len(ops) - 1, 0.5)); ops.append(("sqrt", n_in + len(ops) - 1))
            else:
                ops.append((name, pick()))
        elif kind == 1:
            ops.append((binary[int(rng.integers(0, len(binary)))], pick(), pick()))
        else:
            ops.append((ternary[int(rng.integers(0, len(ternary)))], pick(), pick(), pick()))
    # tie the last registers together so that most of the graph is live, and keep the output a vector
    last = n_in + len(ops) - 1
    ops.append(("add", last, max(last - 1, 0))); ops.append(("add", n_in + len(ops) - 1, 0))
    return Program(ins, ops, mode=mode, fwd_leaf=0)
