/*
 * include/enoki_hip.h -- C ABI of libenoki-hip.so, the MI355X (gfx950) array runtime.
 *
 * This is the drop-in boundary for the reference's `libenoki-cuda.so` (exports declared in
 * /root/reference/include/enoki/cuda.h:27-200).  The reference's exports are C++-mangled and
 * operate on *trace indices* of a PTX JIT; this library is eager: every entry point takes raw
 * device pointers + element counts and launches one pre-compiled HIP kernel on the library's
 * stream.  `include/enoki/hip.h` (HIPArray<T>) is the C++ binding that sits on top of it and
 * plays the role of `CUDAArray<T>` (cuda.h:205-954); INTEGRATION.md shows the reference-side
 * stub a maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++ / torch types.
 *  - every function returns 0 on success or a negative EK_ERR_* code; ek_hip_last_error()
 *    returns a thread-local human readable message.  HIP API failures are reported the same
 *    way (the reference exits the process, src/cuda/common.cu:268-286; a library should not).
 *  - all work is enqueued on ONE stream (ek_hip_stream()/ek_hip_set_stream()); nothing
 *    synchronizes unless documented (ek_hip_sync, *_to_host copies, all/any/count).
 *  - masks are arrays of uint8_t (0/1), as in the reference (src/cuda/jit.cu:1137-1144).
 *  - operands of vertical ops are `ek_operand`: a device array of `size` n or 1 (size-1
 *    operands broadcast, cuda.h semantics via jit.cu:776-782), or an immediate scalar.
 */
#ifndef ENOKI_HIP_C_API_H
#define ENOKI_HIP_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#  define EK_API
#else
#  define EK_API __attribute__((visibility("default")))
#endif

/* Element types -- the subset of EnokiType (src/cuda/common.cuh:25-40) on the hot path. */
typedef enum {
    EK_BOOL = 0, EK_I32 = 1, EK_U32 = 2, EK_I64 = 3, EK_U64 = 4, EK_F32 = 5, EK_F64 = 6,
    EK_TYPE_COUNT = 7
} ek_type;

typedef enum {
    EK_OK = 0,
    EK_ERR_INVALID = -1,      /* bad argument (null pointer, unknown op/type, size mismatch) */
    EK_ERR_UNSUPPORTED = -2,  /* op not defined for this element type */
    EK_ERR_HIP = -3,          /* a HIP runtime call failed; see ek_hip_last_error() */
    EK_ERR_OOM = -4           /* allocation failed even after trimming the cache */
} ek_status;

/* Unary ops.  cuda.h:418-543 (abs_ .. tzcnt_) and array_math.h (sin/cos/exp/log, CPU algorithm).
 * EK_TAN .. EK_CBRT (f32 only): the "second wave" the reference composes generically from traced
 * primitives (array_math.h:466-474, 555, 666, 900, 997-1348); here one fused kernel each. */
typedef enum {
    EK_NEG = 0, EK_ABS, EK_NOT, EK_SQRT, EK_RCP, EK_RSQRT, EK_FLOOR, EK_CEIL, EK_ROUND, EK_TRUNC,
    EK_SIN, EK_COS, EK_EXP, EK_LOG, EK_POPCNT, EK_LZCNT, EK_TZCNT, EK_SIGN, EK_COPY,
    EK_TAN, EK_COT, EK_ASIN, EK_ACOS, EK_ATAN, EK_SINH, EK_COSH, EK_TANH, EK_ASINH, EK_ACOSH, EK_ATANH,
    EK_CBRT,
    EK_ERF, EK_ERFC, EK_ERFINV, EK_I0E, EK_DAWSON, EK_ERFI, EK_LGAMMA, EK_TGAMMA,   /* special.h:56-312, f32 and f64 */
    /* what the derivatives of rcp and rsqrt are made of, as ONE function of the argument each (autodiff.h:381-403: -sqr(result),
     * -.5 result^3 with result = rcp(x) / rsqrt(x)): r r with r = 1 / x;  r r and r (r r) with r = 1 / sqrt(x) -- the roundings of
     * the eager products, so that a consumer which applies an op while it loads can form them from x */
    EK_RCP_SQR, EK_RSQRT_SQR, EK_RSQRT_CUBE,
    /* round 6: the derivative weights of tan, tanh and atan as ONE function of the argument each, with the roundings of the compositions
     * the reference records (autodiff.h:532-541 sqr(sec(x)), :685-696 sqr(sech(x)), :606-616 rcp(1 + sqr(x)); sec = rcp(cos),
     * sech = rcp(cosh), array_math.h:463-464, 1181-1183): r r with r = 1 / cos(x);  r r with r = 1 / cosh(x);  1 / (1 + x x) */
    EK_SEC_SQR, EK_SECH_SQR, EK_RCP_1P_SQR,
    EK_UNARY_COUNT
} ek_unary_op;

/* Binary ops.  cuda.h:341-385, 408-416, 499-580; safe_mul = autodiff.cpp:1191-1205;
 * atan2(y, x) / pow / fmod / ldexp = array_math.h:603-664, 956-958, 1381-1383, 677-680 (atan2, pow, ldexp: f32). */
typedef enum {
    EK_ADD = 0, EK_SUB, EK_MUL, EK_DIV, EK_MOD, EK_MIN, EK_MAX, EK_MULHI, EK_AND, EK_OR, EK_XOR,
    EK_SL, EK_SR, EK_SAFE_MUL, EK_ATAN2, EK_POW, EK_FMOD, EK_LDEXP,
    EK_BINARY_COUNT
} ek_binary_op;

/* Ternary ops.  cuda.h:387-406; safe_fmadd = autodiff.cpp:1207-1221.
 * EK_MULADD / EK_MULSUB / EK_NMULADD (floating point): a * b + c, a * b - c, c - a * b written with OPERATORS -- a multiplication
 * and an addition with a rounding each, what `a*x+b` means in the reference (separate packet operations are never contracted,
 * SURVEY 8c) -- as ONE pass over the operands.  A caller-visible op only so that a product left unevaluated by HIPArray can be
 * finished by the addition that consumes it; bit-identical to ek_hip_binary(EK_MUL) followed by EK_ADD / EK_SUB. */
typedef enum {
    EK_FMADD = 0, EK_FMSUB, EK_FNMADD, EK_FNMSUB, EK_SAFE_FMADD, EK_MULADD, EK_MULSUB, EK_NMULADD,
    EK_TERNARY_COUNT
} ek_ternary_op;

/* Comparisons -> u8 mask.  cuda.h:582-630. */
typedef enum { EK_EQ = 0, EK_NEQ, EK_LT, EK_LE, EK_GT, EK_GE, EK_COMPARE_COUNT } ek_compare_op;

/* Horizontal reductions.  cuda.h:693-759 / src/cuda/horiz.cu:162-268. */
typedef enum { EK_HSUM = 0, EK_HPROD, EK_HMIN, EK_HMAX, EK_REDUCE_COUNT } ek_reduce_op;

/* Mask reductions.  cuda.h:761-794 / horiz.cu:284-354. */
typedef enum { EK_ALL = 0, EK_ANY, EK_COUNT, EK_MASK_REDUCE_COUNT } ek_mask_reduce_op;

/* An operand of a vertical op: device array (`ptr` != NULL, `size` == n or 1) or an immediate
   (`ptr` == NULL; the scalar's bit pattern in the low bytes of `imm`). */
typedef struct {
    const void *ptr;
    uint64_t imm;
    size_t size;
} ek_operand;

/* ---------------------------------------------------------------------------------------------
 *  Runtime: device, stream, memory, diagnostics
 *  replaces cuda_malloc, cuda_free, cuda_malloc_trim, cuda_sync, cuda_memcpy_to_device/from_device,
 *  cuda_mem_get_info, cuda_whos, cuda_set_log_level
 *  (cuda.h:109-200; src/cuda/jit.cu:1683-1896, common.cu:104-122)
 * ------------------------------------------------------------------------------------------- */
EK_API int ek_hip_init(int device);                 /* select device, create stream (idempotent) */
EK_API int ek_hip_device(void);                     /* device ordinal in use, -1 before init */
EK_API int ek_hip_device_count(void);
EK_API void *ek_hip_stream(void);                   /* hipStream_t */
EK_API int ek_hip_set_stream(void *hip_stream);     /* adopt a caller-owned stream (e.g. torch's) */
EK_API int ek_hip_sync(void);                       /* hipStreamSynchronize on the library stream */
EK_API const char *ek_hip_last_error(void);

EK_API int ek_hip_malloc(size_t bytes, void **out);         /* caching allocator, 256-B aligned */
EK_API int ek_hip_free(void *ptr);                          /* returns the block to the cache */
EK_API int ek_hip_malloc_trim(void);                        /* release cached blocks to the driver */
EK_API int ek_hip_host_malloc(size_t bytes, void **out);    /* pinned host memory */
EK_API int ek_hip_host_free(void *ptr);
EK_API int ek_hip_mem_get_info(size_t *free_bytes, size_t *total_bytes);
EK_API int ek_hip_memcpy_to_device(void *dst, const void *src, size_t bytes);   /* blocking */
EK_API int ek_hip_memcpy_to_host(void *dst, const void *src, size_t bytes);     /* blocking */
EK_API int ek_hip_memcpy_device(void *dst, const void *src, size_t bytes);      /* async d2d */
EK_API int ek_hip_memset(void *dst, int byte, size_t bytes);                    /* async */
/* malloc'd, NUL-terminated allocator report; caller frees with free() (cuda_whos, cuda.cpp:40) */
EK_API char *ek_hip_whos(void);
EK_API void ek_hip_set_log_level(uint32_t level);   /* 0 silent .. 3 every launch (cuda.h:195-200) */
EK_API uint32_t ek_hip_log_level(void);
EK_API uint64_t ek_hip_launch_count(void);          /* kernels launched since init (diagnostics) */
/* Kernels that callers launch THEMSELVES on ek_hip_stream() (the fused kernels that enoki/vectorize.h instantiates from user
   code) report here so that they show up in ek_hip_launch_count(), the ENOKI_HIP_LOG=3 trace and ek_hip_profile_*;
   `bytes` = algorithmic bytes of the launch.  `name` must outlive the profile (a string literal). */
EK_API int ek_hip_note_launch(const char *name, size_t n, size_t bytes);
/* One process-wide pointer variable for the C++ binding (include/enoki/hip.h keeps the head of its list of unevaluated
   buffers here): header-only code that is compiled into several shared objects needs ONE home for such state, and this
   library is the one object they all link.  Returns the variable's address; never NULL; no initialisation needed. */
EK_API void **ek_hip_binding_slot(void);
/* Step graphs (hipGraph): between ek_hip_graph_begin() and ek_hip_graph_end() every launch of this library -- kernels,
   memsets, device-to-device copies -- is CAPTURED on the library stream instead of executed; ek_hip_graph_launch()
   replays the captured step without any host-side work (no tape walk, no allocator, no per-kernel launch call), which
   is what strong scaling needs once a shard's step is a dozen kernels of 10-50 us.  Memory that the captured code
   allocates comes from a pool private to the graph and stays reserved until ek_hip_graph_destroy(): the arrays that
   are alive when the capture ends (results, gradients) keep their addresses, and every replay refreshes their contents.
   Not capturable (fail with EK_ERR_INVALID / EK_ERR_HIP): anything that reads back to the host -- ek_hip_memcpy_to_host,
   ek_hip_mask_reduce (all / any / count), deterministic scatter_add, ek_hip_profile_*.
   The reference has no counterpart: its cuda_eval() re-assembles and re-launches PTX per evaluation (jit.cu:1385-1471). */
typedef struct ek_hip_graph ek_hip_graph;
EK_API int ek_hip_graph_begin(void);
EK_API int ek_hip_graph_end(ek_hip_graph **out);
EK_API int ek_hip_graph_launch(ek_hip_graph *graph);
EK_API uint64_t ek_hip_graph_launch_count(const ek_hip_graph *graph);   /* kernel launches inside one replay */
EK_API int ek_hip_graph_destroy(ek_hip_graph *graph);
EK_API int ek_hip_set_tuning(const char *key, int value);   /* "reduce_blocks_per_cu", "scatter_add_binned", "deterministic", "gather_records", "bucket_ordered", "early_adjoint" */
/* Per-kernel timing: between begin and end one HIP event is recorded on the library stream after every
   launch.  ek_hip_profile_end() synchronizes and returns a malloc'd JSON array (caller free()s) of
   {"kernel", "launches", "total_ms", "bytes", "elements"}; `bytes` are the ALGORITHMIC bytes of the
   launches (distinct operand bytes + output bytes; broadcast operands count 0). */
EK_API int ek_hip_profile_begin(void);
EK_API char *ek_hip_profile_end(void);

/* ---------------------------------------------------------------------------------------------
 *  Vertical (elementwise) ops: out[i] = op(a[i or 0], ...), i in [0, n)
 * ------------------------------------------------------------------------------------------- */
EK_API int ek_hip_unary(int op, int type, void *out, const ek_operand *a, size_t n);
EK_API int ek_hip_binary(int op, int type, void *out, const ek_operand *a, const ek_operand *b, size_t n);
EK_API int ek_hip_ternary(int op, int type, void *out, const ek_operand *a, const ek_operand *b,
                          const ek_operand *c, size_t n);
/* sincos_: both outputs from one pass over the input (array_math.h:261-367) */
EK_API int ek_hip_sincos(int type, void *out_sin, void *out_cos, const ek_operand *a, size_t n);
/* sincosh: sinh and cosh from one exp() (array_math.h:1067-1127), what DiffArray::sinh_/cosh_ call (autodiff.h:635-657) */
EK_API int ek_hip_sincosh(int type, void *out_sinh, void *out_cosh, const ek_operand *a, size_t n);
EK_API int ek_hip_compare(int op, int type, uint8_t *out_mask, const ek_operand *a, const ek_operand *b, size_t n);
/* select_: out = mask ? t : f (cuda.h:632-639); `mask` is an EK_BOOL operand */
EK_API int ek_hip_select(int type, void *out, const ek_operand *mask, const ek_operand *t,
                         const ek_operand *f, size_t n);
/* converting constructor (cuda.h:236-247): float->int truncates, int->float rounds to nearest */
EK_API int ek_hip_cast(int src_type, int dst_type, void *out, const ek_operand *a, size_t n);

/* ---------------------------------------------------------------------------------------------
 *  Initialization (cuda.h:641-691; horiz.cu:28-32; common.cu:56-102)
 * ------------------------------------------------------------------------------------------- */
EK_API int ek_hip_fill(int type, void *out, uint64_t imm_bits, size_t n);
/* out[i] = start + i*step, computed in the element type (integer types wrap) */
EK_API int ek_hip_arange(int type, void *out, int64_t start, int64_t step, size_t n);
/* out[i] = fmadd(i, step, min), step = (max-min)/(n-1)  (cuda.h:655-663) */
EK_API int ek_hip_linspace(int type, void *out, double min, double max, size_t n);
EK_API int ek_hip_reverse(int type, void *out, const void *in, size_t n);
/* out = srcs[0] | srcs[1] | ... (sizes in elements, count <= 8) in ONE launch: the staging step of the packed all-reduce
 * (scalar loss + K-element gradients -> one flat buffer), where three small copies would cost three launches */
EK_API int ek_hip_concat(int type, void *out, int count, const void *const *srcs, const size_t *sizes);
/* The staging step of a REDUCE-SCATTER over `rows` ranks: every source (sizes[k] a multiple of `rows`) is seen as [rows, c_k]
 * and the sources are concatenated along the columns -- out[r] = srcs[0][r] | srcs[1][r] | ..., i.e. row r holds the chunks of
 * all tables that rank r will own -- in ONE launch (count <= 8, 4- and 8-byte types).  A source of ONE entry is copied into
 * every row: a scalar partial sum (the loss) rides on the same reduce-scatter -- every rank then receives sum_r(loss_r) in
 * that column -- instead of on a collective of its own. */
EK_API int ek_hip_concat_rows(int type, void *out, size_t rows, int count, const void *const *srcs, const size_t *sizes);

/* ---------------------------------------------------------------------------------------------
 *  Indexed memory ops (cuda.h:845-905).  `index_type` in {EK_I32, EK_U32, EK_I64, EK_U64};
 *  element stride = sizeof(type); `mask` may be an immediate operand (all lanes on/off).
 * ------------------------------------------------------------------------------------------- */
EK_API int ek_hip_gather(int type, int index_type, void *out, const void *base,
                         const ek_operand *index, const ek_operand *mask, size_t n);
/* out[i] = mask[i] && address[i] ? *(type *) (address[i] + byte_offset) : 0: a data member read out of INSTANCE memory through an
 * array of 64-bit object addresses -- what ENOKI_CALL_SUPPORT_GETTER does on the device (array_call.h:269-283:
 * gather<Return, 1>(nullptr, self + offset, mask) from managed instance memory).  The addresses must be readable by the GPU
 * (ek_hip_host_malloc / ENOKI_PINNED_OPERATOR_NEW, like the reference's pinned instances, array_macro.h:361). */
EK_API int ek_hip_gather_address(int type, void *out, const ek_operand *address, int64_t byte_offset, const ek_operand *mask, size_t n);
/* Gather of a structure-of-arrays value: `count` (2..4) tables of the same element type share ONE index / mask array
 * (gather<Array<HIPArray<T>, N>>(...), array_struct.h:9-40 calls gather_ once per component).  outs[c][i] =
 * mask[i] ? bases[c][index[i]] : 0.  4- and 8-byte element types. */
EK_API int ek_hip_gather_multi(int type, int index_type, int count, void *const *outs, const void *const *bases,
                               const ek_operand *index, const ek_operand *mask, size_t n);
/* The same with the tables' common length known (`base_size` entries each; 0 = unknown, plain ek_hip_gather_multi).  When
 * the tables are too large for the L2 and there are enough lookups to pay for it, the components are first staged as
 * packed 8- or 16-byte records {x, y(, z, w)} and every lane issues ONE request per element instead of `count`: random
 * lookups are bound by the number of memory requests, not by bytes.  Same results bit for bit.
 * Replaces the per-component gather_ calls of array_struct.h:9-40 for Array<CUDAArray<T>, N> sources. */
enum { EK_GATHER_PER_TABLE = 0, EK_GATHER_ONE_LAUNCH = 1, EK_GATHER_RECORDS = 2 };
/* which of the three ek_hip_gather_multi_sized() would take for this shape (callers that can leave per-table gathers
 * unevaluated -- enoki/hip.h consumes them inside the next arithmetic kernel -- ask before committing to a launch) */
EK_API int ek_hip_gather_multi_plan(int type, int index_type, int count, size_t base_size, size_t n);
EK_API int ek_hip_gather_multi_sized(int type, int index_type, int count, void *const *outs, const void *const *bases,
                                     size_t base_size, const ek_operand *index, const ek_operand *mask, size_t n);
/* An operand that is read THROUGH an index array: value[i] = mask[i] ? table[index[i]] : 0 -- a gather (cuda.h:845-864)
 * whose result is consumed by the next vertical op instead of being written out.  The reference gets this for free:
 * its JIT emits the gather's `ld.global` into the consumer's kernel (jit.cu:1066-1217). */
typedef struct {
    const void *table;       /* `table_size` elements of the op's element type */
    size_t table_size;
    ek_operand index;        /* EK_U32 / EK_I32 ARRAY of the op's size */
    int index_type;
    ek_operand mask;         /* EK_BOOL array of the op's size, or an immediate */
} ek_gathered;
/* out[i] = op(x0[i], x1[i] (, x2[i])) where operand k is `*gathered[k]` when that pointer is non-NULL and `*operands[k]`
 * otherwise; `arity` 2 (`op` an ek_binary_op: ADD, SUB, MUL) or 3 (`op` an ek_ternary_op: FMADD .. FNMSUB), EK_F32 / EK_F64.
 * One gathered operand per launch -- or, for the fma family, a gathered first (or second) AND third operand that share
 * their index and mask arrays and table size: the two tables are interleaved into {a, c} records first and every element
 * issues ONE 8-byte lookup (random lookups are request-rate bound, not byte bound).  Bit-identical to ek_hip_gather followed by
 * ek_hip_binary / ek_hip_ternary; returns EK_ERR_UNSUPPORTED for every other combination (callers then do exactly that). */
EK_API int ek_hip_map_gathered(int arity, int op, int type, void *out, const ek_operand *const *operands,
                               const ek_gathered *const *gathered, size_t n);
/* Bucket-ordered evaluation of  u[i] = op(A[index[i]], x[i], C[index[i]])  (op of the fma family) for consumers that do not
 * care about the ELEMENT order of u: horizontal reductions, and the adjoint scatter_add of the two gathers through the same
 * index array.  The reference gets the forward half from its JIT -- the gathers' ld.global and the arithmetic are emitted
 * into one kernel per evaluation (src/cuda/jit.cu:984, 1066-1217, 1418-1471) -- and pays atom.global.add per element in the
 * adjoint (cuda.h:892-905).  Here (index, x) is partitioned ONCE by bucket of 16 Ki table entries (8 Ki for 8-byte types);
 * every bucket's {A, C} slice is then served from LDS (no lookup leaves the CU: element-order lookups into tables beyond
 * the 4 MiB L2 of an XCD run at 0.25 of the HBM roofline), and the adjoint reuses the partition instead of counting,
 * scanning and partitioning again.
 *   create        the partition -- 4-byte element types: ONE streaming pass into pages (csrc/ek_paged.h: full-line pages per bucket,
 *                 no count pass, no scans) + the page directory; 8-byte types: count / scan / partition into contiguous runs;
 *                 4-byte tables beyond 256 buckets are cut into slices of 256 buckets, one filtered pass per slice.
 *                 `table_a`, `table_c` (`table_size` entries each), are read by later calls:
 *                 the caller keeps them alive and unchanged for the lifetime of the object.  x and index: n-element arrays.
 *                 EK_ERR_UNSUPPORTED for shapes the path does not cover (ask ek_hip_bucketed_applicable first): tables of
 *                 one bucket, 8-byte tables of more than 256 buckets, 4-byte tables of more than 32 x 256, fewer than 256 Ki or
 *                 (4-byte types) 2^30 or more lookups, deterministic mode, non-fp types.  Indices outside the table are dropped.
 *   reduce        out[0] = reduce_op over map_op(u)  (map_op: EK_COPY or an op ek_hip_reduce_map accepts); keep_values != 0
 *                 also keeps u in bucket order for later calls (4 B/elt more) -- or, when {map_op, keep_op} = {EK_SIN, EK_COS},
 *                 the OTHER half of sincos(u): one sincos per element yields the reduced and the kept half, and a later
 *                 scatter_add of that half (the cos(u) of d/du sin(u)) streams it as is.  keep_op = EK_COPY: keep u.
 *   scatter_add   bases[c][index[i]] += (weighted[c] ? safe_mul(x[i], v_c[i]) : v_c[i]),  v_c = from_u[c] ? map_ops[c](u) :
 *                 the scalar imm_bits[c];  count 1..4 tables of table_size entries.  fresh (may be NULL): fresh[c] != 0 says
 *                 that table c holds no data yet -- a gradient buffer that would otherwise be zero-filled first: its sums are
 *                 WRITTEN (bases[c][k] = sum), saving the fill and the read of the old contents.  The tables must be
 *                 distinct buffers (Tape::flush_pending never queues two scatters into one node's gradient, and
 *                 HIPArray::scatter_add_multi_ un-shares copy-on-write handles before it calls this).
 *   create_hinted hints = EK_BUCKETED_HINT_ADJOINT: the caller expects `reduce(EK_HSUM, sin | cos, keep the other half)` followed
 *                 by the scatter_add of that half and of x times that half -- y = hsum(sin(u)) recorded on a tape whose gathers
 *                 need gradients (autodiff.cpp:899-918 gather, :1191-1199 the edge product).  Because that adjoint is linear in
 *                 its seed, the partition is then made with buckets of HALF the size, the reduce call forms both sums in the
 *                 same pass over (l16, x) -- table slice and gradient tables share the LDS; the kept half is neither written nor
 *                 read back -- and the scatter_add call only folds the per-piece tables.  Any other sequence of calls on a
 *                 hinted object works as on an unhinted one.  Ignored for tables beyond 256 half-size buckets and when
 *                 ek_hip_set_tuning("early_adjoint", 0).
 *                 hints |= EK_BUCKETED_HINT_BOUNDED (round 6): the pair of functions is bounded by 1 (sin / cos).  The partition is
 *                 then made with buckets of a QUARTER of the size and the two sums per table entry are formed in 64-bit FIXED
 *                 POINT by non-returning LDS atomics (scale from the bound 1, from max |x| -- which the partition records -- and
 *                 from n, so that no sum can leave 63 bits): no locks, and sums that do not depend on the order of the additions --
 *                 the gradients are BIT-IDENTICAL from run to run (the reference's scatter_add is fixed-order: dynamic.h:517-534).
 *                 A term of 24 significant bits is exact from 2^-13 of its bound upwards; below that its error is < 2^-37 of the
 *                 bound at 64 Mi elements.  Not finite max |x|, or a NaN term (non-finite table entries, an overflowing u): the
 *                 pieces concerned run under the exchange locks as without the hint.  Ignored beyond 256 quarter-size buckets
 *                 (K > 1 Mi entries) and under ENOKI_HIP_EARLY_SUMS=locks.
 * Values of u are bit-identical to the element-order kernels; reductions and sums differ by the ORDER of their fp
 * additions only (unspecified, like ek_hip_reduce / ek_hip_scatter_add mode 0). */
typedef struct ek_hip_bucketed ek_hip_bucketed;
enum { EK_BUCKETED_HINT_ADJOINT = 1, EK_BUCKETED_HINT_BOUNDED = 2 };
EK_API int ek_hip_bucketed_applicable(int type, int index_type, size_t table_size, size_t n);
EK_API int ek_hip_bucketed_pair_create(int type, int index_type, int op, const void *table_a, const void *table_c,
                                       size_t table_size, const void *x, const void *index, size_t n, ek_hip_bucketed **out);
EK_API int ek_hip_bucketed_pair_create_hinted(int type, int index_type, int op, const void *table_a, const void *table_c,
                                              size_t table_size, const void *x, const void *index, size_t n, unsigned hints,
                                              ek_hip_bucketed **out);
/* table_c == NULL with op == EK_MULADD: the product  u = gather(A, idx) * x  alone (no addend table; u = a x + (-0) = a x bit for
 * bit) -- reductions and the adjoint scatter_add of the ONE gather (the stream x * f'(u)) run in bucket order like the pair's.
 * Both gathers under ONE mask array (cuda.h:845-864: masked-out lanes gather 0): inactive entries are dropped by the partition,
 * and what they would have contributed is added by the final step of a reduction: u = fma(0, x, 0) is 0 for a finite x (the lane
 * enters as map_op(0)) and NaN for an infinite or NaN x -- hsum / hprod then are NaN, exactly as the reference's lane-by-lane
 * evaluation (dynamic.h:632-650) and this library's element-order kernels say.  Dropped lanes scatter nothing.
 * ONE rule for every dropped lane: an index outside [0, table_size) -- unspecified in the reference (cuda.h:845-905) -- is
 * treated like a cleared mask bit with a finite x (u = 0, contributes map_op(0), scatters nothing), for a single object and for
 * the slices of a large table alike.
 * mask: n bytes or NULL; 4-byte element types only. */
EK_API int ek_hip_bucketed_pair_create_masked(int type, int index_type, int op, const void *table_a, const void *table_c,
                                              size_t table_size, const void *x, const void *index, const uint8_t *mask, size_t n,
                                              unsigned hints, ek_hip_bucketed **out);
EK_API int ek_hip_bucketed_reduce(ek_hip_bucketed *b, int reduce_op, int map_op, void *out, int keep_values, int keep_op);
EK_API int ek_hip_bucketed_scatter_add(ek_hip_bucketed *b, int count, void *const *bases, const int *from_u, const int *map_ops,
                                       const uint64_t *imm_bits, const int *weighted, const int *fresh);
/* scatter_add with a host scalar factor per stream: v_c = scale_c * (from_u[c] ? map_ops[c](u) : imm_c) -- what the tape sends
 * down for backward(c * y), for the -sin(u) of d/du cos(u), ...: the product of an unevaluated function of u with a host scalar
 * (enoki/hip.h: HIPArray::scaled_map_).  scale_bits = NULL: all ones.  Sums that the reduce call of a hinted object formed
 * early are folded with the factor. */
EK_API int ek_hip_bucketed_scatter_add_scaled(ek_hip_bucketed *b, int count, void *const *bases, const int *from_u, const int *map_ops,
                                              const uint64_t *imm_bits, const int *weighted, const int *fresh, const uint64_t *scale_bits);
/* != 0: a hinted object sums keep_op(u) and x * keep_op(u) per table entry inside reduce(EK_HSUM, map_op, keep, keep_op):
 * {sin, cos}, {cos, sin}, {log, rcp} and {f, f} for f in neg abs sqrt rcp rsqrt sin cos exp log */
EK_API int ek_hip_bucketed_early_pair(int map_op, int keep_op);
EK_API int ek_hip_bucketed_destroy(ek_hip_bucketed *b);
/* ---- multi-GPU without python / torch (csrc/dist.cpp) -------------------------------------------------------------------------
 * One process per GPU.  Index-range sharding: rank r owns [r n / P, (r + 1) n / P) of EVERY size-n array (ek_hip_dist_shard_range),
 * so all vertical operations are local; size-1 arrays and gather tables are replicated.  The exchange steps -- a horizontal result,
 * the gradient of a replicated table -- are RCCL collectives on the library stream, ordered with the kernels (no host wait):
 *   rank 0: ek_hip_dist_unique_id(id);  ship the 128 bytes to the other ranks;  every rank: ek_hip_dist_init(rank, world, id)
 *   y = hsum over all shards:   ek_hip_reduce(EK_HSUM, ...local shard...) then ek_hip_dist_all_reduce(type, EK_HSUM, y, 1)
 *   table gradients:            ek_hip_dist_all_reduce(type, EK_HSUM, g, K), or ek_hip_dist_reduce_scatter (rank r receives bins
 *                               [r c, (r + 1) c) of the sum; send holds world * c entries) + ek_hip_dist_all_gather when needed
 * reduce_op: EK_HSUM | EK_HPROD | EK_HMIN | EK_HMAX.  world == 1 with a NULL id needs no RCCL at all (collectives are local).
 * librccl.so is loaded on first use (no link-time dependency) and ONE copy per process: ENOKI_HIP_RCCL_PATH when set, else the copy
 * that is already mapped (a python caller has torch's own librccl.so: the collectives of torch.distributed and these then share
 * one RCCL runtime), else the loader's search path; ek_hip_dist_rccl_path() says which file it was.  The collectives refuse to run
 * while a step graph is being captured (EK_ERR_UNSUPPORTED, the capture stays valid) and are no-ops for n == 0.  The reference
 * has no counterpart; python callers use enoki_amd.dist (torch.distributed) for the same exchange. */
EK_API int ek_hip_dist_unique_id(void *id128);
EK_API int ek_hip_dist_init(int rank, int world, const void *id128);
EK_API int ek_hip_dist_world(int *rank, int *world);
EK_API int ek_hip_dist_shard_range(size_t n, int rank, int world, size_t *begin, size_t *end);
EK_API int ek_hip_dist_all_reduce(int type, int reduce_op, void *buf, size_t n);
EK_API int ek_hip_dist_reduce_scatter(int type, int reduce_op, void *recv, const void *send, size_t recv_count);
EK_API int ek_hip_dist_all_gather(int type, void *recv, const void *send, size_t send_count);
EK_API int ek_hip_dist_finalize(void);
EK_API const char *ek_hip_dist_rccl_path(void);
/* Partition of an INDEX array by bucket of the range it points into: the active entries (mask) of `index` are grouped by
 * bucket = index >> shift and stored as bucket-local indices (index & ((1 << shift) - 1)), bucket b at local[bucket_base[b] ..
 * bucket_base[b + 1]).  shift is the smallest of {12, 14, 17, 19} with <= 256 buckets for `range` entries (range <= 128 Mi).
 * This is what a program needs whose gather and scatter go through the SAME index array and whose body depends on the
 * gathered values only (tests/sphere.cpp:58-83 behind a pixel permutation): it can then run per TARGET entry, bucket by
 * bucket, with coalesced reads of the source slice and coalesced writes of the target slice instead of two random accesses
 * per element (enoki/vectorize_indexed.h builds that kernel from a user functor).  count + scan + partition: 5 + 9 B per
 * entry with a mask array.  The pointers in the info struct are DEVICE pointers owned by the object. */
typedef struct ek_hip_index_partition ek_hip_index_partition;
typedef struct {
    int shift, n_buckets;
    size_t n, range;
    const uint32_t *bucket_base;      /* n_buckets + 1 entries; bucket_base[n_buckets] = number of active entries */
    const uint32_t *local;            /* bucket-local indices in bucket order */
    /* round 6 -- page_shift != 0: the PAGED layout (single-pass partition, csrc/ek_paged.h; large inputs).  `local` then holds
       pages of 2^page_shift bucket-local indices; bucket b owns the complete pages pages_full[bucket_base[b] .. bucket_base[b + 1])
       and the partially filled ones pages_part[part_base[b] .. part_base[b + 1]) (entry = page << 6 | count - 1).  The order of
       the elements inside a bucket is unspecified.  page_shift == 0: one contiguous run per bucket as before. */
    int page_shift;
    const uint32_t *pages_full, *pages_part, *part_base;
} ek_hip_index_partition_info;
EK_API int ek_hip_index_partition_create(int index_type, const void *index, const ek_operand *mask, size_t n, size_t range,
                                         ek_hip_index_partition **out);
EK_API int ek_hip_index_partition_get(const ek_hip_index_partition *p, ek_hip_index_partition_info *info);
EK_API int ek_hip_index_partition_destroy(ek_hip_index_partition *p);
EK_API int ek_hip_scatter(int type, int index_type, void *base, const ek_operand *value,
                          const ek_operand *index, const ek_operand *mask, size_t n);
/* mode 0: fastest -- LDS-binned accumulation for large inputs (needs `base_size`), hardware atomics otherwise;
           the order of fp additions is unspecified, like the reference's atom.global.add;
   mode 1: deterministic -- stable radix sort by index + sequential per-bin sums: bit-identical to the CPU
           reference's element-order accumulation (dynamic.h:517-534).  Needs `base_size` and an
           index array; under a mask ARRAY it synchronizes once (the number of active pairs is read back), otherwise not
           at all.  Integer types are exact in either mode.
   64-bit index arrays of 256 Ki+ elements into a table of known size <= 2^32 are narrowed once (the reference's tape
   records gather offsets as Int64, autodiff.cpp:355-366) and then take the same paths as 32-bit ones. */
EK_API int ek_hip_scatter_add(int type, int index_type, void *base, size_t base_size,
                              const ek_operand *value, const ek_operand *index,
                              const ek_operand *mask, size_t n, int mode);
/* `count` (1..4) scatter_adds through ONE index / mask array into `count` tables of `base_size` entries each:
 *     bases[c][index[i]] += weights && weights[c] ? safe_mul(*weights[c], *values[c])[i] : (*values[c])[i]
 * This is what Tape::backward() issues for gathers that share their index array (the adjoint of gather is
 * scatter_add, autodiff.cpp:362-381); a weight is the pending edge product w * g of the sweep (autodiff.cpp:871-876,
 * safe_mul 1191-1199), fused into the read instead of being materialised first.  The indices are read and binned
 * once for all tables.  Same modes and accumulation-order caveats as ek_hip_scatter_add; inputs the fused path does
 * not cover are executed as `count` separate scatter_adds (same results). */
EK_API int ek_hip_scatter_add_multi(int type, int index_type, int count, void *const *bases, size_t base_size,
                                    const ek_operand *const *values, const ek_operand *const *weights,
                                    const ek_operand *index, const ek_operand *mask, size_t n, int mode);

/* ek_hip_scatter_add_multi with a unary operation applied to value stream c while it is loaded:
 *     bases[c][index[i]] += w_c[i] * map_c(values[c][i]),   map_c = value_ops[c]  (EK_COPY: none; value_ops == NULL: none)
 * Ops as for ek_hip_reduce_map, floating point types.  This is the adjoint of `gather` when the incoming gradient is
 * itself an unevaluated unary result -- the cos(u) of d/du sin(u) in `hsum(sin(gather(A, i) * x + gather(B, i)))` --
 * so the derivative array is never written or re-read (HIPArray defers fusable unary ops until their first consumer).
 * Paths without an on-load map (deterministic mode, tables of <= 16 Ki bins, 64-bit indices) evaluate the op first. */
EK_API int ek_hip_scatter_add_multi_map(int type, int index_type, int count, void *const *bases, size_t base_size,
                                        const ek_operand *const *values, const int *value_ops,
                                        const ek_operand *const *weights, const ek_operand *index, const ek_operand *mask,
                                        size_t n, int mode);

/* ---------------------------------------------------------------------------------------------
 *  Horizontal ops.  Results of ek_hip_reduce* stay on the device (`out` = 1 element, async);
 *  mask reductions return to the host and therefore synchronize (cuda.h:761-794).
 * ------------------------------------------------------------------------------------------- */
/* Diagnostics of the bucket partition's load balancing.  The single-pass page partition runs one workgroup per compute unit and
 * deals the input to the eight workgroup CLASSES w % 8 (the XCD a workgroup is dispatched to) in proportion to weights that the
 * library feeds back on the device from the classes' loop durations (OPT-IN: tuning "xcd_balance" / ENOKI_HIP_XCD_BALANCE = 1; 2 records
 * the durations of equal chunks without dealing by them; 0, the default, does neither and this call reports zeros; Q16: 65536 =
 * an equal share).  weights8 / ticks8 receive the current weights and the mean loop duration per class of the last stamped launch
 * in 100 MHz ticks (zeros before the first), *dealt (may be NULL) whether the weights are currently applied -- they are once a class
 * is more than 4 % away from an equal share, until all are back within 2 %; the call synchronises. */
EK_API int ek_hip_partition_class_state(uint32_t *weights8, uint32_t *ticks8, uint32_t *dealt);
EK_API int ek_hip_reduce(int op, int type, void *out, const void *in, size_t n);
/* A CHAIN: base(src[0 .. arity)) under n_maps unary ops (map_ops[0] first), evaluated in ONE pass over the operands -- the fused
 * kernel that the reference's JIT assembles for the vertical ops between two evaluation points (src/cuda/jit.cu:1066-1217,
 * :1418-1508), here as a descriptor interpreted by a pre-compiled kernel (a wave-uniform switch per stage around the loop over a
 * lane's 16-byte vector; no kernel per combination).  arity 1: the source itself; arity 2: base_op = EK_ADD | EK_SUB | EK_MUL;
 * arity 3: the fma family and EK_MULADD / EK_MULSUB / EK_NMULADD.  map_ops: the ops ek_hip_reduce_map accepts.  Operands are
 * arrays of n elements, arrays of one element or immediates.  Values are bit-identical to the op-by-op evaluation
 * (ek_hip_ternary, then ek_hip_unary per map); ek_hip_reduce_chain reduces them (order unspecified, like ek_hip_reduce),
 * ek_hip_map_chain writes them.  hsum(sin(exp(fmadd(a, x, b)))) then moves 12 B/elt instead of 28. */
typedef struct {
    int arity, base_op;
    ek_operand src[3];
    int n_maps;
    int map_ops[3];
} ek_chain;
EK_API int ek_hip_reduce_chain(int reduce_op, int type, void *out, const ek_chain *chain, size_t n);
EK_API int ek_hip_map_chain(int type, void *out, const ek_chain *chain, size_t n);
/* ek_hip_map_chain with a tail -- a factor and a second output:
 *     out[i]  = chain(i) * scale           scale: an immediate (ptr == NULL), or NULL for none; one more rounding, like the eager product
 *     out2[i] = op2(w[i], out[i])          op2 = EK_MUL | EK_SAFE_MUL, w an array of n elements
 * Either of out / out2 may be NULL (not both; out2 == NULL: op2 and w are ignored).  This is what the sweep of Tape::backward()
 * asks for at u = fmadd(a, x, b) under y = hsum(sin(u)) (src/autodiff/autodiff.cpp:838-988: grad_b = 1 * grad_u,
 * grad_a = x * grad_u with grad_u = cos(u)): a, x, b are read once and both gradients written, 20 B/elt, where the op-by-op
 * evaluation writes cos(u), re-reads it and multiplies (24 B/elt after a stored u).  Bits as op by op. */
EK_API int ek_hip_map_chain_product(int type, void *out, void *out2, const ek_chain *chain, const ek_operand *scale, int op2,
                                    const ek_operand *w, size_t n);
/* op(map_op(in[0..n))) in ONE pass over `in`: the unary operation is applied while loading (hsum(sin(x)): 4 B/element
 * instead of 12).  map_op: EK_NEG, EK_ABS, EK_SQRT, EK_RCP, EK_RSQRT, EK_SIN, EK_COS, EK_EXP, EK_LOG; floating point
 * types; n >= 1.  Same element values and the same reduction tree as ek_hip_unary followed by ek_hip_reduce (bit-identical
 * results).  The reference reaches this through its trace: `hsum(sin(x))` is one PTX kernel + the CUB reduction
 * (jit.cu:560-760, horiz.cu:162-268); HIPArray defers fusable unary ops until their first consumer. */
EK_API int ek_hip_reduce_map(int op, int map_op, int type, void *out, const void *in, size_t n);
/* fused backward edge to a scalar source: out[0] = hsum(safe_mul(w, g))  (autodiff.cpp:867-871) */
EK_API int ek_hip_hsum_safe_mul(int type, void *out, const ek_operand *w, const ek_operand *g, size_t n);
EK_API int ek_hip_mask_reduce(int op, const uint8_t *mask, size_t n, uint64_t *host_result);
/* inclusive prefix sum (cuda.h:717-726 / horiz.cu:182-200) */
EK_API int ek_hip_psum(int type, void *out, const void *in, size_t n);

/* Stable sort of (key, element number) pairs by the low `key_bits` bits of 32-bit keys:
 *     keys_out = keys sorted ascending, perm_out[j] = original position of keys_out[j]; equal keys keep their order.
 * The building block of partition() -- the reference sorts (pointer, lane) pairs with a 64-bit CUB radix sort and run-length
 * encodes them (cuda_partition, src/cuda/horiz.cu:35-122); callers here first map the pointers to dense 32-bit keys.
 * ceil(key_bits / 8) ballot-ranked LSD passes, 20 B per entry and pass. */
EK_API int ek_hip_sort_pairs(int key_bits, const uint32_t *keys, size_t n, uint32_t *keys_out, uint32_t *perm_out);

/* PCG32 draw (include/enoki/random.h:68-133): one fused kernel that advances `state` by `inc` where `mask`
 * is set (state_out[i] = mask[i] ? state[i] * 0x5851f42d4c957f2d + inc[i] : state[i]) and writes the output
 * function of the OLD state: u32[n], f32[n] in [0,1), f64[n] in [0,1), or u64[n] (two steps; the FIRST draw is the high
 * word, which is what the pinned g++ reference build produces for random.h:87-89). */
typedef enum { EK_PCG32_UINT32 = 0, EK_PCG32_FLOAT32, EK_PCG32_UINT64, EK_PCG32_FLOAT64 } ek_pcg32_kind;
EK_API int ek_hip_pcg32_next(int kind, void *out, uint64_t *state_out, const ek_operand *state,
                             const ek_operand *inc, const ek_operand *mask, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* ENOKI_HIP_C_API_H */
