// Internal helpers shared by the translation units of libenoki-hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "../../include/enoki_hip.h"

namespace ek {

// ---- error reporting -------------------------------------------------------------------------
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_fail(hipError_t err, const char *what, const char *file, int line);

#define EK_HIP_CHECK(expr)                                                                     \
    do {                                                                                       \
        hipError_t ek_err_ = (expr);                                                           \
        if (ek_err_ != hipSuccess) return ::ek::hip_fail(ek_err_, #expr, __FILE__, __LINE__);  \
    } while (0)

// ---- runtime state ---------------------------------------------------------------------------
struct Tuning {
    int blocks_per_cu = 8;   // grid cap for streaming kernels = blocks_per_cu * #CU
    int reduce_blocks_per_cu = 4;
    int scatter_add_binned = 1;   // 1: LDS-binned scatter_add for large inputs, 0: global atomics only
    int deterministic = 0;        // 1: fp scatter_add always takes the bit-reproducible sorted path (ENOKI_HIP_DETERMINISTIC)
    int gather_records = 1;       // struct gathers through staged {x, y, ..} records: 1 by size, 2 always, 0 never
    int bucket_ordered = 1;       // gather -> fma -> {reduction, scatter_add} chains in bucket order (bucketed.hip); 0: element order only
    int early_adjoint = 1;        // EK_BUCKETED_HINT_ADJOINT is honoured (half-size buckets, adjoint sums formed in the forward pass)
    int xcd_balance = 0;          // 1: the page partition deals its tiles to the workgroup classes w % 8 by fed-back weights (ek_paged.h);
                                  // 2: equal chunks, but the classes' loop durations are recorded (ek_hip_partition_class_state)
};

struct Context {
    bool initialized = false;
    int device = -1;
    int num_cu = 256;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    uint32_t log_level = 0;
    uint64_t launches = 0;
    Tuning tuning;
    void *reduce_scratch = nullptr;   // partials of two-stage reductions
    size_t reduce_scratch_bytes = 0;
    bool profiling = false;           // ek_hip_profile_begin(): one event after every launch
};

Context &ctx();
int ensure_init();
int reduce_scratch(size_t bytes, void **out);

void profile_mark(const char *name, size_t n, size_t bytes);

/// `bytes`: ALGORITHMIC bytes of this launch (distinct input bytes + output bytes), see DESIGN.md
inline void note_launch(const char *name, size_t n, size_t bytes) {
    Context &c = ctx();
    c.launches++;
    if (c.log_level >= 3)
        fprintf(stderr, "enoki-hip: launch %s (n=%zu, %zu bytes)\n", name, n, bytes);
    if (c.profiling) profile_mark(name, n, bytes);
}

// ---- roctx ranges ------------------------------------------------------------------------------
// One range per kernel FAMILY (the multi-launch pipelines: scatter_add, reductions, prefix sums, fused gathers), so a
// rocprofv3 --marker-trace timeline groups the 5-7 launches of e.g. one scatter_add under one bar.  The roctx library
// (librocprofiler-sdk-roctx.so) is loaded lazily and only when ENOKI_HIP_ROCTX=1: no link dependency, no cost otherwise.
void roctx_push(const char *name);
void roctx_pop();
struct RoctxRange {
    explicit RoctxRange(const char *name) { roctx_push(name); }
    ~RoctxRange() { roctx_pop(); }
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
};

// post-launch check: kernel launch failures surface through hipGetLastError
#define EK_LAUNCH_CHECK(name, n, bytes)                                                        \
    do {                                                                                       \
        ::ek::note_launch(name, n, bytes);                                                     \
        hipError_t ek_err_ = hipGetLastError();                                                \
        if (ek_err_ != hipSuccess) return ::ek::hip_fail(ek_err_, name, __FILE__, __LINE__);   \
    } while (0)

// ---- type mapping ------------------------------------------------------------------------------
template <int Type> struct ctype;
template <> struct ctype<EK_BOOL> { using type = uint8_t; };
template <> struct ctype<EK_I32> { using type = int32_t; };
template <> struct ctype<EK_U32> { using type = uint32_t; };
template <> struct ctype<EK_I64> { using type = int64_t; };
template <> struct ctype<EK_U64> { using type = uint64_t; };
template <> struct ctype<EK_F32> { using type = float; };
template <> struct ctype<EK_F64> { using type = double; };

inline size_t type_size(int type) {
    switch (type) {
        case EK_BOOL: return 1;
        case EK_I32: case EK_U32: case EK_F32: return 4;
        case EK_I64: case EK_U64: case EK_F64: return 8;
        default: return 0;
    }
}

// arithmetic type whose overflow wraps: unsigned counterpart for integers, T itself for fp
template <typename T, bool = std::is_floating_point_v<T>> struct wrap_type { using type = T; };
template <typename T> struct wrap_type<T, false> { using type = std::make_unsigned_t<T>; };
template <typename T> using wrap_t = typename wrap_type<T>::type;

// Kernel-side view of an ek_operand
template <typename T> struct Arg {
    const T *ptr;   // device pointer or nullptr
    T imm;          // immediate when ptr == nullptr
    uint32_t vec;   // 1: one element per index, 0: broadcast
};

template <typename T> inline int make_arg(const ek_operand *o, size_t n, Arg<T> &out, const char *what) {
    if (!o) return fail(EK_ERR_INVALID, "%s: null operand", what);
    if (o->ptr == nullptr) {
        T v;
        memcpy(&v, &o->imm, sizeof(T));
        out = Arg<T>{ nullptr, v, 0u };
        return EK_OK;
    }
    if (o->size != n && o->size != 1)
        return fail(EK_ERR_INVALID, "%s: arrays of incompatible size (%zu vs %zu)", what, o->size, n);
    out = Arg<T>{ (const T *) o->ptr, T(0), o->size != 1 ? 1u : 0u };
    return EK_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
template <typename T> inline bool arg_aligned(const Arg<T> &a) { return !a.vec || aligned16(a.ptr); }
/// bytes this operand contributes per launch of n elements (broadcast operands are free)
template <typename T> inline size_t arg_bytes(const Arg<T> &a, size_t n) { return a.vec ? n * sizeof(T) : 0; }

// Grid for a streaming kernel that handles `work_items` per-thread items with 256-thread blocks
inline unsigned stream_grid(size_t work_items, int blocks_per_cu) {
    Context &c = ctx();
    size_t blocks = (work_items + 255) / 256;
    size_t cap = (size_t) c.num_cu * (size_t) blocks_per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (unsigned) blocks;
}

// LDS-binned scatter_add (scatter_binned.hip)
bool scatter_add_binned_applicable(size_t table_size, size_t n, bool index_is_array, size_t elem_size = 4);
template <typename T, typename I>
int scatter_add_binned(T *base, size_t table_size, const Arg<T> &value, const Arg<I> &index, const Arg<uint8_t> &mask,
                       size_t n);

// A step graph is being captured on the library stream (ek_hip_graph_begin): anything that makes the HOST wait for the
// device -- a read-back, a synchronisation -- cannot be recorded and would invalidate the capture.  Such entry points call
// this first and fail cleanly (the capture stays valid, the caller ends it and runs the step eagerly).
int refuse_while_capturing(const char *what);
int refuse_while_capturing_quiet();        // EK_OK when no capture is in progress; sets no error message
void release_meta_ring();                  // bucketed.hip: the context's counter blocks go back to the allocator (device switch)

// The unary ops that a consumer may apply on load (HIPArray defers exactly these: include/enoki/hip.h)
constexpr inline bool unary_fusable(int op) {
    switch (op) {
        case EK_NEG: case EK_ABS: case EK_SQRT: case EK_RCP: case EK_RSQRT: case EK_SIN: case EK_COS: case EK_EXP: case EK_LOG:
        case EK_RCP_SQR: case EK_RSQRT_SQR: case EK_RSQRT_CUBE:
            return true;
        default: return false;
    }
}

// ... and what a CHAIN may carry on top of those (ek_hip_reduce_chain / ek_hip_map_chain / ek_hip_reduce_map, round 6): the second-wave
// functions whose derivative is one map of the same argument, and those derivative maps
constexpr inline bool unary_chainable(int op) {
    switch (op) {
        case EK_TAN: case EK_TANH: case EK_ATAN: case EK_SINH: case EK_COSH: case EK_SEC_SQR: case EK_SECH_SQR: case EK_RCP_1P_SQR:
            return true;
        default: return unary_fusable(op);
    }
}

// one f32 value stream through the single-pass page partition (bucketed.hip): 20 B per pair instead of 26
bool scatter_add_paged_applicable(size_t table_size, size_t n);
int scatter_add_paged(float *base, size_t table_size, const float *value, const uint32_t *index, const Arg<uint8_t> &mask, size_t n);

bool scatter_add_binned_multi_applicable(size_t table_size, size_t n, bool index_is_array, size_t elem_size = 4);
template <typename T, typename I, int C>
int scatter_add_binned_multi(T *const *bases, size_t table_size, const Arg<T> *values, const Arg<T> *weights, unsigned weighted,
                             const Arg<I> &index, const Arg<uint8_t> &mask, size_t n, const int *value_ops = nullptr);

// deterministic scatter_add: stable radix sort by index + sequential per-bin sums (scatter_binned.hip)
template <typename T, typename I>
int scatter_add_sorted(T *base, size_t table_size, const Arg<T> &value, const Arg<I> &index, const Arg<uint8_t> &mask,
                       size_t n);

template <typename T, typename I, int C>
int scatter_add_sorted_multi(T *const *bases, size_t table_size, const Arg<T> *values, const Arg<T> *weights, unsigned weighted,
                             const Arg<I> &index, const Arg<uint8_t> &mask, size_t n);

// stable LSD radix sort of (key, element number) pairs by the low `key_bits` bits (scatter_binned.hip)
int sort_pairs_u32(int key_bits, const uint32_t *keys, size_t n, uint32_t *keys_out, uint32_t *perm_out);

} // namespace ek
