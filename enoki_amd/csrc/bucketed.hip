// Bucket-ordered evaluation of  u = fma(gather(A, idx), x, gather(C, idx))  and of what consumes it.
//
// What this replaces.  The reference's JIT emits a gather into the kernel of its consumer and fuses the whole chain
// up to the next horizontal operation into ONE kernel (src/cuda/jit.cu:984 cuda_jit_assemble, :1066-1217 the gather's
// ld.global inside the consumer, :1418-1471 cuda_eval); the adjoint scatter_add goes through atom.global.add
// (cuda.h:892-905).  An eager kernel that does the same lookups in ELEMENT order is bound by the L2 miss rate of the
// parameter tables, not by HBM: for K = 1 Mi entries the interleaved {A, C} table is 8 MiB, twice the 4 MiB L2 of an
// XCD, 54 % of the lookups miss and the kernel stops at 0.25 of the HBM roofline (profiles/rocprof_l2_r02.txt).
//
// What it does instead.  When the consumer of u does not care about the element order -- a horizontal reduction
// (hsum(sin(u))), and the adjoint scatter_add of the two gathers through the SAME index array -- the elements are
// processed BUCKET BY BUCKET:
//
//   1. count / scan / partition  (ek_binned.h, the scatter_add pipeline's own kernels) sorts (idx, x) by bucket of
//      16 Ki table entries (8 Ki for 8-byte types): idx is read twice (count, partition), x once; written: a 16-bit
//      bucket-local index and x in bucket order                                                4 + 8 + 6 B/elt
//   2. forward   one 1024-thread workgroup per piece of a bucket stages the bucket's {A, C} slice in LDS (128 KiB),
//                streams (l16, x_b), computes u = fma(A[l], x, C[l]) from LDS, reduces map(u) and -- when somebody
//                else still holds u -- keeps u in bucket order                                     6 (+ 4) B/elt
//   3. adjoint   Tape::backward() finds the partition on the node: one workgroup per piece streams (l16, u_b, x_b),
//                evaluates the value streams (cos(u_b), safe_mul(x_b, cos(u_b))) and adds them into LDS tables under
//                the exchange lock of the scatter_add pipeline; partial tables are folded            10 B/elt
//
// No lookup leaves the CU, no second count / scan / partition in the backward.  Every access that needs ELEMENT order
// (data(), an elementwise consumer, another index array) takes the element-order kernels (gathered.hip) -- same bits
// for u; reductions and gradients differ from the element-order path only by the order of their fp additions (parity
// class D, like every reduction / scatter_add of this library).  Deterministic mode never comes here.
#include "ek_bucketed.h"

namespace ek {

// ---- 2. forward ------------------------------------------------------------------------------------
// The streaming part, specialised for the unary op that is applied to u before the reduction (Map, compile time: the
// kernel switches ONCE, outside the loops -- a runtime switch per element would inline nine transcendental bodies into an
// eight-fold unrolled loop and blow the instruction cache).
// KeepPartner (Map = sin or cos only): what is kept in bucket order is not u but the OTHER half of sincos(u) -- one sincos
// evaluation yields the reduced half and the kept half (the cos(u) that the adjoint of sin needs), like the stand-alone
// sincos kernel fills both halves of a linked pair.
// FromKept: u is not formed from the table slice but read back from what an earlier pass kept (the lists have holes in the
// paged layout, so such a reduction walks the pages too).
template <typename T, int ROp, int V, int Map, bool KeepPartner, bool FromKept>
struct ForwardBody {
    using R = BucketReducer<ROp, T>;
    struct Step { Pack<uint16_t, 4> pi[V]; T px[V][4]; size_t pos[V]; };
    const PairRec<T> *rec;
    T *u_out;
    const uint16_t *pair_idx;
    const T *x_b, *kept;
    uint32_t lmask;
    int two;                 // != 0: u = a * x + c with a rounding each (EK_MULADD family) instead of one fma
    T acc[4];

    __device__ __forceinline__ T elem(uint32_t l, T x, int slot) {
        T u;
        if constexpr (FromKept) {
            u = x;
        } else {
            const PairRec<T> r = rec[l & lmask];
            u = pair_value(r.a, x, r.c, two);
        }
        if constexpr (KeepPartner) {
            static_assert(Map == EK_SIN || Map == EK_COS);
            T sn, cs;
            SinCosOp::apply(u, sn, cs);
            acc[slot] = R::combine(acc[slot], Map == EK_SIN ? sn : cs);
            return Map == EK_SIN ? cs : sn;
        } else {
            if constexpr (ROp != EK_REDUCE_NONE) acc[slot] = R::combine(acc[slot], UnaryOp<Map, T>::apply(u));
            return u;
        }
    }
    __device__ __forceinline__ void fetch(Step &s, int h, size_t pos) {
        s.pos[h] = pos;
        if constexpr (FromKept) {
            load4<T, true>(kept + pos, s.px[h]);
        } else {
            s.pi[h] = pack_load<uint16_t, 4, true>(pair_idx + pos);
            load4<T, true>(x_b + pos, s.px[h]);
        }
    }
    __device__ __forceinline__ void apply(const Step &s) {
#pragma unroll
        for (int h = 0; h < V; ++h) {
            T u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = elem(FromKept ? 0u : (uint32_t) s.pi[h].v[j], s.px[h][j], j);
            if (u_out) {
                if constexpr (sizeof(T) == 4) {
                    Pack<T, 4> po;
#pragma unroll
                    for (int j = 0; j < 4; ++j) po.v[j] = u[j];
                    pack_store<T, 4, true>(u_out + s.pos[h], po);
                } else {
                    Pack<T, 2> p0, p1;
                    p0.v[0] = u[0]; p0.v[1] = u[1]; p1.v[0] = u[2]; p1.v[1] = u[3];
                    pack_store<T, 2, true>(u_out + s.pos[h], p0);
                    pack_store<T, 2, true>(u_out + s.pos[h] + 2, p1);
                }
            }
        }
    }
    __device__ __forceinline__ void one(size_t pos, bool on, int slot) {
        if (on) {
            const T u = FromKept ? elem(0u, kept[pos], slot) : elem(pair_idx[pos], x_b[pos], slot);
            if (u_out) u_out[pos] = u;
        }
    }
};

// flip_a / flip_c: the fma family differs by the signs of its first and third operand (fmsub: -c, fnmadd: -a, fnmsub:
// both); the signs are applied ONCE to the staged table entries -- exact -- and the inner loop is always one fma.
template <typename T, int ROp, int V, int PS, bool FromKept = false>
__global__ __launch_bounds__(kBucketThreads) void k_bucket_pair_forward(T *__restrict__ partials, T *__restrict__ u_out,
                                                                        const T *__restrict__ table_a, const T *__restrict__ table_c,
                                                                        size_t table_size, int flip_a, int flip_c, int two,
                                                                        const uint16_t *__restrict__ pair_idx,
                                                                        const T *__restrict__ x_b, const T *__restrict__ kept,
                                                                        BucketLists bl, int map_op, int keep_partner, int shift,
                                                                        BucketFinish<T> fin) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    PairRec<T> *rec = reinterpret_cast<PairRec<T> *>(lds_raw);
    __shared__ T wave_part[kBucketWaves];
    using R = BucketReducer<ROp, T>;
    const int Bins = 1 << shift;
    const uint32_t lmask = (uint32_t) Bins - 1u;
    int bucket;
    PieceRange range;
    const bool live = bucket_piece<PS>(bl, bucket, range);        // (workgroup-uniform; a launch has a few more workgroups than pieces)
    if (!live) {
        if constexpr (ROp != EK_REDUCE_NONE) bucket_finish<T, ROp>(R::identity(), partials, fin.ticket, fin.out, fin.active, fin.n, fin.zero_op, wave_part, fin.counters);
        return;
    }
    if constexpr (!FromKept) {
        stage_pair_slice<T, false>(rec, nullptr, table_a, table_c, (size_t) bucket * Bins, table_size, Bins, flip_a, flip_c);
        __syncthreads();
    }

    T result = R::identity();
    auto run = [&](auto body) {
        body.rec = rec; body.u_out = u_out; body.pair_idx = pair_idx; body.x_b = x_b; body.kept = kept; body.lmask = lmask; body.two = two;
#pragma unroll
        for (int k = 0; k < 4; ++k) body.acc[k] = R::identity();
        walk_piece<PS, V>(bl, range, body);
        result = R::combine(R::combine(body.acc[0], body.acc[1]), R::combine(body.acc[2], body.acc[3]));
    };
#define EK_FWD_CASE(OP) case OP: run(ForwardBody<T, ROp, V, OP, false, FromKept>{}); break;
    if constexpr (ROp == EK_REDUCE_NONE) {
        run(ForwardBody<T, ROp, V, EK_COPY, false, FromKept>{});
    } else if (!FromKept && keep_partner && map_op == EK_SIN) {
        run(ForwardBody<T, ROp, V, EK_SIN, true, false>{});
    } else if (!FromKept && keep_partner && map_op == EK_COS) {
        run(ForwardBody<T, ROp, V, EK_COS, true, false>{});
    } else {
        switch (map_op) {
            EK_FWD_CASE(EK_NEG) EK_FWD_CASE(EK_ABS) EK_FWD_CASE(EK_SQRT) EK_FWD_CASE(EK_RCP) EK_FWD_CASE(EK_RSQRT)
            EK_FWD_CASE(EK_SIN) EK_FWD_CASE(EK_COS) EK_FWD_CASE(EK_EXP) EK_FWD_CASE(EK_LOG)
            EK_FWD_CASE(EK_RCP_SQR) EK_FWD_CASE(EK_RSQRT_SQR) EK_FWD_CASE(EK_RSQRT_CUBE)
            default: run(ForwardBody<T, ROp, V, EK_COPY, false, FromKept>{}); break;
        }
    }
#undef EK_FWD_CASE
    if constexpr (ROp != EK_REDUCE_NONE) {
        T v = result;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v = R::combine(v, bucket_shfl_down(v, d));
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) wave_part[wave] = v;
        __syncthreads();
        if (threadIdx.x < 64) {
            v = threadIdx.x < kBucketWaves ? wave_part[threadIdx.x] : R::identity();
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) v = R::combine(v, bucket_shfl_down(v, d));
        }
        bucket_finish<T, ROp>(v, partials, fin.ticket, fin.out, fin.active, fin.n, fin.zero_op, wave_part, fin.counters);
    }
}


// ---- 3. adjoint ------------------------------------------------------------------------------------
// Value stream c of the scatter_add:  v_c = from_u(c) ? map_c(u) : imm_c,  times x with safe_mul semantics when weighted(c)
// (the tape's pending edge product w * g, autodiff.cpp:1191-1199) -- added to table c at the element's index.
template <typename T, int C> struct BucketStreams {
    int map_op[C];
    T imm[C];
    T scale[C];           // host scalar factor on the stream's value (before the weight)
    unsigned from_u, weighted;
    unsigned plain_x = 0;  // C == 1: the stream is x itself (scatter_add_paged)
};
// The streaming part.  Map >= 0: every stream that is a function of u applies THIS op (compile time; evaluated once per
// element however many streams share it -- the usual pair cos(u), x * cos(u)); Map < 0: per-stream ops chosen at run time.
// Spec = 1: the adjoint of a gathered pair as the tape issues it -- two streams, both the (kept / mapped) function of u, the
// SECOND one weighted by x: known at compile time, no per-element selects on the stream description.
// Spec = 2: ONE stream that is the partitioned value x itself -- a plain scatter_add(value, index) whose (index, value) pairs went
// through the page partition (scatter_add_paged below).
template <typename T, int C, int V, int Map, int Spec>
struct AccumulateBody {
    static constexpr bool Paired = C == 2 && sizeof(T) == 4;
    static_assert(Spec == 0 || (Spec == 1 && C == 2 && Map >= 0) || (Spec == 2 && C == 1));
    struct Step { Pack<uint16_t, 4> pi[V]; T pu[V][4], px[V][4]; };
    T *acc;
    BucketStreams<T, C> st;
    const uint16_t *pair_idx;
    const T *u_b, *x_b;
    int Bins;
    bool need_u, need_x;

    __device__ __forceinline__ void values(T u, T x, T (&v)[C]) const {
        T m = T(0);
        if constexpr (Map >= 0) m = UnaryOp<Map, T>::apply(u);
        if constexpr (Spec == 2) {
            v[0] = x;
        } else if constexpr (Spec == 1) {
            v[0] = m * st.scale[0];
            v[C - 1] = dev::safe_mul(x, m * st.scale[C - 1]);
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if constexpr (Map >= 0) v[c] = ((st.from_u >> c) & 1u) ? m : st.imm[c];
                else v[c] = ((st.from_u >> c) & 1u) ? unary_fused<T>(st.map_op[c], u) : st.imm[c];
                v[c] = v[c] * st.scale[c];
                if ((st.weighted >> c) & 1u) v[c] = dev::safe_mul(x, v[c]);
            }
        }
    }
    // every lane of a wave passes through the lock (its retry loop is wave-uniform): inactive lanes add nothing
    __device__ __forceinline__ void one(size_t pos, bool on, int) {
        const uint32_t l = on ? (uint32_t) pair_idx[pos] : 0u;
        const T u = (on && need_u) ? u_b[pos] : T(0), x = (on && need_x) ? x_b[pos] : T(0);
        T v[C];
        values(u, x, v);
        if constexpr (Paired) {
            lds_add_pair(reinterpret_cast<unsigned long long *>(acc) + (l & (Bins - 1)), v[0], v[C - 1], on);
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) lds_add<true>(&acc[c * Bins + (l & (Bins - 1))], v[c], on);
        }
    }
    __device__ __forceinline__ void fetch(Step &s, int h, size_t pos) {
        s.pi[h] = pack_load<uint16_t, 4, true>(pair_idx + pos);
        if (need_u) load4<T, true>(u_b + pos, s.pu[h]);
        if (need_x) load4<T, true>(x_b + pos, s.px[h]);
    }
    __device__ __forceinline__ void apply(const Step &s) {
        // the values of all 4 V elements; two f32 tables: each element's pair is added right away, otherwise one batch of LDS
        // updates per table afterwards
        constexpr int NB = 4 * V;
        uint32_t l[NB];
        T v[C][NB];
#pragma unroll
        for (int h = 0; h < V; ++h) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = h * 4 + j;
                const T u = need_u ? s.pu[h][j] : T(0), x = need_x ? s.px[h][j] : T(0);
                l[k] = (uint32_t) s.pi[h].v[j] & (Bins - 1);
                T vk[C];
                values(u, x, vk);
#pragma unroll
                for (int c = 0; c < C; ++c) v[c][k] = vk[c];
                if constexpr (Paired) {
                    // one claim right behind its element (claim, add, release, batched retry).  Claiming all 4 V bins of
                    // a step first -- one LDS round trip instead of 4 V dependent ones -- was measured again once the
                    // kernels took the bucket size at run time and lost: 0.185 vs 0.152 ms for {cos(u), x cos(u)}, 0.138
                    // vs 0.108 ms for {1, x}, equal for the rest (same box, 64 Mi elements): a lock held across a batch
                    // is met by the other 15 waves more often than its round trip costs.
                    lds_add_pair_one(reinterpret_cast<unsigned long long *>(acc), l[k], vk[0], vk[C - 1]);
                }
            }
        }
        if constexpr (!Paired) {
#pragma unroll
            for (int c = 0; c < C; ++c) lds_add_batch<T, NB>(acc + c * Bins, l, v[c]);
        }
    }
};

template <typename T, int C, int V, int PS>
__global__ __launch_bounds__(kBucketThreads) void k_bucket_accumulate(T *__restrict__ partials,
                                                                      const uint16_t *__restrict__ pair_idx,
                                                                      const T *__restrict__ u_b, const T *__restrict__ x_b,
                                                                      BucketLists bl, BucketStreams<T, C> st, int shift) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T *acc = reinterpret_cast<T *>(lds_raw);                 // C tables of Bins entries; two f32 tables: Bins {t0, t1} pairs
    const int Bins = 1 << shift;
    constexpr bool Paired = C == 2 && sizeof(T) == 4;
    int bucket;
    PieceRange range;
    if (!bucket_piece<PS>(bl, bucket, range)) return;
    for (int j = threadIdx.x; j < C * Bins; j += kBucketThreads) acc[j] = T(0);
    __syncthreads();

    // one op for all streams that read u?  (streams that do not read u do not care)
    int op = EK_COPY;
    bool uniform = true, first = true;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (!((st.from_u >> c) & 1u)) continue;
        if (first) { op = st.map_op[c]; first = false; }
        else uniform = uniform && st.map_op[c] == op;
    }
    auto run = [&](auto body, bool spec) {
        body.acc = acc; body.st = st; body.pair_idx = pair_idx; body.u_b = u_b; body.x_b = x_b; body.Bins = Bins;
        body.need_u = spec || st.from_u != 0;
        body.need_x = spec || st.weighted != 0;
        walk_piece<PS, V>(bl, range, body);
    };
#define EK_ACC_CASE(OP) case OP: run(AccumulateBody<T, C, V, OP, 0>{}, false); break;
#define EK_ACC_SPEC(OP) case OP: run(AccumulateBody<T, C, V, OP, 1>{}, true); break;
    bool done = false;
    if constexpr (C == 1) {
        if (st.plain_x) {
            auto body = AccumulateBody<T, C, V, EK_COPY, 2>{};
            body.acc = acc; body.st = st; body.pair_idx = pair_idx; body.u_b = u_b; body.x_b = x_b; body.Bins = Bins;
            body.need_u = false; body.need_x = true;
            walk_piece<PS, V>(bl, range, body);
            done = true;
        }
    }
    if constexpr (C == 2) {
        if (uniform && st.from_u == 3u && st.weighted == 2u) {         // (host side: the weighted stream is put second)
            done = true;
            switch (op) {
                EK_ACC_SPEC(EK_NEG) EK_ACC_SPEC(EK_ABS) EK_ACC_SPEC(EK_SQRT) EK_ACC_SPEC(EK_RCP) EK_ACC_SPEC(EK_RSQRT)
                EK_ACC_SPEC(EK_SIN) EK_ACC_SPEC(EK_COS) EK_ACC_SPEC(EK_EXP) EK_ACC_SPEC(EK_LOG)
                EK_ACC_SPEC(EK_RCP_SQR) EK_ACC_SPEC(EK_RSQRT_SQR) EK_ACC_SPEC(EK_RSQRT_CUBE)
                case EK_COPY: run(AccumulateBody<T, C, V, EK_COPY, 1>{}, true); break;
                default: done = false; break;       // an op without a compile-time body: the per-stream run-time form below
            }
        }
    }
#undef EK_ACC_SPEC
    if (done) {
    } else if (uniform) {
        // (an op that unary_fusable() accepts but that has no case here must never fall into the EK_COPY body -- it would scatter u
        // instead of op(u): it takes the run-time form, which evaluates unary_fused(op, u) per stream)
        switch (op) {
            EK_ACC_CASE(EK_NEG) EK_ACC_CASE(EK_ABS) EK_ACC_CASE(EK_SQRT) EK_ACC_CASE(EK_RCP) EK_ACC_CASE(EK_RSQRT)
            EK_ACC_CASE(EK_SIN) EK_ACC_CASE(EK_COS) EK_ACC_CASE(EK_EXP) EK_ACC_CASE(EK_LOG)
            EK_ACC_CASE(EK_RCP_SQR) EK_ACC_CASE(EK_RSQRT_SQR) EK_ACC_CASE(EK_RSQRT_CUBE)
            case EK_COPY: run(AccumulateBody<T, C, V, EK_COPY, 0>{}, false); break;
            default: run(AccumulateBody<T, C, V, -1, 0>{}, false); break;
        }
    } else {
        run(AccumulateBody<T, C, V, -1, 0>{}, false);
    }
#undef EK_ACC_CASE
    __syncthreads();
    // one bucket-sized partial per piece and table: table c at partials + c * gridDim.x * Bins (k_bin_fold_pieces)
#pragma unroll
    for (int c = 0; c < C; ++c) {
        T *out = partials + ((size_t) c * gridDim.x + blockIdx.x) * Bins;
        for (int j = threadIdx.x; j < Bins; j += kBucketThreads) out[j] = Paired ? acc[2 * j + c] : acc[c * Bins + j];
    }
}

// ---- host side -------------------------------------------------------------------------------------
// Counter blocks of the paged objects, kept by the context and handed from object to object: the first reducing launch of an object
// leaves its block's counters zeroed (bucket_finish: the last workgroup clears what the directory launch has consumed), so the next
// object that takes the block needs no fill -- the 768-word memset in front of every partition was 4.5 us of a 100 us shard step.
// Four blocks cover a tape that keeps a few objects alive; whoever finds none free (the slices of a large table, a captured step,
// whose blocks must come from the graph's own pool) allocates and fills as before.
struct MetaRing {
    static constexpr int kBlocks = 4;
    static constexpr size_t kPartials = 2048;                   // reduce partials a block has room for (max_pieces)
    void *block[kBlocks] = {};
    bool busy[kBlocks] = {}, clean[kBlocks] = {};
    // (16 bytes per piece: a float partial, or -- the fixed-point forward + adjoint kernel -- a 64-bit integer partial and a float one)
    static size_t bytes() { return (kPgCounterWords + 3 * (kMaxBuckets + 1) + 1) * sizeof(uint32_t) + kPartials * 16 + 16; }
};
static MetaRing &meta_ring() { static MetaRing *r = new MetaRing(); return *r; }

// The class weights of the page partition (ek_paged.h: kPgClasses) and the workgroups' loop stamps: one small device block per
// context, weights initialised to 1.  ENOKI_HIP_XCD_BALANCE=0 (tuning "xcd_balance") deals equal chunks as before.
struct ClassState {
    void *block = nullptr;             // uint32: weights[kPgClasses] | "the weights are being dealt" | - | stamps[1024] from word kPgClassStamps
    uint32_t *weights() { return (uint32_t *) block; }
    uint32_t *stamps() { return (uint32_t *) block + kPgClassStamps; }
};
static ClassState &class_state() { static ClassState *s = new ClassState(); return *s; }
static uint32_t *class_weights_or_null() {
    ClassState &s = class_state();
    if (!s.block) {
        if (refuse_while_capturing_quiet() != EK_OK) return nullptr;            // (a copy from the host is not part of a step graph)
        if (ek_hip_malloc((kPgClassStamps + 1024) * sizeof(uint32_t), &s.block) != EK_OK) { s.block = nullptr; return nullptr; }
        static const uint32_t ones[kPgClasses] = { kPgWeightOne, kPgWeightOne, kPgWeightOne, kPgWeightOne,
                                                   kPgWeightOne, kPgWeightOne, kPgWeightOne, kPgWeightOne };
        if (hipMemsetAsync(s.block, 0, (kPgClassStamps + 1024) * sizeof(uint32_t), ctx().stream) != hipSuccess ||
            hipMemcpyAsync(s.block, ones, sizeof(ones), hipMemcpyHostToDevice, ctx().stream) != hipSuccess) {
            ek_hip_free(s.block);
            s.block = nullptr;
            return nullptr;
        }
    }
    return s.weights();
}

} // namespace ek

/* diagnostics: the weights of the page partition's workgroup classes and the mean loop duration per class of the last stamped
   launch (100 MHz ticks); synchronises.  weights8 / ticks8: 8 entries each, zeros when no launch has used the weights yet. */
extern "C" EK_API int ek_hip_partition_class_state(uint32_t *weights8, uint32_t *ticks8, uint32_t *dealt) {
    using namespace ek;
    if (int rc = ensure_init()) return rc;
    if (!weights8 || !ticks8) return fail(EK_ERR_INVALID, "ek_hip_partition_class_state(): null pointer");
    for (int k = 0; k < kPgClasses; ++k) weights8[k] = ticks8[k] = 0;
    if (dealt) *dealt = 0;
    ClassState &s = class_state();
    if (!s.block) return EK_OK;
    if (int busy = refuse_while_capturing("ek_hip_partition_class_state()")) return busy;
    std::vector<uint32_t> host(kPgClassStamps + 1024);
    EK_HIP_CHECK(hipMemcpyAsync(host.data(), s.block, host.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx().stream));
    EK_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    const unsigned W = (unsigned) ctx().num_cu < 1024u ? (unsigned) ctx().num_cu : 1024u;
    for (int k = 0; k < kPgClasses; ++k) {
        weights8[k] = host[k];
        uint64_t sum = 0, cnt = 0;
        for (unsigned w = k; w < W; w += kPgClasses) { sum += host[kPgClassStamps + w]; ++cnt; }
        ticks8[k] = cnt ? (uint32_t) (sum / cnt) : 0u;
    }
    if (dealt) *dealt = host[kPgClasses];
    return EK_OK;
}

namespace ek {

void release_meta_ring() {
    ClassState &cs = class_state();
    if (cs.block) { ek_hip_free(cs.block); cs.block = nullptr; }
    MetaRing &r = meta_ring();
    for (int k = 0; k < MetaRing::kBlocks; ++k) {
        if (r.block[k] && !r.busy[k]) { ek_hip_free(r.block[k]); r.block[k] = nullptr; r.clean[k] = false; }
    }
}

Bucketed::~Bucketed() {
    if (meta_slot >= 0) {
        MetaRing &r = meta_ring();
        r.busy[meta_slot] = false;
        r.clean[meta_slot] = meta_clean;
        meta = nullptr;
    }
    for (void *p : { meta, pair_idx, x_b, u_b, m_b, early, page_lists })
        if (p) ek_hip_free(p);
}

static size_t bucket_target_pieces(size_t n, int n_buckets) {
    Context &c = ctx();
    static const int per_cu = [] { const char *e = getenv("ENOKI_HIP_BUCKET_PIECES_PER_CU"); return e ? atoi(e) : 1; }();
    static const size_t piece_elems = [] { const char *e = getenv("ENOKI_HIP_PIECE_ELEMS"); return e ? (size_t) atol(e) : (size_t) 32768; }();
    return std::max<size_t>(std::min<size_t>((size_t) std::max(per_cu, 1) * (size_t) c.num_cu, n / piece_elems), (size_t) n_buckets);
}


template <typename T, typename I, int Shift>
static int bucketed_create(Bucketed *b, const T *x, const I *index) {
    RoctxRange range("enoki-hip: bucket partition");
    Context &c = ctx();
    constexpr int Bins = 1 << Shift;
    b->shift = Shift;
    const size_t n = b->n;
    const int n_buckets = b->n_buckets = (int) ((b->table_size + Bins - 1) / Bins);
    b->positions = n;

    unsigned blocks = (unsigned) std::min<size_t>((size_t) c.num_cu * 4, (n + kTile - 1) / kTile);
    if (blocks == 0) blocks = 1;
    size_t chunk = (n + blocks - 1) / blocks;
    chunk = (chunk + kTile - 1) / kTile * kTile;
    blocks = (unsigned) ((n + chunk - 1) / chunk);

    const Arg<uint8_t> mask{ nullptr, 1, 0u };
    const Arg<T> xv{ x, T(0), 1u };
    const int vec_ok = aligned16(index) && aligned16(x);
    int rep_shift = 0;
    while ((n_buckets << (rep_shift + 1)) <= kMaxBuckets && rep_shift < 4) ++rep_shift;

    const uint32_t target_pieces = (uint32_t) bucket_target_pieces(n, n_buckets);
    b->max_pieces = target_pieces + (unsigned) n_buckets;
    const size_t count_entries = (size_t) n_buckets * blocks;
    const size_t meta_words = count_entries + 3 * kMaxBuckets + 2;
    if (int rc = ek_hip_malloc(meta_words * sizeof(uint32_t) + (size_t) b->max_pieces * sizeof(T) + 16, &b->meta)) return rc;
    if (int rc = ek_hip_malloc(n * sizeof(uint16_t), &b->pair_idx)) return rc;
    if (int rc = ek_hip_malloc(n * sizeof(T), &b->x_b)) return rc;
    uint32_t *counts = (uint32_t *) b->meta, *row_total = counts + count_entries;
    b->bucket_base = row_total + kMaxBuckets;
    b->piece_prefix = b->bucket_base + kMaxBuckets + 1;
    b->reduce_partials = (void *) (((uintptr_t) (counts + meta_words) + 15) & ~(uintptr_t) 15);

    hipLaunchKernelGGL((k_bin_count<I, Shift>), dim3(blocks), dim3(kThreads), 0, c.stream, counts, index, mask, n, chunk,
                       n_buckets, rep_shift, vec_ok);
    EK_LAUNCH_CHECK("bucket_count", n, n * sizeof(I));
    // (both scans in ONE single-workgroup launch were measured: 28 us against 4.8 + 5.3 us for these two -- 64 rows of 1024
    // counters are too much latency for one CU)
    hipLaunchKernelGGL(k_bin_scan_rows, dim3(n_buckets), dim3(1024), 0, c.stream, counts, row_total, blocks);
    hipLaunchKernelGGL(k_bin_scan_buckets, dim3(1), dim3(256), 0, c.stream, b->bucket_base, b->piece_prefix,
                       (const uint32_t *) row_total, n_buckets, target_pieces);
    EK_LAUNCH_CHECK("bucket_scan", count_entries, 2 * count_entries * sizeof(uint32_t));
    BinStreams<T, 1> st;
    st.value[0] = xv;
    st.weight[0] = Arg<T>{ nullptr, T(1), 0u };
    st.pair_val[0] = (T *) b->x_b;
    st.weighted = 0u;
    st.value_op[0] = EK_COPY;
    // (histogram replicas for the tile ranking, as in the count kernel: measured, no difference -- 0.2006 / 0.2014 / 0.2021 ms
    // for 1 / 2 / 4 replicas on one box; the ranking is not what bounds this kernel)
    hipLaunchKernelGGL((k_bin_partition<T, I, Shift, uint16_t, 1>), dim3(blocks), dim3(kThreads), 0, c.stream,
                       (uint16_t *) b->pair_idx, st, (const uint32_t *) counts, (const uint32_t *) b->bucket_base, index, mask, n,
                       chunk, n_buckets, 0, vec_ok);
    EK_LAUNCH_CHECK("bucket_partition", n, n * (sizeof(I) + sizeof(T)) + n * (sizeof(uint16_t) + sizeof(T)));
    return EK_OK;
}

/// A table of three or more slices: (index, x) are first split by SLICE into contiguous runs -- the count / scan / partition
/// kernels of ek_binned.h with 32-bit slice-local indices -- so that every slice's page partition reads only its own elements
/// (20 B/elt once instead of 8 B/elt per slice).  The slice populations come back to the host (which sizes the per-slice work).
struct CoarseSplit {
    void *idx = nullptr, *x = nullptr, *meta = nullptr;      // slice-local indices and x in slice order; counts / bases
    std::vector<uint32_t> base;                              // host copy of bucket_base[0 .. S]
    ~CoarseSplit() {
        for (void *p : { idx, x, meta })
            if (p) ek_hip_free(p);
    }
};

template <int Shift>
static int coarse_split(CoarseSplit &cs, const float *x, const uint32_t *index, const Arg<uint8_t> &mask, size_t n, int S) {
    RoctxRange range("enoki-hip: slice partition");
    Context &c = ctx();
    unsigned blocks = (unsigned) std::min<size_t>((size_t) c.num_cu * 4, (n + kTile - 1) / kTile);
    if (blocks == 0) blocks = 1;
    size_t chunk = (n + blocks - 1) / blocks;
    chunk = (chunk + kTile - 1) / kTile * kTile;
    blocks = (unsigned) ((n + chunk - 1) / chunk);
    const int vec_ok = aligned16(index) && aligned16(x) && arg_aligned(mask);
    int rep_shift = 0;
    while ((S << (rep_shift + 1)) <= kMaxBuckets && rep_shift < 4) ++rep_shift;
    const size_t count_entries = (size_t) S * blocks;
    if (int rc = ek_hip_malloc((count_entries + 2 * kMaxBuckets + 2) * sizeof(uint32_t), &cs.meta)) return rc;
    if (int rc = ek_hip_malloc(n * sizeof(uint32_t), &cs.idx)) return rc;
    if (int rc = ek_hip_malloc(n * sizeof(float), &cs.x)) return rc;
    uint32_t *counts = (uint32_t *) cs.meta, *row_total = counts + count_entries, *bucket_base = row_total + kMaxBuckets;
    hipLaunchKernelGGL((k_bin_count<uint32_t, Shift>), dim3(blocks), dim3(kThreads), 0, c.stream, counts, index, mask, n, chunk, S,
                       rep_shift, vec_ok);
    EK_LAUNCH_CHECK("bucket_slice_count", n, n * sizeof(uint32_t) + arg_bytes(mask, n));
    hipLaunchKernelGGL(k_bin_scan_rows, dim3(S), dim3(1024), 0, c.stream, counts, row_total, blocks);
    hipLaunchKernelGGL(k_bin_scan_buckets, dim3(1), dim3(256), 0, c.stream, bucket_base, (uint32_t *) nullptr, (const uint32_t *) row_total, S, 0u);
    EK_LAUNCH_CHECK("bucket_slice_scan", count_entries, 2 * count_entries * sizeof(uint32_t));
    BinStreams<float, 1> st;
    st.value[0] = Arg<float>{ x, 0.f, 1u };
    st.weight[0] = Arg<float>{ nullptr, 1.f, 0u };
    st.pair_val[0] = (float *) cs.x;
    st.weighted = 0u;
    st.value_op[0] = EK_COPY;
    hipLaunchKernelGGL((k_bin_partition<float, uint32_t, Shift, uint32_t, 1>), dim3(blocks), dim3(kThreads), 0, c.stream,
                       (uint32_t *) cs.idx, st, (const uint32_t *) counts, (const uint32_t *) bucket_base, index, mask, n, chunk, S, 0,
                       vec_ok);
    EK_LAUNCH_CHECK("bucket_slice_partition", n, n * 16 + arg_bytes(mask, n));
    cs.base.resize(S + 1);
    EK_HIP_CHECK(hipMemcpyAsync(cs.base.data(), bucket_base, (S + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
    EK_HIP_CHECK(hipStreamSynchronize(c.stream));
    return EK_OK;
}

/// The same object from the single-pass paged partition (ek_paged.h): 4-byte element types, n <= 2^30.  One streaming pass
/// over (index, x) + the page directory: no count pass, no scans.
template <typename I>
static int bucketed_create_paged(Bucketed *b, const float *x, const I *index, const Arg<uint8_t> &mask, int shift) {
    RoctxRange range("enoki-hip: bucket partition (pages)");
    Context &c = ctx();
    b->shift = shift;
    const size_t n = b->n;
    const int n_buckets = b->n_buckets = (int) ((b->table_size + ((size_t) 1 << shift) - 1) >> shift);
    const PagedPlan p = paged_plan(n, n_buckets, c.num_cu, c.tuning.xcd_balance != 0);
    if (p.W > 1024) return fail(EK_ERR_UNSUPPORTED, "ek_hip_bucketed_pair_create(): %u workgroups", p.W);
    b->page_shift = p.page_shift;
    b->positions = p.page_slots << p.page_shift;
    const uint32_t target_pieces = (uint32_t) bucket_target_pieces(n, n_buckets);
    b->max_pieces = target_pieces + (unsigned) n_buckets;
    // meta: page totals [kPgReplicas][2][256] + meta row [256] (kPgCounterWords) | base_full[257] | base_part[257] | piece_prefix[257] | reduce partials
    const size_t meta_words = kPgCounterWords + 3 * (kMaxBuckets + 1) + 1;
    // the counter block: one of the context's ring when one is free (then usually without a fill, see MetaRing), else an allocation
    bool filled = false;
    static const bool use_ring = [] { const char *e = getenv("ENOKI_HIP_META_RING"); return !e || atoi(e) != 0; }();
    if (use_ring && b->max_pieces <= MetaRing::kPartials && refuse_while_capturing_quiet() == EK_OK) {
        MetaRing &r = meta_ring();
        for (int k = 0; k < MetaRing::kBlocks && b->meta_slot < 0; ++k) {
            if (r.busy[k]) continue;
            if (!r.block[k]) {
                if (ek_hip_malloc(MetaRing::bytes(), &r.block[k]) != EK_OK) break;
                r.clean[k] = false;
            }
            r.busy[k] = true;
            b->meta_slot = k;
            b->meta = r.block[k];
            filled = r.clean[k];
        }
    }
    if (b->meta_slot < 0)
        if (int rc = ek_hip_malloc(meta_words * sizeof(uint32_t) + (size_t) b->max_pieces * 16 + 16, &b->meta)) return rc;
    if (int rc = ek_hip_malloc(b->positions * sizeof(uint16_t), &b->pair_idx)) return rc;
    if (int rc = ek_hip_malloc(b->positions * sizeof(float), &b->x_b)) return rc;
    const size_t part_entries = (size_t) p.W * n_buckets;
    if (int rc = ek_hip_malloc((p.page_slots + part_entries + 1) * sizeof(uint32_t), &b->page_lists)) return rc;
    b->glist_full = (uint32_t *) b->page_lists;
    b->glist_part = b->glist_full + p.page_slots;
    uint32_t *gtotal = (uint32_t *) b->meta;
    b->bucket_base = gtotal + kPgCounterWords;
    b->active = gtotal + kPgMetaBase + kPgMetaResult;        // (written by the directory launch from the partition's accumulators)
    b->ticket = gtotal + kPgMetaBase + kPgMetaFinishTicket;  // (zero between launches: reset by whoever draws the last ticket)
    b->has_mask = mask.vec != 0;
    b->base_part = b->bucket_base + kMaxBuckets + 1;
    b->piece_prefix = b->base_part + kMaxBuckets + 1;
    b->reduce_partials = (void *) (((uintptr_t) (gtotal + meta_words) + 15) & ~(uintptr_t) 15);
    // what only the two kernels below need: the workgroups' own page lists and counts
    Scratch work;
    if (int rc = work.alloc((2 * p.page_slots + 3 * part_entries) * sizeof(uint32_t))) return rc;
    PagedOut<float> out;
    out.lp = (uint16_t *) b->pair_idx;
    out.xp = (float *) b->x_b;
    out.wdir = (uint32_t *) work.ptr;
    out.wlist = out.wdir + p.page_slots;
    out.cnt_full = out.wlist + p.page_slots;
    out.loff = out.cnt_full + part_entries;
    out.part = out.loff + part_entries;
    out.gtotal = gtotal;
    out.active = gtotal + kPgMetaBase + kPgMetaAccum;
    out.lo = b->win_lo; out.span = b->win_span ? b->win_span : (uint32_t) std::min<size_t>(b->table_size, 0xFFFFFFFFu);
    // tiles dealt to the classes w % 8 by the weights the previous launches fed back (ek_paged.h); a launch of at least 32 tiles per
    // workgroup stamps its loops and lets the directory launch update the weights
    // (ENOKI_HIP_XCD_BALANCE=2: the slots are provisioned and the loops stamped, but the chunks stay equal -- what
    // ek_hip_partition_class_state() then reports is the imbalance itself)
    uint32_t *class_block = p.balanced ? class_weights_or_null() : nullptr;
    out.class_w = c.tuning.xcd_balance == 1 ? class_block : nullptr;
    const bool stamped = class_block && n >= (size_t) 32 * kPgTile * p.W && p.W <= 1024;
    const bool feedback = stamped && out.class_w;
    out.class_stamp = stamped ? class_state().stamps() : nullptr;
    static const uint32_t band = [] { const char *e = getenv("ENOKI_HIP_XCD_BAND"); return e ? (uint32_t) atoi(e) : kPgWeightBand; }();
    out.class_band = band;
    if (!filled) {
        EK_HIP_CHECK(hipMemsetAsync(gtotal, 0, kPgCounterWords * sizeof(uint32_t), c.stream));
        note_launch("bucket_meta_clear", kPgCounterWords, kPgCounterWords * sizeof(uint32_t));   // (its own mark: a profiled run must not bill the fill to the partition)
    }
    const int vec_ok = aligned16(index) && aligned16(x) && arg_aligned(mask);
    auto launch = [&](auto kernel) -> int {
        // the workgroup's own page directory in the LDS behind the records when it fits next to them and the kernel's static
        // arrays (160 KiB per workgroup on gfx950: inputs up to ~64 Mi elements); ENOKI_HIP_WDIR_LDS=0: through global memory
        static const bool want = [] { const char *e = getenv("ENOKI_HIP_WDIR_LDS"); return !e || atoi(e) != 0; }();
        size_t lds = p.lds;
        out.wdir_lds = 0;
        const size_t fixed_lds = want ? static_lds_of(kernel) : SIZE_MAX;
        if (fixed_lds != SIZE_MAX && fixed_lds + p.lds + (size_t) p.slots * sizeof(uint32_t) <= (size_t) 160 * 1024) {
            lds += (size_t) p.slots * sizeof(uint32_t);
            out.wdir_lds = 1;
        }
        if (int rc = allow_big_lds(kernel, lds)) return rc;
        hipLaunchKernelGGL(kernel, dim3(p.W), dim3(kPgThreads), lds, c.stream, out, index, mask, x, n, p.chunk, n_buckets, shift,
                           p.cap, p.slots, vec_ok);
        return EK_OK;
    };
    int rc;
    if (p.page_shift == 6) rc = mask.vec ? launch(k_page_partition<float, I, 6, true>) : launch(k_page_partition<float, I, 6, false>);
    else rc = mask.vec ? launch(k_page_partition<float, I, 5, true>) : launch(k_page_partition<float, I, 5, false>);
    if (rc) return rc;
    EK_LAUNCH_CHECK("bucket_partition", n, n * (sizeof(I) + sizeof(float)) + arg_bytes(mask, n) + n * (sizeof(uint16_t) + sizeof(float)));
    hipLaunchKernelGGL(k_page_directory, dim3(n_buckets, kPgDirSlices), dim3(256), 0, c.stream, b->glist_full, b->glist_part, b->bucket_base,
                       b->base_part, b->piece_prefix, gtotal, (const uint32_t *) out.cnt_full,
                       (const uint32_t *) out.loff, (const uint32_t *) out.part, (const uint32_t *) out.wlist, p.W, p.slots, n_buckets,
                       target_pieces, feedback ? class_state().weights() : (uint32_t *) nullptr, (const uint32_t *) out.class_stamp, band);
    EK_LAUNCH_CHECK("bucket_directory", p.page_slots, 2 * p.page_slots * sizeof(uint32_t));
    return EK_OK;
}


template <typename T, int ROp>
static int bucketed_forward_launch(Bucketed *b, void *out, int map_op, bool keep, int keep_op = EK_COPY) {
    Context &c = ctx();
    const size_t Bins = b->bins();
    const size_t lds = Bins * sizeof(PairRec<T>);
    // vectors per lane and step.  float: two (one: 6 % slower, four: 13 % slower, same box); double: one -- with two the
    // kernel needs more than the 128 registers a 1024-thread workgroup leaves a lane (12 spilled; the adjoint: 250)
    constexpr int VV = sizeof(T) == 8 ? 1 : 2;
    // keep the OTHER half of a sincos pair instead of u: only when it is exactly that (sin reduced, cos kept or vice versa)
    const bool partner = keep && ROp != EK_REDUCE_NONE &&
                         ((map_op == EK_SIN && keep_op == EK_COS) || (map_op == EK_COS && keep_op == EK_SIN));
    void **kept = partner ? &b->m_b : &b->u_b;
    if (keep && !*kept)
        if (int rc = ek_hip_malloc(b->positions * sizeof(T), kept)) return rc;
    const int flip_a = b->flip_a(), flip_c = b->flip_c(), two = b->two_roundings();
    EK_BY_LAYOUT(b, {
        if (int rc = allow_big_lds(k_bucket_pair_forward<T, ROp, VV, PS>, lds)) return rc;
        hipLaunchKernelGGL((k_bucket_pair_forward<T, ROp, VV, PS>), dim3(b->max_pieces), dim3(kBucketThreads), lds, c.stream,
                           (T *) b->reduce_partials, keep ? (T *) *kept : (T *) nullptr, (const T *) b->table_a,
                           (const T *) b->table_c, b->table_size, flip_a, flip_c, two, (const uint16_t *) b->pair_idx,
                           (const T *) b->x_b, (const T *) nullptr, b->lists(), map_op, partner ? 1 : 0, b->shift,
                           b->template finish<T>(out, map_op, ROp != EK_REDUCE_NONE));
    });
    EK_LAUNCH_CHECK(ROp == EK_REDUCE_NONE ? "bucket_pair_fma" : "bucket_pair_fma_reduce", b->n,
                    b->n * (sizeof(uint16_t) + sizeof(T) + (keep ? sizeof(T) : 0)) + (b->table_c ? 2 : 1) * b->table_size * sizeof(T));
    b->launched_reducing();
    if (keep && partner) { b->has_m = true; b->m_op = keep_op; }
    else if (keep) b->has_u = true;
    if constexpr (ROp != EK_REDUCE_NONE) {
        if (!b->ticket) {
            hipLaunchKernelGGL((k_bucket_reduce_final<T, ROp>), dim3(1), dim3(256), 0, c.stream, (T *) out,
                               (const T *) b->reduce_partials, b->max_pieces, b->masked_ptr(), b->n, map_op);
            EK_LAUNCH_CHECK("reduce_stage2", (size_t) b->max_pieces, (size_t) b->max_pieces * sizeof(T) + sizeof(T));
        }
    }
    return EK_OK;
}

/// reduce_op over map_op(values) of what an earlier pass kept in list order (u, or the kept half of a sincos pair)
template <typename T, int ROp>
static int bucketed_reduce_kept_launch(Bucketed *b, void *out, int map_op, const void *values, int zero_op) {
    Context &c = ctx();
    constexpr int VV = sizeof(T) == 8 ? 1 : 2;
    EK_BY_LAYOUT(b, {
        hipLaunchKernelGGL((k_bucket_pair_forward<T, ROp, VV, PS, true>), dim3(b->max_pieces), dim3(kBucketThreads), 0, c.stream,
                           (T *) b->reduce_partials, (T *) nullptr, (const T *) nullptr, (const T *) nullptr, b->table_size, 0, 0, 0,
                           (const uint16_t *) b->pair_idx, (const T *) b->x_b, (const T *) values, b->lists(), map_op, 0, b->shift,
                           b->template finish<T>(out, zero_op));
    });
    EK_LAUNCH_CHECK("bucket_reduce_kept", b->n, b->n * sizeof(T));
    b->launched_reducing();
    if (!b->ticket) {
        hipLaunchKernelGGL((k_bucket_reduce_final<T, ROp>), dim3(1), dim3(256), 0, c.stream, (T *) out, (const T *) b->reduce_partials,
                           b->max_pieces, b->masked_ptr(), b->n, zero_op);
        EK_LAUNCH_CHECK("reduce_stage2", (size_t) b->max_pieces, (size_t) b->max_pieces * sizeof(T) + sizeof(T));
    }
    return EK_OK;
}

template <typename T>
static int bucketed_reduce_kept(Bucketed *b, int reduce_op, int map_op, void *out, const void *values, int zero_op) {
    // zero_op: what the masked-out lanes (u = 0) contribute, as a function of 0
    if (b->page_shift == 0) {          // contiguous lists have no holes: an ordinary reduction
        if (map_op == EK_COPY) return ek_hip_reduce(reduce_op, b->type, out, values, b->n);
        return ek_hip_reduce_map(reduce_op, map_op, b->type, out, values, b->n);
    }
    switch (reduce_op) {
        case EK_HSUM: return bucketed_reduce_kept_launch<T, EK_HSUM>(b, out, map_op, values, zero_op);
        case EK_HPROD: return bucketed_reduce_kept_launch<T, EK_HPROD>(b, out, map_op, values, zero_op);
        case EK_HMIN: return bucketed_reduce_kept_launch<T, EK_HMIN>(b, out, map_op, values, zero_op);
        case EK_HMAX: return bucketed_reduce_kept_launch<T, EK_HMAX>(b, out, map_op, values, zero_op);
        default: return fail(EK_ERR_INVALID, "ek_hip_bucketed_reduce(): unknown op %d", reduce_op);
    }
}

/// hsum of map_op(u) with the adjoint of the gathers -- the sums of keep_op(u) and x * keep_op(u) per entry -- formed in the
/// same pass (k_bucket_pair_forward_adjoint)

template <typename T>
static int bucketed_reduce(Bucketed *b, int reduce_op, int map_op, void *out, bool keep, int keep_op) {
    RoctxRange range("enoki-hip: bucket-ordered gather + fma + reduction");
    // a plan with half-size buckets was made for this: the sum of one half of sincos(u) whose other half is to be kept
    if (b->shift < bin_shift_of<T> && reduce_op == EK_HSUM && keep && !b->has_u && !b->has_m && !b->has_early &&
        early_pair_supported(map_op, keep_op))
        return bucketed_forward_adjoint_launch<T>(b, out, map_op, keep_op);
    if (b->has_m && map_op == b->m_op) return bucketed_reduce_kept<T>(b, reduce_op, EK_COPY, out, b->m_b, b->m_op);   // the kept half itself
    if (b->has_u) return bucketed_reduce_kept<T>(b, reduce_op, map_op, out, b->u_b, map_op);      // u already exists in list order
    switch (reduce_op) {
        case EK_HSUM: return bucketed_forward_launch<T, EK_HSUM>(b, out, map_op, keep, keep_op);
        case EK_HPROD: return bucketed_forward_launch<T, EK_HPROD>(b, out, map_op, keep, keep_op);
        case EK_HMIN: return bucketed_forward_launch<T, EK_HMIN>(b, out, map_op, keep, keep_op);
        case EK_HMAX: return bucketed_forward_launch<T, EK_HMAX>(b, out, map_op, keep, keep_op);
        default: return fail(EK_ERR_INVALID, "ek_hip_bucketed_reduce(): unknown op %d", reduce_op);
    }
}

template <typename T, int C>
static int bucketed_accumulate(Bucketed *b, T *const *bases, const BucketStreams<T, C> &st, const void *u_src, unsigned fresh) {
    Context &c = ctx();
    const size_t Bins = b->bins();
    const size_t lds = (size_t) C * Bins * sizeof(T);
    constexpr int VV = sizeof(T) == 8 ? 1 : 2;          // as in the forward kernel
    Scratch partials;
    if (int rc = partials.alloc((size_t) C * b->max_pieces * Bins * sizeof(T))) return rc;
    EK_BY_LAYOUT(b, {
        if (int rc = allow_big_lds(k_bucket_accumulate<T, C, VV, PS>, lds)) return rc;
        hipLaunchKernelGGL((k_bucket_accumulate<T, C, VV, PS>), dim3(b->max_pieces), dim3(kBucketThreads), lds, c.stream,
                           (T *) partials.ptr, (const uint16_t *) b->pair_idx, (const T *) u_src, (const T *) b->x_b, b->lists(), st,
                           b->shift);
    });
    EK_LAUNCH_CHECK("bucket_accumulate", (size_t) C * b->n,
                    b->n * (sizeof(uint16_t) + (st.from_u ? sizeof(T) : 0) + (st.weighted ? sizeof(T) : 0)) +
                    (size_t) C * b->max_pieces * Bins * sizeof(T));
    FoldTargets<T, C> targets;
    for (int s = 0; s < C; ++s) { targets.table[s] = bases[s]; targets.scale[s] = T(1); }
    hipLaunchKernelGGL((k_bin_fold_pieces<T, C>), dim3(fold_grid(b->table_size), C), dim3(256), 0, c.stream, targets,
                       (const T *) partials.ptr, (const uint32_t *) b->piece_prefix, b->table_size, (size_t) b->max_pieces * Bins, fresh,
                       b->shift);
    EK_LAUNCH_CHECK("scatter_add_fold", (size_t) C * b->table_size,
                    (size_t) C * ((size_t) b->max_pieces * Bins * sizeof(T) + 2 * b->table_size * sizeof(T)));
    return EK_OK;
}


// The per-piece tables of the fixed-point forward + adjoint kernel (bucketed_early.hip) into the gradient tables: 64-bit sums of
// all pieces of a bucket are added as INTEGERS (order-free: the result does not depend on which piece an element went to, nor on
// anything else that varies from run to run), converted once, scaled back by the power of two they were scaled up with, times the
// caller's factor.  A piece that ran under locks (piece_mode 1: max |x| not finite, or a NaN term) left floats, which are added on top.
template <int C>
__global__ __launch_bounds__(256) void k_fold_fixed_pieces(FoldTargets<float, C> targets, const long long *__restrict__ partials,
                                                           const uint32_t *__restrict__ piece_mode, const uint32_t *__restrict__ piece_prefix,
                                                           size_t table_size, size_t partial_stride, unsigned fresh, int shift,
                                                           const uint32_t *__restrict__ xmax_bits, int S0, int first_table) {
    const size_t k0 = ((size_t) blockIdx.x * 256 + threadIdx.x) * kFoldPerLane;
    if (k0 >= table_size) return;
    const int c = first_table + (int) blockIdx.y;                 // 0: the plain sums, 1: the x-weighted sums
    float *__restrict__ target = targets.table[blockIdx.y];
    partials += (size_t) c * partial_stride;
    const uint32_t b = (uint32_t) (k0 >> shift), local = (uint32_t) (k0 & (((size_t) 1 << shift) - 1));
    const bool is_fresh = (fresh >> blockIdx.y) & 1u;
    const uint32_t p0 = piece_prefix[b], p1 = piece_prefix[b + 1];
    // (128 bits: a piece's sums stay below 2^62 by the choice of the scale, a skewed bucket of many pieces may not)
    unsigned long long ilo[kFoldPerLane];
    long long ihi[kFoldPerLane];
    float fsum[kFoldPerLane], old[kFoldPerLane];
#pragma unroll
    for (int j = 0; j < kFoldPerLane; ++j) { ilo[j] = 0; ihi[j] = 0; fsum[j] = 0.f; old[j] = 0.f; }
    const int lim = (int) (table_size - k0 < (size_t) kFoldPerLane ? table_size - k0 : (size_t) kFoldPerLane);
    if (!is_fresh) {
#pragma unroll
        for (int j = 0; j < kFoldPerLane; ++j) if (j < lim) old[j] = target[k0 + j];
    }
    const uint32_t xm = xmax_bits[0];
    if (p1 - p0 == 1u) {
        // a bucket that is one piece wrote floats whatever it ran under (bucketed_early.hip): no mode to wait for, a quarter of the bytes
        const float *slotf = reinterpret_cast<const float *>(partials + ((size_t) p0 << shift));
#pragma unroll
        for (int j = 0; j < kFoldPerLane; ++j) if (j < lim) fsum[j] = __builtin_nontemporal_load(slotf + local + j);
    } else
    for (uint32_t p = p0; p < p1; ++p) {
        // (mode and sums requested together: the 64-bit words of a slot can be read whatever the piece wrote into it)
        const long long *slot = partials + ((size_t) p << shift);
        const uint32_t mode = piece_mode[p];
        long long v[kFoldPerLane];
#pragma unroll
        for (int j = 0; j < kFoldPerLane; ++j) v[j] = j < lim ? __builtin_nontemporal_load(slot + local + j) : 0ll;
        if (mode == 0u) {
#pragma unroll
            for (int j = 0; j < kFoldPerLane; ++j) {
                const unsigned long long before = ilo[j];
                ilo[j] += (unsigned long long) v[j];
                ihi[j] += (v[j] >> 63) + (ilo[j] < before ? 1 : 0);
            }
        } else {
            const float *slotf = reinterpret_cast<const float *>(slot);
#pragma unroll
            for (int j = 0; j < kFoldPerLane; ++j) if (j < lim) fsum[j] += slotf[local + j];
        }
    }
    const float back = fixed_scale(S0, xm, c == 1).back, f = targets.scale[blockIdx.y];
#pragma unroll
    for (int j = 0; j < kFoldPerLane; ++j) {
        if (j >= lim) continue;
        // (one conversion of the exact integer: the same integer gives the same float whatever the pieces were)
        const bool narrow = ihi[j] == ((long long) ilo[j] >> 63);
        float v = narrow ? (float) (long long) ilo[j] : (float) ((double) ihi[j] * 18446744073709551616.0 + (double) ilo[j]);
        v = v * back + fsum[j];
        if (f != 1.f) v = v * f;
        target[k0 + j] = old[j] + v;
    }
}

template <typename T>
static int bucketed_scatter_add(Bucketed *b, int count, void *const *bases, const int *from_u, const int *map_ops,
                                const uint64_t *imm_bits, const int *weighted, const int *fresh, const uint64_t *scale_bits) {
    RoctxRange range("enoki-hip: bucket-ordered scatter_add");
    T scale[4] = { T(1), T(1), T(1), T(1) };
    if (scale_bits)
        for (int s = 0; s < count; ++s) memcpy(&scale[s], &scale_bits[s], sizeof(T));
    if (b->has_early && count >= 1 && count <= 2) {
        // exactly the streams that the forward pass summed already -- early_op(u), and x * early_op(u)?  Fold them.
        bool match = true;
        for (int s = 0; s < count; ++s)
            match = match && from_u[s] && (map_ops ? map_ops[s] : (int) EK_COPY) == b->early_op;
        if (count == 2) match = match && (weighted[0] != 0) != (weighted[1] != 0);
        if (match) {
            Context &c = ctx();
            const size_t stride = (size_t) b->max_pieces * b->bins();
            const unsigned grid = fold_grid(b->table_size);
            if constexpr (std::is_same_v<T, float>) {
                if (b->early_fixed) {
                    const uint32_t *xmax = b->active + (kPgMetaResultXmax - kPgMetaResult);
                    if (count == 2) {
                        const int s_plain = weighted[0] ? 1 : 0, s_weighted = 1 - s_plain;
                        FoldTargets<float, 2> targets;
                        targets.table[0] = (float *) bases[s_plain];
                        targets.table[1] = (float *) bases[s_weighted];
                        targets.scale[0] = scale[s_plain];
                        targets.scale[1] = scale[s_weighted];
                        const unsigned fr = fresh ? ((fresh[s_plain] ? 1u : 0u) | (fresh[s_weighted] ? 2u : 0u)) : 0u;
                        hipLaunchKernelGGL((k_fold_fixed_pieces<2>), dim3(grid, 2), dim3(256), 0, c.stream, targets, (const long long *) b->early,
                                           (const uint32_t *) b->early_modes, (const uint32_t *) b->piece_prefix, b->table_size, stride, fr, b->shift,
                                           xmax, b->early_S0, 0);
                    } else {
                        FoldTargets<float, 1> targets;
                        targets.table[0] = (float *) bases[0];
                        targets.scale[0] = scale[0];
                        hipLaunchKernelGGL((k_fold_fixed_pieces<1>), dim3(grid, 1), dim3(256), 0, c.stream, targets, (const long long *) b->early,
                                           (const uint32_t *) b->early_modes, (const uint32_t *) b->piece_prefix, b->table_size, stride,
                                           (fresh && fresh[0]) ? 1u : 0u, b->shift, xmax, b->early_S0, weighted[0] ? 1 : 0);
                    }
                    EK_LAUNCH_CHECK("scatter_add_fold", (size_t) count * b->table_size,
                                    (size_t) count * (stride * sizeof(long long) + 2 * b->table_size * sizeof(T)));
                    return EK_OK;
                }
            }
            if (count == 2) {
                // partial table 0: unweighted, 1: weighted
                const int s_plain = weighted[0] ? 1 : 0, s_weighted = 1 - s_plain;
                FoldTargets<T, 2> targets;
                targets.table[0] = (T *) bases[s_plain];
                targets.table[1] = (T *) bases[s_weighted];
                targets.scale[0] = scale[s_plain];
                targets.scale[1] = scale[s_weighted];
                const unsigned fr = fresh ? ((fresh[s_plain] ? 1u : 0u) | (fresh[s_weighted] ? 2u : 0u)) : 0u;
                hipLaunchKernelGGL((k_bin_fold_pieces<T, 2>), dim3(grid, 2), dim3(256), 0, c.stream, targets, (const T *) b->early,
                                   (const uint32_t *) b->piece_prefix, b->table_size, stride, fr, b->shift);
            } else {
                FoldTargets<T, 1> targets;
                targets.table[0] = (T *) bases[0];
                targets.scale[0] = scale[0];
                hipLaunchKernelGGL((k_bin_fold_pieces<T, 1>), dim3(grid, 1), dim3(256), 0, c.stream, targets,
                                   (const T *) b->early + (weighted[0] ? stride : 0), (const uint32_t *) b->piece_prefix,
                                   b->table_size, stride, (fresh && fresh[0]) ? 1u : 0u, b->shift);
            }
            EK_LAUNCH_CHECK("scatter_add_fold", (size_t) count * b->table_size,
                            (size_t) count * (stride * sizeof(T) + 2 * b->table_size * sizeof(T)));
            return EK_OK;
        }
    }
    bool need_u = false, all_kept = b->has_m;
    for (int s = 0; s < count; ++s) {
        need_u = need_u || from_u[s];
        if (from_u[s]) all_kept = all_kept && (map_ops ? map_ops[s] : (int) EK_COPY) == b->m_op;
    }
    // every stream that reads u wants exactly the half of sincos(u) that the forward kept: stream it as is
    const bool use_kept = need_u && all_kept;
    if (need_u && !use_kept && !b->has_u)
        if (int rc = bucketed_forward_launch<T, EK_REDUCE_NONE>(b, nullptr, EK_COPY, true)) return rc;
    const void *u_src = use_kept ? b->m_b : b->u_b;
    // two tables per launch: their LDS tables fill the 128 KiB a workgroup may use
    for (int s0 = 0; s0 < count; s0 += 2) {
        const int C = std::min(2, count - s0);
        T *tb[2] = { (T *) bases[s0], C == 2 ? (T *) bases[s0 + 1] : nullptr };
        if (C == 2) {
            BucketStreams<T, 2> st{};
            // the weighted stream second (the tape queues the adjoints of the addend and of the factor in either order): the
            // kernel has a compile-time form for { f(u), x * f(u) }
            const bool swap = weighted[s0] && !weighted[s0 + 1];
            if (swap) std::swap(tb[0], tb[1]);
            for (int s = 0; s < 2; ++s) {
                const int src = s0 + (swap ? 1 - s : s);
                st.map_op[s] = (map_ops && !use_kept) ? map_ops[src] : (int) EK_COPY;
                memcpy(&st.imm[s], &imm_bits[src], sizeof(T));
                st.scale[s] = scale[src];
                st.from_u |= (from_u[src] ? 1u : 0u) << s;
                st.weighted |= (weighted[src] ? 1u : 0u) << s;
            }
            const int f0 = swap ? s0 + 1 : s0, f1 = swap ? s0 : s0 + 1;
            const unsigned fr = fresh ? ((fresh[f0] ? 1u : 0u) | (fresh[f1] ? 2u : 0u)) : 0u;
            if (int rc = bucketed_accumulate<T, 2>(b, tb, st, u_src, fr)) return rc;
        } else {
            BucketStreams<T, 1> st{};
            st.map_op[0] = (map_ops && !use_kept) ? map_ops[s0] : (int) EK_COPY;
            memcpy(&st.imm[0], &imm_bits[s0], sizeof(T));
            st.scale[0] = scale[s0];
            st.from_u = from_u[s0] ? 1u : 0u;
            st.weighted = weighted[s0] ? 1u : 0u;
            if (int rc = bucketed_accumulate<T, 1>(b, tb, st, u_src, (fresh && fresh[s0]) ? 1u : 0u)) return rc;
        }
    }
    return EK_OK;
}

// ---- plain scatter_add(value, index, mask) through the page partition ------------------------------------------------------
// The LDS-binned scatter_add of scatter_binned.hip reads the indices twice and needs two scans before it can place a pair
// (count 5 + partition 15 + accumulate 6 B per pair, five launches).  The single-pass page partition of the bucket-ordered path
// does the same job for ONE f32 value stream in 14 + 6 B and three launches + the fold: the (index, value) pairs are partitioned
// into pages by bucket of 16 Ki table entries, every piece of a bucket adds its values into an LDS table under the exchange
// lock, the per-piece tables are folded into the target.  Same sums up to the order of the additions (class D, like the binned
// path); out-of-range indices and masked-out pairs are dropped.
int scatter_add_paged(float *base, size_t table_size, const float *value, const uint32_t *index, const Arg<uint8_t> &mask, size_t n) {
    RoctxRange range("enoki-hip: scatter_add (paged)");
    ek::Bucketed obj;
    obj.type = EK_F32; obj.index_type = EK_U32; obj.op = EK_FMADD;
    obj.n = n; obj.table_size = table_size;
    if (int rc = bucketed_create_paged<uint32_t>(&obj, value, index, mask, bin_shift_of<float>)) return rc;
    BucketStreams<float, 1> st{};
    st.map_op[0] = EK_COPY; st.imm[0] = 0.f; st.scale[0] = 1.f;
    st.from_u = 0u; st.weighted = 1u; st.plain_x = 1u;
    float *bases[1] = { base };
    return bucketed_accumulate<float, 1>(&obj, bases, st, nullptr, 0u);
}

bool scatter_add_paged_applicable(size_t table_size, size_t n) {
    // (from 32 buckets on: with a handful of buckets every element of a tile meets the same few LDS counters -- 64 Mi adds into
    // 64 Ki bins: 0.51 ms paged against 0.36 ms binned, 256 Ki bins: equal, 1 Mi: 0.28 against 0.33, 4 Mi: 0.31 against 0.39;
    // profiles/probe_scatter_paged_r05.txt)
    return ctx().tuning.bucket_ordered && !ctx().tuning.deterministic && n >= ((size_t) 1 << 18) && n < ((size_t) 1 << 30) &&
           table_size >= (size_t) 32 * bins_of<float> && table_size <= (size_t) kMaxBuckets * bins_of<float>;
}

// ---- partition of an index array alone (ek_hip_index_partition_*) ------------------------------------------------------
struct IndexPartition {
    ek_hip_index_partition_info info{};
    void *meta = nullptr, *local = nullptr, *lists = nullptr;
    ~IndexPartition() {
        for (void *p : { meta, local, lists })
            if (p) ek_hip_free(p);
    }
};

// The index array alone through the single-pass page partition (k_page_partition<.., IndexOnly>): idx 4 (+ mask 1) read, 4 written per
// entry, two launches -- instead of count + two scans + partition (idx read twice: 14 B per entry, four launches; 124 us of cfg4's 245 us
// step at 32 Mi rays).  4-byte records give every one of up to 256 buckets two 64-element pages of LDS.
static int index_partition_run_paged(IndexPartition *ip, const uint32_t *index, const Arg<uint8_t> &mask, size_t n, int n_buckets, int shift) {
    RoctxRange range("enoki-hip: index partition (pages)");
    Context &c = ctx();
    const PagedPlan p = paged_plan(n, n_buckets, c.num_cu, false, true);
    if (p.W > 1024) return fail(EK_ERR_UNSUPPORTED, "ek_hip_index_partition_create(): %u workgroups", p.W);
    const size_t meta_words = kPgCounterWords + 3 * (kMaxBuckets + 1) + 1;
    if (int rc = ek_hip_malloc(meta_words * sizeof(uint32_t), &ip->meta)) return rc;
    if (int rc = ek_hip_malloc((p.page_slots << p.page_shift) * sizeof(uint32_t), &ip->local)) return rc;
    const size_t part_entries = (size_t) p.W * n_buckets;
    if (int rc = ek_hip_malloc((p.page_slots + part_entries + 1) * sizeof(uint32_t), &ip->lists)) return rc;
    uint32_t *gtotal = (uint32_t *) ip->meta, *base_full = gtotal + kPgCounterWords, *base_part = base_full + kMaxBuckets + 1,
             *piece_prefix = base_part + kMaxBuckets + 1;
    uint32_t *glist_full = (uint32_t *) ip->lists, *glist_part = glist_full + p.page_slots;
    Scratch work;
    if (int rc = work.alloc((2 * p.page_slots + 3 * part_entries) * sizeof(uint32_t))) return rc;
    PagedOut<float> out{};
    out.lp = nullptr;
    out.xp = (float *) ip->local;
    out.wdir = (uint32_t *) work.ptr;
    out.wlist = out.wdir + p.page_slots;
    out.cnt_full = out.wlist + p.page_slots;
    out.loff = out.cnt_full + part_entries;
    out.part = out.loff + part_entries;
    out.gtotal = gtotal;
    out.active = gtotal + kPgMetaBase + kPgMetaAccum;
    out.lo = 0; out.span = (uint32_t) std::min<size_t>(ip->info.range, 0xFFFFFFFFu);
    out.class_w = nullptr; out.class_stamp = nullptr; out.class_band = kPgWeightBand;
    // (a kernel of our own: the runtime's memset is a launch with its own barrier packets around it -- the step's widest gap)
    if (int rc = ek_hip_fill(EK_U32, gtotal, 0, kPgCounterWords)) return rc;
    const int vec_ok = aligned16(index) && arg_aligned(mask);
    auto launch = [&](auto kernel) -> int {
        size_t lds = p.lds;
        out.wdir_lds = 0;
        const size_t fixed_lds = static_lds_of(kernel);
        if (fixed_lds != SIZE_MAX && fixed_lds + p.lds + (size_t) p.slots * sizeof(uint32_t) <= (size_t) 160 * 1024) {
            lds += (size_t) p.slots * sizeof(uint32_t);
            out.wdir_lds = 1;
        }
        if (int rc = allow_big_lds(kernel, lds)) return rc;
        hipLaunchKernelGGL(kernel, dim3(p.W), dim3(kPgThreads), lds, c.stream, out, index, mask, (const float *) nullptr, n, p.chunk, n_buckets,
                           shift, p.cap, p.slots, vec_ok);
        return EK_OK;
    };
    int rc;
    if (p.page_shift == 6) rc = mask.vec ? launch(k_page_partition<float, uint32_t, 6, true, true>) : launch(k_page_partition<float, uint32_t, 6, false, true>);
    else rc = mask.vec ? launch(k_page_partition<float, uint32_t, 5, true, true>) : launch(k_page_partition<float, uint32_t, 5, false, true>);
    if (rc) return rc;
    EK_LAUNCH_CHECK("index_partition", n, n * 2 * sizeof(uint32_t) + arg_bytes(mask, n));
    hipLaunchKernelGGL(k_page_directory, dim3(n_buckets, kPgDirSlices), dim3(256), 0, c.stream, glist_full, glist_part, base_full,
                       base_part, piece_prefix, gtotal, (const uint32_t *) out.cnt_full, (const uint32_t *) out.loff,
                       (const uint32_t *) out.part, (const uint32_t *) out.wlist, p.W, p.slots, n_buckets, 0u,
                       (uint32_t *) nullptr, (const uint32_t *) nullptr, kPgWeightBand);
    EK_LAUNCH_CHECK("index_partition_directory", p.page_slots, 2 * p.page_slots * sizeof(uint32_t));
    ip->info.bucket_base = base_full;
    ip->info.local = (const uint32_t *) ip->local;
    ip->info.page_shift = p.page_shift;
    ip->info.pages_full = glist_full;
    ip->info.pages_part = glist_part;
    ip->info.part_base = base_part;
    return EK_OK;
}

template <typename I, int Shift>
static int index_partition_run(IndexPartition *ip, const I *index, const Arg<uint8_t> &mask, size_t n, int n_buckets) {
    RoctxRange range("enoki-hip: index partition");
    Context &c = ctx();
    unsigned blocks = (unsigned) std::min<size_t>((size_t) c.num_cu * 4, (n + kTile - 1) / kTile);
    if (blocks == 0) blocks = 1;
    size_t chunk = (n + blocks - 1) / blocks;
    chunk = (chunk + kTile - 1) / kTile * kTile;
    blocks = (unsigned) ((n + chunk - 1) / chunk);
    const int vec_ok = aligned16(index) && arg_aligned(mask);
    int rep_shift = 0;
    while ((n_buckets << (rep_shift + 1)) <= kMaxBuckets && rep_shift < 4) ++rep_shift;
    const size_t count_entries = (size_t) n_buckets * blocks;
    if (int rc = ek_hip_malloc((count_entries + 2 * kMaxBuckets + 2) * sizeof(uint32_t), &ip->meta)) return rc;
    if (int rc = ek_hip_malloc((n ? n : 1) * sizeof(uint32_t), &ip->local)) return rc;
    uint32_t *counts = (uint32_t *) ip->meta, *row_total = counts + count_entries, *bucket_base = row_total + kMaxBuckets;
    hipLaunchKernelGGL((k_bin_count<I, Shift>), dim3(blocks), dim3(kThreads), 0, c.stream, counts, index, mask, n, chunk, n_buckets,
                       rep_shift, vec_ok);
    EK_LAUNCH_CHECK("index_partition_count", n, n * sizeof(I) + arg_bytes(mask, n));
    hipLaunchKernelGGL(k_bin_scan_rows, dim3(n_buckets), dim3(1024), 0, c.stream, counts, row_total, blocks);
    hipLaunchKernelGGL(k_bin_scan_buckets, dim3(1), dim3(256), 0, c.stream, bucket_base, (uint32_t *) nullptr,
                       (const uint32_t *) row_total, n_buckets, 0u);
    EK_LAUNCH_CHECK("index_partition_scan", count_entries, 2 * count_entries * sizeof(uint32_t));
    BinStreams<uint32_t, 1> st{};
    st.value[0] = Arg<uint32_t>{ nullptr, 0u, 0u };
    st.weight[0] = Arg<uint32_t>{ nullptr, 1u, 0u };
    st.pair_val[0] = nullptr;
    st.value_op[0] = EK_COPY;
    hipLaunchKernelGGL((k_bin_partition<uint32_t, I, Shift, uint32_t, 1, false, true>), dim3(blocks), dim3(kThreads), 0, c.stream,
                       (uint32_t *) ip->local, st, (const uint32_t *) counts, (const uint32_t *) bucket_base, index, mask, n, chunk,
                       n_buckets, 0, vec_ok);
    EK_LAUNCH_CHECK("index_partition", n, n * (sizeof(I) + sizeof(uint32_t)) + arg_bytes(mask, n));
    ip->info.bucket_base = bucket_base;
    ip->info.local = (const uint32_t *) ip->local;
    return EK_OK;
}

} // namespace ek

using namespace ek;

// A table beyond 256 buckets is cut into SLICES of 256 buckets: one object per slice, each made by a pass over (index, x) that
// keeps the indices of its slice (PagedOut::lo / span) -- S passes of 8 B/elt instead of one, still without a lookup that
// leaves the CU.  Reductions combine the slices' results (and add the masked-out lanes' terms once), the scatter_add of slice s
// goes to entries [s span, (s + 1) span) of its tables.
struct ek_hip_bucketed : ek::Bucketed {
    std::vector<ek_hip_bucketed *> slices;
    size_t slice_span = 0;
    bool slices_split = false;       // the slices hold disjoint parts of the input (CoarseSplit) instead of filtered views of all of it
    bool empty_slice = false;        // a slice that received no element
    ~ek_hip_bucketed() { for (ek_hip_bucketed *s : slices) delete s; }
};
constexpr int kMaxSlices = 64;
struct SliceCounts { const uint32_t *active[kMaxSlices]; const uint32_t *flag; };

template <typename T, int ROp>
__global__ __launch_bounds__(64) void k_slices_combine(T *__restrict__ out, const T *__restrict__ partial, int slices, SliceCounts counts,
                                                       size_t n, int map_op) {
    using R = ek::BucketReducer<ROp, T>;
    if (threadIdx.x != 0) return;
    T r = R::identity();
    size_t kept = 0;
    bool nonfinite = counts.flag && counts.flag[0];
    for (int s = 0; s < slices; ++s) { r = R::combine(r, partial[s]); kept += counts.active[s][0]; nonfinite = nonfinite || counts.active[s][1]; }
    out[0] = ek::bucket_dropped_lanes<T, ROp>(r, n - kept, nonfinite, map_op);
}

// (tables of three or more slices under a mask: the split by slice drops the masked-out lanes before any page partition sees
// them, so their x is looked at here -- 5 B/elt of a shape that already pays 20 B/elt for the split)
__global__ __launch_bounds__(256) void k_masked_nonfinite(uint32_t *__restrict__ flag, const uint32_t *__restrict__ xbits,
                                                          const uint8_t *__restrict__ mask, size_t n) {
    bool bad = false;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256)
        bad = bad || (!mask[i] && (xbits[i] & 0x7F800000u) == 0x7F800000u);
    if (bad) atomicOr(flag, 1u);
}
struct ek_hip_index_partition : ek::IndexPartition { };


extern "C" {

int ek_hip_bucketed_applicable(int type, int index_type, size_t table_size, size_t n) {
    if (type != EK_F32 && type != EK_F64) return 0;
    if (index_type != EK_U32 && index_type != EK_I32) return 0;
    if (ctx().tuning.deterministic || !ctx().tuning.bucket_ordered) return 0;
    const size_t bins = type == EK_F64 ? (size_t) bins_of<double> : (size_t) bins_of<float>;
    // (4-byte types go through pages whose numbers are packed with an element count: 2^30 elements at most; their tables may
    // have up to kMaxSlices slices of 256 half-size buckets)
    return n >= ((size_t) 1 << 18) && n < ((size_t) 1 << (type == EK_F64 ? 32 : 30)) && table_size > bins &&
           table_size <= (size_t) kMaxBuckets * bins * (type == EK_F64 ? 1 : kMaxSlices / 2);
}

int ek_hip_bucketed_pair_create(int type, int index_type, int op, const void *table_a, const void *table_c, size_t table_size,
                                const void *x, const void *index, size_t n, ek_hip_bucketed **out) {
    return ek_hip_bucketed_pair_create_hinted(type, index_type, op, table_a, table_c, table_size, x, index, n, 0u, out);
}

int ek_hip_bucketed_pair_create_hinted(int type, int index_type, int op, const void *table_a, const void *table_c, size_t table_size,
                                       const void *x, const void *index, size_t n, unsigned hints, ek_hip_bucketed **out) {
    return ek_hip_bucketed_pair_create_masked(type, index_type, op, table_a, table_c, table_size, x, index, nullptr, n, hints, out);
}

int ek_hip_bucketed_pair_create_masked(int type, int index_type, int op, const void *table_a, const void *table_c, size_t table_size,
                                       const void *x, const void *index, const uint8_t *mask, size_t n, unsigned hints,
                                       ek_hip_bucketed **out) {
    if (int rc = ensure_init()) return rc;
    if (!out || !table_a || !x || !index) return fail(EK_ERR_INVALID, "ek_hip_bucketed_pair_create(): null pointer");
    if (!table_c && op != EK_MULADD)
        return fail(EK_ERR_INVALID, "ek_hip_bucketed_pair_create(): a NULL addend table (the product gather(A, idx) * x alone) goes with EK_MULADD");
    *out = nullptr;
    if (op != EK_FMADD && op != EK_FMSUB && op != EK_FNMADD && op != EK_FNMSUB && op != EK_MULADD && op != EK_MULSUB && op != EK_NMULADD)
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_bucketed_pair_create(): op %d is neither of the fma family nor a product-then-sum", op);
    if (!ek_hip_bucketed_applicable(type, index_type, table_size, n))
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_bucketed_pair_create(): shape not covered (type %d, %zu lookups into %zu entries%s)",
                    type, n, table_size, ctx().tuning.deterministic ? ", deterministic mode" : "");
    ek_hip_bucketed *b = new ek_hip_bucketed();
    b->type = type; b->index_type = index_type; b->op = op;
    b->n = n; b->table_size = table_size;
    b->table_a = table_a; b->table_c = table_c;
    int rc;
    // valid int32 indices are non-negative: same bits as uint32
    // EK_BUCKETED_HINT_ADJOINT: buckets of half the size, so that a bucket's table slice AND its two gradient tables fit the LDS
    // together (k_bucket_pair_forward_adjoint) -- when the table still fits kMaxBuckets of those
    const size_t bins = type == EK_F64 ? (size_t) bins_of<double> : (size_t) bins_of<float>;
    const bool half = (hints & EK_BUCKETED_HINT_ADJOINT) && ctx().tuning.early_adjoint && table_size <= (size_t) kMaxBuckets * (bins / 2);
    if (type == EK_F32) {
        const Arg<uint8_t> m{ mask, 1, mask ? 1u : 0u };
        // half-size buckets (early adjoint) while the table needs at most two slices of them; beyond that the per-slice costs
        // (a partition launch, table slices staged and folded per piece) weigh more than the second pass over the lists: full-size
        // buckets halve the number of slices (K = 16 Mi, 64 Mi lookups, same box: 8 slices 53.8, 4 slices 57.0 Gelem/s)
        const bool want_half = (hints & EK_BUCKETED_HINT_ADJOINT) && ctx().tuning.early_adjoint &&
                               table_size <= (size_t) 2 * kMaxBuckets * (bins / 2);
        const size_t sel_bins = want_half ? bins / 2 : bins, span = (size_t) kMaxBuckets * sel_bins;
        if (table_size > span) {
            const int S = (int) ((table_size + span - 1) / span);
            b->slice_span = span;
            rc = EK_OK;
            // two slices: each reads all of (index, x) and keeps its own; three or more: split by slice first (needs the slice
            // populations on the host, which a captured step cannot wait for)
            CoarseSplit cs;
            const bool split = S >= 3;
            if (split) {
                rc = refuse_while_capturing("ek_hip_bucketed_pair_create(): a table of three or more slices (slice populations are read back)");
                if (rc == EK_OK)
                    rc = want_half ? coarse_split<bin_shift_of<float> - 1 + 8>(cs, (const float *) x, (const uint32_t *) index, m, n, S)
                                   : coarse_split<bin_shift_of<float> + 8>(cs, (const float *) x, (const uint32_t *) index, m, n, S);
            }
            if (split && rc == EK_OK && mask) {
                // the split drops the masked-out lanes: whether one of them carried a non-finite x is found out here
                rc = ek_hip_malloc(16, &b->meta);
                if (rc == EK_OK) {
                    EK_HIP_CHECK(hipMemsetAsync(b->meta, 0, 16, ctx().stream));
                    const unsigned grid = (unsigned) std::min<size_t>((n + 255) / 256, (size_t) ctx().num_cu * 8);
                    hipLaunchKernelGGL(k_masked_nonfinite, dim3(grid), dim3(256), 0, ctx().stream, (uint32_t *) b->meta, (const uint32_t *) x, mask, n);
                    EK_LAUNCH_CHECK("bucket_masked_nonfinite", n, n * 5);
                }
            }
            const Arg<uint8_t> all{ nullptr, 1, 0u };
            for (int sl = 0; sl < S && rc == EK_OK; ++sl) {
                ek_hip_bucketed *sub = new ek_hip_bucketed();
                b->slices.push_back(sub);
                sub->type = type; sub->index_type = index_type; sub->op = op;
                sub->table_size = std::min(span, table_size - (size_t) sl * span);
                sub->table_a = (const float *) table_a + (size_t) sl * span;
                sub->table_c = table_c ? (const float *) table_c + (size_t) sl * span : nullptr;
                sub->correct_masked = false;
                if (split) {
                    // the slice's own elements, indices already local to the slice
                    sub->n = cs.base[sl + 1] - cs.base[sl];
                    sub->win_lo = 0;
                    sub->win_span = (uint32_t) sub->table_size;
                    if (sub->n == 0) { sub->empty_slice = true; continue; }
                    rc = bucketed_create_paged<uint32_t>(sub, (const float *) cs.x + cs.base[sl], (const uint32_t *) cs.idx + cs.base[sl], all,
                                                         bin_shift_of<float> - (want_half ? 1 : 0));
                } else {
                    sub->n = n;
                    sub->win_lo = (uint32_t) ((size_t) sl * span);
                    sub->win_span = (uint32_t) sub->table_size;
                    rc = bucketed_create_paged<uint32_t>(sub, (const float *) x, (const uint32_t *) index, m, bin_shift_of<float> - (want_half ? 1 : 0));
                }
            }
            b->slices_split = split;
        } else {
            // EK_BUCKETED_HINT_BOUNDED on top of the adjoint hint: buckets of a QUARTER of the size (4 Ki entries), whose records and two planes
            // of 64-bit fixed-point sums fit the LDS (bucketed_early.hip) -- while the table is within kMaxBuckets of those
            const bool quarter = half && (hints & EK_BUCKETED_HINT_BOUNDED) && early_fixed_enabled() &&
                                 table_size <= (size_t) kMaxBuckets * (bins / 4);
            const int down = half ? (quarter ? 2 : 1) : 0;
            rc = bucketed_create_paged<uint32_t>(b, (const float *) x, (const uint32_t *) index, m, bin_shift_of<float> - down);
        }
    } else if (mask) {
        rc = fail(EK_ERR_UNSUPPORTED, "ek_hip_bucketed_pair_create_masked(): masks with 4-byte element types only");
    } else {
        if (half) rc = bucketed_create<double, uint32_t, bin_shift_of<double> - 1>(b, (const double *) x, (const uint32_t *) index);
        else rc = bucketed_create<double, uint32_t, bin_shift_of<double>>(b, (const double *) x, (const uint32_t *) index);
    }
    if (rc != EK_OK) { delete b; return rc; }
    *out = b;
    return EK_OK;
}

int ek_hip_bucketed_reduce(ek_hip_bucketed *b, int reduce_op, int map_op, void *out, int keep_values, int keep_op) {
    if (int rc = ensure_init()) return rc;
    if (!b || !out) return fail(EK_ERR_INVALID, "ek_hip_bucketed_reduce(): null pointer");
    if (map_op != EK_COPY && !unary_fusable(map_op))
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_bucketed_reduce(): op %d cannot be applied on load", map_op);
    if (!b->slices.empty()) {
        Scratch partial;
        if (int rc = partial.alloc(b->slices.size() * sizeof(float))) return rc;
        SliceCounts counts{};
        int S = 0;                                     // slices that hold elements
        for (ek_hip_bucketed *sub : b->slices) {
            if (sub->empty_slice) continue;
            if (int rc = bucketed_reduce<float>(sub, reduce_op, map_op, (float *) partial.ptr + S, keep_values != 0, keep_op)) return rc;
            counts.active[S++] = sub->active;
        }
        counts.flag = (const uint32_t *) b->meta;          // (tables of three or more slices under a mask; null otherwise)
        Context &c = ctx();
#define EK_COMBINE(OP) hipLaunchKernelGGL((k_slices_combine<float, OP>), dim3(1), dim3(64), 0, c.stream, (float *) out, (const float *) partial.ptr, S, counts, b->n, map_op)
        switch (reduce_op) {
            case EK_HSUM: EK_COMBINE(EK_HSUM); break;
            case EK_HPROD: EK_COMBINE(EK_HPROD); break;
            case EK_HMIN: EK_COMBINE(EK_HMIN); break;
            case EK_HMAX: EK_COMBINE(EK_HMAX); break;
            default: return fail(EK_ERR_INVALID, "ek_hip_bucketed_reduce(): unknown op %d", reduce_op);
        }
#undef EK_COMBINE
        EK_LAUNCH_CHECK("reduce_stage2", (size_t) S, (size_t) S * sizeof(float));
        return EK_OK;
    }
    if (b->type == EK_F32) return bucketed_reduce<float>(b, reduce_op, map_op, out, keep_values != 0, keep_op);
    return bucketed_reduce<double>(b, reduce_op, map_op, out, keep_values != 0, keep_op);
}

int ek_hip_bucketed_scatter_add(ek_hip_bucketed *b, int count, void *const *bases, const int *from_u, const int *map_ops,
                                const uint64_t *imm_bits, const int *weighted, const int *fresh) {
    return ek_hip_bucketed_scatter_add_scaled(b, count, bases, from_u, map_ops, imm_bits, weighted, fresh, nullptr);
}

int ek_hip_bucketed_early_pair(int map_op, int keep_op) { return early_pair_supported(map_op, keep_op) ? 1 : 0; }

int ek_hip_bucketed_scatter_add_scaled(ek_hip_bucketed *b, int count, void *const *bases, const int *from_u, const int *map_ops,
                                       const uint64_t *imm_bits, const int *weighted, const int *fresh, const uint64_t *scale_bits) {
    if (int rc = ensure_init()) return rc;
    if (!b || !bases || !from_u || !imm_bits || !weighted || count < 1 || count > 4)
        return fail(EK_ERR_INVALID, "ek_hip_bucketed_scatter_add(): bad arguments");
    for (int s = 0; s < count; ++s) {
        if (!bases[s]) return fail(EK_ERR_INVALID, "ek_hip_bucketed_scatter_add(): null table");
        if (from_u[s] && map_ops && map_ops[s] != EK_COPY && !unary_fusable(map_ops[s]))
            return fail(EK_ERR_UNSUPPORTED, "ek_hip_bucketed_scatter_add(): op %d cannot be applied on load", map_ops[s]);
    }
    if (!b->slices.empty()) {
        for (size_t sl = 0; sl < b->slices.size(); ++sl) {
            void *sb[4];
            for (int s = 0; s < count; ++s) sb[s] = (float *) bases[s] + sl * b->slice_span;
            if (b->slices[sl]->empty_slice) {          // nothing to add; a fresh table still has to hold zeros there
                for (int s = 0; s < count; ++s)
                    if (fresh && fresh[s])
                        EK_HIP_CHECK(hipMemsetAsync(sb[s], 0, b->slices[sl]->table_size * sizeof(float), ctx().stream));
                continue;
            }
            if (int rc = bucketed_scatter_add<float>(b->slices[sl], count, sb, from_u, map_ops, imm_bits, weighted, fresh, scale_bits)) return rc;
        }
        return EK_OK;
    }
    if (b->type == EK_F32) return bucketed_scatter_add<float>(b, count, bases, from_u, map_ops, imm_bits, weighted, fresh, scale_bits);
    return bucketed_scatter_add<double>(b, count, bases, from_u, map_ops, imm_bits, weighted, fresh, scale_bits);
}

int ek_hip_bucketed_destroy(ek_hip_bucketed *b) {
    delete b;
    return EK_OK;
}

int ek_hip_index_partition_create(int index_type, const void *index, const ek_operand *mask, size_t n, size_t range,
                                  ek_hip_index_partition **out) {
    if (int rc = ensure_init()) return rc;
    if (!out || !index || !mask) return fail(EK_ERR_INVALID, "ek_hip_index_partition_create(): null pointer");
    *out = nullptr;
    if (index_type != EK_U32 && index_type != EK_I32)
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_index_partition_create(): 32-bit index arrays only");
    if (n == 0 || n >= ((size_t) 1 << 32) || range == 0 || range > ((size_t) kMaxBuckets << 19))
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_index_partition_create(): %zu indices into %zu entries is out of range", n, range);
    Arg<uint8_t> m;
    if (int rc = make_arg<uint8_t>(mask, n, m, "ek_hip_index_partition_create")) return rc;
    // the smallest bucket size of {4 Ki, 16 Ki, 128 Ki, 512 Ki} entries that covers the range with <= 256 buckets
    static const int shifts[] = { 12, 14, 17, 19 };
    int shift = 19;
    for (int sft : shifts)
        if (((range + ((size_t) 1 << sft) - 1) >> sft) <= (size_t) kMaxBuckets) { shift = sft; break; }
    ek_hip_index_partition *ip = new ek_hip_index_partition();
    ip->info.shift = shift;
    ip->info.n_buckets = (int) ((range + ((size_t) 1 << shift) - 1) >> shift);
    ip->info.n = n;
    ip->info.range = range;
    int rc;
    const uint32_t *idx = (const uint32_t *) index;          // valid int32 indices are non-negative: same bits as uint32
    // large inputs over many buckets: the single-pass page partition (ENOKI_HIP_INDEX_PAGED=0: count / scan / partition as before)
    static const bool paged = [] { const char *e = getenv("ENOKI_HIP_INDEX_PAGED"); return !e || atoi(e) != 0; }();
    if (paged && ctx().tuning.bucket_ordered && n >= ((size_t) 1 << 20) && n < ((size_t) 1 << 30) && ip->info.n_buckets >= 32) {
        rc = index_partition_run_paged(ip, idx, m, n, ip->info.n_buckets, shift);
        if (rc == EK_OK) { *out = ip; return EK_OK; }
        if (rc != EK_ERR_UNSUPPORTED && rc != EK_ERR_OOM) { delete ip; return rc; }
        (void) hipGetLastError();
        for (void **q : { &ip->meta, &ip->local, &ip->lists }) { if (*q) ek_hip_free(*q); *q = nullptr; }
        ip->info.page_shift = 0;
    }
    switch (shift) {
        case 12: rc = index_partition_run<uint32_t, 12>(ip, idx, m, n, ip->info.n_buckets); break;
        case 14: rc = index_partition_run<uint32_t, 14>(ip, idx, m, n, ip->info.n_buckets); break;
        case 17: rc = index_partition_run<uint32_t, 17>(ip, idx, m, n, ip->info.n_buckets); break;
        default: rc = index_partition_run<uint32_t, 19>(ip, idx, m, n, ip->info.n_buckets); break;
    }
    if (rc != EK_OK) { delete ip; return rc; }
    *out = ip;
    return EK_OK;
}

int ek_hip_index_partition_get(const ek_hip_index_partition *p, ek_hip_index_partition_info *info) {
    if (!p || !info) return fail(EK_ERR_INVALID, "ek_hip_index_partition_get(): null pointer");
    *info = p->info;
    return EK_OK;
}

int ek_hip_index_partition_destroy(ek_hip_index_partition *p) {
    delete p;
    return EK_OK;
}

} // extern "C"
