// Shared by the translation units of the bucket-ordered path (bucketed.hip: partitions, forward, adjoint, the C ABI;
// bucketed_early.hip: the forward + adjoint kernel of the headline step): where a bucket's elements are and how a piece is
// walked, the table slice in the LDS, the exchange locks of the gradient sums, the reduction finished by the last workgroup,
// and the host-side object.  Split out of bucketed.hip in round 6 so that a change to one kernel family recompiles one unit.
#pragma once
#include <mutex>
#include <unordered_map>
#include <cstdlib>
#include <cstring>
#include "ek_binned.h"
#include "ek_paged.h"
#include <limits>

namespace ek {

template <typename T> struct alignas(2 * sizeof(T)) PairRec { T a, c; };

// The bucket's slice of both tables as interleaved {a, c} records in the LDS (every element then costs ONE ds_read_b64; b128 for
// doubles), and -- ZeroTables -- the two gradient tables behind them cleared.  A thread requests all of its entries before it
// uses the first: the loop used to wait for each pair of loads (Bins / 1024 = 8 or 16 dependent round trips per piece, 6-10 us
// of every launch).  Entries beyond the table read its last entry and count as zero.
/// max_bits (optional, [2]): max |a|, max |c| over this thread's share of the slice as bit patterns (a NaN or an infinity comes out on top)
template <typename T, bool ZeroTables>
__device__ __forceinline__ void stage_pair_slice(PairRec<T> *rec, T *tables, const T *__restrict__ table_a, const T *__restrict__ table_c,
                                                 size_t first, size_t table_size, int Bins, int flip_a, int flip_c, uint32_t *max_bits = nullptr) {
    constexpr int U = 8;
    const size_t last = table_size - 1;
    for (int j0 = threadIdx.x; j0 < Bins; j0 += U * (int) blockDim.x) {
        T a[U], c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = first + (size_t) (j0 + u * (int) blockDim.x);
            const size_t kc = k < last ? k : last;
            a[u] = table_a[kc];
            c[u] = table_c ? table_c[kc] : T(-0.0);       // no addend table: u = a x + (-0) = a x, bit for bit (signed zeros included)
            if (k > last) { a[u] = T(0); c[u] = T(0); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * (int) blockDim.x;
            if (j < Bins) {
                rec[j] = PairRec<T>{ flip_a ? -a[u] : a[u], flip_c ? -c[u] : c[u] };
                if constexpr (sizeof(T) == 4) {
                    if (max_bits) {
                        max_bits[0] = max(max_bits[0], __float_as_uint((float) a[u]) & 0x7FFFFFFFu);
                        max_bits[1] = max(max_bits[1], __float_as_uint((float) c[u]) & 0x7FFFFFFFu);
                    }
                }
                if constexpr (ZeroTables) { tables[2 * j] = T(0); tables[2 * j + 1] = T(0); }
            }
        }
    }
}

constexpr int kBucketThreads = 1024;        // one workgroup per CU: the {A, C} slice / two gradient tables fill 128 KiB of LDS
constexpr int kBucketWaves = kBucketThreads / 64;
enum { EK_REDUCE_NONE = EK_REDUCE_COUNT };   // forward kernel without a reduction: only keeps u in bucket order

template <int Op, typename T> struct BucketReducer {
    static __device__ __host__ __forceinline__ T identity() {
        if constexpr (Op == EK_HSUM || Op == EK_REDUCE_NONE) return T(0);
        else if constexpr (Op == EK_HPROD) return T(1);
        else return std::numeric_limits<T>::quiet_NaN();           // minNum / maxNum (reduce.hip)
    }
    static __device__ __forceinline__ T combine(T acc, T v) {
        if constexpr (Op == EK_HSUM) return acc + v;
        else if constexpr (Op == EK_HPROD) return acc * v;
        else if constexpr (Op == EK_HMIN) { if constexpr (sizeof(T) == 4) return __builtin_fminf(acc, v); else return __builtin_fmin(acc, v); }
        else if constexpr (Op == EK_HMAX) { if constexpr (sizeof(T) == 4) return __builtin_fmaxf(acc, v); else return __builtin_fmax(acc, v); }
        else return acc;
    }
};

/// where the last workgroup of a reducing launch puts the result (bucket_finish); ticket == nullptr: a separate launch does it
template <typename T> struct BucketFinish {
    uint32_t *ticket;
    T *out;
    const uint32_t *active;
    size_t n;
    int zero_op;             // what a dropped lane (u = 0) contributes, as a function of 0
    uint32_t *counters = nullptr;   // the object's block of partition counters (gtotal): cleared by the last workgroup for the block's next user
};

template <typename T> __device__ __forceinline__ T bucket_shfl_down(T v, int delta) {
    if constexpr (sizeof(T) == 8) {
        uint64_t u;
        __builtin_memcpy(&u, &v, 8);
        uint32_t lo = (uint32_t) u, hi = (uint32_t) (u >> 32);
        lo = __shfl_down(lo, delta, 64);
        hi = __shfl_down(hi, delta, 64);
        u = ((uint64_t) hi << 32) | lo;
        __builtin_memcpy(&v, &u, 8);
        return v;
    } else {
        return __shfl_down(v, delta, 64);
    }
}

template <typename T> __device__ __forceinline__ T fma_t(T a, T b, T c) {
    if constexpr (sizeof(T) == 4) return __builtin_fmaf(a, b, c); else return __builtin_fma(a, b, c);
}
// u of one element: fma(a, x, c), or -- `a * x + c` written with operators, EK_MULADD / EK_MULSUB / EK_NMULADD -- the product and
// the sum with a rounding each (this translation unit is built with -ffp-contract=off).  `two` is uniform over the launch: both
// forms are two or three vector instructions next to the ~50 of the function that follows, so it is a select, not a kernel.
template <typename T> __device__ __forceinline__ T pair_value(T a, T x, T c, int two) {
    const T p = a * x;
    const T s = p + c, f = fma_t(a, x, c);
    return two ? s : f;
}

// ---- where a bucket's elements are ----------------------------------------------------------------------------------
// Two layouts of the bucket-ordered lists (l16, x_b and what is kept next to them):
//   contiguous  (PS = 0; count / scan / partition of ek_binned.h -- 8-byte element types): bucket b owns positions
//               [base[b], base[b + 1]), a piece is a sub-range that starts anywhere
//   paged       (PS = 5 | 6; the single-pass partition of ek_paged.h -- 4-byte element types): bucket b owns the pages
//               glist_full[base[b] .. base[b + 1]) (2^PS elements each, complete) and glist_part[base_part[b] .. base_part[b + 1])
//               (page << 6 | count - 1); a piece is a range of both lists
struct BucketLists {
    const uint32_t *base, *piece_prefix;
    const uint32_t *base_part, *glist_full, *glist_part;
    int n_buckets;
};

struct PieceRange {
    size_t begin, end;               // contiguous
    uint32_t f0, f1, p0, p1;         // paged
    uint32_t pieces;                 // of the bucket this piece belongs to
};

// piece `blockIdx.x` -> its bucket and its share of the bucket's elements (the same cut as k_bin_accumulate)
template <int PS>
__device__ __forceinline__ bool bucket_piece(const BucketLists &bl, int &bucket, PieceRange &r) {
    // ONE round trip to global memory: every thread asks for everything its candidate bucket would need (prefix, bases) at once and
    // the thread whose bucket holds this piece publishes it -- the search, then the prefix of the bucket found, then its bases were
    // four dependent round trips, ~2.4 us at the start of every workgroup (profiles/probe_early_phases_r06.txt: "which piece am I")
    __shared__ uint32_t s_piece[8];
    const int n_buckets = bl.n_buckets;
    const uint32_t total = bl.piece_prefix[n_buckets];
    for (int b = threadIdx.x; b < n_buckets; b += blockDim.x) {
        const uint32_t pp0 = bl.piece_prefix[b], pp1 = bl.piece_prefix[b + 1], lo = bl.base[b], hi = bl.base[b + 1];
        const uint32_t plo = PS ? bl.base_part[b] : 0u, phi = PS ? bl.base_part[b + 1] : 0u;
        if (pp0 <= blockIdx.x && blockIdx.x < pp1) {
            s_piece[0] = (uint32_t) b; s_piece[1] = pp0; s_piece[2] = pp1; s_piece[3] = lo; s_piece[4] = hi; s_piece[5] = plo; s_piece[6] = phi;
        }
    }
    if (blockIdx.x >= total) return false;
    __syncthreads();
    bucket = (int) s_piece[0];
    const size_t q = blockIdx.x - s_piece[1], pieces = s_piece[2] - s_piece[1];
    auto cut = [&](size_t lo, size_t hi, size_t &b0, size_t &b1) {
        const size_t per = (hi - lo + pieces - 1) / pieces;
        b0 = lo + q * per < hi ? lo + q * per : hi;
        b1 = b0 + per < hi ? b0 + per : hi;
    };
    r = PieceRange{};
    r.pieces = (uint32_t) pieces;
    if constexpr (PS == 0) {
        cut(s_piece[3], s_piece[4], r.begin, r.end);
    } else {
        size_t a, b;
        cut(s_piece[3], s_piece[4], a, b);
        r.f0 = (uint32_t) a; r.f1 = (uint32_t) b;
        cut(s_piece[5], s_piece[6], a, b);
        r.p0 = (uint32_t) a; r.p1 = (uint32_t) b;
    }
    return true;
}

// Walks a piece.  Body:
//   struct Step                                what a lane holds of V vectors of four consecutive elements
//   fetch(Step &, int h, size_t pos)           requests vector h at element position pos (16-byte aligned)
//   apply(const Step &)                        consumes the V vectors
//   one(size_t pos, bool on, int slot)         one element; EVERY lane of a wave calls it together (on = false: no element)
// Contiguous: up to 3 leading elements one per lane, then vectors with the loads of step i + 1 requested before step i is
// consumed, then the tail one element per lane.  Paged: 2^PS / 4 lanes share a page, four elements each -- the same 8- and
// 16-byte vector loads; the pages' numbers are requested two steps ahead; what does not fill a step of the whole workgroup
// (the last complete pages, the partially filled ones) goes element by element under a lane predicate.
#ifdef EK_EARLY_TIMING
template <typename Body> __device__ __forceinline__ auto walk_mark(Body &b, int k) -> decltype(b.mark(k)) { return b.mark(k); }
__device__ __forceinline__ void walk_mark(...) { }
#define EK_WALK_MARK(k) walk_mark(body, k)
#else
#define EK_WALK_MARK(k) do { } while (0)
#endif

template <int PS, int V, typename Body>
__device__ __forceinline__ void walk_piece(const BucketLists &bl, const PieceRange &r, Body &body) {
    using Step = typename Body::Step;
    if constexpr (PS == 0) {
        const size_t begin = r.begin, end = r.end;
        const size_t head_end = ((begin + 3) & ~(size_t) 3) < end ? ((begin + 3) & ~(size_t) 3) : end;
        body.one(begin + threadIdx.x, begin + threadIdx.x < head_end, 0);
        constexpr size_t kStep = (size_t) 4 * V * kBucketThreads;
        size_t base = head_end;
        if (base + kStep <= end) {
            Step cur, next;
#pragma unroll
            for (int h = 0; h < V; ++h) body.fetch(cur, h, base + (size_t) h * (kStep / V) + (size_t) threadIdx.x * 4);
            for (; base + 2 * kStep <= end; base += kStep) {
#pragma unroll
                for (int h = 0; h < V; ++h) body.fetch(next, h, base + kStep + (size_t) h * (kStep / V) + (size_t) threadIdx.x * 4);
                body.apply(cur);
                cur = next;
            }
            body.apply(cur);
            base += kStep;
        }
        for (; base < end; base += kBucketThreads) body.one(base + threadIdx.x, base + threadIdx.x < end, 0);
    } else {
        constexpr uint32_t LX = (1u << PS) / 4, G = kBucketThreads / LX;        // lanes per page, pages per vector of the workgroup
        const uint32_t g = threadIdx.x / LX, i = threadIdx.x % LX;
        const uint32_t nfull = r.f1 - r.f0, steps = nfull / (G * V);
        auto at = [&](uint32_t page) { return ((size_t) page << PS) + 4 * i; };
        if (steps) {
            const uint32_t *list = bl.glist_full + r.f0 + g;
            uint32_t e1[V], e2[V];
            Step cur, next;
#pragma unroll
            for (int h = 0; h < V; ++h) e1[h] = list[h * G];
#pragma unroll
            for (int h = 0; h < V; ++h) body.fetch(cur, h, at(e1[h]));
#pragma unroll
            for (int h = 0; h < V; ++h) e1[h] = steps > 1 ? list[(V + h) * G] : 0u;
            for (uint32_t s = 0; s + 1 < steps; ++s) {
#pragma unroll
                for (int h = 0; h < V; ++h) e2[h] = s + 2 < steps ? list[((s + 2) * V + h) * G] : 0u;
#pragma unroll
                for (int h = 0; h < V; ++h) body.fetch(next, h, at(e1[h]));
                body.apply(cur);
                cur = next;
#pragma unroll
                for (int h = 0; h < V; ++h) e1[h] = e2[h];
            }
            body.apply(cur);
        }
        EK_WALK_MARK(0);
        // wave-uniform trip counts: the lock protocol inside one() wants every lane of a wave to come along
        const uint32_t done = steps * G * V, rest = nfull - done;
        for (uint32_t r0 = 0; r0 < rest; r0 += G) {
            const bool valid = r0 + g < rest;
            const uint32_t page = valid ? bl.glist_full[r.f0 + done + r0 + g] : 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) body.one(at(page) + j, valid, j);
        }
        EK_WALK_MARK(1);
        const uint32_t nparts = r.p1 - r.p0;
        for (uint32_t r0 = 0; r0 < nparts; r0 += G) {
            const bool valid = r0 + g < nparts;
            const uint32_t e = valid ? bl.glist_part[r.p0 + r0 + g] : 0u, count = valid ? (e & 63u) + 1u : 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) body.one(at(e >> 6) + j, 4 * i + j < count, j);
        }
    }
}

// `active` (may be null): [0] number of elements in the lists, [1] != 0 when a lane whose mask bit is clear carries a non-finite x.
// The other n - active[0] lanes were dropped by the partition -- masked out of both gathers, or pointing outside the table (the
// reference leaves that case unspecified, cuda.h:845-905; here it counts as masked out).  Their u is fma(0, x, 0): 0 for a finite
// x, so they contribute map_op(0) to the reduction; NaN for an infinite or NaN x -- then the reference's result is NaN
// (dynamic.h:632-650 sums every lane) and so is this one (hsum, hprod; hmin / hmax skip NaNs here as everywhere: DESIGN section 5).
template <typename T, int ROp>
__device__ __forceinline__ T bucket_dropped_lanes(T r, size_t masked, bool nonfinite, int map_op) {
    using R = BucketReducer<ROp, T>;
    if (masked) {
        const T f0 = unary_fused<T>(map_op, T(0));
        if constexpr (ROp == EK_HSUM) {
            r = r + (T) masked * f0;
        } else if constexpr (ROp == EK_HPROD) {
            T p = T(1), base = f0;
            for (size_t e = masked; e; e >>= 1) { if (e & 1) p = p * base; base = base * base; }
            r = r * p;
        } else {
            r = R::combine(r, f0);
        }
    }
    if (nonfinite) r = R::combine(r, std::numeric_limits<T>::quiet_NaN());
    return r;
}

// The reduction over the pieces' partial results, finished by the LAST workgroup to arrive instead of a launch of its own
// (k_bucket_reduce_final: 4.5 us + a launch gap per step; a quarter of what an 8 Mi-element shard step spends outside its two
// big kernels).  Every workgroup publishes its partial and takes a ticket; the one that draws the last ticket reads all
// partials (agent-scope loads: they were written on other XCDs), adds what the dropped lanes contribute and resets the ticket
// for the next launch -- launches on one object are ordered by the stream.  `ticket` == nullptr: the host launches
// k_bucket_reduce_final (objects whose meta block is not zero-filled: 8-byte element types).
/// The finish ticket of a launch, drawn by ONE thread of every workgroup after its partials are in memory; true for the workgroup that
/// draws the last one.  A RELAXED agent-scope atomic behind a wait for the wave's outstanding stores (the partials are agent-scope
/// stores, written through to memory; the last workgroup reads them with agent-scope loads): an acquire-release ticket is that wait
/// PLUS a write-back of the XCD's whole L2 -- the tables of 32 workgroups, which only the next launch reads -- and an invalidate:
/// 4.5 us per workgroup on average, 14-18 us for the unluckiest one, at the end of every launch; the forward + adjoint kernel went
/// 107-111 -> 99-100 us at 64 Mi elements, 41.8 -> 35.2 in an 8 Mi shard, same call (profiles/probe_early_fixed_phases_r06.txt).
/// A two-level ticket (groups of 32 workgroups) on top measured no better: what remains of the wait is the latency of the wave's
/// own table stores.  -DEK_FINISH_ACQ_REL restores the fenced form for A/B runs.
__device__ __forceinline__ bool finish_ticket(uint32_t *__restrict__ ticket) {
#ifdef EK_FINISH_ACQ_REL
    return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
#endif
}
__device__ __forceinline__ void finish_ticket_reset(uint32_t *__restrict__ ticket) {
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T, int ROp>
__device__ __forceinline__ void bucket_finish(T block_result /* thread 0 */, T *__restrict__ partials, uint32_t *__restrict__ ticket,
                                              T *__restrict__ out, const uint32_t *__restrict__ active, size_t n, int map_op,
                                              T *wave_part /* [kBucketWaves] shared */, uint32_t *__restrict__ counters = nullptr) {
    using R = BucketReducer<ROp, T>;
    using Bits = std::conditional_t<sizeof(T) == 4, uint32_t, unsigned long long>;
    __shared__ uint32_t s_last;
    if (threadIdx.x == 0) {
        Bits b;
        __builtin_memcpy(&b, &block_result, sizeof(T));
        __hip_atomic_store(reinterpret_cast<Bits *>(partials) + blockIdx.x, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket && finish_ticket(ticket);
    }
    __syncthreads();
    if (!s_last) return;
    // (the partition's page totals and accumulators have been consumed by the directory launch: cleared here, the block of counters
    // can go to the next object without a fill -- MetaRing)
    if (counters) {
        // (page totals [2][kMaxBuckets], then the accumulators: elements kept, non-finite flag, the ticket -- stored as 0 again below --,
        //  max |x|)
        for (unsigned k = threadIdx.x; k < kPgTotalsWords + 4u; k += blockDim.x) counters[k] = 0u;
        for (unsigned k = kPgMetaBase + kPgMetaAccumRep + threadIdx.x; k < kPgMetaBase + kPgMetaAccumRepEnd; k += blockDim.x) counters[k] = 0u;
    }
    T v = R::identity();
    for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x) {
        const Bits b = __hip_atomic_load(reinterpret_cast<const Bits *>(partials) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        T p;
        __builtin_memcpy(&p, &b, sizeof(T));
        v = R::combine(v, p);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = R::combine(v, bucket_shfl_down(v, d));
    __syncthreads();                                  // (wave_part may still hold this workgroup's own wave results)
    if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        v = threadIdx.x < blockDim.x / 64 ? wave_part[threadIdx.x] : R::identity();
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) v = R::combine(v, bucket_shfl_down(v, d));
        if (threadIdx.x == 0) {
            out[0] = bucket_dropped_lanes<T, ROp>(v, active ? n - (size_t) active[0] : 0, active && active[1], map_op);
            finish_ticket_reset(ticket);
        }
    }
}

template <typename T, int ROp>
__global__ __launch_bounds__(256) void k_bucket_reduce_final(T *__restrict__ out, const T *__restrict__ partials, unsigned count,
                                                             const uint32_t *__restrict__ active, size_t n, int map_op) {
    using R = BucketReducer<ROp, T>;
    __shared__ T wave_part[4];
    T v = R::identity();
    for (unsigned i = threadIdx.x; i < count; i += 256) v = R::combine(v, partials[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = R::combine(v, bucket_shfl_down(v, d));
    if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        T r = R::combine(R::combine(wave_part[0], wave_part[1]), R::combine(wave_part[2], wave_part[3]));
        out[0] = bucket_dropped_lanes<T, ROp>(r, active ? n - (size_t) active[0] : 0, active && active[1], map_op);
    }
}

// ---- exchange locks on {sum0, sum1} pairs ---------------------------------------------------------------------
// Two f32 tables share ONE lock per bin: the LDS holds {table 0, table 1} pairs and a bin pair is claimed by a 64-bit
// exchange (ds_wrxchg_rtn_b64), updated and released by one 64-bit store -- half the LDS atomics and half the dependent
// round trips of two independent 32-bit locks.  Same protocol as lds_add (ek_binned.h), see there for why the retry loop
// is wave-uniform.
constexpr unsigned long long kLockedPair = 0xFFC00001FFC00001ull;

__device__ __forceinline__ unsigned long long pair_sum(unsigned long long old, float v0, float v1) {
    const float s0 = __uint_as_float((unsigned) old) + v0, s1 = __uint_as_float((unsigned) (old >> 32)) + v1;
    // a pair is LOCKED iff its low word carries the lock pattern (a NaN payload no sum produces): one compare per test, and
    // the pattern is never published as a value
    unsigned lo = __float_as_uint(s0);
    if (lo == kLockedBits) lo = 0x7FC00000u;
    return (unsigned long long) lo | ((unsigned long long) __float_as_uint(s1) << 32);
}

__device__ __forceinline__ void lds_add_pair(unsigned long long *p, float v0, float v1, bool active) {
    bool pending = active;
    if (pending) {
        const unsigned long long old = atomicExch(p, kLockedPair);
        if ((unsigned) old != kLockedBits) {
            __hip_atomic_store(p, pair_sum(old, v0, v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pending = false;
        }
    }
    const unsigned key = (unsigned) (uintptr_t) p;
    const int lane = threadIdx.x & 63;
    while (__any(pending)) {
        const unsigned long long pend = __ballot(pending);
        const int leader = __ffsll((long long) pend) - 1;
        const unsigned leader_key = __shfl(key, leader);
        const bool grouped = pending && key == leader_key;
        float t0 = grouped ? v0 : 0.0f, t1 = grouped ? v1 : 0.0f;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { t0 += __shfl_xor(t0, d); t1 += __shfl_xor(t1, d); }
        if (lane == leader) {
            unsigned long long old;
            do { old = atomicExch(p, kLockedPair); } while ((unsigned) old == kLockedBits);
            __hip_atomic_store(p, pair_sum(old, t0, t1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        pending = pending && !grouped;
    }
}

// One update per lane, the form the streaming loops use: claim, add, release; the lanes that met a lock (another wave's, or a
// lane of this wave with the same bin) retry together once, what is still locked then takes the combining path above.  The
// "did anybody meet a lock" tests are wave ballots consumed by scalar branches (no vector instruction spent on them).
__device__ __forceinline__ void lds_add_pair_one(unsigned long long *table, uint32_t l, float v0, float v1) {
    unsigned long long *p = table + l;
    const unsigned long long old = atomicExch(p, kLockedPair);
    unsigned lo = (unsigned) old;
    asm volatile("" : "+v"(lo));                       // (keeps the lock test a 32-bit compare of the low word)
    const unsigned long long met = __builtin_amdgcn_uicmp(lo, kLockedBits, 32 /* == */);     // lanes that met a lock
    if (lo != kLockedBits) __hip_atomic_store(p, pair_sum(old, v0, v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (met) {
        bool pending = lo == kLockedBits;
        if (pending) {
            const unsigned long long again = atomicExch(p, kLockedPair);
            if ((unsigned) again != kLockedBits) {
                __hip_atomic_store(p, pair_sum(again, v0, v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pending = false;
            }
        }
        if (__builtin_amdgcn_ballot_w64(pending)) lds_add_pair(p, v0, v1, pending);
    }
}

// The same update WITHOUT a branch: claim, add, release -- a lane that met a lock stores its (meaningless) sum to a dummy slot
// and reports the update as still to be made.  A step's updates then form one basic block (the compiler overlaps an exchange's
// round trip with the next element's arithmetic), and what met a lock is retried once per step, all slots in flight together
// (lds_add_pair_retry).
__device__ __forceinline__ bool lds_try_add_pair(unsigned long long *table, unsigned long long *dummy, uint32_t l, float v0, float v1) {
    unsigned long long *p = table + l;
    const unsigned long long old = atomicExch(p, kLockedPair);
    const bool met = (unsigned) old == kLockedBits;
    __hip_atomic_store(met ? dummy : p, pair_sum(old, v0, v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return met;
}

// the slots of `pending` once more, their exchanges in flight together (the locks they met have been released long since); what
// is locked even then -- hot bins, or two slots of one lane with the same bin -- goes one slot at a time with wave combining
template <int N>
__device__ __forceinline__ void lds_add_pair_retry(unsigned long long *table, const uint32_t (&l)[N], const float (&v0)[N],
                                                   const float (&v1)[N], unsigned pending) {
    unsigned long long old[N];
#pragma unroll
    for (int j = 0; j < N; ++j)
        if ((pending >> j) & 1u) old[j] = atomicExch(table + l[j], kLockedPair);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (((pending >> j) & 1u) && (unsigned) old[j] != kLockedBits) {
            __hip_atomic_store(table + l[j], pair_sum(old[j], v0[j], v1[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pending &= ~(1u << j);
        }
    }
    if (__builtin_amdgcn_ballot_w64(pending != 0)) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (__builtin_amdgcn_ballot_w64((pending >> j) & 1u)) lds_add_pair(table + l[j], v0[j], v1[j], (pending >> j) & 1u);
    }
}

// N independent updates per lane: all N bins are CLAIMED first (N exchanges in flight -- one LDS round trip instead of N
// dependent ones), then every claim that succeeded is added to and released; the few that met a lock (another lane's, or
// this lane's own claim of the same bin in an earlier slot) are retried together, and what is still locked then goes through
// the one-at-a-time path with wave-level combining.  Nobody spins while holding a lock, so there is no circular wait.
// The kernels below call it with N = 1 (see bucket_accumulate_stream for the measurement that retired N = 4 / 8); what they
// keep from it is the cheap retry round before the wave-combining loop.
template <int N>
__device__ __forceinline__ void lds_add_pair_batch(unsigned long long *table, const uint32_t (&l)[N], const float (&v0)[N],
                                                   const float (&v1)[N]) {
    unsigned long long old[N];
    unsigned pending = 0;
    // round 1: every slot
#pragma unroll
    for (int j = 0; j < N; ++j) old[j] = atomicExch(table + l[j], kLockedPair);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if ((unsigned) old[j] != kLockedBits)
            __hip_atomic_store(table + l[j], pair_sum(old[j], v0[j], v1[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            pending |= 1u << j;
    }
    // With 64 N claims in flight per wave and 16 Ki bins a few claims per batch meet a lock (64 N = 512: ~3 % of them), so
    // "somebody in the wave is pending" is the rule, not the exception.  Round 2 retries exactly those slots, again all in
    // flight together -- their locks were released by the stores above -- and leaves only true repeat offenders (hot
    // bins) to the one-at-a-time path with its wave-level combining.
    if (__any(pending != 0)) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if ((pending >> j) & 1u) old[j] = atomicExch(table + l[j], kLockedPair);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (((pending >> j) & 1u) && (unsigned) old[j] != kLockedBits) {
                __hip_atomic_store(table + l[j], pair_sum(old[j], v0[j], v1[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pending &= ~(1u << j);
            }
        }
        if (__any(pending != 0)) {
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (__any((pending >> j) & 1u)) lds_add_pair(table + l[j], v0[j], v1[j], (pending >> j) & 1u);
        }
    }
}

template <typename T, int N>
__device__ __forceinline__ void lds_add_batch(T *table, const uint32_t (&l)[N], const T (&v)[N]) {
    if constexpr (std::is_same_v<T, float>) {
        unsigned *t = reinterpret_cast<unsigned *>(table);
        unsigned old[N];
        unsigned pending = 0;
        auto sum = [](unsigned o, float x) {
            unsigned bits = __float_as_uint(__uint_as_float(o) + x);
            return bits == kLockedBits ? 0x7FC00000u : bits;
        };
#pragma unroll
        for (int j = 0; j < N; ++j) old[j] = atomicExch(t + l[j], kLockedBits);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (old[j] != kLockedBits) __hip_atomic_store(t + l[j], sum(old[j], v[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else pending |= 1u << j;
        }
        if (__any(pending != 0)) {
#pragma unroll
            for (int j = 0; j < N; ++j)
                if ((pending >> j) & 1u) old[j] = atomicExch(t + l[j], kLockedBits);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (((pending >> j) & 1u) && old[j] != kLockedBits) {
                    __hip_atomic_store(t + l[j], sum(old[j], v[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pending &= ~(1u << j);
                }
            }
            if (__any(pending != 0)) {
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (__any((pending >> j) & 1u)) lds_add<true>(table + l[j], v[j], ((pending >> j) & 1u) != 0);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) lds_add<true>(table + l[j], v[j], true);
    }
}

/// which {reduced op, kept op} pairs the forward + adjoint kernel is built for
__host__ __device__ constexpr bool early_pair_supported(int map_op, int keep_op) {
    if ((map_op == EK_SIN && keep_op == EK_COS) || (map_op == EK_COS && keep_op == EK_SIN)) return true;
    if (map_op == EK_LOG && keep_op == EK_RCP) return true;
    if (map_op == EK_SQRT && keep_op == EK_RSQRT) return true;
    if (map_op == EK_RCP && keep_op == EK_RCP_SQR) return true;
    if (map_op == EK_RSQRT && keep_op == EK_RSQRT_CUBE) return true;
    return map_op == keep_op && (map_op == EK_SIN || map_op == EK_COS || map_op == EK_EXP || map_op == EK_SQRT || map_op == EK_RCP ||
                                 map_op == EK_RSQRT || map_op == EK_LOG || map_op == EK_ABS || map_op == EK_NEG);
}

// ---- 64-bit fixed-point adjoint sums (bucketed_early.hip forms them, k_fold_fixed_pieces folds them) ---------------------------
struct FixedScale { double upd /* 2^S */; float back /* 2^-S */; };

__host__ __device__ inline float pow2f(int e) {                     // 2^e, -126 <= e <= 127
    const uint32_t bits = (uint32_t) (e + 127) << 23;
    float f;
    __builtin_memcpy(&f, &bits, 4);
    return f;
}
/// S0 such that n terms of magnitude <= 1 (scaled by 2^S0) cannot leave 62 bits
__host__ __device__ inline int fixed_shift_for(size_t n) {
    int lg = 1;
    while (lg < 40 && ((size_t) 1 << lg) <= n) ++lg;                 // 2^lg > n
    return 62 - lg;
}
/// scale of the plain sums (terms bounded by 1) and of the x-weighted sums (terms bounded by max |x|, given as its bits: finite)
__host__ __device__ inline FixedScale fixed_scale(int S0, uint32_t xmax_bits, bool weighted) {
    int S = S0;
    if (weighted) {
        int E = (int) (xmax_bits >> 23);                            // max |x| < 2^(E - 126)
        if (E < S0 + 1) E = S0 + 1;                                 // (tiny or zero: scaled as if max |x| were 2^(S0 - 126); every factor stays a normal number)
        S = S0 - (E - 126);
    }
    const unsigned long long bits = (unsigned long long) (S + 1023) << 52;
    double up;
    __builtin_memcpy(&up, &bits, 8);
    return FixedScale{ up, pow2f(-S) };
}

/// ENOKI_HIP_EARLY_SUMS=locks: never (exchange locks for every pair, half-size buckets as in round 5)
inline bool early_fixed_enabled() {
    static const bool on = [] { const char *e = getenv("ENOKI_HIP_EARLY_SUMS"); return !(e && !strcmp(e, "locks")); }();
    return on;
}

// ---- host side -------------------------------------------------------------------------------------
struct Bucketed {
    int type = 0, index_type = 0, op = 0;
    size_t n = 0, table_size = 0;
    const void *table_a = nullptr, *table_c = nullptr;    // not owned: the caller keeps the tables alive and unchanged
    int n_buckets = 0;
    unsigned max_pieces = 0;
    void *meta = nullptr;          // counts[n_buckets][blocks] | row_total | bucket_base | piece_prefix | reduce partials
    uint32_t *bucket_base = nullptr, *piece_prefix = nullptr;
    void *reduce_partials = nullptr;
    void *pair_idx = nullptr;      // uint16_t[n]: index within the bucket, bucket order
    void *x_b = nullptr;           // x in bucket order
    void *u_b = nullptr;           // u in bucket order (allocated by the first consumer that keeps it)
    bool has_u = false;
    void *m_b = nullptr;           // m_op(u) in bucket order: the kept half of a sincos pair (see ek_hip_bucketed_reduce)
    int m_op = EK_COPY;
    bool has_m = false;
    int shift = 0;                 // buckets of 2^shift table entries: bin_shift_of<T>, or one less (EK_BUCKETED_HINT_ADJOINT)
    void *early = nullptr;         // k_bucket_pair_forward_adjoint: per piece, sums of early_op(u) and of x * early_op(u) per entry
    int early_op = EK_COPY;
    bool has_early = false;
    bool early_fixed = false;      // `early` holds 64-bit fixed-point sums (8 B per entry and piece) + one mode word per piece behind them
    int early_S0 = 0;              // their scale: terms bounded by 1 times 2^S0 (fixed_scale())
    uint32_t *early_modes = nullptr;   // device, [max_pieces]: 0 fixed point, 1 floats (a piece that ran under locks)
    // paged lists (ek_paged.h; 4-byte element types): page_shift = 5 | 6, pair_idx / x_b / u_b / m_b hold `positions` elements
    // (whole pages, the workgroups' unused slots in between), bucket_base counts complete pages
    int page_shift = 0;
    size_t positions = 0;
    void *page_lists = nullptr;    // glist_full[page slots] | glist_part[W * n_buckets]
    uint32_t *glist_full = nullptr, *glist_part = nullptr, *base_part = nullptr;
    const uint32_t *active = nullptr;      // device: [0] number of elements in the lists, [1] non-finite x under a cleared mask bit (null: contiguous lists)
    uint32_t *ticket = nullptr;            // device, zero between launches: the last workgroup of a reducing launch finishes the reduction (bucket_finish)
    int meta_slot = -1;                    // >= 0: `meta` is a block of the context's ring (meta_ring()), not an allocation of its own
    bool meta_clean = false;               // a reducing launch RAN (not merely recorded): its last workgroup left the counters of the block zeroed
    bool meta_clean_pending = false;       // finish() handed the counters to a launch that has not been checked yet (launched_reducing())
    uint32_t win_lo = 0, win_span = 0;     // a slice of a large table: only indices in [win_lo, win_lo + win_span) (ek_hip_bucketed::slices)
    bool correct_masked = true;            // the final reduction adds the masked-out lanes' map_op(0) terms (slices: their owner does)

    bool has_mask = false;
    // (with or without a mask array: lanes whose index points outside the table are dropped by the partition too, and count like
    //  masked-out ones -- one rule for a single object and for the slices of a large table)
    const uint32_t *masked_ptr() const { return correct_masked ? active : nullptr; }
    template <typename T> BucketFinish<T> finish(void *out, int zero_op, bool reducing = true) {
        // (a ring block under a launch that reduces: its last workgroup clears the partition's counters -- from then on the block is clean)
        BucketFinish<T> f{ ticket, (T *) out, masked_ptr(), n, zero_op };
        // (the block only counts as clean once that launch was really ENQUEUED FOR EXECUTION -- a launch recorded into a
        //  step graph may never run, a failed launch did not run: launched_reducing() below, called behind EK_LAUNCH_CHECK)
        if (reducing && meta_slot >= 0 && ticket) { f.counters = (uint32_t *) meta; meta_clean_pending = true; }
        return f;
    }
    void launched_reducing() {
        if (meta_clean_pending && refuse_while_capturing_quiet() == EK_OK) meta_clean = true;
        meta_clean_pending = false;
    }
    // signs applied ONCE to the staged table entries (exact): fmsub / mulsub: -c;  fnmadd / nmuladd (c - a x = (-a) x + c): -a;  fnmsub: both
    int flip_a() const { return op == EK_FNMADD || op == EK_FNMSUB || op == EK_NMULADD; }
    int flip_c() const { return op == EK_FMSUB || op == EK_FNMSUB || op == EK_MULSUB; }
    int two_roundings() const { return op == EK_MULADD || op == EK_MULSUB || op == EK_NMULADD; }
    size_t bins() const { return (size_t) 1 << shift; }
    BucketLists lists() const { return BucketLists{ bucket_base, piece_prefix, base_part, glist_full, glist_part, n_buckets }; }
    ~Bucketed();
};

template <typename K> static inline int allow_big_lds(K kernel, size_t bytes) {
    // beyond 64 KiB of dynamic LDS a kernel has to opt in -- once per kernel and size (a runtime call per launch otherwise)
    static std::unordered_map<const void *, size_t> allowed;
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    size_t &have = allowed[reinterpret_cast<const void *>(kernel)];
    if (bytes <= have) return EK_OK;
    EK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
    have = bytes;
    return EK_OK;
}

/// static LDS of a kernel (hipFuncGetAttributes, asked once per kernel); SIZE_MAX when the runtime cannot say
template <typename K> static inline size_t static_lds_of(K kernel) {
    static std::unordered_map<const void *, size_t> known;
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    auto it = known.find(reinterpret_cast<const void *>(kernel));
    if (it != known.end()) return it->second;
    hipFuncAttributes attr;
    const size_t v = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kernel)) == hipSuccess ? (size_t) attr.sharedSizeBytes : SIZE_MAX;
    known[reinterpret_cast<const void *>(kernel)] = v;
    return v;
}

// kernel variants by list layout: contiguous for 8-byte element types, pages of 32 / 64 elements for 4-byte ones
#define EK_BY_LAYOUT(b, call)                                                                                  \
    do {                                                                                                       \
        if constexpr (sizeof(T) == 8) { constexpr int PS = 0; call; }                                          \
        else if ((b)->page_shift == 6) { constexpr int PS = 6; call; }                                         \
        else { constexpr int PS = 5; call; }                                                                   \
    } while (0)

/// the forward + adjoint launch of an object made with half-size buckets (bucketed_early.hip; instantiated for float and double)
template <typename T> int bucketed_forward_adjoint_launch(Bucketed *b, void *out, int map_op, int keep_op);

} // namespace ek
