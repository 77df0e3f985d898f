// The reduction of f(u) AND the adjoint of the two gathers in ONE pass over a bucket's elements (the headline step's second big
// kernel).  See bucketed.hip for the path as a whole; what the translation units share is in ek_bucketed.h.
#include "ek_bucketed.h"

#include <cstdlib>
#include <cstring>

namespace ek {

// ---- 2 + 3 in one pass: the reduction of f(u) AND the adjoint of the gathers ------------------------------------------
// y = hsum(f(u)) is linear in its seed: whatever gradient g the tape later sends down, the tables receive g * sum(f'(u)) and
// g * sum(x f'(u)) per entry.  When the reduction is asked to KEEP the function of u that the derivative will be made of --
// the other half of a sincos pair for sin / cos, the value itself for exp, rcp(u) for log, ... -- the sums are formed right
// here, while u, the two function values and the bucket's table slice are at hand: the LDS holds the {A, C} slice AND the two
// gradient tables of a half-size bucket (2 x 64 KiB), the kept function is never written (4 B/elt) or read back (4 B/elt),
// (l16, x_b) is streamed once instead of twice, and the adjoint's LDS round trips overlap the forward's arithmetic.  The
// tape's scatter_add of exactly these streams then only folds the partial tables (ek_hip_bucketed_scatter_add, with the seed
// as a factor); anything else it asks for takes the ordinary kernels.
template <int Map, int Keep, typename T> struct EarlyPair {
    // reduced value and kept function of one u
    static __device__ __forceinline__ void apply(T u, T &val, T &kept) {
        if constexpr ((Map == EK_SIN && Keep == EK_COS) || (Map == EK_COS && Keep == EK_SIN)) {
            T sn, cs;
            SinCosOp::apply(u, sn, cs);
            val = Map == EK_SIN ? sn : cs;
            kept = Map == EK_SIN ? cs : sn;
        } else if constexpr (Map == Keep) {
            val = kept = UnaryOp<Map, T>::apply(u);
        } else {
            val = UnaryOp<Map, T>::apply(u);
            kept = UnaryOp<Keep, T>::apply(u);
        }
    }
};

#ifdef EK_EARLY_TIMING
__device__ unsigned long long g_early_timing[32];       // [16..31]: the fixed-point path per workgroup (thread 0)
#endif

template <typename T, int V, int Map, int Keep, bool Two>
struct EarlyBody {
    static constexpr bool Paired = sizeof(T) == 4;
    struct Step { Pack<uint16_t, 4> pi[V]; T px[V][4]; };
    const PairRec<T> *rec;
    T *tables;
    unsigned long long *dummy;
    const uint16_t *pair_idx;
    const T *x_b;
    int Bins;
    uint32_t lmask;
    T acc[4];
#ifdef EK_EARLY_TIMING
    // measurement builds only (tools/probe_early_phases.py): cycles of a wave per phase of a step
    //   0 wait for the step's (l16, x) loads   1 record reads   2 arithmetic + claim / add / release   3 retry round   4 steps
    unsigned long long tacc[5] = { 0, 0, 0, 0, 0 }, tlast = 0, tmark[2] = { 0, 0 };
    __device__ __forceinline__ void mark(int k) { tmark[k] = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void stamp(int k) { const unsigned long long now = __builtin_readcyclecounter(); tacc[k] += now - tlast; tlast = now; }
#endif

    // reduced value into `sum`, kept function m and x * m out
    __device__ __forceinline__ void values(uint32_t l, T x, T &sum, T &v0, T &v1) const {
        const PairRec<T> r = rec[l];
        EarlyPair<Map, Keep, T>::apply(pair_value(r.a, x, r.c, Two), sum, v0);
        v1 = dev::safe_mul(x, v0);
    }
    __device__ __forceinline__ void one(size_t pos, bool on, int slot) {
        const uint32_t l = on ? (uint32_t) pair_idx[pos] & lmask : 0u;
        T sum, v0, v1;
        values(l, on ? x_b[pos] : T(0), sum, v0, v1);
        if (on) acc[slot] += sum;
        if constexpr (Paired) {
            lds_add_pair(reinterpret_cast<unsigned long long *>(tables) + l, v0, v1, on);
        } else {
            lds_add<true>(&tables[l], v0, on);
            lds_add<true>(&tables[Bins + l], v1, on);
        }
    }
    __device__ __forceinline__ void fetch(Step &s, int h, size_t pos) {
        s.pi[h] = pack_load<uint16_t, 4, true>(pair_idx + pos);
        load4<T, true>(x_b + pos, s.px[h]);
    }
    __device__ __forceinline__ void apply(const Step &s) {
        constexpr int NB = 4 * V;
        uint32_t l[NB];
        T v0[NB], v1[NB];
        if constexpr (Paired) {
            // f32: the step's table records first (four independent reads), then element by element: value and kept function,
            // ONE claim -- add -- release without a branch (lds_try_add_pair), and what met a lock retried once per step, all
            // slots in flight (lds_add_pair_retry).  A lock is held for one LDS round trip.  Measured on 64 Mi lookups into
            // 1 Mi entries, same box, us per launch: 8 claims per lane in flight 199, 4: 165, 2: 149, one claim behind each
            // element with its own retry round 143-147, this form 137-143; the next element's arithmetic pinned under the
            // exchange's round trip (a longer hold) 147-149; sincos / exp in packed-fp32 instructions (v_pk_fma_f32: two
            // passes on gfx950's SIMD-32) 150-153 against 147-150.  With the claims removed the kernel takes 100, with the
            // arithmetic removed 124, with both 80-87 (profiles/probe_early_r04.txt).  Round 5: a LOCK-FREE form -- the pairs of
            // all four elements read up front, four compare-and-swaps in flight, two dependent LDS round trips per step instead of
            // five -- is 11 % SLOWER (171 against 154 us): it moves 40 B per element through the LDS instead of 32, and that, not
            // the latency of the claim -- add -- release chain, is what bounds the adjoint part (profiles/probe_early_r05.txt).
            unsigned long long *tb = reinterpret_cast<unsigned long long *>(tables);
            PairRec<T> r[NB];
#ifdef EK_EARLY_TIMING
            tlast = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");        // (the next step's two loads and the list entry behind them stay in flight)
            stamp(0);
#endif
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                l[k] = (uint32_t) s.pi[k / 4].v[k % 4] & lmask;
                r[k] = rec[l[k]];
            }
#ifdef EK_EARLY_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(1);
#endif
            unsigned pending = 0;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const T x = s.px[k / 4][k % 4];
                T sum;
                EarlyPair<Map, Keep, T>::apply(pair_value(r[k].a, x, r[k].c, Two), sum, v0[k]);
                v1[k] = dev::safe_mul(x, v0[k]);
                acc[k % 4] += sum;
                pending |= lds_try_add_pair(tb, dummy, l[k], v0[k], v1[k]) ? 1u << k : 0u;
            }
#ifdef EK_EARLY_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(2);
#endif
            if (__builtin_amdgcn_ballot_w64(pending != 0)) lds_add_pair_retry<NB>(tb, l, v0, v1, pending);
#ifdef EK_EARLY_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(3);
            tacc[4] += 1;
#endif
            return;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            l[k] = (uint32_t) s.pi[k / 4].v[k % 4] & lmask;
            T sum;
            values(l[k], s.px[k / 4][k % 4], sum, v0[k], v1[k]);
            acc[k % 4] += sum;
        }
        if constexpr (!Paired) {
            lds_add_batch<T, NB>(tables, l, v0);
            lds_add_batch<T, NB>(tables + Bins, l, v1);
        }
    }
};



// ---- round 6: the same pass with the piece's pages handed out DYNAMICALLY ----------------------------------------------------
// What the phase stamps of the static walk showed (tools/probe_early_phases.py, profiles/probe_early_phases_r06.txt): the loads of a
// step are there when the step starts (8 cycles of waiting) -- the kernel is not short of memory parallelism; a step is ~450 cycles
// of record reads, ~1300 of arithmetic + claims and ~1100 of RETRY ROUND (one more dependent LDS round trip for the handful of lanes
// that met a lock -- which some lane of a wave does in two steps out of three: 64 lanes x 4 slots over 8 Ki entries); behind the steps,
// the complete pages that do not fill a step of the whole workgroup and the partially filled pages go element by element, every
// element a chain of list load -> data load -> record read -> claim (18 % of the walk at 64 Mi elements, 75 % in an 8 Mi shard);
// and the waves of a workgroup, which all take the same number of steps, finish up to 48 % apart (the slowest sets the barrier).
// Here instead:
//   * a wave takes BATCHES of 64 / LX pages from a counter in the LDS (one returning ds_add per step, requested three steps ahead):
//     fast waves take more, everybody finishes within a step of everybody else;
//   * complete and partially filled pages are ONE sequence; every batch goes through the same two vector loads, the elements beyond
//     a page's count are switched off by a lane predicate (they claim the lane's private slot: still no branch in the step);
// (A lock-protocol body on this walk -- what met a lock carried into the next step, its exchange in flight with that step's record
// reads -- measured 155-157 us against 151-155 for the static walk: git 4c0... has it; the walk pays off with the fixed-point body below.)
// ---- round 6: the adjoint sums as 64-bit FIXED POINT, added by non-returning LDS atomics -------------------------------------
// ds_add_u64 costs what ds_write_b64 costs (13 LDS cycles per wave instruction under random entries, profiles/probe_lds64_r06.txt;
// ds_add_f32 193, ds_add_f64 26): a record read + two adds move through the LDS in the time of today's read + exchange + release,
// without a lock -- no retry round (two of every five LDS instructions of the lock protocol), no claim -> add -> release chain
// (four exposed LDS round trips per step), no lock bookkeeping in the vector ALU, and the sums do not depend on the order of the
// additions: the default mode becomes bit-reproducible.  A term v becomes round-down(v * 2^S), S from a bound of |v| over the
// launch and the number of elements (no sum can leave 63 bits); terms of 24 significant bits are exact from 2^-13 of the bound
// upwards, below that the error is < 2^-S per term -- 2^-37 of the bound at 64 Mi elements.
__device__ __forceinline__ unsigned long long to_fixed64(float v, double up) {
    // t = v 2^S is exact in a double (24 significant bits, any S); d = t + 1.5 * 2^52 rounds it to the nearest integer k (ties to
    // even; |k| < 2^51 by the choice of S: S0 <= 48) and carries k in its low 52 bits: bits(d) = bits(1.5 * 2^52) + k as 64-bit
    // integers -- the low word of the constant is zero, so the subtraction is ONE 32-bit add on the high word.  v_cvt_f64_f32 +
    // v_fma_f64 + v_add_u32 (~9 issue cycles of a SIMD, profiles/probe_valu_r06.txt) against the float form's two multiplies,
    // v_rndne, fma, two conversions, a shift and an add (~17): the split of t into two exactly representable floats
    // (hf = rint(t / 2^32), lf = t - hf 2^32) is in git 89107d7.
    const double d = __builtin_fma((double) v, up, 6755399441055744.0);
    return __builtin_bit_cast(unsigned long long, d) - 0x4338000000000000ull;
}

/// bytes of each of the three areas of the fixed-point kernel's LDS: the {a, c} records of a 4 Ki-entry bucket, the plane of the
/// plain sums, the plane of the x-weighted sums.  Constants whatever Bins is, so that the byte offset 8 l of an entry -- ONE
/// shift-and-mask of the packed 16-bit indices -- addresses all three through the instructions' immediate offsets (the third
/// plane lies one byte beyond the 16-bit offset field: one add).  PLANES, not {s0, s1} pairs: with 16-byte entries one
/// ds_add_u64 instruction touches only every other pair of banks -- measured, SQ_LDS_BANK_CONFLICT 32.7 M against 22.9 M cycles
/// per launch and the kernel LDS-bound at 77 % (profiles/rocprof_sq_early_pairs_r06.txt).
constexpr uint32_t kFixedRecBytes = 4096u * 8u;

template <int PS, int Map, int Keep, bool Two>
struct EarlyFixed {
    using T = float;
    static constexpr uint32_t Page = 1u << PS, LX = Page / 4, PW = 64 / LX;
    struct Step { Pack<uint16_t, 4> pi; T px[4]; };
    const unsigned char *lds;                // records at lds + 8 l, sums at lds + kFixedRecBytes + 8 l and lds + 2 kFixedRecBytes + 8 l
    const uint16_t *pair_idx;
    const T *x_b;
    uint32_t m8;                             // (Bins - 1) << 3, in a VECTOR register (an SGPR operand halves the issue rate of a VOP2, probe_valu)
    double up0, up1;
    // the reduced value (|sin|, |cos| <= 1) as an INTEGER too: a term becomes trunc(f 2^28) -- four of them fit 32 bits, so a batch
    // costs one 64-bit addition -- and the sum over the launch is exact in 64 bits: y does not depend on which lane, wave or piece
    // took which element (the float accumulators of the lock path do).  2^-28 per term is 1/16 of the float's own last bit at 1.
    long long iacc;

    __device__ __forceinline__ void fetch(Step &s, size_t pos) {
#ifdef EK_FX_DIAG_IDXPAIR
        // measurement only (wrong data): the two lane groups of a pair read the two halves of ONE 128-byte line of indices
        const uint32_t lane = threadIdx.x & 63u, even = (lane / (2 * LX)) * 2 * LX + lane % LX;
        const uint32_t page = (uint32_t) (pos >> PS), pa = (uint32_t) __shfl((int) page, (int) even, 64);
        const size_t posl = ((size_t) (pa & ~1u) << PS) + (((lane / LX) & 1u) << PS) + (pos & (Page - 1u));
        s.pi = pack_load<uint16_t, 4, true>(pair_idx + posl);
#else
        s.pi = pack_load<uint16_t, 4, true>(pair_idx + pos);
#endif
        load4<T, true>(x_b + pos, s.px);
    }
    struct Recs { uint32_t a8[4]; PairRec<T> r[4]; };
    /// the records of a step's four entries: issued ONE STEP AHEAD of the arithmetic that uses them (walk_pages_dynamic) -- behind the
    /// 8 x 16 non-returning adds the workgroup's other waves keep queued in the LDS, a read takes longer than a step's arithmetic
    __device__ __forceinline__ void lookup(const Step &s, Recs &o) const {
        uint32_t w[2];
        __builtin_memcpy(w, &s.pi, 8);
        o.a8[0] = (w[0] << 3) & m8; o.a8[1] = (w[0] >> 13) & m8;
        o.a8[2] = (w[1] << 3) & m8; o.a8[3] = (w[1] >> 13) & m8;
#pragma unroll
        for (int k = 0; k < 4; ++k) o.r[k] = *reinterpret_cast<const PairRec<T> *>(lds + o.a8[k]);
#ifdef EK_FX_DIAG_NOLOOKUP
#pragma unroll
        for (int k = 0; k < 4; ++k) o.r[k] = PairRec<T>{ __uint_as_float(o.a8[k] | 0x3f000000u), T(0.25) };
#endif
    }
    template <bool Masked>
    __device__ __forceinline__ void apply(const Step &s, const Recs &rc, int rem) {
        const uint32_t *a8 = rc.a8;
        const PairRec<T> *r = rc.r;
        int batch = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const T x = s.px[k];
            // (every u of a piece that got here is finite: the piece guard of the kernel -- no fix-up branch per element)
            const T u = pair_value(r[k].a, x, r[k].c, Two);
            T sn = T(0), cs = T(0);
#ifdef EK_FX_DIAG_NOSINCOS
            sn = u; cs = u * T(0.5);
#else
            dev::sincos_f32<Map == EK_SIN || Keep == EK_SIN, Map == EK_COS || Keep == EK_COS, true>(u, sn, cs);
#endif
            const T sum = Map == EK_SIN ? sn : cs;
            T v0 = Keep == EK_SIN ? sn : cs;
            int q = dev::cvt_sat_i32(sum * 268435456.0f);
            if constexpr (Masked) {
                // beyond a page's count: the lane adds ZERO to whatever entry its stale index names (in range by the mask) -- no
                // branch, no private slot; x * 0 = 0 even for a stale infinity (safe_mul is v_mul_legacy_f32)
                const bool on = k < rem;
                q = on ? q : 0;
                v0 = on ? v0 : T(0);
            }
            batch += q;
            const T v1 = dev::safe_mul(x, v0);
            unsigned long long *p = reinterpret_cast<unsigned long long *>(const_cast<unsigned char *>(lds) + kFixedRecBytes + a8[k]);
            unsigned long long *p1 = reinterpret_cast<unsigned long long *>(const_cast<unsigned char *>(lds) + 2 * kFixedRecBytes + a8[k]);
#ifdef EK_FX_DIAG_NOADDS
            batch += (int) (to_fixed64(v0, up0) >> 20) + (int) (to_fixed64(v1, up1) >> 20) + (int) (size_t) p;
#else
            (void) __hip_atomic_fetch_add(p, to_fixed64(v0, up0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void) __hip_atomic_fetch_add(p1, to_fixed64(v1, up1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        }
        iacc += (long long) batch;
    }
    __device__ __forceinline__ void flush() { }
};

/// batches of a piece's pages handed to the waves from `s_next` (initialised to 5 * kBucketWaves: a wave's first five batches are
/// wave, wave + 16, ..., wave + 64)
template <int PS, typename Body>
__device__ __forceinline__ void walk_pages_dynamic(const BucketLists &bl, const PieceRange &r, Body &body, uint32_t *s_next) {
    using Step = typename Body::Step;
    constexpr uint32_t LX = Body::LX, PW = Body::PW, Page = Body::Page;
    const uint32_t lane = threadIdx.x & 63u, gw = lane / LX, i = lane % LX, wave = threadIdx.x >> 6;
    const uint32_t nfull = r.f1 - r.f0, nparts = r.p1 - r.p0, total = nfull + nparts, nb = (total + PW - 1u) / PW;
    auto grab = [&]() -> uint32_t {
        uint32_t v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(s_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return v;                                                                     // (read through readfirstlane when it is needed)
    };
    // the list entry of this lane's page in batch b (one load whichever list it is in; beyond the piece: any valid word)
    auto entry = [&](uint32_t b) -> uint32_t {
        const uint32_t q = b * PW + gw;
        const uint32_t *a = q < nfull ? bl.glist_full + r.f0 + q : q < total ? bl.glist_part + r.p0 + (q - nfull) : bl.base;
        return __builtin_nontemporal_load(a);
    };
    auto place = [&](uint32_t b, uint32_t ev, size_t &pos, int &rem) {
        const uint32_t q = b * PW + gw;
        const uint32_t page = q < nfull ? ev : q < total ? ev >> 6 : 0u;
        const uint32_t count = q < nfull ? Page : q < total ? (ev & 63u) + 1u : 0u;
        pos = ((size_t) page << PS) + 4u * i;
        rem = (int) count - (int) (4u * i);
    };
    // a batch moves through five stages: counter -> list entry -> (l16, x) loads (TWO steps in flight: a step is shorter than the
    // memory's latency under load) -> record reads -> arithmetic + adds.  The four step buffers, the two record sets and the two
    // list entries change ROLES from phase to phase instead of being copied: a register move of a value still in flight would
    // wait for it (s_waitcnt vmcnt(0) at the top of every step is what the rolled form compiles to).
    uint32_t id0 = wave, id1 = wave + kBucketWaves, id2 = wave + 2 * kBucketWaves, id3 = wave + 3 * kBucketWaves, id4 = wave + 4 * kBucketWaves;
    if (id0 >= nb) return;
    uint32_t evA = entry(id0), evB = entry(id1);
    const uint32_t ev2 = entry(id2);
    Step st0, st1, st2, st3;
    typename Body::Recs r0, r1;
    size_t pos;
    int rem0, rem1, rem2, rem3 = 0;
    place(id0, evA, pos, rem0);
    body.fetch(st0, pos);
    place(id1, evB, pos, rem1);
    body.fetch(st1, pos);
    evB = entry(id3);                                        // (in the order of a phase: the list entry before the data in front of it)
    place(id2, ev2, pos, rem2);
    body.fetch(st2, pos);
    body.lookup(st0, r0);
    auto phase = [&](const Step &cur, const Step &next, Step &last, const typename Body::Recs &rc, typename Body::Recs &rn, int remc, int &reml,
                     uint32_t &ev_new, uint32_t ev_old) -> bool {
        const uint32_t g5 = grab();                          // five batches ahead: the counter
        ev_new = entry(id4);                                 // four ahead: the list entry
        place(id3, ev_old, pos, reml);                       // three ahead: the data
        body.fetch(last, pos);
        body.lookup(next, rn);                               // one ahead: the records
        if (__builtin_amdgcn_ballot_w64(remc < 4)) body.template apply<true>(cur, rc, remc);
        else body.template apply<false>(cur, rc, 4);
        id0 = id1; id1 = id2; id2 = id3; id3 = id4; id4 = (uint32_t) __builtin_amdgcn_readfirstlane((int) g5);
        return id0 < nb;
    };
    while (true) {
        if (!phase(st0, st1, st3, r0, r1, rem0, rem3, evA, evB)) break;
        if (!phase(st1, st2, st0, r1, r0, rem1, rem0, evB, evA)) break;
        if (!phase(st2, st3, st1, r0, r1, rem2, rem1, evA, evB)) break;
        if (!phase(st3, st0, st2, r1, r0, rem3, rem2, evB, evA)) break;
    }
    body.flush();
}


/// bucket_finish for the fixed-point kernel: every workgroup publishes an INTEGER partial (units of 2^-28; pieces that ran under locks: 0)
/// and a float one (pieces under locks; 0 otherwise); the last workgroup adds the integers exactly -- the order cannot matter --,
/// converts once, and adds the float partials in the order of the pieces.
/// first half: thread 0 publishes the workgroup's partials and draws the ticket -- BEFORE the workgroup writes its tables, so that the
/// ticket's round trip (and its wait for the wave's stores, none outstanding yet) runs under the conversion of the tables
__device__ __forceinline__ void bucket_finish_fixed_begin(long long iblock /* thread 0 */, float fblock /* thread 0 */, float *__restrict__ partials,
                                                          uint32_t *__restrict__ ticket, uint32_t *s_last_fixed /* shared */) {
    unsigned long long *ipartials = reinterpret_cast<unsigned long long *>(partials);             // [gridDim.x] integers, then [gridDim.x] floats
    uint32_t *fpartials = reinterpret_cast<uint32_t *>(ipartials + gridDim.x);
    if (threadIdx.x == 0) {
        __hip_atomic_store(ipartials + blockIdx.x, (unsigned long long) iblock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(fpartials + blockIdx.x, __float_as_uint(fblock), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef EK_EARLY_TIMING
        const unsigned long long t_tk = __builtin_readcyclecounter();
#endif
        *s_last_fixed = finish_ticket(ticket) ? 1u : 0u;  // (relaxed, behind a wait for the stores above: ek_bucketed.h)
#ifdef EK_EARLY_TIMING
        const unsigned long long dt_tk = __builtin_readcyclecounter() - t_tk;
        atomicAdd(&g_early_timing[24], dt_tk); atomicMax(&g_early_timing[25], dt_tk);
#endif
    }
}
__device__ __forceinline__ void bucket_finish_fixed_end(const uint32_t *s_last_fixed /* shared */, float *__restrict__ partials,
                                                        uint32_t *__restrict__ ticket, float *__restrict__ out, const uint32_t *__restrict__ active,
                                                        size_t n, int map_op, float *wave_part, long long *wave_ipart, uint32_t *__restrict__ counters) {
    unsigned long long *ipartials = reinterpret_cast<unsigned long long *>(partials);
    uint32_t *fpartials = reinterpret_cast<uint32_t *>(ipartials + gridDim.x);
    __syncthreads();
    if (!*s_last_fixed) return;
    if (counters) {
        for (unsigned k = threadIdx.x; k < kPgTotalsWords + 4u; k += blockDim.x) counters[k] = 0u;
        for (unsigned k = kPgMetaBase + kPgMetaAccumRep + threadIdx.x; k < kPgMetaBase + kPgMetaAccumRepEnd; k += blockDim.x) counters[k] = 0u;
    }
    long long iv = 0;
    float fv = 0.f;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x) {
        iv += (long long) __hip_atomic_load(ipartials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fv += __uint_as_float(__hip_atomic_load(fpartials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { iv += bucket_shfl_down(iv, d); fv += bucket_shfl_down(fv, d); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { wave_ipart[threadIdx.x >> 6] = iv; wave_part[threadIdx.x >> 6] = fv; }
    __syncthreads();
    if (threadIdx.x < 64) {
        iv = threadIdx.x < blockDim.x / 64 ? wave_ipart[threadIdx.x] : 0ll;
        fv = threadIdx.x < blockDim.x / 64 ? wave_part[threadIdx.x] : 0.f;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) { iv += bucket_shfl_down(iv, d); fv += bucket_shfl_down(fv, d); }
        if (threadIdx.x == 0) {
            const float r = (float) iv * 3.7252902984619140625e-9f /* 2^-28 */ + fv;
            out[0] = bucket_dropped_lanes<float, EK_HSUM>(r, active ? n - (size_t) active[0] : 0, active && active[1], map_op);
            finish_ticket_reset(ticket);
        }
    }
}

template <typename T, int V, int PS, bool Fixed, bool Two>
__global__ __launch_bounds__(kBucketThreads) void k_bucket_pair_forward_adjoint(T *__restrict__ partials, T *__restrict__ table_partials,
                                                                                const T *__restrict__ table_a,
                                                                                const T *__restrict__ table_c, size_t table_size,
                                                                                int flip_a, int flip_c,
                                                                                const uint16_t *__restrict__ pair_idx,
                                                                                const T *__restrict__ x_b, BucketLists bl,
                                                                                int map_op, int keep_op, int shift, BucketFinish<T> fin,
                                                                                const uint32_t *__restrict__ xmax_bits, int S0,
                                                                                uint32_t *__restrict__ piece_mode) {
    extern __shared__ __align__(16) unsigned char lds_dynamic[];
    // Fixed: a STATIC block (records of 4 Ki entries + 4 Ki pairs of sums, whatever Bins is) -- its address is a constant of the
    // program, every LDS instruction of the walk carries it in its immediate offset
    __shared__ __align__(16) unsigned char lds_static[Fixed ? 3 * kFixedRecBytes : 16];
    unsigned char *lds_raw = Fixed ? lds_static : lds_dynamic;
    const int Bins = 1 << shift;
    PairRec<T> *rec = reinterpret_cast<PairRec<T> *>(lds_raw);
    // f32: Bins {t0, t1} pairs under one lock;  f64: two tables;  Fixed: two planes of 64-bit sums (areas of constant size) behind the records
    T *tables = Fixed ? reinterpret_cast<T *>(lds_raw + kFixedRecBytes) : reinterpret_cast<T *>(rec + Bins);
    __shared__ T wave_part[kBucketWaves];
    __shared__ unsigned long long s_dummy;
    __shared__ uint32_t s_next;
    __shared__ uint32_t s_guard[2];
    __shared__ long long wave_ipart[Fixed ? kBucketWaves : 1];
    __shared__ uint32_t s_last_fixed;
    constexpr bool Paired = sizeof(T) == 4;
    int bucket;
    PieceRange range;
#ifdef EK_EARLY_TIMING
    const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
    // (asked for before anything depends on it: a dependent round trip behind the staging barrier otherwise)
    [[maybe_unused]] uint32_t xm_early = 0;
    if constexpr (Fixed) {
        xm_early = xmax_bits[0];
        if (threadIdx.x == 0) { s_next = 5u * kBucketWaves; s_guard[0] = 0u; s_guard[1] = 0u; }       // (ordered before their users by bucket_piece's barrier)
    }
    if (!bucket_piece<PS>(bl, bucket, range)) {
        if constexpr (Fixed) {
            bucket_finish_fixed_begin(0ll, T(0), partials, fin.ticket, &s_last_fixed);
            bucket_finish_fixed_end(&s_last_fixed, partials, fin.ticket, fin.out, fin.active, fin.n, fin.zero_op, wave_part, wave_ipart, fin.counters);
        }
        else bucket_finish<T, EK_HSUM>(T(0), partials, fin.ticket, fin.out, fin.active, fin.n, fin.zero_op, wave_part, fin.counters);
        return;
    }
#ifdef EK_EARLY_TIMING
    const unsigned long long t_piece = __builtin_readcyclecounter();
#endif
#ifdef EK_EARLY_TIMING
    unsigned long long fx_t[6] = { 0, 0, 0, 0, 0, 0 };
    fx_t[0] = __builtin_readcyclecounter();           // piece known
#endif
    [[maybe_unused]] uint32_t slice_max[2] = { 0u, 0u };
    stage_pair_slice<T, true>(rec, tables, table_a, table_c, (size_t) bucket * Bins, table_size, Bins, flip_a, flip_c, Fixed ? slice_max : nullptr);
    if constexpr (Fixed) {
        for (int j = threadIdx.x; j < 2 * 4096; j += kBucketThreads) reinterpret_cast<unsigned long long *>(tables)[j] = 0ull;     // both planes, whole areas
    }
    if constexpr (Fixed) {
        // max |a|, max |c| over the slice, taken from the registers the slice was staged from (one pass, one barrier)
        uint32_t am = slice_max[0], cm = slice_max[1];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { am = max(am, (uint32_t) __shfl_xor((int) am, d, 64)); cm = max(cm, (uint32_t) __shfl_xor((int) cm, d, 64)); }
        if ((threadIdx.x & 63) == 0) { atomicMax(&s_guard[0], am); atomicMax(&s_guard[1], cm); }
    }
    __syncthreads();
    T v = T(0);
#ifdef EK_EARLY_TIMING
    unsigned long long t_walk_end = 0;
#endif
    auto run = [&](auto body) {
        body.rec = rec; body.tables = tables; body.pair_idx = pair_idx; body.x_b = x_b; body.Bins = Bins;
        body.lmask = (uint32_t) Bins - 1u; body.dummy = &s_dummy;
#pragma unroll
        for (int k = 0; k < 4; ++k) body.acc[k] = T(0);
#ifdef EK_EARLY_TIMING
        const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
        walk_piece<PS, V>(bl, range, body);
#ifdef EK_EARLY_TIMING
        if ((threadIdx.x & 255) == 0) {              // (one wave per SIMD-quad reports: 4 of the 16 waves)
            const unsigned long long t_end = __builtin_readcyclecounter();
            for (int k = 0; k < 5; ++k) atomicAdd(&g_early_timing[k], body.tacc[k]);
            atomicAdd(&g_early_timing[5], t_end - t_begin);
            atomicAdd(&g_early_timing[6], 1ull);
            atomicAdd(&g_early_timing[7], body.tmark[0] - t_begin);           // the main steps
            atomicAdd(&g_early_timing[8], body.tmark[1] - body.tmark[0]);     // the complete pages that do not fill a step
            atomicAdd(&g_early_timing[9], t_end - body.tmark[1]);             // the partially filled pages
            atomicAdd(&g_early_timing[10], t_piece - t_entry);                // which piece am I
            atomicAdd(&g_early_timing[11], t_begin - t_piece);                // table slice staged, tables cleared, barrier
            atomicMax(&g_early_timing[12], t_end - t_begin);                  // the slowest wave's walk
            t_walk_end = t_end;
        }
#endif
        v = (body.acc[0] + body.acc[1]) + (body.acc[2] + body.acc[3]);
    };
    // Fixed: the sums in fixed point -- unless no scale exists (max |x| over the launch infinite, NaN, or so small, below
    // 2^(S0 - 125) and not zero, that 2^S would leave the range of normal numbers), or THIS piece could meet a term that fixed point
    // cannot carry: a NaN from a non-finite table entry, or from a u beyond the range in which the reference's sincos stays finite
    // (|u| >= 1.8e19: its reduced argument squared overflows, array_math.h:331-340) -- decided from max |a|, max |c| of the staged
    // slice and max |x|, once per piece, not per element; or the piece is larger than the scale allows for.  Such pieces run under
    // the exchange locks like every piece of round 5; piece_mode tells the fold which kind of table a piece wrote.
    [[maybe_unused]] bool locks = !Fixed, single = false;
    [[maybe_unused]] long long iv = 0;            // Fixed: this lane's share of the reduced value, in units of 2^-28
    [[maybe_unused]] FixedScale fixed0{}, fixed1{};
    if constexpr (Fixed) {
        const uint32_t xm = (uint32_t) __builtin_amdgcn_readfirstlane((int) xm_early);
        const uint32_t E = xm >> 23;
        locks = E >= 255u || (E < (uint32_t) S0 + 1u && xm != 0u);
        const uint32_t am = s_guard[0], cm = s_guard[1];         // (max |a|, max |c| over the slice as bits: behind the staging barrier)
        const float ubound = __uint_as_float(am) * __uint_as_float(xm) + __uint_as_float(cm);
        const size_t piece_pages = (size_t) (range.f1 - range.f0) + (range.p1 - range.p0);
        if (am >= 0x7F800000u || cm >= 0x7F800000u || !(ubound < 1.0e18f) || (piece_pages << PS) >> (62 - S0)) locks = true;
        // a bucket that is ONE piece (the rule at the headline size: 256 buckets, 256 pieces): its sums are final -- converted here,
        // exactly as the fold would convert them, and written as floats (half the bytes out and back in; mode 1 like a piece under locks)
        single = range.pieces == 1u;
        if (threadIdx.x == 0) piece_mode[blockIdx.x] = (locks || single) ? 1u : 0u;
        if (!locks) {
            fixed0 = fixed_scale(S0, xm, false);
            fixed1 = fixed_scale(S0, xm, true);
            auto run_fixed = [&](auto body) {
                body.lds = lds_raw; body.pair_idx = pair_idx; body.x_b = x_b;
                asm volatile("v_mov_b32 %0, %1" : "=v"(body.m8) : "s"(((uint32_t) Bins - 1u) << 3));
                body.up0 = fixed0.upd; body.up1 = fixed1.upd; body.iacc = 0;
#ifdef EK_EARLY_TIMING
                fx_t[1] = __builtin_readcyclecounter();   // slice staged, planes cleared, guards decided
#endif
                walk_pages_dynamic<PS>(bl, range, body, &s_next);
#ifdef EK_EARLY_TIMING
                fx_t[2] = __builtin_readcyclecounter();   // this wave's walk done
#endif
                iv = body.iacc;
            };
            // (launched for the pairs whose functions are bounded by 1 only: bucketed_forward_adjoint_launch)
            if (map_op == EK_SIN && keep_op == EK_COS) run_fixed(EarlyFixed<PS, EK_SIN, EK_COS, Two>{});
            else if (map_op == EK_COS && keep_op == EK_SIN) run_fixed(EarlyFixed<PS, EK_COS, EK_SIN, Two>{});
            else if (map_op == EK_SIN) run_fixed(EarlyFixed<PS, EK_SIN, EK_SIN, Two>{});
            else run_fixed(EarlyFixed<PS, EK_COS, EK_COS, Two>{});
        }
    }
    if (locks) {
#define EK_EARLY_CASE(M, K) else if (map_op == M && keep_op == K) run(EarlyBody<T, V, M, K, Two>{});
        if (map_op == EK_SIN && keep_op == EK_COS) run(EarlyBody<T, V, EK_SIN, EK_COS, Two>{});
        EK_EARLY_CASE(EK_COS, EK_SIN) EK_EARLY_CASE(EK_LOG, EK_RCP) EK_EARLY_CASE(EK_SQRT, EK_RSQRT) EK_EARLY_CASE(EK_RCP, EK_RCP_SQR)
        EK_EARLY_CASE(EK_RSQRT, EK_RSQRT_CUBE) EK_EARLY_CASE(EK_SIN, EK_SIN) EK_EARLY_CASE(EK_COS, EK_COS)
        EK_EARLY_CASE(EK_EXP, EK_EXP) EK_EARLY_CASE(EK_SQRT, EK_SQRT) EK_EARLY_CASE(EK_RCP, EK_RCP) EK_EARLY_CASE(EK_RSQRT, EK_RSQRT)
        EK_EARLY_CASE(EK_LOG, EK_LOG) EK_EARLY_CASE(EK_ABS, EK_ABS) EK_EARLY_CASE(EK_NEG, EK_NEG)
#undef EK_EARLY_CASE
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += bucket_shfl_down(v, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (Fixed) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) iv += bucket_shfl_down(iv, d);
        if (lane == 0) wave_ipart[wave] = iv;
    }
    if (lane == 0) wave_part[wave] = v;
    __syncthreads();
#ifdef EK_EARLY_TIMING
    fx_t[3] = __builtin_readcyclecounter();               // every wave's walk done
#endif
    if (threadIdx.x < 64) {
        v = threadIdx.x < kBucketWaves ? wave_part[threadIdx.x] : T(0);
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) v += bucket_shfl_down(v, d);
        if constexpr (Fixed) {
            iv = threadIdx.x < kBucketWaves ? wave_ipart[threadIdx.x] : 0ll;
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) iv += bucket_shfl_down(iv, d);
        }
    }
    if constexpr (Fixed) bucket_finish_fixed_begin(iv, v, partials, fin.ticket, &s_last_fixed);
    // table c (0: sum of the kept function, 1: sum of x * kept function) of this piece at table_partials + (c * gridDim.x + piece) * Bins
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        T *out = table_partials + ((size_t) c * gridDim.x + blockIdx.x) * Bins;
        if constexpr (Fixed) {
            // the piece's slot holds Bins 64-bit sums (fixed point), or Bins floats at its start (a piece under locks)
            long long *out64 = reinterpret_cast<long long *>(table_partials) + ((size_t) c * gridDim.x + blockIdx.x) * Bins;
            if (!locks && single) {
                const long long *sums = reinterpret_cast<const long long *>(lds_raw + (c + 1) * kFixedRecBytes);
                const float back = c ? fixed1.back : fixed0.back;
                T *outf = reinterpret_cast<T *>(out64);
                for (int j = threadIdx.x; j < Bins; j += kBucketThreads) outf[j] = (float) sums[j] * back;
            } else if (!locks) {
                const long long *sums = reinterpret_cast<const long long *>(lds_raw + (c + 1) * kFixedRecBytes);
                for (int j = threadIdx.x; j < Bins; j += kBucketThreads) out64[j] = sums[j];
            } else {
                T *outf = reinterpret_cast<T *>(out64);
                for (int j = threadIdx.x; j < Bins; j += kBucketThreads) outf[j] = tables[2 * j + c];
            }
        } else {
            for (int j = threadIdx.x; j < Bins; j += kBucketThreads) out[j] = Paired ? tables[2 * j + c] : tables[c * Bins + j];
        }
    }
#ifdef EK_EARLY_TIMING
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_out = __builtin_readcyclecounter();
        atomicAdd(&g_early_timing[13], t_out - t_walk_end);                   // wave 0: end of its walk -> the piece's tables written (incl. the barrier = waiting for the slowest wave)
        atomicAdd(&g_early_timing[14], t_out - t_entry);                      // the whole workgroup
        atomicMax(&g_early_timing[15], t_out - t_entry);
    }
#endif
#ifdef EK_EARLY_TIMING
    if constexpr (Fixed) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fx_t[4] = __builtin_readcyclecounter();           // tables written
    }
#endif
    if constexpr (Fixed) bucket_finish_fixed_end(&s_last_fixed, partials, fin.ticket, fin.out, fin.active, fin.n, fin.zero_op, wave_part, wave_ipart, fin.counters);
    else bucket_finish<T, EK_HSUM>(v, partials, fin.ticket, fin.out, fin.active, fin.n, fin.zero_op, wave_part, fin.counters);
#ifdef EK_EARLY_TIMING
    if constexpr (Fixed) {
        if (threadIdx.x == 0 && fx_t[1]) {
            fx_t[5] = __builtin_readcyclecounter();
            atomicAdd(&g_early_timing[16], fx_t[0] - t_entry);      // which piece am I
            atomicAdd(&g_early_timing[17], fx_t[1] - fx_t[0]);      // staging, clearing, guards
            atomicAdd(&g_early_timing[18], fx_t[2] - fx_t[1]);      // wave 0's walk
            atomicAdd(&g_early_timing[19], fx_t[3] - fx_t[2]);      // waiting for the slowest wave
            atomicAdd(&g_early_timing[20], fx_t[4] - fx_t[3]);      // tables converted and written
            atomicAdd(&g_early_timing[21], fx_t[5] - fx_t[4]);      // finish (ticket; the last workgroup: the reduction)
            atomicAdd(&g_early_timing[22], 1ull);
            atomicMax(&g_early_timing[23], fx_t[5] - t_entry);
        }
    }
#endif
}

template <typename T>
int bucketed_forward_adjoint_launch(Bucketed *b, void *out, int map_op, int keep_op) {
    Context &c = ctx();
    const size_t Bins = b->bins();
    size_t lds = Bins * (sizeof(PairRec<T>) + 2 * sizeof(T));
    constexpr int VV = 1;            // one 4-element vector per lane and step (two: 0.151 ms against 0.143, same box)
    const int flip_a = b->flip_a(), flip_c = b->flip_c(), two = b->two_roundings();
    // 64-bit fixed-point sums: 4-byte elements in pages, a pair of functions bounded by 1 (sin / cos), a bucket whose records AND two
    // planes of 64-bit sums fit the LDS (24 B per entry: buckets of 4 Ki entries -- EK_BUCKETED_HINT_BOUNDED makes them), and the
    // partition's max |x| at hand.  ENOKI_HIP_EARLY_SUMS=locks: exchange locks for every pair (A/B runs).
    const bool bounded = (map_op == EK_SIN || map_op == EK_COS) && (keep_op == EK_SIN || keep_op == EK_COS);
    const bool fixed = sizeof(T) == 4 && early_fixed_enabled() && bounded && b->page_shift != 0 && b->active && Bins * 8 <= kFixedRecBytes;
    const size_t slot = fixed ? sizeof(long long) : sizeof(T);            // bytes per entry of a piece's partial table
    if (!b->early)
        if (int rc = ek_hip_malloc((size_t) 2 * b->max_pieces * Bins * slot + (fixed ? (size_t) b->max_pieces * sizeof(uint32_t) : 0), &b->early)) return rc;
    // scale: no piece is larger than ~1.5 n / (pieces aimed at) + a page per partition workgroup (k_page_directory cuts a bucket in
    // proportion to its population); the kernel checks its own piece against 2^(62 - S0) and runs it under locks otherwise
    const size_t piece_bound = 2 * (b->n / std::max<size_t>(1, b->max_pieces - (size_t) b->n_buckets)) + 65536;
    const int S0 = std::min(fixed_shift_for(piece_bound), 48);
    uint32_t *modes = fixed ? reinterpret_cast<uint32_t *>(static_cast<char *>(b->early) + (size_t) 2 * b->max_pieces * Bins * slot) : nullptr;
    auto go = [&](auto kernel) -> int {
        if (int rc = allow_big_lds(kernel, lds)) return rc;
        hipLaunchKernelGGL(kernel, dim3(b->max_pieces), dim3(kBucketThreads), lds, c.stream,
                           (T *) b->reduce_partials, (T *) b->early, (const T *) b->table_a, (const T *) b->table_c, b->table_size,
                           flip_a, flip_c, (const uint16_t *) b->pair_idx, (const T *) b->x_b, b->lists(), map_op, keep_op, b->shift,
                           b->template finish<T>(out, map_op), fixed ? b->active + (kPgMetaResultXmax - kPgMetaResult) : nullptr, S0, modes);
        return EK_OK;
    };
    int rc;
    if constexpr (sizeof(T) == 8) {
        rc = two ? go(k_bucket_pair_forward_adjoint<T, VV, 0, false, true>) : go(k_bucket_pair_forward_adjoint<T, VV, 0, false, false>);
    } else {
        if (fixed) lds = 0;            // (the fixed-point kernels' LDS is static)
#define EK_GO(PSV) (fixed ? (two ? go(k_bucket_pair_forward_adjoint<T, VV, PSV, true, true>) : go(k_bucket_pair_forward_adjoint<T, VV, PSV, true, false>)) \
                          : (two ? go(k_bucket_pair_forward_adjoint<T, VV, PSV, false, true>) : go(k_bucket_pair_forward_adjoint<T, VV, PSV, false, false>)))
        rc = b->page_shift == 6 ? EK_GO(6) : EK_GO(5);
#undef EK_GO
    }
    if (rc) return rc;
    EK_LAUNCH_CHECK("bucket_pair_fma_reduce_adjoint", b->n,
                    b->n * (sizeof(uint16_t) + sizeof(T)) + (b->table_c ? 2 : 1) * b->table_size * sizeof(T) + (size_t) 2 * b->max_pieces * Bins * slot);
    b->launched_reducing();
    b->early_fixed = fixed;
    b->early_S0 = S0;
    b->early_modes = modes;
    b->has_early = true;
    b->early_op = keep_op;
    if (!b->ticket) {
        hipLaunchKernelGGL((k_bucket_reduce_final<T, EK_HSUM>), dim3(1), dim3(256), 0, c.stream, (T *) out, (const T *) b->reduce_partials,
                           b->max_pieces, b->masked_ptr(), b->n, map_op);
        EK_LAUNCH_CHECK("reduce_stage2", (size_t) b->max_pieces, (size_t) b->max_pieces * sizeof(T) + sizeof(T));
    }
    return EK_OK;
}

template int bucketed_forward_adjoint_launch<float>(Bucketed *, void *, int, int);
template int bucketed_forward_adjoint_launch<double>(Bucketed *, void *, int, int);

} // namespace ek

#ifdef EK_EARLY_TIMING
/// measurement builds only: reads and clears the phase counters of k_bucket_pair_forward_adjoint (summed over the waves)
extern "C" EK_API int ek_hip_debug_early_timing(unsigned long long *out8) {
    (void) hipDeviceSynchronize();
    (void) hipMemcpyFromSymbol(out8, HIP_SYMBOL(ek::g_early_timing), 32 * sizeof(unsigned long long));
    unsigned long long zero[32] = {};
    (void) hipMemcpyToSymbol(HIP_SYMBOL(ek::g_early_timing), zero, sizeof(zero));
    return EK_OK;
}
#endif
