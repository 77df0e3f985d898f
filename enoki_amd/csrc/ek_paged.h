// Single-pass bucket partition into PAGES (round 4).
//
// What it replaces: k_bin_count + k_bin_scan_rows + k_bin_scan_buckets + k_bin_partition (ek_binned.h) on the bucket-ordered
// path -- the role of the reference's sort + run-length partition (src/cuda/horiz.cu:35-122), done in ONE streaming pass.
//
// Why pages.  A partition whose output is one contiguous run per bucket has to know every bucket's size before the first
// element is written: a count pass over the indices (4 B/elt) and two scans.  And a workgroup that writes its share of a
// bucket's run appends ~64 elements per tile at an arbitrary alignment: partially written cache lines, 2-byte stores,
// 1.19x write amplification (profiles/rocprof_pmc_r03.txt) -- the write-out was 56 % of the kernel.
// Here the output of a bucket is a LIST OF PAGES instead: a page is 64 elements (32 when the table has more than 128
// buckets) = one full 128-byte line of 16-bit bucket-local indices + two full lines of values, written exactly once by 16-byte
// stores.  Every workgroup owns a contiguous range of page slots and hands them out itself, so no offset depends on another
// workgroup: no count pass, no scan, no look-back, no global atomics in the loop.
//
//   k_page_partition   one 1024-thread workgroup per CU walks its chunk of (index, x) in tiles of 4096 elements.  Per bucket the
//                      LDS holds a circular buffer of `cap` elements (16 Ki elements over all buckets = 96 KiB); an element
//                      takes its slot with ONE returning LDS atomic on a {origin, fill} word, complete pages leave as 16-byte
//                      vectors.  A tile that brings a bucket more than its buffer holds (skewed indices) takes further rounds
//                      of the same three phases.  What is left at the end of the chunk leaves as one partially filled page per
//                      bucket.  The workgroup finally lists its pages bucket by bucket (wlist) -- the order is that of the
//                      input, nothing depends on timing.                               idx 4 + x 4 read, 6 written per element
//   k_page_directory   one workgroup per bucket gathers the workgroups' lists into the bucket's page list (full pages first,
//                      partially filled ones with their element count behind) and cuts the bucket into pieces for the
//                      consumers.                                                                        ~8 B per PAGE
//
// Consumers (bucketed.hip) walk page lists: 16 lanes take a page, four elements each -- the same 8- and 16-byte vector loads
// as over a contiguous run.
#pragma once
#include "ek_binned.h"

namespace ek {

constexpr int kPgThreads = 1024;
constexpr int kPgTile = 4 * kPgThreads;
constexpr int kPgLdsElems = 16384;             // elements staged per workgroup, all buckets together (128 KiB of 8-byte records)
constexpr uint32_t kNoPage = 0xFFFFFFFFu;
// The counter block of a partition: kPgReplicas copies of the per-bucket page totals ([replica][full | partially filled][bucket]; a
// workgroup adds to copy w % kPgReplicas, the directory launch adds the copies up), then the meta row.  ONE copy meant 256 workgroups
// queueing on every word: device-scope atomics on one address are served one after the other (~40 ns each), and the workgroups wait
// for their acknowledgements before they may list their pages -- 10-12 us at the end of every workgroup, which an 8 Mi-element shard
// (all workgroups finishing together) shows in full: partition 49 -> 37 us with 8 copies (profiles/probe_paged_r06.txt, section 6).
#ifndef EK_PG_REPLICAS
#define EK_PG_REPLICAS 8            // (measurement builds: -DEK_PG_REPLICAS=1 in EVERY unit that includes this header)
#endif
constexpr int kPgReplicas = EK_PG_REPLICAS;
constexpr uint32_t kPgTotalsWords = (uint32_t) kPgReplicas * 2u * kMaxBuckets;
constexpr uint32_t kPgMetaBase = kPgTotalsWords;                       // the meta row (kPgMeta* below)
constexpr uint32_t kPgCounterWords = kPgTotalsWords + kMaxBuckets;     // what is zero when a launch starts
// words of the meta row of the counter block (gtotal + kPgMetaBase): what the partition accumulates, the tickets, what the consumers read
enum { kPgMetaAccum = 0 /* [2], unused since the accumulators are kept per replica */, kPgMetaFinishTicket = 2, kPgMetaAccumXmax = 3 /* unused */,
       kPgMetaResult = 4 /* [2] */, kPgMetaResultXmax = 6,
       kPgMetaAccumRep = 16 /* [kPgReplicas][4]: elements kept, non-finite flag, max |x| bits, - : one set per copy w % kPgReplicas, like the page totals */,
       kPgMetaAccumRepEnd = kPgMetaAccumRep + 4 * kPgReplicas };
// Workgroup w is dispatched to XCD w % 8, and on every box seen so far one XCD runs the same streaming work ~9 % slower than the
// other seven (profiles/probe_paged_phases_r05.txt): with equal chunks the kernel ends when that XCD ends.  The eight CLASSES
// w % 8 therefore get shares of the tiles in proportion to WEIGHTS (Q16, 65536 = 1) that live on the device and are fed back by
// the directory launch from the loop durations the workgroups stamp -- no host round trip, no assumption about which XCD (or
// whether any) is the slow one: equal durations leave the weights where they are.  OPT-IN (tuning "xcd_balance", default 0): on the
// boxes it could be measured on it changed nothing (profiles/probe_xcd_balance_r05.txt), see DESIGN.md section 9 item 2.
constexpr int kPgClasses = 8, kPgClassStamps = 16;      // class block: weights[8] | dealt flag | - | stamps[W] from word 16
constexpr uint32_t kPgWeightOne = 65536u, kPgWeightMin = 57672u /* 0.88 */, kPgWeightMax = 73400u /* 1.12 */, kPgWeightBand = 2621u /* 0.04 */;

template <typename T> struct PagedOut {
    uint16_t *lp;          // pages of bucket-local indices
    T *xp;                 // pages of values, same positions
    uint32_t *wdir;        // [W][slots]  page slot -> (sequence number within its bucket) << 8 | bucket
    uint32_t *wlist;       // [W][slots]  the workgroup's full pages, bucket by bucket, in input order
    uint32_t *cnt_full;    // [n_buckets][W]  full pages of the workgroup per bucket
    uint32_t *loff;        // [n_buckets][W]  where they start in wlist[w]
    uint32_t *part;        // [n_buckets][W]  page << 6 | (count - 1) of the partially filled page, or kNoPage
    uint32_t *gtotal;      // [kPgReplicas][2][kMaxBuckets]  full / partially filled pages per bucket over the workgroups of a replica (zero at launch)
    uint32_t *active;      // [0] number of elements kept, [1] != 0: a lane whose mask bit is clear carries a non-finite x (zeroed by the host)
    uint32_t lo, span;     // only indices in [lo, lo + span) are kept, rebased to lo: the table (lo = 0, span = its size) or a slice of it
    const uint32_t *class_w;   // [kPgClasses] weights of the classes w % 8 (nullptr: equal chunks, `chunk` elements each)
    uint32_t *class_stamp;     // [W] loop duration of every workgroup in 100 MHz ticks (nullptr: not recorded)
    uint32_t class_band;       // weights that all stay within this distance of 1 (Q16) count as equal
    uint32_t wdir_lds;         // != 0: the workgroup's wdir entries live in the LDS behind the records (>= slots words) instead of out.wdir
#ifdef EK_PG_TIMING
    unsigned long long *dbg;   // [W][2][8] cycles per phase of waves 0 and 1 (measurement builds only)
#endif
};

#ifdef EK_PG_TIMING
#define EK_PG_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define EK_PG_T(k) do { } while (0)
#endif

using PgV4 = __attribute__((ext_vector_type(4))) uint32_t;
using PgV2 = __attribute__((ext_vector_type(2))) uint32_t;

// IndexOnly (round 6): no value stream -- the pages carry the bucket-local INDEX as a 32-bit word (the value pages' slots), records of
// 4 bytes (twice the buffer per bucket: 256 buckets keep 64-element pages), nothing is written to the 16-bit pages.  This is the
// partition of an index array alone (ek_hip_index_partition_*: cfg4's pixel permutation): idx 4 read, 4 written per element.
template <typename T, typename I, int PS, bool HasMask, bool IndexOnly = false>
__global__ __launch_bounds__(kPgThreads) void k_page_partition(PagedOut<T> out, const I *__restrict__ index, Arg<uint8_t> mask,
                                                               const T *__restrict__ x, size_t n, size_t chunk, int n_buckets,
                                                               int shift, uint32_t cap, uint32_t slots, int vec_ok) {
    static_assert(sizeof(T) == 4, "pages carry 4-byte values");
    constexpr uint32_t Page = 1u << PS;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    using Rec = std::conditional_t<IndexOnly, uint32_t, unsigned long long>;
    Rec *rec = reinterpret_cast<Rec *>(lds_raw);                                // [n_buckets][cap] of {value bits, local index} (IndexOnly: the local index alone)
    __shared__ uint32_t cnt[kMaxBuckets];      // origin << 16 | fill of the bucket's circular buffer
    __shared__ uint32_t npg[kMaxBuckets];      // full pages written so far per bucket
    __shared__ uint32_t dbase[kMaxBuckets];    // overflowing bucket: first of its directly written pages (slot within the round)
    __shared__ uint32_t dtail[kMaxBuckets];    //                     fill from which its elements stay in the LDS
    __shared__ uint32_t jobs[1024];            // page slots of this round: page in the buffer << 8 | bucket (kNoPage: written directly)
    __shared__ uint32_t s_pages, s_jobs, s_over;

    const uint32_t W = gridDim.x, w = blockIdx.x;
#ifdef EK_PG_TIMING
    const unsigned long long t_start = wall_clock64();
#endif
    size_t begin = (size_t) w * chunk < n ? (size_t) w * chunk : n, end = begin + chunk < n ? begin + chunk : n;
    if (out.class_w) {
        // the tiles of the whole input, dealt to the classes by weight and equally to a class's workgroups (W is a multiple of 8)
        const uint64_t NT = (n + kPgTile - 1) / kPgTile, per = W / kPgClasses, cls = w % kPgClasses, r = w / kPgClasses;
        uint64_t before = 0, mine = 0, all = 0;
#pragma unroll
        for (int k = 0; k < kPgClasses; ++k) {
            const uint64_t wk = out.class_w[k];
            all += wk;
            before += (uint64_t) k < cls ? wk : 0u;
            mine = (uint64_t) k == cls ? wk : mine;
        }
        // (class_w[kPgClasses]: the weights are being dealt -- set by the feedback once a class is more than the band away from 1
        // and cleared when all are back within half of it; the launch-to-launch scatter of a balanced box stays below and the
        // chunks stay equal, as without weights)
        if (out.class_w[kPgClasses]) {
            const uint64_t t0 = NT * before / all, t1 = NT * (before + mine) / all, tx = t1 - t0;
            const uint64_t b0 = (t0 + tx * r / per) * kPgTile, b1 = (t0 + tx * (r + 1) / per) * kPgTile;
            begin = b0 < n ? (size_t) b0 : n;
            end = b1 < n ? (size_t) b1 : n;
        }
    }
    const size_t wbase = (size_t) w * slots;                     // first page slot of this workgroup
    const uint32_t lowmask = (1u << shift) - 1u, cap_pages = cap >> PS, cap_shift = 31u - (uint32_t) __builtin_clz(cap);
    const uint32_t spare = (uint32_t) n_buckets << cap_shift;         // one record behind the buffers
    // The workgroup's own page directory (slot -> sequence number << 8 | bucket) is written while pages are announced and read
    // once, by the lists phase at the end.  Through global memory that read has to wait for ALL of the workgroup's stores (the
    // counter that orders them is in-order: 8-12 us at the end of every workgroup, profiles/probe_paged_phases_r05.txt); when
    // the entries fit the LDS behind the records (inputs up to ~64 Mi elements) they stay there and nobody waits.
    uint32_t *wd = reinterpret_cast<uint32_t *>(rec + spare + 2);
    for (int b = threadIdx.x; b < kMaxBuckets; b += kPgThreads) { cnt[b] = 0; npg[b] = 0; }
    if (threadIdx.x == 0) { s_pages = 0; s_jobs = 0; s_over = 0; }
    __syncthreads();

    auto make_rec = [&](uint32_t xv, uint32_t local) -> Rec {
        if constexpr (IndexOnly) return local; else return (unsigned long long) xv | ((unsigned long long) local << 32);
    };
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    // a tile as it arrives: nothing is decoded before the tile is placed, so that two tiles of loads stay in flight
    // IndexOnly places TWO tiles per round (8 elements per lane, 8192 per workgroup): 256 buckets of 128 records take 32 arrivals per
    // round on average, and the two barriers, the waits in front of them and the page announcements are paid once for twice the
    // elements (the index partition moves 9 B per element: it is bound by the rounds, not by HBM).  With a value stream a bucket's
    // buffer is half as deep in elements per bucket and round -- one tile per round as before.
    constexpr int NT = IndexOnly ? 2 : 1, NE = 4 * NT;
    struct Raw { I pi[4]; PgV4 xv; uint32_t m; };
    struct Tile { uint32_t ix[NE], xv[NE], on; };          // xv: the values' bits;  elements 4 s .. 4 s + 3: sub-tile s
    auto load_raw = [&](size_t base, Raw &r) {
        const size_t e = base + (size_t) threadIdx.x * 4;
        load4<I, true>(index + e, r.pi);
        if constexpr (!IndexOnly) r.xv = __builtin_nontemporal_load(reinterpret_cast<const PgV4 *>(x + e));
        else r.xv = PgV4{ 0u, 0u, 0u, 0u };
        if constexpr (HasMask) r.m = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(mask.ptr + e));
        else r.m = 0;
    };
    const uint32_t win_lo = out.lo, win_span = out.span;
    // A lane whose mask bit is clear gathers 0 from both tables (cuda.h:845-864): its u is fma(0, x, 0) -- 0 for a finite x, NaN
    // for an infinite or NaN x, and the reference's reduction then is NaN (dynamic.h:632-650).  The lane is dropped here; that
    // it would have produced a NaN is remembered (one compare per masked-out lane) and applied by the final reduction.
    uint32_t nonfinite_masked = 0;
    // max |x| over the chunk, as bits (|x| as an integer is monotonic; a NaN or an infinity comes out on top): what bounds the terms
    // x f'(u) of the adjoint sums when they are formed in fixed point (bucketed_early.hip).  Dropped lanes are included -- the
    // bound only gets more careful.
    uint32_t xmax_bits = 0;
    auto decode = [&](const Raw &r, Tile &t, int sub) {
        const int o = 4 * sub;
        if (sub == 0) t.on = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t.ix[o + j] = (uint32_t) r.pi[j];
            t.xv[o + j] = r.xv[j];
            if constexpr (!IndexOnly) xmax_bits = max(xmax_bits, t.xv[o + j] & 0x7FFFFFFFu);
            if constexpr (HasMask) {
                const uint32_t on = ((r.m >> (8 * j)) & 0xFFu) ? 1u : 0u;
                t.on |= on << (o + j);
                if constexpr (!IndexOnly) nonfinite_masked |= (on ^ 1u) & (uint32_t) ((t.xv[o + j] & 0x7F800000u) == 0x7F800000u);
            }
        }
        if constexpr (!HasMask) t.on |= (sm ? 0xFu : 0u) << o;
        // indices outside the table (or outside this slice of it) are dropped like masked-out lanes: an out-of-range index must
        // not reach another bucket's counters (what it gathers / scatters is unspecified anyway, cuda.h:845-905)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t.ix[o + j] -= win_lo;
            if (t.ix[o + j] >= win_span) { t.on &= ~(1u << (o + j)); t.ix[o + j] = 0; }
        }
    };
    auto load_ragged = [&](size_t base, Tile &t, int sub) {
        const size_t e = base + (size_t) threadIdx.x * 4;
        const int o = 4 * sub;
        if (sub == 0) t.on = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t.ix[o + j] = 0; t.xv[o + j] = 0;
            if (e + j < end) {
                t.ix[o + j] = (uint32_t) index[e + j];
                if constexpr (!IndexOnly) { t.xv[o + j] = __builtin_bit_cast(uint32_t, x[e + j]); xmax_bits = max(xmax_bits, t.xv[o + j] & 0x7FFFFFFFu); }
                const uint32_t on = (mask.vec ? mask.ptr[e + j] : sm) ? 1u : 0u;
                t.on |= on << (o + j);
                if constexpr (HasMask && !IndexOnly) nonfinite_masked |= (on ^ 1u) & (uint32_t) ((t.xv[o + j] & 0x7F800000u) == 0x7F800000u);
                t.ix[o + j] -= win_lo;
                if (t.ix[o + j] >= win_span) { t.on &= ~(1u << (o + j)); t.ix[o + j] = 0; }
            }
        }
    };

    // the pages in slots [ps0, ps0 + njobs) leave the LDS: one page per group of lanes, four elements per lane (16 bytes of
    // values, 8 bytes of indices)
    auto write_out = [&](uint32_t ps0, uint32_t njobs) {
        constexpr int LX = (int) (Page / 4);
        const uint32_t g = threadIdx.x / LX, i = threadIdx.x % LX;
        for (uint32_t j = g; j < njobs; j += kPgThreads / LX) {
            const uint32_t jb = jobs[j];
            if (jb == kNoPage) continue;
            const uint32_t src = ((jb & 0xFFu) << cap_shift) + ((jb >> 8) << PS) + 4 * i;
            const size_t at = ((wbase + ps0 + j) << PS) + 4 * i;
            if constexpr (IndexOnly) {
                *reinterpret_cast<PgV4 *>(out.xp + at) = *reinterpret_cast<const PgV4 *>(rec + src);
            } else {
                const PgV4 r01 = *reinterpret_cast<const PgV4 *>(rec + src), r23 = *reinterpret_cast<const PgV4 *>(rec + src + 2);
                const PgV4 vx = { r01[0], r01[2], r23[0], r23[2] };
                const PgV2 vl = { r01[1] | (r01[3] << 16), r23[1] | (r23[3] << 16) };
                *reinterpret_cast<PgV4 *>(out.xp + at) = vx;
                *reinterpret_cast<PgV2 *>(out.lp + at) = vl;
            }
        }
    };

    uint32_t ps0 = 0;
#ifdef EK_PG_TIMING
    unsigned long long tacc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tlast = __builtin_readcyclecounter();
#endif
    // One round of two barriers per tile.  An element takes a fill number of its bucket's buffer with ONE returning LDS atomic
    // and is staged at origin + fill; the element that takes the last fill of a page announces the page (slot, job, directory
    // entry) on the spot -- no serial bookkeeping.  After the barrier the announced pages leave as 16-byte vectors and their
    // announcers move the buckets' {origin, fill} words on.
    // A tile that brings a bucket more than its buffer holds (skewed indices) sets a flag; such a tile takes one more barrier:
    // the bucket's elements beyond the buffer are complete pages + a rest, the pages get slots like any other and their
    // elements go STRAIGHT to global memory (4- and 2-byte stores, as the contiguous-run partition writes all of its output),
    // the rest is staged once the buffer has been written out.
    auto process = [&](const Tile &t) {
        uint32_t old[NE], pending = 0, done = 0;
        EK_PG_T(0);                       // waiting for the tile's loads + decode
#pragma unroll
        for (int k = 0; k < NE; ++k) old[k] = ((t.on >> k) & 1u) ? atomicAdd(&cnt[t.ix[k] >> shift], 1u) : 0u;
        // staged without a branch: an element that found its bucket's buffer full (or is masked out) writes to a spare record
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const uint32_t b = t.ix[k] >> shift, fill = old[k] & 0xFFFFu, pos = ((old[k] >> 16) + fill) & (cap - 1u);
            const bool on = (t.on >> k) & 1u, ok = on && fill < cap;
            rec[ok ? ((b << cap_shift) | pos) : spare] = make_rec(t.xv[k], t.ix[k] & lowmask);
            pending |= (on && !ok) ? 1u << k : 0u;
            done |= (ok && ((fill + 1u) & (Page - 1u)) == 0u) ? 1u << k : 0u;
        }
        if (pending) s_over = 1u;
        // The elements that took the last fill of a page announce it.  ONE slot request per WAVE (round 6): every announcing lane
        // used to add to s_pages itself -- 64 (128 with 32-element pages) returning atomics on ONE LDS address per tile, which the
        // LDS serves one after the other (~5 cycles each: the whole difference between 64- and 32-element pages,
        // profiles/probe_paged_r06.txt).  Now the wave counts its pages with ballots, lane 0 requests the range and every
        // announcing lane finds its slots by its rank among the wave's announcements.
        unsigned long long dm[NE];
        uint32_t before[NE + 1];
        before[0] = 0;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            dm[k] = __builtin_amdgcn_ballot_w64(((done >> k) & 1u) != 0);
            before[k + 1] = before[k] + (uint32_t) __builtin_popcountll(dm[k]);
        }
        if (before[NE]) {
            uint32_t wave_first = 0;
            if ((threadIdx.x & 63) == 0) wave_first = atomicAdd(&s_pages, before[NE]);
            wave_first = (uint32_t) __builtin_amdgcn_readfirstlane((int) wave_first);
            if (done) {
                // rank of this lane's announcements: slot k of every lane before slot k + 1 of any (the order inside a wave is immaterial)
                auto below = [](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)); };
                uint32_t seq[NE];
#pragma unroll
                for (int k = 0; k < NE; ++k) seq[k] = ((done >> k) & 1u) ? npg[t.ix[k] >> shift] : 0u;
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    if ((done >> k) & 1u) {
                        const uint32_t ps = wave_first + before[k] + below(dm[k]);
                        const uint32_t b = t.ix[k] >> shift, fill = old[k] & 0xFFFFu, pos = ((old[k] >> 16) + fill) & (cap - 1u);
                        jobs[ps - ps0] = b | ((pos >> PS) << 8);
                        const uint32_t entry = ((seq[k] + (fill >> PS)) << 8) | b;
                        if (out.wdir_lds) wd[ps] = entry; else out.wdir[wbase + ps] = entry;
                    }
                }
            }
        }
        EK_PG_T(1);                       // placement
        __syncthreads();
        EK_PG_T(2);                       // barrier 1
        const bool over = s_over != 0u;
        if (over) {
            if ((int) threadIdx.x < n_buckets) {
                const uint32_t b = threadIdx.x, c = cnt[b], f = c & 0xFFFFu;
                if (f > cap) {
                    const uint32_t nd = (f >> PS) - cap_pages, p = nd ? atomicAdd(&s_pages, nd) : ps0, seq0 = npg[b] + cap_pages;
                    for (uint32_t q = 0; q < nd; ++q) {
                        jobs[p - ps0 + q] = kNoPage;
                        if (out.wdir_lds) wd[p + q] = ((seq0 + q) << 8) | b; else out.wdir[wbase + p + q] = ((seq0 + q) << 8) | b;
                    }
                    dbase[b] = p - ps0;
                    dtail[b] = (f >> PS) << PS;
                }
            }
            __syncthreads();
        }
        EK_PG_T(3);
        const uint32_t ps1 = s_pages;
        write_out(ps0, ps1 - ps0);
        if (!over) {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                if ((done >> k) & 1u) {
                    const uint32_t b = t.ix[k] >> shift;
                    atomicAdd(&cnt[b], (Page << 16) - Page);          // origin + Page, fill - Page
                    atomicAdd(&npg[b], 1u);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                if ((pending >> k) & 1u) {
                    const uint32_t b = t.ix[k] >> shift, fill = old[k] & 0xFFFFu;
                    if (fill < dtail[b]) {
                        const size_t at = ((wbase + ps0 + dbase[b]) << PS) + (fill - cap);
                        if constexpr (IndexOnly) {
                            reinterpret_cast<uint32_t *>(out.xp)[at] = t.ix[k] & lowmask;
                        } else {
                            reinterpret_cast<uint32_t *>(out.xp)[at] = t.xv[k];
                            out.lp[at] = (uint16_t) (t.ix[k] & lowmask);
                        }
                        pending &= ~(1u << k);
                    } else {
                        old[k] = (old[k] & 0xFFFF0000u) | (fill - dtail[b]);     // its place in the emptied buffer
                    }
                }
            }
            if ((int) threadIdx.x < n_buckets) {
                const uint32_t b = threadIdx.x, c = cnt[b], f = c & 0xFFFFu, org = c >> 16, np = f >> PS;
                if (f > cap) cnt[b] = (org << 16) | (f & (Page - 1u));
                else cnt[b] = (((org + (np << PS)) & 0xFFFFu) << 16) | (f & (Page - 1u));
                npg[b] += np;
            }
            if (threadIdx.x == 0) s_over = 0u;
        }
        ps0 = ps1;
        EK_PG_T(5);                       // write-out
        __syncthreads();
        EK_PG_T(6);                       // barrier 2
        if (pending) {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                if ((pending >> k) & 1u) {
                    const uint32_t b = t.ix[k] >> shift;
                    const uint32_t slot = (b << cap_shift) | (((old[k] >> 16) + (old[k] & 0xFFFFu)) & (cap - 1u));
                    rec[slot] = make_rec(t.xv[k], t.ix[k] & lowmask);
                }
            }
        }
    };

#ifdef EK_PG_TIMING
    const unsigned long long t_loop = wall_clock64();
#endif
    const unsigned long long stamp0 = out.class_stamp ? wall_clock64() : 0ull;
    // whole tiles of 16-byte aligned operands: two tiles of loads in flight ahead of the one that is being placed.  The two
    // register sets alternate (a rotation by moves would have to wait for the loads it moves).
    // (Tiles handed out round robin -- at every moment the W workgroups reading W consecutive 16-KiB stretches instead of W
    // stretches a whole chunk apart -- were measured in round 5: 188-191 us against 185-186, profiles/probe_paged_r05.txt.)
    size_t base = begin;
    const size_t ntiles = vec_ok ? (end - begin) / kPgTile : 0, ngroups = ntiles / NT;        // rounds of NT whole tiles
    if (ngroups > 0) {
        Raw buf0[NT], buf1[NT];
#pragma unroll
        for (int sub = 0; sub < NT; ++sub) load_raw(begin + (size_t) sub * kPgTile, buf0[sub]);
        if (ngroups > 1) {
#pragma unroll
            for (int sub = 0; sub < NT; ++sub) load_raw(begin + (size_t) (NT + sub) * kPgTile, buf1[sub]);
        }
        for (size_t i = 0; i < ngroups; i += 2) {
            {
                Tile t;
#pragma unroll
                for (int sub = 0; sub < NT; ++sub) decode(buf0[sub], t, sub);
                if (i + 2 < ngroups) {
#pragma unroll
                    for (int sub = 0; sub < NT; ++sub) load_raw(begin + ((i + 2) * NT + sub) * kPgTile, buf0[sub]);
                }
                process(t);
            }
            if (i + 1 < ngroups) {
                Tile t;
#pragma unroll
                for (int sub = 0; sub < NT; ++sub) decode(buf1[sub], t, sub);
                if (i + 3 < ngroups) {
#pragma unroll
                    for (int sub = 0; sub < NT; ++sub) load_raw(begin + ((i + 3) * NT + sub) * kPgTile, buf1[sub]);
                }
                process(t);
            }
        }
        base = begin + ngroups * NT * kPgTile;
    }
    for (; base < end; base += (size_t) NT * kPgTile) {
        Tile t;
#pragma unroll
        for (int sub = 0; sub < NT; ++sub) load_ragged(base + (size_t) sub * kPgTile, t, sub);
        process(t);
    }
    if (out.class_stamp && threadIdx.x == 0) out.class_stamp[w] = (uint32_t) (wall_clock64() - stamp0);

#ifdef EK_PG_TIMING
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 128)
        for (int k = 0; k < 8; ++k) out.dbg[((size_t) w * 2 + (threadIdx.x >> 6)) * 8 + k] = tacc[k];
    if (threadIdx.x == 0) { out.dbg[(size_t) W * 16 + w * 4 + 0] = t_start; out.dbg[(size_t) W * 16 + w * 4 + 1] = t_loop; out.dbg[(size_t) W * 16 + w * 4 + 2] = wall_clock64(); }
#endif
    if constexpr (HasMask) {
        if (nonfinite_masked) atomicOr(out.active + (kPgMetaAccumRep - kPgMetaAccum) + 4u * (w % kPgReplicas) + 1u, 1u);
    }
    {
        // (per wave into the LDS word that announced the overflow rounds -- free by now --, ONE global atomic per workgroup behind the
        //  barrier below: 4096 atomics on one address cost 30 us of every launch)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) xmax_bits = max(xmax_bits, (uint32_t) __shfl_xor((int) xmax_bits, d, 64));
        if (!IndexOnly && (threadIdx.x & 63) == 0 && xmax_bits) atomicMax(&s_over, xmax_bits);
    }
    // what is left: one partially filled page per bucket; the workgroup's page lists
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        uint32_t fl[4], org[4], full[4], tot = 0, ftot = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = 4 * l + j;
            fl[j] = org[j] = full[j] = 0;
            if (b < n_buckets) {
                const uint32_t c = cnt[b];
                org[j] = c >> 16;
                fl[j] = c & 0xFFFFu;
                full[j] = npg[b];
            }
            tot += fl[j] ? 1u : 0u;
            ftot += full[j];
        }
        uint32_t incl = tot, fincl = ftot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64), fup = __shfl_up(fincl, d, 64);
            if (l >= d) { incl += up; fincl += fup; }
        }
        uint32_t p = incl - tot, lo = fincl - ftot;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = 4 * l + j;
            if (b < n_buckets) {
                uint32_t entry = kNoPage;
                if (fl[j]) {
                    jobs[p] = (uint32_t) b | (((org[j] >> PS) & (cap_pages - 1u)) << 8);
                    entry = (uint32_t) ((wbase + ps0 + p) << 6) | (fl[j] - 1u);
                    ++p;
                    atomicAdd(&out.gtotal[(w % kPgReplicas) * 2u * kMaxBuckets + kMaxBuckets + b], 1u);
                }
                out.part[(size_t) b * W + w] = entry;
                out.cnt_full[(size_t) b * W + w] = full[j];
                out.loff[(size_t) b * W + w] = lo;
                if (full[j]) atomicAdd(&out.gtotal[(w % kPgReplicas) * 2u * kMaxBuckets + b], full[j]);
                cnt[b] = lo;                                    // from here on: where the bucket's pages start in wlist[w]
                lo += full[j];
            }
        }
        if (l == 63) s_jobs = incl;
        // elements this workgroup kept (masked-out entries are dropped): complete pages + what is left
        uint32_t kept = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) kept += (full[j] << PS) + fl[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) kept += __shfl_xor(kept, d, 64);
        if (l == 0 && kept) atomicAdd(out.active + (kPgMetaAccumRep - kPgMetaAccum) + 4u * (w % kPgReplicas), kept);
    }
#ifdef EK_PG_TIMING
    if (threadIdx.x == 0) out.dbg[(size_t) W * 16 + W * 4 + w * 4 + 0] = wall_clock64();
#endif
    // wdir in global memory was written by this workgroup: its stores have to be done, and it is read back past the L1
    if (!out.wdir_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!IndexOnly && threadIdx.x == 0 && s_over) atomicMax(out.active + (kPgMetaAccumRep - kPgMetaAccum) + 4u * (w % kPgReplicas) + 2u, s_over);
#ifdef EK_PG_TIMING
    if (threadIdx.x == 0) out.dbg[(size_t) W * 16 + W * 4 + w * 4 + 1] = wall_clock64();
#endif
    const uint32_t nfull = ps0;
    write_out(ps0, s_jobs);
#ifdef EK_PG_TIMING
    if (threadIdx.x == 0) out.dbg[(size_t) W * 16 + W * 4 + w * 4 + 2] = wall_clock64();
#endif
    for (uint32_t s0 = threadIdx.x; s0 < nfull; s0 += 4 * kPgThreads) {
        uint32_t d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t s = s0 + u * kPgThreads;
            d[u] = s >= nfull ? 0u : out.wdir_lds ? wd[s] : __hip_atomic_load(out.wdir + wbase + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t s = s0 + u * kPgThreads;
            if (s < nfull) out.wlist[wbase + cnt[d[u] & 0xFFu] + (d[u] >> 8)] = (uint32_t) (wbase + s);
        }
    }
#ifdef EK_PG_TIMING
    if (threadIdx.x == 0) out.dbg[(size_t) W * 16 + w * 4 + 3] = wall_clock64();
#endif
}

// inclusive scan of one value per thread over a workgroup of 256 threads (wave shuffles + 4 wave totals)
__device__ __forceinline__ uint32_t pg_block_scan(uint32_t v, uint32_t *wave_tot /* [4] */, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    __syncthreads();                      // the previous scan's totals have been read
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t tk = wave_tot[k];
        before += k < wave ? tk : 0u;
        all += tk;
    }
    total = all;
    return incl + before;
}

constexpr int kPgDirSlices = 4;

// The bucket's page list = the workgroups' lists one after the other (full pages), then the partially filled pages; bucket
// bases of both lists; pieces for the consumers (as k_bin_scan_buckets: a share of `target_pieces` in proportion to the
// bucket's population, at least one when it is not empty).  Grid: (bucket, slice of the bucket's list).
static __global__ __launch_bounds__(256) void k_page_directory(uint32_t *__restrict__ glist_full, uint32_t *__restrict__ glist_part,
                                                               uint32_t *__restrict__ base_full, uint32_t *__restrict__ base_part,
                                                               uint32_t *__restrict__ piece_prefix,
                                                               uint32_t *__restrict__ gtotal, const uint32_t *__restrict__ cnt_full,
                                                               const uint32_t *__restrict__ loff, const uint32_t *__restrict__ part,
                                                               const uint32_t *__restrict__ wlist, uint32_t W, uint32_t slots,
                                                               int n_buckets, uint32_t target_pieces,
                                                               uint32_t *__restrict__ class_w, const uint32_t *__restrict__ class_stamp,
                                                               uint32_t class_band) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t row[1025], lrow[1024];
    __shared__ uint32_t s_fb, s_pb, s_f;
    const int t = threadIdx.x, b = blockIdx.x, slice = blockIdx.y;
    // everything this workgroup reads from global memory, requested up front
    uint32_t f = 0, p = 0;
    if (t < n_buckets) {
#pragma unroll
        for (int r = 0; r < kPgReplicas; ++r) { f += gtotal[r * 2 * kMaxBuckets + t]; p += gtotal[r * 2 * kMaxBuckets + kMaxBuckets + t]; }
    }
    uint32_t c[4], ent[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t wq = 4 * t + k;
        c[k] = wq < W ? cnt_full[(size_t) b * W + wq] : 0u;
        lo[k] = wq < W ? loff[(size_t) b * W + wq] : 0u;
        ent[k] = (wq < W && slice == 0) ? part[(size_t) b * W + wq] : kNoPage;
    }
    uint32_t total_f, total_p, total_q, total_c, total_h;
    const uint32_t fi = pg_block_scan(f, wave_tot, total_f), pi = pg_block_scan(p, wave_tot, total_p);
    const uint64_t pop = (uint64_t) f + p, total = (uint64_t) total_f + total_p;
    uint32_t pieces = 0;
    if (t < n_buckets && pop > 0 && target_pieces > 0) {
        pieces = (uint32_t) ((pop * target_pieces + total / 2) / total);
        if (pieces == 0) pieces = 1;
    }
    const uint32_t qi = pg_block_scan(pieces, wave_tot, total_q);
    if (t == b) {
        s_fb = fi - f; s_pb = pi - p; s_f = f;
        if (slice == 0) {
            base_full[b] = fi - f; base_part[b] = pi - p; piece_prefix[b] = qi - pieces;
            if (b == n_buckets - 1) { base_full[n_buckets] = total_f; base_part[n_buckets] = total_p; piece_prefix[n_buckets] = total_q; }
        }
    }
    // row b of cnt_full: exclusive prefix over the workgroups (four per thread)
    const uint32_t csum = c[0] + c[1] + c[2] + c[3], ci = pg_block_scan(csum, wave_tot, total_c);
    uint32_t run = ci - csum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t wq = 4 * t + k;
        if (wq < W) { row[wq] = run; lrow[wq] = (uint32_t) ((size_t) wq * slots) + lo[k]; }
        run += c[k];
    }
    if (t == 0) row[W] = total_c;
    const uint32_t hsum = (ent[0] != kNoPage) + (ent[1] != kNoPage) + (ent[2] != kNoPage) + (ent[3] != kNoPage);
    const uint32_t hi = pg_block_scan(hsum, wave_tot, total_h);
    __syncthreads();
    const uint32_t fb = s_fb, pb = s_pb, F = s_f;
    if (slice == 0) {
        uint32_t at = pb + hi - hsum;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (ent[k] != kNoPage) glist_part[at++] = ent[k];
    }
    const uint32_t e_begin = (uint32_t) ((uint64_t) F * slice / kPgDirSlices), e_end = (uint32_t) ((uint64_t) F * (slice + 1) / kPgDirSlices);
    if (F > 65536u) {
        // A HOT bucket (skewed indices: hundreds of thousands of pages): the list is copied run by run -- the pages of one partition
        // workgroup are contiguous in its wlist -- instead of entry by entry with a binary search over the runs per entry (8 dependent
        // LDS reads each: the directory was 0.83 ms of the 1.6 ms zipf step, profiles/bench_r06.json: also.cfg3b_zipf)
        for (uint32_t wq = (uint32_t) slice; wq < W; wq += kPgDirSlices) {
            const uint32_t r0 = row[wq], len = row[wq + 1] - r0;
            const uint32_t *from = wlist + lrow[wq];
            uint32_t *to = glist_full + fb + r0;
            for (uint32_t e0 = t; e0 < len; e0 += 8 * 256) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = e0 + u * 256 < len ? __builtin_nontemporal_load(from + e0 + u * 256) : 0u;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u * 256 < len) to[e0 + u * 256] = v[u];
            }
        }
    } else
    for (uint32_t e0 = e_begin + t; e0 < e_end; e0 += 4 * 256) {
        uint32_t src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t e = e0 + u * 256;
            uint32_t lo2 = 0, hi2 = W;                         // last w with row[w] <= e
            while (hi2 - lo2 > 1) {
                const uint32_t mid = (lo2 + hi2) / 2;
                if (row[mid] <= e) lo2 = mid; else hi2 = mid;
            }
            src[u] = e < e_end ? __builtin_nontemporal_load(wlist + lrow[lo2] + (e - row[lo2])) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + u * 256 < e_end) glist_full[fb + e0 + u * 256] = src[u];
    }
    // what the partition accumulated (elements kept, "non-finite x under a cleared mask bit") goes to where the consumers read it;
    // the accumulators and the page totals are cleared by the last workgroup of the first reducing launch (bucket_finish), after
    // which the block can serve the next object without a fill (csrc/bucketed.hip: MetaRing)
    if (b == 0 && slice == 0 && t < 3) {
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < kPgReplicas; ++r) {
            const uint32_t v = gtotal[kPgMetaBase + kPgMetaAccumRep + 4 * r + t];
            acc = t == 2 ? max(acc, v) : acc + v;            // (elements kept: a sum; the flag: any; max |x|: as bits)
        }
        gtotal[kPgMetaBase + (t == 2 ? (int) kPgMetaResultXmax : (int) kPgMetaResult + t)] = acc;
    }
    // Feedback for the next partition launch (class_w != nullptr only after a launch long enough to say something): a class's speed
    // is its share of the tiles over the mean loop duration of its workgroups; the new weight moves an eighth of the way towards
    // the share that would have made the durations equal, within [0.88, 1.12].  Equal durations are a fixed point; the class means
    // of a balanced box scatter by ~2 % from launch to launch, which at this gain leaves the weights within ~0.5 % of 1 (simulated:
    // +0.6 % on the kernel where there is nothing to balance, 1.095 -> 1.02 of the balanced time where one class is 9.5 % slow).
    if (class_w && b == 0 && slice == 1 && t < 64) {
        float dur = 0.f, cnt = 0.f;                                     // lane t: class t % 8, every eighth of its workgroups
        for (uint32_t w = (uint32_t) t; w < W; w += 64) { dur += (float) class_stamp[w]; cnt += 1.f; }
#pragma unroll
        for (int d = kPgClasses; d < 64; d <<= 1) { dur += __shfl_xor(dur, d, 64); cnt += __shfl_xor(cnt, d, 64); }
        const float old_w = t < kPgClasses ? (float) class_w[t] : (float) kPgWeightOne;
        const bool was_dealt = class_w[kPgClasses] != 0u;
        const float dealt = was_dealt ? old_w : (float) kPgWeightOne;                           // the shares the launch really used
        const float speed = (t < kPgClasses && dur > 0.f) ? dealt * cnt / dur : 0.f;
        float sum = speed, least = t < kPgClasses ? speed : 1.f;
#pragma unroll
        for (int d = 1; d < kPgClasses; d <<= 1) { sum += __shfl_xor(sum, d, 64); least = fminf(least, __shfl_xor(least, d, 64)); }
        if (t < kPgClasses && least > 0.f) {
            const float target = speed / (sum / kPgClasses) * (float) kPgWeightOne;
            const float next = fminf(fmaxf(0.875f * old_w + 0.125f * target, (float) kPgWeightMin), (float) kPgWeightMax);
            class_w[t] = (uint32_t) (next + 0.5f);
            // hysteresis: dealt from the band on, equal again below half of it
            float far = fabsf(next - (float) kPgWeightOne);
#pragma unroll
            for (int d = 1; d < kPgClasses; d <<= 1) far = fmaxf(far, __shfl_xor(far, d, 64));
            if (t == 0) class_w[kPgClasses] = (far > (float) class_band || (was_dealt && far > 0.5f * (float) class_band)) ? 1u : 0u;
        }
    }
}



// ---- host side -------------------------------------------------------------------------------------
struct PagedPlan {
    int page_shift = 6;
    uint32_t cap = 0, W = 0, slots = 0;
    size_t chunk = 0, page_slots = 0;          // page_slots: W * slots (positions = page_slots << page_shift)
    size_t lds = 0;
    bool balanced = false;                     // the launch may deal its tiles by class weights (slots are provisioned for it)
};

/// geometry of the paged partition of n elements into n_buckets buckets (4-byte values)
static inline PagedPlan paged_plan(size_t n, int n_buckets, int num_cu, bool weighted = false, bool index_only = false) {
    PagedPlan p;
    // (4-byte records: twice as many fit, 256 buckets keep two 64-element pages each)
    p.page_shift = n_buckets > (index_only ? 256 : 128) ? 5 : 6;
    int nb2 = 2;
    while (nb2 < n_buckets) nb2 <<= 1;
    p.cap = (uint32_t) ((index_only ? 2 * kPgLdsElems : kPgLdsElems) / nb2);
    const size_t tiles = (n + kPgTile - 1) / kPgTile;
    p.W = (uint32_t) std::max<size_t>(1, std::min<size_t>((size_t) num_cu, tiles));
    p.chunk = ((tiles + p.W - 1) / p.W) * kPgTile;
    p.W = (uint32_t) std::max<size_t>(1, (n + p.chunk - 1) / p.chunk);
    p.slots = (uint32_t) ((p.chunk >> p.page_shift) + (size_t) n_buckets);
    // weighted classes (k_page_partition): a workgroup of the class with the largest possible share takes up to
    // max / (7 min + max) of the tiles instead of 1 / 8 (+ a tile of rounding)
    p.balanced = weighted && p.W >= 64 && p.W % kPgClasses == 0 && tiles >= (size_t) 4 * p.W;
    if (p.balanced) {
        const double share = (double) kPgWeightMax / (7.0 * kPgWeightMin + kPgWeightMax);
        const size_t most = (size_t) ((double) tiles * share / (p.W / kPgClasses)) + 2;
        p.slots = (uint32_t) (((most * kPgTile) >> p.page_shift) + (size_t) n_buckets);
    }
    p.page_slots = (size_t) p.W * p.slots;
    p.lds = ((size_t) n_buckets * p.cap + 2) * (index_only ? 4 : 8);
    return p;
}

} // namespace ek
