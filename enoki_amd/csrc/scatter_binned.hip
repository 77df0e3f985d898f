// scatter_add through the LDS: the MI355X-native replacement for `atom.global.add` (cuda.h:892-905).
//
// Why: device-scope floating point atomics on gfx950 retire at ~21 G atomics/s no matter how the
// addresses are spread (measured: 64 Mi random adds into a 1 Mi-entry table take 3.19 ms, the same
// with one private table per XCD -- profiles/probe_r01.txt), i.e. 2 % of the HBM roofline.  A CU's
// 160 KiB LDS, on the other hand, holds 16 Ki f32 bins (64 KiB, two workgroups per CU) and executes
// ds_add_f32 at LDS speed.  So the adjoint of gather is restructured as
//
//   1. count      every workgroup owns a contiguous chunk of the n elements and histograms its
//                 indices by BUCKET (= index >> 14) in LDS                        reads  4 B/elt
//   2. scan       per-bucket exclusive scan over the workgroups' counts + scan of the bucket totals
//   3. partition  each workgroup re-reads its chunk in tiles of 8192 elements, sorts a tile by bucket
//                 in LDS (so that a bucket's elements leave the CU as one coalesced run) and appends
//                 (index within the bucket: 16 bit, value) to the bucket's pair list  reads 8, writes 6 B/elt
//   4. accumulate S workgroups per bucket stream the bucket's pairs and ds_add them into a zeroed LDS
//                 table, then write their partial table                           reads  6 B/elt
//   5. fold       target[k] += sum_s partial[s][k]                                (S + 2) * 4 B per bin
//
// = 24 B/elt of streaming traffic and no global atomics.  Tables of <= 16 Ki bins skip steps 1-3.
// The result is the same set of additions as the atomic version in a different (unspecified) order
// -- parity class D, like the reference's own GPU path.
#include "ek_unary.h"

#include <algorithm>
#include <vector>

namespace ek {

constexpr int kBinShift = 14;
constexpr int kBins = 1 << kBinShift;      // bins per bucket (64 KiB of f32 / i32 in LDS)
// 8-byte element types get half as many bins per bucket: the LDS table stays at 64 KiB (two workgroups per CU)
template <typename T> constexpr int bin_shift_of = sizeof(T) == 8 ? kBinShift - 1 : kBinShift;
template <typename T> constexpr int bins_of = 1 << bin_shift_of<T>;

// four consecutive elements: one 16-byte load for 4-byte types, two for 8-byte types
template <typename T, bool NT> __device__ __forceinline__ void load4(const T *p, T (&out)[4]) {
    if constexpr (sizeof(T) * 4 <= 16) {
        Pack<T, 4> v = pack_load<T, 4, NT>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = v.v[j];
    } else {
        Pack<T, 2> a = pack_load<T, 2, NT>(p), b = pack_load<T, 2, NT>(p + 2);
        out[0] = a.v[0]; out[1] = a.v[1]; out[2] = b.v[0]; out[3] = b.v[1];
    }
}
constexpr int kMaxBuckets = 256;
constexpr int kThreads = 512;
constexpr int kPerThread = 16;
constexpr int kTile = kThreads * kPerThread;   // elements sorted per LDS pass of the partition

template <typename I> __device__ __forceinline__ uint32_t index_u32(I i) { return (uint32_t) i; }

// Loads one tile (kTile elements) of indices / mask bits / values into registers: the lane owns kPerThread / 4 runs of 4
// consecutive elements.  A run that lies inside the input and whose arrays are 16-byte aligned is ONE vector load per
// array; the runs of a ragged last tile (and unaligned operands) are read element by element from the same addresses,
// so the two cases share their address registers.
template <typename T>
__device__ __forceinline__ void load_run4(const T *__restrict__ p, size_t e, size_t end, bool wide, T fill, T (&out)[4]) {
    if (wide) {
        load4<T, true>(p + e, out);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = e + j < end ? p[e + j] : fill;
    }
}

template <bool WithValue, bool Full = false, typename I, typename T>
__device__ __forceinline__ void load_tile(const I *__restrict__ index, const Arg<uint8_t> &mask, uint8_t sm,
                                          const Arg<T> &value, T sv, size_t base, size_t end, int vec_ok,
                                          uint32_t (&ix)[kPerThread], bool (&on)[kPerThread], T *val) {
    static_assert(kPerThread % 4 == 0 && sizeof(I) == 4);
    constexpr int kRuns = kPerThread / 4;
#pragma unroll
    for (int h = 0; h < kRuns; ++h) {
        const size_t e = base + (size_t) h * (kTile / kRuns) + (size_t) threadIdx.x * 4;
        const bool wide = Full || (vec_ok && e + 4 <= end);      // Full: a whole tile of 16-byte aligned operands
        I pi[4];
        load_run4<I>(index, e, end, wide, I(0), pi);
        uint8_t pm[4] = { sm, sm, sm, sm };
        if (mask.vec) load_run4<uint8_t>(mask.ptr, e, end, wide, uint8_t(0), pm);
        T pv[4] = { sv, sv, sv, sv };
        if constexpr (WithValue) { if (value.vec) load_run4<T>(value.ptr, e, end, wide, sv, pv); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ix[h * 4 + j] = index_u32(pi[j]);
            on[h * 4 + j] = pm[j] != 0 && (Full || e + j < end);
            if constexpr (WithValue) val[h * 4 + j] = pv[j];
        }
    }
}

// Same addressing as load_tile for one more operand array (further value streams and their weights)
template <bool Full = false, typename T>
__device__ __forceinline__ void load_tile_operand(const Arg<T> &a, T s, size_t base, size_t end, int vec_ok, T (&val)[kPerThread]) {
    constexpr int kRuns = kPerThread / 4;
    if (!a.vec) {
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) val[k] = s;
        return;
    }
#pragma unroll
    for (int h = 0; h < kRuns; ++h) {
        const size_t e = base + (size_t) h * (kTile / kRuns) + (size_t) threadIdx.x * 4;
        T pv[4];
        load_run4<T>(a.ptr, e, end, Full || (vec_ok && e + 4 <= end), s, pv);
#pragma unroll
        for (int j = 0; j < 4; ++j) val[h * 4 + j] = pv[j];
    }
}

// Value streams of one partition pass: `count` tables receive contributions through ONE index / mask array
// (the adjoints of gathers that share their index array).  Stream c scatters value[c], or -- when bit c of
// `weighted` is set -- safe_mul(weight[c], value[c]): the tape's edge product w * g fused into the read, so the
// product array is never materialised (autodiff.cpp:1191-1199 for the formula).
template <typename T, int C> struct BinStreams {
    Arg<T> value[C];
    Arg<T> weight[C];
    T *pair_val[C];
    unsigned weighted;
    int value_op[C];      // fusable unary op applied to value[c] on load (EK_COPY: none); partition kernels with Mapped = true only
};

// ---- 1. count ------------------------------------------------------------------------------------
template <typename I, int Shift = kBinShift>
__global__ __launch_bounds__(kThreads) void k_bin_count(uint32_t *__restrict__ counts, const I *__restrict__ index,
                                                        Arg<uint8_t> mask, size_t n, size_t chunk, int n_buckets,
                                                        int rep_shift, int vec_ok) {
    // Each bucket owns 2^rep_shift counters; a lane uses counter (lane mod 2^rep_shift).  With 64 buckets and
    // 64 lanes several lanes of a wave hit the same LDS address and serialise; replication spreads them.
    __shared__ uint32_t hist[kMaxBuckets];
    const uint32_t rep = threadIdx.x & ((1u << rep_shift) - 1u);
    for (int b = threadIdx.x; b < kMaxBuckets; b += kThreads) hist[b] = 0;
    __syncthreads();
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    const size_t begin = (size_t) blockIdx.x * chunk, end = begin + chunk < n ? begin + chunk : n;
    for (size_t base = begin; base < end; base += kTile) {
        uint32_t ix[kPerThread];
        bool on[kPerThread];
        load_tile<false>(index, mask, sm, Arg<uint32_t>{ nullptr, 0u, 0u }, 0u, base, end, vec_ok, ix, on, (uint32_t *) nullptr);
#pragma unroll
        for (int k = 0; k < kPerThread; ++k)
            if (on[k]) atomicAdd(&hist[((ix[k] >> Shift) << rep_shift) | rep], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n_buckets; b += kThreads) {
        uint32_t c = 0;
        for (int r = 0; r < (1 << rep_shift); ++r) c += hist[(b << rep_shift) + r];
        counts[(size_t) b * gridDim.x + blockIdx.x] = c;
    }
}

// ---- 2. scan ---------------------------------------------------------------------------------------
// counts is [n_buckets][n_blocks]; workgroup b turns row b into its exclusive prefix and emits the row total
__global__ __launch_bounds__(1024) void k_bin_scan_rows(uint32_t *__restrict__ counts, uint32_t *__restrict__ row_total,
                                                        unsigned n_blocks) {
    // 1024 entries per step: wave64 shuffle scan, then a scan of the 16 wave totals (no 20-barrier Hillis-Steele)
    __shared__ uint32_t wave_total[16];
    __shared__ uint32_t step_total;
    uint32_t *row = counts + (size_t) blockIdx.x * n_blocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (unsigned base = 0; base < n_blocks; base += 1024) {
        unsigned i = base + threadIdx.x;
        uint32_t v = i < n_blocks ? row[i] : 0u, incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        if (wave == 0) {
            uint32_t w = lane < 16 ? wave_total[lane] : 0u, wi = w;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                uint32_t up = __shfl_up(wi, d, 64);
                if (lane >= d) wi += up;
            }
            if (lane < 16) wave_total[lane] = wi - w;          // exclusive offset of every wave
            if (lane == 15) step_total = wi;
        }
        __syncthreads();
        if (i < n_blocks) row[i] = carry + wave_total[wave] + incl - v;
        carry += step_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}

// bucket_base[b] = sum of row totals of buckets < b; bucket_base[n_buckets] = grand total.
// piece_prefix[b] = number of accumulate work items ("pieces") of buckets < b.  Every bucket gets a share of the
// `target_pieces` workgroups proportional to its population (at least one when it is not empty) and is cut into
// that many equal pieces: with evenly spread indices all buckets get the same number of pieces, with skewed indices
// the crowded buckets get most of them -- the accumulate phase stays balanced either way.
// (target_pieces == 0: the caller does not need pieces.)
__global__ __launch_bounds__(256) void k_bin_scan_buckets(uint32_t *__restrict__ bucket_base, uint32_t *__restrict__ piece_prefix,
                                                          const uint32_t *__restrict__ row_total, int n_buckets,
                                                          uint32_t target_pieces) {
    __shared__ uint32_t part[256];
    uint32_t v = (int) threadIdx.x < n_buckets ? row_total[threadIdx.x] : 0u;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        uint32_t add = threadIdx.x >= (unsigned) d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    const uint32_t lo = part[threadIdx.x] - v, hi = part[threadIdx.x];
    if ((int) threadIdx.x < n_buckets) bucket_base[threadIdx.x] = lo;
    if (threadIdx.x == 255) bucket_base[n_buckets] = part[255];
    if (target_pieces == 0) return;
    __syncthreads();
    const uint64_t total = part[255], size = hi - lo;
    uint32_t pieces = 0;
    if ((int) threadIdx.x < n_buckets && size > 0) {
        pieces = (uint32_t) ((size * target_pieces + total / 2) / total);
        if (pieces == 0) pieces = 1;
    }
    __syncthreads();
    part[threadIdx.x] = pieces;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        uint32_t add = threadIdx.x >= (unsigned) d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    if ((int) threadIdx.x < n_buckets) piece_prefix[threadIdx.x] = part[threadIdx.x] - pieces;
    if (threadIdx.x == 255) piece_prefix[n_buckets] = part[255];
}

// ---- 3. partition ----------------------------------------------------------------------------------
template <typename T, typename I, int Shift = kBinShift, typename OutIdx = uint16_t, int C = 1, bool Mapped = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4))) void k_bin_partition(OutIdx *__restrict__ pair_idx, BinStreams<T, C> st,
                                                            const uint32_t *__restrict__ offsets,
                                                            const uint32_t *__restrict__ bucket_base,
                                                            const I *__restrict__ index, Arg<uint8_t> mask, size_t n,
                                                            size_t chunk, int n_buckets, int rep_shift, int vec_ok) {
    __shared__ uint32_t cursor[kMaxBuckets];       // next free global slot of this workgroup per bucket
    __shared__ uint32_t tile_hist[kMaxBuckets];    // elements of the current tile per (bucket, replica) slot
    __shared__ uint32_t tile_off[kMaxBuckets];     // exclusive prefix of tile_hist
    const uint32_t rep = threadIdx.x & ((1u << rep_shift) - 1u);
    __shared__ uint32_t stage_idx[kTile];
    __shared__ T stage_val[kTile];

    for (int b = threadIdx.x; b < kMaxBuckets; b += kThreads) {
        cursor[b] = b < n_buckets ? bucket_base[b] + offsets[(size_t) b * gridDim.x + blockIdx.x] : 0u;
        tile_hist[b] = 0;
    }
    __syncthreads();
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    T sv[C], sw[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        sv[c] = st.value[c].vec ? T(0) : arg_scalar(st.value[c]);
        sw[c] = (((st.weighted >> c) & 1u) && !st.weight[c].vec) ? arg_scalar(st.weight[c]) : T(1);
    }
    const size_t begin = (size_t) blockIdx.x * chunk, end = begin + chunk < n ? begin + chunk : n;

    // values of stream c for the current tile (times their weights).  `val` still holds the values of stream c - 1: when
    // that stream was unweighted and reads the same array (g and w * g of one gradient g -- the usual pair), the array
    // is not loaded a second time.
    auto load_stream = [&](auto full, int c, size_t base, T (&val)[kPerThread]) {
        constexpr bool Full = decltype(full)::value;
        bool reuse = c > 0 && st.value[c].vec && st.value[c].ptr == st.value[c > 0 ? c - 1 : 0].ptr &&
                     !((st.weighted >> (c > 0 ? c - 1 : 0)) & 1u);
        if constexpr (Mapped) reuse = reuse && st.value_op[c] == st.value_op[c > 0 ? c - 1 : 0];
        if (!reuse) {                      // otherwise `val` still holds the (mapped) values of stream c - 1
            load_tile_operand<Full>(st.value[c], sv[c], base, end, vec_ok, val);
            if constexpr (Mapped && std::is_floating_point_v<T>) {
                // the producer of this stream was left unevaluated (HIPArray defers fusable unary ops): apply it here
                const int op = st.value_op[c];
                if (op != EK_COPY) {
#pragma unroll
                    for (int k = 0; k < kPerThread; ++k) val[k] = unary_fused<T>(op, val[k]);
                }
            }
        }
        if ((st.weighted >> c) & 1u) {
            T w[kPerThread];
            load_tile_operand<Full>(st.weight[c], sw[c], base, end, vec_ok, w);
#pragma unroll
            for (int k = 0; k < kPerThread; ++k) val[k] = dev::safe_mul(w[k], val[k]);
        }
    };

    // one tile; `full`: the tile lies inside the input and every operand array is 16-byte aligned (no bounds checks, no
    // element-wise loads -- the ragged variant needs ~50 more registers and would spill in the common case)
    auto tile = [&](auto full, size_t base) {
        constexpr bool Full = decltype(full)::value;
        // indices + first value stream; stream c + 1 is requested while stream c is written out (measured against
        // requesting all streams up front: 5 % faster, the extra registers cost more than the early loads bring)
        uint32_t ix[kPerThread], rank[kPerThread];
        T val[kPerThread];
        uint32_t on = 0;                   // bit k: element k of this lane is active
        {
            bool flag[kPerThread];
            load_tile<false, Full>(index, mask, sm, Arg<T>{ nullptr, T(0), 0u }, T(0), base, end, vec_ok, ix, flag, (T *) nullptr);
#pragma unroll
            for (int k = 0; k < kPerThread; ++k) on |= (flag[k] ? 1u : 0u) << k;
        }
        load_stream(full, 0, base, val);
#pragma unroll
        for (int k = 0; k < kPerThread; ++k)
            rank[k] = ((on >> k) & 1u) ? atomicAdd(&tile_hist[((ix[k] >> Shift) << rep_shift) | rep], 1u) : 0u;
        __syncthreads();
        // exclusive scan of the tile histogram (256 entries) by ONE wave: 4 entries per lane + shuffle scan
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            uint32_t h0 = tile_hist[4 * l], h1 = tile_hist[4 * l + 1], h2 = tile_hist[4 * l + 2], h3 = tile_hist[4 * l + 3];
            uint32_t sum = h0 + h1 + h2 + h3, incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t up = __shfl_up(incl, d, 64);
                if (l >= d) incl += up;
            }
            uint32_t excl = incl - sum;
            tile_off[4 * l] = excl;
            tile_off[4 * l + 1] = excl + h0;
            tile_off[4 * l + 2] = excl + h0 + h1;
            tile_off[4 * l + 3] = excl + h0 + h1 + h2;
        }
        __syncthreads();
        const uint32_t tile_count = tile_off[kMaxBuckets - 1] + tile_hist[kMaxBuckets - 1];
        // bucket-sorted staging (rank becomes the position inside the sorted tile)
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            if ((on >> k) & 1u) {
                uint32_t p = tile_off[((ix[k] >> Shift) << rep_shift) | rep] + rank[k];
                rank[k] = p;
                stage_idx[p] = ix[k];
                stage_val[p] = val[k];
            }
        }
        __syncthreads();
        // coalesced runs: consecutive staged elements of one bucket go to consecutive global slots
        for (uint32_t j = threadIdx.x; j < tile_count; j += kThreads) {
            uint32_t key = stage_idx[j], b = key >> Shift;
            uint32_t g = cursor[b] + (j - tile_off[b << rep_shift]);
            pair_idx[g] = (OutIdx) (key & ((1u << Shift) - 1u));   // the bucket is implied by the position
            st.pair_val[0][g] = stage_val[j];
        }
        // further streams reuse the sorted positions: restage the values, same output addresses
#pragma unroll
        for (int c = 1; c < C; ++c) {
            load_stream(full, c, base, val);           // `val` still holds stream c - 1 (reused when both read one array)
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kPerThread; ++k)
                if ((on >> k) & 1u) stage_val[rank[k]] = val[k];
            __syncthreads();
            for (uint32_t j = threadIdx.x; j < tile_count; j += kThreads) {
                uint32_t b = stage_idx[j] >> Shift;
                st.pair_val[c][cursor[b] + (j - tile_off[b << rep_shift])] = stage_val[j];
            }
        }
        __syncthreads();
        if ((int) threadIdx.x < n_buckets) {
            const uint32_t first = threadIdx.x << rep_shift, next = (threadIdx.x + 1) << rep_shift;
            cursor[threadIdx.x] += (next < kMaxBuckets ? tile_off[next] : tile_count) - tile_off[first];
        }
        __syncthreads();
        if (threadIdx.x < kMaxBuckets) tile_hist[threadIdx.x] = 0;
        __syncthreads();
    };
    size_t base = begin;
    if (vec_ok)
        for (; base + kTile <= end; base += kTile) tile(std::true_type{}, base);
    for (; base < end; base += kTile) tile(std::false_type{}, base);
}

// ---- 4. accumulate ---------------------------------------------------------------------------------
// LDS accumulation.  Integer ds_add_u32 runs at ~10 cycles per wave instruction, but ds_add_f32 is
// microcoded on gfx950: ~194 cycles per wave instruction, conflicts or not (profiles/probe_lds_r01.txt),
// which would cap 64 Mi float adds at 0.33 ms.  Floats therefore take a per-bin EXCHANGE LOCK built
// from the fast integer path:
//     old = ds_wrxchg_rtn_b32(bin, LOCKED)      claim the bin (LOCKED = a NaN payload we never store)
//     if (old == LOCKED) retry                   someone else holds it for the next few cycles
//     ds_write_b32(bin, old + v)                 plain store releases it
// A wave executes these in lockstep, so of the lanes that collide on one bin exactly one wins per
// iteration and the holder never waits for a spinner -> always progresses.  With random bins almost
// every lane succeeds on the first try: ~2 LDS instructions per element instead of a 194-cycle atomic.
constexpr uint32_t kLockedBits = 0xFFC00001u;
constexpr unsigned long long kLockedBits64 = 0xFFF8000000000001ull;

template <bool UseLock, typename T> __device__ __forceinline__ void lds_add(T *addr, T v, bool active) {
    if constexpr (std::is_same_v<T, float>) {
        if constexpr (UseLock) {
            // The loop condition is WAVE-UNIFORM (__any): every lane stays inside until the whole wave is
            // done, so a winner's releasing store is issued in the iteration in which it won.  (With a
            // per-lane `while (pending)` the compiler may sink the store behind the loop exit, where the
            // winner waits for reconvergence with the very lanes that spin on its lock -- a deadlock.)
            unsigned *p = reinterpret_cast<unsigned *>(addr);
            bool pending = active;
            // (1) optimistic round: with well-spread bins nearly every lane wins here
            if (pending) {
                unsigned old = atomicExch(p, kLockedBits);
                if (old != kLockedBits) {
                    float sum = __uint_as_float(old) + v;
                    unsigned bits = __float_as_uint(sum);
                    if (bits == kLockedBits) bits = 0x7FC00000u;         // never publish the lock pattern
                    __hip_atomic_store(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pending = false;
                }
            }
            // (2) losers collided inside the wave (or met another wave's lock).  Skewed index distributions would
            // serialise here lane by lane, so the lanes that share the first loser's bin first combine their values
            // with a wave reduction and ONE lane adds the total: the number of rounds is the number of distinct
            // contended bins, not the number of colliding lanes.  Only that one lane ever spins, and never on a
            // lock held inside its own wave, so it always gets through.
            const unsigned key = (unsigned) (uintptr_t) addr;
            const int lane = threadIdx.x & 63;
            while (__any(pending)) {
                const unsigned long long pend = __ballot(pending);
                const int leader = __ffsll((long long) pend) - 1;
                const unsigned leader_key = __shfl(key, leader);
                const bool grouped = pending && key == leader_key;
                float total = grouped ? v : 0.0f;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d);
                if (lane == leader) {
                    unsigned old;
                    do { old = atomicExch(p, kLockedBits); } while (old == kLockedBits);
                    unsigned bits = __float_as_uint(__uint_as_float(old) + total);
                    if (bits == kLockedBits) bits = 0x7FC00000u;
                    __hip_atomic_store(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                pending = pending && !grouped;
            }
        } else {
            if (active) atomicAdd(addr, v);                                          // ds_add_f32
        }
    } else if constexpr (std::is_same_v<T, double>) {
        // the same exchange lock on 64-bit bins (ds_wrxchg_rtn_b64); tiny tables use ds_add_f64 directly
        if constexpr (UseLock) {
            unsigned long long *p = reinterpret_cast<unsigned long long *>(addr);
            bool pending = active;
            if (pending) {
                unsigned long long old = atomicExch(p, kLockedBits64);
                if (old != kLockedBits64) {
                    unsigned long long bits = (unsigned long long) __double_as_longlong(__longlong_as_double((long long) old) + v);
                    if (bits == kLockedBits64) bits = 0x7FF8000000000000ull;
                    __hip_atomic_store(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pending = false;
                }
            }
            const unsigned key = (unsigned) (uintptr_t) addr;
            const int lane = threadIdx.x & 63;
            while (__any(pending)) {
                const unsigned long long pend = __ballot(pending);
                const int leader = __ffsll((long long) pend) - 1;
                const unsigned leader_key = __shfl(key, leader);
                const bool grouped = pending && key == leader_key;
                double total = grouped ? v : 0.0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d);
                if (lane == leader) {
                    unsigned long long old;
                    do { old = atomicExch(p, kLockedBits64); } while (old == kLockedBits64);
                    unsigned long long bits = (unsigned long long) __double_as_longlong(__longlong_as_double((long long) old) + total);
                    if (bits == kLockedBits64) bits = 0x7FF8000000000000ull;
                    __hip_atomic_store(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                pending = pending && !grouped;
            }
        } else {
            if (active) atomicAdd(addr, v);                                          // ds_add_f64
        }
    } else if constexpr (sizeof(T) == 8) {
        if (active) atomicAdd(reinterpret_cast<unsigned long long *>(addr), (unsigned long long) v);   // ds_add_u64
    } else {
        if (active) atomicAdd(reinterpret_cast<unsigned int *>(addr), (unsigned int) v);   // ds_add_u32
    }
}

// Pairs come either from the partition (Direct = false: bucket b owns [bucket_base[b], bucket_base[b+1]))
// or straight from the operands when the whole table fits one bucket (Direct = true).
template <typename T, typename I, bool Direct, bool UseLock>
__global__ __launch_bounds__(kThreads) void k_bin_accumulate(T *__restrict__ partials, size_t table_size,
                                                             const uint16_t *__restrict__ pair_idx,
                                                             const T *__restrict__ pair_val,
                                                             const uint32_t *__restrict__ bucket_base, Arg<T> value,
                                                             const I *__restrict__ index, Arg<uint8_t> mask, size_t n,
                                                             int slices, const uint32_t *__restrict__ piece_prefix) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T *acc = reinterpret_cast<T *>(lds_raw);
    constexpr int Bins = bins_of<T>;
    size_t begin, end;
    if constexpr (Direct) {
        const size_t per = ((n + slices - 1) / slices + 4095) / 4096 * 4096;     // multiple of the vector step
        begin = (size_t) blockIdx.x * per < n ? (size_t) blockIdx.x * per : n;
        end = begin + per < n ? begin + per : n;
    } else {
        // work item = piece number blockIdx.x (the grid is an upper bound on the number of pieces): find its bucket
        // (piece_prefix is ascending, <= 257 entries; `slices` carries n_buckets here), then its range: the q-th of the
        // bucket's equal pieces
        __shared__ int s_bucket;
        const int n_buckets = slices;
        if (blockIdx.x >= piece_prefix[n_buckets]) return;
        for (int b = threadIdx.x; b < n_buckets; b += kThreads)
            if (piece_prefix[b] <= blockIdx.x && blockIdx.x < piece_prefix[b + 1]) s_bucket = b;
        __syncthreads();
        const int bucket = s_bucket;
        // value stream blockIdx.y: its pair values follow those of the previous stream (`n` = pairs per stream), its
        // partial tables likewise
        pair_val += (size_t) blockIdx.y * n;
        partials += (size_t) blockIdx.y * gridDim.x * Bins;
        const size_t lo = bucket_base[bucket], hi = bucket_base[bucket + 1], q = blockIdx.x - piece_prefix[bucket];
        const size_t pieces = piece_prefix[bucket + 1] - piece_prefix[bucket], per = (hi - lo + pieces - 1) / pieces;
        begin = lo + q * per < hi ? lo + q * per : hi;
        end = begin + per < hi ? begin + per : hi;
    }
    for (int j = threadIdx.x; j < Bins; j += kThreads) acc[j] = T(0);
    __syncthreads();

    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    const T sv = value.vec ? T(0) : arg_scalar(value);
    constexpr int kAcc = 8;      // loads in flight per lane
    const bool plain = Direct && !mask.vec && sm != 0 && value.vec;
    if constexpr (Direct) {
        // fast path of the single-bucket case: no mask array, value array, 16-byte aligned operands -> every lane
        // moves two 16-byte vectors of indices and of values per step (begin is a multiple of the step)
        const bool aligned = ((reinterpret_cast<uintptr_t>(index) | reinterpret_cast<uintptr_t>(value.ptr)) & 15u) == 0;
        if (plain && aligned) {
            constexpr size_t kStep = (size_t) kAcc * kThreads;
            size_t base = begin;
            for (; base + kStep <= end; base += kStep) {
                Pack<I, 4> pi[2];
                T pv[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const size_t e = base + (size_t) h * (kStep / 2) + (size_t) threadIdx.x * 4;
                    pi[h] = pack_load<I, 4, true>(index + e);
                    load4<T, true>(value.ptr + e, pv[h]);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        lds_add<UseLock>(&acc[index_u32(pi[h].v[j]) & (Bins - 1)], pv[h][j], true);
            }
            begin = base;          // the generic loop below finishes the tail
        }
    } else {
        // pair lists: a piece starts anywhere; up to 3 leading pairs go one per lane, then every lane moves two
        // 4-element vectors (8 bytes of bucket-local indices, 16 bytes of values) per step
        const size_t head_end = ((begin + 3) & ~(size_t) 3) < end ? ((begin + 3) & ~(size_t) 3) : end;
        {
            const size_t i = begin + threadIdx.x;
            const bool on = i < head_end;
            const uint32_t ix = on ? (uint32_t) pair_idx[i] : 0u;
            const T v = on ? pair_val[i] : T(0);
            lds_add<UseLock>(&acc[ix & (Bins - 1)], v, on);
        }
        // software pipelined: the loads of step i + 1 are issued before the LDS adds of step i, so that every wave always
        // has 48 B per lane in flight (without it a wave alternates between waiting for memory and for the LDS and
        // the phase stops at ~3.5 TB/s)
        constexpr size_t kStep = (size_t) kAcc * kThreads;
        size_t base = head_end;
        struct Step { Pack<uint16_t, 4> pi[2]; T pv[2][4]; };
        auto fetch = [&](Step &s, size_t at) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const size_t e = at + (size_t) h * (kStep / 2) + (size_t) threadIdx.x * 4;
                s.pi[h] = pack_load<uint16_t, 4, true>(pair_idx + e);
                load4<T, true>(pair_val + e, s.pv[h]);
            }
        };
        auto apply = [&](const Step &s) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    lds_add<UseLock>(&acc[(uint32_t) s.pi[h].v[j] & (Bins - 1)], s.pv[h][j], true);
        };
        if (base + kStep <= end) {
            Step cur, next;
            fetch(cur, base);
            for (; base + 2 * kStep <= end; base += kStep) {
                fetch(next, base + kStep);
                apply(cur);
                cur = next;
            }
            apply(cur);
            base += kStep;
        }
        begin = base;
    }
    for (size_t base = begin; base < end; base += (size_t) kAcc * kThreads) {
        uint32_t ix[kAcc];
        T val[kAcc];
        bool on[kAcc];
#pragma unroll
        for (int k = 0; k < kAcc; ++k) {
            size_t i = base + (size_t) k * kThreads + threadIdx.x;
            on[k] = i < end;
            if constexpr (Direct) {
                if (plain) {          // no mask array, value array: the common case without per-element operand tests
                    ix[k] = i < end ? index_u32(__builtin_nontemporal_load(index + i)) : 0u;
                    val[k] = i < end ? __builtin_nontemporal_load(value.ptr + i) : T(0);
                } else {
                    on[k] = on[k] && (mask.vec ? mask.ptr[i] : sm);
                    ix[k] = i < end ? index_u32(__builtin_nontemporal_load(index + i)) : 0u;
                    val[k] = (value.vec && i < end) ? __builtin_nontemporal_load(value.ptr + i) : sv;
                }
            } else {
                ix[k] = i < end ? (uint32_t) __builtin_nontemporal_load(pair_idx + i) : 0u;
                val[k] = i < end ? __builtin_nontemporal_load(pair_val + i) : T(0);
            }
        }
#pragma unroll
        for (int k = 0; k < kAcc; ++k)
            lds_add<UseLock>(&acc[ix[k] & (Bins - 1)], val[k], on[k]);
    }
    __syncthreads();

    // Direct: one table-sized partial per slice; binned: one bucket-sized partial per piece
    T *out = Direct ? partials + (size_t) blockIdx.x * table_size : partials + (size_t) blockIdx.x * Bins;
    const size_t valid = Direct ? table_size : (size_t) Bins;
    for (int j = threadIdx.x; j < Bins; j += kThreads)
        if ((size_t) j < valid) out[j] = acc[j];
}

// ---- 5. fold ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_bin_fold(T *__restrict__ target, const T *__restrict__ partials, size_t table_size,
                                                  int slices) {
    size_t k = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (k >= table_size) return;
    using U = wrap_t<T>;
    T s = target[k];
    for (int j = 0; j < slices; ++j) s = (T) ((U) s + (U) partials[(size_t) j * table_size + k]);
    target[k] = s;
}

// first stage of a two-stage fold for small tables with many slices (a 16 Ki-bin table has only 64 workgroups
// worth of bins): group g sums the slices s = g, g + groups, g + 2 groups, ... into out[g][k]
template <typename T>
__global__ __launch_bounds__(256) void k_bin_fold_groups(T *__restrict__ out, const T *__restrict__ partials, size_t table_size,
                                                         int slices, int groups) {
    size_t k = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (k >= table_size) return;
    using U = wrap_t<T>;
    const int g = blockIdx.y;
    T s = T(0);
    for (int j = g; j < slices; j += groups) s = (T) ((U) s + (U) partials[(size_t) j * table_size + k]);
    out[(size_t) g * table_size + k] = s;
}

// binned path: bin k of bucket b sums the partials of the bucket's pieces; blockIdx.y = value stream (table)
template <typename T, int C> struct FoldTargets { T *table[C]; };

template <typename T, int C>
__global__ __launch_bounds__(256) void k_bin_fold_pieces(FoldTargets<T, C> targets, const T *__restrict__ partials,
                                                         const uint32_t *__restrict__ piece_prefix, size_t table_size,
                                                         size_t partial_stride) {
    size_t k = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (k >= table_size) return;
    using U = wrap_t<T>;
    T *__restrict__ target = targets.table[blockIdx.y];
    partials += (size_t) blockIdx.y * partial_stride;
    const uint32_t b = (uint32_t) (k >> bin_shift_of<T>), local = (uint32_t) (k & (bins_of<T> - 1));
    T s = target[k];
    for (uint32_t p = piece_prefix[b]; p < piece_prefix[b + 1]; ++p)
        s = (T) ((U) s + (U) partials[(size_t) p * bins_of<T> + local]);
    target[k] = s;
}

struct Scratch {
    void *ptr = nullptr;
    ~Scratch() { if (ptr) ek_hip_free(ptr); }      // stream-ordered: safe to hand back right after enqueueing
    int alloc(size_t bytes) { return ek_hip_malloc(bytes, &ptr); }
};

template <typename T, typename I>
int scatter_add_binned_large(T *base, size_t table_size, const Arg<T> &value, const Arg<I> &index, const Arg<uint8_t> &mask,
                             size_t n);
template <typename T, typename I, int C>
int scatter_add_binned_multi(T *const *bases, size_t table_size, const Arg<T> *values, const Arg<T> *weights, unsigned weighted,
                             const Arg<I> &index, const Arg<uint8_t> &mask, size_t n, const int *value_ops);

template <typename T, typename I>
int scatter_add_binned(T *base, size_t table_size, const Arg<T> &value, const Arg<I> &index, const Arg<uint8_t> &mask,
                       size_t n) {
    constexpr int Bins = bins_of<T>;
    if (table_size > (size_t) kMaxBuckets * Bins)
        return scatter_add_binned_large<T, I>(base, table_size, value, index, mask, n);
    Context &c = ctx();
    const int n_buckets = (int) ((table_size + Bins - 1) / Bins);
    const size_t lds_bytes = (size_t) Bins * sizeof(T);
    const size_t algo_bytes = arg_bytes(value, n) + arg_bytes(index, n) + arg_bytes(mask, n);

    if (n_buckets == 1) {
        int slices = std::max(1, std::min(2 * c.num_cu, (int) (n / 65536)));
        Scratch partials;
        if (int rc = partials.alloc((size_t) slices * table_size * sizeof(T))) return rc;
        // tiny tables: many lanes of a wave collide on one bin, where the (conflict-insensitive) ds_add_f32
        // beats the exchange lock; from ~1 Ki bins on collisions inside a wave are rare
        if (table_size > 1024)
            hipLaunchKernelGGL((k_bin_accumulate<T, I, true, true>), dim3(slices), dim3(kThreads), lds_bytes, c.stream,
                               (T *) partials.ptr, table_size, nullptr, nullptr, nullptr, value, index.ptr, mask, n, slices,
                               nullptr);
        else
            hipLaunchKernelGGL((k_bin_accumulate<T, I, true, false>), dim3(slices), dim3(kThreads), lds_bytes, c.stream,
                               (T *) partials.ptr, table_size, nullptr, nullptr, nullptr, value, index.ptr, mask, n, slices,
                               nullptr);
        EK_LAUNCH_CHECK("scatter_add_lds", n, algo_bytes);
        const unsigned bin_blocks = (unsigned) ((table_size + 255) / 256);
        if (slices > 32) {
            // few bins, many slices: fold in two stages so that the first one has slices/2 x more workgroups
            const int groups = 16;
            Scratch grouped;
            if (int rc = grouped.alloc((size_t) groups * table_size * sizeof(T))) return rc;
            hipLaunchKernelGGL((k_bin_fold_groups<T>), dim3(bin_blocks, groups), dim3(256), 0, c.stream, (T *) grouped.ptr,
                               (const T *) partials.ptr, table_size, slices, groups);
            hipLaunchKernelGGL((k_bin_fold<T>), dim3(bin_blocks), dim3(256), 0, c.stream, base, (const T *) grouped.ptr,
                               table_size, groups);
        } else {
            hipLaunchKernelGGL((k_bin_fold<T>), dim3(bin_blocks), dim3(256), 0, c.stream, base, (const T *) partials.ptr,
                               table_size, slices);
        }
        EK_LAUNCH_CHECK("scatter_add_fold", table_size, (size_t) (slices + 2) * table_size * sizeof(T));
        return EK_OK;
    }

    const Arg<T> values[1] = { value }, weights[1] = { Arg<T>{ nullptr, T(1), 0u } };
    T *bases[1] = { base };
    return scatter_add_binned_multi<T, I, 1>(bases, table_size, values, weights, 0u, index, mask, n);
}

// `C` value streams through one index / mask array into `C` tables of the same size (2 .. 256 buckets): one count, one
// scan, one partition pass that reads the indices once; accumulate + fold per stream.
template <typename T, typename I, int C>
int scatter_add_binned_multi(T *const *bases, size_t table_size, const Arg<T> *values, const Arg<T> *weights, unsigned weighted,
                             const Arg<I> &index, const Arg<uint8_t> &mask, size_t n, const int *value_ops) {
    RoctxRange range("enoki-hip: scatter_add (LDS-binned)");
    Context &c = ctx();
    constexpr int Bins = bins_of<T>, Shift = bin_shift_of<T>;
    const int n_buckets = (int) ((table_size + Bins - 1) / Bins);
    const size_t lds_bytes = (size_t) Bins * sizeof(T);
    if (n_buckets < 2 || n_buckets > kMaxBuckets) return fail(EK_ERR_INVALID, "scatter_add_binned_multi(): table size out of range");

    // chunked passes over the input: a few workgroups per CU, chunks are multiples of the tile
    unsigned blocks = (unsigned) std::min<size_t>((size_t) c.num_cu * 4, (n + kTile - 1) / kTile);
    if (blocks == 0) blocks = 1;
    size_t chunk = (n + blocks - 1) / blocks;
    chunk = (chunk + kTile - 1) / kTile * kTile;
    blocks = (unsigned) ((n + chunk - 1) / chunk);

    int vec_ok = arg_aligned(index) && arg_aligned(mask);
    size_t stream_bytes = 0;
    BinStreams<T, C> st;
    st.weighted = weighted;
    bool mapped = false;
    for (int s = 0; s < C; ++s) {
        st.value[s] = values[s];
        st.weight[s] = weights[s];
        st.value_op[s] = value_ops ? value_ops[s] : (int) EK_COPY;
        mapped = mapped || st.value_op[s] != EK_COPY;
        vec_ok = vec_ok && arg_aligned(values[s]) && (!((weighted >> s) & 1u) || arg_aligned(weights[s]));
        const bool reused = s > 0 && values[s].vec && values[s].ptr == values[s - 1].ptr && !((weighted >> (s - 1)) & 1u) &&
                            st.value_op[s] == st.value_op[s - 1];
        stream_bytes += (reused ? 0 : arg_bytes(values[s], n)) + (((weighted >> s) & 1u) ? arg_bytes(weights[s], n) : 0);
    }
    int rep_shift = 0;
    while ((n_buckets << (rep_shift + 1)) <= kMaxBuckets && rep_shift < 4) ++rep_shift;
    const size_t count_entries = (size_t) n_buckets * blocks;
    Scratch counts, pairs_idx, pairs_val, partials;
    // layout: counts[n_buckets][blocks] | row_total[kMaxBuckets] | bucket_base[kMaxBuckets + 1] | piece_prefix[kMaxBuckets + 1]
    if (int rc = counts.alloc((count_entries + 3 * kMaxBuckets + 2) * sizeof(uint32_t))) return rc;
    if (int rc = pairs_idx.alloc(n * sizeof(uint16_t))) return rc;
    if (int rc = pairs_val.alloc((size_t) C * n * sizeof(T))) return rc;        // stream s at offset s * n
    for (int s = 0; s < C; ++s) st.pair_val[s] = (T *) pairs_val.ptr + (size_t) s * n;
    uint32_t *row_total = (uint32_t *) counts.ptr + count_entries;
    uint32_t *bucket_base = row_total + kMaxBuckets;
    uint32_t *piece_prefix = bucket_base + kMaxBuckets + 1;
    // accumulate work items: about two workgroups per CU (64 KiB of LDS each), shared out by bucket population
    // ... but not more pieces than the input can feed: every piece costs a zeroed 64 KiB LDS table and a 64 KiB partial
    // that the fold reads back, which for small inputs (the 8 Mi-element shards of an 8-GPU run) is more traffic than
    // the pairs themselves
    static const size_t piece_elems = [] { const char *e = getenv("ENOKI_HIP_PIECE_ELEMS"); return e ? (size_t) atol(e) : (size_t) 32768; }();
    const uint32_t target_pieces = (uint32_t) std::max<size_t>(std::min<size_t>(2 * (size_t) c.num_cu, n / piece_elems), (size_t) n_buckets);
    const unsigned max_pieces = target_pieces + (unsigned) n_buckets;       // rounding + "at least one" slack

    hipLaunchKernelGGL((k_bin_count<I, Shift>), dim3(blocks), dim3(kThreads), 0, c.stream, (uint32_t *) counts.ptr, index.ptr,
                       mask, n, chunk, n_buckets, rep_shift, vec_ok);
    EK_LAUNCH_CHECK("scatter_add_count", n, arg_bytes(index, n) + arg_bytes(mask, n));
    hipLaunchKernelGGL(k_bin_scan_rows, dim3(n_buckets), dim3(1024), 0, c.stream, (uint32_t *) counts.ptr, row_total, blocks);
    hipLaunchKernelGGL(k_bin_scan_buckets, dim3(1), dim3(256), 0, c.stream, bucket_base, piece_prefix,
                       (const uint32_t *) row_total, n_buckets, target_pieces);
    EK_LAUNCH_CHECK("scatter_add_scan", count_entries, 2 * count_entries * sizeof(uint32_t));
    if constexpr (std::is_floating_point_v<T>) {
        if (mapped)
            hipLaunchKernelGGL((k_bin_partition<T, I, Shift, uint16_t, C, true>), dim3(blocks), dim3(kThreads), 0, c.stream,
                               (uint16_t *) pairs_idx.ptr, st, (const uint32_t *) counts.ptr, (const uint32_t *) bucket_base,
                               index.ptr, mask, n, chunk, n_buckets, 0, vec_ok);
    } else if (mapped) {
        return fail(EK_ERR_UNSUPPORTED, "scatter_add_binned_multi(): mapped value streams need a floating point type");
    }
    if (!mapped)
        hipLaunchKernelGGL((k_bin_partition<T, I, Shift, uint16_t, C>), dim3(blocks), dim3(kThreads), 0, c.stream,
                           (uint16_t *) pairs_idx.ptr, st, (const uint32_t *) counts.ptr, (const uint32_t *) bucket_base,
                           index.ptr, mask, n, chunk, n_buckets, 0, vec_ok);
    EK_LAUNCH_CHECK("scatter_add_partition", n, stream_bytes + arg_bytes(index, n) + arg_bytes(mask, n) +
                                                n * (sizeof(uint16_t) + C * sizeof(T)));

    // ONE accumulate launch and ONE fold launch for all value streams (grid.y = stream): fewer launches per backward()
    if (int rc = partials.alloc((size_t) C * max_pieces * Bins * sizeof(T))) return rc;
    hipLaunchKernelGGL((k_bin_accumulate<T, I, false, true>), dim3(max_pieces, C), dim3(kThreads), lds_bytes, c.stream,
                       (T *) partials.ptr, table_size, (const uint16_t *) pairs_idx.ptr, (const T *) pairs_val.ptr,
                       (const uint32_t *) bucket_base, values[0], index.ptr, mask, n, n_buckets, (const uint32_t *) piece_prefix);
    EK_LAUNCH_CHECK("scatter_add_accumulate", (size_t) C * n,
                    (size_t) C * (n * (sizeof(uint16_t) + sizeof(T)) + (size_t) max_pieces * Bins * sizeof(T)));
    FoldTargets<T, C> targets;
    for (int s = 0; s < C; ++s) targets.table[s] = bases[s];
    hipLaunchKernelGGL((k_bin_fold_pieces<T, C>), dim3((unsigned) ((table_size + 255) / 256), C), dim3(256), 0, c.stream, targets,
                       (const T *) partials.ptr, (const uint32_t *) piece_prefix, table_size, (size_t) max_pieces * Bins);
    EK_LAUNCH_CHECK("scatter_add_fold", (size_t) C * table_size,
                    (size_t) C * ((size_t) max_pieces * Bins * sizeof(T) + 2 * table_size * sizeof(T)));
    return EK_OK;
}

// ---- tables beyond 256 buckets (4 Mi bins): split by SUPER-bucket first ---------------------------------
// One more count / scan / partition pass with shift 22 groups the (index, value) pairs by 4 Mi-bin slice of the
// table (<= 256 slices: tables up to 2^30 bins) and rewrites the indices relative to their slice; every populated
// slice is then an ordinary binned scatter_add on `base + slice * 4 Mi`.  The slice populations are read back once
// (the only synchronisation).  Versus the global-atomic fallback this is ~5x faster on uniform indices and does not
// collapse on skewed ones (same-address device atomics retire at 0.08 G/s).
template <typename T> constexpr int super_shift_of = bin_shift_of<T> + 8;       // 2^22 (2^21 for 8-byte types) bins per super-bucket

template <typename T, int N> __global__ __launch_bounds__(256) void k_scatter_add_pairs(T *__restrict__ base, const T *__restrict__ val,
                                                                                       const uint32_t *__restrict__ idx, size_t n) {
    // small remainder slices: plain device atomics
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    using U = wrap_t<T>;
    if constexpr (std::is_floating_point_v<T>) unsafeAtomicAdd(base + idx[i], val[i]);
    else atomicAdd(reinterpret_cast<U *>(base) + idx[i], (U) val[i]);
}

template <typename T, typename I>
int scatter_add_binned_large(T *base, size_t table_size, const Arg<T> &value, const Arg<I> &index, const Arg<uint8_t> &mask,
                             size_t n) {
    Context &c = ctx();
    constexpr int kSuperShift = super_shift_of<T>;
    constexpr size_t kSuperBins = (size_t) 1 << kSuperShift;
    const int n_super = (int) ((table_size + kSuperBins - 1) / kSuperBins);
    unsigned blocks = (unsigned) std::min<size_t>((size_t) c.num_cu * 4, (n + kTile - 1) / kTile);
    if (blocks == 0) blocks = 1;
    size_t chunk = (n + blocks - 1) / blocks;
    chunk = (chunk + kTile - 1) / kTile * kTile;
    blocks = (unsigned) ((n + chunk - 1) / chunk);
    const int vec_ok = arg_aligned(index) && arg_aligned(mask) && arg_aligned(value);
    int rep_shift = 0;
    while ((n_super << (rep_shift + 1)) <= kMaxBuckets && rep_shift < 4) ++rep_shift;
    const size_t count_entries = (size_t) n_super * blocks;
    Scratch counts, pairs_idx, pairs_val;
    if (int rc = counts.alloc((count_entries + 2 * kMaxBuckets + 1) * sizeof(uint32_t))) return rc;
    if (int rc = pairs_idx.alloc(n * sizeof(uint32_t))) return rc;
    if (int rc = pairs_val.alloc(n * sizeof(T))) return rc;
    uint32_t *row_total = (uint32_t *) counts.ptr + count_entries;
    uint32_t *bucket_base = row_total + kMaxBuckets;

    hipLaunchKernelGGL((k_bin_count<I, kSuperShift>), dim3(blocks), dim3(kThreads), 0, c.stream, (uint32_t *) counts.ptr,
                       index.ptr, mask, n, chunk, n_super, rep_shift, vec_ok);
    EK_LAUNCH_CHECK("scatter_add_count", n, arg_bytes(index, n) + arg_bytes(mask, n));
    hipLaunchKernelGGL(k_bin_scan_rows, dim3(n_super), dim3(1024), 0, c.stream, (uint32_t *) counts.ptr, row_total, blocks);
    hipLaunchKernelGGL(k_bin_scan_buckets, dim3(1), dim3(256), 0, c.stream, bucket_base, (uint32_t *) nullptr,
                       (const uint32_t *) row_total, n_super, 0u);
    EK_LAUNCH_CHECK("scatter_add_scan", count_entries, 2 * count_entries * sizeof(uint32_t));
    BinStreams<T, 1> st;
    st.value[0] = value;
    st.weight[0] = Arg<T>{ nullptr, T(1), 0u };
    st.pair_val[0] = (T *) pairs_val.ptr;
    st.weighted = 0u;
    hipLaunchKernelGGL((k_bin_partition<T, I, kSuperShift, uint32_t, 1>), dim3(blocks), dim3(kThreads), 0, c.stream,
                       (uint32_t *) pairs_idx.ptr, st, (const uint32_t *) counts.ptr, (const uint32_t *) bucket_base, index.ptr,
                       mask, n, chunk, n_super, 0, vec_ok);
    EK_LAUNCH_CHECK("scatter_add_partition", n, arg_bytes(value, n) + arg_bytes(index, n) + arg_bytes(mask, n) +
                                                n * (sizeof(uint32_t) + sizeof(T)));

    std::vector<uint32_t> offsets((size_t) n_super + 1);
    if (int busy = refuse_while_capturing("scatter_add into a table of more than 4 Mi bins (slice populations are read back)")) return busy;
    EK_HIP_CHECK(hipMemcpyAsync(offsets.data(), bucket_base, offsets.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
    EK_HIP_CHECK(hipStreamSynchronize(c.stream));

    for (int sb = 0; sb < n_super; ++sb) {
        const size_t lo = offsets[(size_t) sb], cnt = offsets[(size_t) sb + 1] - lo;
        if (cnt == 0) continue;
        T *sub_base = base + (size_t) sb * kSuperBins;
        const size_t sub_size = std::min(kSuperBins, table_size - (size_t) sb * kSuperBins);
        const T *sub_val = (const T *) pairs_val.ptr + lo;
        const uint32_t *sub_idx = (const uint32_t *) pairs_idx.ptr + lo;
        if (cnt >= ((size_t) 1 << 18)) {
            Arg<T> v{ sub_val, T(0), 1u };
            Arg<uint32_t> ix{ sub_idx, 0u, 1u };
            Arg<uint8_t> all_on{ nullptr, 1, 0u };
            if (int rc = scatter_add_binned<T, uint32_t>(sub_base, sub_size, v, ix, all_on, cnt)) return rc;
        } else {
            hipLaunchKernelGGL((k_scatter_add_pairs<T, 1>), dim3((unsigned) ((cnt + 255) / 256)), dim3(256), 0, c.stream, sub_base,
                               sub_val, sub_idx, cnt);
            EK_LAUNCH_CHECK("scatter_add", cnt, cnt * (sizeof(uint32_t) + sizeof(T)));
        }
    }
    return EK_OK;
}

// entry points used by ek_hip_scatter_add (memory.hip)
bool scatter_add_binned_applicable(size_t table_size, size_t n, bool index_is_array, size_t elem_size) {
    const int shift = elem_size == 8 ? kBinShift - 1 : kBinShift;
    return index_is_array && table_size > 0 && n >= ((size_t) 1 << 18) &&
           table_size <= ((size_t) kMaxBuckets << (shift + 8)) && n < ((size_t) 1 << 32);
}

bool scatter_add_binned_multi_applicable(size_t table_size, size_t n, bool index_is_array, size_t elem_size) {
    const int shift = elem_size == 8 ? kBinShift - 1 : kBinShift;
    return scatter_add_binned_applicable(table_size, n, index_is_array, elem_size) && table_size > ((size_t) 1 << shift) &&
           table_size <= ((size_t) kMaxBuckets << shift);
}

#define EK_BINNED_INSTANCE(T, I)                                                                                      \
    template int scatter_add_binned<T, I>(T *, size_t, const Arg<T> &, const Arg<I> &, const Arg<uint8_t> &, size_t);  \
    template int scatter_add_binned_multi<T, I, 1>(T *const *, size_t, const Arg<T> *, const Arg<T> *, unsigned,       \
                                                   const Arg<I> &, const Arg<uint8_t> &, size_t, const int *);        \
    template int scatter_add_binned_multi<T, I, 2>(T *const *, size_t, const Arg<T> *, const Arg<T> *, unsigned,       \
                                                   const Arg<I> &, const Arg<uint8_t> &, size_t, const int *);        \
    template int scatter_add_binned_multi<T, I, 3>(T *const *, size_t, const Arg<T> *, const Arg<T> *, unsigned,       \
                                                   const Arg<I> &, const Arg<uint8_t> &, size_t, const int *);
EK_BINNED_INSTANCE(float, uint32_t) EK_BINNED_INSTANCE(float, int32_t)
EK_BINNED_INSTANCE(uint32_t, uint32_t) EK_BINNED_INSTANCE(uint32_t, int32_t)
EK_BINNED_INSTANCE(double, uint32_t) EK_BINNED_INSTANCE(double, int32_t)
EK_BINNED_INSTANCE(uint64_t, uint32_t) EK_BINNED_INSTANCE(uint64_t, int32_t)

} // namespace ek

// =================================================================================================
//  Deterministic scatter_add (mode 1): bit-identical to the CPU reference's element-order accumulation
//  (dynamic.h:517-534 -> sequential transform, array_static.h:982-991).
//
//  A STABLE least-significant-digit radix sort of the (index, value) pairs by index (8 bits per pass,
//  ceil(log2(table) / 8) passes) leaves every bin's contributions contiguous AND in element order; one
//  lane per bin then adds its run sequentially, starting from the bin's current value -- the exact
//  sequence of fp additions the CPU performs.  Stability comes from ranking with wave64 ballots instead
//  of atomics: a tile is laid out so that (wave, item, lane) order is element order, lanes holding the
//  same digit find each other with 8 ballots ("match any"), and per-wave digit counters in LDS are only
//  ever touched by their own wave.
// =================================================================================================
namespace ek {

constexpr int kRadixBits = 8, kRadix = 1 << kRadixBits;
constexpr int kSortWaves = kThreads / 64;

template <typename I>
__global__ __launch_bounds__(kThreads) void k_radix_count(uint32_t *__restrict__ counts, const I *__restrict__ keys,
                                                          Arg<uint8_t> mask, size_t n, size_t chunk, int shift) {
    __shared__ uint32_t hist[kRadix];
    for (int b = threadIdx.x; b < kRadix; b += kThreads) hist[b] = 0;
    __syncthreads();
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    const size_t begin = (size_t) blockIdx.x * chunk, end = begin + chunk < n ? begin + chunk : n;
    for (size_t i = begin + threadIdx.x; i < end; i += kThreads)
        if (mask.vec ? mask.ptr[i] : sm)
            atomicAdd(&hist[(index_u32(keys[i]) >> shift) & (kRadix - 1)], 1u);
    __syncthreads();
    for (int b = threadIdx.x; b < kRadix; b += kThreads)
        counts[(size_t) b * gridDim.x + blockIdx.x] = hist[b];
}

template <typename T, typename I, int C = 1>
__global__ __launch_bounds__(kThreads) void k_radix_partition_stable(uint32_t *__restrict__ out_keys, BinStreams<T, C> st,
                                                                     const I *__restrict__ keys,
                                                                     Arg<uint8_t> mask, const uint32_t *__restrict__ offsets,
                                                                     const uint32_t *__restrict__ bucket_base, size_t n,
                                                                     size_t chunk, int shift) {
    __shared__ uint32_t wave_count[kSortWaves][kRadix];   // per-wave digit counters, then exclusive over waves
    __shared__ uint32_t total[kRadix], tile_off[kRadix], cursor[kRadix];
    __shared__ uint32_t stage_key[kTile];
    __shared__ T stage_val[kTile];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int b = threadIdx.x; b < kRadix; b += kThreads)
        cursor[b] = bucket_base[b] + offsets[(size_t) b * gridDim.x + blockIdx.x];
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    T sv[C], sw[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        sv[c] = st.value[c].vec ? T(0) : arg_scalar(st.value[c]);
        sw[c] = (((st.weighted >> c) & 1u) && !st.weight[c].vec) ? arg_scalar(st.weight[c]) : T(1);
    }
    const size_t begin = (size_t) blockIdx.x * chunk, end = begin + chunk < n ? begin + chunk : n;

    // values of stream c for the current tile, in the tile's (wave, item, lane) layout, times their weights
    auto load_stream = [&](int c, size_t base, T (&val)[kPerThread]) {
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const size_t i = base + (size_t) wave * (kTile / kSortWaves) + (size_t) j * 64 + lane;
            T v = (st.value[c].vec && i < end) ? st.value[c].ptr[i] : sv[c];
            if constexpr (std::is_floating_point_v<T>) {
                if ((st.weighted >> c) & 1u) v = dev::safe_mul((st.weight[c].vec && i < end) ? st.weight[c].ptr[i] : sw[c], v);
            }
            val[j] = v;
        }
    };

    for (size_t base = begin; base < end; base += kTile) {
        for (int b = threadIdx.x; b < kSortWaves * kRadix; b += kThreads) (&wave_count[0][0])[b] = 0;
        __syncthreads();

        uint32_t key[kPerThread], rank[kPerThread];
        T val[kPerThread];
        bool on[kPerThread];
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            // striped inside the wave: (wave, item, lane) order == element order
            const size_t i = base + (size_t) wave * (kTile / kSortWaves) + (size_t) j * 64 + lane;
            on[j] = i < end && (mask.vec ? mask.ptr[i] != 0 : sm != 0);
            key[j] = i < end ? index_u32(keys[i]) : 0u;
        }
        load_stream(0, base, val);
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const uint32_t d = (key[j] >> shift) & (kRadix - 1);
            unsigned long long peers = __ballot(on[j]);
#pragma unroll
            for (int bit = 0; bit < kRadixBits; ++bit) {
                const bool set = (d >> bit) & 1u;
                const unsigned long long m = __ballot(on[j] && set);
                peers &= set ? m : ~m;
            }
            const uint32_t below = (uint32_t) __popcll(peers & lt_mask);
            uint32_t old = 0;
            if (on[j] && below == 0) {                       // leader of its digit group in this item
                volatile uint32_t *slot = &wave_count[wave][d];
                old = *slot;
                *slot = old + (uint32_t) __popcll(peers);
            }
            const int leader = on[j] ? __ffsll((long long) peers) - 1 : lane;
            old = __shfl(old, leader, 64);
            rank[j] = old + below;
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // per digit: exclusive prefix over the waves, and the tile total
        if (threadIdx.x < kRadix) {
            uint32_t run = 0;
            for (int w = 0; w < kSortWaves; ++w) {
                uint32_t c = wave_count[w][threadIdx.x];
                wave_count[w][threadIdx.x] = run;
                run += c;
            }
            total[threadIdx.x] = run;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            uint32_t h0 = total[4 * l], h1 = total[4 * l + 1], h2 = total[4 * l + 2], h3 = total[4 * l + 3];
            uint32_t sum = h0 + h1 + h2 + h3, incl = sum;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                uint32_t up = __shfl_up(incl, dd, 64);
                if (l >= dd) incl += up;
            }
            uint32_t excl = incl - sum;
            tile_off[4 * l] = excl; tile_off[4 * l + 1] = excl + h0;
            tile_off[4 * l + 2] = excl + h0 + h1; tile_off[4 * l + 3] = excl + h0 + h1 + h2;
        }
        __syncthreads();
        const uint32_t tile_count = tile_off[kRadix - 1] + total[kRadix - 1];
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            if (on[j]) {
                const uint32_t d = (key[j] >> shift) & (kRadix - 1);
                const uint32_t p = tile_off[d] + wave_count[wave][d] + rank[j];
                rank[j] = p;                       // position inside the sorted tile, reused by the other streams
                stage_key[p] = key[j];
                stage_val[p] = val[j];
            }
        }
        if constexpr (C > 1) load_stream(1, base, val);
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < tile_count; s += kThreads) {
            const uint32_t k = stage_key[s], d = (k >> shift) & (kRadix - 1);
            const uint32_t g = cursor[d] + (s - tile_off[d]);
            out_keys[g] = k;
            st.pair_val[0][g] = stage_val[s];
        }
#pragma unroll
        for (int c = 1; c < C; ++c) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kPerThread; ++j)
                if (on[j]) stage_val[rank[j]] = val[j];
            if (c + 1 < C) load_stream(c + 1, base, val);
            __syncthreads();
            for (uint32_t s = threadIdx.x; s < tile_count; s += kThreads) {
                const uint32_t d = (stage_key[s] >> shift) & (kRadix - 1);
                st.pair_val[c][cursor[d] + (s - tile_off[d])] = stage_val[s];
            }
        }
        __syncthreads();
        if (threadIdx.x < kRadix) cursor[threadIdx.x] += total[threadIdx.x];
        __syncthreads();
    }
}

__device__ __forceinline__ float wave_read(float v, int lane) {
    return __uint_as_float((unsigned) __builtin_amdgcn_readlane((int) __float_as_uint(v), lane));
}
__device__ __forceinline__ uint32_t wave_read(uint32_t v, int lane) { return (uint32_t) __builtin_amdgcn_readlane((int) v, lane); }
__device__ __forceinline__ int32_t wave_read(int32_t v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
template <typename T, std::enable_if_t<sizeof(T) == 8, int> = 0> __device__ __forceinline__ T wave_read(T v, int lane) {
    uint64_t bits;
    __builtin_memcpy(&bits, &v, 8);
    uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) bits, lane),
             hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (bits >> 32), lane);
    bits = (uint64_t) lo | ((uint64_t) hi << 32);
    T r;
    __builtin_memcpy(&r, &bits, 8);
    return r;
}

/// One lane per bin: locate the bin's run in the sorted keys and add it sequentially -- the exact chain of additions
/// of the CPU.  Runs longer than a wave (hot bins) are walked by the whole wave on behalf of their lane: 64 values are
/// fetched with one coalesced load and folded in element order through readlane, ~16x faster than a single lane
/// chasing its own loads (a serial chain cannot be parallelised without changing the rounding).
/// Run boundaries of the sorted keys: bin k owns [starts[k], ends[k]) (both zero-initialised: bins without elements keep
/// an empty run).  One streaming pass over the keys instead of two binary searches per bin, shared by all value streams.
__global__ __launch_bounds__(256) void k_segment_bounds(uint32_t *__restrict__ starts, uint32_t *__restrict__ ends,
                                                        const uint32_t *__restrict__ keys, size_t m, size_t table_size) {
    // four consecutive keys per lane (one 16-byte load; the scratch buffer is 16-byte aligned) plus the two neighbours
    const size_t i0 = ((size_t) blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= m) return;
    uint32_t k[6];
    if (i0 + 4 <= m) {
        Pack<uint32_t, 4> p = pack_load<uint32_t, 4, false>(keys + i0);
        k[1] = p.v[0]; k[2] = p.v[1]; k[3] = p.v[2]; k[4] = p.v[3];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) k[1 + j] = i0 + j < m ? keys[i0 + j] : 0xffffffffu;
    }
    k[0] = i0 > 0 ? keys[i0 - 1] : 0xffffffffu;
    k[5] = i0 + 4 < m ? keys[i0 + 4] : 0xffffffffu;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t i = i0 + j;
        const uint32_t key = k[1 + j];
        if (i >= m || key >= table_size) continue;   // out-of-range index: undefined in the reference, ignored here
        if (i == 0 || k[j] != key) starts[key] = (uint32_t) i;
        if (i + 1 == m || k[2 + j] != key) ends[key] = (uint32_t) (i + 1);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_segment_sum(T *__restrict__ target, size_t table_size,
                                                     const uint32_t *__restrict__ starts, const uint32_t *__restrict__ ends,
                                                     const T *__restrict__ vals) {
    const size_t k = (size_t) blockIdx.x * 256 + threadIdx.x;
    const bool valid = k < table_size;
    size_t lo = 0, hi = 0;
    if (valid) { lo = starts[k]; hi = ends[k]; }
    const bool has_run = hi > lo;
    T acc = has_run ? target[k] : T(0);

    // hot bins first, one at a time, cooperatively
    const int lane = threadIdx.x & 63;
    unsigned long long hot = __ballot(has_run && hi - lo > 64);
    while (hot) {
        const int owner = __ffsll((long long) hot) - 1;
        hot &= hot - 1;
        const size_t begin = __shfl(lo, owner), end = __shfl(hi, owner);
        T sum = __shfl(acc, owner);
        T next = begin + (size_t) lane < end ? vals[begin + (size_t) lane] : T(0);
        for (size_t base = begin; base < end; base += 64) {
            const T v = next;
            const size_t ahead = base + 64 + (size_t) lane;            // prefetch the following 64 values
            next = ahead < end ? vals[ahead] : T(0);
            const int cnt = end - base < 64 ? (int) (end - base) : 64;
            // element order; every lane carries the same sum.  v_readlane with a constant / uniform lane index
            // (a generic shuffle would go through the LDS crossbar: ~100 cycles per element)
            if (cnt == 64) {
#pragma unroll
                for (int j = 0; j < 64; ++j) sum += wave_read(v, j);
            } else {
                for (int j = 0; j < cnt; ++j) sum += wave_read(v, j);
            }
        }
        if (lane == owner) { acc = sum; lo = hi; }                     // run consumed
    }
    if (has_run) {
        for (size_t i = lo; i < hi; ++i) acc += vals[i];
        target[k] = acc;
    }
}

// `C` value streams sorted by ONE key array: the keys are ranked, moved and re-read once per pass for all streams,
// then every table sums its own sorted values in element order.
template <typename T, typename I, int C>
int scatter_add_sorted_multi(T *const *bases, size_t table_size, const Arg<T> *values, const Arg<T> *weights, unsigned weighted,
                             const Arg<I> &index, const Arg<uint8_t> &mask, size_t n) {
    Context &c = ctx();
    int bits = 1;
    while (((size_t) 1 << bits) < table_size && bits < 32) ++bits;
    const int passes = (bits + kRadixBits - 1) / kRadixBits;

    unsigned blocks = (unsigned) std::min<size_t>((size_t) c.num_cu * 4, (n + kTile - 1) / kTile);
    if (blocks == 0) blocks = 1;
    size_t chunk = (n + blocks - 1) / blocks;
    chunk = (chunk + kTile - 1) / kTile * kTile;
    blocks = (unsigned) ((n + chunk - 1) / chunk);
    const size_t count_entries = (size_t) kRadix * blocks;

    Scratch counts, keys_a, keys_b, vals_a[C], vals_b[C];
    if (int rc = counts.alloc((count_entries + 2 * kRadix + 1) * sizeof(uint32_t))) return rc;
    if (int rc = keys_a.alloc(n * sizeof(uint32_t))) return rc;
    if (passes > 1)
        if (int rc = keys_b.alloc(n * sizeof(uint32_t))) return rc;
    for (int s = 0; s < C; ++s) {
        if (int rc = vals_a[s].alloc(n * sizeof(T))) return rc;
        if (passes > 1)
            if (int rc = vals_b[s].alloc(n * sizeof(T))) return rc;
    }
    uint32_t *row_total = (uint32_t *) counts.ptr + count_entries;
    uint32_t *bucket_base = row_total + kRadix;

    size_t m = n;                         // valid pairs after the first pass dropped the masked ones
    const uint32_t *in_keys = nullptr;
    const T *in_vals[C] = {};
    for (int p = 0; p < passes; ++p) {
        const int shift = p * kRadixBits;
        uint32_t *out_keys = (uint32_t *) ((p & 1) ? keys_b.ptr : keys_a.ptr);
        BinStreams<T, C> st;
        st.weighted = p == 0 ? weighted : 0u;
        for (int s = 0; s < C; ++s) {
            st.pair_val[s] = (T *) ((p & 1) ? vals_b[s].ptr : vals_a[s].ptr);
            st.value[s] = p == 0 ? values[s] : Arg<T>{ in_vals[s], T(0), 1u };
            st.weight[s] = p == 0 ? weights[s] : Arg<T>{ nullptr, T(1), 0u };
        }
        if (p == 0) {
            hipLaunchKernelGGL((k_radix_count<I>), dim3(blocks), dim3(kThreads), 0, c.stream, (uint32_t *) counts.ptr,
                               index.ptr, mask, n, chunk, shift);
        } else {
            Arg<uint8_t> all_on{ nullptr, 1, 0 };
            hipLaunchKernelGGL((k_radix_count<uint32_t>), dim3(blocks), dim3(kThreads), 0, c.stream,
                               (uint32_t *) counts.ptr, in_keys, all_on, m, chunk, shift);
        }
        hipLaunchKernelGGL(k_bin_scan_rows, dim3(kRadix), dim3(1024), 0, c.stream, (uint32_t *) counts.ptr, row_total, blocks);
        hipLaunchKernelGGL(k_bin_scan_buckets, dim3(1), dim3(256), 0, c.stream, bucket_base, (uint32_t *) nullptr,
                           (const uint32_t *) row_total, kRadix, 0u);
        if (p == 0) {
            hipLaunchKernelGGL((k_radix_partition_stable<T, I, C>), dim3(blocks), dim3(kThreads), 0, c.stream, out_keys, st,
                               index.ptr, mask, (const uint32_t *) counts.ptr, (const uint32_t *) bucket_base, n, chunk, shift);
            if (!mask.vec && mask.ptr == nullptr && mask.imm != 0) {
                m = n;                    // a host-known `true`: every pair is active, nothing to read back (and the path can be
                                          // part of a captured step graph)
            } else {
                uint32_t valid = 0;       // the only synchronisation of the deterministic path
                if (int busy = refuse_while_capturing("deterministic scatter_add under a mask array (the number of active pairs is read back)")) return busy;
                EK_HIP_CHECK(hipMemcpyAsync(&valid, bucket_base + kRadix, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
                EK_HIP_CHECK(hipStreamSynchronize(c.stream));
                m = valid;
            }
        } else {
            Arg<uint8_t> all_on{ nullptr, 1, 0 };
            hipLaunchKernelGGL((k_radix_partition_stable<T, uint32_t, C>), dim3(blocks), dim3(kThreads), 0, c.stream, out_keys, st,
                               in_keys, all_on, (const uint32_t *) counts.ptr, (const uint32_t *) bucket_base, m, chunk, shift);
        }
        // per pass: the count reads the keys (4 B), the partition reads and writes the keys and C value streams
        EK_LAUNCH_CHECK("scatter_add_sort_pass", n, m * (sizeof(uint32_t) + 2 * (sizeof(uint32_t) + C * sizeof(T))));
        in_keys = out_keys;
        for (int s = 0; s < C; ++s) in_vals[s] = st.pair_val[s];
        if (m == 0) return EK_OK;
    }
    Scratch bounds;
    if (int rc = bounds.alloc(2 * table_size * sizeof(uint32_t))) return rc;
    uint32_t *starts = (uint32_t *) bounds.ptr, *ends = starts + table_size;
    EK_HIP_CHECK(hipMemsetAsync(bounds.ptr, 0, 2 * table_size * sizeof(uint32_t), c.stream));
    hipLaunchKernelGGL(k_segment_bounds, dim3((unsigned) ((m + 1023) / 1024)), dim3(256), 0, c.stream, starts, ends, in_keys, m,
                       table_size);
    EK_LAUNCH_CHECK("scatter_add_segment_bounds", m, m * sizeof(uint32_t) + 2 * table_size * sizeof(uint32_t));
    for (int s = 0; s < C; ++s) {
        hipLaunchKernelGGL((k_segment_sum<T>), dim3((unsigned) ((table_size + 255) / 256)), dim3(256), 0, c.stream, bases[s],
                           table_size, (const uint32_t *) starts, (const uint32_t *) ends, in_vals[s]);
        EK_LAUNCH_CHECK("scatter_add_segment_sum", table_size, m * sizeof(T) + table_size * (2 * sizeof(uint32_t) + 2 * sizeof(T)));
    }
    return EK_OK;
}

template <typename T, typename I>
int scatter_add_sorted(T *base, size_t table_size, const Arg<T> &value, const Arg<I> &index, const Arg<uint8_t> &mask,
                       size_t n) {
    const Arg<T> values[1] = { value }, weights[1] = { Arg<T>{ nullptr, T(1), 0u } };
    T *bases[1] = { base };
    return scatter_add_sorted_multi<T, I, 1>(bases, table_size, values, weights, 0u, index, mask, n);
}

// Stable sort of (key, element number) pairs by the low `key_bits` bits of 32-bit keys: the building block of
// partition() (the reference sorts (pointer, lane) pairs with cub::DeviceRadixSort, horiz.cu:35-122).  Same ballot-ranked
// LSD passes as above with the element numbers as the value stream.
int sort_pairs_u32(int key_bits, const uint32_t *keys, size_t n, uint32_t *keys_out, uint32_t *perm_out) {
    RoctxRange range("enoki-hip: sort (key, lane) pairs");
    Context &c = ctx();
    if (key_bits < 1) key_bits = 1;
    const int passes = (key_bits + kRadixBits - 1) / kRadixBits;
    unsigned blocks = (unsigned) std::min<size_t>((size_t) c.num_cu * 4, (n + kTile - 1) / kTile);
    if (blocks == 0) blocks = 1;
    size_t chunk = (n + blocks - 1) / blocks;
    chunk = (chunk + kTile - 1) / kTile * kTile;
    blocks = (unsigned) ((n + chunk - 1) / chunk);
    const size_t count_entries = (size_t) kRadix * blocks;
    Scratch counts, keys_tmp, perm_tmp, iota;
    if (int rc = counts.alloc((count_entries + 2 * kRadix + 1) * sizeof(uint32_t))) return rc;
    if (int rc = keys_tmp.alloc(n * sizeof(uint32_t))) return rc;
    if (int rc = perm_tmp.alloc(n * sizeof(uint32_t))) return rc;
    if (int rc = iota.alloc(n * sizeof(uint32_t))) return rc;
    if (int rc = ek_hip_arange(EK_U32, iota.ptr, 0, 1, n)) return rc;
    uint32_t *row_total = (uint32_t *) counts.ptr + count_entries, *bucket_base = row_total + kRadix;
    const Arg<uint8_t> all_on{ nullptr, 1, 0 };
    const uint32_t *in_keys = keys, *in_vals = (const uint32_t *) iota.ptr;
    for (int p = 0; p < passes; ++p) {
        // ping-pong so that the LAST pass writes the caller's buffers
        const bool to_caller = ((passes - 1 - p) & 1) == 0;
        uint32_t *out_keys = to_caller ? keys_out : (uint32_t *) keys_tmp.ptr;
        BinStreams<uint32_t, 1> st;
        st.weighted = 0u;
        st.pair_val[0] = to_caller ? perm_out : (uint32_t *) perm_tmp.ptr;
        st.value[0] = Arg<uint32_t>{ in_vals, 0u, 1u };
        st.weight[0] = Arg<uint32_t>{ nullptr, 1u, 0u };
        const int shift = p * kRadixBits;
        hipLaunchKernelGGL((k_radix_count<uint32_t>), dim3(blocks), dim3(kThreads), 0, c.stream, (uint32_t *) counts.ptr, in_keys,
                           all_on, n, chunk, shift);
        hipLaunchKernelGGL(k_bin_scan_rows, dim3(kRadix), dim3(1024), 0, c.stream, (uint32_t *) counts.ptr, row_total, blocks);
        hipLaunchKernelGGL(k_bin_scan_buckets, dim3(1), dim3(256), 0, c.stream, bucket_base, (uint32_t *) nullptr,
                           (const uint32_t *) row_total, kRadix, 0u);
        hipLaunchKernelGGL((k_radix_partition_stable<uint32_t, uint32_t, 1>), dim3(blocks), dim3(kThreads), 0, c.stream, out_keys, st,
                           in_keys, all_on, (const uint32_t *) counts.ptr, (const uint32_t *) bucket_base, n, chunk, shift);
        EK_LAUNCH_CHECK("sort_pass", n, n * 5 * sizeof(uint32_t));
        in_keys = out_keys;
        in_vals = st.pair_val[0];
    }
    return EK_OK;
}

#define EK_SORTED_INSTANCE(T, I)                                                                                      \
    template int scatter_add_sorted<T, I>(T *, size_t, const Arg<T> &, const Arg<I> &, const Arg<uint8_t> &, size_t);  \
    template int scatter_add_sorted_multi<T, I, 2>(T *const *, size_t, const Arg<T> *, const Arg<T> *, unsigned,       \
                                                   const Arg<I> &, const Arg<uint8_t> &, size_t);                     \
    template int scatter_add_sorted_multi<T, I, 3>(T *const *, size_t, const Arg<T> *, const Arg<T> *, unsigned,       \
                                                   const Arg<I> &, const Arg<uint8_t> &, size_t);
EK_SORTED_INSTANCE(float, uint32_t) EK_SORTED_INSTANCE(float, int32_t)
EK_SORTED_INSTANCE(double, uint32_t) EK_SORTED_INSTANCE(double, int32_t)

} // namespace ek
