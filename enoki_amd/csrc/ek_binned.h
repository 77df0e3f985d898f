// Building blocks of the LDS-binned pipelines: count / scan / partition by table bucket, LDS accumulation under an
// exchange lock, folds.  Shared by scatter_binned.hip (scatter_add) and bucketed.hip (bucket-ordered evaluation of
// gather -> arithmetic -> {reduction, scatter_add} chains).  See scatter_binned.hip for the design notes.
#pragma once
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
        // (an index beyond the table is dropped like a masked-out lane: it must not reach another bucket's counter)
#pragma unroll
        for (int k = 0; k < kPerThread; ++k)
            if (on[k] && (ix[k] >> Shift) < (uint32_t) n_buckets) atomicAdd(&hist[((ix[k] >> Shift) << rep_shift) | rep], 1u);
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
static __global__ __launch_bounds__(1024) void k_bin_scan_rows(uint32_t *__restrict__ counts, uint32_t *__restrict__ row_total,
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
static __global__ __launch_bounds__(256) void k_bin_scan_buckets(uint32_t *__restrict__ bucket_base, uint32_t *__restrict__ piece_prefix,
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
// NoValues: partition of the INDICES alone (ek_hip_index_partition_*): no value stream is read, staged or written.
// amdgpu_waves_per_eu(4) = 128 registers: two workgroups per CU.  The forms with two or three mapped value streams spill under
// it (up to 68 registers, 148 B of scratch per lane) -- measured against a build that lets them have 155 registers (no scratch,
// one workgroup per CU): 0.355 ms against 0.40-0.41 ms per 64 Mi elements, same box.  The spills stay.
template <typename T, typename I, int Shift = kBinShift, typename OutIdx = uint16_t, int C = 1, bool Mapped = false, bool NoValues = false>
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
    __shared__ T stage_val[NoValues ? 1 : kTile];

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
            for (int k = 0; k < kPerThread; ++k) on |= ((flag[k] && (ix[k] >> Shift) < (uint32_t) n_buckets) ? 1u : 0u) << k;
        }
        if constexpr (!NoValues) load_stream(full, 0, base, val);
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
                if constexpr (!NoValues) stage_val[p] = val[k];
            }
        }
        __syncthreads();
        // coalesced runs: consecutive staged elements of one bucket go to consecutive global slots
        for (uint32_t j = threadIdx.x; j < tile_count; j += kThreads) {
            uint32_t key = stage_idx[j], b = key >> Shift;
            uint32_t g = cursor[b] + (j - tile_off[b << rep_shift]);
            pair_idx[g] = (OutIdx) (key & ((1u << Shift) - 1u));   // the bucket is implied by the position
            if constexpr (!NoValues) st.pair_val[0][g] = stage_val[j];
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
template <typename T, int C> struct FoldTargets { T *table[C]; T scale[C]; };     // scale: factor on the folded sum

// Four consecutive entries per lane (one 16-byte access per piece and per target for 4-byte types): a bucket is a multiple of four
// entries, so the four share their bucket and its pieces.  (One entry per lane: 7.9 us for two 1 Mi-entry tables of two pieces each,
// a latency-bound trickle of 4-byte loads.)
constexpr int kFoldPerLane = 4;
inline unsigned fold_grid(size_t table_size) { return (unsigned) ((table_size + 256 * kFoldPerLane - 1) / (256 * kFoldPerLane)); }

template <typename T, int C>
__global__ __launch_bounds__(256) void k_bin_fold_pieces(FoldTargets<T, C> targets, const T *__restrict__ partials,
                                                         const uint32_t *__restrict__ piece_prefix, size_t table_size,
                                                         size_t partial_stride, unsigned fresh = 0u,
                                                         int shift = bin_shift_of<T>) {
    const size_t k0 = ((size_t) blockIdx.x * 256 + threadIdx.x) * kFoldPerLane;
    if (k0 >= table_size) return;
    using U = wrap_t<T>;
    T *__restrict__ target = targets.table[blockIdx.y];
    partials += (size_t) blockIdx.y * partial_stride;
    const uint32_t b = (uint32_t) (k0 >> shift), local = (uint32_t) (k0 & (((size_t) 1 << shift) - 1));
    const bool is_fresh = (fresh >> blockIdx.y) & 1u;       // fresh: the table holds no data yet, its sums are written
    const uint32_t p0 = piece_prefix[b], p1 = piece_prefix[b + 1];
    U sum[kFoldPerLane];
    T old[kFoldPerLane];
#pragma unroll
    for (int j = 0; j < kFoldPerLane; ++j) { sum[j] = U(0); old[j] = T(0); }
    const bool vec = sizeof(T) == 4 && k0 + kFoldPerLane <= table_size && ((reinterpret_cast<uintptr_t>(target + k0) | reinterpret_cast<uintptr_t>(partials)) & 15u) == 0;
    if (vec) {
        if constexpr (sizeof(T) == 4) {
            if (!is_fresh) {
                const Pack<T, 4> t = pack_load<T, 4, false>(target + k0);
#pragma unroll
                for (int j = 0; j < 4; ++j) old[j] = t.v[j];
            }
            for (uint32_t p = p0; p < p1; ++p) {
                const Pack<T, 4> v = pack_load<T, 4, true>(partials + ((size_t) p << shift) + local);
#pragma unroll
                for (int j = 0; j < 4; ++j) sum[j] = (U) (sum[j] + (U) v.v[j]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < kFoldPerLane; ++j) {
            if (k0 + j >= table_size) continue;
            if (!is_fresh) old[j] = target[k0 + j];
            for (uint32_t p = p0; p < p1; ++p) sum[j] = (U) (sum[j] + (U) partials[((size_t) p << shift) + local + j]);
        }
    }
    T out[kFoldPerLane];
#pragma unroll
    for (int j = 0; j < kFoldPerLane; ++j) {
        U v = sum[j];
        if constexpr (std::is_floating_point_v<T>) {
            const T f = targets.scale[blockIdx.y];
            if (f != T(1)) v = v * f;
        }
        out[j] = (T) ((U) old[j] + v);
    }
    if (vec) {
        if constexpr (sizeof(T) == 4) {
            Pack<T, 4> o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o.v[j] = out[j];
            pack_store<T, 4, false>(target + k0, o);
        }
    } else {
#pragma unroll
        for (int j = 0; j < kFoldPerLane; ++j)
            if (k0 + j < table_size) target[k0 + j] = out[j];
    }
}

struct Scratch {
    void *ptr = nullptr;
    ~Scratch() { if (ptr) ek_hip_free(ptr); }      // stream-ordered: safe to hand back right after enqueueing
    int alloc(size_t bytes) { return ek_hip_malloc(bytes, &ptr); }
};

} // namespace ek
