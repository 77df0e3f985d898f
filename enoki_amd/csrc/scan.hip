// Inclusive prefix sum in ONE pass over the data: "decoupled look-back" (Merrill & Garland 2016) on CDNA4.
//
// Replaces the reference's cub::DeviceScan::InclusiveSum (src/cuda/horiz.cu:182-200).  psum() feeds compress() and
// partition() (horiz.cu:124-160) and the PrefixSum special of the tape (autodiff.cpp:473-521).
//
//   * Tiles of 256 lanes x 16 rows x one 16-byte vector (16384 four-byte elements = 64 KiB) are handed out in START order by an
//     atomic ticket, so a tile only ever waits for tiles that are already running -- no deadlock whatever the dispatch
//     order of workgroups is.
//   * A tile scans itself in registers (vector-local scan, wave64 __shfl_up scan, 4 wave totals through LDS), publishes
//     its AGGREGATE, then looks back over the descriptors of its predecessors, 256 per round trip (one per lane), adding
//     aggregates until it meets a tile that already knows its INCLUSIVE prefix; it then publishes its own.
//   * Descriptors cross workgroups (and XCDs, whose L2s are not coherent) as self-validating 8-byte granules
//     {status : 32 | payload : 32} written by ONE write-through (`sc1`) store and polled with `sc1` loads: no fences, no
//     separate flag (MI355X_MICROARCH.md, "Persistent kernels", rows handoff-1to1 / transport-variants).  8-byte element
//     types use two granules (low / high half); a reader that sees two different statuses simply polls again.
//
// Traffic: 4 B read + 4 B written per 4-byte element (the three-pass scan this replaces moved 12 B and ran one of its
// passes on a single thread).  Integer sums are exact.  Floating point sums depend on where the look-back of a tile
// stopped, i.e. on timing: class D like every other fp reduction here, and NOT run-to-run reproducible -- with the
// `deterministic` tuning switch fp prefix sums take the fixed-shape three-pass kernels of reduce.hip instead.
#include "ek_map.h"

namespace ek {

constexpr uint32_t kScanInvalid = 0, kScanAggregate = 1, kScanInclusive = 2;
constexpr int kScanRows = 16;

template <typename T> struct ScanBits;
template <> struct ScanBits<uint32_t> { using type = uint32_t; };
template <> struct ScanBits<int32_t> { using type = uint32_t; };
template <> struct ScanBits<float> { using type = uint32_t; };
template <> struct ScanBits<uint64_t> { using type = uint64_t; };
template <> struct ScanBits<int64_t> { using type = uint64_t; };
template <> struct ScanBits<double> { using type = uint64_t; };

template <typename T> __device__ __forceinline__ void scan_publish(uint64_t *desc, size_t tile, uint32_t status, T value) {
    typename ScanBits<T>::type bits;
    __builtin_memcpy(&bits, &value, sizeof(T));
    if constexpr (sizeof(T) == 4) {
        __hip_atomic_store(desc + tile, ((uint64_t) status << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_store(desc + 2 * tile, ((uint64_t) status << 32) | (uint32_t) bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(desc + 2 * tile + 1, ((uint64_t) status << 32) | (uint32_t) (bits >> 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T> __device__ __forceinline__ void scan_poll(const uint64_t *desc, size_t tile, uint32_t &status, T &value) {
    typename ScanBits<T>::type bits;
    if constexpr (sizeof(T) == 4) {
        const uint64_t g = __hip_atomic_load(desc + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        status = (uint32_t) (g >> 32);
        bits = (uint32_t) g;
    } else {
        const uint64_t lo = __hip_atomic_load(desc + 2 * tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t hi = __hip_atomic_load(desc + 2 * tile + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        status = (uint32_t) (lo >> 32) == (uint32_t) (hi >> 32) ? (uint32_t) (lo >> 32) : kScanInvalid;   // halves of two publications
        bits = ((uint64_t) (uint32_t) hi << 32) | (uint32_t) lo;
    }
    __builtin_memcpy(&value, &bits, sizeof(T));
}

template <typename T> __device__ __forceinline__ T scan_shfl_up(T v, int delta) {
    if constexpr (sizeof(T) == 8) {
        uint64_t u;
        __builtin_memcpy(&u, &v, 8);
        uint32_t lo = __shfl_up((uint32_t) u, delta, 64), hi = __shfl_up((uint32_t) (u >> 32), delta, 64);
        u = ((uint64_t) hi << 32) | lo;
        __builtin_memcpy(&v, &u, 8);
        return v;
    } else {
        return __shfl_up(v, delta, 64);
    }
}

template <typename T> __device__ __forceinline__ T scan_shfl_xor(T v, int mask) {
    if constexpr (sizeof(T) == 8) {
        uint64_t u;
        __builtin_memcpy(&u, &v, 8);
        uint32_t lo = __shfl_xor((uint32_t) u, mask, 64), hi = __shfl_xor((uint32_t) (u >> 32), mask, 64);
        u = ((uint64_t) hi << 32) | lo;
        __builtin_memcpy(&v, &u, 8);
        return v;
    } else {
        return __shfl_xor(v, mask, 64);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_scan_lookback(T *__restrict__ out, const T *__restrict__ in, size_t n, uint64_t *__restrict__ desc,
                                                       unsigned *__restrict__ ticket, int vec_ok) {
    using W = wrap_t<T>;                                    // unsigned arithmetic for integers: wrap-around, no UB
    constexpr int V = 16 / sizeof(T), kRow = 256 * V, kTileElems = kScanRows * kRow;
    __shared__ unsigned s_tile;
    __shared__ T s_wave[kScanRows][4];
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const size_t tile = s_tile, tile_base = tile * (size_t) kTileElems;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // ---- load 4 rows (all in flight), scan each vector locally ----
    Pack<T, V> v[kScanRows];
#pragma unroll
    for (int r = 0; r < kScanRows; ++r) {
        const size_t e = tile_base + (size_t) r * kRow + (size_t) threadIdx.x * V;
        if (vec_ok && e + V <= n) {
            v[r] = pack_load<T, V, true>(in + e);
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) v[r].v[j] = e + j < n ? in[e + j] : T(0);
        }
    }
    T incl[kScanRows];             // (exclusive over the lanes: what is added to this lane's vector)
#pragma unroll
    for (int r = 0; r < kScanRows; ++r) {
#pragma unroll
        for (int j = 1; j < V; ++j) v[r].v[j] = (T) ((W) v[r].v[j] + (W) v[r].v[j - 1]);
        T s = v[r].v[V - 1];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            T up = scan_shfl_up(s, d);
            if (lane >= d) s = (T) ((W) s + (W) up);
        }
        const T below = scan_shfl_up(s, 1);                     // exclusive prefix over the lanes of this wave
        incl[r] = lane == 0 ? T(0) : below;
        if (lane == 63) s_wave[r][wave] = s;
    }
    __syncthreads();
    T aggregate = T(0);
#pragma unroll
    for (int r = 0; r < kScanRows; ++r)
#pragma unroll
        for (int w = 0; w < 4; ++w) aggregate = (T) ((W) aggregate + (W) s_wave[r][w]);

    // ---- publish, look back (all four waves: 256 predecessors per round trip), publish again ----
    __shared__ T s_part[4];
    __shared__ int s_found[4];
    T exclusive = T(0);
    if (tile == 0) {
        if (threadIdx.x == 0) scan_publish(desc, 0, kScanInclusive, aggregate);
    } else {
        if (threadIdx.x == 0) scan_publish(desc, tile, kScanAggregate, aggregate);
        long long base = (long long) tile - 1;
        while (true) {
            // wave w inspects predecessors base - 64 w - lane; the nearest tile that already knows its inclusive prefix
            // ends the walk, everything nearer contributes its aggregate
            const long long t = base - (long long) threadIdx.x;
            uint32_t st = kScanInclusive;                    // "tiles" before tile 0: inclusive prefix 0
            T val = T(0);
            do {
                if (t >= 0) scan_poll(desc, (size_t) t, st, val);
                if (__any(st == kScanInvalid)) __builtin_amdgcn_s_sleep(1);
            } while (__any(st == kScanInvalid));
            const unsigned long long have_prefix = __ballot(st == kScanInclusive);
            const int nearest = have_prefix ? __ffsll((long long) have_prefix) - 1 : 64;
            T part = lane <= nearest ? val : T(0);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part = (T) ((W) part + (W) scan_shfl_xor(part, d));
            if (lane == 0) { s_part[wave] = part; s_found[wave] = have_prefix != 0; }
            __syncthreads();
            bool done = false;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (!done) exclusive = (T) ((W) exclusive + (W) s_part[w]);
                done = done || s_found[w];
            }
            __syncthreads();                                 // s_part / s_found are rewritten by the next round
            if (done) break;
            base -= 256;
        }
        if (threadIdx.x == 0) scan_publish(desc, tile, kScanInclusive, (T) ((W) exclusive + (W) aggregate));
    }

    // ---- store: prefix of the tile + rows before this one + waves before this one + lanes before this one ----
    T running = exclusive;
#pragma unroll
    for (int r = 0; r < kScanRows; ++r) {
        const size_t e = tile_base + (size_t) r * kRow + (size_t) threadIdx.x * V;
        T before = running;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) before = (T) ((W) before + (W) s_wave[r][w]);
            running = (T) ((W) running + (W) s_wave[r][w]);
        }
        const T add = (T) ((W) before + (W) incl[r]);
#pragma unroll
        for (int j = 0; j < V; ++j) v[r].v[j] = (T) ((W) v[r].v[j] + (W) add);
        if (vec_ok && e + V <= n) {
            pack_store<T, V, true>(out + e, v[r]);
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j)
                if (e + j < n) out[e + j] = v[r].v[j];
        }
    }
}

template <typename T> int psum_single_pass(void *out, const void *in, size_t n) {
    RoctxRange range("enoki-hip: prefix sum");
    Context &c = ctx();
    constexpr size_t kTileElems = (size_t) kScanRows * 256 * (16 / sizeof(T));
    const size_t tiles = (n + kTileElems - 1) / kTileElems;
    if (tiles > 0xFFFFFFFFull) return fail(EK_ERR_UNSUPPORTED, "ek_hip_psum(): array too large");
    // descriptors (zero = invalid) followed by the ticket counter
    const size_t desc_bytes = tiles * sizeof(uint64_t) * (sizeof(T) == 8 ? 2 : 1);
    void *scratch = nullptr;
    if (int rc = ek_hip_malloc(desc_bytes + 256, &scratch)) return rc;
    hipError_t e = hipMemsetAsync(scratch, 0, desc_bytes + 256, c.stream);
    if (e != hipSuccess) { ek_hip_free(scratch); return hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__); }
    hipLaunchKernelGGL((k_scan_lookback<T>), dim3((unsigned) tiles), dim3(256), 0, c.stream, (T *) out, (const T *) in, n,
                       (uint64_t *) scratch, (unsigned *) ((char *) scratch + desc_bytes), (int) (aligned16(out) && aligned16(in)));
    ek_hip_free(scratch);          // stream-ordered reuse
    EK_LAUNCH_CHECK("psum", n, 2 * n * sizeof(T));
    return EK_OK;
}

template int psum_single_pass<int32_t>(void *, const void *, size_t);
template int psum_single_pass<uint32_t>(void *, const void *, size_t);
template int psum_single_pass<int64_t>(void *, const void *, size_t);
template int psum_single_pass<uint64_t>(void *, const void *, size_t);
template int psum_single_pass<float>(void *, const void *, size_t);
template int psum_single_pass<double>(void *, const void *, size_t);

} // namespace ek
