// scatter_add through the LDS: the MI355X-native replacement for `atom.global.add` (cuda.h:892-905).
//
// Why: device-scope floating point atomics on gfx950 retire at ~21 G atomics/s no matter how the
// addresses are spread (measured: 64 Mi random adds into a 1 Mi-entry table take 3.19 ms, the same
// with one private table per XCD -- profiles/probe_r01.txt), i.e. 2 % of the HBM roofline.  A CU's
// 160 KiB LDS, on the other hand, holds 16 Ki f32 bins (64 KiB, two workgroups per CU) and executes
// ds_add_f32 at LDS speed.  So the adjoint of gather is restructured as
//
//   1. count     every workgroup owns a contiguous chunk of the n elements and histograms its
//                indices by BUCKET (= index >> 14) in LDS                       reads  4 B/elt
//   2. scan      exclusive scan of the (bucket-major, workgroup-minor) counts   tiny
//   3. partition each workgroup re-reads its chunk and appends (index & 16383, value) to its slice of
//                the bucket's pair list (LDS cursors)                           reads 8, writes 8 B/elt
//   4. accumulate S workgroups per bucket stream the bucket's pairs and ds_add them into a
//                zeroed LDS table, then write their partial table               reads  8 B/elt
//   5. fold      target[k] += sum_s partial[s][k]                               (S + 2) * 4 B per bin
//
// = 28 B/elt of coalesced streaming traffic and no global atomics.  Tables of <= 16 Ki bins skip
// steps 1-3.  The result is the same set of additions as the atomic version in a different
// (unspecified) order -- parity class D, like the reference's own GPU path.
#include "ek_map.h"

#include <algorithm>

namespace ek {

constexpr int kBinShift = 14;
constexpr int kBins = 1 << kBinShift;      // bins per bucket (64 KiB of f32 / i32 in LDS)
constexpr int kMaxBuckets = 256;
constexpr int kThreads = 512;

template <typename I> __device__ __forceinline__ uint32_t index_u32(I i) { return (uint32_t) i; }

// ---- 1. count ------------------------------------------------------------------------------------
template <typename I>
__global__ __launch_bounds__(kThreads) void k_bin_count(uint32_t *__restrict__ counts, const I *__restrict__ index,
                                                        Arg<uint8_t> mask, size_t n, size_t chunk, int n_buckets) {
    __shared__ uint32_t hist[kMaxBuckets];
    for (int b = threadIdx.x; b < n_buckets; b += kThreads) hist[b] = 0;
    __syncthreads();
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    const size_t begin = (size_t) blockIdx.x * chunk, end = begin + chunk < n ? begin + chunk : n;
    for (size_t i = begin + threadIdx.x; i < end; i += kThreads) {
        if (mask.vec ? mask.ptr[i] : sm)
            atomicAdd(&hist[index_u32(index[i]) >> kBinShift], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n_buckets; b += kThreads)
        counts[(size_t) b * gridDim.x + blockIdx.x] = hist[b];
}

// ---- 2. scan (single workgroup; the array has n_buckets * n_blocks <= 256 * 2048 entries) -----------
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t *__restrict__ counts, size_t count, uint32_t *__restrict__ total) {
    __shared__ uint32_t part[1024];
    const size_t per = (count + 1023) / 1024;
    const size_t begin = threadIdx.x * per, end = begin + per < count ? begin + per : count;
    uint32_t s = 0;
    for (size_t i = begin; i < end; ++i) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        uint32_t add = threadIdx.x >= (unsigned) d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
    for (size_t i = begin; i < end; ++i) {
        uint32_t c = counts[i];
        counts[i] = run;
        run += c;
    }
    if (threadIdx.x == 1023) *total = part[1023];
}

// ---- 3. partition ----------------------------------------------------------------------------------
template <typename T, typename I>
__global__ __launch_bounds__(kThreads) void k_bin_partition(uint32_t *__restrict__ pair_bin, T *__restrict__ pair_val,
                                                            const uint32_t *__restrict__ offsets, Arg<T> value,
                                                            const I *__restrict__ index, Arg<uint8_t> mask, size_t n,
                                                            size_t chunk, int n_buckets) {
    __shared__ uint32_t cursor[kMaxBuckets];
    for (int b = threadIdx.x; b < n_buckets; b += kThreads)
        cursor[b] = offsets[(size_t) b * gridDim.x + blockIdx.x];
    __syncthreads();
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    const T sv = value.vec ? T(0) : arg_scalar(value);
    const size_t begin = (size_t) blockIdx.x * chunk, end = begin + chunk < n ? begin + chunk : n;
    for (size_t i = begin + threadIdx.x; i < end; i += kThreads) {
        if (mask.vec ? mask.ptr[i] : sm) {
            uint32_t ix = index_u32(index[i]);
            uint32_t pos = atomicAdd(&cursor[ix >> kBinShift], 1u);
            pair_bin[pos] = ix & (kBins - 1);
            pair_val[pos] = value.vec ? value.ptr[i] : sv;
        }
    }
}

// ---- 4. accumulate ---------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void lds_add(T *addr, T v) {
    if constexpr (std::is_same_v<T, float>) atomicAdd(addr, v);                 // ds_add_f32
    else atomicAdd(reinterpret_cast<unsigned int *>(addr), (unsigned int) v);   // ds_add_u32
}

// Pairs come either from the partition (Direct = false: bucket b owns pairs [bucket_begin[b], bucket_begin[b+1]))
// or straight from the operands when the whole table fits one bucket (Direct = true).
template <typename T, typename I, bool Direct>
__global__ __launch_bounds__(kThreads) void k_bin_accumulate(T *__restrict__ partials, size_t table_size,
                                                             const uint32_t *__restrict__ pair_bin,
                                                             const T *__restrict__ pair_val,
                                                             const uint32_t *__restrict__ offsets, size_t offsets_stride,
                                                             const uint32_t *__restrict__ total, Arg<T> value,
                                                             const I *__restrict__ index, Arg<uint8_t> mask, size_t n,
                                                             int slices) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T *acc = reinterpret_cast<T *>(lds_raw);
    const int bucket = blockIdx.x / slices, slice = blockIdx.x % slices;
    for (int j = threadIdx.x; j < kBins; j += kThreads) acc[j] = T(0);
    __syncthreads();

    if constexpr (Direct) {
        const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
        const T sv = value.vec ? T(0) : arg_scalar(value);
        const size_t per = (n + slices - 1) / slices;
        const size_t begin = (size_t) slice * per, end = begin + per < n ? begin + per : n;
        for (size_t i = begin + threadIdx.x; i < end; i += kThreads)
            if (mask.vec ? mask.ptr[i] : sm)
                lds_add(&acc[index_u32(index[i]) & (kBins - 1)], value.vec ? value.ptr[i] : sv);
    } else {
        const int n_buckets = gridDim.x / slices;
        const size_t lo = offsets[(size_t) bucket * offsets_stride];
        const size_t hi = bucket + 1 < n_buckets ? (size_t) offsets[(size_t) (bucket + 1) * offsets_stride] : (size_t) *total;
        const size_t cnt = hi - lo, per = ((cnt + slices - 1) / slices + 3) & ~size_t(3);
        size_t begin = lo + (size_t) slice * per, end = begin + per < hi ? begin + per : hi;
        if (begin > hi) begin = hi;
        for (size_t i = begin + threadIdx.x; i < end; i += kThreads)
            lds_add(&acc[pair_bin[i]], pair_val[i]);
    }
    __syncthreads();

    T *out = partials + (size_t) slice * table_size + (size_t) bucket * kBins;
    const size_t valid = table_size - (size_t) bucket * kBins;
    for (int j = threadIdx.x; j < kBins; j += kThreads)
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

struct Scratch {
    void *ptr = nullptr;
    ~Scratch() { if (ptr) ek_hip_free(ptr); }      // stream-ordered: safe to hand back right after enqueueing
    int alloc(size_t bytes) { return ek_hip_malloc(bytes, &ptr); }
};

template <typename T, typename I>
int scatter_add_binned(T *base, size_t table_size, const Arg<T> &value, const Arg<I> &index, const Arg<uint8_t> &mask,
                       size_t n) {
    Context &c = ctx();
    const int n_buckets = (int) ((table_size + kBins - 1) / kBins);
    const size_t lds_bytes = (size_t) kBins * sizeof(T);
    const size_t algo_bytes = arg_bytes(value, n) + arg_bytes(index, n) + arg_bytes(mask, n);

    if (n_buckets == 1) {
        int slices = std::max(1, std::min(2 * c.num_cu, (int) (n / 65536)));
        Scratch partials;
        if (int rc = partials.alloc((size_t) slices * table_size * sizeof(T))) return rc;
        hipLaunchKernelGGL((k_bin_accumulate<T, I, true>), dim3(slices), dim3(kThreads), lds_bytes, c.stream,
                           (T *) partials.ptr, table_size, nullptr, nullptr, nullptr, (size_t) 0, nullptr, value,
                           index.ptr, mask, n, slices);
        EK_LAUNCH_CHECK("scatter_add_lds", n, algo_bytes);
        hipLaunchKernelGGL((k_bin_fold<T>), dim3((unsigned) ((table_size + 255) / 256)), dim3(256), 0, c.stream, base,
                           (const T *) partials.ptr, table_size, slices);
        EK_LAUNCH_CHECK("scatter_add_fold", table_size, (size_t) (slices + 2) * table_size * sizeof(T));
        return EK_OK;
    }

    // chunked passes over the input: enough workgroups to fill the chip, chunks of >= 32 Ki elements
    unsigned blocks = (unsigned) std::min<size_t>((size_t) c.num_cu * 4, (n + 32767) / 32768);
    if (blocks == 0) blocks = 1;
    size_t chunk = (n + blocks - 1) / blocks;
    chunk = (chunk + 1023) / 1024 * 1024;
    blocks = (unsigned) ((n + chunk - 1) / chunk);

    const size_t count_entries = (size_t) n_buckets * blocks;
    Scratch counts, pairs_bin, pairs_val, partials;
    if (int rc = counts.alloc((count_entries + 1) * sizeof(uint32_t))) return rc;
    if (int rc = pairs_bin.alloc(n * sizeof(uint32_t))) return rc;
    if (int rc = pairs_val.alloc(n * sizeof(T))) return rc;
    uint32_t *total = (uint32_t *) counts.ptr + count_entries;

    hipLaunchKernelGGL((k_bin_count<I>), dim3(blocks), dim3(kThreads), 0, c.stream, (uint32_t *) counts.ptr, index.ptr,
                       mask, n, chunk, n_buckets);
    EK_LAUNCH_CHECK("scatter_add_count", n, arg_bytes(index, n) + arg_bytes(mask, n));
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, c.stream, (uint32_t *) counts.ptr, count_entries, total);
    EK_LAUNCH_CHECK("scatter_add_scan", count_entries, 2 * count_entries * sizeof(uint32_t));
    hipLaunchKernelGGL((k_bin_partition<T, I>), dim3(blocks), dim3(kThreads), 0, c.stream, (uint32_t *) pairs_bin.ptr,
                       (T *) pairs_val.ptr, (const uint32_t *) counts.ptr, value, index.ptr, mask, n, chunk, n_buckets);
    EK_LAUNCH_CHECK("scatter_add_partition", n, algo_bytes + n * (sizeof(uint32_t) + sizeof(T)));

    int slices = std::max(1, (4 * c.num_cu + n_buckets - 1) / n_buckets);
    if (int rc = partials.alloc((size_t) slices * table_size * sizeof(T))) return rc;
    hipLaunchKernelGGL((k_bin_accumulate<T, I, false>), dim3((unsigned) (n_buckets * slices)), dim3(kThreads), lds_bytes,
                       c.stream, (T *) partials.ptr, table_size, (const uint32_t *) pairs_bin.ptr,
                       (const T *) pairs_val.ptr, (const uint32_t *) counts.ptr, (size_t) blocks, (const uint32_t *) total,
                       value, index.ptr, mask, n, slices);
    EK_LAUNCH_CHECK("scatter_add_accumulate", n, n * (sizeof(uint32_t) + sizeof(T)) + (size_t) slices * table_size * sizeof(T));
    hipLaunchKernelGGL((k_bin_fold<T>), dim3((unsigned) ((table_size + 255) / 256)), dim3(256), 0, c.stream, base,
                       (const T *) partials.ptr, table_size, slices);
    EK_LAUNCH_CHECK("scatter_add_fold", table_size, (size_t) (slices + 2) * table_size * sizeof(T));
    return EK_OK;
}

// entry points used by ek_hip_scatter_add (memory.hip)
bool scatter_add_binned_applicable(size_t table_size, size_t n, bool index_is_array) {
    return index_is_array && table_size > 0 && n >= ((size_t) 1 << 18) &&
           table_size <= (size_t) kMaxBuckets * kBins && n < ((size_t) 1 << 32);
}

#define EK_BINNED_INSTANCE(T, I)                                                                                      \
    template int scatter_add_binned<T, I>(T *, size_t, const Arg<T> &, const Arg<I> &, const Arg<uint8_t> &, size_t);
EK_BINNED_INSTANCE(float, uint32_t) EK_BINNED_INSTANCE(float, int32_t)
EK_BINNED_INSTANCE(uint32_t, uint32_t) EK_BINNED_INSTANCE(uint32_t, int32_t)

} // namespace ek
