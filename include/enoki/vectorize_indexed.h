/*
    enoki/vectorize_indexed.h -- vectorize_through(f, index, mask, target, sources...): a program whose gather and scatter go
    through the SAME index array, executed per TARGET entry instead of per element

        v = gather(sources..., index, mask);   (value, hit) = f(v...);   scatter(target, value, index, hit & mask);   count(hit & mask)

    The reference runs this shape (tests/sphere.cpp:58-83 behind a pixel permutation: BASELINE config 4) as three vectorize()
    calls around a gather and a scatter; its JIT fuses them into one kernel with TWO random accesses per element (the gather's
    ld.global and the scatter's st.global, cuda.h:845-890).  On the MI355X a random 4-byte access into arrays that do not fit
    the L2 is a 64-byte memory transaction served at ~55 G accesses/s, so the fused kernel of enoki/vectorize.h stops at
    27 G elements/s (44 with packed source records) whatever its arithmetic.

    When `f` depends on the gathered values only -- which is what makes gather and scatter through one index array a
    "permuted map" -- the random accesses can be traded for streaming ones:

      1. ek_hip_index_partition_create() groups the active entries of `index` by bucket of 4 Ki .. 512 Ki target entries
         (count + scan + partition: 14 B per entry, all streaming);
      2. ONE workgroup per bucket (k_vectorize_through below)
           a. marks, in an LDS bitmap, the target entries that at least one active element points at   4 B per element
           b. walks its slice of the target range IN ORDER: coalesced loads of the sources at the marked entries, f, coalesced
              stores of the value where `hit`, and a second bitmap of the entries that were hit   sources + 4 B per target entry
           c. counts the active elements whose entry was hit (what count(hit & mask) returns)          4 B per element (L2)

    Results are identical to the element-order program: every element that points at entry j computes f(sources[j]) -- the same
    value, so which of several duplicates writes last does not matter -- and an entry nobody points at is not touched.

    Requirements as for enoki/vectorize.h (hipcc translation unit, this header or vectorize.h first, user templates between
    ENOKI_DEVICE_CODE_BEGIN / END).  `f` takes one one-element packet per source and returns std::pair<Packet, mask_t<Packet>>.
*/
#pragma once

#include <enoki/vectorize.h>

namespace enoki {

namespace detail {
    template <size_t N> struct ThroughSources { const float *ptr[N]; };

    template <typename Func, size_t N, size_t... Is>
    __device__ __forceinline__ auto through_eval(const Func &f, const float (&v)[N], std::index_sequence<Is...>) {
        using Packet = Array<float, 1>;
        return f(Packet(v[Is])...);
    }

    // LDS: two bitmaps of 2^shift bits (marked / hit); 1024 threads; one workgroup per bucket.  Every loop keeps several
    // loads per lane in flight (8 list entries; 2 x 4 consecutive target entries per source): with one load per lane and
    // iteration the kernel would be bound by memory latency, not bandwidth.
    template <typename Func, size_t N>
    __global__ __launch_bounds__(1024) void k_vectorize_through(Func f, float *__restrict__ target, ThroughSources<N> src, size_t range,
                                                                int shift, int vec_ok, const uint32_t *__restrict__ bucket_base,
                                                                const uint32_t *__restrict__ local,
                                                                unsigned long long *__restrict__ hit_count) {
        extern __shared__ uint32_t through_bits[];
        typedef float __attribute__((ext_vector_type(4))) float4v;
        const uint32_t words = 1u << (shift - 5);
        uint32_t *marked = through_bits, *hit = through_bits + words;
        for (uint32_t w = threadIdx.x; w < 2 * words; w += 1024) through_bits[w] = 0u;
        __syncthreads();
        const uint32_t begin = bucket_base[blockIdx.x], end = bucket_base[blockIdx.x + 1];
        constexpr int U = 16;
        // (a) which target entries does an active element point at?
        for (uint32_t base = begin; base < end; base += U * 1024) {
            uint32_t l[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const uint32_t i = base + k * 1024 + threadIdx.x;
                l[k] = i < end ? __builtin_nontemporal_load(local + i) : ~0u;
            }
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (l[k] != ~0u) atomicOr(&marked[l[k] >> 5], 1u << (l[k] & 31u));
        }
        __syncthreads();
        // (b) the bucket's slice of the target range, in order
        const size_t first = (size_t) blockIdx.x << shift;
        const uint32_t entries = (uint32_t) (range - first < ((size_t) 1 << shift) ? range - first : ((size_t) 1 << shift));
        auto one = [&](uint32_t l, const float (&v)[N]) {
            auto r = through_eval(f, v, std::make_index_sequence<N>());
            if (r.second.coeff(0)) {
                target[first + l] = r.first.coeff(0);
                atomicOr(&hit[l >> 5], 1u << (l & 31u));
            }
        };
        if (vec_ok) {
            constexpr int V = 4;
            for (uint32_t base = 0; base < entries; base += V * 4096) {
                float4v v[V][N];
                uint32_t lv[V], bits[V];
#pragma unroll
                for (int h = 0; h < V; ++h) {
                    lv[h] = base + h * 4096 + threadIdx.x * 4;
                    bits[h] = lv[h] + 4 <= entries ? (marked[lv[h] >> 5] >> (lv[h] & 31u)) & 15u : 0u;
                    if (bits[h]) {
#pragma unroll
                        for (size_t s = 0; s < N; ++s)
                            v[h][s] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(src.ptr[s] + first + lv[h]));
                    }
                }
#pragma unroll
                for (int h = 0; h < V; ++h) {
                    if (!bits[h]) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!((bits[h] >> j) & 1u)) continue;
                        float in[N];
#pragma unroll
                        for (size_t s = 0; s < N; ++s) in[s] = v[h][s][j];
                        one(lv[h] + j, in);
                    }
                }
            }
            // ragged end of the range (fewer than 4 entries left for a lane)
            const uint32_t tail = entries & ~3u;
            if (threadIdx.x < entries - tail) {
                const uint32_t l = tail + threadIdx.x;
                if ((marked[l >> 5] >> (l & 31u)) & 1u) {
                    float in[N];
#pragma unroll
                    for (size_t s = 0; s < N; ++s) in[s] = src.ptr[s][first + l];
                    one(l, in);
                }
            }
        } else {
            for (uint32_t l = threadIdx.x; l < entries; l += 1024) {
                if (!((marked[l >> 5] >> (l & 31u)) & 1u)) continue;
                float in[N];
#pragma unroll
                for (size_t s = 0; s < N; ++s) in[s] = src.ptr[s][first + l];
                one(l, in);
            }
        }
        __syncthreads();
        // (c) count(hit & mask): the active elements whose entry was hit
        unsigned count = 0;
        for (uint32_t base = begin; base < end; base += U * 1024) {
            uint32_t l[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const uint32_t i = base + k * 1024 + threadIdx.x;
                l[k] = i < end ? local[i] : ~0u;
            }
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (l[k] != ~0u) count += (hit[l[k] >> 5] >> (l[k] & 31u)) & 1u;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) count += __shfl_down(count, d, 64);
        __shared__ unsigned wave_count[16];
        if ((threadIdx.x & 63) == 0) wave_count[threadIdx.x >> 6] = count;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long total = 0;
            for (int w = 0; w < 16; ++w) total += wave_count[w];
            if (total) atomicAdd(hit_count, total);
        }
    }
}

/// Returns count(hit & mask).  `target` (entries nobody hits keep their contents) and every source have the same length, the
/// range that `index` points into.  Synchronises once (the count is read back), like count() itself.
template <typename Func, typename... Sources>
size_t vectorize_through(Func f, const HIPArray<uint32_t> &index, const HIPArray<bool> &mask, HIPArray<float> &target,
                         const Sources &... sources) {
    static_assert((std::is_same_v<Sources, HIPArray<float>> && ...), "vectorize_through(): float32 source arrays expected");
    constexpr size_t N = sizeof...(Sources);
    static_assert(N >= 1, "vectorize_through(): at least one source array");
    const size_t n = index.size(), range = target.size();
    if (((sources.size() != range) || ...))
        throw std::runtime_error("vectorize_through(): the sources and the target must have the same length");
    if (n == 0 || range == 0) return 0;
    ek_operand om = mask.operand();
    ek_hip_index_partition *part = nullptr;
    detail::hip_check(ek_hip_index_partition_create(HIPArray<uint32_t>::Type, index.data(), &om, n, range, &part), "vectorize_through");
    ek_hip_index_partition_info info;
    ek_hip_index_partition_get(part, &info);
    target.make_unique();
    float *out = target.data();
    detail::ThroughSources<N> src{ { sources.data()... } };
    int vec_ok = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    for (size_t s = 0; s < N; ++s) vec_ok = vec_ok && (reinterpret_cast<uintptr_t>(src.ptr[s]) & 15u) == 0;
    void *counter = nullptr;
    detail::hip_check(ek_hip_malloc(sizeof(unsigned long long), &counter), "vectorize_through");
    detail::hip_check(ek_hip_memset(counter, 0, sizeof(unsigned long long)), "vectorize_through");
    hipStream_t stream = (hipStream_t) ek_hip_stream();
    const size_t lds = (size_t) 2 << (info.shift - 3);
    auto kernel = detail::k_vectorize_through<Func, N>;
    if (lds > 65536) (void) hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    hipLaunchKernelGGL(kernel, dim3((unsigned) info.n_buckets), dim3(1024), lds, stream, f, out, src, range, info.shift, vec_ok,
                       info.bucket_base, info.local, (unsigned long long *) counter);
    // algorithmic bytes: the bucket lists twice, the sources and the target once
    detail::hip_check(ek_hip_note_launch("vectorize_through", n, 8 * n + (N + 1) * 4 * range), "vectorize_through");
    unsigned long long hits = 0;
    int rc = ek_hip_memcpy_to_host(&hits, counter, sizeof(hits));       // synchronises
    ek_hip_free(counter);
    ek_hip_index_partition_destroy(part);
    detail::hip_check(rc, "vectorize_through");
    return (size_t) hits;
}

} // namespace enoki
