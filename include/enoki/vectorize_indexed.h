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
           a. counts, in one LDS byte per target entry, the active elements that point at it             4 B per element
           b. walks its slice of the target range IN ORDER: coalesced loads of the sources at the entries somebody points
              at, f, coalesced stores of the value where `hit`; count(hit & mask) is the sum of the bytes of the entries
              that were hit                                                                sources + 4 B per target entry
         (buckets of 512 Ki entries do not fit the LDS as bytes: two bitmaps instead, marked / hit, and a second pass over
         the bucket's list that counts the elements whose entry was hit; the same after a byte counter overflowed)
      3. the workgroup that finishes last publishes the count to pinned host memory: the host synchronises the stream and
         reads it -- no memset launch, no device -> host copy.

    vectorize_through_fill(f, index, mask, fill_value, target, sources...) is `target = full(fill_value); vectorize_through(...)`
    in one pass: step 2b writes EVERY entry (the fill value where nobody hit) with full-width stores.

    Results are identical to the element-order program: every element that points at entry j computes f(sources[j]) -- the same
    value, so which of several duplicates writes last does not matter -- and an entry nobody points at is not touched (Fill:
    holds the fill value).

    Requirements as for enoki/vectorize.h (hipcc translation unit, this header or vectorize.h first, user templates between
    ENOKI_DEVICE_CODE_BEGIN / END).  `f` takes one one-element packet per source and returns std::pair<Packet, mask_t<Packet>>.
*/
#pragma once

#include <enoki/vectorize.h>
#include <mutex>

namespace enoki {

namespace detail {
    template <size_t N> struct ThroughSources { const float *ptr[N]; };

    /// the paged layout of ek_hip_index_partition_info (page_shift == 0: one contiguous run per bucket)
    struct ThroughPages { int page_shift; const uint32_t *full, *part, *part_base; };

    template <typename Func, size_t N, size_t... Is>
    __device__ __forceinline__ auto through_eval(const Func &f, const float (&v)[N], std::index_sequence<Is...>) {
        using Packet = Array<float, 1>;
        return f(Packet(v[Is])...);
    }

    /// What one call leaves behind for the host: {count(hit & mask), "a multiplicity counter overflowed"}.  `state` lives in
    /// device memory and cleans up after itself (the workgroup that finishes last publishes the totals to the pinned host
    /// slot and zeroes the state), so a call costs no memset launch and no device -> host copy: the host synchronises the
    /// stream and reads two words of pinned memory.
    struct ThroughState { unsigned long long total; unsigned int done, overflow; };

    struct ThroughScratch {
        std::mutex mutex;                       // held for the duration of a call (launch .. readback)
        ThroughState *state = nullptr;          // device
        volatile unsigned long long *host = nullptr;   // pinned: [0] = total, [1] = overflow
    };
    inline ThroughScratch &through_scratch() { static ThroughScratch s; return s; }

    // One workgroup of 1024 threads per bucket of 2^shift target entries.  Every loop keeps several loads per lane in flight
    // (16 list entries; 4 x 4 consecutive target entries per source): with one load per lane and iteration the kernel would
    // be bound by memory latency, not bandwidth.
    //
    //   Counters (shift <= 17): LDS holds one BYTE per target entry, the number of active elements that point at it -- an entry
    //     is marked when its byte is not zero, and count(hit & mask) is the sum of the bytes of the entries that were hit, so
    //     the bucket's list is read ONCE.  A 256th duplicate of one entry would carry into its neighbour: the workgroup
    //     notices (the atomic returns the old word), raises `overflow` and gives up, and the host repeats the call with
    //     bitmaps (every store of the first attempt is repeated with the same value).
    //   Bitmaps (any shift): two bitmaps of 2^shift bits (marked / hit), and a second pass over the list that counts the
    //     elements whose entry was hit.
    //   Fill: EVERY entry of the target is written -- f's value where an active element hit, `fill_value` elsewhere -- which
    //     is `target = full(fill_value); scatter(target, ...)` without the separate pass over the target, and with full-width
    //     stores instead of masked ones.
    template <typename Func, size_t N, bool Fill, bool Counters>
    __global__ __launch_bounds__(1024) void k_vectorize_through(Func f, float *__restrict__ target, ThroughSources<N> src, size_t range,
                                                                int shift, int vec_ok, float fill_value,
                                                                const uint32_t *__restrict__ bucket_base,
                                                                const uint32_t *__restrict__ local, ThroughState *__restrict__ state,
                                                                volatile unsigned long long *__restrict__ host, ThroughPages pages) {
        extern __shared__ uint32_t through_lds[];
        typedef float __attribute__((ext_vector_type(4))) float4v;
        __shared__ unsigned wave_count[16];
        __shared__ unsigned gave_up;
        // Counters: 2^shift bytes;  bitmaps: 2 x 2^shift bits
        const uint32_t words = Counters ? 1u << (shift - 2) : 2u << (shift - 5);
        uint32_t *marked = through_lds, *hit = through_lds + (1u << (shift - 5));
        for (uint32_t w = threadIdx.x; w < words; w += 1024) through_lds[w] = 0u;
        if (threadIdx.x == 0) gave_up = 0u;
        __syncthreads();
        const uint32_t begin = bucket_base[blockIdx.x], end = bucket_base[blockIdx.x + 1];
        constexpr int U = 16;
        // Every bucket-local index of this bucket once, many loads per lane in flight.  Contiguous run: 16 x 4 bytes per lane and
        // trip.  Pages (round 6, the single-pass partition): 2^page_shift / 4 lanes share a page, one 16-byte vector each, four
        // pages per group of lanes and trip; the partially filled pages (entry = page << 6 | count - 1) behind them under a
        // lane predicate.
        auto for_each_local = [&](auto &&fn) {
            if (pages.page_shift == 0) {
                for (uint32_t base = begin; base < end; base += U * 1024) {
                    uint32_t l[U];
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        const uint32_t i = base + k * 1024 + threadIdx.x;
                        l[k] = i < end ? __builtin_nontemporal_load(local + i) : ~0u;
                    }
#pragma unroll
                    for (int k = 0; k < U; ++k)
                        if (l[k] != ~0u) fn(l[k]);
                }
            } else {
                typedef uint32_t __attribute__((ext_vector_type(4))) uint4v;
                const uint32_t lx = (1u << pages.page_shift) / 4u, groups = 1024u / lx, g = threadIdx.x / lx, i = threadIdx.x % lx;
                constexpr int P = 4;
                for (uint32_t q0 = begin + g; q0 < end; q0 += P * groups) {
                    uint32_t pg[P];
                    uint4v v[P];
#pragma unroll
                    for (int k = 0; k < P; ++k) pg[k] = q0 + k * groups < end ? __builtin_nontemporal_load(pages.full + q0 + k * groups) : ~0u;
#pragma unroll
                    for (int k = 0; k < P; ++k)
                        if (pg[k] != ~0u) v[k] = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(local + ((size_t) pg[k] << pages.page_shift) + 4u * i));
#pragma unroll
                    for (int k = 0; k < P; ++k)
                        if (pg[k] != ~0u) { fn(v[k][0]); fn(v[k][1]); fn(v[k][2]); fn(v[k][3]); }
                }
                const uint32_t pb = pages.part_base[blockIdx.x], pe = pages.part_base[blockIdx.x + 1];
                for (uint32_t q = pb + g; q < pe; q += groups) {
                    const uint32_t e = __builtin_nontemporal_load(pages.part + q), cnt = (e & 63u) + 1u;
                    const uint4v v = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(local + ((size_t) (e >> 6) << pages.page_shift) + 4u * i));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4u * i + j < cnt) fn(v[j]);
                }
            }
        };
        // (a) which target entries does an active element point at (Counters: how many of them)?
        for_each_local([&](uint32_t l) {
            if constexpr (Counters) {
                const uint32_t sh = (l & 3u) * 8u;
                const uint32_t old = atomicAdd(&through_lds[l >> 2], 1u << sh);
                if (((old >> sh) & 255u) == 255u) gave_up = 1u;
            } else {
                atomicOr(&marked[l >> 5], 1u << (l & 31u));
            }
        });
        __syncthreads();
        unsigned count = 0;
        const bool abandon = Counters && gave_up;
        if (!abandon) {
            // (b) the bucket's slice of the target range, in order
            const size_t first = (size_t) blockIdx.x << shift;
            const uint32_t entries = (uint32_t) (range - first < ((size_t) 1 << shift) ? range - first : ((size_t) 1 << shift));
            // multiplicity of entry l (bitmaps: 0 / 1)
            auto mult = [&](uint32_t l) -> uint32_t {
                if constexpr (Counters) return (through_lds[l >> 2] >> ((l & 3u) * 8u)) & 255u;
                else return (marked[l >> 5] >> (l & 31u)) & 1u;
            };
            // f at one marked entry; returns whether it hit
            auto one = [&](uint32_t l, uint32_t m, const float (&v)[N], float &value) -> bool {
                auto r = through_eval(f, v, std::make_index_sequence<N>());
                if (!r.second.coeff(0)) return false;
                value = r.first.coeff(0);
                if constexpr (Counters) count += m;
                else atomicOr(&hit[l >> 5], 1u << (l & 31u));
                return true;
            };
            auto scalar_entry = [&](uint32_t l) {
                const uint32_t m = mult(l);
                float value = fill_value;
                bool write = Fill;
                if (m) {
                    float in[N];
#pragma unroll
                    for (size_t s = 0; s < N; ++s) in[s] = src.ptr[s][first + l];
                    write |= one(l, m, in, value);
                }
                if (write) target[first + l] = value;
            };
            if (vec_ok) {
                // four groups of 4 consecutive entries per lane and step; the loads of step i + 1 are requested before
                // the arithmetic of step i starts (two register sets), so f runs in the shadow of the memory system
                constexpr int V = 4;
                constexpr uint32_t kStep = V * 4096;
                struct Set {
                    float4v v[V][N];
                    uint32_t lv[V], bits[V];          // Counters: the four multiplicity bytes;  bitmaps: four bits
                };
                auto load_set = [&](uint32_t base, Set &q) {
#pragma unroll
                    for (int h = 0; h < V; ++h) {
                        q.lv[h] = base + h * 4096 + threadIdx.x * 4;
                        q.bits[h] = 0u;
                        if (q.lv[h] + 4 <= entries) {
                            if constexpr (Counters) q.bits[h] = through_lds[q.lv[h] >> 2];
                            else q.bits[h] = (marked[q.lv[h] >> 5] >> (q.lv[h] & 31u)) & 15u;
                        }
                        if (q.bits[h]) {
#pragma unroll
                            for (size_t s = 0; s < N; ++s)
                                q.v[h][s] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(src.ptr[s] + first + q.lv[h]));
                        }
                    }
                };
                auto compute_set = [&](Set &q) {
#pragma unroll
                    for (int h = 0; h < V; ++h) {
                        if (q.lv[h] + 4 > entries) continue;
                        float4v out = { fill_value, fill_value, fill_value, fill_value };
                        if (q.bits[h]) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t m = Counters ? (q.bits[h] >> (8 * j)) & 255u : (q.bits[h] >> j) & 1u;
                                if (!m) continue;
                                float in[N], value;
#pragma unroll
                                for (size_t s = 0; s < N; ++s) in[s] = q.v[h][s][j];
                                if (one(q.lv[h] + j, m, in, value)) {
                                    if constexpr (Fill) out[j] = value;
                                    else target[first + q.lv[h] + j] = value;
                                }
                            }
                        }
                        if constexpr (Fill) __builtin_nontemporal_store(out, reinterpret_cast<float4v *>(target + first + q.lv[h]));
                    }
                };
                Set qa, qb;
                if (entries) load_set(0, qa);
                for (uint32_t base = 0; base < entries; base += 2 * kStep) {
                    const bool second = base + kStep < entries;
                    if (second) load_set(base + kStep, qb);
                    compute_set(qa);
                    if (second) {
                        if (base + 2 * kStep < entries) load_set(base + 2 * kStep, qa);
                        compute_set(qb);
                    }
                }
                // ragged end of the range (fewer than 4 entries left for a lane)
                const uint32_t tail = entries & ~3u;
                if (threadIdx.x < entries - tail) scalar_entry(tail + threadIdx.x);
            } else {
                for (uint32_t l = threadIdx.x; l < entries; l += 1024) scalar_entry(l);
            }
            if constexpr (!Counters) {
                __syncthreads();
                // (c) count(hit & mask): the active elements whose entry was hit
                for_each_local([&](uint32_t l) { count += (hit[l >> 5] >> (l & 31u)) & 1u; });
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) count += __shfl_down(count, d, 64);
        if ((threadIdx.x & 63) == 0) wave_count[threadIdx.x >> 6] = count;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long total = 0;
            for (int w = 0; w < 16; ++w) total += wave_count[w];
            if (total) atomicAdd(&state->total, total);
            if (abandon) atomicOr(&state->overflow, 1u);
            __threadfence();
            if (atomicAdd(&state->done, 1u) == gridDim.x - 1) {
                // last one out: publish, and leave the state as the next call expects it
                __threadfence();
                host[0] = atomicExch(&state->total, 0ull);
                host[1] = atomicExch(&state->overflow, 0u);
                atomicExch(&state->done, 0u);
                __threadfence_system();
            }
        }
    }

    template <bool Fill, typename Func, typename... Sources>
    size_t vectorize_through_impl(Func f, const HIPArray<uint32_t> &index, const HIPArray<bool> &mask, float fill_value,
                                  HIPArray<float> &target, const Sources &... sources) {
        static_assert((std::is_same_v<Sources, HIPArray<float>> && ...), "vectorize_through(): float32 source arrays expected");
        constexpr size_t N = sizeof...(Sources);
        static_assert(N >= 1, "vectorize_through(): at least one source array");
        const char *what = "vectorize_through";
        const size_t n = index.size();
        size_t sizes[N] = { sources.size()... };
        const size_t range = Fill ? sizes[0] : target.size();
        for (size_t s = 0; s < N; ++s)
            if (sizes[s] != range)
                throw std::runtime_error("vectorize_through(): the sources and the target must have the same length");
        if (mask.size() != n && mask.size() != 1)
            throw std::runtime_error("vectorize_through(): the mask and the index array must have the same length");
        if (Fill) {
            if (range == 0) { target = HIPArray<float>(); return 0; }
            if (n == 0) { target = HIPArray<float>::full_(fill_value, range); return 0; }
            // a target of the right length is overwritten in place (a mapped image, say); anything else is replaced
            if (target.size() == range) target.make_unique();
            else target = HIPArray<float>::empty_(range);
        } else {
            if (n == 0 || range == 0) return 0;
            target.make_unique();
        }
        ek_operand om = mask.operand();
        ek_hip_index_partition *part = nullptr;
        hip_check(ek_hip_index_partition_create(HIPArray<uint32_t>::Type, index.data(), &om, n, range, &part), what);
        ek_hip_index_partition_info info;
        ek_hip_index_partition_get(part, &info);
        float *out = target.data();
        ThroughSources<N> src{ { sources.data()... } };
        // the kernel reads the sources while it writes the target (and may run twice: the byte-counter overflow retry): in-place
        // use would apply f to entries that the first attempt already replaced
        for (size_t s = 0; s < N; ++s)
            if ((const void *) src.ptr[s] == (const void *) out) {
                ek_hip_index_partition_destroy(part);
                throw std::runtime_error(std::string(what) + ": the target must not be one of the sources");
            }
        int vec_ok = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
        for (size_t s = 0; s < N; ++s) vec_ok = vec_ok && (reinterpret_cast<uintptr_t>(src.ptr[s]) & 15u) == 0;

        ThroughScratch &scratch = through_scratch();
        std::lock_guard<std::mutex> guard(scratch.mutex);
        int rc = 0;
        if (!scratch.state) {
            void *state = nullptr, *host = nullptr;
            rc = ek_hip_malloc(sizeof(ThroughState), &state);
            if (!rc) rc = ek_hip_memset(state, 0, sizeof(ThroughState));
            if (!rc) rc = ek_hip_host_malloc(2 * sizeof(unsigned long long), &host);
            if (rc) {
                if (state) ek_hip_free(state);
                ek_hip_index_partition_destroy(part);
                hip_check(rc, what);
            }
            scratch.state = (ThroughState *) state;
            scratch.host = (volatile unsigned long long *) host;
        }
        hipStream_t stream = (hipStream_t) ek_hip_stream();
        unsigned long long hits = 0;
        // byte counters where the bucket's entries fit the LDS as bytes; bitmaps otherwise, and after a counter overflowed
        for (int attempt = info.shift <= 17 ? 0 : 1; attempt < 2; ++attempt) {
            const bool counters = attempt == 0;
            const size_t lds = counters ? (size_t) 1 << info.shift : (size_t) 2 << (info.shift - 3);
            auto kernel = counters ? k_vectorize_through<Func, N, Fill, true> : k_vectorize_through<Func, N, Fill, false>;
            if (lds > 65536)
                (void) hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            hipLaunchKernelGGL(kernel, dim3((unsigned) info.n_buckets), dim3(1024), lds, stream, f, out, src, range, info.shift, vec_ok,
                               fill_value, info.bucket_base, info.local, scratch.state, scratch.host,
                               ThroughPages{ info.page_shift, info.pages_full, info.pages_part, info.part_base });
            // algorithmic bytes: the bucket lists once (bitmaps: twice), the sources once, the target once
            rc = ek_hip_note_launch("vectorize_through", n, (counters ? 4 : 8) * n + (N + 1) * 4 * range);
            if (!rc) rc = ek_hip_sync();
            if (rc) break;
            hits = scratch.host[0];
            if (!scratch.host[1]) break;          // no counter overflowed
        }
        ek_hip_index_partition_destroy(part);
        hip_check(rc, what);
        return (size_t) hits;
    }
}

/// Returns count(hit & mask).  `target` (entries nobody hits keep their contents) and every source have the same length, the
/// range that `index` points into.  Synchronises once (the count is read back), like count() itself.
template <typename Func, typename... Sources>
size_t vectorize_through(Func f, const HIPArray<uint32_t> &index, const HIPArray<bool> &mask, HIPArray<float> &target,
                         const Sources &... sources) {
    return detail::vectorize_through_impl<false>(f, index, mask, 0.f, target, sources...);
}

/// `target = full(fill_value, range); vectorize_through(f, index, mask, target, sources...)` in one pass: the target is
/// (re)allocated with the length of the sources and EVERY entry is written.
template <typename Func, typename... Sources>
size_t vectorize_through_fill(Func f, const HIPArray<uint32_t> &index, const HIPArray<bool> &mask, float fill_value,
                              HIPArray<float> &target, const Sources &... sources) {
    return detail::vectorize_through_impl<true>(f, index, mask, fill_value, target, sources...);
}

} // namespace enoki
