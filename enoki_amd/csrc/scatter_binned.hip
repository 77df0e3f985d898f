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
#include "ek_binned.h"

namespace ek {


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
    if constexpr (std::is_same_v<T, float>) {
        // one pass over (index, value) instead of count + scans + partition (valid int32 indices are non-negative: same bits as uint32)
        static const bool paged = [] { const char *e = getenv("ENOKI_HIP_SCATTER_PAGED"); return !e || atoi(e) != 0; }();
        if (paged && value.vec && index.vec && scatter_add_paged_applicable(table_size, n)) {
            // (what the page partition cannot take -- a shape its plan refuses, no memory for its ~6 n bytes of lists --
            //  is not a failure of scatter_add: the count / scan / partition path below handled every such call before)
            const int rc = scatter_add_paged(base, table_size, value.ptr, reinterpret_cast<const uint32_t *>(index.ptr), mask, n);
            if (rc != EK_ERR_UNSUPPORTED && rc != EK_ERR_OOM) return rc;
            (void) hipGetLastError();
        }
    }
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
    for (int s = 0; s < C; ++s) { targets.table[s] = bases[s]; targets.scale[s] = T(1); }
    hipLaunchKernelGGL((k_bin_fold_pieces<T, C>), dim3(fold_grid(table_size), C), dim3(256), 0, c.stream, targets,
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
