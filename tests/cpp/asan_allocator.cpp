// Host side of the runtime (caching allocator, graph pools, stream / device bookkeeping: enoki_amd/csrc/runtime.cpp) under
// AddressSanitizer + UBSan: runtime.cpp is plain host C++, so it is compiled HERE with g++ -fsanitize=address,undefined
// (one translation unit with this driver) and linked against the HIP runtime only.  Run on a GPU box:
//     ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0 tests/cpp/asan_allocator.bin
// Exercises: size classes and reuse, out-of-range frees, trim, whos, memcpy round trips, step-graph pools (blocks stay
// reserved until the graph is destroyed, eager allocations never receive them), refusal of host reads while capturing,
// device re-initialisation rules.
#include "../../enoki_amd/csrc/runtime.cpp"
namespace ek { void release_meta_ring() { } }      // (defined next to the kernels in csrc/bucketed.hip, which this checker does not link)

#include <cassert>
#include <cstdio>
#include <random>
#include <set>
#include <vector>

#define CHECK(expr) do { if (!(expr)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #expr, ek_hip_last_error()); return 1; } } while (0)

int main() {
    CHECK(ek_hip_init(-1) == EK_OK);
    std::mt19937 rng(7);
    // ---- random malloc / free traffic with content checks ----
    struct Block { void *p; size_t bytes; uint8_t tag; };
    std::vector<Block> live;
    std::vector<uint8_t> host(1 << 22), back(1 << 22);
    for (int it = 0; it < 4000; ++it) {
        if (live.size() < 64 && (rng() % 3) != 0) {
            size_t bytes = (size_t) 1 << (rng() % 22);
            bytes += rng() % (bytes / 2 + 1);
            void *p = nullptr;
            CHECK(ek_hip_malloc(bytes, &p) == EK_OK && p != nullptr);
            uint8_t tag = (uint8_t) (rng() & 0xff);
            CHECK(ek_hip_memset(p, tag, bytes) == EK_OK);
            live.push_back({ p, bytes, tag });
        } else if (!live.empty()) {
            size_t k = rng() % live.size();
            Block b = live[k];
            size_t probe = std::min(b.bytes, back.size());
            CHECK(ek_hip_memcpy_to_host(back.data(), b.p, probe) == EK_OK);
            for (size_t i = 0; i < probe; i += 997) CHECK(back[i] == b.tag);       // nobody else wrote into a live block
            CHECK(ek_hip_free(b.p) == EK_OK);
            live[k] = live.back();
            live.pop_back();
        }
    }
    for (Block &b : live) CHECK(ek_hip_free(b.p) == EK_OK);
    live.clear();
    int bogus;
    CHECK(ek_hip_free(&bogus) == EK_ERR_INVALID);                                  // not ours: reported, not crashed
    CHECK(ek_hip_free(nullptr) == EK_OK);
    char *w = ek_hip_whos();
    CHECK(w != nullptr);
    free(w);
    CHECK(ek_hip_malloc_trim() == EK_OK);

    // ---- step-graph pools ----
    void *warm = nullptr;
    CHECK(ek_hip_malloc(1 << 20, &warm) == EK_OK);
    CHECK(ek_hip_free(warm) == EK_OK);                                             // now cached
    CHECK(ek_hip_graph_begin() == EK_OK);
    void *a = nullptr, *b = nullptr, *c = nullptr;
    CHECK(ek_hip_malloc(1 << 20, &a) == EK_OK);                                    // taken from the cache into the pool
    CHECK(ek_hip_memset(a, 1, 1 << 20) == EK_OK);                                  // captured
    CHECK(ek_hip_malloc(1 << 20, &b) == EK_OK);
    CHECK(ek_hip_memcpy_device(b, a, 1 << 20) == EK_OK);
    CHECK(ek_hip_free(a) == EK_OK);                                                // back to the POOL
    CHECK(ek_hip_malloc(1 << 20, &c) == EK_OK);
    CHECK(c == a);                                                                 // reused inside the capture
    CHECK(ek_hip_memset(c, 2, 1 << 20) == EK_OK);
    uint8_t probe = 0;
    CHECK(ek_hip_memcpy_to_host(&probe, b, 1) == EK_ERR_INVALID);                  // host reads are refused while capturing
    ek_hip_graph *g = nullptr;
    CHECK(ek_hip_graph_end(&g) == EK_OK && g != nullptr);
    CHECK(ek_hip_free(c) == EK_OK);                                                // stays reserved for the graph
    std::set<void *> eager;
    for (int i = 0; i < 8; ++i) {
        void *p = nullptr;
        CHECK(ek_hip_malloc(1 << 20, &p) == EK_OK);
        CHECK(p != a && p != b);                                                   // eager code never gets pool blocks
        eager.insert(p);
    }
    for (int r = 0; r < 3; ++r) CHECK(ek_hip_graph_launch(g) == EK_OK);
    CHECK(ek_hip_memcpy_to_host(&probe, b, 1) == EK_OK && probe == 1);             // b = copy of a's first contents
    for (void *p : eager) CHECK(ek_hip_free(p) == EK_OK);
    CHECK(ek_hip_graph_destroy(g) == EK_OK);
    CHECK(ek_hip_free(b) == EK_OK);                                                // an ordinary block again
    CHECK(ek_hip_graph_end(&g) == EK_ERR_INVALID);                                 // no capture in progress

    // ---- device re-initialisation ----
    void *keep = nullptr;
    CHECK(ek_hip_malloc(4096, &keep) == EK_OK);
    CHECK(ek_hip_init(ek_hip_device()) == EK_OK);                                  // same device: no-op
    if (ek_hip_device_count() > 1) CHECK(ek_hip_init(1) == EK_ERR_INVALID);         // live allocations: refused
    CHECK(ek_hip_free(keep) == EK_OK);
    CHECK(ek_hip_malloc_trim() == EK_OK);
    printf("asan_allocator: all checks passed (%llu launches noted)\n", (unsigned long long) ek_hip_launch_count());
    return 0;
}
