// Raw-memory gather / scatter of packed records (array.h gather_packed / scatter_packed; the reference's
// gather<Array<Packet, N>>(mem, index, mask), array_router.h:1097-1107): component k of record i is mem[index[i] * N + k].
// Host packets of 1 and 4 lanes, 2- and 3-component records, masked and unmasked; clang builds additionally take the
// one-instruction path for 8- and 16-byte records (tests/test_sphere_gpu.py checks that one on the device).
//
//     g++ -O1 -std=c++17 -Iinclude tests/cpp/packed_host.cpp -o tests/cpp/packed_host.bin
#include <enoki/array.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace enoki;

#define CHECK(expr) do { if (!(expr)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #expr); exit(1); } } while (0)

template <size_t Lanes, size_t N> static void run() {
    using FloatP = Array<float, Lanes>;
    using UIntP = Array<uint32_t, Lanes>;
    using MaskP = mask_t<FloatP>;
    using Record = Array<FloatP, N>;
    const size_t count = 37;
    std::vector<float> table(count * N + 1);
    for (size_t i = 0; i < table.size(); ++i) table[i] = 0.5f * (float) i - 3.f;

    for (int misalign = 0; misalign < 2; ++misalign) {                  // an odd float offset defeats the aligned path
        std::vector<float> store(table.size() + 1);
        float *mem = store.data() + misalign;
        for (size_t i = 0; i + 1 < table.size(); ++i) mem[i] = table[i];
        UIntP idx; MaskP mask;
        for (size_t l = 0; l < Lanes; ++l) { idx.coeff(l) = (uint32_t) ((7 * l + 5) % count); mask.coeff(l) = (l % 3) != 1; }

        Record all = gather<Record>(mem, idx);
        Record some = gather<Record>(mem, idx, mask);
        for (size_t l = 0; l < Lanes; ++l)
            for (size_t k = 0; k < N; ++k) {
                CHECK(all.coeff(k).coeff(l) == mem[idx.coeff(l) * N + k]);
                CHECK(some.coeff(k).coeff(l) == (mask.coeff(l) ? mem[idx.coeff(l) * N + k] : 0.f));
            }

        Record v;
        for (size_t l = 0; l < Lanes; ++l)
            for (size_t k = 0; k < N; ++k) v.coeff(k).coeff(l) = 100.f + (float) (l * N + k);
        std::vector<float> before(mem, mem + count * N);
        scatter(mem, v, idx, mask);
        for (size_t l = 0; l < Lanes; ++l)
            for (size_t k = 0; k < N; ++k)
                before[idx.coeff(l) * N + k] = mask.coeff(l) ? v.coeff(k).coeff(l) : before[idx.coeff(l) * N + k];
        for (size_t i = 0; i < count * N; ++i) CHECK(mem[i] == before[i]);
    }
}

int main() {
    run<1, 2>(); run<1, 3>(); run<1, 4>(); run<4, 2>(); run<4, 3>(); run<4, 4>();
    printf("packed_host: gather / scatter of 2-, 3- and 4-component records on 1- and 4-lane packets, aligned and not, masked and not\n");
    return 0;
}
