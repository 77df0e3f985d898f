// include/enoki/morton.h on host scalars against the DEFINITION of the code (bit b of coordinate i -> bit b * D + i, low
// floor(bits / D) bits of every coordinate), 200000 random inputs per (word size, dimension); run by tests/test_morton.py.
#include <enoki/morton.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
using namespace enoki;
template <typename S, size_t D> static int check() {
    using C = Array<S, D>;
    const size_t most = sizeof(S) * 8 / D;
    int bad = 0;
    for (int it = 0; it < 200000; ++it) {
        C c;
        S want = 0;
        for (size_t i = 0; i < D; ++i) { c.coeff(i) = (S) (((uint64_t) rand() << 33) ^ ((uint64_t) rand() << 11) ^ rand()); }
        for (size_t b = 0; b < most; ++b)
            for (size_t i = 0; i < D; ++i) want |= (S) (((c.coeff(i) >> b) & S(1)) << (b * D + i));
        S got = morton_encode(c);
        C back = morton_decode<C>(got);
        bool ok = got == want;
        const S low = most >= sizeof(S) * 8 ? S(~S(0)) : S((S(1) << most) - 1);
        for (size_t i = 0; i < D; ++i) ok = ok && back.coeff(i) == (c.coeff(i) & low);
        bad += !ok;
    }
    printf("%zu-bit, %zu-D: %d mismatches\n", sizeof(S) * 8, D, bad);
    return bad;
}
int main() { return check<uint32_t, 2>() + check<uint32_t, 3>() + check<uint64_t, 2>() + check<uint64_t, 3>() + check<uint32_t, 4>() + check<uint64_t, 1>(); }
