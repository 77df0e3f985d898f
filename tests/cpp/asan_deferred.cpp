// Lifetime and value semantics of HIPArray's deferred nodes (deferred gathers, deferred unary maps, sincos pairs:
// include/enoki/hip.h) under AddressSanitizer + LeakSanitizer + UBSan, WITHOUT a GPU: the C ABI below is a host stand-in
// (malloc'd "device" memory, scalar loops) that implements just the entry points hip.h reaches from this file, so that the
// header's own logic -- reference counts, reader lists, partner links, copy-on-write, forcing before a source changes --
// runs for real.  This is a CHECKER of the binding's host logic; numerics are whatever libm gives (both sides of every
// comparison go through the same stand-in).
//
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iinclude tests/cpp/asan_deferred.cpp -o tests/cpp/asan_deferred.bin
//
// Part 1: directed scenarios.  Part 2: a fuzzer -- random programs over a pool of arrays are executed twice, once with
// deferred evaluation and once without; every array that is looked at must hold the same bits, and the stand-in's
// allocation count must return to zero.
#include <enoki/hip.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include "host_abi_stub.h"


// ------------------------------------------------------------------------------------------------------------------
using namespace enoki;
using F = HIPArray<float>;
using U = HIPArray<uint32_t>;
using M = HIPArray<bool>;

#define CHECK(expr) do { if (!(expr)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #expr); exit(1); } } while (0)

static std::vector<float> host(const F &a) {
    std::vector<float> v(a.size());
    for (size_t i = 0; i < v.size(); ++i) v[i] = a.coeff(i);
    return v;
}
/* Bit for bit, except that any NaN equals any NaN: a negation the binding folds into a map's scale
   multiplies by -1 (which hands a NaN through with its sign) where the eager kernel flips the sign bit.
   IEEE 754 leaves the sign of a NaN result open and so does the reference (PTX neg.f32 of a NaN). */
static bool same(const std::vector<float> &a, const std::vector<float> &b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (memcmp(&a[i], &b[i], 4) != 0 && !(a[i] != a[i] && b[i] != b[i])) return false;
    return true;
}

static constexpr size_t N = 1 << 16;       // the smallest size the binding defers unary maps for

static F input(size_t n, float scale) {
    F x = linspace<F>(-3.f, 3.f, n) * F(scale);
    (void) x.data();
    return x;
}

static void directed() {
    F x = input(N, 1.f);
    std::vector<float> hx = host(x), hs(N), hc(N);
    for (size_t i = 0; i < N; ++i) { hs[i] = std::sin(hx[i]); hc[i] = std::cos(hx[i]); }
    U idx = arange<U>(N);

    // --- a deferred map is an ordinary array to everybody who looks at it ---
    long u0 = g_unary_calls;
    F s = sin(x);
    CHECK(g_unary_calls == u0);                       // nothing ran yet
    F s2 = s;                                         // second handle on the unevaluated buffer
    CHECK(same(host(s2), hs) && same(host(s), hs));
    CHECK(g_unary_calls == u0 + 1);                   // evaluated once for both handles

    // --- consumed on load: no kernel, and the node stays usable ---
    s = sin(x);
    u0 = g_unary_calls;
    float total = hsum(s).coeff(0);
    CHECK(g_unary_calls == u0);
    float expect = 0.f;
    for (float v : hs) expect += v;
    CHECK(total == expect && same(host(s), hs));

    // --- the source changes, goes away, or is overwritten through another handle ---
    {
        F t = input(N, 1.f);
        F st = sin(t);
        scatter(t, F(0.f), idx);                      // in-place write into the source: the map runs first
        CHECK(same(host(st), hs) && t.coeff(5) == 0.f);
    }
    {
        F st;
        { F t = input(N, 1.f); st = sin(t); }         // the only handle on the source dies: the node keeps the buffer
        CHECK(same(host(st), hs));
    }
    {
        F t = input(N, 1.f), alias = t;
        F st = sin(t);
        scatter(alias, F(1.f), idx);                  // copy-on-write of a shared source
        CHECK(same(host(st), hs) && same(host(t), hx));
    }
    {
        F st = sin(input(N, 1.f));                    // never looked at: node and source are released with the handle
    }

    // --- the factors of the derivatives of sqrt / rcp / rsqrt stay ONE unevaluated map of the source each (autodiff.h:353-403) ---
    {
        F v = abs(input(N, 3.f)) + F(1.f);
        std::vector<float> hv = host(v), want(N);
        u0 = g_unary_calls;
        F r = sqrt(v), w = F(0.5f) / r;               // .5 / sqrt(v): an unevaluated multiple of rsqrt(v)
        F q = rcp(v), wq = -sqr(q);                   // -(1 / v)^2
        F t = rsqrt(v), t2 = sqr(t), w3 = F(-0.5f) * (t * t2);
        CHECK(g_unary_calls == u0);                   // nothing ran
        for (size_t i = 0; i < N; ++i) want[i] = 0.5f / std::sqrt(hv[i]);
        CHECK(same(host(w), want));
        for (size_t i = 0; i < N; ++i) { volatile float a = 1.0f / hv[i]; want[i] = -(a * a); }
        CHECK(same(host(wq), want));
        for (size_t i = 0; i < N; ++i) { volatile float a = 1.0f / std::sqrt(hv[i]), a2 = a * a; want[i] = -0.5f * (a * a2); }
        CHECK(same(host(w3), want));
        CHECK(g_unary_calls == u0 + 3);               // ONE kernel per weight: neither sqrt / rcp / rsqrt nor the products ran
    }

    // --- sincos pairs: every order of touching / dropping the halves ---
    for (int order = 0; order < 6; ++order) {
        long c0 = g_sincos_calls, un0 = g_unary_calls;
        auto [a, b] = sincos(x);
        CHECK(g_sincos_calls == c0);
        switch (order) {
            case 0: CHECK(same(host(a), hs) && same(host(b), hc)); CHECK(g_sincos_calls == c0 + 1 && g_unary_calls == un0); break;
            case 1: CHECK(same(host(b), hc) && same(host(a), hs)); CHECK(g_sincos_calls == c0 + 1 && g_unary_calls == un0); break;
            case 2: a = F(); CHECK(same(host(b), hc)); CHECK(g_sincos_calls == c0 && g_unary_calls == un0 + 1); break;
            case 3: b = F(); CHECK(same(host(a), hs)); CHECK(g_sincos_calls == c0 && g_unary_calls == un0 + 1); break;
            case 4: { float t2 = hsum(a).coeff(0); CHECK(t2 == expect); CHECK(same(host(b), hc) && same(host(a), hs)); break; }
            case 5: { F keep = a; a = F(); CHECK(same(host(b), hc) && same(host(keep), hs)); CHECK(g_sincos_calls == c0 + 1); break; }
        }
    }

    // --- maps and gathers on top of each other ---
    {
        F table = input(1024, 1.f);
        U gi = arange<U>(N) & U(1023u);
        F e = exp(gather<F>(table, gi));              // map of a deferred gather: the gather runs, the map stays deferred
        F g = gather<F>(exp(table), gi);              // gather from a small (eagerly evaluated) map
        CHECK(same(host(e), host(g)));
        F big = input(N, 1.f);
        F gg = gather<F>(sin(big), idx);              // gather whose table is a deferred map
        CHECK(same(host(gg), hs));
        F fused = fmadd(gather<F>(table, gi), x, gather<F>(table, gi));
        std::vector<float> ht = host(table), hf = host(fused);
        for (size_t i = 0; i < N; i += 997) CHECK(hf[i] == std::fma(ht[i & 1023], hx[i], ht[i & 1023]));
    }

    // --- a deferred gather reads its INDEX array: a write through data() / scatter into the indices comes after the gather ---
    {
        F table = input(4096, 1.f);
        U gi = arange<U>(N) & U(4095u);
        F g = gather<F>(table, gi);
        scatter(gi, U(0u), arange<U>(N));             // in-place write into the index array
        std::vector<float> ht = host(table), hg = host(g);
        for (size_t i = 0; i < N; i += 499) CHECK(hg[i] == ht[i & 4095]);
        U gj = arange<U>(N) & U(4095u);
        F g2 = gather<F>(table, gj);
        uint32_t *raw = gj.data();                    // mutable pointer: the pending gather runs first
        (void) raw;
        CHECK(!g2.deferred_());
    }

    // --- the list of unevaluated buffers (hip_graph_begin() evaluates them all before a capture starts) ---
    {
        F table = input(4096, 1.f);
        U gi = arange<U>(N) & U(4095u);
        F a = sin(x), b = gather<F>(table, gi);
        { F dropped = cos(x); }                               // leaves the list when it dies
        auto sc = sincos(x);
        CHECK(a.mapped_() && b.deferred_() && sc.first.mapped_() && sc.second.mapped_());
        enoki::detail::HIPBuffer::force_all_pending();
        CHECK(!a.mapped_() && !b.deferred_() && !sc.first.mapped_() && !sc.second.mapped_());
        CHECK(enoki::detail::HIPBuffer::pending_head() == nullptr);
        CHECK(same(host(a), hs) && same(host(sc.second), hc));
    }

    // --- zeros stay unevaluated until somebody looks (kind 3) ---
    {
        F z = zero<F>(N);
        CHECK(z.zeroed_());
        F z2 = z;                                      // a second handle: no longer a candidate for "write instead of add"
        CHECK(z.coeff(17) == 0.f && !z.zeroed_() && !z2.zeroed_());
        F w = zero<F>(N) + x;                          // consumed by an ordinary op: memset first
        CHECK(same(host(w), hx));
        F t = zero<F>(N);
        scatter_add(t, x, idx);                        // target of an element-order scatter_add
        CHECK(same(host(t), hx));
        { F dead = zero<F>(N); }
    }

    // --- fma over a gathered pair (kind 2): bucket-ordered consumers, and every way of asking for element order ---
    {
        const size_t K = 4096;
        F A = input(K, 1.f), C = input(K, 0.5f);
        U gi = (arange<U>(N) * U(2654435761u)) & U((uint32_t) K - 1u);
        std::vector<float> hA = host(A), hC = host(C);
        std::vector<uint32_t> hi(N);
        for (size_t i = 0; i < N; ++i) hi[i] = (uint32_t) ((i * 2654435761ull) & (K - 1));
        std::vector<float> hu(N);
        for (size_t i = 0; i < N; ++i) hu[i] = std::fma(hA[hi[i]], hx[i], hC[hi[i]]);
        auto make_u = [&]() { return fmadd(gather<F>(A, gi), x, gather<F>(C, gi)); };

        long f0 = g_fused_calls, r0 = g_bucketed_reduces, s0 = g_bucketed_scatters;
        F u = make_u();
        CHECK(u.paired_() && g_fused_calls == f0);                     // nothing ran
        // (1) the forward of BASELINE config 3b: sincos(u), hsum(sin) in bucket order; u stays unevaluated with its partition
        auto [su, cu] = sincos(u);
        CHECK(su.mapped_() && cu.mapped_() && u.paired_());
        float y = hsum(su).coeff(0), ye = 0.f;
        for (size_t i = 0; i < N; ++i) ye += std::sin(hu[i]);
        CHECK(y == ye && g_bucketed_reduces == r0 + 1 && g_fused_calls == f0 && u.paired_() && g_bucketed_live == 1);
        // (2) the adjoint: cos(u) and x * cos(u) through the same index array reuse the partition
        F ga = zero<F>(K), gc = zero<F>(K);
        F *targets[2] = { &gc, &ga };
        const F *values[2] = { &cu, &cu }, *weights[2] = { nullptr, &x };
        long fr0 = g_fresh_targets;
        CHECK(ga.zeroed_() && gc.zeroed_());
        F::scatter_add_multi_(2, targets, values, weights, gi, M(true));
        CHECK(g_bucketed_scatters == s0 + 1 && g_fused_calls == f0 && cu.mapped_());
        CHECK(g_fresh_targets == fr0 + 2 && !ga.zeroed_());            // both gradient buffers were written, not memset + added to
        {   // a zeros target that somebody else also holds is an ordinary (memset) target
            F shared = zero<F>(K), alias = shared, other2 = zero<F>(K);
            F *t2[2] = { &other2, &shared };
            long fr1 = g_fresh_targets;
            F::scatter_add_multi_(2, t2, values, weights, gi, M(true));
            CHECK(g_fresh_targets == fr1 + 1 && alias.coeff(0) == 0.f);
            CHECK(same(host(shared), host(ga)) && same(host(other2), host(gc)));
        }
        std::vector<float> ea(K, 0.f), ec(K, 0.f);
        for (size_t i = 0; i < N; ++i) {
            float c = std::cos(hu[i]);
            ec[hi[i]] += c;
            ea[hi[i]] += (hx[i] == 0 || c == 0) ? 0.f : hx[i] * c;
        }
        CHECK(same(host(ga), ea) && same(host(gc), ec));
        // (3) a second, element-order consumer: u is evaluated once by the fused gather kernel, the partition goes away, and
        // the maps that were built on the unevaluated u read the evaluated one
        F twice = u + u;
        CHECK(!u.paired_() && g_fused_calls == f0 + 1 && g_bucketed_live == 0);
        CHECK(same(host(u), hu) && twice.coeff(77) == hu[77] + hu[77]);
        std::vector<float> hcu(N);
        for (size_t i = 0; i < N; ++i) hcu[i] = std::cos(hu[i]);
        CHECK(same(host(cu), hcu));
        // (4) forced through data()
        F u2 = make_u();
        CHECK(u2.paired_());
        CHECK(hsum(u2).coeff(0) != 0.f && u2.paired_());                // reduced in bucket order, still unevaluated
        const float *raw = ((const F &) u2).data();
        CHECK(raw && !u2.paired_() && g_bucketed_live == 0 && same(host(u2), hu));
        // (5) a table changes while u is pending: u is evaluated with the OLD contents first
        F A2 = input(K, 1.f);
        F u3 = fmadd(gather<F>(A2, gi), x, gather<F>(C, gi));
        float m3 = hmax(u3).coeff(0);
        scatter(A2, F(100.f), arange<U>(K));
        CHECK(!u3.paired_() && same(host(u3), hu) && g_bucketed_live == 0);
        float em = hu[0];
        for (float v : hu) em = std::fmax(em, v);
        CHECK(m3 == em);
        // (6) ... and so does x, and the index array
        F x2 = input(N, 1.f);
        U gi2 = (arange<U>(N) * U(2654435761u)) & U((uint32_t) K - 1u);
        F u4 = fmadd(gather<F>(A, gi2), x2, gather<F>(C, gi2));
        (void) hsum(sin(u4));
        scatter(x2, F(0.f), idx);
        CHECK(!u4.paired_() && same(host(u4), hu));
        F u5 = fmadd(gather<F>(A, gi2), x, gather<F>(C, gi2));
        scatter(gi2, U(0u), idx);
        CHECK(!u5.paired_() && same(host(u5), hu));
        // (7) the adjoint with a DIFFERENT weight array, or through a different index array: element-order pipeline, same sums
        F u6 = make_u();
        F c6 = cos(u6);
        F other = input(N, 1.f);                                      // same values as x, different buffer
        F g6 = zero<F>(K), g7 = zero<F>(K);
        F *t6[2] = { &g6, &g7 };
        const F *v6[2] = { &c6, &c6 }, *w6[2] = { nullptr, &other };
        long s1 = g_bucketed_scatters;
        F::scatter_add_multi_(2, t6, v6, w6, gi, M(true));
        CHECK(g_bucketed_scatters == s1 && !u6.paired_());
        CHECK(same(host(g7), ea) && same(host(g6), ec));
        // (8) never consumed: released with the handle
        { F dead = make_u(); F sd = sin(dead); }
        // (9) the four ops of the family and a target that aliases a table
        F un = fnmsub(gather<F>(A, gi), x, gather<F>(C, gi));
        CHECK(un.paired_());
        float sn = hsum(un).coeff(0), esn = 0.f;
        for (size_t i = 0; i < N; ++i) esn += std::fma(-hA[hi[i]], hx[i], -hC[hi[i]]);
        CHECK(sn == esn);
        F A3 = input(K, 1.f);
        F u7 = fmadd(gather<F>(A3, gi), x, gather<F>(C, gi));
        F c7 = cos(u7);
        F *t7[1] = { &A3 };
        const F *v7[1] = { &c7 };
        F::scatter_add_multi_(1, t7, v7, nullptr, gi, M(true));       // the target IS table A of the pending u
        std::vector<float> hA3 = host(A3), eA3 = hA;
        for (size_t i = 0; i < N; ++i) eA3[hi[i]] += std::cos(hu[i]);     // u7 was evaluated with the OLD table first
        CHECK(same(hA3, eA3) && !u7.paired_());
    }
    CHECK(g_bucketed_live == 0);

    // --- scatter_add_multi_ with mapped values; one target IS the map's source ---
    {
        F u = input(N, 1.f);
        auto [su, cu] = sincos(u);
        F ta = zero<F>(N), tb = zero<F>(N);
        F *targets[2] = { &ta, &tb };
        const F *values[2] = { &cu, &cu }, *weights[2] = { &x, nullptr };
        long f0 = g_unary_calls + g_sincos_calls;
        F::scatter_add_multi_(2, targets, values, weights, idx, M(true));
        CHECK(g_unary_calls + g_sincos_calls == f0);  // cos applied on load
        std::vector<float> ha = host(ta), hb = host(tb);
        for (size_t i = 0; i < N; i += 101) CHECK(hb[i] == hc[i] && ha[i] == ((hx[i] == 0 || hc[i] == 0) ? 0.f : hx[i] * hc[i]));
        CHECK(same(host(su), hs));
        // the mapped value is also a weight: evaluated once, used as both
        {
            F cw = cos(input(N, 1.f));
            F t0 = zero<F>(N), t1 = zero<F>(N);
            F *t3[2] = { &t0, &t1 };
            const F *v3[2] = { &cw, &cw }, *w3[2] = { &cw, nullptr };
            F::scatter_add_multi_(2, t3, v3, w3, idx, M(true));
            std::vector<float> h0 = host(t0), h1 = host(t1);
            for (size_t i = 0; i < N; i += 101) CHECK(h1[i] == hc[i] && h0[i] == ((hc[i] == 0) ? 0.f : hc[i] * hc[i]));
        }
        // target == source of the mapped value: evaluated first, then accumulated
        F v = input(N, 1.f);
        F cv = cos(v);
        F other = zero<F>(N);
        F *t2[2] = { &v, &other };
        const F *v2[2] = { &cv, &cv };
        F::scatter_add_multi_(2, t2, v2, nullptr, idx, M(true));
        std::vector<float> hv = host(v), ho = host(other);
        for (size_t i = 0; i < N; i += 101) CHECK(hv[i] == hx[i] + hc[i] && ho[i] == hc[i]);
    }

    // --- BASELINE configs[1] as a C++ caller writes it, ONE expression: the temporaries between the calls let go of their handles
    //     as soon as the next function has its result (array.h, expiring arguments), so the reduction absorbs the whole chain ---
    {
        F a = input(N, 1.f), b = input(N, 0.5f), xx = input(N, 0.25f);
        const long c0 = g_chain_calls, l0 = g_array_launches, u1 = g_unary_calls, f0 = g_fused_calls;
        float y = hsum(sin(exp(fmadd(a, xx, b)))).coeff(0);
        CHECK(g_chain_calls == c0 + 1 && g_array_launches == l0 && g_unary_calls == u1 && g_fused_calls == f0);
        float y2 = hsum(sin(exp(a * xx + b))).coeff(0);                 // the operator spelling: a product and a sum, still one pass
        CHECK(g_chain_calls == c0 + 2 && g_array_launches == l0 && g_unary_calls == u1);
        std::vector<float> ha = host(a), hb = host(b), hxx = host(xx);
        float e1 = 0.f, e2 = 0.f;
        for (size_t i = 0; i < N; ++i) {
            e1 += std::sin(std::exp(std::fma(ha[i], hxx[i], hb[i])));
            volatile float p = ha[i] * hxx[i];
            e2 += std::sin(std::exp(p + hb[i]));
        }
        CHECK(y == e1 && y2 == e2);
        // a NAMED intermediate is not expiring: it is evaluated once and keeps its value
        F u = fmadd(a, xx, b);
        float y3 = hsum(sin(exp(u))).coeff(0);
        CHECK(y3 == e1 && u.valid() && u.coeff(7) == std::fma(ha[7], hxx[7], hb[7]));
    }
}

// ------------------------------------------------------------------------------------------------------------------
//  Fuzzer: the same random program with and without deferred evaluation
// ------------------------------------------------------------------------------------------------------------------
static std::vector<std::vector<float>> run_program(uint32_t seed, bool defer) {
    hip_set_defer(defer);
    std::mt19937 rng(seed);
    std::vector<std::vector<float>> seen;
    {
        const size_t n = N, K = 4096;
        std::vector<F> pool;
        for (int i = 0; i < 4; ++i) pool.push_back(input(n, 0.25f + 0.5f * (float) i));
        std::vector<F> tables;
        for (int i = 0; i < 2; ++i) tables.push_back(input(K, 1.f + (float) i));
        U idx = (arange<U>(n) * U(2654435761u)) & U((uint32_t) K - 1u);
        U ident = arange<U>(n);
        static const int maps[] = { EK_NEG, EK_ABS, EK_SIN, EK_COS, EK_EXP };
        auto pick = [&]() -> F & { return pool[rng() % pool.size()]; };
        for (int step = 0; step < 60; ++step) {
            const int what = rng() % 20;
            if (getenv("TRACE")) fprintf(stderr, "seed %u step %d op %d\n", seed, step, what);
            switch (what) {
                case 0: {   // unary map
                    F &a = pick();
                    int op = maps[rng() % 5];
                    F r = op == EK_NEG ? -a : op == EK_ABS ? abs(a) : op == EK_SIN ? sin(a) : op == EK_COS ? cos(a) : exp(a * F(0.1f));
                    pick() = r;
                    break;
                }
                case 1: { auto [s, c] = sincos(pick()); pick() = s; if (rng() & 1) pick() = c; break; }
                case 2: { F &a = pick(); seen.push_back({ hsum(a).coeff(0), hmax(a).coeff(0) }); break; }
                case 3: { F r = pick() + pick(); pick() = r; break; }
                case 4: { F r = gather<F>(tables[rng() % 2], idx); pick() = r; break; }
                case 5: { F r = fmadd(gather<F>(tables[0], idx), pick(), gather<F>(tables[1], idx)); pick() = r; break; }
                case 6: { F r = gather<F>(tables[rng() % 2], idx) * pick(); pick() = r; break; }
                case 7: {   // in-place write into an array that may have deferred readers
                    F &a = pick();
                    scatter(a, F(0.5f), ident, ident < U((uint32_t) (rng() % n)));
                    break;
                }
                case 8: {   // in-place accumulation into a table that may have deferred gathers
                    F &t = tables[rng() % 2];
                    U ti = arange<U>(K);
                    scatter_add(t, F(0.125f), ti, ti < U((uint32_t) (rng() % K)));
                    break;
                }
                case 9: { seen.push_back(host(pick())); break; }
                case 12: {  // the shape of BASELINE config 3b: parameter lookup, sincos, reduction, adjoint through the same indices
                    F &xx = pick();
                    F u = fmadd(gather<F>(tables[0], idx), xx, gather<F>(tables[1], idx));
                    auto [su, cu] = sincos(u);
                    seen.push_back({ hsum(su).coeff(0) });
                    if (rng() & 1) seen.push_back({ hmin(u).coeff(0) });
                    F ta = zero<F>(K), tb = zero<F>(K);
                    F *targets[2] = { &tb, &ta };
                    const F *values[2] = { &cu, &cu }, *weights[2] = { nullptr, (rng() & 3) ? &xx : &pick() };
                    F::scatter_add_multi_(2, targets, values, weights, idx, M(true));
                    seen.push_back(host(ta));
                    seen.push_back(host(tb));
                    if (rng() & 1) pick() = u;
                    if (rng() & 1) pick() = cu;
                    break;
                }
                case 10: { pick() = pick(); break; }                     // handle copy
                // round 5: unevaluated arithmetic (kind 4) under stacks of maps, consumed as a chain or forced; the operator spellings
                case 14: { F r = pick() * pick() + pick(); if (rng() & 1) seen.push_back({ hsum(r).coeff(0) }); pick() = r; break; }
                case 15: {
                    F u = fmadd(pick(), pick(), pick());
                    F e = exp(u * F(0.125f));
                    F t = rng() & 1 ? sin(e) : abs(-e);
                    if (rng() & 1) seen.push_back({ hsum(t).coeff(0), hmin(t).coeff(0) });      // a chain reduction (or its pieces, if held)
                    if (rng() & 1) pick() = u;                                                  // u held by somebody else: evaluated once
                    pick() = t;
                    break;
                }
                case 16: { F r = gather<F>(tables[0], idx) * pick() + gather<F>(tables[1], idx); if (rng() & 1) seen.push_back({ hsum(cos(r)).coeff(0) }); pick() = r; break; }
                case 17: { F p = gather<F>(tables[1], idx) * pick(); F r = rng() & 1 ? p - gather<F>(tables[0], idx) : gather<F>(tables[0], idx) - p; pick() = r; break; }
                case 19: {  // an unevaluated map times an array: the map and the product are two outputs of one pass
                    F u = fmadd(pick(), pick(), pick());
                    F c = rng() & 1 ? cos(u) : -sin(exp(u * F(0.125f)));
                    F &w = pick();
                    F p = rng() & 1 ? c * w : w * c;
                    if (rng() & 1) seen.push_back(host(p));
                    if (rng() & 1) pick() = c;
                    pick() = p;
                    break;
                }
                case 18: { F p = pick() * gather<F>(tables[0], idx); if (rng() & 1) seen.push_back({ hsum(p).coeff(0) }); else pick() = p; break; }
                case 13: {  // sqrt and its derivative's factor .5 / sqrt(v): an unevaluated multiple of rsqrt(v) when deferred
                    F v = abs(pick()) + F(1.f);
                    F r = sqrt(v);
                    F w = F(0.5f) / r, w2 = F(3.f) / r;                  // (3 is no power of two: an ordinary division)
                    seen.push_back({ hsum(r).coeff(0), hsum(w).coeff(0) });
                    seen.push_back(host(w));
                    seen.push_back(host(w2));
                    if (rng() & 1) pick() = w * F(2.f);
                    if (rng() & 1) seen.push_back(host(w * F(3.f)));          // .5 rsqrt times a seed: still a map (1.5 rsqrt)
                    if (rng() & 1) pick() = r;
                    // the weights of rcp and rsqrt (autodiff.h:381-403): products of unevaluated maps of one source
                    F q = rcp(v), wq = -sqr(q);
                    F s = rsqrt(v), s2 = sqr(s), w3 = F(-0.5f) * (s * s2);
                    seen.push_back({ hsum(q).coeff(0), hsum(s).coeff(0) });
                    seen.push_back(host(wq));
                    seen.push_back(host(w3));
                    if (rng() & 1) seen.push_back(host(s2));
                    if (rng() & 1) pick() = q * s;                       // maps of one source, but no derivative's product
                    break;
                }
                case 11: {  // adjoint-style multi scatter with (possibly) mapped values
                    F v = cos(pick());
                    F ta = zero<F>(K), tb = zero<F>(K);
                    F *targets[2] = { &ta, &tb };
                    const F *values[2] = { &v, &v }, *weights[2] = { &pick(), nullptr };
                    F::scatter_add_multi_(2, targets, values, weights, idx, M(true));
                    seen.push_back(host(ta));
                    if (rng() & 1) tables[rng() % 2] = tb;
                    break;
                }
            }
        }
        for (F &a : pool) seen.push_back(host(a));
        for (F &t : tables) seen.push_back(host(t));
    }
    return seen;
}

int main() {
    directed();
    CHECK(g_live.empty());
    long fused_total = 0;
    for (uint32_t seed = 1; seed <= 40; ++seed) {
        long f0 = g_fused_calls;
        auto with = run_program(seed, true);
        fused_total += g_fused_calls - f0;
        CHECK(g_live.empty() && g_bucketed_live == 0);
        f0 = g_fused_calls;
        auto without = run_program(seed, false);
        CHECK(g_fused_calls == f0);                    // switched off means off
        CHECK(g_live.empty());
        CHECK(with.size() == without.size());
        for (size_t i = 0; i < with.size(); ++i)
            if (!same(with[i], without[i])) { fprintf(stderr, "seed %u: observation %zu differs\n", seed, i); return 1; }
    }
    hip_set_defer(true);
    CHECK(fused_total > 100 && g_bucketed_reduces > 20 && g_bucketed_scatters > 10);      // the deferred paths were really taken
    CHECK(g_chain_calls > 10);                                                            // ... chains among them,
    CHECK(g_chain_product_calls > 5);                                                     // and maps evaluated together with a product
    printf("asan_deferred: directed scenarios + 40 fuzzed programs agree with eager evaluation (%ld fused consumer launches), no block left allocated\n",
           fused_total);
    return 0;
}
