// The tape (enoki_amd/src/autodiff_impl.h, Tape<HIPArray<float>>) on top of HIPArray's deferred nodes, under
// AddressSanitizer + LeakSanitizer + UBSan and WITHOUT a GPU: the C ABI is the host stand-in of host_abi_stub.h.
//
// Random differentiable programs -- leaves, tables, shared-index gathers, struct gathers through records, arithmetic, unary maps, select, hsum, broadcast
// scalars -- are recorded and differentiated twice, once with every gather / unary result deferred (ENOKI_HIP_DEFER_MIN=1)
// and once with deferral switched off.  The tape keeps deferred arrays as edge weights, hands them to the fused
// scatter_add and hsum consumers, shares buffers between gradients: every value and every gradient must come out with the
// same bits both ways, and the stand-in's allocation count must return to zero after each program.
//
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iinclude tests/cpp/asan_tape.cpp -o tests/cpp/asan_tape.bin
#include <enoki/hip.h>
#include <enoki/autodiff.h>

#if !defined(EK_FUZZ_DEVICE)
#  include "host_abi_stub.h"
#  include "../../enoki_amd/src/autodiff_impl.h"
namespace enoki { template struct Tape<HIPArray<float>>; }
#else
// The same programs against the REAL library on a GPU (tests/test_deferred_map_gpu.py, fuzz_tape_hip.bin: linked with
// libenoki-hip-autodiff.so / libenoki-hip.so, no sanitizers): deferred and eager evaluation must agree bit for bit on the
// device kernels too -- fused consumers, record gathers (forced) and the deterministic scatter_add order included.
static long g_fused_calls = 0, g_record_gathers = 1000;
static struct { bool empty() const { return true; } } g_live;
#endif

#include <random>
#include <vector>

using namespace enoki;
#if defined(EK_FUZZ_DEVICE) && defined(EK_FUZZ_DOUBLE)
using Real = double;                               // the device build also runs in float64 (fuzz_tape_hip_f64.bin)
#else
using Real = float;
#endif
using F = HIPArray<Real>;
using U = HIPArray<uint32_t>;
using D = DiffArray<F>;
using UD = DiffArray<U>;

#define CHECK(expr) do { if (!(expr)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #expr); exit(1); } } while (0)

static std::vector<Real> host(const F &a) {
#if defined(EK_FUZZ_DEVICE)
    return a.to_host();                             // one copy per array
#else
    std::vector<Real> v(a.size());
    for (size_t i = 0; i < v.size(); ++i) v[i] = a.coeff(i);
    return v;
#endif
}
static bool same(const std::vector<Real> &a, const std::vector<Real> &b) {
    return a.size() == b.size() && (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(Real)) == 0);
}

static std::vector<std::vector<Real>> run_program(uint32_t seed, bool defer) {
    hip_set_defer(defer);
    std::mt19937 rng(seed);
    std::vector<std::vector<Real>> seen;
    {
        const size_t n = 3000 + rng() % 2000, K = 64 + rng() % 200;
        // leaves: two arrays of size n, two tables of size K, one scalar
        std::vector<D> leaves = { D(linspace<F>(-2.f, 2.f, n)), D(sin(linspace<F>(0.f, 9.f, n))),
                                  D(cos(linspace<F>(0.f, 5.f, K))), D(linspace<F>(0.5f, 1.5f, K)), D(F(0.75f)) };
        for (D &l : leaves) set_requires_gradient(l);
        UD idx = UD((arange<U>(n) * U(2654435761u)) >> U(8u)) % UD(U((uint32_t) K));
        std::vector<D> pool = { leaves[0], leaves[1], leaves[0] * leaves[4] };
        auto pick = [&]() -> D & { return pool[rng() % pool.size()]; };
        for (int step = 0; step < 14; ++step) {
            D r;
            switch (rng() % 12) {
                case 0: r = sin(pick()); break;
                case 1: r = cos(pick()) * pick(); break;
                case 2: r = exp(pick() * D(F(0.2f))); break;
                case 3: r = fmadd(gather<D>(leaves[2], idx), pick(), gather<D>(leaves[3], idx)); break;   // shared-index pair
                case 4: r = gather<D>(leaves[2 + rng() % 2], idx) * pick(); break;
                case 5: r = pick() + pick() * leaves[4]; break;
                case 6: r = select(pick() > D(F(0.1f)), pick(), -pick()); break;
                case 7: r = sqrt(abs(pick()) + D(F(1.f))); break;
                case 8: { auto sc = sincos(pick()); r = sc.first * sc.second; break; }
                case 9: r = pick() - hsum(sin(pick())) * D(F(1e-3f)); break;                             // reduction mid-graph
                case 10: r = abs(pick()) * rcp(abs(pick()) + D(F(2.f))); break;
                case 11: {                                                                       // struct gather: one record lookup,
                    Array<D, 2> tables(leaves[2], leaves[3]);                                    // one tape node per component
                    Array<D, 2> g = gather<Array<D, 2>>(tables, idx);
                    r = fmadd(g.x(), pick(), g.y() * leaves[4]);
                    break;
                }
            }
            pool[rng() % pool.size()] = r;
        }
        D loss = hsum(sin(pool[0])) + hsum(pool[1] * pool[2]) * D(F(0.5f)) + hsum(exp(pool[2] * D(F(0.1f))));
        seen.push_back(host(detach(loss)));
        backward(loss);
        for (D &l : leaves) seen.push_back(host(gradient(l)));
        if (rng() & 1) seen.push_back(host(detach(pool[rng() % pool.size()])));       // a value that may still be deferred
    }
    return seen;
}

#if !defined(EK_FUZZ_DEVICE)
/// BASELINE configs[2] with leaf ARRAYS (cfg3a): y = hsum(sin(fmadd(a, x, b))), backward().  Deferred, the forward pass is ONE
/// chain reduction that reads a, x, b (u = fmadd is never written: the sin and the cos that differentiating sin records are
/// its only holders), and the sweep's grad_b = cos(u), grad_a = safe_mul(x, cos(u)) are the two outputs of ONE pass.
static void directed_cfg3a() {
    std::vector<std::vector<Real>> res[2];
    for (int pass = 3; pass >= 0; --pass) {
        const int defer = pass & 1;
        const bool one_expression = pass >= 2;
        hip_set_defer(defer != 0);
        const size_t n = 5000;
        const long c0 = g_chain_calls, p0 = g_chain_product_calls, a0 = g_array_launches, u0 = g_unary_calls, s0 = g_sincos_calls;
        {
            F x = linspace<F>(-1.f, 1.f, n);
            (void) x.data();
            D a = D(linspace<F>(-2.f, 2.f, n)), b = D(sin(linspace<F>(0.f, 9.f, n)));
            (void) detach(a).data(); (void) detach(b).data();
            set_requires_gradient(a); set_requires_gradient(b);
            const long a1 = g_array_launches, u1 = g_unary_calls;
            // statement by statement, as a python caller's temporaries die -- or in ONE C++ expression: the routed unary functions
            // take an expiring argument's handle away as soon as their result exists (array.h), so the temporary that holds u
            // does not count as somebody who still wants it
            D y;
            if (one_expression) {
                y = hsum(sin(fmadd(a, D(x), b)));
            } else {
                D s;
                { D u = fmadd(a, D(x), b); s = sin(u); }
                y = hsum(s);
            }
            backward(y);
            F ga = gradient(a), gb = gradient(b);
            res[defer] = { host(detach(y)), host(ga), host(gb) };
            if (defer) {
                CHECK(g_chain_calls == c0 + 2);                 // the forward reduction and the backward pass
                CHECK(g_chain_product_calls == p0 + 1);         // which writes both gradients
                CHECK(g_array_launches == a1 && g_unary_calls == u1 && g_sincos_calls == s0);      // and nothing else ran over the arrays
            }
        }
        (void) c0; (void) p0; (void) a0; (void) u0;
        CHECK(g_live.empty());
    }
    for (size_t i = 0; i < 3; ++i) CHECK(same(res[0][i], res[1][i]));
    hip_set_defer(true);
    // a named variable is not expiring: the function leaves it alone; std::move() hands it over
    D keep = D(linspace<F>(0.f, 1.f, 5000));
    D s1 = sin(keep);
    CHECK(detach(keep).valid() && detach(keep).size() == 5000);
    D s2 = sin(std::move(keep));
    CHECK(!detach(keep).valid());
    CHECK(same(host(detach(s1)), host(detach(s2))));
}
#endif

int main() {
    setenv("ENOKI_HIP_DEFER_MIN", "1", 1);         // read once, on first use: every gather / fusable unary result is deferred
    long fused_total = 0;
#if !defined(EK_FUZZ_DEVICE)
    directed_cfg3a();
#endif
#if defined(EK_FUZZ_DEVICE)
    setenv("ENOKI_HIP_GATHER_RECORDS", "2", 1);    // struct gathers always through staged records
    if (ek_hip_init(-1) != EK_OK) { fprintf(stderr, "%s\n", ek_hip_last_error()); return 2; }
    ek_hip_set_tuning("deterministic", 1);         // fp scatter_add in element order: comparable bit for bit
    uint64_t launches_with = 0, launches_without = 0;
    {
        // ONE C++ expression launches what the statement-by-statement form launches (expiring temporaries let go of their
        // handles, array.h): BASELINE configs[1] is a chain reduction + its second stage, configs[2] on leaf arrays the same
        // forward pass and one backward pass with two outputs
        const size_t n = 1 << 18;
        F a = linspace<F>(-1.f, 1.f, n), x = sin(linspace<F>(0.f, 40.f, n)), b = cos(linspace<F>(0.f, 9.f, n));
        (void) a.data(); (void) x.data(); (void) b.data();
        uint64_t l0 = ek_hip_launch_count();
        F y = hsum(sin(exp(fmadd(a, x, b))));
        CHECK(ek_hip_launch_count() - l0 == 2);
        uint64_t counts[2];
        std::vector<Real> grads[2];
        for (int one_expression = 0; one_expression < 2; ++one_expression) {
            D da = D(a), db = D(b);
            set_requires_gradient(da); set_requires_gradient(db);
            l0 = ek_hip_launch_count();
            D yd;
            if (one_expression) {
                yd = hsum(sin(fmadd(da, D(x), db)));
            } else {
                D s;
                { D u = fmadd(da, D(x), db); s = sin(u); }
                yd = hsum(s);
            }
            CHECK(ek_hip_launch_count() - l0 == 2);
            backward(yd);
            F ga = gradient(da), gb = gradient(db);
            (void) ga.data(); (void) gb.data();
            counts[one_expression] = ek_hip_launch_count() - l0;
            grads[one_expression] = host(ga);
        }
        CHECK(counts[0] == counts[1]);
        CHECK(same(grads[0], grads[1]));
    }
#endif
#if defined(EK_FUZZ_DEVICE)
    const uint32_t programs = 400;                 // 0.01 s each on the device
#else
    const uint32_t programs = 60;                  // ASan + UBSan on the host stand-in
#endif
    for (uint32_t seed = 1; seed <= programs; ++seed) {
        long f0 = g_fused_calls;
#if defined(EK_FUZZ_DEVICE)
        uint64_t l0 = ek_hip_launch_count();
#endif
        auto with = run_program(seed, true);
#if defined(EK_FUZZ_DEVICE)
        launches_with += ek_hip_launch_count() - l0; l0 = ek_hip_launch_count();
        g_fused_calls += 2;
#endif
        fused_total += g_fused_calls - f0;
        CHECK(g_live.empty());
        f0 = g_fused_calls;
        auto without = run_program(seed, false);
#if defined(EK_FUZZ_DEVICE)
        launches_without += ek_hip_launch_count() - l0;
#endif
        CHECK(g_fused_calls == f0);
        CHECK(g_live.empty());
        CHECK(with.size() == without.size());
        for (size_t i = 0; i < with.size(); ++i)
            if (!same(with[i], without[i])) {
                size_t bad = 0;
                for (size_t j = 0; j < with[i].size() && j < without[i].size(); ++j) bad += with[i][j] != without[i][j];
                fprintf(stderr, "seed %u: observation %zu differs (%zu of %zu entries)\n", seed, i, bad, with[i].size());
                return 1;
            }
    }
    hip_set_defer(true);
    CHECK(fused_total > 100);
    CHECK(g_record_gathers > 20);
#if defined(EK_FUZZ_DEVICE)
    CHECK(launches_with < launches_without);
    printf("fuzz_tape_hip: %u fuzzed differentiable programs (float%d) give identical values and gradients with and without deferred "
           "evaluation on the device (%llu kernel launches deferred, %llu eager)\n", programs, (int) (8 * sizeof(Real)),
           (unsigned long long) launches_with, (unsigned long long) launches_without);
    return 0;
#endif
    printf("asan_tape: 60 fuzzed differentiable programs give identical values and gradients with and without deferred evaluation "
           "(%ld fused consumer launches), no block left allocated\n", fused_total);
    return 0;
}
