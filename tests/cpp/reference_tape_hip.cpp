// The reference's OWN tape (src/autodiff/autodiff.cpp) and its OWN test suite (tests/autodiff.cpp, 47 tests), both
// compiled unmodified from where they lie, on top of the reference's OWN headers -- with this repository's device backend
// underneath: integration/enoki/hip.h is the array type, integration/hip_hooks.cpp the JIT hooks of an eager backend.
// Everything above the C ABI in this binary is reference code; compare tests/cpp/reftest_autodiff_hip.cpp, where everything
// above the C ABI is this repository's code.  Built only where /root/reference exists (enoki_amd/_build.py).
#define ENOKI_CUDA 1                       /* activates the "device array" branches of autodiff.cpp (safe_mul, callbacks) */
#include "test.h"                          /* tests/cpp/refshim/test.h (the build passes -iquote): collects ENOKI_TEST cases */
#include <enoki/dynamic.h>
#include <enoki/hip.h>                     /* integration/enoki/hip.h */
#include <enoki/autodiff.h>
#include <enoki/color.h>

#include REFERENCE_TAPE_FILE               /* /root/reference/src/autodiff/autodiff.cpp */

NAMESPACE_BEGIN(enoki)
template struct Tape<HIPArray<float>>;
template struct DiffArray<HIPArray<float>>;
/// what the test's `DynamicArray<Packet<T>>` aliases turn into below
template <typename P> using HIPArrayOfPacket = HIPArray<scalar_t<P>>;
NAMESPACE_END(enoki)

// every enoki header the test asks for is included by now (include guards): from here on only the test's own aliases
// `using FloatX = DynamicArray<FloatP>` / `using UInt32X = DynamicArray<UInt32P>` see this name
#define DynamicArray HIPArrayOfPacket
#include REFERENCE_TEST_FILE               /* /root/reference/tests/autodiff.cpp */

// ---- `--bench n`: BASELINE config 3b on this stack (reference tape + reference router + integration header), for the table in
// DESIGN.md that separates what the C ABI's kernels deliver from what this repository's own binding adds on top ------------
#include <chrono>
static int bench_cfg3b(size_t n, int steps) {
    const size_t K = 1 << 20;
    FloatD A = sin(linspace<FloatD>(0.f, 100.f, K)), B = cos(linspace<FloatD>(0.f, 70.f, K));
    FloatD x = linspace<FloatD>(-1.f, 1.f, n);
    UInt32D idx = (arange<UInt32D>(n) * 2654435761u) >> 12;                    // < 2^20, pseudo-random
    double best = 1e30, y_value = 0;
    for (int s = 0; s < steps + 2; ++s) {
        ek_hip_sync();
        auto t0 = std::chrono::steady_clock::now();
        set_requires_gradient(A); set_requires_gradient(B);
        FloatD u = fmadd(gather<FloatD>(A, idx), x, gather<FloatD>(B, idx));
        FloatD y = hsum(sin(u));
        backward(y);
        FloatX gA = gradient(A), gB = gradient(B);
        ek_hip_sync();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (s >= 2) best = std::min(best, ms);
        y_value = detach(y).coeff(0);
        (void) gA; (void) gB;
        A = detach(A); B = detach(B);
    }
    printf("reference tape + reference router + integration/enoki/hip.h: cfg3b n=%zu K=%zu: %.3f ms per step (best of %d), y = %.4f\n",
           n, K, best, steps, y_value);
    // one more step under the library's per-launch events: which kernels the reference's layers issue
    ek_hip_profile_begin();
    {
        set_requires_gradient(A); set_requires_gradient(B);
        FloatD y = hsum(sin(fmadd(gather<FloatD>(A, idx), x, gather<FloatD>(B, idx))));
        backward(y);
        ek_hip_sync();
    }
    char *report = ek_hip_profile_end();
    if (report) { printf("kernels of one step: %s\n", report); free(report); }
    return 0;
}

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_crash(int sig) {
    void *frames[48];
    int n = backtrace(frames, 48);
    const char msg[] = "\n*** crash inside the reference's code path, backtrace:\n";
    (void) !write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    _exit(128 + sig);
}

int main(int argc, char **argv) {
    signal(SIGSEGV, on_crash); signal(SIGABRT, on_crash); signal(SIGBUS, on_crash);
    if (ek_hip_init(-1) != EK_OK) { std::cerr << ek_hip_last_error() << std::endl; return 2; }
    if (argc >= 3 && strcmp(argv[1], "--bench") == 0) return bench_cfg3b((size_t) atoll(argv[2]), 5);
    return test::run_all();
}
