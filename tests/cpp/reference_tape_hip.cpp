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

int main() {
    signal(SIGSEGV, on_crash); signal(SIGABRT, on_crash); signal(SIGBUS, on_crash);
    if (ek_hip_init(-1) != EK_OK) { std::cerr << ek_hip_last_error() << std::endl; return 2; }
    return test::run_all();
}
