// The reference's OWN autodiff test suite (tests/autodiff.cpp, 47 tests), compiled UNMODIFIED from where it lies against
// this repository's headers and libenoki-hip-autodiff.so, with its four array aliases landing on the device types:
//     FloatX -> HIPArray<float>, FloatD -> DiffArray<HIPArray<float>>, UInt32X / UInt32D likewise
// (tests/cpp/refshim/enoki/dynamic.h).  This is the proof of the template ("backend concept") half of the drop-in
// boundary, SURVEY.md 8b(i): the reference's templated test code is the caller, HIPArray is the backend.
// Built only where /root/reference exists (enoki_amd/_build.py); the binary travels to the GPU box, where
// tests/test_reference_sources_gpu.py runs it.  The file is never copied: REFERENCE_TEST_FILE is a path.
#include REFERENCE_TEST_FILE

int main() {
    try {
        enoki::detail::hip_check(ek_hip_init(-1), "ek_hip_init");
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 2;
    }
    return test::run_all();
}
