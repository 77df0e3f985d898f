// The reference's tests/sphere.cpp (make_rays / intersect_rays / shade_hits over ray.h's ENOKI_STRUCT Ray, run through
// vectorize() separately and combined), compiled UNMODIFIED by hipcc from where it lies: DynamicArray<Array<float>>
// becomes HIPArray<float> (tests/cpp/refshim/enoki/dynamic.h) and vectorize() is this repository's compile-time fusion
// (include/enoki/vectorize.h): each of the reference's four *_dynamic wrappers is ONE kernel.  The reference program
// has no assertion of its own; it writes sphere1.ppm / sphere2.ppm, which tests/test_reference_sources_gpu.py compares
// pixel for pixel with the CPU oracle.
#include <enoki/vectorize.h>

#include <enoki/dynamic.h>      // (the shim) everything the reference file includes is loaded before it is ...
#include <chrono>
#include <fstream>
#include <iostream>

// HIP needs the reference's kernels marked for the device: see refshim/enoki/array.h for how that is done without
// touching the files (a device-code region from ray.h to the last kernel, closed level by level by ENOKI_INLINE).
#undef ENOKI_INLINE
#define ENOKI_INLINE _Pragma("clang force_cuda_host_device end") __host__ __device__ inline __attribute__((always_inline))
#define REFSHIM_DEVICE_REGION_ARMED 1
#define main reference_sphere_main
#include REFERENCE_TEST_FILE
#undef main

int main(int argc, char **argv) {
    try {
        uint64_t before = ek_hip_launch_count();
        int rc = reference_sphere_main(argc, argv);
        std::cerr << "kernel launches: " << (ek_hip_launch_count() - before) << std::endl;
        return rc;
    } catch (const std::exception &e) {
        std::cerr << "exception: " << e.what() << std::endl;
        return 2;
    }
}
