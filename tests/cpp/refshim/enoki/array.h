// tests/cpp/refshim/enoki/array.h -- forwards to the real <enoki/array.h>.  One extra duty in the hipcc build of the
// reference's tests/sphere.cpp (tests/cpp/reftest_sphere_hip.cpp): HIP only lets a __global__ kernel call functions that
// are marked for the device, and the reference's kernels are plain C++:
//     tests/ray.h            Ray::operator() -- ray.h includes <enoki/array.h> right before the struct
//     tests/sphere.cpp:58    make_rays (a plain template)
//     tests/sphere.cpp:67-88 intersect_rays, shade_hits, combined (declared ENOKI_INLINE)
// When the driver has armed REFSHIM_DEVICE_REGION_ARMED, this shim opens a device-code region at the include inside
// ray.h, three levels deep; the driver's ENOKI_INLINE closes one level per use and marks its function for the device
// explicitly, so the region ends exactly after the reference's last kernel and everything behind it (the *_dynamic
// wrappers, write_image(), main()) stays ordinary host code.  (No include guard on purpose: the real header has one.)
#include_next <enoki/array.h>

#if defined(__HIP__) && defined(REFSHIM_DEVICE_REGION_ARMED)
#  undef REFSHIM_DEVICE_REGION_ARMED
#  pragma clang force_cuda_host_device begin
#  pragma clang force_cuda_host_device begin
#  pragma clang force_cuda_host_device begin
#endif
