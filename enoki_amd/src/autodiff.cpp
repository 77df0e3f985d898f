// libenoki-hip-autodiff.so: explicit instantiations of the tape for the device array types
// (the analogue of src/autodiff/autodiff.cpp:1223-1241 in the reference, which instantiates
// Tape<CUDAArray<float>> / Tape<CUDAArray<double>>).
#include <enoki/hip.h>
#include "autodiff_impl.h"

namespace enoki {
template struct __attribute__((visibility("default"))) Tape<HIPArray<float>>;
template struct __attribute__((visibility("default"))) Tape<HIPArray<double>>;
} // namespace enoki
