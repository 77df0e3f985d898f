/*
    enoki/cuda.h -- source compatibility for programs written against the reference's GPU backend

    `#include <enoki/cuda.h>`, `CUDAArray<float>`, `DiffArray<CUDAArray<float>>`, `cuda_eval()`, `cuda_sync()`,
    `cuda_whos()`, `cuda_malloc_trim()`, `cuda_set_log_level()` keep compiling: the array template is an alias of
    HIPArray<T> (include/enoki/hip.h) and the runtime calls forward to libenoki-hip.so (reference cuda.h:27-200, 205-954).
    `cuda_eval()` stays a no-op -- the backend launches pre-compiled kernels as operations are called, there is no trace --
    and `is_cuda_array_v<T>` stays false (array.h): generic code that asks it only does so to decide whether a trace must
    be flushed.  `is_device_array_v<T>` is the trait for "lives in GPU memory".
*/
#pragma once

#include <enoki/hip.h>

#include <cstdint>
#include <string>

namespace enoki {

template <typename Value> using CUDAArray = HIPArray<Value>;

inline std::string cuda_whos_string() { return hip_whos(); }
/// returns a malloc()ed string like the reference (caller frees)
inline char *cuda_whos() { return ek_hip_whos(); }
inline void cuda_malloc_trim() { hip_malloc_trim(); }
inline void cuda_set_log_level(uint32_t level) { ek_hip_set_log_level(level); }
inline void cuda_eval_var(uint32_t, bool = false) { }
/// the device synchronisation that the reference's cuda_sync() performs (array.h's placeholder of the same name does nothing:
/// code that includes THIS header asked for the GPU runtime)
inline void cuda_device_sync() { hip_sync(); }

} // namespace enoki
