// GPU instantiation: the product's DiffArray<HIPArray<float>> driven by the same register machine.
#include <enoki/hip.h>
#include <enoki/autodiff.h>
#include "tape_program.h"

using namespace enoki;
using FloatD = DiffArray<HIPArray<float>>;
using UInt32D = DiffArray<HIPArray<uint32_t>>;

extern "C" __attribute__((visibility("default")))
int hip_tape_program(const int32_t *prog, size_t n_ops, const float *const *inputs, const uint64_t *sizes,
                     const uint8_t *leaf, size_t n_in, const uint32_t *const *index_inputs,
                     const uint64_t *index_sizes, size_t n_idx, int mode, int fwd_leaf, int simplify,
                     float *out_value, uint64_t *out_size, float *const *grads) {
    auto to_host = [](const HIPArray<float> &a, float *dst, size_t n) {
        if (a.size() == 1 && n != 1) { float v = a.coeff(0); for (size_t i = 0; i < n; ++i) dst[i] = v; }
        else { auto h = a.to_host(); memcpy(dst, h.data(), n * sizeof(float)); }
    };
    try {
        return run_tape_program<FloatD, UInt32D>(prog, n_ops, inputs, sizes, leaf, n_in, index_inputs, index_sizes,
                                                 n_idx, mode, fwd_leaf, simplify, out_value, out_size, grads, to_host);
    } catch (const std::exception &e) {
        fprintf(stderr, "hip_tape_program: %s\n", e.what());
        return -3;
    }
}

extern "C" __attribute__((visibility("default"))) size_t hip_tape_live_nodes() {
    return Tape<HIPArray<float>>::get()->node_count();
}
