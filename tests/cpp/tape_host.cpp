// CPU instantiation of the product's generic tape over the oracle array type (test infrastructure).
#include "../../oracle/host_array.h"
#include "../../enoki_amd/src/autodiff_impl.h"
#include "tape_program.h"

namespace enoki {
template struct Tape<HostArray<float>>;
}

using namespace enoki;
using FloatD = DiffArray<HostArray<float>>;
using UInt32D = DiffArray<HostArray<uint32_t>>;

extern "C" __attribute__((visibility("default")))
int host_tape_program(const int32_t *prog, size_t n_ops, const float *const *inputs, const uint64_t *sizes,
                      const uint8_t *leaf, size_t n_in, const uint32_t *const *index_inputs,
                      const uint64_t *index_sizes, size_t n_idx, int mode, int fwd_leaf, int simplify,
                      float *out_value, uint64_t *out_size, float *const *grads) {
    auto to_host = [](const HostArray<float> &a, float *dst, size_t n) {
        if (a.size() == 1 && n != 1) { for (size_t i = 0; i < n; ++i) dst[i] = a.coeff(0); }
        else memcpy(dst, a.data(), n * sizeof(float));
    };
    try {
        return run_tape_program<FloatD, UInt32D>(prog, n_ops, inputs, sizes, leaf, n_in, index_inputs, index_sizes,
                                                 n_idx, mode, fwd_leaf, simplify, out_value, out_size, grads, to_host);
    } catch (const std::exception &e) {
        fprintf(stderr, "host_tape_program: %s\n", e.what());
        return -3;
    }
}

extern "C" __attribute__((visibility("default"))) size_t host_tape_live_nodes() {
    return Tape<HostArray<float>>::get()->node_count();
}
