// BASELINE config 5 (3-bounce path tracer with gradients w.r.t. an albedo texture) on the device, two ways, from the ONE
// templated source examples/path_trace.h (which oracle/ref_driver.cpp instantiates on the reference's arrays: ref_cfg5).
//
//   path_trace_device         the template on DiffArray<HIPArray<float>>: every operation one kernel, the three texture lookups on
//                             the tape, backward() = three scatter_adds (~1000 launches, 4 kB of array traffic per path)
//   path_trace_fused_device   what the reference's JIT gives such a program -- ONE kernel per evaluation (src/cuda/jit.cu:1066-1217,
//                             1418-1508) -- without a JIT: the same template instantiated on ONE-ELEMENT PACKETS inside one
//                             __global__ kernel (the vocabulary of enoki::vectorize()), with the differentiable value type replaced
//                             by a forward-mode dual that carries one derivative slot per texture lookup.  The kernel writes what
//                             the adjoint needs -- per bounce the texel index and d(radiance)/d(albedo of that bounce), 24 B per
//                             path -- and the loss; on the tape it is ONE node (Tape::append_custom) whose adjoint is ONE
//                             scatter_add of those records into grad(tex).  No tape node per elementwise operation.
//
// Build: hipcc --offload-arch=gfx950 -x hip -ffp-contract=off (enoki_amd/_build.py) -> examples/libpath_trace.so
#include <enoki/vectorize.h>

#include <enoki/random.h>

ENOKI_DEVICE_CODE_BEGIN
#include "path_trace.h"

/// PCG32 (include/enoki/random.h:62-119 of the reference: seed, XSH-RR output, 23-bit float) for ONE lane of a fused kernel
template <typename FloatP> struct LanePCG32 {
    uint64_t state, inc;
    LanePCG32(uint64_t initstate, uint64_t initseq) {
        state = 0;
        inc = (initseq << 1) | 1u;
        next_uint32();
        state += initstate;
        next_uint32();
    }
    uint32_t next_uint32() {
        const uint64_t old = state;
        state = old * PCG32_MULT + inc;
        const uint32_t xorshifted = (uint32_t) (((old >> 18) ^ old) >> 27), rot = (uint32_t) (old >> 59);
        return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
    }
    FloatP next_float32() {
        const uint32_t bits = (next_uint32() >> 9) | 0x3f800000u;
        float f;
        __builtin_memcpy(&f, &bits, 4);
        return FloatP(f - 1.f);
    }
};

/// value + its derivative w.r.t. each of N gathered table entries (forward mode; N is small: one slot per lookup of a path).
/// The operations a multilinear throughput / radiance recurrence needs.
template <typename Value, int N> struct GatherDual {
    Value v;
    Value d[N];
    GatherDual() = default;
    GatherDual(float c) : v(c) {
        for (int i = 0; i < N; ++i) d[i] = Value(0.f);
    }
    friend GatherDual operator+(const GatherDual &a, const GatherDual &b) {
        GatherDual r;
        r.v = a.v + b.v;
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
        return r;
    }
    friend GatherDual operator*(const GatherDual &a, const GatherDual &b) {
        GatherDual r;
        r.v = a.v * b.v;
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
        return r;
    }
    friend GatherDual operator*(const GatherDual &a, float c) {
        GatherDual r;
        r.v = a.v * c;
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c;
        return r;
    }
};
ENOKI_DEVICE_CODE_END

#include <enoki/autodiff.h>

#include <cstdio>

using namespace enoki;
using FloatC = HIPArray<float>;
using UInt32C = HIPArray<uint32_t>;
using UInt64C = HIPArray<uint64_t>;
using FloatD = DiffArray<FloatC>;
using UInt32D = DiffArray<UInt32C>;

template <int Bounces>
__global__ __launch_bounds__(256) void k_path_trace(const float *__restrict__ tex, size_t n, uint64_t seed, uint64_t first_lane,
                                                    uint32_t width, float *__restrict__ partials, uint32_t *__restrict__ rec_idx,
                                                    float *__restrict__ rec_w) {
    using FloatP = Array<float, 1>;
    using UInt32P = Array<uint32_t, 1>;
    using Dual = GatherDual<FloatP, Bounces>;
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f;
    if (i < n) {
        LanePCG32<FloatP> rng(seed, (uint64_t) i + first_lane);
        uint32_t idx[Bounces];
        int k = 0;
        auto lookup = [&](const UInt32P &texel) {
            const uint32_t t = texel.coeff(0);
            Dual r(0.f);
            r.v = FloatP(tex[t]);
            r.d[k] = FloatP(1.f);
            idx[k] = t;
            ++k;
            return r;
        };
        const Dual rad = cfg5::path_trace<Bounces, Dual, FloatP>(rng, lookup, width);
        loss = rad.v.coeff(0);
#pragma unroll
        for (int b = 0; b < Bounces; ++b) {
            rec_idx[(size_t) b * n + i] = idx[b];
            rec_w[(size_t) b * n + i] = rad.d[b].coeff(0);
        }
    }
    __shared__ float wave_part[4];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) loss += __shfl_down(loss, d, 64);
    if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]);
}

/// loss = hsum(radiance of n paths) as ONE tape node over `tex`
static FloatD path_trace_fused(const FloatD &tex, size_t n, uint64_t seed, uint64_t first_lane, uint32_t width) {
    constexpr int B = 3;
    const FloatC &t = detach(tex);
    const size_t K = t.size();
    UInt32C rec_idx = empty<UInt32C>(B * n);
    FloatC rec_w = empty<FloatC>(B * n);
    const unsigned blocks = (unsigned) ((n + 255) / 256);
    FloatC partials = empty<FloatC>(blocks);
    hipLaunchKernelGGL(k_path_trace<B>, dim3(blocks), dim3(256), 0, (hipStream_t) ek_hip_stream(), t.data(), n, seed, first_lane, width,
                       partials.data(), rec_idx.data(), rec_w.data());
    if (hipGetLastError() != hipSuccess) throw std::runtime_error("path_trace_fused(): kernel launch failed");
    FloatC loss = hsum(partials);
    return FloatD::custom_(tex, std::move(loss), "path_trace", [rec_idx, rec_w, K](const FloatC &g) {
        // d loss / d tex[k] = sum over the lookups that hit k of their recorded derivative; times the incoming gradient
        FloatC out = zero<FloatC>(K);
        scatter_add(out, rec_w * g, rec_idx);
        return out;
    });
}

static FloatD path_trace_unfused(const FloatD &tex, size_t n, uint64_t seed, uint64_t first_lane, uint32_t width) {
    PCG32<FloatC> rng(UInt64C(seed), arange<UInt64C>(n) + UInt64C(first_lane));
    auto lookup = [&](const UInt32C &texel) { return gather<FloatD>(tex, UInt32D(texel)); };
    return hsum(cfg5::path_trace<3, FloatD, FloatC>(rng, lookup, width));
}

static int run(bool fused, const float *tex_, size_t K, size_t n, uint64_t seed, uint64_t first_lane, int bounces, uint32_t width,
               float *loss, float *grad_tex) {
    if (bounces != 3) { fprintf(stderr, "path_trace: built for 3 bounces\n"); return -1; }
    try {
        FloatD tex = FloatC::map((void *) tex_, K);
        set_requires_gradient(tex);
        FloatD y = fused ? path_trace_fused(tex, n, seed, first_lane, width) : path_trace_unfused(tex, n, seed, first_lane, width);
        backward(y);
        FloatC g = gradient(tex);
        if (ek_hip_memcpy_device(grad_tex, g.data(), K * sizeof(float)) != EK_OK) return -2;
        if (ek_hip_memcpy_device(loss, detach(y).data(), sizeof(float)) != EK_OK) return -2;
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "path_trace: %s\n", e.what());
        return -1;
    }
}

/// All pointers are DEVICE pointers; loss: 1 float, grad_tex: K floats (written).  One forward + backward() per call.
extern "C" __attribute__((visibility("default")))
int path_trace_fused_device(const float *tex, size_t K, size_t n, uint64_t seed, uint64_t first_lane, int bounces, uint32_t width,
                            float *loss, float *grad_tex) {
    return run(true, tex, K, n, seed, first_lane, bounces, width, loss, grad_tex);
}

extern "C" __attribute__((visibility("default")))
int path_trace_device(const float *tex, size_t K, size_t n, uint64_t seed, uint64_t first_lane, int bounces, uint32_t width,
                      float *loss, float *grad_tex) {
    return run(false, tex, K, n, seed, first_lane, bounces, width, loss, grad_tex);
}
