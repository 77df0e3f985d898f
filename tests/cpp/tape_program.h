// tests/cpp/tape_program.h -- a tiny register machine over DiffArray values, shared by the three
// implementations that the tape parity tests compare:
//   oracle/ref_driver.cpp  ref_tape_program   the unmodified reference (DiffArray<DynamicArray<Packet<float>>>)
//   tests/cpp/tape_host.cpp  host_tape_program  the product's generic Tape over oracle::HostArray (CPU)
//   tests/cpp/tape_hip.cpp   hip_tape_program   the product's Tape over HIPArray<float> (GPU)
// Program encoding: n_ops quadruples (opcode, a, b, c) of int32; operands index a register file whose
// first n_in entries are the inputs; every op writes the next register.  See tests/test_tape_parity.py.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

enum {
    P_ADD = 0, P_SUB, P_MUL, P_DIV, P_FMADD, P_NEG, P_ABS, P_SQRT, P_RCP, P_RSQRT, P_SIN, P_COS,
    P_EXP, P_LOG, P_HSUM, P_HPROD, P_MIN, P_MAX, P_GATHER, P_SCATTER_ADD, P_SCATTER, P_SELECT_GT0,
    P_MULC, P_ADDC, P_TANH, P_TAN, P_ATAN2, P_FMSUB, P_FNMADD, P_FNMSUB, P_SINH, P_COSH, P_ASIN,
    P_ACOS, P_ATAN, P_PSUM, P_REVERSE, P_ASINH, P_ACOSH, P_ATANH, P_CBRT, P_POW, P_COT
};

/// FloatD / UInt32D: differentiable float array and its index array type; `to_host(array, dst, n)` copies out
template <typename FloatD, typename UInt32D, typename ToHost>
int run_tape_program(const int32_t *prog, size_t n_ops, const float *const *inputs, const uint64_t *sizes,
                     const uint8_t *leaf, size_t n_in, const uint32_t *const *index_inputs,
                     const uint64_t *index_sizes, size_t n_idx, int mode, int fwd_leaf, int simplify,
                     float *out_value, uint64_t *out_size, float *const *grads, ToHost to_host) {
    using namespace enoki;
    using FloatX = typename FloatD::Type;
    using UInt32X = typename UInt32D::Type;
    std::vector<FloatD> reg(n_in + n_ops);
    std::vector<UInt32D> ireg(n_idx);
    for (size_t i = 0; i < n_in; ++i) {
        reg[i] = FloatD(FloatX::copy(inputs[i], sizes[i]));
        if (leaf[i]) set_requires_gradient(reg[i]);
    }
    for (size_t i = 0; i < n_idx; ++i) ireg[i] = UInt32D(UInt32X::copy(index_inputs[i], index_sizes[i]));

    size_t last = n_in ? n_in - 1 : 0;
    for (size_t k = 0; k < n_ops; ++k) {
        const int32_t *p = prog + 4 * k;
        size_t d = n_in + k;
        auto R = [&](int32_t i) -> FloatD & { return reg[(size_t) i]; };
        float cst;
        memcpy(&cst, &p[2], sizeof(float));
        switch (p[0]) {
            case P_ADD: reg[d] = R(p[1]) + R(p[2]); break;
            case P_SUB: reg[d] = R(p[1]) - R(p[2]); break;
            case P_MUL: reg[d] = R(p[1]) * R(p[2]); break;
            case P_DIV: reg[d] = R(p[1]) / R(p[2]); break;
            case P_FMADD: reg[d] = fmadd(R(p[1]), R(p[2]), R(p[3])); break;
            case P_FMSUB: reg[d] = fmsub(R(p[1]), R(p[2]), R(p[3])); break;
            case P_FNMADD: reg[d] = fnmadd(R(p[1]), R(p[2]), R(p[3])); break;
            case P_FNMSUB: reg[d] = fnmsub(R(p[1]), R(p[2]), R(p[3])); break;
            case P_NEG: reg[d] = -R(p[1]); break;
            case P_ABS: reg[d] = abs(R(p[1])); break;
            case P_SQRT: reg[d] = sqrt(R(p[1])); break;
            case P_RCP: reg[d] = rcp(R(p[1])); break;
            case P_RSQRT: reg[d] = rsqrt(R(p[1])); break;
            case P_SIN: reg[d] = sin(R(p[1])); break;
            case P_COS: reg[d] = cos(R(p[1])); break;
            case P_EXP: reg[d] = exp(R(p[1])); break;
            case P_LOG: reg[d] = log(R(p[1])); break;
            case P_HSUM: reg[d] = hsum(R(p[1])); break;
            case P_HPROD: reg[d] = hprod(R(p[1])); break;
            case P_PSUM: reg[d] = psum(R(p[1])); break;
            case P_REVERSE: reg[d] = reverse(R(p[1])); break;
            case P_MIN: reg[d] = min(R(p[1]), R(p[2])); break;
            case P_MAX: reg[d] = max(R(p[1]), R(p[2])); break;
            case P_MULC: reg[d] = R(p[1]) * cst; break;
            case P_ADDC: reg[d] = R(p[1]) + cst; break;
            case P_SELECT_GT0: reg[d] = select(R(p[1]) > 0.f, R(p[2]), R(p[3])); break;
            case P_GATHER: reg[d] = gather<FloatD>(R(p[1]), ireg[(size_t) p[2]]); break;
            case P_SCATTER_ADD:
                scatter_add(R(p[1]), R(p[2]), ireg[(size_t) p[3]]);
                reg[d] = R(p[1]);
                break;
            case P_SCATTER:
                scatter(R(p[1]), R(p[2]), ireg[(size_t) p[3]]);
                reg[d] = R(p[1]);
                break;
            case P_TAN: reg[d] = tan(R(p[1])); break;
            case P_COT: reg[d] = cot(R(p[1])); break;
            case P_ASIN: reg[d] = asin(R(p[1])); break;
            case P_ACOS: reg[d] = acos(R(p[1])); break;
            case P_ATAN: reg[d] = atan(R(p[1])); break;
            case P_ATAN2: reg[d] = atan2(R(p[1]), R(p[2])); break;
            case P_SINH: reg[d] = sinh(R(p[1])); break;
            case P_COSH: reg[d] = cosh(R(p[1])); break;
            case P_TANH: reg[d] = tanh(R(p[1])); break;
            case P_ASINH: reg[d] = asinh(R(p[1])); break;
            case P_ACOSH: reg[d] = acosh(R(p[1])); break;
            case P_ATANH: reg[d] = atanh(R(p[1])); break;
            case P_CBRT: reg[d] = cbrt(R(p[1])); break;
            case P_POW: reg[d] = pow(R(p[1]), R(p[2])); break;
            default: return -1;
        }
        last = d;
    }

    FloatD &y = reg[last];
    *out_size = y.size();
    to_host(y.value_(), out_value, y.size());
    if (simplify) FloatD::simplify_graph_();

    if (mode == 0) {
        backward(y);
        for (size_t i = 0; i < n_in; ++i) {
            if (!leaf[i]) continue;
            const FloatX &g = gradient(reg[i]);
            if (g.size() == 0) {
                for (size_t j = 0; j < sizes[i]; ++j) grads[i][j] = 0.f;
            } else {
                to_host(g, grads[i], sizes[i]);
            }
        }
    } else {
        forward(reg[(size_t) fwd_leaf]);
        to_host(gradient(y), grads[0], y.size());
    }
    return 0;
}
