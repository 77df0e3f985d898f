/*
    enoki/color.h -- sRGB <-> linear transfer functions for any array type (device, differentiable, static, scalar)

    Same approximation as the reference's include/enoki/color.h:19-118: below the knee the transfer function is the linear
    segment, above it a rational function p(t) / q(t) (t = sqrt(x) for linear -> sRGB, t = x for the inverse) fitted to
    the power law, degree 5 / 4 in single and 10 / 9 in double precision, evaluated with the polyN association of
    array_math.h so that results agree with the reference bit for bit.  Written with select() instead of masked
    assignment + an any() early-out: on a device array the early-out would cost a synchronisation per call, and
    DiffArray differentiates through select() exactly like through the masked assignment (autodiff.h:407-453).
*/
#pragma once

#include <enoki/special.h>

namespace enoki {

/// linear radiance -> sRGB encoded value (x <= 0.0031308: 12.92 x, above: 1.055 x^(1/2.4) - 0.055 as p(sqrt x) / q(sqrt x) * x)
template <typename T> expr_t<T> linear_to_srgb(const T &x) {
    using Value = expr_t<T>;
    using Scalar = scalar_t<Value>;
    Value t = sqrt(x), p, q;
    if constexpr (std::is_same_v<Scalar, float>) {
        p = detail::poly5(t, -0.0016829072605308378, 0.03453868659826638, 0.7642611304733891, 2.0041169284241644,
                          0.7551545191665577, -0.016202083165206348);
        q = detail::poly5(t, 4.178892964897981e-7, -0.00004375359692957097, 0.03467195408529984, 0.6085338522168684,
                          1.8970238036421054, 1.);
    } else {
        p = detail::poly10(t, -3.7113872202050023e-6, -0.00021805827098915798, 0.002531335520959116, 0.2263810267005674,
                           3.0477578489880823, 15.374469584296442, 32.44669922192121, 27.901125077137042,
                           8.450947414259522, 0.5838023820686707, -0.0031151377052754843);
        q = detail::poly10(t, 2.2380622409188757e-11, -8.387527630781522e-9, 0.00007045228641004039, 0.007244514696840552,
                           0.21749170309546628, 2.575446652731678, 13.297981743005433, 30.50364355650628,
                           29.70548706952188, 10.723011300050162, 1.);
    }
    Value slope = select(x > Scalar(0.0031308), p / q, Value(Scalar(12.92)));
    return slope * x;
}

/// sRGB encoded value -> linear radiance (x <= 0.04045: x / 12.92, above: ((x + 0.055) / 1.055)^2.4 as p(x) / q(x) * x)
template <typename T> expr_t<T> srgb_to_linear(const T &x) {
    using Value = expr_t<T>;
    using Scalar = scalar_t<Value>;
    Value p, q;
    if constexpr (std::is_same_v<Scalar, float>) {
        p = detail::poly4(x, -0.0163933279112946, -0.7386328024653209, -11.199318357635072, -47.46726633009393,
                          -36.04572663838034);
        q = detail::poly4(x, -0.004261480793199332, -19.140923959601675, -59.096406619244426, -18.225745396846637, 1.);
    } else {
        p = detail::poly9(x, -0.008042950896814532, -0.5489744177844188, -14.786385491859248, -200.19589605282445,
                          -1446.951694673217, -5548.704065887224, -10782.158977031822, -9735.250875334352,
                          -3483.4445569178347, -342.62884098034357);
        q = detail::poly9(x, -2.2132610916769585e-8, -9.646075249097724, -237.47722999429413, -2013.8039726540235,
                          -7349.477378676199, -11916.470977597566, -8059.219012060384, -1884.7738197074218,
                          -84.8098437770271, 1.);
    }
    Value slope = select(x > Scalar(0.04045), p / q, Value(Scalar(1.0 / 12.92)));
    return slope * x;
}

} // namespace enoki
