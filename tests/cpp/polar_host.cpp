// polar_decomp (include/enoki/matrix.h) and transform_decompose / transform_compose / transform_compose_inverse
// (include/enoki/transform.h) on host scalars: properties of 2000 random affine matrices -- Q orthogonal, P symmetric,
// Q P = A, compose(decompose(A)) = A, compose * compose_inverse = I.  Run by tests/test_matrix.py.
#include <enoki/transform.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
using namespace enoki;
int main() {
    using M3 = Matrix<double, 3>; using M4 = Matrix<double, 4>;
    double worst_orth = 0, worst_sym = 0, worst_rec = 0, worst_round = 0, worst_inv = 0;
    srand(3);
    for (int it = 0; it < 2000; ++it) {
        M4 A = identity<M4>();
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) A(i, j) = (rand() / (double) RAND_MAX - 0.5) * 4;
        M3 sub; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) sub(i, j) = A(i, j);
        if (std::fabs(det(sub)) < 0.05) continue;
        auto [Q, P] = polar_decomp(sub);
        M3 QtQ = transpose(Q) * Q, QP = Q * P;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            worst_orth = std::fmax(worst_orth, std::fabs(QtQ(i, j) - (i == j)));
            worst_sym = std::fmax(worst_sym, std::fabs(P(i, j) - P(j, i)));
            worst_rec = std::fmax(worst_rec, std::fabs(QP(i, j) - sub(i, j)));
        }
        auto [S, q, t] = transform_decompose(A);
        M4 B = transform_compose(S, q, t), Bi = transform_compose_inverse(S, q, t), I = B * Bi;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            worst_round = std::fmax(worst_round, std::fabs(B(i, j) - A(i, j)));
            worst_inv = std::fmax(worst_inv, std::fabs(I(i, j) - (i == j)));
        }
    }
    printf("orthogonality %.2e  symmetry %.2e  Q P - A %.2e  compose(decompose) - A %.2e  A A^-1 - I %.2e\n", worst_orth, worst_sym, worst_rec, worst_round, worst_inv);
    // ten Newton rounds (the reference's default) leave ~1e-8 on the worst conditioned of these matrices
    return (worst_orth < 1e-6 && worst_sym < 1e-9 && worst_rec < 1e-6 && worst_round < 1e-6 && worst_inv < 1e-7) ? 0 : 1;
}
