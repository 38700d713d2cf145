// ref_shim.cpp -- C entry points onto the parts of the reference that compile stand-alone:
// headers/util.h and headers/pcgsolver/{pcg_solver,sparse_matrix,blas_wrapper}.h.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile straight from the reference checkout
// (no reference source is copied into this repository); the output goes to oracle/_ref/.
// It validates oracle/flip_oracle.c and generates the golden vectors under tests/golden/.
// The rest of the reference needs OpenVDB, Eigen, TBB and SFML, none of which exist in the
// build container, so it cannot be built here.
#include <cassert>
#include <climits>
#include <cstdint>
#include <vector>

#include "util.h"
#include "pcgsolver/pcg_solver.h"

extern "C" {

double ref_trilinear_hat_d(double x) { return calculate_trilinear_hat<double>(x); }
float ref_trilinear_hat_f(float x) { return calculate_trilinear_hat<float>(x); }
float ref_clamp_f(float a, float lo, float hi) { return clamp<float>(a, lo, hi); }
float ref_sqr_f(float a) { return sqr<float>(a); }
float ref_fraction_inside2_f(float l, float r) { return fraction_inside<float>(l, r); }
float ref_fraction_inside4_f(float bl, float br, float tl, float tr) { return fraction_inside<float>(bl, br, tl, tr); }
double ref_fraction_inside2_d(double l, double r) { return fraction_inside<double>(l, r); }

// PCGSolver<double>::solve on a CSR matrix (ascending columns per row).
int ref_pcg_solve(int32_t n, const int32_t *rowptr, const int32_t *col, const double *val, const double *rhs,
                  double tolerance_factor, int32_t max_iterations, double *result, double *residual_out,
                  int32_t *iterations_out) {
    SparseMatrixd A(n);
    for (int32_t i = 0; i < n; ++i)
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) A.set_element(i, col[q], val[q]);
    std::vector<double> b(rhs, rhs + n), x(n, 0.0);
    PCGSolver<double> solver;
    solver.set_solver_parameters(tolerance_factor, max_iterations);
    double res = 0;
    int it = 0;
    bool ok = solver.solve(A, b, x, res, it);
    for (int32_t i = 0; i < n; ++i) result[i] = x[i];
    *residual_out = res;
    *iterations_out = it;
    return ok ? 1 : 0;
}

// factor_modified_incomplete_cholesky0: returns the inverse diagonal of the factor.
void ref_mic0_invdiag(int32_t n, const int32_t *rowptr, const int32_t *col, const double *val, double *invdiag) {
    SparseMatrixd A(n);
    for (int32_t i = 0; i < n; ++i)
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) A.set_element(i, col[q], val[q]);
    SparseColumnLowerFactor<double> F;
    factor_modified_incomplete_cholesky0(A, F);
    for (int32_t i = 0; i < n; ++i) invdiag[i] = F.invdiag[i];
}

// multiply(FixedSparseMatrix, x, y)
void ref_spmv(int32_t n, const int32_t *rowptr, const int32_t *col, const double *val, const double *x, double *y) {
    SparseMatrixd A(n);
    for (int32_t i = 0; i < n; ++i)
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) A.set_element(i, col[q], val[q]);
    FixedSparseMatrixd F;
    F.construct_from_matrix(A);
    std::vector<double> xv(x, x + n), yv(n);
    multiply(F, xv, yv);
    for (int32_t i = 0; i < n; ++i) y[i] = yv[i];
}

} // extern "C"
