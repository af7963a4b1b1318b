// =====================================================================================================
//  CAPE ORACLE  --  TEST INFRASTRUCTURE ONLY (see cape_oracle.hpp for the full header / parity status:
//  **parity unpinned** -- the reference cannot be built here and holds no golden vectors for this path).
//
//  All arithmetic is written so that the sequence of IEEE-754 operations equals the reference's:
//  no FMA contraction (-ffp-contract=off), no fast-math, products of floats stay float (types.hpp:84),
//  reductions on 3-vectors in Eigen's order.
// =====================================================================================================
#include "cape_oracle.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <random>
#include <utility>

namespace cape_oracle {

// ---------------------------------------------------------------------------------------------------
// small helpers restating Eigen fixed-size semantics (Appendix A.2)
// ---------------------------------------------------------------------------------------------------

// ---- risk-measurement variants (oracle/variants.py): every third-party choice the restatement could not pin against
// the real Eigen / OpenCV / glibc is a compile-time switch, default 0 = the choice argued in SURVEY.md Appendix A.  The
// variants exist to MEASURE how much of the output depends on each choice; nothing ships them.
#ifndef CAPE_VAR_DOT_ORDER
#define CAPE_VAR_DOT_ORDER 0   // 1: a0 + (a1 + a2), the order without SSE2 packet vectorisation
#endif
#ifndef CAPE_VAR_NORMALIZE
#define CAPE_VAR_NORMALIZE 0   // 1: v *= 1 / sqrt(z) instead of v /= sqrt(z)
#endif
#ifndef CAPE_VAR_EIGEN
#define CAPE_VAR_EIGEN 0       // 1: cyclic Jacobi (another backward-stable solver), 2: closed form (computeDirect-style)
#endif
#ifndef CAPE_VAR_DET
#define CAPE_VAR_DET 0         // 1: Eigen's bruteforce_det3_helper association
#endif
#ifndef CAPE_VAR_GEMM
#define CAPE_VAR_GEMM 0        // cylinder covariance sum over k: 1 blocks of 256, 2 descending, 3 two interleaved accumulators
#endif
#ifndef CAPE_VAR_LIBM
#define CAPE_VAR_LIBM 0        // acos / atan2 of the histogram: 1 result + 1 ulp, 2 result - 1 ulp
#endif

// Vector3d::dot / squaredNorm with SSE2 Packet2d linear vectorisation: (a0 + a1) + a2
#if CAPE_VAR_DOT_ORDER
static inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }
static inline double sqnorm3(const double a[3]) { return a[0] * a[0] + (a[1] * a[1] + a[2] * a[2]); }
#else
static inline double dot3(const double a[3], const double b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline double sqnorm3(const double a[3]) { return (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]; }
#endif
static inline double libm_variant(double v)
{
#if CAPE_VAR_LIBM == 1
    return std::nextafter(v, std::numeric_limits<double>::infinity());
#elif CAPE_VAR_LIBM == 2
    return std::nextafter(v, -std::numeric_limits<double>::infinity());
#else
    return v;
#endif
}

// Eigen::MatrixBase::normalize(): z = squaredNorm(); if (z > 0) v /= sqrt(z)
void normalize3(double v[3])
{
    const double z = sqnorm3(v);
    if (z > 0)
    {
        const double s = std::sqrt(z);
#if CAPE_VAR_NORMALIZE
        const double r = 1.0 / s;
        v[0] *= r;
        v[1] *= r;
        v[2] *= r;
#else
        v[0] /= s;
        v[1] /= s;
        v[2] /= s;
#endif
    }
}

static inline void cross3(const double a[3], const double b[3], double out[3])
{
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}

// utils::double_equal, src/utils/distance_utils.cpp:11 (epsilon defaults to DBL_EPSILON, distance_utils.hpp:24-26)
static inline bool double_equal(double a, double b, double eps = std::numeric_limits<double>::epsilon())
{
    return std::abs(a - b) <= eps;
}

// ---------------------------------------------------------------------------------------------------
// utils::get_depth_quantization, src/utils/covariances.cpp:12-19 ; constants src/parameters.hpp:16-18
// ---------------------------------------------------------------------------------------------------
double depth_quantization(double depth)
{
    static const double depthSigmaError = 2.73 * ((1.0 / 1000.0) * (1.0 / 1000.0)); // sigmaE * SQR(1/1000)
    constexpr double depthSigmaMultiplier = 0.74 / 1000.0;
    constexpr double depthSigmaMargin = -0.53;
    return std::max(depthSigmaMargin + depthSigmaMultiplier * depth + depthSigmaError * (depth * depth), 0.5);
}

// ---------------------------------------------------------------------------------------------------
// Eigen 3.4.0 SelfAdjointEigenSolver<Matrix3d>::compute (iterative path), SURVEY.md Appendix A.1.
// Call sites: plane_segment.cpp:251, cylinder_segment.cpp:92.
// evecs[r][c]: column c is the eigenvector of evals[c] (ascending).
// ---------------------------------------------------------------------------------------------------
static inline double eigen_hypot(double x, double y)
{
    // numext::hypot -> positive_real_hypot(|x|,|y|)
    x = std::abs(x);
    y = std::abs(y);
    if (std::isinf(x) || std::isinf(y))
        return std::numeric_limits<double>::infinity();
    if (std::isnan(x) || std::isnan(y))
        return std::numeric_limits<double>::quiet_NaN();
    const double p = (x < y) ? y : x; // numext::maxi(x,y)
    if (p == 0.0)
        return 0.0;
    const double qp = ((x < y) ? x : y) / p; // numext::mini(y,x) / p
    return p * std::sqrt(1.0 + qp * qp);
}

static inline void make_givens(double p, double q, double& c, double& s)
{
    // JacobiRotation<double>::makeGivens (real case)
    if (q == 0.0)
    {
        c = p < 0.0 ? -1.0 : 1.0;
        s = 0.0;
    }
    else if (p == 0.0)
    {
        c = 0.0;
        s = q < 0.0 ? 1.0 : -1.0;
    }
    else if (std::abs(p) > std::abs(q))
    {
        const double t = q / p;
        double u = std::sqrt(1.0 + t * t);
        if (p < 0.0)
            u = -u;
        c = 1.0 / u;
        s = -t * c;
    }
    else
    {
        const double t = p / q;
        double u = std::sqrt(1.0 + t * t);
        if (q < 0.0)
            u = -u;
        s = -1.0 / u;
        c = -t * s;
    }
}

static void tridiagonal_qr_step(double* diag, double* subdiag, int start, int end, double Q[3][3])
{
    // Wilkinson shift
    const double td = (diag[end - 1] - diag[end]) * 0.5;
    const double e = subdiag[end - 1];
    double mu = diag[end];
    if (td == 0.0)
    {
        mu -= std::abs(e);
    }
    else if (e != 0.0)
    {
        const double e2 = e * e;
        const double h = eigen_hypot(td, e);
        if (e2 == 0.0)
            mu -= e / ((td + (td > 0.0 ? h : -h)) / e);
        else
            mu -= e2 / (td + (td > 0.0 ? h : -h));
    }

    double x = diag[start] - mu;
    double z = subdiag[start];
    for (int k = start; k < end && z != 0.0; ++k)
    {
        double c, s;
        make_givens(x, z, c, s);

        // T = G' T G
        const double sdk = s * diag[k] + c * subdiag[k];
        const double dkp1 = s * subdiag[k] + c * diag[k + 1];

        diag[k] = c * (c * diag[k] - s * subdiag[k]) - s * (c * subdiag[k] - s * diag[k + 1]);
        diag[k + 1] = s * sdk + c * dkp1;
        subdiag[k] = c * sdk - s * dkp1;

        if (k > start)
            subdiag[k - 1] = c * subdiag[k - 1] - s * z;

        // chase the bulge
        x = subdiag[k];
        if (k < end - 1)
        {
            z = -s * subdiag[k + 1];
            subdiag[k + 1] = c * subdiag[k + 1];
        }

        // Q = Q * G : q.applyOnTheRight(k, k+1, rot)
        for (int r = 0; r < 3; ++r)
        {
            const double xi = Q[r][k];
            const double yi = Q[r][k + 1];
            Q[r][k] = c * xi - s * yi;
            Q[r][k + 1] = s * xi + c * yi;
        }
    }
}

static void self_adjoint_eigen3_iterative(const double lower[3][3], double evals[3], double evecs[3][3], int* iterations)
{
    double m00 = lower[0][0], m10 = lower[1][0], m11 = lower[1][1];
    double m20 = lower[2][0], m21 = lower[2][1], m22 = lower[2][2];

    // map coefficients to [-1,1]
    double scale = std::abs(m00);
    scale = std::max(scale, std::abs(m10));
    scale = std::max(scale, std::abs(m20));
    scale = std::max(scale, std::abs(m11));
    scale = std::max(scale, std::abs(m21));
    scale = std::max(scale, std::abs(m22));
    if (scale == 0.0)
        scale = 1.0;
    m00 /= scale;
    m10 /= scale;
    m11 /= scale;
    m20 /= scale;
    m21 /= scale;
    m22 /= scale;

    // tridiagonalization_inplace_selector<MatrixType,3,false>::run
    double diag[3], sub[2];
    double Q[3][3];
    const double tol = std::numeric_limits<double>::min();
    diag[0] = m00;
    const double v1norm2 = m20 * m20;
    if (v1norm2 <= tol)
    {
        diag[1] = m11;
        diag[2] = m22;
        sub[0] = m10;
        sub[1] = m21;
        Q[0][0] = 1; Q[0][1] = 0; Q[0][2] = 0;
        Q[1][0] = 0; Q[1][1] = 1; Q[1][2] = 0;
        Q[2][0] = 0; Q[2][1] = 0; Q[2][2] = 1;
    }
    else
    {
        const double beta = std::sqrt(m10 * m10 + v1norm2);
        const double invBeta = 1.0 / beta;
        const double m01 = m10 * invBeta;
        const double m02 = m20 * invBeta;
        const double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
        diag[1] = m11 + m02 * q;
        diag[2] = m22 - m02 * q;
        sub[0] = beta;
        sub[1] = m21 - m01 * q;
        Q[0][0] = 1; Q[0][1] = 0;   Q[0][2] = 0;
        Q[1][0] = 0; Q[1][1] = m01; Q[1][2] = m02;
        Q[2][0] = 0; Q[2][1] = m02; Q[2][2] = -m01;
    }

    // computeFromTridiagonal_impl
    const int n = 3;
    int end = n - 1;
    int start = 0;
    int iter = 0;
    const int maxIterations = 30;
    const double considerAsZero = std::numeric_limits<double>::min();
    const double precision_inv = 1.0 / std::numeric_limits<double>::epsilon();
    while (end > 0)
    {
        for (int i = start; i < end; ++i)
        {
            if (std::abs(sub[i]) < considerAsZero)
            {
                sub[i] = 0.0;
            }
            else
            {
                const double scaled_subdiag = precision_inv * sub[i];
                if (scaled_subdiag * scaled_subdiag <= (std::abs(diag[i]) + std::abs(diag[i + 1])))
                    sub[i] = 0.0;
            }
        }
        while (end > 0 && sub[end - 1] == 0.0)
            end--;
        if (end <= 0)
            break;
        iter++;
        if (iter > maxIterations * n)
            break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.0)
            start--;
        tridiagonal_qr_step(diag, sub, start, end, Q);
    }
    if (iterations)
        *iterations = iter;

    if (iter <= maxIterations * n)
    {
        // selection sort, ascending
        for (int i = 0; i < n - 1; ++i)
        {
            int k = 0;
            double best = diag[i];
            for (int j = 1; j < n - i; ++j)
            {
                if (diag[i + j] < best)
                {
                    best = diag[i + j];
                    k = j;
                }
            }
            if (k > 0)
            {
                std::swap(diag[i], diag[k + i]);
                for (int r = 0; r < 3; ++r)
                    std::swap(Q[r][i], Q[r][k + i]);
            }
        }
    }

    for (int i = 0; i < 3; ++i)
    {
        evals[i] = diag[i] * scale;
        for (int r = 0; r < 3; ++r)
            evecs[r][i] = Q[r][i];
    }
}

#if CAPE_VAR_EIGEN == 1
// variant: cyclic Jacobi sweeps to convergence, eigenvalues ascending.  A different backward-stable algorithm: its
// eigenvectors agree with the QR iteration to a few ulp (times the inverse eigenvalue gap), which is the size of
// disagreement to expect from any other correct implementation of the solver.
static void self_adjoint_eigen3_variant(const double lower[3][3], double evals[3], double evecs[3][3])
{
    double a[3][3] = {{lower[0][0], lower[1][0], lower[2][0]}, {lower[1][0], lower[1][1], lower[2][1]}, {lower[2][0], lower[2][1], lower[2][2]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep)
    {
        const double off = std::abs(a[0][1]) + std::abs(a[0][2]) + std::abs(a[1][2]);
        if (off == 0.0)
            break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q)
            {
                if (a[p][q] == 0.0)
                    continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; ++k)
                {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - sn * akq;
                    a[k][q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k)
                {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - sn * aqk;
                    a[q][k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k)
                {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - sn * vkq;
                    v[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    int idx[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (a[idx[j]][idx[j]] < a[idx[i]][idx[i]])
                std::swap(idx[i], idx[j]);
    for (int c = 0; c < 3; ++c)
    {
        evals[c] = a[idx[c]][idx[c]];
        for (int r = 0; r < 3; ++r)
            evecs[r][c] = v[r][idx[c]];
    }
}
#elif CAPE_VAR_EIGEN == 2
// variant: closed form in the manner of SelfAdjointEigenSolver::computeDirect (shift by the mean, scale, trigonometric
// roots of the characteristic polynomial, eigenvectors from cross products of rows of A - lambda I).  Documented by Eigen
// as less accurate than the iterative solver when eigenvalues are close: the pessimistic end of the range.
static void self_adjoint_eigen3_variant(const double lower[3][3], double evals[3], double evecs[3][3])
{
    double A[3][3] = {{lower[0][0], lower[1][0], lower[2][0]}, {lower[1][0], lower[1][1], lower[2][1]}, {lower[2][0], lower[2][1], lower[2][2]}};
    const double shift = (A[0][0] + A[1][1] + A[2][2]) / 3.0;
    for (int i = 0; i < 3; ++i)
        A[i][i] -= shift;
    double scale = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            scale = std::max(scale, std::abs(A[i][j]));
    if (scale > 0)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                A[i][j] /= scale;
    // roots of x^3 - c2 x^2 + c1 x - c0 with c2 = trace = 0 after the shift
    const double c0 = A[0][0] * A[1][1] * A[2][2] + 2.0 * A[1][0] * A[2][0] * A[2][1] - A[0][0] * A[2][1] * A[2][1] -
                      A[1][1] * A[2][0] * A[2][0] - A[2][2] * A[1][0] * A[1][0];
    const double c1 = A[0][0] * A[1][1] - A[1][0] * A[1][0] + A[0][0] * A[2][2] - A[2][0] * A[2][0] + A[1][1] * A[2][2] - A[2][1] * A[2][1];
    const double a_over_3 = -c1 / 3.0 > 0 ? -c1 / 3.0 : 0.0;
    const double half_b = 0.5 * c0;
    double q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
    if (q < 0)
        q = 0;
    const double rho = std::sqrt(a_over_3);
    const double theta = std::atan2(std::sqrt(q), half_b) / 3.0;
    const double ct = std::cos(theta), st = std::sin(theta);
    double roots[3] = {-rho * (ct + std::sqrt(3.0) * st), -rho * (ct - std::sqrt(3.0) * st), 2.0 * rho * ct};
    std::sort(roots, roots + 3);
    auto kernel = [&](double lambda, double out[3]) {
        double B[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                B[i][j] = A[i][j] - (i == j ? lambda : 0.0);
        double best = -1.0;
        for (int i = 0; i < 3; ++i)
            for (int j = i + 1; j < 3; ++j)
            {
                double c[3];
                cross3(B[i], B[j], c);
                const double n2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
                if (n2 > best)
                {
                    best = n2;
                    out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
                }
            }
        const double n = std::sqrt(best);
        if (n > 0)
        {
            out[0] /= n; out[1] /= n; out[2] /= n;
        }
        else
        {
            out[0] = 1; out[1] = 0; out[2] = 0;
        }
    };
    double v0[3], v2[3], v1[3];
    kernel(roots[0], v0);
    kernel(roots[2], v2);
    // orthogonalise v2 against v0, v1 = v2 x v0
    const double dp = v0[0] * v2[0] + v0[1] * v2[1] + v0[2] * v2[2];
    for (int k = 0; k < 3; ++k)
        v2[k] -= dp * v0[k];
    const double n2 = std::sqrt(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]);
    if (n2 > 0)
        for (int k = 0; k < 3; ++k)
            v2[k] /= n2;
    cross3(v2, v0, v1);
    for (int r = 0; r < 3; ++r)
    {
        evecs[r][0] = v0[r];
        evecs[r][1] = v1[r];
        evecs[r][2] = v2[r];
    }
    for (int c = 0; c < 3; ++c)
        evals[c] = roots[c] * scale + shift;
}
#endif

void self_adjoint_eigen3(const double lower[3][3], double evals[3], double evecs[3][3], int* iterations)
{
#if CAPE_VAR_EIGEN
    if (iterations)
        *iterations = 0;
    self_adjoint_eigen3_variant(lower, evals, evecs);
#else
    self_adjoint_eigen3_iterative(lower, evals, evecs, iterations);
#endif
}

// Matrix3d::determinant (first-row expansion, Appendix A.2)
static inline double det3(const double m[3][3])
{
#if CAPE_VAR_DET
    // Eigen's bruteforce_det3_helper(m,0,1,2) + (m,1,2,0) + (m,2,0,1): m(0,a) * (m(1,b) m(2,c) - m(1,c) m(2,b))
    auto h = [&](int a, int b, int c) { return m[0][a] * (m[1][b] * m[2][c] - m[1][c] * m[2][b]); };
    return h(0, 1, 2) + h(1, 2, 0) + h(2, 0, 1);
#endif
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

// Matrix3d::inverse, cofactor method (Appendix A.3)
static inline double cof3(const double m[3][3], int i, int j)
{
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
}
static void inverse3(const double m[3][3], double r[3][3])
{
    const double c0[3] = {cof3(m, 0, 0), cof3(m, 1, 0), cof3(m, 2, 0)};
    const double det = (c0[0] * m[0][0] + c0[1] * m[1][0]) + c0[2] * m[2][0];
    const double invdet = 1.0 / det;
    const double c01 = cof3(m, 0, 1) * invdet;
    const double c11 = cof3(m, 1, 1) * invdet;
    const double c02 = cof3(m, 0, 2) * invdet;
    r[1][2] = cof3(m, 2, 1) * invdet;
    r[2][1] = cof3(m, 1, 2) * invdet;
    r[2][2] = cof3(m, 2, 2) * invdet;
    r[1][0] = c01;
    r[1][1] = c11;
    r[2][0] = c02;
    r[0][0] = c0[0] * invdet;
    r[0][1] = c0[1] * invdet;
    r[0][2] = c0[2] * invdet;
}

// ---------------------------------------------------------------------------------------------------
// PlaneCoordinates semantics, src/coordinates/plane_coordinates.hpp:16-40: every ctor / copy / assignment
// re-normalises the normal.  The helpers below are called exactly where the reference constructs, copies
// or assigns a PlaneCoordinates (vector reallocations are assumed not to happen = steady-state capacity,
// SURVEY.md 3.3 / A16).
// ---------------------------------------------------------------------------------------------------
static inline void plane_coord_construct(PlaneSeg& s, const double n[3], double d)
{
    // _parametrization = PlaneCoordinates(n, d): ctor normalises, operator= normalises again
    double t[3] = {n[0], n[1], n[2]};
    normalize3(t); // PlaneCoordinates(const vector3&, double)
    normalize3(t); // operator=
    s.normal[0] = t[0];
    s.normal[1] = t[1];
    s.normal[2] = t[2];
    s.d = d;
}

// Plane_Segment(const Plane_Segment&), plane_segment.cpp:18-36
static inline PlaneSeg copy_segment(const PlaneSeg& s)
{
    PlaneSeg c = s;
    normalize3(c.normal); // PlaneCoordinates(const PlaneCoordinates&)
    return c;
}

// Plane_Segment::clear_plane_parameters, plane_segment.cpp:289-310
static inline void clear_segment(PlaneSeg& s) { s = PlaneSeg(); }

// Plane_Segment::fit_plane, plane_segment.cpp:232-284 ; Huygens covariance :205-230
void fit_plane(PlaneSeg& s)
{
    s.planar = false;
    const double oneOverCount = 1.0 / static_cast<double>(s.n);

    s.centroid[0] = s.Sx * oneOverCount;
    s.centroid[1] = s.Sy * oneOverCount;
    s.centroid[2] = s.Sz * oneOverCount;

    const double xx = std::max(0.0, s.Sxs - (s.Sx * s.Sx) * oneOverCount);
    const double yy = std::max(0.0, s.Sys - (s.Sy * s.Sy) * oneOverCount);
    const double zz = std::max(0.0, s.Szs - (s.Sz * s.Sz) * oneOverCount);
    const double xy = s.Sxy - s.Sx * s.Sy * oneOverCount;
    const double xz = s.Szx - s.Sx * s.Sz * oneOverCount;
    const double yz = s.Syz - s.Sy * s.Sz * oneOverCount;

    const double cov[3][3] = {{xx, xy, xz}, {xy, yy, yz}, {xz, yz, zz}};
    if (double_equal(det3(cov), 0.0))
        return;

    double evals[3], evecs[3][3];
    self_adjoint_eigen3(cov, evals, evecs, nullptr);
    const double ev0 = std::abs(evals[0]);
    const double ev1 = std::abs(evals[1]);

    // eigenVector.normalized()
    double normal[3] = {evecs[0][0], evecs[1][0], evecs[2][0]};
    {
        const double z = sqnorm3(normal);
        if (z > 0)
        {
            const double sq = std::sqrt(z);
#if CAPE_VAR_NORMALIZE
            const double rq = 1.0 / sq;
            normal[0] *= rq;
            normal[1] *= rq;
            normal[2] *= rq;
#else
            normal[0] /= sq;
            normal[1] /= sq;
            normal[2] /= sq;
#endif
        }
    }
    const double d = -dot3(normal, s.centroid);

    if (d <= 0)
    {
        const double neg[3] = {-normal[0], -normal[1], -normal[2]};
        plane_coord_construct(s, neg, -d);
    }
    else
    {
        plane_coord_construct(s, normal, d);
    }

    s.mse = ev0 * oneOverCount;
    s.score = ev1 / std::max(ev0, 1e-6);
    s.planar = true;
}

// Plane_Segment::expand_segment, plane_segment.cpp:170-190
static inline void expand_segment(PlaneSeg& a, const PlaneSeg& b)
{
    a.Sx += b.Sx;
    a.Sy += b.Sy;
    a.Sz += b.Sz;
    a.Sxs += b.Sxs;
    a.Sys += b.Sys;
    a.Szs += b.Szs;
    a.Sxy += b.Sxy;
    a.Syz += b.Syz;
    a.Szx += b.Szx;
    a.n += b.n;
}

// Plane_Segment::can_be_merged, plane_segment.cpp:322-326
bool can_be_merged(const PlaneSeg& a, const PlaneSeg& p, double maxMatchDistance)
{
    static const double maximumMergeAngle = std::cos(static_cast<double>(kMaxAngleForMerge_d) * M_PI / 180.0);
    const double cosAngle = dot3(a.normal, p.normal);
    if (!(cosAngle > maximumMergeAngle))
        return false;
    const double dist = dot3(a.normal, p.centroid) + a.d;
    return std::abs(dist) < maxMatchDistance;
}

// ---------------------------------------------------------------------------------------------------
// Histogram<20>, src/features/primitives/histogram.hpp:20-133
// ---------------------------------------------------------------------------------------------------
struct Histogram20
{
    static constexpr size_t Size = 20;
    unsigned hist[Size * Size];
    std::vector<int> bins;
    unsigned pointCount = 0;

    void reset()
    {
        std::fill(hist, hist + Size * Size, 0u);
        bins.clear();
    }

    // :35-62 ; points(i,0)=acos(-nz), points(i,1)=atan2(nx,ny)
    void init(const std::vector<double>& p0, const std::vector<double>& p1, const std::vector<uint8_t>& mask)
    {
        constexpr double minX = 0;
        constexpr double minY = -M_PI;
        constexpr double maxXminX = M_PI - minX;
        constexpr double maxYminY = M_PI - minY;
        pointCount = static_cast<unsigned>(p0.size());
        bins.assign(pointCount, -1);
        for (unsigned i = 0; i < pointCount; ++i)
        {
            if (mask[i])
            {
                const int xQ = static_cast<int>(std::floor((Size - 1) * (p0[i] - minX) / maxXminX));
                int yQ = 0;
                if (xQ > 0)
                    yQ = static_cast<int>(std::floor((Size - 1) * (p1[i] - minY) / maxYminY));
                const unsigned bin = static_cast<unsigned>(yQ * Size + xQ);
                bins[i] = static_cast<int>(bin);
                hist[bin] += 1;
            }
        }
    }

    // :69-98
    std::vector<unsigned> points_from_most_frequent_bin() const
    {
        int mostFrequentBin = -1;
        unsigned maxOcc = 0;
        for (unsigned i = 0; i < Size * Size; ++i)
        {
            if (hist[i] > maxOcc)
            {
                mostFrequentBin = static_cast<int>(i);
                maxOcc = hist[i];
            }
        }
        std::vector<unsigned> ids;
        if (mostFrequentBin >= 0)
        {
            for (unsigned i = 0; i < pointCount; ++i)
                if (bins[i] == mostFrequentBin)
                    ids.push_back(i);
        }
        return ids;
    }

    // :103-113  (quirk: bin becomes 1, not -1)
    void remove_point(unsigned id)
    {
        if (hist[bins[id]] != 0)
            hist[bins[id]] -= 1;
        bins[id] = 1;
    }
};

// ---------------------------------------------------------------------------------------------------
// OpenCV 3x3 morphology on the cell grid (Appendix A.4)
// ---------------------------------------------------------------------------------------------------
enum class Border
{
    Zero,   // BORDER_CONSTANT, Scalar(0)
    Ignore, // morphologyDefaultBorderValue(): outside never changes the result
};

static void morph3(const std::vector<uint8_t>& src, std::vector<uint8_t>& dst, int rows, int cols, bool cross,
                   bool erode, Border border)
{
    std::vector<uint8_t> out(src.size());
    for (int r = 0; r < rows; ++r)
    {
        for (int c = 0; c < cols; ++c)
        {
            uint8_t acc = erode ? 255 : 0;
            for (int dr = -1; dr <= 1; ++dr)
            {
                for (int dc = -1; dc <= 1; ++dc)
                {
                    if (cross && dr != 0 && dc != 0)
                        continue;
                    const int rr = r + dr, cc = c + dc;
                    uint8_t v;
                    if (rr < 0 || rr >= rows || cc < 0 || cc >= cols)
                    {
                        if (border == Border::Ignore)
                            continue;
                        v = 0;
                    }
                    else
                    {
                        v = src[rr * cols + cc];
                    }
                    acc = erode ? std::min(acc, v) : std::max(acc, v);
                }
            }
            out[r * cols + c] = acc;
        }
    }
    dst.swap(out);
}

// ---------------------------------------------------------------------------------------------------
// Cylinder_Segment, src/features/primitives/cylinder_segment.cpp:35-322
// ---------------------------------------------------------------------------------------------------
struct CylinderSeg
{
    double axis[3] = {0, 0, 0};
    unsigned cellActivatedCount = 0;
    unsigned segmentCount = 0;
    std::vector<unsigned> local2global;
    std::vector<std::vector<uint8_t>> inliers;
    std::vector<double> mse;
    std::vector<double> radius;
    std::vector<std::array<double, 3>> centers;
};

struct Rng
{
    // utils::Random, src/utils/random.hpp:17-58 ; thread_local engine re-created per find_primitives call because
    // the reference runs it on a fresh std::async thread each frame (rgbd_slam.cpp:291), seed 0 = MAKE_DETERMINISTIC
    std::mt19937 engine{0u};
    std::uniform_real_distribution<double> dist{0.0, 1.0};
    explicit Rng(unsigned seed = 0u) : engine(seed) {}
    double next_double() { return dist(engine); }
    unsigned next_uint(unsigned maxValue)
    {
        return 0u + static_cast<unsigned>(std::floor(next_double() * (maxValue - 0u)));
    }
};

double mt19937_first_double(unsigned seed, int index)
{
    std::mt19937 e(seed);
    std::uniform_real_distribution<double> d(0.0, 1.0);
    double v = 0;
    for (int i = 0; i <= index; ++i)
        v = d(e);
    return v;
}

// cylinder_segment.cpp:227-322
static size_t run_ransac_loop(unsigned maximumIterations, const std::vector<unsigned>& idsLeft,
                              const std::vector<double>& N, // 3 x n, column-major
                              const std::vector<double>& C, // projected centroids 3 x n
                              const std::vector<uint8_t>& idsLeftMask, std::vector<uint8_t>& isInlierFinal,
                              unsigned cellActivatedCount, Rng& rng)
{
    if (idsLeft.size() < 3)
        return 0;

    const unsigned planeIdsLeft = static_cast<unsigned>(idsLeft.size());
    const unsigned inliersAcceptedCount = static_cast<unsigned>(std::floor(0.9 * planeIdsLeft));

    constexpr float maximumSqrtDistance = kCylSqrtMaxDist;
    double minHypothesisDist = maximumSqrtDistance * static_cast<float>(planeIdsLeft);
    std::vector<unsigned> finalInlierIndexes;

    for (unsigned iteration = 0; iteration < maximumIterations; ++iteration)
    {
        const unsigned id1 = idsLeft[rng.next_uint(planeIdsLeft)];
        const unsigned id2 = idsLeft[rng.next_uint(planeIdsLeft)];
        const unsigned id3 = idsLeft[rng.next_uint(planeIdsLeft)];
        const double* n1 = &N[3 * id1];
        const double* n2 = &N[3 * id2];
        const double* n3 = &N[3 * id3];
        const double* c1 = &C[3 * id1];
        const double* c2 = &C[3 * id2];
        const double* c3 = &C[3 * id3];

        double sumN[3], sumC[3];
        for (int k = 0; k < 3; ++k)
        {
            sumN[k] = (n1[k] + n2[k]) + n3[k];
            sumC[k] = (c1[k] + c2[k]) + c3[k];
        }

        const double a = 1.0 - sqnorm3(sumN) / 9.0;
        double prod[3];
        for (int k = 0; k < 3; ++k)
            prod[k] = (n1[k] * c1[k] + n2[k] * c2[k]) + n3[k] * c3[k];
        const double b = ((prod[0] + prod[1]) + prod[2]) / 3.0 - (dot3(sumN, sumC) / 9.0);
        const double radius = b / a;
        const double oneOverRadiusSquared = 1.0 / (radius * radius);
        double center[3];
        for (int k = 0; k < 3; ++k)
            center[k] = (sumC[k] - radius * sumN[k]) / 3.0;

        std::vector<unsigned> inlierIndexes;
        double dist = 0.0;
        for (unsigned i = 0; i < cellActivatedCount; ++i)
        {
            if (!idsLeftMask[i])
                continue;
            double v[3];
            for (int k = 0; k < 3; ++k)
                v[k] = (C[3 * i + k] - radius * N[3 * i + k]) - center[k];
            const double distance = sqnorm3(v) * oneOverRadiusSquared;
            if (distance < maximumSqrtDistance)
            {
                dist += distance;
                inlierIndexes.push_back(i);
            }
            else
            {
                dist += maximumSqrtDistance;
            }
        }

        if (dist < minHypothesisDist)
        {
            minHypothesisDist = dist;
            finalInlierIndexes.swap(inlierIndexes);
            // early stop (quirk: tests the swapped-out previous best, :308-312)
            if (inlierIndexes.size() > inliersAcceptedCount)
                break;
        }
    }

    std::fill(isInlierFinal.begin(), isInlierFinal.end(), 0);
    for (const unsigned idx : finalInlierIndexes)
        isInlierFinal[idx] = 1;
    return finalInlierIndexes.size();
}

// cylinder_segment.cpp:35-225
static CylinderSeg make_cylinder_segment(const std::vector<PlaneSeg>& planeGrid, const std::vector<uint8_t>& isActivated,
                                         unsigned cellActivatedCount, Rng& rng)
{
    CylinderSeg cs;
    cs.cellActivatedCount = cellActivatedCount;
    const size_t samplesCount = isActivated.size();
    const unsigned n = cellActivatedCount;
    cs.local2global.assign(n, 0);

    std::vector<double> normals(3 * 2 * n); // 3 x 2n column-major
    std::vector<double> centroids(3 * n);

    unsigned j = 0;
    for (unsigned i = 0; i < samplesCount; ++i)
    {
        if (isActivated[i])
        {
            for (int k = 0; k < 3; ++k)
            {
                normals[3 * j + k] = planeGrid[i].normal[k];
                centroids[3 * j + k] = planeGrid[i].centroid[k];
            }
            cs.local2global[j] = i;
            ++j;
        }
    }
    for (unsigned i = 0; i < samplesCount; ++i)
    {
        if (isActivated[i])
        {
            for (int k = 0; k < 3; ++k)
                normals[3 * j + k] = -planeGrid[i].normal[k];
            ++j;
        }
    }

    // cov = (M * M^T) / (cols - 1) ; summation order of the dynamic GEMM is unpinned: ascending column index
    double cov[3][3];
    const unsigned cols = 2 * n;
    for (int r = 0; r < 3; ++r)
    {
        for (int c = 0; c < 3; ++c)
        {
            double acc = 0.0;
#if CAPE_VAR_GEMM == 1
            for (unsigned k0 = 0; k0 < cols; k0 += 256) // depth blocking of a GEMM kernel: block sums added to C
            {
                double blk = 0.0;
                for (unsigned k = k0; k < std::min(cols, k0 + 256u); ++k)
                    blk += normals[3 * k + r] * normals[3 * k + c];
                acc += blk;
            }
#elif CAPE_VAR_GEMM == 2
            for (unsigned k = cols; k-- > 0;)
                acc += normals[3 * k + r] * normals[3 * k + c];
#elif CAPE_VAR_GEMM == 3
            double even = 0.0, odd = 0.0; // a depth-vectorised inner product: two accumulators, reduced at the end
            for (unsigned k = 0; k + 1 < cols; k += 2)
            {
                even += normals[3 * k + r] * normals[3 * k + c];
                odd += normals[3 * (k + 1) + r] * normals[3 * (k + 1) + c];
            }
            if (cols & 1u)
                even += normals[3 * (cols - 1) + r] * normals[3 * (cols - 1) + c];
            acc = even + odd;
#else
            for (unsigned k = 0; k < cols; ++k)
                acc += normals[3 * k + r] * normals[3 * k + c];
#endif
            cov[r][c] = acc / static_cast<double>(cols - 1);
        }
    }

    double evals[3], evecs[3][3];
    self_adjoint_eigen3(cov, evals, evecs, nullptr);
    const double score = evals[2] / evals[0];
    if (score < kCylMinScore)
        return cs;

    const double axis[3] = {evecs[0][0], evecs[1][0], evecs[2][0]};
    cs.axis[0] = axis[0];
    cs.axis[1] = axis[1];
    cs.axis[2] = axis[2];

    std::vector<double> N(3 * n), PC(3 * n);
    for (unsigned i = 0; i < n; ++i)
    {
        const double cdt = dot3(axis, &centroids[3 * i]);
        for (int k = 0; k < 3; ++k)
            PC[3 * i + k] = centroids[3 * i + k] - cdt * axis[k];
        const double ndt = dot3(axis, &normals[3 * i]);
        double pn[3];
        for (int k = 0; k < 3; ++k)
            pn[k] = normals[3 * i + k] - ndt * axis[k];
        const double nrm = std::sqrt(sqnorm3(pn));
        for (int k = 0; k < 3; ++k)
            N[3 * i + k] = pn[k] / nrm;
    }

    static const unsigned maximumIterations =
            static_cast<unsigned>(logf(1.0f - kCylPSuccess) / logf(1.0f - powf(kCylInlierProp, 3.0f)));

    unsigned planeSegmentsLeft = n;
    std::vector<uint8_t> idsLeftMask(n, 1);
    std::vector<unsigned> idsLeft;
    for (unsigned i = 0; i < n; ++i)
        idsLeft.push_back(i);

    const size_t minimumCellActivated =
            static_cast<unsigned>(kMinActivatedProportion * static_cast<double>(samplesCount));
    while (planeSegmentsLeft > minimumCellActivated && planeSegmentsLeft > 0.1 * n)
    {
        std::vector<uint8_t> isInlierFinal(n, 0);
        const size_t maxInliersCount = run_ransac_loop(maximumIterations, idsLeft, N, PC, idsLeftMask, isInlierFinal, n, rng);
        if (maxInliersCount < 6)
            break;

        double b = 0;
        double sumN[3] = {0, 0, 0}, sumC[3] = {0, 0, 0};
        idsLeft.clear();
        for (unsigned i = 0; i < n; ++i)
        {
            if (isInlierFinal[i])
            {
                idsLeftMask[i] = 0;
                planeSegmentsLeft--;
                for (int k = 0; k < 3; ++k)
                {
                    sumN[k] += N[3 * i + k];
                    sumC[k] += PC[3 * i + k];
                }
                b += (N[3 * i] * PC[3 * i] + N[3 * i + 1] * PC[3 * i + 1]) + N[3 * i + 2] * PC[3 * i + 2];
            }
            else if (idsLeftMask[i])
            {
                idsLeft.push_back(i);
            }
        }

        const double oneOverSq = 1.0 / static_cast<double>(maxInliersCount * maxInliersCount);
        const double a = 1 - sqnorm3(sumN) * oneOverSq;
        b /= static_cast<double>(maxInliersCount);
        b -= dot3(sumN, sumC) * oneOverSq;
        double radius = b / a;
        double center[3];
        for (int k = 0; k < 3; ++k)
            center[k] = (sumC[k] - radius * sumN[k]) / static_cast<double>(maxInliersCount);
        if (radius < 0)
            radius = -radius;

        cs.segmentCount += 1;
        cs.radius.push_back(radius);
        cs.centers.push_back({center[0], center[1], center[2]});
        cs.inliers.push_back(isInlierFinal);

        double P1[3], P2[3], P21[3];
        for (int k = 0; k < 3; ++k)
        {
            P1[k] = center[k];
            P2[k] = center[k] + axis[k];
        }
        for (int k = 0; k < 3; ++k)
            P21[k] = P2[k] - P1[k];
        const double P1P2d = std::sqrt(sqnorm3(P21));

        double mse = 0;
        for (unsigned i = 0; i < n; ++i)
        {
            if (isInlierFinal[i])
            {
                double w[3], cr[3];
                for (int k = 0; k < 3; ++k)
                    w[k] = centroids[3 * i + k] - P2[k];
                cross3(P21, w, cr);
                const double t = std::sqrt(sqnorm3(cr)) / P1P2d - radius;
                mse += t * t;
            }
        }
        mse /= static_cast<double>(maxInliersCount);
        cs.mse.push_back(mse);
    }
    return cs;
}

// ---------------------------------------------------------------------------------------------------
// Oracle
// ---------------------------------------------------------------------------------------------------
Oracle::Oracle(const Config& cfg) : cfg_(cfg)
{
    hCells_ = cfg.width / static_cast<int>(kCell);
    vCells_ = cfg.height / static_cast<int>(kCell);
    totalCells_ = hCells_ * vCells_;
    planeGrid.assign(totalCells_, PlaneSeg());
    cellTols.assign(totalCells_, 0.0f);

    // K^-1 = Parameters::get_camera_1_intrinsics().inverse(), point_coordinates.cpp:81 ; Appendix A.3
    const double K[3][3] = {{cfg.fx, 0, cfg.cx}, {0, cfg.fy, cfg.cy}, {0, 0, 1}};
    double Ki[3][3];
    inverse3(K, Ki);
    k00_ = Ki[0][0];
    k02_ = Ki[0][2];
    k11_ = Ki[1][1];
    k12_ = Ki[1][2];

    // Depth_Map_Transformation::init_matrices, depth_map_transformation.cpp:147-173
    cellMap_.assign(static_cast<size_t>(cfg.width) * cfg.height, 0);
    for (int row = 0; row < cfg.height; ++row)
    {
        const unsigned cellR = row / kCell, localR = row % kCell;
        for (int col = 0; col < cfg.width; ++col)
        {
            const unsigned cellC = col / kCell, localC = col % kCell;
            cellMap_[static_cast<size_t>(row) * cfg.width + col] =
                    static_cast<int>((cellR * hCells_ + cellC) * kPtsPerCell + localR * kCell + localC);
        }
    }
}

void Oracle::back_project(double col, double row, double z, double out[3]) const
{
    // transform_screen_to_camera: (K^-1 * [u v 1]).head<2>() evaluated as K^-1[:, :2]*[u v] + K^-1[:,2]
    // (Eigen homogeneous product); k01 = k10 = 0 structurally, so x' = fl(fl(k00*u) + k02)
    const double xp = (k00_ * col + 0.0 * row) + k02_;
    const double yp = (0.0 * col + k11_ * row) + k12_;
    out[0] = z * xp;
    out[1] = z * yp;
    out[2] = z;
}

void Oracle::organized_cloud(const float* depth, std::vector<float>& cloud) const
{
    const size_t N = static_cast<size_t>(cfg_.width) * cfg_.height;
    cloud.assign(3 * N, 0.0f);
    for (int row = 0; row < cfg_.height; ++row)
    {
        for (int col = 0; col < cfg_.width; ++col)
        {
            const float z = depth[static_cast<size_t>(row) * cfg_.width + col];
            if (z > 0)
            {
                const int id = cellMap_[static_cast<size_t>(row) * cfg_.width + col];
                double p[3];
                back_project(col, row, z, p);
                cloud[id] = static_cast<float>(p[0]);
                cloud[N + id] = static_cast<float>(p[1]);
                cloud[2 * N + id] = z;
            }
        }
    }
}

// plane_segment.cpp:44-60
static inline bool is_continuous(float pixelDepth, float& lastPixelDepth)
{
    if (pixelDepth > 0)
    {
        if (fabsf(pixelDepth - lastPixelDepth) <= 4.0 * depth_quantization(pixelDepth))
        {
            lastPixelDepth = pixelDepth;
            return true;
        }
        return false;
    }
    return true;
}

// Plane_Segment::init_plane_segment, plane_segment.cpp:102-168
static void init_plane_segment(PlaneSeg& s, const float* xM, const float* yM, const float* zM)
{
    clear_segment(s);

    // is_cell_horizontal_continuous :82-100
    {
        const unsigned startValue = static_cast<unsigned>(kCell * (kCell / 2.0));
        const unsigned endValue = startValue + kCell;
        float last = std::max(zM[startValue], zM[startValue + 1]);
        if (last <= 0)
            return;
        for (unsigned i = startValue + 1; i < endValue; ++i)
            if (!is_continuous(zM[i], last))
                return;
    }
    // is_cell_vertical_continuous :62-80
    {
        const unsigned startValue = kCell / 2;
        const unsigned endValue = kPtsPerCell - startValue;
        float last = std::max(zM[startValue], zM[startValue + kCell]);
        if (last <= 0)
            return;
        for (unsigned i = startValue + kCell; i < endValue; i += kCell)
            if (!is_continuous(zM[i], last))
                return;
    }
    // :120
    unsigned positive = 0;
    for (unsigned i = 0; i < kPtsPerCell; ++i)
        positive += (zM[i] > 0) ? 1u : 0u;
    if (positive < kPtsPerCell / 2)
        return;

    s.n = 0;
    for (unsigned i = 0; i < kPtsPerCell; ++i)
    {
        const float z = zM[i];
        if (z > 0)
        {
            ++s.n;
            const float x = xM[i];
            const float y = yM[i];
            s.Sx += x;
            s.Sy += y;
            s.Sz += z;
            s.Sxs += x * x; // SQR(float) is a float product, types.hpp:84
            s.Sys += y * y;
            s.Szs += z * z;
            s.Sxy += x * y;
            s.Szx += x * z;
            s.Syz += y * z;
        }
    }

    // _minZeroPointCount = floor(400 * 0.7f), plane_segment.hpp:33-34
    static const unsigned minZeroPointCount =
            static_cast<unsigned>(std::floor(static_cast<float>(kPtsPerCell) * kMinZeroDepthProportion));
    if (s.n < minZeroPointCount)
        return;

    fit_plane(s);
    const double q = depth_quantization(s.centroid[2]);
    s.planar = s.mse <= q * q;
}

void Oracle::rectify_depth(const float* depth, const double T[16], std::vector<float>& rectified) const
{
    const int W = cfg_.width, H = cfg_.height;
    rectified.assign(static_cast<size_t>(W) * H, 0.0f);
    for (int row = 0; row < H; ++row)
    {
        for (int column = 0; column < W; ++column)
        {
            const float originalZ = depth[static_cast<size_t>(row) * W + column];
            if (!(originalZ > 0)) // `<= 0 -> continue`; NaN would reach exit(-1) in the reference, dropped here
                continue;
            // _Xpre / _Ypre = static_cast<float>(ScreenCoordinate2D(col,row).to_camera_coordinates()), :156-161
            const float preX = static_cast<float>((k00_ * column + 0.0 * row) + k02_);
            const float preY = static_cast<float>((0.0 * column + k11_ * row) + k12_);
            const double o[3] = {static_cast<double>(preX * originalZ), static_cast<double>(preY * originalZ),
                                 static_cast<double>(originalZ)};
            // (T * original.homogeneous()).head<3>(): T[:, :3] * original + T[:, 3]
            double pr[3];
            for (int i = 0; i < 3; ++i)
                pr[i] = ((T[4 * i] * o[0] + T[4 * i + 1] * o[1]) + T[4 * i + 2] * o[2]) + T[4 * i + 3];
            // CameraCoordinate::to_screen_coordinates (point_coordinates.cpp:201-210): 1.0 / z * (K1 * p).head<2>()
            const double u = (cfg_.fx * pr[0] + 0.0 * pr[1]) + cfg_.cx * pr[2];
            const double v = (0.0 * pr[0] + cfg_.fy * pr[1]) + cfg_.cy * pr[2];
            const double s = 1.0 / pr[2];
            const double sx = s * u, sy = s * v;
            if (std::isnan(sx) || std::isnan(sy))
                continue;
            const double fx_ = std::floor(sx), fy_ = std::floor(sy);
            if (!(fx_ > 0.0 && fy_ > 0.0 && fx_ < W && fy_ < H))
                continue;
            rectified[static_cast<size_t>(fy_) * W + static_cast<size_t>(fx_)] = static_cast<float>(pr[2]);
        }
    }
}

void Oracle::run(const float* depth, FrameResult& out)
{
    organized_cloud(depth, lastCloud);
    find_primitives(lastCloud.data(), depth, out);
}

void Oracle::find_primitives(const float* cloud, const float* depth, FrameResult& out)
{
    const size_t N = static_cast<size_t>(cfg_.width) * cfg_.height;
    const int cells = totalCells_;
    const int H = hCells_, V = vCells_;

    // ---- reset_data, primitive_detection.cpp:168-185
    Histogram20 histogram;
    histogram.reset();
    std::vector<PlaneSeg> planeSegments;
    std::vector<CylinderSeg> cylinderSegments;
    std::vector<int32_t> gridPlane(cells, 0), gridCyl(cells, 0);
    std::vector<uint8_t> isUnassigned(cells, 0);
    Rng rng(cfg_.rngSeed); // fresh thread_local engine per call (see Rng)

    out = FrameResult();

    // ---- init_planar_cell_fitting, :187-237
    static const float sinAngleForMerge = sinf(static_cast<float>(kMaxAngleForMerge_d * M_PI / 180.0));
    constexpr float planeMergeDistanceThreshold = kMaxDistForMerge_mm;
    for (int cell = 0; cell < cells; ++cell)
    {
        const size_t offset = static_cast<size_t>(cell) * kPtsPerCell;
        PlaneSeg& seg = planeGrid[cell];
        init_plane_segment(seg, cloud + offset, cloud + N + offset, cloud + 2 * N + offset);
        if (seg.planar)
        {
            // (row(offset+399) - row(offset)).norm() on float 1x3: a0 + (a1 + a2) (no packet for 3 floats)
            const float dx = cloud[offset + kPtsPerCell - 1] - cloud[offset];
            const float dy = cloud[N + offset + kPtsPerCell - 1] - cloud[N + offset];
            const float dz = cloud[2 * N + offset + kPtsPerCell - 1] - cloud[2 * N + offset];
            const float cellDiameter = sqrtf(dx * dx + (dy * dy + dz * dz));
            cellTols[cell] = std::min(planeMergeDistanceThreshold,
                                      cellDiameter * sinAngleForMerge * sqrtf(static_cast<float>(seg.n)));
        }
        else
        {
            cellTols[cell] = 0;
        }
    }

    // ---- init_histogram, :239-265
    unsigned remainingPlanarCells = 0;
    {
        std::vector<double> p0(cells, 0.0), p1(cells, 0.0);
        for (int cell = 0; cell < cells; ++cell)
        {
            const PlaneSeg& seg = planeGrid[cell];
            if (seg.planar)
            {
                p0[cell] = libm_variant(acos(-seg.normal[2]));
                p1[cell] = libm_variant(atan2(seg.normal[0], seg.normal[1]));
                ++remainingPlanarCells;
                isUnassigned[cell] = 1;
            }
        }
        histogram.init(p0, p1, isUnassigned);
        initialBins.assign(histogram.bins.begin(), histogram.bins.end());
    }

    // ---- grow_planes_and_cylinders, :267-310
    std::vector<std::pair<int, int>> cylinder2regionMap;
    unsigned untried = remainingPlanarCells;
    const unsigned planeSeedCount = static_cast<unsigned>(kMinSeedProportion * cells);
    const unsigned minimumCellActivated = static_cast<unsigned>(kMinActivatedProportion * cells);

    std::vector<uint8_t> isActivated(cells, 0);
    // region_growing, :778-818 (recursive 4-neighbour DFS: left, right, up, down)
    struct Grower
    {
        const std::vector<PlaneSeg>& grid;
        const std::vector<float>& tols;
        const std::vector<uint8_t>& unassigned;
        std::vector<uint8_t>& activated;
        int H, V;
        void grow(unsigned x, unsigned y, const PlaneSeg& planeToExpand)
        {
            const int index = static_cast<int>(x + H * y);
            if (index >= H * V)
                return;
            if (!unassigned[index] || activated[index])
                return;
            const PlaneSeg& patch = grid[index];
            if (can_be_merged(planeToExpand, patch, tols[index]))
            {
                activated[index] = 1;
                if (x > 0)
                    grow(x - 1, y, patch);
                if (x < static_cast<unsigned>(H - 1))
                    grow(x + 1, y, patch);
                if (y > 0)
                    grow(x, y - 1, patch);
                if (y < static_cast<unsigned>(V - 1))
                    grow(x, y + 1, patch);
            }
        }
    };

    while (untried > 0)
    {
        const std::vector<unsigned> seedCandidates = histogram.points_from_most_frequent_bin();
        if (seedCandidates.size() < planeSeedCount)
            break;

        unsigned seedId = 0;
        double minMSE = std::numeric_limits<double>::max();
        for (const unsigned cand : seedCandidates)
        {
            const double m = planeGrid[cand].mse;
            if (m >= minMSE)
                continue;
            seedId = cand;
            minMSE = m;
            if (minMSE <= 0)
                break;
        }
        if (minMSE >= std::numeric_limits<double>::max())
        {
            ++out.logInvalidSeed; // log_warning, :302
            break;
        }

        // ---- grow_plane_segment_at_seed, :312-389
        out.seeds.push_back(static_cast<int32_t>(seedId));
        out.seedOutcome.push_back(0);
        out.seedActivated.push_back(0);
        const PlaneSeg& planeToGrow = planeGrid[seedId];
        if (!planeToGrow.planar)
            continue; // unreachable: non planar cells keep MSE = DBL_MAX

        PlaneSeg newSeg = copy_segment(planeToGrow);
        const unsigned y = seedId / H;
        const unsigned x = seedId % H;
        std::fill(isActivated.begin(), isActivated.end(), 0);
        Grower g{planeGrid, cellTols, isUnassigned, isActivated, H, V};
        g.grow(x, y, newSeg);

        unsigned cellActivatedCount = 0;
        bool isPlaneFitable = false;
        for (int i = 0; i < cells; ++i)
        {
            if (isActivated[i])
            {
                const PlaneSeg& ps = planeGrid[i];
                if (ps.planar)
                {
                    expand_segment(newSeg, ps);
                    ++cellActivatedCount;
                    histogram.remove_point(i);
                    isUnassigned[i] = 0;
                    --untried;
                    isPlaneFitable = true;
                }
            }
        }
        out.seedActivated.back() = cellActivatedCount;

        if (!isPlaneFitable || cellActivatedCount < minimumCellActivated)
        {
            histogram.remove_point(seedId);
            continue;
        }

        fit_plane(newSeg);
        if (!newSeg.planar)
        {
            out.seedOutcome.back() = 4;
            ++out.logNotPlanarAfterMerge; // log, :374
            continue;
        }

        if (newSeg.score > 100)
        {
            // add_plane_segment_to_features, :391-411
            planeSegments.push_back(copy_segment(newSeg));
            const int currentPlaneCount = static_cast<int>(planeSegments.size());
            for (int i = 0; i < cells; ++i)
                if (isActivated[i])
                    gridPlane[i] = currentPlaneCount;
            out.seedOutcome.back() = 1;
        }
        else if (cellActivatedCount > 5)
        {
            if (!cfg_.cylinders)
            {
                out.seedOutcome.back() = 3;
                continue;
            }
            out.seedOutcome.back() = 2;
            // cylinder_fitting, :478-501
            CylinderSeg cyl = make_cylinder_segment(planeGrid, isActivated, cellActivatedCount, rng);
            cylinderSegments.push_back(cyl);
            for (unsigned segId = 0; segId < cyl.segmentCount; ++segId)
            {
                PlaneSeg merged; // default ctor = cleared
                bool fitable = false;
                // find_plane_segment_in_cylinder, :413-435
                for (unsigned col = 0; col < cellActivatedCount; ++col)
                {
                    if (cyl.inliers[segId][col])
                    {
                        const PlaneSeg& ps = planeGrid[cyl.local2global[col]];
                        if (ps.planar)
                        {
                            expand_segment(merged, ps);
                            fitable = true;
                        }
                    }
                }
                if (!fitable)
                    continue;
                fit_plane(merged);
                if (!merged.planar)
                    ++out.logNotPlanarAfterMerge; // log, :497 (the sub-segment still goes through the model selection)
                // add_cylinder_to_features, :437-476
                if (merged.mse < cyl.mse[segId])
                {
                    planeSegments.push_back(copy_segment(merged));
                    const int currentPlaneCount = static_cast<int>(planeSegments.size());
                    for (unsigned col = 0; col < cellActivatedCount; ++col)
                        if (cyl.inliers[segId][col])
                            gridPlane[cyl.local2global[col]] = currentPlaneCount;
                }
                else
                {
                    cylinder2regionMap.emplace_back(static_cast<int>(cylinderSegments.size()) - 1, static_cast<int>(segId));
                    const int cylinderCount = static_cast<int>(cylinder2regionMap.size());
                    for (unsigned col = 0; col < cellActivatedCount; ++col)
                        if (cyl.inliers[segId][col])
                            gridCyl[cyl.local2global[col]] = cylinderCount;
                }
            }
        }
        else
        {
            out.seedOutcome.back() = 3;
        }
    }

    // ---- merge_planes, :503-560 ; get_connected_components_matrix :736-776
    const unsigned planeCount = static_cast<unsigned>(planeSegments.size());
    std::vector<uint8_t> connected(static_cast<size_t>(planeCount) * planeCount, 0);
    if (planeCount > 0)
    {
        for (int row = 0; row < V - 1; ++row)
        {
            for (int col = 0; col < H - 1; ++col)
            {
                const int planeId = gridPlane[row * H + col];
                if (planeId <= 0)
                    continue;
                const int nextPlaneId = gridPlane[row * H + col + 1];
                const int belowPlaneId = gridPlane[(row + 1) * H + col];
                if (nextPlaneId > 0 && planeId != nextPlaneId)
                {
                    connected[(planeId - 1) * planeCount + (nextPlaneId - 1)] = 1;
                    connected[(nextPlaneId - 1) * planeCount + (planeId - 1)] = 1;
                }
                if (belowPlaneId > 0 && planeId != belowPlaneId)
                {
                    connected[(planeId - 1) * planeCount + (belowPlaneId - 1)] = 1;
                    connected[(belowPlaneId - 1) * planeCount + (planeId - 1)] = 1;
                }
            }
        }
    }
    std::vector<uint32_t> mergeLabels(planeCount);
    for (unsigned i = 0; i < planeCount; ++i)
        mergeLabels[i] = i;
    for (unsigned row = 0; row < planeCount; ++row)
    {
        bool wasExpanded = false;
        const unsigned planeId = mergeLabels[row];
        PlaneSeg& planeToExpand = planeSegments[planeId];
        if (!planeToExpand.planar)
            continue;
        for (unsigned col = row + 1; col < planeCount; ++col)
        {
            if (!connected[row * planeCount + col])
                continue;
            const PlaneSeg& mergePlane = planeSegments[col];
            if (!mergePlane.planar)
                continue;
            if (can_be_merged(planeToExpand, mergePlane, kMaxDistForMerge_mm))
            {
                expand_segment(planeToExpand, mergePlane);
                mergeLabels[col] = planeId;
                wasExpanded = true;
            }
            else
            {
                connected[row * planeCount + col] = 0;
                connected[col * planeCount + row] = 0;
            }
        }
        if (wasExpanded)
            fit_plane(planeToExpand);
    }

    // ---- add_planes_to_primitives, :562-648 ; compute_plane_segment_boundary :650-703
    std::vector<uint8_t> mask(cells), eroded, dilated;
    for (unsigned planeIndex = 0; planeIndex < planeCount; ++planeIndex)
    {
        const unsigned label = mergeLabels[planeIndex];
        if (planeIndex != label)
            continue;
        const PlaneSeg& ps = planeSegments[planeIndex];
        if (!ps.planar)
            continue;

        std::fill(mask.begin(), mask.end(), 0);
        for (unsigned jj = planeIndex; jj < planeCount; ++jj)
        {
            if (mergeLabels[jj] == label)
            {
                for (int i = 0; i < cells; ++i)
                    if (gridPlane[i] == static_cast<int>(jj + 1))
                        mask[i] = 1;
            }
        }

        const double maxBoundaryDistance = 3 * std::sqrt(ps.mse);
        morph3(mask, eroded, V, H, /*cross*/ true, /*erode*/ true, Border::Zero);
        morph3(mask, dilated, V, H, /*cross*/ false, /*erode*/ false, Border::Ignore);

        PlaneOut po;
        for (int row = 0; row < V; ++row)
        {
            for (int col = 0; col < H; ++col)
            {
                const int i = row * H + col;
                const uint8_t ring = dilated[i] > eroded[i] ? static_cast<uint8_t>(dilated[i] - eroded[i]) : 0;
                if (ring <= 0)
                    continue;
                const int centerX = static_cast<int>(col * kCell + kCell / 2);
                const int centerY = static_cast<int>(row * kCell + kCell / 2);
                const double dpt = depth[static_cast<size_t>(centerY) * cfg_.width + centerX];
                if (dpt > 0)
                {
                    double p[3];
                    back_project(centerX, centerY, dpt, p);
                    const double dist = dot3(ps.normal, p) + ps.d;
                    if (std::abs(dist) < maxBoundaryDistance)
                    {
                        po.boundary.push_back(p[0]);
                        po.boundary.push_back(p[1]);
                        po.boundary.push_back(p[2]);
                    }
                }
            }
        }
        if (po.boundary.size() / 3 < 3)
            continue;

        // Plane(planeSeg, polygon), shape_primitives.cpp:48-56 (polygon construction itself is OUT)
        po.segment_index = static_cast<int>(planeIndex);
        double nrm[3] = {ps.normal[0], ps.normal[1], ps.normal[2]};
        normalize3(nrm);
        for (int k = 0; k < 3; ++k)
        {
            po.normal[k] = nrm[k];
            po.centroid[k] = ps.centroid[k];
        }
        po.d = ps.d;
        po.mse = ps.mse;
        po.score = ps.score;
        po.n = ps.n;
        const double hess[3][3] = {{ps.Sxs, ps.Sxy, ps.Szx}, {ps.Sxy, ps.Sys, ps.Syz}, {ps.Szx, ps.Syz, ps.Szs}};
        double inv[3][3];
        inverse3(hess, inv);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                po.cov[3 * r + c] = inv[r][c];
        out.planes.push_back(std::move(po));
    }

    // ---- add_cylinders_to_primitives, :705-734
    for (size_t ci = 0; ci < cylinder2regionMap.size(); ++ci)
    {
        std::fill(mask.begin(), mask.end(), 0);
        for (int i = 0; i < cells; ++i)
            if (gridCyl[i] == static_cast<int>(ci + 1))
                mask[i] = 1;
        morph3(mask, mask, V, H, true, false, Border::Ignore);
        morph3(mask, mask, V, H, true, true, Border::Ignore);
        morph3(mask, eroded, V, H, true, true, Border::Ignore);
        uint8_t mn = 255, mx = 0;
        for (int i = 0; i < cells; ++i)
        {
            mn = std::min(mn, eroded[i]);
            mx = std::max(mx, eroded[i]);
        }
        if (mx <= 0 || mn >= mx)
            continue;
        const int regId = cylinder2regionMap[ci].first;
        CylinderOut co;
        for (int k = 0; k < 3; ++k)
            co.axis[k] = cylinderSegments[regId].axis[k];
        co.radius = std::numeric_limits<double>::quiet_NaN(); // 0/0, see header
        out.cylinders.push_back(co);
    }

    out.planeLabels = gridPlane;
    out.cylLabels = gridCyl;
    out.planeSegments = planeSegments;
    out.mergeLabels = mergeLabels;
}

} // namespace cape_oracle

// =====================================================================================================
// C interface for ctypes (tests / smoke / bench cpu_baseline only)
// =====================================================================================================
using namespace cape_oracle;

namespace {
struct Handle
{
    Oracle oracle;
    FrameResult res;
    explicit Handle(const Config& c) : oracle(c) {}
};
} // namespace

extern "C" {

void* cape_oracle_create(int width, int height, double fx, double fy, double cx, double cy, int cylinders)
{
    Config c;
    c.width = width;
    c.height = height;
    c.fx = fx;
    c.fy = fy;
    c.cx = cx;
    c.cy = cy;
    c.cylinders = cylinders != 0;
    return new Handle(c);
}

// utils::Random::_seed of a build without MAKE_DETERMINISTIC (random.hpp:59-64): the next frames restart their engine at `seed`
void cape_oracle_set_rng_seed(void* h, unsigned seed) { static_cast<Handle*>(h)->oracle.set_rng_seed(seed); }

void cape_oracle_destroy(void* h) { delete static_cast<Handle*>(h); }

int cape_oracle_cells(void* h) { return static_cast<Handle*>(h)->oracle.cells(); }

int cape_oracle_run(void* h, const float* depth)
{
    Handle* H = static_cast<Handle*>(h);
    H->oracle.run(depth, H->res);
    return 0;
}

// run n frames back to back (cpu_baseline timing loop: no Python in the timed region)
int cape_oracle_run_many(void* h, const float* depth, int n_frames, long long* total_planes)
{
    Handle* H = static_cast<Handle*>(h);
    const size_t stride = static_cast<size_t>(H->oracle.config().width) * H->oracle.config().height;
    long long acc = 0;
    for (int f = 0; f < n_frames; ++f)
    {
        H->oracle.run(depth + f * stride, H->res);
        acc += static_cast<long long>(H->res.planes.size());
    }
    if (total_planes)
        *total_planes = acc;
    return 0;
}

void cape_oracle_rectify(void* h, const float* depth, const double* T16, float* out)
{
    Handle* H = static_cast<Handle*>(h);
    std::vector<float> r;
    H->oracle.rectify_depth(depth, T16, r);
    std::memcpy(out, r.data(), r.size() * sizeof(float));
}

void cape_oracle_get_cloud(void* h, float* cloud)
{
    Handle* H = static_cast<Handle*>(h);
    std::memcpy(cloud, H->oracle.lastCloud.data(), H->oracle.lastCloud.size() * sizeof(float));
}

// per-cell stats, SoA: sums is cells x 9 in the order Sx,Sy,Sz,Sxs,Sys,Szs,Sxy,Syz,Szx
void cape_oracle_get_cell_stats(void* h, uint32_t* n, uint8_t* planar, double* sums, double* centroid, double* normal,
                                double* d, double* mse, double* score, float* tol, int32_t* bins)
{
    Handle* H = static_cast<Handle*>(h);
    const int cells = H->oracle.cells();
    for (int i = 0; i < cells; ++i)
    {
        const PlaneSeg& s = H->oracle.planeGrid[i];
        n[i] = s.n;
        planar[i] = s.planar ? 1 : 0;
        const double ss[9] = {s.Sx, s.Sy, s.Sz, s.Sxs, s.Sys, s.Szs, s.Sxy, s.Syz, s.Szx};
        for (int k = 0; k < 9; ++k)
            sums[9 * i + k] = ss[k];
        for (int k = 0; k < 3; ++k)
        {
            centroid[3 * i + k] = s.centroid[k];
            normal[3 * i + k] = s.normal[k];
        }
        d[i] = s.d;
        mse[i] = s.mse;
        score[i] = s.score;
        tol[i] = H->oracle.cellTols[i];
        bins[i] = H->oracle.initialBins[i];
    }
}

void cape_oracle_get_labels(void* h, int32_t* plane, int32_t* cyl)
{
    Handle* H = static_cast<Handle*>(h);
    std::memcpy(plane, H->res.planeLabels.data(), H->res.planeLabels.size() * sizeof(int32_t));
    std::memcpy(cyl, H->res.cylLabels.data(), H->res.cylLabels.size() * sizeof(int32_t));
}

void cape_oracle_log_counts(void* h, int* invalid_seed, int* not_planar_after_merge)
{
    Handle* H = static_cast<Handle*>(h);
    *invalid_seed = H->res.logInvalidSeed;
    *not_planar_after_merge = H->res.logNotPlanarAfterMerge;
}

int cape_oracle_num_seeds(void* h) { return static_cast<int>(static_cast<Handle*>(h)->res.seeds.size()); }
void cape_oracle_get_seeds(void* h, int32_t* seeds, int32_t* outcome, uint32_t* activated)
{
    Handle* H = static_cast<Handle*>(h);
    for (size_t i = 0; i < H->res.seeds.size(); ++i)
    {
        seeds[i] = H->res.seeds[i];
        outcome[i] = H->res.seedOutcome[i];
        activated[i] = H->res.seedActivated[i];
    }
}

int cape_oracle_num_plane_segments(void* h)
{
    return static_cast<int>(static_cast<Handle*>(h)->res.planeSegments.size());
}
// rec: P x 20 doubles = normal[3], d, centroid[3], mse, score, sums[9], n, planar ; merge: P
void cape_oracle_get_plane_segments(void* h, double* rec, uint32_t* merge)
{
    Handle* H = static_cast<Handle*>(h);
    for (size_t i = 0; i < H->res.planeSegments.size(); ++i)
    {
        const PlaneSeg& s = H->res.planeSegments[i];
        double* r = rec + 20 * i;
        r[0] = s.normal[0]; r[1] = s.normal[1]; r[2] = s.normal[2]; r[3] = s.d;
        r[4] = s.centroid[0]; r[5] = s.centroid[1]; r[6] = s.centroid[2];
        r[7] = s.mse; r[8] = s.score;
        r[9] = s.Sx; r[10] = s.Sy; r[11] = s.Sz; r[12] = s.Sxs; r[13] = s.Sys; r[14] = s.Szs;
        r[15] = s.Sxy; r[16] = s.Syz; r[17] = s.Szx;
        r[18] = static_cast<double>(s.n);
        r[19] = s.planar ? 1.0 : 0.0;
        merge[i] = H->res.mergeLabels[i];
    }
}

int cape_oracle_num_planes(void* h) { return static_cast<int>(static_cast<Handle*>(h)->res.planes.size()); }
// rec: P x 20 doubles = normal[3], d, centroid[3], mse, score, n, cov[9], segment_index ; nb: boundary point counts
void cape_oracle_get_planes(void* h, double* rec, int32_t* nb)
{
    Handle* H = static_cast<Handle*>(h);
    for (size_t i = 0; i < H->res.planes.size(); ++i)
    {
        const PlaneOut& p = H->res.planes[i];
        double* r = rec + 20 * i;
        r[0] = p.normal[0]; r[1] = p.normal[1]; r[2] = p.normal[2]; r[3] = p.d;
        r[4] = p.centroid[0]; r[5] = p.centroid[1]; r[6] = p.centroid[2];
        r[7] = p.mse; r[8] = p.score; r[9] = static_cast<double>(p.n);
        for (int k = 0; k < 9; ++k)
            r[10 + k] = p.cov[k];
        r[19] = static_cast<double>(p.segment_index);
        nb[i] = static_cast<int32_t>(p.boundary.size() / 3);
    }
}
void cape_oracle_get_boundary(void* h, int plane, double* pts)
{
    Handle* H = static_cast<Handle*>(h);
    const PlaneOut& p = H->res.planes[plane];
    std::memcpy(pts, p.boundary.data(), p.boundary.size() * sizeof(double));
}

int cape_oracle_num_cylinders(void* h) { return static_cast<int>(static_cast<Handle*>(h)->res.cylinders.size()); }
void cape_oracle_get_cylinders(void* h, double* rec /* C x 4: axis[3], radius */)
{
    Handle* H = static_cast<Handle*>(h);
    for (size_t i = 0; i < H->res.cylinders.size(); ++i)
    {
        const CylinderOut& c = H->res.cylinders[i];
        rec[4 * i + 0] = c.axis[0];
        rec[4 * i + 1] = c.axis[1];
        rec[4 * i + 2] = c.axis[2];
        rec[4 * i + 3] = c.radius;
    }
}

// known-answer helpers
double cape_oracle_depth_quantization(double z) { return depth_quantization(z); }
void cape_oracle_eigen3(const double* lower9 /* row-major 3x3 */, double* evals, double* evecs9, int* iters)
{
    double m[3][3], ev[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            m[r][c] = lower9[3 * r + c];
    self_adjoint_eigen3(m, evals, ev, iters);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            evecs9[3 * r + c] = ev[r][c];
}
double cape_oracle_mt19937_double(unsigned seed, int index) { return mt19937_first_double(seed, index); }
void cape_oracle_back_project(void* h, double col, double row, double z, double* out)
{
    static_cast<Handle*>(h)->oracle.back_project(col, row, z, out);
}
unsigned cape_oracle_ransac_max_iterations()
{
    return static_cast<unsigned>(logf(1.0f - kCylPSuccess) / logf(1.0f - powf(kCylInlierProp, 3.0f)));
}
double cape_oracle_cos_merge_angle() { return std::cos(static_cast<double>(kMaxAngleForMerge_d) * M_PI / 180.0); }
float cape_oracle_sin_merge_angle() { return sinf(static_cast<float>(kMaxAngleForMerge_d * M_PI / 180.0)); }

} // extern "C"
