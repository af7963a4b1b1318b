// Device-side scalar math of the CAPE path (gfx950).  Every function performs the IEEE-754 operation
// sequence of the reference function it cites (paths relative to the reference root); the translation unit is
// compiled with -ffp-contract=off so no mul+add pair is fused, and f64 div/sqrt expand to the correctly
// rounded sequences.  Comparisons mirror std::max/std::min argument order so NaN/-0 behave the same.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cape_hip.h"

namespace cape {

constexpr int kCell = 20;
constexpr int kPts = 400;
constexpr double kDblMax = 1.7976931348623157e308;
constexpr double kDblMin = 2.2250738585072014e-308;
constexpr double kDblEps = 2.220446049250313e-16;

// std::max(a,b) = (a < b) ? b : a ; std::min(a,b) = (b < a) ? b : a
__device__ __forceinline__ double std_max(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ float std_maxf(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float std_minf(float a, float b) { return (b < a) ? b : a; }

// utils::get_depth_quantization, src/utils/covariances.cpp:12-19 (constants src/parameters.hpp:16-18)
__device__ __forceinline__ double depth_quantization(double depth)
{
    constexpr double sigmaError = 2.73 * ((1.0 / 1000.0) * (1.0 / 1000.0));
    constexpr double sigmaMultiplier = 0.74 / 1000.0;
    constexpr double sigmaMargin = -0.53;
    return std_max(sigmaMargin + sigmaMultiplier * depth + sigmaError * (depth * depth), 0.5);
}

// Eigen fixed-size Vector3d reductions: (a0 + a1) + a2
__device__ __forceinline__ double dot3(double a0, double a1, double a2, double b0, double b1, double b2)
{
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

// Eigen normalize(): z = squaredNorm(); if (z > 0) v /= sqrt(z)   (plane_coordinates.hpp:19-40)
__device__ __forceinline__ void normalize3(double& x, double& y, double& z)
{
    const double n2 = (x * x + y * y) + z * z;
    if (n2 > 0)
    {
        const double s = sqrt(n2);
        x /= s;
        y /= s;
        z /= s;
    }
}

// numext::hypot (positive_real_hypot), used by tridiagonal_qr_step's Wilkinson shift
__device__ __forceinline__ double eigen_hypot(double x, double y)
{
    x = fabs(x);
    y = fabs(y);
    if (isinf(x) || isinf(y))
        return __builtin_huge_val();
    if (isnan(x) || isnan(y))
        return __builtin_nan("");
    const double p = (x < y) ? y : x;
    if (p == 0.0)
        return 0.0;
    const double qp = ((x < y) ? x : y) / p;
    return p * sqrt(1.0 + qp * qp);
}

// JacobiRotation<double>::makeGivens (real case).  Eigen's two general branches (|p| > |q| and the other) are the same
// statements with p and q swapped: ONE division, one square root and one reciprocal on selected operands give the same bits
// as either branch (-(1/u) == (-1)/u), and the lanes of a wave no longer serialise the two.
__device__ __forceinline__ void make_givens(double p, double q, double& c, double& s)
{
    const bool big = fabs(p) > fabs(q);
    const double num = big ? q : p, den = big ? p : q;
    const double t = num / den;
    double u = sqrt(1.0 + t * t);
    if (den < 0.0)
        u = -u;
    const double r = 1.0 / u;
    const double first = big ? r : -r; // c of the first branch, s of the second
    const double second = -t * first;  // s = -t * c, resp. c = -t * s
    c = big ? first : second;
    s = big ? second : first;
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
}

// Eigen 3.4.0 SelfAdjointEigenSolver<Matrix3d>::compute, iterative QL/QR path (call sites plane_segment.cpp:251,
// cylinder_segment.cpp:92).  Reads the lower triangle.  Outputs eigenvalues ascending and the matching
// eigenvector columns q[r][c].  Arrays are indexed with compile-time constants only (kept in registers).
struct Eig3
{
    double val[3];
    double q[3][3];
};

__device__ __forceinline__ void qr_rotate(double (&Q)[3][3], int k, double c, double s)
{
    // q.applyOnTheRight(k, k+1, rot) ; k is 0 or 1
#pragma unroll
    for (int r = 0; r < 3; ++r)
    {
        if (k == 0)
        {
            const double xi = Q[r][0], yi = Q[r][1];
            Q[r][0] = c * xi - s * yi;
            Q[r][1] = s * xi + c * yi;
        }
        else
        {
            const double xi = Q[r][1], yi = Q[r][2];
            Q[r][1] = c * xi - s * yi;
            Q[r][2] = s * xi + c * yi;
        }
    }
}

__device__ inline void self_adjoint_eigen3(double m00, double m10, double m11, double m20, double m21, double m22, Eig3& out)
{
    double scale = fabs(m00);
    scale = std_max(scale, fabs(m10));
    scale = std_max(scale, fabs(m20));
    scale = std_max(scale, fabs(m11));
    scale = std_max(scale, fabs(m21));
    scale = std_max(scale, fabs(m22));
    if (scale == 0.0)
        scale = 1.0;
    m00 /= scale;
    m10 /= scale;
    m11 /= scale;
    m20 /= scale;
    m21 /= scale;
    m22 /= scale;

    double d0, d1, d2, s0, s1;
    double Q[3][3];
    d0 = m00;
    const double v1norm2 = m20 * m20;
    if (v1norm2 <= kDblMin)
    {
        d1 = m11;
        d2 = m22;
        s0 = m10;
        s1 = m21;
        Q[0][0] = 1; Q[0][1] = 0; Q[0][2] = 0;
        Q[1][0] = 0; Q[1][1] = 1; Q[1][2] = 0;
        Q[2][0] = 0; Q[2][1] = 0; Q[2][2] = 1;
    }
    else
    {
        const double beta = sqrt(m10 * m10 + v1norm2);
        const double invBeta = 1.0 / beta;
        const double m01 = m10 * invBeta;
        const double m02 = m20 * invBeta;
        const double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
        d1 = m11 + m02 * q;
        d2 = m22 - m02 * q;
        s0 = beta;
        s1 = m21 - m01 * q;
        Q[0][0] = 1; Q[0][1] = 0;   Q[0][2] = 0;
        Q[1][0] = 0; Q[1][1] = m01; Q[1][2] = m02;
        Q[2][0] = 0; Q[2][1] = m02; Q[2][2] = -m01;
    }

    // computeFromTridiagonal_impl, n = 3, m_maxIterations = 30.  The generic loop over (start,end) is unrolled
    // into its three possible blocks: [0,2], [1,2], [0,1].
    const double precision_inv = 1.0 / kDblEps;
    int end = 2, start = 0, iter = 0;
    while (end > 0)
    {
        // deflation test for i in [start, end)
        if (start <= 0 && end > 0)
        {
            if (fabs(s0) < kDblMin)
                s0 = 0.0;
            else
            {
                const double sc = precision_inv * s0;
                if (sc * sc <= (fabs(d0) + fabs(d1)))
                    s0 = 0.0;
            }
        }
        if (start <= 1 && end > 1)
        {
            if (fabs(s1) < kDblMin)
                s1 = 0.0;
            else
            {
                const double sc = precision_inv * s1;
                if (sc * sc <= (fabs(d1) + fabs(d2)))
                    s1 = 0.0;
            }
        }
        // find the largest unreduced block at the end of the matrix
        while (end > 0 && ((end == 2) ? s1 : s0) == 0.0)
            end--;
        if (end <= 0)
            break;
        iter++;
        if (iter > 90)
            break;
        start = end - 1;
        while (start > 0 && ((start == 2) ? s1 : s0) != 0.0) // subdiag[start-1], start-1 in {0}
            start--;

        // tridiagonal_qr_step(diag, subdiag, start, end, Q)
        const double dem1 = (end == 2) ? d1 : d0; // diag[end-1]
        const double de = (end == 2) ? d2 : d1;   // diag[end]
        const double e = (end == 2) ? s1 : s0;    // subdiag[end-1]
        const double td = (dem1 - de) * 0.5;
        double mu = de;
        if (td == 0.0)
        {
            mu -= fabs(e);
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

        // k = start: ONE rotation on (k, k + 1), k in {0, 1} -- the reference's generic loop body.  The lanes of a wave sit in
        // different (start, end) blocks, and three unrolled copies of it used to run one after the other; the body works on
        // SELECTED operands (SEL) when some lane of the wave is in block [1,2], and as plain k = 0 code -- no selects -- when
        // none is, which is the rule: the shift makes the bottom of the matrix deflate first, so start stays 0.
        auto qr_first = [&](auto sel) {
            constexpr bool SEL = decltype(sel)::value;
            const bool k1 = SEL ? (start != 0) : false;
            const double dk = k1 ? d1 : d0, sk = k1 ? s1 : s0, dk1 = k1 ? d2 : d1;
            double x = dk - mu;
            double z = sk;
            if (z != 0.0)
            {
                double c, s;
                make_givens(x, z, c, s);
                const double sdk = s * dk + c * sk;
                const double dkp1 = s * sk + c * dk1;
                const double ndk = c * (c * dk - s * sk) - s * (c * sk - s * dk1);
                const double ndk1 = s * sdk + c * dkp1;
                const double nsk = c * sdk - s * dkp1;
                // q.applyOnTheRight(k, k + 1, rot)
#pragma unroll
                for (int r = 0; r < 3; ++r)
                {
                    const double xi = k1 ? Q[r][1] : Q[r][0], yi = k1 ? Q[r][2] : Q[r][1];
                    const double nx = c * xi - s * yi, ny = s * xi + c * yi;
                    Q[r][0] = k1 ? Q[r][0] : nx;
                    Q[r][1] = k1 ? nx : ny;
                    Q[r][2] = k1 ? ny : Q[r][2];
                }
                if (k1)
                {
                    d1 = ndk;
                    d2 = ndk1;
                    s1 = nsk;
                }
                else
                {
                    d0 = ndk;
                    d1 = ndk1;
                    s0 = nsk;
                    x = s0;
                    if (0 < end - 1)
                    {
                        z = -s * s1;
                        s1 = c * s1;
                    }
                    // k = 1 (only when start == 0 and end == 2)
                    if (end == 2 && z != 0.0)
                    {
                        double c2, s2;
                        make_givens(x, z, c2, s2);
                        const double sdk2 = s2 * d1 + c2 * s1;
                        const double dkp12 = s2 * s1 + c2 * d2;
                        const double nd1 = c2 * (c2 * d1 - s2 * s1) - s2 * (c2 * s1 - s2 * d2);
                        d2 = s2 * sdk2 + c2 * dkp12;
                        s1 = c2 * sdk2 - s2 * dkp12;
                        d1 = nd1;
                        s0 = c2 * s0 - s2 * z; // k > start
                        qr_rotate(Q, 1, c2, s2);
                    }
                }
            }
        };
        if (__any(start != 0))
            qr_first(std::true_type{});
        else
            qr_first(std::false_type{});
    }

    if (iter <= 90)
    {
        // selection sort ascending, swapping eigenvector columns (first strict minimum wins)
        // i = 0: argmin over (d0,d1,d2)
        {
            int k = 0;
            double best = d0;
            if (d1 < best) { best = d1; k = 1; }
            if (d2 < best) { best = d2; k = 2; }
            if (k == 1)
            {
                double t = d0; d0 = d1; d1 = t;
#pragma unroll
                for (int r = 0; r < 3; ++r) { double u = Q[r][0]; Q[r][0] = Q[r][1]; Q[r][1] = u; }
            }
            else if (k == 2)
            {
                double t = d0; d0 = d2; d2 = t;
#pragma unroll
                for (int r = 0; r < 3; ++r) { double u = Q[r][0]; Q[r][0] = Q[r][2]; Q[r][2] = u; }
            }
        }
        // i = 1: argmin over (d1,d2)
        if (d2 < d1)
        {
            double t = d1; d1 = d2; d2 = t;
#pragma unroll
            for (int r = 0; r < 3; ++r) { double u = Q[r][1]; Q[r][1] = Q[r][2]; Q[r][2] = u; }
        }
    }

    out.val[0] = d0 * scale;
    out.val[1] = d1 * scale;
    out.val[2] = d2 * scale;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            out.q[r][c] = Q[r][c];
}

// Plane fit result of Plane_Segment::fit_plane (plane_segment.cpp:232-284)
struct PlaneFit
{
    double nx, ny, nz, d;
    double cx, cy, cz;
    double mse, score;
    bool planar;
};

// sums order: Sx Sy Sz Sxs Sys Szs Sxy Syz Szx
__device__ inline void fit_plane(const double (&S)[9], uint32_t n, PlaneFit& f)
{
    f.planar = false;
    f.nx = f.ny = f.nz = 0.0;
    f.d = 0.0;
    f.mse = kDblMax;
    f.score = 0.0;
    const double inv = 1.0 / (double)n;
    f.cx = S[0] * inv;
    f.cy = S[1] * inv;
    f.cz = S[2] * inv;

    // get_point_cloud_Huygen_covariance, plane_segment.cpp:205-230
    const double xx = std_max(0.0, S[3] - (S[0] * S[0]) * inv);
    const double yy = std_max(0.0, S[4] - (S[1] * S[1]) * inv);
    const double zz = std_max(0.0, S[5] - (S[2] * S[2]) * inv);
    const double xy = S[6] - S[0] * S[1] * inv;
    const double xz = S[8] - S[0] * S[2] * inv;
    const double yz = S[7] - S[1] * S[2] * inv;

    // Matrix3d::determinant, first-row expansion
    const double det = xx * (yy * zz - yz * yz) - xy * (xy * zz - yz * xz) + xz * (xy * yz - yy * xz);
    if (fabs(det - 0.0) <= kDblEps)
        return;

    Eig3 e;
    self_adjoint_eigen3(xx, xy, yy, xz, yz, zz, e);
    const double ev0 = fabs(e.val[0]);
    const double ev1 = fabs(e.val[1]);

    double nx = e.q[0][0], ny = e.q[1][0], nz = e.q[2][0];
    normalize3(nx, ny, nz); // eigenVector.normalized()
    const double d = -dot3(nx, ny, nz, f.cx, f.cy, f.cz);
    double px, py, pz, pd;
    if (d <= 0)
    {
        px = -nx; py = -ny; pz = -nz; pd = -d;
    }
    else
    {
        px = nx; py = ny; pz = nz; pd = d;
    }
    normalize3(px, py, pz); // PlaneCoordinates(normal, d) constructor
    normalize3(px, py, pz); // PlaneCoordinates::operator=
    f.nx = px;
    f.ny = py;
    f.nz = pz;
    f.d = pd;
    f.mse = ev0 * inv;
    f.score = ev1 / std_max(ev0, 1e-6);
    f.planar = true;
}

// Plane_Segment::can_be_merged (plane_segment.cpp:322-326): parent plane (n,d) vs child normal/centroid
__device__ __forceinline__ bool can_be_merged(double pnx, double pny, double pnz, double pd, double cnx, double cny,
                                              double cnz, double ccx, double ccy, double ccz, double maxDist,
                                              double cosMerge)
{
    const double cosAngle = dot3(pnx, pny, pnz, cnx, cny, cnz);
    const double dist = dot3(pnx, pny, pnz, ccx, ccy, ccz) + pd;
    return (cosAngle > cosMerge) & (fabs(dist) < maxDist); // no short-circuit: keeps the callers branch-free
}

// "Plane segment is not planar after merge" (primitive_detection.cpp:374, :497): counted in bits 8..15 of the frame's status.  Every
// lane of the frame holds the SAME count (the callers bump it in uniform control flow), so the OR-fold of the status keeps it.
__device__ __forceinline__ void status_count_not_planar(uint32_t& status)
{
    if (CAPE_FRAME_NOT_PLANAR_COUNT(status) < 255u)
        status += 1u << CAPE_FRAME_NOT_PLANAR_SHIFT;
}
// a parked frame's status: the flag bits live in lane 0 (the fold ORs them back), the count in every lane
__device__ __forceinline__ uint32_t status_resume(uint32_t parked, bool first)
{
    const uint32_t count = parked & (0xFFu << CAPE_FRAME_NOT_PLANAR_SHIFT);
    return (first ? (parked & ~count) : 0u) | count;
}

} // namespace cape
