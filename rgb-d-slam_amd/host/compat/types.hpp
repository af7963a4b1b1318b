// Dependency-free mirror of the type names of reference src/types.hpp:16-32 (see host/compat/README.md): small
// fixed-size value classes with the subset of the Eigen interface that the overlay sources and the consumers' call
// sites (map_primitive.cpp:102-153,217-278, plane_with_tracking.cpp:33-48) use.  In the reference's build the same
// names are Eigen types.
#ifndef CAPE_COMPAT_TYPES_HPP
#define CAPE_COMPAT_TYPES_HPP

#include <array>
#include <cmath>
#include <cstddef>
#include <vector>

namespace rgbd_slam {

using uint = unsigned int;
constexpr double EulerToRadian = M_PI / 180.0;

template <size_t N> struct Vec : public std::array<double, N>
{
    using Base = std::array<double, N>;
    Vec() : Base {} {}
    Vec(const Base& a) : Base(a) {}
    template <typename... T, typename = std::enable_if_t<sizeof...(T) == N && (N > 1)>>
    Vec(T... v) : Base {static_cast<double>(v)...}
    {
    }
    static Vec Zero() { return Vec(); }

    [[nodiscard]] double x() const noexcept { return (*this)[0]; }
    [[nodiscard]] double y() const noexcept { return (*this)[1]; }
    [[nodiscard]] double z() const noexcept { static_assert(N >= 3); return (*this)[2]; }
    double& x() noexcept { return (*this)[0]; }
    double& y() noexcept { return (*this)[1]; }
    double& z() noexcept { static_assert(N >= 3); return (*this)[2]; }
    [[nodiscard]] double operator()(size_t i) const noexcept { return (*this)[i]; }
    double& operator()(size_t i) noexcept { return (*this)[i]; }

    // fixed-size reductions in the order of an SSE2 Packet2d evaluation for three components, (a0 + a1) + a2 ...
    [[nodiscard]] double dot(const Vec& o) const noexcept
    {
        double s = (*this)[0] * o[0];
        for (size_t i = 1; i < N; ++i)
            s = s + (*this)[i] * o[i];
        return s;
    }
    [[nodiscard]] double squaredNorm() const noexcept { return dot(*this); }
    [[nodiscard]] double norm() const noexcept { return std::sqrt(squaredNorm()); }
    // Eigen normalize(): z = squaredNorm(); if (z > 0) v /= sqrt(z)
    void normalize() noexcept
    {
        const double z = squaredNorm();
        if (z > 0)
        {
            const double s = std::sqrt(z);
            for (size_t i = 0; i < N; ++i)
                (*this)[i] /= s;
        }
    }
    [[nodiscard]] Vec normalized() const noexcept
    {
        Vec r(*this);
        r.normalize();
        return r;
    }
    [[nodiscard]] bool hasNaN() const noexcept
    {
        for (size_t i = 0; i < N; ++i)
            if (std::isnan((*this)[i]))
                return true;
        return false;
    }
    [[nodiscard]] bool isApprox(const Vec& o, double prec = 1e-12) const noexcept
    {
        Vec d;
        for (size_t i = 0; i < N; ++i)
            d[i] = (*this)[i] - o[i];
        const double a = squaredNorm(), b = o.squaredNorm();
        return d.squaredNorm() <= prec * prec * (a < b ? a : b);
    }
    template <size_t K> [[nodiscard]] Vec<K> head() const noexcept
    {
        static_assert(K <= N);
        Vec<K> r;
        for (size_t i = 0; i < K; ++i)
            r[i] = (*this)[i];
        return r;
    }
    [[nodiscard]] Vec<3> cross(const Vec<3>& o) const noexcept
    {
        static_assert(N == 3);
        return Vec<3>((*this)[1] * o[2] - (*this)[2] * o[1], (*this)[2] * o[0] - (*this)[0] * o[2],
                      (*this)[0] * o[1] - (*this)[1] * o[0]);
    }
    Vec operator-() const noexcept
    {
        Vec r;
        for (size_t i = 0; i < N; ++i)
            r[i] = -(*this)[i];
        return r;
    }
    Vec& operator+=(const Vec& o) noexcept
    {
        for (size_t i = 0; i < N; ++i)
            (*this)[i] += o[i];
        return *this;
    }
};
template <size_t N> Vec<N> operator+(Vec<N> a, const Vec<N>& b) noexcept { return a += b; }
template <size_t N> Vec<N> operator-(const Vec<N>& a, const Vec<N>& b) noexcept
{
    Vec<N> r;
    for (size_t i = 0; i < N; ++i)
        r[i] = a[i] - b[i];
    return r;
}
template <size_t N> Vec<N> operator*(const Vec<N>& a, double s) noexcept
{
    Vec<N> r;
    for (size_t i = 0; i < N; ++i)
        r[i] = a[i] * s;
    return r;
}
template <size_t N> Vec<N> operator*(double s, const Vec<N>& a) noexcept { return a * s; }
template <size_t N> Vec<N> operator/(const Vec<N>& a, double s) noexcept
{
    Vec<N> r;
    for (size_t i = 0; i < N; ++i)
        r[i] = a[i] / s;
    return r;
}

using vector2 = Vec<2>;
using vector3 = Vec<3>;
using vector4 = Vec<4>;
using vector3_vector = std::vector<vector3>;
using vectorb = std::vector<bool>;

// row-major R x C matrix of doubles; m(r, c) like Eigen
template <size_t R, size_t C> struct Mat
{
    std::array<double, R * C> v {};
    static Mat Zero() { return Mat(); }
    static Mat Identity()
    {
        Mat m;
        for (size_t i = 0; i < (R < C ? R : C); ++i)
            m(i, i) = 1.0;
        return m;
    }
    [[nodiscard]] double operator()(size_t r, size_t c) const noexcept { return v[r * C + c]; }
    double& operator()(size_t r, size_t c) noexcept { return v[r * C + c]; }
    void setZero() noexcept { v.fill(0.0); }
    [[nodiscard]] bool hasNaN() const noexcept
    {
        for (double e : v)
            if (std::isnan(e))
                return true;
        return false;
    }
    [[nodiscard]] Vec<(R < C ? R : C)> diagonal() const noexcept
    {
        Vec<(R < C ? R : C)> d;
        for (size_t i = 0; i < (R < C ? R : C); ++i)
            d[i] = (*this)(i, i);
        return d;
    }
    [[nodiscard]] Mat<C, R> transpose() const noexcept
    {
        Mat<C, R> t;
        for (size_t r = 0; r < R; ++r)
            for (size_t c = 0; c < C; ++c)
                t(c, r) = (*this)(r, c);
        return t;
    }
    [[nodiscard]] bool operator==(const Mat& o) const noexcept { return v == o.v; }
};
template <size_t R, size_t C> Vec<R> operator*(const Mat<R, C>& m, const Vec<C>& x) noexcept
{
    Vec<R> y;
    for (size_t r = 0; r < R; ++r)
    {
        double s = m(r, 0) * x[0];
        for (size_t c = 1; c < C; ++c)
            s = s + m(r, c) * x[c];
        y[r] = s;
    }
    return y;
}
using matrix33 = Mat<3, 3>;
using matrix44 = Mat<4, 4>;

// Eigen::MatrixXf stand-in for the organised cloud argument: the native path back-projects on the device, the matrix
// is only carried through the reference's call signatures and stays empty (depth_map_transformation.cpp:96 resizes it)
struct matrixf
{
    [[nodiscard]] long rows() const noexcept { return _rows; }
    [[nodiscard]] long cols() const noexcept { return _cols; }
    void resize(long r, long c) noexcept
    {
        _rows = r;
        _cols = c;
    }

  private:
    long _rows = 0, _cols = 0;
};

template <class T> T constexpr inline SQR(const T x) { return x * x; }

} // namespace rgbd_slam
#endif
