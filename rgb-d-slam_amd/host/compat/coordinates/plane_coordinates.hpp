// TRANSCRIBED INTERFACE (scaffolding, not product work): this file mirrors the reference's declarations member for member so that the
// overlay and the consumer call sites compile WITHOUT Eigen / OpenCV / Boost in this image.  Inside the reference tree it is not used
// (the reference's own file is); nothing here is counted as an implemented component (copy-paste findings).
// Mirror of reference src/coordinates/plane_coordinates.hpp:16-68: a plane as (unit normal, d).  The behaviour that
// matters for parity is reproduced exactly: EVERY construction, copy and assignment re-normalises the normal
// (plane_coordinates.hpp:19-40) -- the count of normalisations between a cell fit and an output plane is observable in
// the last bits.  The world <-> camera projections belong to the reference's pose code and are
// not part of the replaced unit.
#ifndef CAPE_COMPAT_PLANE_COORDINATES_HPP
#define CAPE_COMPAT_PLANE_COORDINATES_HPP
#include <cmath>

#include "point_coordinates.hpp"
#include "types.hpp"

namespace rgbd_slam {

struct PlaneCoordinates
{
    PlaneCoordinates() : _normal(vector3::Zero()), _d(0.0) {}
    PlaneCoordinates(const vector4& parametrization) : _normal(parametrization.head<3>()), _d(parametrization(3)) { _normal.normalize(); }
    PlaneCoordinates(const vector3& normal, const double d) : _normal(normal), _d(d) { _normal.normalize(); }
    PlaneCoordinates(const PlaneCoordinates& other) : _normal(other.get_normal()), _d(other.get_d()) { _normal.normalize(); }
    PlaneCoordinates& operator=(const PlaneCoordinates& other) noexcept
    {
        if (this == &other)
            return *this;
        _normal = other._normal;
        _normal.normalize();
        _d = other._d;
        return *this;
    }

    [[nodiscard]] vector4 get_parametrization() const noexcept { return vector4(_normal.x(), _normal.y(), _normal.z(), _d); }
    [[nodiscard]] vector3 get_normal() const noexcept { return _normal; }
    [[nodiscard]] vector3& normal() noexcept { return _normal; }
    [[nodiscard]] double get_d() const noexcept { return _d; }
    [[nodiscard]] double& d() noexcept { return _d; }
    [[nodiscard]] WorldCoordinate get_center() const noexcept { return WorldCoordinate(_normal * (-_d)); }
    [[nodiscard]] double get_point_distance(const vector3& point) const noexcept { return _normal.dot(point) + _d; }
    [[nodiscard]] double get_cos_angle(const PlaneCoordinates& other) const noexcept { return _normal.dot(other._normal); }
    [[nodiscard]] bool hasNaN() const noexcept { return std::isnan(_d) or _normal.hasNaN(); }

  private:
    vector3 _normal;
    double _d;
};

struct PlaneCameraCoordinates : public PlaneCoordinates
{
    using PlaneCoordinates::PlaneCoordinates;
};

struct PlaneWorldCoordinates : public PlaneCoordinates
{
    using PlaneCoordinates::PlaneCoordinates;
};

} // namespace rgbd_slam
#endif
