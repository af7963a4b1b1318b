// TRANSCRIBED INTERFACE (scaffolding, not product work): this file mirrors the reference's declarations member for member so that the
// overlay and the consumer call sites compile WITHOUT Eigen / OpenCV / Boost in this image.  Inside the reference tree it is not used
// (the reference's own file is); nothing here is counted as an implemented component (copy-paste findings).
// See shape_primitives.hpp.  Follows reference src/features/primitives/shape_primitives.cpp:17-113 member by member.
#include "shape_primitives.hpp"

#include <cmath>

#include "outputs/logger.hpp"
#include "parameters.hpp"

namespace rgbd_slam::features::primitives {

namespace {
// |cos| of the largest angle at which two normals still match (shape_primitives.cpp:28-30, :72-73)
double minimum_normal_dot() noexcept
{
    static const double v = std::abs(std::cos(parameters::matching::maximumAngleForPlaneMatch_d * M_PI / 180.0));
    return v;
}
} // namespace

Cylinder::Cylinder(const Cylinder_Segment& cylinderSeg) : _radius(0)
{
    // mean radius of the fitted sub-segments (:17-24); a segment copy holds none, which gives 0 / 0 = NaN
    const uint count = cylinderSeg.get_segment_count();
    for (uint i = 0; i < count; ++i)
        _radius += cylinderSeg.get_radius(i);
    _radius /= count;
    _normal = cylinderSeg.get_normal();
}

Cylinder::Cylinder(const Cylinder& cylinder) : IPrimitive(), _normal(cylinder._normal), _radius(cylinder._radius) {}

bool Cylinder::is_similar(const Cylinder& cylinder) const noexcept
{
    return std::abs(_normal.dot(cylinder._normal)) > minimum_normal_dot();
}

double Cylinder::get_distance(const vector3&) const noexcept
{
    outputs::log_error("Error: get_point_distance is not implemented for Cylinder objects"); // as in the reference (:33-38)
    return 0;
}

Plane::Plane(const Plane_Segment& planeSeg, const CameraPolygon& boundaryPolygon) :
    _parametrization(planeSeg.get_normal(), planeSeg.get_plane_d()), // one more normalisation (:49)
    _pointCloudCovariance(planeSeg.get_point_cloud_covariance()),
    _boundaryPolygon(boundaryPolygon)
{
}

Plane::Plane(const Plane& plane) :
    IPrimitive(),
    _parametrization(plane._parametrization),
    _pointCloudCovariance(plane._pointCloudCovariance),
    _boundaryPolygon(plane._boundaryPolygon)
{
}

bool Plane::is_normal_similar(const Plane& plane) const noexcept { return is_normal_similar(plane._parametrization); }

bool Plane::is_normal_similar(const PlaneCameraCoordinates& planeParametrization) const noexcept
{
    return std::abs(_parametrization.get_cos_angle(planeParametrization)) > minimum_normal_dot();
}

bool Plane::is_distance_similar(const Plane& plane) const noexcept { return is_distance_similar(plane._parametrization); }

bool Plane::is_distance_similar(const PlaneCameraCoordinates& planeParametrization) const noexcept
{
    constexpr double maximumPlaneMatchDistance = parameters::matching::maximumDistanceForPlaneMatch_mm;
    return std::abs(_parametrization.get_d() - planeParametrization.get_d()) < maximumPlaneMatchDistance;
}

bool Plane::is_similar(const Cylinder&) const noexcept
{
    outputs::log_error("is_similar is not implemented between plane and cylinder"); // as in the reference (:88-93)
    return false;
}

double Plane::get_distance(const vector3& point) const noexcept { return get_parametrization().get_point_distance(point); }

} // namespace rgbd_slam::features::primitives
