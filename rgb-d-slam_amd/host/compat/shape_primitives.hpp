// TRANSCRIBED INTERFACE (scaffolding, not product work): this file mirrors the reference's declarations member for member so that the
// overlay and the consumer call sites compile WITHOUT Eigen / OpenCV / Boost in this image.  Inside the reference tree it is not used
// (the reference's own file is); nothing here is counted as an implemented component (copy-paste findings).
// Dependency-free twin of reference src/features/primitives/shape_primitives.hpp:31-130 -- the value types
// find_primitives returns.  Class names, constructors, member functions, return types and public data members are the
// reference's; when the overlay is built inside the reference tree the reference's OWN shape_primitives.{hpp,cpp} are
// kept instead of this pair (they compile against the overlay's plane_segment.hpp / cylinder_segment.hpp unchanged).
#ifndef RGBDSLAM_FEATURES_PRIMITIVES_PRIMITIVES_HPP
#define RGBDSLAM_FEATURES_PRIMITIVES_PRIMITIVES_HPP

#include <vector>

#include "coordinates/point_coordinates.hpp"
#include "coordinates/polygon_coordinates.hpp"
#include "cylinder_segment.hpp"
#include "plane_segment.hpp"
#include "types.hpp"

namespace rgbd_slam::features::primitives {

class IPrimitive
{
  public:
    IPrimitive() = default;

  private:
    IPrimitive& operator=(const IPrimitive&) = delete;
};

class Cylinder : public IPrimitive
{
  public:
    Cylinder(const Cylinder_Segment& cylinderSeg);
    Cylinder(const Cylinder& cylinder);

    [[nodiscard]] bool is_similar(const Cylinder& prim) const noexcept;
    [[nodiscard]] double get_distance(const vector3& point) const noexcept;

    vector3 _normal;
    double _radius;

    ~Cylinder() = default;

  private:
    Cylinder() = delete;
    Cylinder& operator=(const Cylinder&) = delete;
};

class Plane : public IPrimitive
{
  public:
    Plane(const Plane_Segment& planeSeg, const CameraPolygon& boundaryPolygon);
    Plane(const Plane& plane);

    [[nodiscard]] bool is_normal_similar(const Plane& prim) const noexcept;
    [[nodiscard]] bool is_normal_similar(const PlaneCameraCoordinates& planeParametrization) const noexcept;
    [[nodiscard]] bool is_distance_similar(const Plane& prim) const noexcept;
    [[nodiscard]] bool is_distance_similar(const PlaneCameraCoordinates& planeParametrization) const noexcept;
    [[nodiscard]] bool is_similar(const Cylinder& prim) const noexcept;

    [[nodiscard]] vector3 get_normal() const noexcept { return _parametrization.get_normal(); }
    [[nodiscard]] double get_d() const noexcept { return _parametrization.get_d(); }
    [[nodiscard]] PlaneCameraCoordinates get_parametrization() const noexcept { return _parametrization; }
    [[nodiscard]] CameraCoordinate get_center() const noexcept { return CameraCoordinate(_parametrization.get_center()); }
    [[nodiscard]] matrix33 get_point_cloud_covariance() const noexcept { return _pointCloudCovariance; }
    [[nodiscard]] CameraPolygon get_boundary_polygon() const noexcept { return _boundaryPolygon; }

    ~Plane() = default;

  private:
    [[nodiscard]] double get_distance(const vector3& point) const noexcept;

    PlaneCameraCoordinates _parametrization;
    matrix33 _pointCloudCovariance;
    const CameraPolygon _boundaryPolygon;

    Plane() = delete;
    Plane& operator=(const Plane&) = delete;
};

using cylinder_container = std::vector<Cylinder>;
using plane_container = std::vector<Plane>;

} // namespace rgbd_slam::features::primitives
#endif
