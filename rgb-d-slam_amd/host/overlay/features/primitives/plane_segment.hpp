// Replacement for reference src/features/primitives/plane_segment.hpp inside the `primitives` library
// (CMakeLists.txt:117-123).  Same class name, namespace and public read interface (plane_segment.hpp:66-113), plus ONE
// addition: a constructor that injects the result the GPU computed (cape_plane_segment, include/cape_hip.h).  The
// reference's Plane(const Plane_Segment&, const CameraPolygon&) (shape_primitives.cpp:48-56) only reads get_normal(),
// get_plane_d() and get_point_cloud_covariance(), so its own shape_primitives.{hpp,cpp} compile against this header
// unchanged.
//
// What is NOT here: init_plane_segment / fit_plane -- the per-cell accumulation and the PCA are the hot path and run
// on the device (csrc/cape_cell_fit.hip, csrc/cape_grow.hip); this library has no CPU implementation of them.
#ifndef RGBDSLAM_FEATURES_PRIMITIVES_PLANESEGMENT_HPP
#define RGBDSLAM_FEATURES_PRIMITIVES_PLANESEGMENT_HPP

#include <cmath>
#include <limits>

#include "cape_hip.h"
#include "coordinates/plane_coordinates.hpp"
#include "coordinates/point_coordinates.hpp"
#include "parameters.hpp"
#include "types.hpp"

namespace rgbd_slam::features::primitives {

class Plane_Segment
{
  public:
    Plane_Segment() = default;
    Plane_Segment(const Plane_Segment& seg) = default; // PlaneCoordinates' copy re-normalises, like the reference's copy (:306-320)

    // the injecting constructor: one element of _planeSegments after merge_planes, as the device left it
    explicit Plane_Segment(const cape_plane_segment& record);

    static void set_static_members(const uint cellWidth, const uint pointPerCellCount) noexcept
    {
        _cellWidth = cellWidth;
        _ptsPerCellCount = pointPerCellCount;
    }

    // plane_segment.cpp:286-304: sums and point count of the other segment are added, nothing is refitted
    void expand_segment(const Plane_Segment& planeSegment) noexcept;
    // plane_segment.cpp:322-326
    [[nodiscard]] double get_cos_angle(const Plane_Segment& p) const noexcept { return _parametrization.get_cos_angle(p._parametrization); }
    [[nodiscard]] double get_point_distance(const vector3& point) const noexcept { return _parametrization.get_point_distance(point); }
    [[nodiscard]] bool can_be_merged(const Plane_Segment& p, const double maxMatchDistance) const noexcept;
    void clear_plane_parameters() noexcept;

    // plane_segment.cpp:192-203: inverse of the moment matrix; evaluated on the device (same cofactor inverse) for the
    // segments that become planes, carried here
    [[nodiscard]] matrix33 get_point_cloud_covariance() const { return _pointCloudCovariance; }

    [[nodiscard]] double get_MSE() const noexcept { return _MSE; }
    [[nodiscard]] vector3 get_normal() const noexcept { return _parametrization.get_normal(); }
    [[nodiscard]] CameraCoordinate get_centroid() const noexcept { return _centroid; }
    [[nodiscard]] CameraCoordinate get_center() const noexcept { return CameraCoordinate(_parametrization.get_center()); }
    [[nodiscard]] double get_plane_d() const noexcept { return _parametrization.get_d(); }
    [[nodiscard]] vector4 get_parametrization() const noexcept { return _parametrization.get_parametrization(); }
    [[nodiscard]] bool is_planar() const noexcept { return _isPlanar; }
    [[nodiscard]] double get_score() const noexcept { return _score; }
    [[nodiscard]] uint get_point_count() const noexcept { return _pointCount; }

  private:
    static inline uint _ptsPerCellCount = 400;
    static inline uint _cellWidth = 20;

    uint _pointCount = 0;
    double _score = 0.0;
    double _MSE = std::numeric_limits<double>::max();
    bool _isPlanar = false;
    CameraCoordinate _centroid;
    PlaneCoordinates _parametrization;
    matrix33 _pointCloudCovariance = matrix33::Zero();
    double _Sx = 0.0, _Sy = 0.0, _Sz = 0.0, _Sxs = 0.0, _Sys = 0.0, _Szs = 0.0, _Sxy = 0.0, _Syz = 0.0, _Szx = 0.0;
};

} // namespace rgbd_slam::features::primitives
#endif
