// Replacement for reference src/features/primitives/cylinder_segment.hpp inside the `primitives` library: the read
// interface that Cylinder(const Cylinder_Segment&) uses (shape_primitives.cpp:17-24: get_segment_count, get_radius,
// get_normal) plus the constructor that injects the device's result (cape_cylinder).  The RANSAC fit itself
// (cylinder_segment.cpp:35-322) runs on the device (csrc/cape_cylinder.h).
#ifndef RGBDSLAM_FEATURES_PRIMITIVES_CYLINDERSEGMENT_HPP
#define RGBDSLAM_FEATURES_PRIMITIVES_CYLINDERSEGMENT_HPP

#include <vector>

#include "cape_hip.h"
#include "plane_segment.hpp"
#include "types.hpp"

namespace rgbd_slam::features::primitives {

class Cylinder_Segment
{
  public:
    // One kept entry of cylinder2regionMap.  The reference hands Cylinder a COPY made by Cylinder_Segment(const
    // Cylinder_Segment&) (cylinder_segment.cpp:23-29), which keeps the axis but none of the fitted sub-segments, so the
    // copy has get_segment_count() == 0 and the Cylinder's mean radius comes out as 0/0 = NaN: same here.
    explicit Cylinder_Segment(const cape_cylinder& record) : _axis(record.axis[0], record.axis[1], record.axis[2]) {}
    Cylinder_Segment(const Cylinder_Segment& seg) = default;

    [[nodiscard]] uint get_segment_count() const noexcept { return static_cast<uint>(_radius.size()); }
    [[nodiscard]] double get_MSE_at(const uint index) const noexcept { return _MSE[index]; }
    [[nodiscard]] double get_radius(const uint index) const noexcept { return _radius[index]; }
    [[nodiscard]] vector3 get_normal() const noexcept { return _axis; }
    [[nodiscard]] double get_normal_similarity(const Cylinder_Segment& other) const noexcept { return std::abs(_axis.dot(other._axis)); }

  private:
    vector3 _axis;
    std::vector<double> _radius;
    std::vector<double> _MSE;
};

} // namespace rgbd_slam::features::primitives
#endif
