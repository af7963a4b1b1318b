// Mirror of the two point types of reference src/coordinates/point_coordinates.hpp:119-197 that the primitives
// library's public interface names (CameraCoordinate: Plane::get_center, Plane_Segment::get_centroid).
#ifndef CAPE_COMPAT_POINT_COORDINATES_HPP
#define CAPE_COMPAT_POINT_COORDINATES_HPP
#include "types.hpp"

namespace rgbd_slam {

struct WorldCoordinate : public vector3
{
    using vector3::vector3;
    WorldCoordinate() : vector3(vector3::Zero()) {}
    WorldCoordinate(const vector3& v) : vector3(v) {}
};

struct CameraCoordinate : public vector3
{
    using vector3::vector3;
    CameraCoordinate() : vector3(vector3::Zero()) {}
    CameraCoordinate(const vector3& v) : vector3(v) {}
    CameraCoordinate(const vector4& homogeneous) :
        vector3(homogeneous.x() / homogeneous[3], homogeneous.y() / homogeneous[3], homogeneous.z() / homogeneous[3])
    {
    }
};

} // namespace rgbd_slam
#endif
