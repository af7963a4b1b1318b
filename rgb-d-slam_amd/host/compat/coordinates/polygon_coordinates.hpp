// Mirror of reference src/coordinates/polygon_coordinates.hpp:18-95: polygons tagged by the space they live in.  The
// screen / world projections (to_world_space, to_screen_space, display ...) are the reference's pose and display code,
// outside the replaced unit; what the primitives library and its consumers' matching code need is the utils::Polygon
// interface (area, intersection, validity, boundary).
#ifndef CAPE_COMPAT_POLYGON_COORDINATES_HPP
#define CAPE_COMPAT_POLYGON_COORDINATES_HPP
#include "types.hpp"
#include "utils/polygon.hpp"

namespace rgbd_slam {

class CameraPolygon : public utils::Polygon
{
  public:
    using Polygon::Polygon;
    CameraPolygon() = default;
    CameraPolygon(const Polygon& other) : Polygon(other) {}
};

class WorldPolygon : public utils::Polygon
{
  public:
    using Polygon::Polygon;
    WorldPolygon() = default;
    WorldPolygon(const Polygon& other) : Polygon(other) {}
};

} // namespace rgbd_slam
#endif
