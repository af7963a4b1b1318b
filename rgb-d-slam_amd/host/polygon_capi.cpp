// Test hook: the host boundary-polygon class behind a C entry point, so that tests/test_gpu_polygon.py can compare the
// device polygons (cape_build_polygons) with it vertex for vertex through ctypes.  Not part of the product's C ABI.
#include <cstring>
#include <exception>
#include <vector>

#include "boundary_polygon.hpp"

extern "C" int cape_host_polygon(const double* points3, int n, const double* normal, const double* center, double* ring_out, int capacity,
                                 int* count_out, double* area_out, double* x_axis_out, double* y_axis_out, int* valid_out)
{
    using rgbd_slam::vector3;
    try
    {
        std::vector<vector3> pts;
        pts.reserve(n);
        for (int i = 0; i < n; ++i)
            pts.emplace_back(points3[3 * i], points3[3 * i + 1], points3[3 * i + 2]);
        const rgbd_slam::utils::Polygon poly(pts, vector3(normal[0], normal[1], normal[2]), vector3(center[0], center[1], center[2]));
        const auto& ring = poly.boundary();
        *count_out = static_cast<int>(ring.size());
        for (size_t i = 0; i < ring.size() && static_cast<int>(i) < capacity; ++i)
        {
            ring_out[2 * i] = ring[i][0];
            ring_out[2 * i + 1] = ring[i][1];
        }
        *area_out = poly.get_area();
        for (int k = 0; k < 3; ++k)
        {
            x_axis_out[k] = poly.get_x_axis()[k];
            y_axis_out[k] = poly.get_y_axis()[k];
        }
        *valid_out = (poly.is_valid() && poly.boundary_length() >= 3) ? 1 : 0;
        return 0;
    }
    catch (const std::exception&)
    {
        *count_out = 0;
        *valid_out = 0;
        return 1; // the constructor threw: fewer than 3 points / normal not unit
    }
}

// Polygon::inter_area of two polygons given by their rings and frames (the public explicit-ring constructor, like the overlay
// builds its CameraPolygons from the device's vertices): `a` is the detected polygon, `b` the projected one
// (map_primitive.cpp:137).  tests/test_gpu_match_polygon.py compares cape_match_polygons with it bit for bit.
extern "C" double cape_host_polygon_inter_area(const double* ring_a, int na, const double* x_a, const double* y_a, const double* c_a,
                                               const double* ring_b, int nb, const double* x_b, const double* y_b, const double* c_b,
                                               double* area_a_out, double* area_b_out)
{
    using rgbd_slam::vector2;
    using rgbd_slam::vector3;
    std::vector<vector2> ra, rb;
    for (int i = 0; i < na; ++i)
        ra.emplace_back(ring_a[2 * i], ring_a[2 * i + 1]);
    for (int i = 0; i < nb; ++i)
        rb.emplace_back(ring_b[2 * i], ring_b[2 * i + 1]);
    const rgbd_slam::utils::Polygon a(ra, vector3(x_a[0], x_a[1], x_a[2]), vector3(y_a[0], y_a[1], y_a[2]), vector3(c_a[0], c_a[1], c_a[2]));
    const rgbd_slam::utils::Polygon b(rb, vector3(x_b[0], x_b[1], x_b[2]), vector3(y_b[0], y_b[1], y_b[2]), vector3(c_b[0], c_b[1], c_b[2]));
    if (area_a_out)
        *area_a_out = a.get_area();
    if (area_b_out)
        *area_b_out = b.get_area();
    return a.inter_area(b);
}

// The same with the projected polygon seen through a pose first: a.inter_area(b.to_camera_space(worldToCamera))
// (map_primitive.cpp:103 then :137); plane_in / plane_out: (nx, ny, nz, d) of the map plane before / after
// to_camera_coordinates (map_primitive.cpp:100-101).  tests/test_gpu_match_pose.py compares cape_match_polygons_pose with it.
extern "C" double cape_host_polygon_inter_area_pose(const double* ring_a, int na, const double* x_a, const double* y_a, const double* c_a,
                                                    const double* ring_b, int nb, const double* x_b, const double* y_b, const double* c_b,
                                                    const double* world_to_camera, const double* plane_in, double* plane_out)
{
    using rgbd_slam::vector2;
    using rgbd_slam::vector3;
    std::vector<vector2> ra, rb;
    for (int i = 0; i < na; ++i)
        ra.emplace_back(ring_a[2 * i], ring_a[2 * i + 1]);
    for (int i = 0; i < nb; ++i)
        rb.emplace_back(ring_b[2 * i], ring_b[2 * i + 1]);
    const rgbd_slam::utils::Polygon a(ra, vector3(x_a[0], x_a[1], x_a[2]), vector3(y_a[0], y_a[1], y_a[2]), vector3(c_a[0], c_a[1], c_a[2]));
    const rgbd_slam::utils::Polygon b(rb, vector3(x_b[0], x_b[1], x_b[2]), vector3(y_b[0], y_b[1], y_b[2]), vector3(c_b[0], c_b[1], c_b[2]));
    if (plane_in && plane_out)
        rgbd_slam::utils::plane_to_camera(plane_in, plane_in[3], world_to_camera, plane_out, plane_out + 3);
    return a.inter_area(b.to_camera_space(world_to_camera));
}
