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
