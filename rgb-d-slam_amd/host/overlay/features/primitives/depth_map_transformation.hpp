// Replacement for reference src/features/primitives/depth_map_transformation.hpp:15-57 inside the `primitives` library.
// Same class, same three public members.  The organised cloud the reference builds here (3.7 MB per frame) exists only
// to be read by find_primitives; the native path back-projects inside the cell kernel (csrc/cape_cell_fit.hip), so
// get_organized_cloud_array validates and returns -- and find_primitives recomputes from the depth image, bit-identical
// by construction.  rectify_depth runs on the device (csrc/cape_rectify.hip).
#ifndef RGBDSLAM_FEATURES_PRIMITIVES_DEPTHMAPSEGMENTATION_HPP
#define RGBDSLAM_FEATURES_PRIMITIVES_DEPTHMAPSEGMENTATION_HPP

#include "cape_hip.h"
#include "depth_image.hpp"
#include "types.hpp"

namespace rgbd_slam::features::primitives {

class Depth_Map_Transformation
{
  public:
    Depth_Map_Transformation(const uint width, const uint height, const uint cellSize);
    ~Depth_Map_Transformation();
    Depth_Map_Transformation(const Depth_Map_Transformation&) = delete;
    Depth_Map_Transformation& operator=(const Depth_Map_Transformation&) = delete;

    // depth_map_transformation.cpp:23-87: registers the depth camera's image to the colour camera (Parameters'
    // camera2 -> camera1 transform); collisions keep the last source pixel in row-major order (MAKE_DETERMINISTIC)
    [[nodiscard]] bool rectify_depth(const depth_image& depthImage, depth_image& rectifiedDepth) noexcept;

    // depth_map_transformation.cpp:89-142: always true, like the reference; the matrix comes back with 0 rows
    [[nodiscard]] bool get_organized_cloud_array(const depth_image& depthImage, matrixf& organizedCloudArray) noexcept;

    [[nodiscard]] bool is_ok() const noexcept { return _isOk; }

  private:
    uint _width, _height, _cellSize;
    bool _isOk = false;
    cape_handle _handle = nullptr; // created on the first rectify_depth call
};

} // namespace rgbd_slam::features::primitives
#endif
