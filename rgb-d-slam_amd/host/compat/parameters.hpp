// Mirror of the part of reference src/parameters.hpp the primitives library reads: the detection / matching constants
// (:67-95) and the process-global camera parameters (class Parameters, :119-191).  The reference fills Parameters from
// a YAML file (parse_file) or load_defaut(); here the setters are explicit.
#ifndef CAPE_COMPAT_PARAMETERS_HPP
#define CAPE_COMPAT_PARAMETERS_HPP
#include <string>

#include "types.hpp"

namespace rgbd_slam {

namespace parameters {
constexpr uint coreNumber = 8; // parameters.hpp:11
namespace detection {
constexpr uint depthMapPatchSize_px = 20;            // parameters.hpp:79-80
constexpr float maximumPlaneAngleForMerge_d = 18.0f; // :75
constexpr float maximumPlaneDistanceForMerge_mm = 50.0f;
} // namespace detection
namespace matching {
constexpr double maximumDistanceForPlaneMatch_mm = 100; // :92-95
constexpr double maximumAngleForPlaneMatch_d = 20.0;
constexpr float minimumPlaneOverlapToConsiderMatch = 0.4f;
} // namespace matching
} // namespace parameters

namespace compat_detail {
struct vector2_uint // Eigen::Vector<uint, 2> in the reference
{
    uint v[2] = {0, 0};
    [[nodiscard]] uint x() const noexcept { return v[0]; }
    [[nodiscard]] uint y() const noexcept { return v[1]; }
};
} // namespace compat_detail

class Parameters
{
  public:
    using vector2_uint = compat_detail::vector2_uint;

    static void load_defaut() noexcept; // parameters.cpp:59-74: 640x480, f = 550, c = (320, 240), identity extrinsics
    [[nodiscard]] static bool is_valid() noexcept { return _isValid; }

    [[nodiscard]] static vector2_uint get_camera_1_image_size() noexcept { return _camera1ImageSize; }
    [[nodiscard]] static vector2 get_camera_1_center() noexcept { return _camera1Center; }
    [[nodiscard]] static vector2 get_camera_1_focal() noexcept { return _camera1Focal; }
    [[nodiscard]] static matrix33 get_camera_1_intrinsics() noexcept;
    [[nodiscard]] static matrix44 get_camera_2_to_camera_1_transformation() noexcept { return _camera2toCamera1transformation; }

    // explicit setters (the reference's parse_file reads them from YAML)
    static void set_camera_1(uint width, uint height, double fx, double fy, double cx, double cy) noexcept;
    static void set_camera_2_to_camera_1_transformation(const matrix44& t) noexcept { _camera2toCamera1transformation = t; }

  private:
    inline static bool _isValid = false;
    inline static vector2_uint _camera1ImageSize;
    inline static vector2 _camera1Center;
    inline static vector2 _camera1Focal;
    inline static matrix44 _camera2toCamera1transformation = matrix44::Identity();
};

} // namespace rgbd_slam
#endif
