// Host-side mirror of the reference's `primitives` library interface (namespace rgbd_slam::features::primitives)
// on top of the C ABI in include/cape_hip.h.  Same class names, method names, argument meaning and error
// behaviour (everything noexcept, failures are logged through a callback and the frame yields no primitives) as
//   Depth_Map_Transformation  reference src/features/primitives/depth_map_transformation.hpp:15-57
//   Primitive_Detection       reference src/features/primitives/primitive_detection.hpp:27-241
//   Plane / Cylinder          reference src/features/primitives/shape_primitives.hpp:31-130
// so that src/rgbd_slam.cpp:48-57,109-112,291-297,335 compile against it unchanged once the Eigen / OpenCV
// overloads below are enabled (they are compiled only where those headers exist; this image has neither, so the
// POD views are what the tests exercise).  See INTEGRATION.md.
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/cape_hip.h"
#include "boundary_polygon.hpp"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && __has_include(<opencv2/core.hpp>)
#define CAPE_HAVE_EIGEN_OPENCV 1
#include <Eigen/Dense>
#include <opencv2/core.hpp>
#endif
#endif

namespace rgbd_slam {

using uint = unsigned int;

// Stand-in for the process-global camera parameters of the reference (src/parameters.hpp:119-191): intrinsics are
// read once, when the first detector is constructed (the reference caches them in function-local statics).
class Parameters
{
  public:
    static void load_defaut() noexcept { set_camera_1(640, 480, 550.0, 550.0, 320.0, 240.0); } // parameters.cpp:59-74
    static void set_camera_1(uint width, uint height, double fx, double fy, double cx, double cy) noexcept;
    static bool is_valid() noexcept;
    static void get_camera_1(uint& width, uint& height, double& fx, double& fy, double& cx, double& cy) noexcept;
};

namespace outputs {
// outputs::log / log_warning / log_error (src/outputs/logger.cpp:21-65) collapse into one callback; level 0 info,
// 1 warning, 2 error.  Default: stderr.
using log_callback = std::function<void(int level, const std::string& message)>;
void set_log_callback(log_callback cb);
} // namespace outputs

namespace features::primitives {

// row-major float32 depth image in millimetres (what cv::Mat_<float> holds), not owned
struct DepthImageView
{
    const float* data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0; // elements per row (>= cols)
};

// Plane, shape_primitives.hpp:73-126.  The boundary polygon (reference: CameraPolygon over Boost.Geometry + FLANN,
// primitive_detection.cpp:622) is built on the host by boundary_polygon.{hpp,cpp} ("next" row N1 of SURVEY.md 8f)
// from the boundary candidate points the device emits.
class Plane
{
  public:
    using vector3 = std::array<double, 3>;
    using matrix33 = std::array<double, 9>; // row-major

    Plane(const cape_plane_segment& seg, const double* boundaryPoints /* 3 x count */) noexcept;

    [[nodiscard]] vector3 get_normal() const noexcept { return _normal; }
    [[nodiscard]] double get_d() const noexcept { return _d; }
    [[nodiscard]] std::array<double, 4> get_parametrization() const noexcept { return {_normal[0], _normal[1], _normal[2], _d}; }
    [[nodiscard]] vector3 get_center() const noexcept { return {_normal[0] * (-_d), _normal[1] * (-_d), _normal[2] * (-_d)}; }
    [[nodiscard]] matrix33 get_point_cloud_covariance() const noexcept { return _pointCloudCovariance; }
    [[nodiscard]] const std::vector<vector3>& get_boundary_points() const noexcept { return _boundaryPoints; }
    [[nodiscard]] const utils::Polygon& get_boundary_polygon() const noexcept { return _boundaryPolygon; }
    // shape_primitives.cpp:66-86 (20 degrees, 100 mm; parameters.hpp:92-95)
    [[nodiscard]] bool is_normal_similar(const Plane& prim) const noexcept;
    [[nodiscard]] bool is_distance_similar(const Plane& prim) const noexcept;

  private:
    vector3 _normal;
    double _d;
    matrix33 _pointCloudCovariance;
    std::vector<vector3> _boundaryPoints;
    utils::Polygon _boundaryPolygon;
};

// Cylinder, shape_primitives.hpp:31-68 (public data members as in the reference)
class Cylinder
{
  public:
    explicit Cylinder(const cape_cylinder& c) noexcept : _normal {c.axis[0], c.axis[1], c.axis[2]}, _radius(c.radius) {}
    [[nodiscard]] bool is_similar(const Cylinder& prim) const noexcept;
    std::array<double, 3> _normal;
    double _radius;
};

using cylinder_container = std::vector<Cylinder>;
using plane_container = std::vector<Plane>;

// "Next" row N2: the selection loop of MapPlane::find_matches (src/map_management/map_features/map_primitive.cpp:91-161)
// for one map plane that the caller has already projected into camera space (parametrization nx,ny,nz,d and boundary
// polygon).  A detected plane is a candidate if |d - d'| < 100 mm and |n.n'| > cos 20 deg (shape_primitives.cpp:66-86);
// the candidate with the greatest polygon intersection area wins provided inter / area(detected) >= 0.4 (0.2 with
// useAdvancedSearch; parameters.hpp:90-95).  Returns the index of the selected detected plane or -1; like the
// reference it rejects index 0 (`if (selectedIndex <= 0)`, map_primitive.cpp:146).
int find_plane_match(const plane_container& detectedPlanes, const std::vector<bool>& isDetectedFeatureMatched,
                     const std::array<double, 4>& projectedPlane, const utils::Polygon& projectedPolygon,
                     bool useAdvancedSearch = false) noexcept;

// depth_map_transformation.hpp:15-57.  The organised cloud exists in the reference only to feed find_primitives
// (src/rgbd_slam.cpp:109-121); the native path back-projects inside the cell-fit kernel, so get_organized_cloud_array
// only validates sizes.  rectify_depth runs on the device through cape_rectify_depth.
class Depth_Map_Transformation
{
  public:
    Depth_Map_Transformation(const uint width, const uint height, const uint cellSize);
    ~Depth_Map_Transformation();
    Depth_Map_Transformation(const Depth_Map_Transformation&) = delete;
    Depth_Map_Transformation& operator=(const Depth_Map_Transformation&) = delete;
    [[nodiscard]] bool get_organized_cloud_array(const DepthImageView& depthImage) noexcept;
    // rectify_depth(depthImage, rectifiedDepth): `rectified` must hold rows*cols floats.  The camera2 -> camera1
    // matrix is Parameters::get_camera_2_to_camera_1_transformation() in the reference; here it is set explicitly
    // (identity by default = Parameters::load_defaut).
    [[nodiscard]] bool rectify_depth(const DepthImageView& depthImage, float* rectified) noexcept;
    void set_camera_2_to_camera_1_transformation(const std::array<double, 16>& rowMajor4x4) noexcept { _cam2to1 = rowMajor4x4; }
#ifdef CAPE_HAVE_EIGEN_OPENCV
    [[nodiscard]] bool get_organized_cloud_array(const cv::Mat_<float>& depthImage, Eigen::MatrixXf& organizedCloudArray) noexcept;
#endif
  private:
    uint _width, _height, _cellSize;
    cape_handle _handle = nullptr; // created on the first rectify_depth call
    std::array<double, 16> _cam2to1 {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};

// primitive_detection.hpp:27-241
class Primitive_Detection
{
  public:
    Primitive_Detection(const uint width, const uint height);
    ~Primitive_Detection();
    Primitive_Detection(const Primitive_Detection&) = delete;
    Primitive_Detection& operator=(const Primitive_Detection&) = delete;

    // find_primitives(depthMatrix, depthImage, planeContainer, primitiveContainer): depthMatrix is not needed
    void find_primitives(const DepthImageView& depthImage, plane_container& planeContainer,
                         cylinder_container& primitiveContainer) noexcept;
#ifdef CAPE_HAVE_EIGEN_OPENCV
    void find_primitives(const Eigen::MatrixXf& depthMatrix, const cv::Mat_<float>& depthImage,
                         plane_container& planeContainer, cylinder_container& primitiveContainer) noexcept;
#endif
    // batch extension (frames are independent: SURVEY.md 8e): depth = n_frames contiguous images on the host
    void find_primitives_batch(const float* depth, int n_frames, std::vector<plane_container>& planes,
                               std::vector<cylinder_container>& cylinders) noexcept;

    // "Next" row N2, device part: candidate matches between consecutive frames of the frames still resident on the
    // device (the last chunk of <= 64 frames of find_primitives_batch), computed on cell masks by
    // cape_match_consecutive.  Plane indices count the segments with is_output, i.e. the planes BEFORE the polygon
    // validity test drops any; a host-side polygon check with find_plane_match confirms a candidate.
    bool match_consecutive(int n_frames, std::vector<cape_frame_match>& matches, bool useAdvancedSearch = false,
                           bool allowIndexZero = false) noexcept;

    void show_statistics(const double meanFrameTreatmentDuration, const uint frameCount,
                         const bool shouldDisplayDetails = false) const noexcept;

    [[nodiscard]] bool is_ready() const noexcept { return _handle != nullptr; }

  private:
    cape_handle _handle = nullptr;
    uint _width, _height;
    int _cells = 0, _boundaryCapacity = 0;
    int _maxBatch = 0;
    mutable double _meanPrimitiveTreatmentDuration = 0.0; // seconds, accumulated like primitive_detection.cpp:164
    std::vector<cape_frame_record> _records;
    std::vector<double> _boundary;
    void collect(int frame, plane_container& planes, cylinder_container& cylinders) const;
};

} // namespace features::primitives
} // namespace rgbd_slam
