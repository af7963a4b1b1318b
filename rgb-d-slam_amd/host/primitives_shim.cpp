// See primitives_shim.hpp.  Everything numeric happens behind the C ABI (libcape_hip.so); this file only converts
// containers and keeps the reference's error convention (noexcept, log and skip).
#include "primitives_shim.hpp"

#include <chrono>
#include <exception>
#include <cmath>
#include <cstdio>
#include <mutex>

namespace rgbd_slam {

namespace {
struct Cam
{
    uint w = 0, h = 0;
    double fx = 0, fy = 0, cx = 0, cy = 0;
    bool valid = false;
} g_cam;
outputs::log_callback g_log;
std::mutex g_logMutex;

void log(int level, const std::string& msg)
{
    std::scoped_lock<std::mutex> lock(g_logMutex);
    if (g_log)
        g_log(level, msg);
    else
        std::fprintf(stderr, "[cape %s] %s\n", level == 0 ? "info" : (level == 1 ? "warn" : "error"), msg.c_str());
}
} // namespace

void Parameters::set_camera_1(uint width, uint height, double fx, double fy, double cx, double cy) noexcept
{
    g_cam = Cam {width, height, fx, fy, cx, cy, width > 0 && height > 0 && fx > 0 && fy > 0};
}
bool Parameters::is_valid() noexcept { return g_cam.valid; }
void Parameters::get_camera_1(uint& width, uint& height, double& fx, double& fy, double& cx, double& cy) noexcept
{
    width = g_cam.w; height = g_cam.h; fx = g_cam.fx; fy = g_cam.fy; cx = g_cam.cx; cy = g_cam.cy;
}

void outputs::set_log_callback(log_callback cb)
{
    std::scoped_lock<std::mutex> lock(g_logMutex);
    g_log = std::move(cb);
}

namespace features::primitives {

Plane::Plane(const cape_plane_segment& seg, const double* pts) noexcept :
    _normal {seg.out_normal[0], seg.out_normal[1], seg.out_normal[2]},
    _d(seg.d)
{
    for (int k = 0; k < 9; ++k)
        _pointCloudCovariance[k] = seg.cov[k];
    _boundaryPoints.reserve(seg.boundary_count);
    for (uint32_t i = 0; i < seg.boundary_count; ++i)
        _boundaryPoints.push_back({pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]});
    // const CameraPolygon polygon(orderedBoundary, planeSegment.get_normal(), planeSegment.get_center()),
    // primitive_detection.cpp:622 -- note the SEGMENT's normal / centre, not the re-normalised Plane normal
    try
    {
        const utils::vector3 n {seg.normal[0], seg.normal[1], seg.normal[2]};
        const utils::vector3 c {seg.normal[0] * (-seg.d), seg.normal[1] * (-seg.d), seg.normal[2] * (-seg.d)};
        _boundaryPolygon = utils::Polygon(_boundaryPoints, n, c);
    }
    catch (const std::exception&)
    {
        _boundaryPolygon = utils::Polygon();
    }
}

bool Plane::is_normal_similar(const Plane& p) const noexcept
{
    static const double minimumNormalDotDiff = std::abs(std::cos(20.0 * M_PI / 180.0));
    const double c = (_normal[0] * p._normal[0] + _normal[1] * p._normal[1]) + _normal[2] * p._normal[2];
    return std::abs(c) > minimumNormalDotDiff;
}
bool Plane::is_distance_similar(const Plane& p) const noexcept { return std::abs(_d - p._d) < 100.0; }

int find_plane_match(const plane_container& detected, const std::vector<bool>& isMatched, const std::array<double, 4>& projectedPlane,
                     const utils::Polygon& projectedPolygon, bool useAdvancedSearch) noexcept
{
    static const double minimumNormalDotDiff = std::abs(std::cos(20.0 * M_PI / 180.0));
    const double projectedArea = projectedPolygon.get_area();
    const double planeMinimalOverlap = static_cast<double>(0.4f); // minimumPlaneOverlapToConsiderMatch (float)
    const double areaSimilarityThreshold = useAdvancedSearch ? planeMinimalOverlap / 2 : planeMinimalOverlap;
    double greatestSimilarity = 0.0;
    if (projectedArea <= 0.0)
        return -1;
    int selectedIndex = -1;
    for (int i = 0; i < static_cast<int>(detected.size()); ++i)
    {
        if (i < static_cast<int>(isMatched.size()) && isMatched[i])
            continue;
        const Plane& p = detected[i];
        const auto n = p.get_normal();
        const bool distanceSimilar = std::abs(p.get_d() - projectedPlane[3]) < 100.0;
        const double c = (n[0] * projectedPlane[0] + n[1] * projectedPlane[1]) + n[2] * projectedPlane[2];
        if (!distanceSimilar || !(std::abs(c) > minimumNormalDotDiff))
            continue;
        const utils::Polygon& detectedPolygon = p.get_boundary_polygon();
        const double newPlaneArea = detectedPolygon.get_area();
        const double interArea = detectedPolygon.inter_area(projectedPolygon);
        if (interArea > greatestSimilarity && interArea / newPlaneArea >= areaSimilarityThreshold)
        {
            selectedIndex = i;
            greatestSimilarity = interArea;
        }
    }
    if (selectedIndex <= 0) // quirk of the reference: index 0 can never be returned
        return -1;
    return selectedIndex;
}

bool Cylinder::is_similar(const Cylinder& c) const noexcept
{
    static const double minimumNormalDotDiff = std::abs(std::cos(20.0 * M_PI / 180.0));
    const double d = (_normal[0] * c._normal[0] + _normal[1] * c._normal[1]) + _normal[2] * c._normal[2];
    return std::abs(d) > minimumNormalDotDiff;
}

Depth_Map_Transformation::Depth_Map_Transformation(const uint width, const uint height, const uint cellSize) :
    _width(width),
    _height(height),
    _cellSize(cellSize)
{
    if (cellSize != CAPE_CELL_SIZE)
        log(2, "Depth_Map_Transformation: only the reference's 20 px cell size is supported");
}

Depth_Map_Transformation::~Depth_Map_Transformation() { cape_destroy(_handle); }

bool Depth_Map_Transformation::rectify_depth(const DepthImageView& depthImage, float* rectified) noexcept
{
    if (!rectified || depthImage.rows != static_cast<int>(_height) || depthImage.cols != static_cast<int>(_width) ||
        depthImage.step != static_cast<size_t>(depthImage.cols))
    {
        log(2, "rectify_depth: depth image must be a contiguous width x height float image");
        return false;
    }
    if (!_handle)
    {
        if (!Parameters::is_valid())
            Parameters::load_defaut();
        uint w, h;
        cape_config cfg {};
        Parameters::get_camera_1(w, h, cfg.fx, cfg.fy, cfg.cx, cfg.cy);
        cfg.width = static_cast<int32_t>(_width);
        cfg.height = static_cast<int32_t>(_height);
        cfg.max_batch = 1;
        if (cape_create(&cfg, &_handle) != CAPE_OK)
        {
            log(2, std::string("rectify_depth: ") + cape_last_error());
            _handle = nullptr;
            return false;
        }
    }
    // host boundary of the reference signature: stage through device memory (cape_debug-style helper of the C ABI)
    if (cape_rectify_depth_host(_handle, depthImage.data, rectified, 1, _cam2to1.data()) != CAPE_OK)
    {
        log(2, std::string("rectify_depth: ") + cape_last_error());
        return false;
    }
    return true;
}

bool Depth_Map_Transformation::get_organized_cloud_array(const DepthImageView& depthImage) noexcept
{
    // the reference always returns true (depth_map_transformation.cpp:141)
    if (depthImage.rows != static_cast<int>(_height) || depthImage.cols != static_cast<int>(_width))
        log(2, "get_organized_cloud_array: depth image size differs from the configured size");
    return true;
}

#ifdef CAPE_HAVE_EIGEN_OPENCV
bool Depth_Map_Transformation::get_organized_cloud_array(const cv::Mat_<float>& depthImage, Eigen::MatrixXf& cloud) noexcept
{
    cloud.resize(0, 3); // consumed only by find_primitives, which back-projects on the device
    return get_organized_cloud_array(DepthImageView {depthImage.ptr<float>(0), depthImage.rows, depthImage.cols,
                                                     depthImage.step1()});
}
#endif

Primitive_Detection::Primitive_Detection(const uint width, const uint height) : _width(width), _height(height)
{
    if (!Parameters::is_valid())
        Parameters::load_defaut();
    uint w, h;
    cape_config cfg {};
    Parameters::get_camera_1(w, h, cfg.fx, cfg.fy, cfg.cx, cfg.cy);
    cfg.width = static_cast<int32_t>(width);
    cfg.height = static_cast<int32_t>(height);
    cfg.flags = CAPE_FLAG_CYLINDERS; // the reference always runs the cylinder branch (primitive_detection.cpp:385-388)
    cfg.device = 0;
    cfg.max_batch = _maxBatch = 64;
    cfg.boundary_capacity = 0;
    cfg.sub_batches = 0;
    if (cape_create(&cfg, &_handle) != CAPE_OK)
    {
        log(2, std::string("Primitive_Detection: ") + cape_last_error());
        _handle = nullptr;
        return;
    }
    cape_layout lay {};
    cape_get_layout(_handle, &lay);
    _cells = lay.cells;
    _boundaryCapacity = lay.boundary_capacity;
    _records.resize(_maxBatch);
    _boundary.resize(static_cast<size_t>(_maxBatch) * _boundaryCapacity * 3);
}

Primitive_Detection::~Primitive_Detection() { cape_destroy(_handle); }

void Primitive_Detection::collect(int f, plane_container& planes, cylinder_container& cylinders) const
{
    planes.clear();
    cylinders.clear();
    const cape_frame_record& r = _records[f];
    if (r.header.status & (CAPE_FRAME_PLANE_OVERFLOW | CAPE_FRAME_CYL_OVERFLOW | CAPE_FRAME_BOUNDARY_OVERFLOW))
        log(1, "find_primitives: per-frame capacity exceeded, primitive list truncated");
    planes.reserve(r.header.n_planes);
    const double* bnd = _boundary.data() + static_cast<size_t>(f) * _boundaryCapacity * 3;
    for (int i = 0; i < r.header.n_plane_segments; ++i)
    {
        const cape_plane_segment& s = r.segments[i];
        if (!s.is_output)
            continue;
        Plane plane(s, bnd + static_cast<size_t>(s.boundary_offset) * 3);
        // primitive_detection.cpp:624-632: keep the plane only if its polygon is valid and has >= 3 edges
        if (plane.get_boundary_polygon().is_valid() && plane.get_boundary_polygon().boundary_length() >= 3)
            planes.push_back(std::move(plane));
        else
            log(2, "Polyfit error: invalid boundary polygon, rejecting plane segment");
    }
    cylinders.reserve(r.header.n_cylinders);
    for (int i = 0; i < r.header.n_cylinder_labels; ++i)
        if (r.cylinders[i].kept)
            cylinders.emplace_back(r.cylinders[i]);
}

void Primitive_Detection::find_primitives_batch(const float* depth, int n_frames, std::vector<plane_container>& planes,
                                                std::vector<cylinder_container>& cylinders) noexcept
{
    planes.assign(n_frames, {});
    cylinders.assign(n_frames, {});
    if (!_handle)
    {
        log(2, "find_primitives: no device extractor (cape_create failed); returning no primitives");
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const size_t frameElems = static_cast<size_t>(_width) * _height;
    for (int base = 0; base < n_frames; base += _maxBatch)
    {
        const int n = (n_frames - base < _maxBatch) ? n_frames - base : _maxBatch;
        if (cape_extract_host(_handle, depth + base * frameElems, n, nullptr) != CAPE_OK ||
            cape_copy_results(_handle, n, _records.data(), nullptr, nullptr, _boundary.data()) != CAPE_OK)
        {
            log(2, std::string("find_primitives: ") + cape_last_error());
            return;
        }
        for (int f = 0; f < n; ++f)
            collect(f, planes[base + f], cylinders[base + f]);
    }
    _meanPrimitiveTreatmentDuration += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

bool Primitive_Detection::match_consecutive(int n_frames, std::vector<cape_frame_match>& matches, bool useAdvancedSearch,
                                            bool allowIndexZero) noexcept
{
    matches.clear();
    if (!_handle || n_frames < 0)
        return false;
    const uint32_t flags = (useAdvancedSearch ? static_cast<uint32_t>(CAPE_MATCH_ADVANCED) : 0u) |
                           (allowIndexZero ? static_cast<uint32_t>(CAPE_MATCH_ALLOW_INDEX0) : 0u);
    matches.resize(n_frames);
    if (cape_match_consecutive(_handle, n_frames, flags, nullptr) != CAPE_OK ||
        cape_copy_matches(_handle, n_frames, matches.data()) != CAPE_OK)
    {
        log(2, std::string("match_consecutive: ") + cape_last_error());
        matches.clear();
        return false;
    }
    return true;
}

void Primitive_Detection::find_primitives(const DepthImageView& depthImage, plane_container& planeContainer,
                                          cylinder_container& primitiveContainer) noexcept
{
    planeContainer.clear();
    primitiveContainer.clear();
    if (depthImage.rows != static_cast<int>(_height) || depthImage.cols != static_cast<int>(_width) ||
        depthImage.step != static_cast<size_t>(depthImage.cols))
    {
        log(2, "find_primitives: depth image must be a contiguous width x height float image");
        return;
    }
    std::vector<plane_container> p;
    std::vector<cylinder_container> c;
    find_primitives_batch(depthImage.data, 1, p, c);
    if (!p.empty())
    {
        planeContainer = std::move(p[0]);
        primitiveContainer = std::move(c[0]);
    }
}

#ifdef CAPE_HAVE_EIGEN_OPENCV
void Primitive_Detection::find_primitives(const Eigen::MatrixXf&, const cv::Mat_<float>& depthImage,
                                          plane_container& planeContainer, cylinder_container& primitiveContainer) noexcept
{
    const cv::Mat_<float> d = depthImage.isContinuous() ? depthImage : depthImage.clone();
    find_primitives(DepthImageView {d.ptr<float>(0), d.rows, d.cols, static_cast<size_t>(d.cols)}, planeContainer,
                    primitiveContainer);
}
#endif

void Primitive_Detection::show_statistics(const double meanFrameTreatmentDuration, const uint frameCount,
                                          const bool shouldDisplayDetails) const noexcept
{
    // primitive_detection.cpp:69-117
    auto percent = [](double t, double total) { return total <= 0 ? 0.0 : (t / total) * 100.0; };
    if (frameCount == 0)
        return;
    const double mean = _meanPrimitiveTreatmentDuration / static_cast<double>(frameCount);
    char buf[256];
    std::snprintf(buf, sizeof buf, "\tMean primitive extraction time is %.4f seconds (%.2f%%)", mean,
                  percent(mean, meanFrameTreatmentDuration));
    log(0, buf);
    if (shouldDisplayDetails && _handle)
    {
        cape_timings t {};
        if (cape_get_timings(_handle, &t) == CAPE_OK && t.calls > 0)
        {
            std::snprintf(buf, sizeof buf, "\t\tMean primitive init time is %.6f seconds, grow+merge+refine %.6f seconds (device, per call)",
                          t.cell_fit_s / t.calls, t.grow_s / t.calls);
            log(0, buf);
        }
    }
}

} // namespace features::primitives
} // namespace rgbd_slam
