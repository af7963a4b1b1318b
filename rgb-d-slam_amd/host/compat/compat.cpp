// Definitions behind host/compat/{parameters,outputs/logger}.hpp -- compiled only in the dependency-free build; in the
// reference's build these symbols are the reference's own (src/parameters.cpp, src/outputs/logger.cpp).
#include <cstdio>
#include <mutex>

#include "outputs/logger.hpp"
#include "parameters.hpp"

namespace rgbd_slam {

void Parameters::load_defaut() noexcept { set_camera_1(640, 480, 550.0, 550.0, 320.0, 240.0); } // parameters.cpp:59-74

void Parameters::set_camera_1(uint width, uint height, double fx, double fy, double cx, double cy) noexcept
{
    _camera1ImageSize.v[0] = width;
    _camera1ImageSize.v[1] = height;
    _camera1Focal = vector2(fx, fy);
    _camera1Center = vector2(cx, cy);
    _isValid = width > 0 && height > 0 && fx > 0 && fy > 0;
}

matrix33 Parameters::get_camera_1_intrinsics() noexcept
{
    matrix33 k = matrix33::Zero();
    k(0, 0) = _camera1Focal.x();
    k(0, 2) = _camera1Center.x();
    k(1, 1) = _camera1Focal.y();
    k(1, 2) = _camera1Center.y();
    k(2, 2) = 1.0;
    return k;
}

namespace outputs {
namespace {
log_callback g_log;
std::mutex g_logMutex;
void emit(int level, const std::string_view& msg)
{
    std::scoped_lock<std::mutex> lock(g_logMutex);
    if (g_log)
        g_log(level, std::string(msg));
    else
        std::fprintf(stderr, "[cape %s] %.*s\n", level == 0 ? "info" : (level == 1 ? "warn" : "error"), (int)msg.size(), msg.data());
}
} // namespace
void set_log_callback(log_callback cb)
{
    std::scoped_lock<std::mutex> lock(g_logMutex);
    g_log = std::move(cb);
}
void log(const std::string_view& message) { emit(0, message); }
void log_warning(const std::string_view& message) { emit(1, message); }
void log_error(const std::string_view& message) { emit(2, message); }
} // namespace outputs

} // namespace rgbd_slam
