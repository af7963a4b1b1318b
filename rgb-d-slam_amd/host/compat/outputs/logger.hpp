// Mirror of reference src/outputs/logger.hpp:12-17: the three log functions, routed to one replaceable callback
// (level 0 info, 1 warning, 2 error; default stderr).
#ifndef CAPE_COMPAT_LOGGER_HPP
#define CAPE_COMPAT_LOGGER_HPP
#include <functional>
#include <string>
#include <string_view>

namespace rgbd_slam::outputs {

using log_callback = std::function<void(int level, const std::string& message)>;
void set_log_callback(log_callback cb);

void log(const std::string_view& message);
void log_warning(const std::string_view& message);
void log_error(const std::string_view& message);

} // namespace rgbd_slam::outputs
#endif
