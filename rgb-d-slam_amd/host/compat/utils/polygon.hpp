// reference src/utils/polygon.hpp names rgbd_slam::utils::Polygon; the dependency-free implementation lives in
// host/boundary_polygon.{hpp,cpp} ("next" row N1 of SURVEY.md 8f).
#ifndef CAPE_COMPAT_UTILS_POLYGON_HPP
#define CAPE_COMPAT_UTILS_POLYGON_HPP
#include "../../boundary_polygon.hpp"
#endif
