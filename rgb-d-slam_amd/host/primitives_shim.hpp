// Umbrella header of the host-side replacement of the reference's `primitives` library (CMakeLists.txt:117-123).
// The replacement itself is host/overlay/features/primitives/*: files with the reference's names, class names,
// namespaces and signatures, compiled either inside the reference tree against its real Eigen / OpenCV / Boost types or
// here, dependency-free, against host/compat (see host/compat/README.md and INTEGRATION.md).
#pragma once
#include "overlay/features/primitives/depth_map_transformation.hpp"
#include "overlay/features/primitives/primitive_detection.hpp"
#include "compat/shape_primitives.hpp"
