#include "depth_map_transformation.hpp"

#include <string>

#include "outputs/logger.hpp"
#include "parameters.hpp"

namespace rgbd_slam::features::primitives {

Depth_Map_Transformation::Depth_Map_Transformation(const uint width, const uint height, const uint cellSize) :
    _width(width),
    _height(height),
    _cellSize(cellSize)
{
    _isOk = cellSize == CAPE_CELL_SIZE && width % CAPE_CELL_SIZE == 0 && height % CAPE_CELL_SIZE == 0;
    if (!_isOk)
        outputs::log_error("Depth_Map_Transformation: the image must be a whole number of 20 px cells (depthMapPatchSize_px)");
}

Depth_Map_Transformation::~Depth_Map_Transformation() { cape_destroy(_handle); }

bool Depth_Map_Transformation::get_organized_cloud_array(const depth_image& depthImage, matrixf& organizedCloudArray) noexcept
{
    if (depthImage.rows != static_cast<int>(_height) || depthImage.cols != static_cast<int>(_width))
        outputs::log_error("get_organized_cloud_array: depth image size differs from the configured size");
    organizedCloudArray.resize(0, 3); // consumed only by find_primitives, which back-projects on the device
    return true;
}

bool Depth_Map_Transformation::rectify_depth(const depth_image& depthImage, depth_image& rectifiedDepth) noexcept
{
    if (depthImage.rows != static_cast<int>(_height) || depthImage.cols != static_cast<int>(_width))
    {
        outputs::log_error("rectify_depth: depth image size differs from the configured size");
        return false;
    }
    if (!_handle)
    {
        if (!Parameters::is_valid())
            Parameters::load_defaut();
        cape_config cfg {};
        cfg.width = static_cast<int32_t>(_width);
        cfg.height = static_cast<int32_t>(_height);
        cfg.fx = Parameters::get_camera_1_focal().x();
        cfg.fy = Parameters::get_camera_1_focal().y();
        cfg.cx = Parameters::get_camera_1_center().x();
        cfg.cy = Parameters::get_camera_1_center().y();
        cfg.max_batch = 1;
        if (cape_create(&cfg, &_handle) != CAPE_OK)
        {
            outputs::log_error(std::string("rectify_depth: ") + cape_last_error());
            _handle = nullptr;
            return false;
        }
    }
    const depth_image src = depthImage.isContinuous() ? depthImage : depthImage.clone();
    rectifiedDepth.create(static_cast<int>(_height), static_cast<int>(_width));
    double T[16];
    const auto t = Parameters::get_camera_2_to_camera_1_transformation();
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            T[4 * r + c] = t(r, c);
    if (cape_rectify_depth_host(_handle, src.ptr<float>(0), rectifiedDepth.ptr<float>(0), 1, T) != CAPE_OK)
    {
        outputs::log_error(std::string("rectify_depth: ") + cape_last_error());
        return false;
    }
    return true;
}

} // namespace rgbd_slam::features::primitives
