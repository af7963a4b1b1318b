// The depth image type of the primitives library's public interface.  The reference passes cv::Mat_<float>
// (depth_map_transformation.hpp:29-39, primitive_detection.hpp:41-44): where OpenCV exists `depth_image` IS that type,
// so the overlay's signatures are the reference's to the letter.  In the dependency-free build it is a minimal
// row-major float image with the members of cv::Mat_<float> that the overlay sources use (rows, cols, ptr<float>(r),
// isContinuous(), create(), clone(), empty(), operator()(r, c)).
#ifndef CAPE_OVERLAY_DEPTH_IMAGE_HPP
#define CAPE_OVERLAY_DEPTH_IMAGE_HPP

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#define CAPE_HAVE_OPENCV 1
#endif
#endif

#ifdef CAPE_HAVE_OPENCV
#include <opencv2/core.hpp>
namespace rgbd_slam::features::primitives {
using depth_image = cv::Mat_<float>;
}
#else
#include <cstddef>
#include <memory>
#include <vector>
namespace rgbd_slam::features::primitives {
class depth_image
{
  public:
    int rows = 0, cols = 0;
    depth_image() = default;
    depth_image(int rows_, int cols_) { create(rows_, cols_); }
    // wraps caller-owned pixels (cv::Mat_<float>(rows, cols, data, step)): stepBytes = bytes from one row to the next
    depth_image(int rows_, int cols_, float* data, size_t stepBytes = 0) :
        rows(rows_), cols(cols_), _data(data), _stepElems(stepBytes ? stepBytes / sizeof(float) : (size_t)cols_)
    {
    }
    void create(int rows_, int cols_)
    {
        _own = std::make_shared<std::vector<float>>((size_t)rows_ * cols_, 0.0f);
        _data = _own->data();
        rows = rows_;
        cols = cols_;
        _stepElems = (size_t)cols_;
    }
    template <typename T = float> [[nodiscard]] const T* ptr(int r = 0) const noexcept { return _data + (size_t)r * _stepElems; }
    template <typename T = float> T* ptr(int r = 0) noexcept { return _data + (size_t)r * _stepElems; }
    [[nodiscard]] float operator()(int r, int c) const noexcept { return _data[(size_t)r * _stepElems + c]; }
    float& operator()(int r, int c) noexcept { return _data[(size_t)r * _stepElems + c]; }
    [[nodiscard]] bool isContinuous() const noexcept { return _stepElems == (size_t)cols; }
    [[nodiscard]] bool empty() const noexcept { return _data == nullptr || rows == 0 || cols == 0; }
    [[nodiscard]] depth_image clone() const
    {
        depth_image d(rows, cols);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c)
                d(r, c) = (*this)(r, c);
        return d;
    }

  private:
    float* _data = nullptr;
    size_t _stepElems = 0;
    std::shared_ptr<std::vector<float>> _own;
};
} // namespace rgbd_slam::features::primitives
#endif
#endif
