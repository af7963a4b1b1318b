// N3 ("next" row of SURVEY.md 8f): Depth_Map_Transformation::rectify_depth on the device
// (reference src/features/primitives/depth_map_transformation.cpp:23-87): every valid pixel of the depth camera is
// back-projected (float pre-factors, :45), moved to the colour camera's frame with the 4x4 camera2 -> camera1 matrix
// (:48-49), re-projected with the colour intrinsics (CameraCoordinate::to_screen_coordinates,
// point_coordinates.cpp:201-221) and scattered; where several source pixels land on one target the reference's
// MAKE_DETERMINISTIC loop keeps the LAST one in row-major order.  Two streaming kernels: scatter packs (source
// index + 1, depth bits) into a 64-bit key and resolves collisions with atomicMax = "last writer in row-major order";
// resolve turns keys into the rectified float image and clears them for the next call.  HBM-bound: 4 B read + 8 B
// atomic per source pixel, 8 B read + 12 B written per target pixel.
#include <hip/hip_runtime.h>

#include "cape_internal.h"

namespace cape {

__global__ __launch_bounds__(256) void cape_rectify_scatter_kernel(RectifyParams p, size_t nPixels)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nPixels)
        return;
    const float originalZ = p.in[g];
    if (!(originalZ > 0)) // `if (originalZ <= 0) continue;` -- NaN is dropped as well (the reference would exit(-1) on it)
        return;
    const size_t frameSize = (size_t)p.W * p.H;
    const size_t frame = g / frameSize;
    const uint32_t src = (uint32_t)(g - frame * frameSize);
    const int row = (int)(src / p.W), col = (int)(src - (uint32_t)row * p.W);
    // _Xpre / _Ypre hold static_cast<float>(K1^-1 [col,row,1]) (depth_map_transformation.cpp:156-161); float products
    const double o0 = (double)(p.xpre[col] * originalZ);
    const double o1 = (double)(p.ypre[row] * originalZ);
    const double o2 = (double)originalZ;
    // (T * original.homogeneous()).head<3>() = T[:, :3] * original + T[:, 3]
    const double p0 = ((p.T[0] * o0 + p.T[1] * o1) + p.T[2] * o2) + p.T[3];
    const double p1 = ((p.T[4] * o0 + p.T[5] * o1) + p.T[6] * o2) + p.T[7];
    const double p2 = ((p.T[8] * o0 + p.T[9] * o1) + p.T[10] * o2) + p.T[11];
    // 1.0 / z * (K1 * p).head<2>()
    const double u = (p.fx * p0 + 0.0 * p1) + p.cx * p2;
    const double v = (0.0 * p0 + p.fy * p1) + p.cy * p2;
    const double s = 1.0 / p2;
    const double sx = s * u, sy = s * v;
    if (isnan(sx) || isnan(sy))
        return;
    const double fx_ = floor(sx), fy_ = floor(sy);
    // static_cast<uint>(floor(.)) then `> 0 and < width`: negative / huge values never pass
    if (!(fx_ > 0.0 && fy_ > 0.0 && fx_ < (double)p.W && fy_ < (double)p.H))
        return;
    const uint32_t tcol = (uint32_t)fx_, trow = (uint32_t)fy_;
    const float zOut = (float)p2;
    const unsigned long long key = ((unsigned long long)(src + 1u) << 32) | (unsigned long long)__float_as_uint(zOut);
    atomicMax(&p.keys[frame * frameSize + (size_t)trow * p.W + tcol], key);
}

__global__ __launch_bounds__(256) void cape_rectify_resolve_kernel(RectifyParams p, size_t nPixels)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nPixels)
        return;
    const unsigned long long k = p.keys[g];
    p.out[g] = k ? __uint_as_float((uint32_t)(k & 0xFFFFFFFFull)) : 0.0f; // cv::Mat_<float>::zeros where nothing landed
    p.keys[g] = 0ull;
}

hipError_t launch_rectify(const RectifyParams& p, int nFrames, hipStream_t stream)
{
    const size_t n = (size_t)nFrames * p.W * p.H;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(cape_rectify_scatter_kernel, dim3(blocks), dim3(256), 0, stream, p, n);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(cape_rectify_resolve_kernel, dim3(blocks), dim3(256), 0, stream, p, n);
    return hipGetLastError();
}

} // namespace cape
