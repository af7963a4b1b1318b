// N3 ("next" row of SURVEY.md 8f): Depth_Map_Transformation::rectify_depth on the device
// (reference src/features/primitives/depth_map_transformation.cpp:23-87): every valid pixel of the depth camera is
// back-projected (float pre-factors, :45), moved to the colour camera's frame with the 4x4 camera2 -> camera1 matrix
// (:48-49), re-projected with the colour intrinsics (CameraCoordinate::to_screen_coordinates,
// point_coordinates.cpp:201-221) and scattered; where several source pixels land on one target the reference's
// MAKE_DETERMINISTIC loop keeps the LAST one in row-major order.
//
// cape_rectify_tile_kernel (the path of every ordinary camera rig): one workgroup OWNS a band of R target rows.  It scans the
// source rows that can land there -- the band's own rows moved by the row displacement the host predicted for this rig, plus a
// margin --, keeps the pixels that do and resolves their collisions in LDS with a 64-bit atomicMax on (source index + 1,
// registered depth) = "last writer in row-major order"; then the band's depths are stored -- no key buffer in HBM, no global
// atomic: 4 B read per scanned pixel, 4 B written per target.
// A source pixel whose target row belongs to a band that does not scan its row (the prediction samples a few depths: a scene
// outside them, or a rig whose displacement varies by more than a band can afford) flags its frame, and three small persistent kernels redo the flagged frames with global atomics, the keys living in the
// output image itself (cleared, scattered with atomicMax on the words, resolved in place).  The first version of this row
// ran that general form on everything, with 64-bit (index, depth) keys in a separate buffer: 2.9 ms per 1 024 frames, of
// which 1.9 ms were the 315 M global atomics.
#include <hip/hip_runtime.h>

#include "cape_internal.h"

namespace cape {

namespace {

#ifndef CAPE_RECTIFY_FLAT
#define CAPE_RECTIFY_FLAT 1
#endif
constexpr int kTileThreads = 1024;
constexpr int kFallbackThreads = 256;
constexpr int kFallbackChunk = kFallbackThreads * 4; // pixels per work item of the fallback kernels

// where the source pixel (row, col) of depth z lands, and its depth there.  False: dropped (invalid, behind the camera, outside).
__device__ __forceinline__ bool rectify_target(const RectifyParams& p, int row, int col, float originalZ, int& trow, int& tcol, float& zOut)
{
    if (!(originalZ > 0)) // `if (originalZ <= 0) continue;` -- NaN is dropped as well (the reference would exit(-1) on it)
        return false;
    // _Xpre / _Ypre hold static_cast<float>(K1^-1 [col,row,1]) (depth_map_transformation.cpp:156-161); float products
    const double o0 = (double)(p.xpre[col] * originalZ);
    const double o1 = (double)(p.ypre[row] * originalZ);
    const double o2 = (double)originalZ;
    // (T * original.homogeneous()).head<3>() = T[:, :3] * original + T[:, 3]
    const double p0 = ((p.T[0] * o0 + p.T[1] * o1) + p.T[2] * o2) + p.T[3];
    const double p1 = ((p.T[4] * o0 + p.T[5] * o1) + p.T[6] * o2) + p.T[7];
    const double p2 = ((p.T[8] * o0 + p.T[9] * o1) + p.T[10] * o2) + p.T[11];
    // 1.0 / z * (K1 * p).head<2>() with K1 = [fx 0 cx; 0 fy cy; 0 0 1].  The products with K1's zeros are left out: for finite p
    // they are +-0 and can only turn a -0 sum into +0, which matters to nothing below (a zero u or v never passes `> 0`); a
    // non-finite p0 / p1 (an absurd matrix) fails through its own coordinate either way.
    const double u = p.fx * p0 + p.cx * p2;
    const double v = p.fy * p1 + p.cy * p2;
    const double s = 1.0 / p2;
    const double sx = s * u, sy = s * v;
    if (isnan(sx) || isnan(sy))
        return false;
    const double fx_ = floor(sx), fy_ = floor(sy);
    // static_cast<uint>(floor(.)) then `> 0 and < width`: negative / huge values never pass
    if (!(fx_ > 0.0 && fy_ > 0.0 && fx_ < (double)p.W && fy_ < (double)p.H))
        return false;
    tcol = (int)fx_;
    trow = (int)fy_;
    zOut = (float)p2;
    return true;
}

// The same as straight-line code for the tile kernel, which evaluates four pixels side by side: every value is computed whatever
// the pixel is (an invalid depth or an absurd matrix gives garbage that the returned predicate rejects; nothing here can trap)
// and the tests are ANDed at the end -- one predicate per pixel instead of four nested exec-mask regions.
__device__ __forceinline__ bool rectify_target_flat(const RectifyParams& p, float xpre, float ypre, float originalZ, int& trow, int& tcol, float& zOut)
{
    const double o0 = (double)(xpre * originalZ);
    const double o1 = (double)(ypre * originalZ);
    const double o2 = (double)originalZ;
    const double p0 = ((p.T[0] * o0 + p.T[1] * o1) + p.T[2] * o2) + p.T[3];
    const double p1 = ((p.T[4] * o0 + p.T[5] * o1) + p.T[6] * o2) + p.T[7];
    const double p2 = ((p.T[8] * o0 + p.T[9] * o1) + p.T[10] * o2) + p.T[11];
    const double u = p.fx * p0 + p.cx * p2;
    const double v = p.fy * p1 + p.cy * p2;
    const double s = 1.0 / p2;
    const double sx = s * u, sy = s * v;
    const double fx_ = floor(sx), fy_ = floor(sy);
    // NaN fails every ordered comparison, so `isnan(sx) || isnan(sy)` is covered by the range test
    const bool ok = (originalZ > 0) & (fx_ > 0.0) & (fy_ > 0.0) & (fx_ < (double)p.W) & (fy_ < (double)p.H);
    tcol = (int)fx_;
    trow = (int)fy_;
    zOut = (float)p2;
    return ok;
}

// the registered depth of the target whose winner is source pixel `src` of the frame (0: nothing landed)
__device__ __forceinline__ float winner_depth(const RectifyParams& p, const float* in, unsigned key)
{
    if (!key)
        return 0.0f; // cv::Mat_<float>::zeros where nothing landed
    const int src = (int)(key - 1u);
    const int row = src / p.W, col = src - row * p.W;
    int trow, tcol;
    float z = 0.0f;
    (void)rectify_target(p, row, col, in[src], trow, tcol, z);
    return z;
}

} // namespace

__global__ __launch_bounds__(kTileThreads) void cape_rectify_tile_kernel(RectifyParams p, int bands, int log2R, int lo, int hi)
{
    const int R = 1 << log2R;
    // R x W keys: (source index relative to the first scanned row + 1) << 32 | bits of the registered depth -- the maximum is
    // the last writer in row-major order, and it carries its depth along
    extern __shared__ unsigned long long s_keys[];
    const int frame = blockIdx.x / bands, band = blockIdx.x - frame * bands;
    const int t0 = band * R, t1 = (t0 + R < p.H) ? t0 + R : p.H;
    // a pixel of source row r lands in row r + d, d in [lo, hi] for this rig (launch_rectify's prediction + a margin): the
    // band scans rows [t0 - hi, t1 - lo); the first and the last band reach the image's edge whatever the prediction says
    const int s0 = (band == 0 || t0 - hi < 0) ? 0 : t0 - hi, s1 = (band == bands - 1 || t1 - lo > p.H) ? p.H : t1 - lo;
    const size_t frameSize = (size_t)p.W * p.H;
    const float* in = p.in + (size_t)frame * frameSize;
    const int targets = (t1 - t0) * p.W;
    for (int i = threadIdx.x; i < targets; i += kTileThreads)
        s_keys[i] = 0ull;
    __syncthreads();
    const int quadsPerRow = p.W >> 2; // (the width is a multiple of the 20-pixel cell)
    const int quads = (s1 > s0 ? s1 - s0 : 0) * quadsPerRow;
    bool escaped = false;
    // thread t takes quads t, t + 1024, ... of the window; (row, column) advance without a division
    const int stepRows = kTileThreads / quadsPerRow, stepQuads = kTileThreads - stepRows * quadsPerRow;
    int r = s0 + (int)threadIdx.x / quadsPerRow, cq = (int)threadIdx.x - (r - s0) * quadsPerRow;
    for (int q = threadIdx.x; q < quads; q += kTileThreads)
    {
        const int c4 = cq << 2, rowBase = __mul24(r - s0, p.W);
        const float4 z4 = *reinterpret_cast<const float4*>(in + (size_t)r * p.W + c4);
        const float zs[4] = {z4.x, z4.y, z4.z, z4.w};
#if CAPE_RECTIFY_FLAT
        const float4 xp4 = *reinterpret_cast<const float4*>(p.xpre + c4);
        const float xps[4] = {xp4.x, xp4.y, xp4.z, xp4.w};
        const float yp = p.ypre[r];
        int trow[4], tcol[4];
        float z[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            ok[k] = rectify_target_flat(p, xps[k], yp, zs[k], trow[k], tcol[k], z[k]);
        bool outside[4]; // landed, not in my band, and moved by more than the prediction: rare
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            const bool mine = ok[k] & (trow[k] >= t0) & (trow[k] < t1);
            if (mine)
                atomicMax(&s_keys[__mul24(trow[k] - t0, p.W) + tcol[k]],
                          ((unsigned long long)((unsigned)(rowBase + c4 + k) + 1u) << 32) | (unsigned long long)__float_as_uint(z[k]));
            const int dr = trow[k] - r;
            outside[k] = ok[k] & !mine & ((dr < lo) | (dr > hi));
        }
        if (outside[0] | outside[1] | outside[2] | outside[3])
        {
            // does the band that owns the target scan this row?  (Inside the prediction it does by construction; every row is
            // scanned by somebody, so somebody asks.)
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const int ob = trow[k] >> log2R, o0 = ob << log2R, o1 = (o0 + R < p.H) ? o0 + R : p.H;
                const int os0 = (ob == 0 || o0 - hi < 0) ? 0 : o0 - hi, os1 = (ob == bands - 1 || o1 - lo > p.H) ? p.H : o1 - lo;
                escaped |= outside[k] & ((r < os0) | (r >= os1));
            }
        }
#else
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            int trow, tcol;
            float z;
            if (!rectify_target(p, r, c4 + k, zs[k], trow, tcol, z))
                continue;
            if (trow >= t0 && trow < t1)
                atomicMax(&s_keys[__mul24(trow - t0, p.W) + tcol],
                          ((unsigned long long)((unsigned)(rowBase + c4 + k) + 1u) << 32) | (unsigned long long)__float_as_uint(z));
            else if (trow - r < lo || trow - r > hi)
            {
                // not mine, and moved by more than predicted: does the band that owns the target scan this row?  (Inside the
                // prediction it does by construction; every row is scanned by somebody, so somebody asks.)
                const int ob = trow >> log2R, o0 = ob << log2R, o1 = (o0 + R < p.H) ? o0 + R : p.H;
                const int os0 = (ob == 0 || o0 - hi < 0) ? 0 : o0 - hi, os1 = (ob == bands - 1 || o1 - lo > p.H) ? p.H : o1 - lo;
                if (r < os0 || r >= os1)
                    escaped = true;
            }
        }
#endif
        r += stepRows;
        cq += stepQuads;
        if (cq >= quadsPerRow)
        {
            cq -= quadsPerRow;
            ++r;
        }
    }
    if (__syncthreads_or(escaped ? 1 : 0) && threadIdx.x == 0)
        if (atomicExch(&p.frameFlag[frame], 1u) == 0u)
            p.flagged[1 + atomicAdd(&p.flagged[0], 1u)] = (unsigned)frame;
    float* out = p.out + (size_t)frame * frameSize + (size_t)t0 * p.W;
    for (int i = threadIdx.x; i < (targets >> 2); i += kTileThreads)
    {
        const unsigned long long* k = s_keys + 4 * i;
        float4 o; // (a key of zero has zero depth bits: cv::Mat_<float>::zeros where nothing landed)
        o.x = __uint_as_float((unsigned)k[0]);
        o.y = __uint_as_float((unsigned)k[1]);
        o.z = __uint_as_float((unsigned)k[2]);
        o.w = __uint_as_float((unsigned)k[3]);
        *reinterpret_cast<float4*>(out + 4 * i) = o;
    }
}

// ---- the general form, for the frames the tile kernel flagged: keys in the output image itself
// MODE 0: clear the frame's output, 1: scatter (atomicMax of source index + 1 on the target's word), 2: resolve in place
template <int MODE> __global__ __launch_bounds__(kFallbackThreads) void cape_rectify_fallback_kernel(RectifyParams p)
{
    const unsigned nFlagged = p.flagged[0];
    const size_t frameSize = (size_t)p.W * p.H;
    const unsigned chunks = (unsigned)((frameSize + kFallbackChunk - 1) / kFallbackChunk);
    for (unsigned item = blockIdx.x; item < nFlagged * chunks; item += gridDim.x)
    {
        const unsigned frame = p.flagged[1 + item / chunks], chunk = item % chunks;
        const float* in = p.in + (size_t)frame * frameSize;
        unsigned* keys = reinterpret_cast<unsigned*>(p.out + (size_t)frame * frameSize);
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            const size_t g = (size_t)chunk * kFallbackChunk + (size_t)k * kFallbackThreads + threadIdx.x;
            if (g >= frameSize)
                continue;
            if (MODE == 0)
                keys[g] = 0u;
            else if (MODE == 1)
            {
                const int row = (int)(g / p.W), col = (int)(g - (size_t)row * p.W);
                int trow, tcol;
                float z;
                if (rectify_target(p, row, col, in[g], trow, tcol, z))
                    atomicMax(&keys[(size_t)trow * p.W + tcol], (unsigned)g + 1u);
            }
            else
                reinterpret_cast<float*>(keys)[g] = winner_depth(p, in, keys[g]);
        }
    }
}

// a rig whose row displacement varies by more than the bands can afford to scan: every frame goes to the general kernels
__global__ void cape_rectify_flag_all_kernel(RectifyParams p, int nFrames)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < nFrames)
        p.flagged[1 + f] = (unsigned)f;
    if (f == 0)
        p.flagged[0] = (unsigned)nFrames;
}

hipError_t launch_rectify(const RectifyParams& p, int nFrames, int computeUnits, hipStream_t stream)
{
    // flags of the frames + the list of flagged frames: [0] count, [1..] frames
    if (const hipError_t e = hipMemsetAsync(p.frameFlag, 0, (size_t)nFrames * sizeof(unsigned), stream); e != hipSuccess)
        return e;
    if (const hipError_t e = hipMemsetAsync(p.flagged, 0, sizeof(unsigned), stream); e != hipSuccess)
        return e;
    // band height: the keys of a band (8 B per target) fill half of a CU's LDS, one workgroup of sixteen waves per CU
    int log2R = 4;
    if (p.bandRows > 0)
        for (log2R = 0; (2 << log2R) <= p.bandRows; ++log2R) // (a power of two)
            ;
    // ... of THIS device: 96 KB of the 160 KB a gfx950 workgroup may have, never more than the device reports (ADVICE r3:
    // a 64 KB device got an 80 KB launch and failed instead of taking a smaller band)
    // (minus the few bytes __syncthreads_or keeps in static LDS: a band of exactly the device's limit -- CAPE_RECTIFY_BAND=32 at
    // W = 640 on gfx950 -- aborted the queue with HSA_STATUS_ERROR_INVALID_ALLOCATION instead of taking a smaller band)
    const size_t ldsLimit = (p.ldsLimitBytes > 0 ? (size_t)p.ldsLimitBytes : 64 * 1024) - 256;
    const size_t budget = ldsLimit < 96 * 1024 ? ldsLimit : 96 * 1024;
    while (log2R > 1 && ((size_t)p.W * 8 << log2R) > (p.bandRows <= 0 ? budget : ldsLimit))
        --log2R;
    const int R = 1 << log2R;
    const int bands = (p.H + R - 1) / R;
    const bool fits = ((size_t)R * p.W * 8) <= ldsLimit;
    if (!fits || p.shiftHi - p.shiftLo > 4 * R) // not even two rows of keys fit / each band would scan five times its own rows
        hipLaunchKernelGGL(cape_rectify_flag_all_kernel, dim3((unsigned)((nFrames + 255) / 256)), dim3(256), 0, stream, p, nFrames);
    else
        hipLaunchKernelGGL(cape_rectify_tile_kernel, dim3((unsigned)(bands * nFrames)), dim3(kTileThreads), (size_t)R * p.W * 8, stream, p, bands, log2R,
                           p.shiftLo, p.shiftHi);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    const int grid = (computeUnits > 0 ? computeUnits : 256) * 8;
    hipLaunchKernelGGL(cape_rectify_fallback_kernel<0>, dim3(grid), dim3(kFallbackThreads), 0, stream, p);
    hipLaunchKernelGGL(cape_rectify_fallback_kernel<1>, dim3(grid), dim3(kFallbackThreads), 0, stream, p);
    hipLaunchKernelGGL(cape_rectify_fallback_kernel<2>, dim3(grid), dim3(kFallbackThreads), 0, stream, p);
    return hipGetLastError();
}

} // namespace cape
