// Stage A, shared by the streaming kernel (cape_cell_moments.hip) and the per-cell / strip kernels (cape_cell_fit.hip):
// the pixel accumulators of plane_segment.cpp:131-152 on top of depth_map_transformation.cpp:123-138 and the continuity
// step of plane_segment.cpp:44-60.
#pragma once
#include <hip/hip_runtime.h>

#include "cape_device.h"
#include "cape_internal.h"

namespace cape {


// tuning knobs (overridable for A/B builds, see profiles/sweep.py).  Four measured-and-rejected variants of this kernel
// (LDS-DMA ring prefetch, ds_add_f64 partials, v_pk_mul_f32 pixel pairs, a three-buffer prefetch) lived here behind
// macros through round 2; their numbers are kept in DESIGN.md 4.1, their code in the history (commit 4d9050d).
#ifndef CAPE_A_GROUP
#define CAPE_A_GROUP 2   // image rows per prefetch group
#endif
#ifndef CAPE_A_WAVES
#define CAPE_A_WAVES 4   // __launch_bounds__ waves per SIMD of the streaming kernel (measured: 4 -> 1.46 ms, 5 -> 1.49 ms, 3 -> 2.6 ms)
#endif

constexpr int kThreadsA = 320;
constexpr int kBandThreads = 160;
[[maybe_unused]] constexpr int kPartStride = 11; // 10 f64 per thread, padded against LDS bank conflicts

struct PxAcc
{
    double S[9];
    uint32_t n;
    // z range of the valid pixels, as bit patterns (non-negative floats order like unsigned integers): zmaxBits is the
    // greatest pattern, zminBits1 the smallest (pattern - 1) -- the -1 wraps the +0 of an invalid pixel to 0xFFFFFFFF, the
    // neutral element of the minimum, with one v_sub_u32 (which issues at twice the rate of the selects it replaces)
    uint32_t zminBits1, zmaxBits;
};

// One pixel of plane_segment.cpp:131-152 on top of depth_map_transformation.cpp:123-138.
__device__ __forceinline__ float acc_px(float zr, double a, double b, PxAcc& A)
{
    const bool valid = zr > 0.0f;       // `if (z > 0)` ; NaN is invalid
    const float z = valid ? zr : 0.0f;  // invalid pixels add +0 to every sum
    A.n += valid ? 1u : 0u;
    const double zd = (double)z;
    // ScreenCoordinate::to_camera_coordinates (point_coordinates.cpp:150-167): x = z * fl(fl(k00*u) + k02) in f64, the
    // cloud stores static_cast<float>(x) (depth_map_transformation.cpp:133-135)
    const float x = (float)(zd * a);
    const float y = (float)(zd * b);
    // float values / float products widened into double accumulators (types.hpp:84: SQR on the operand's own type)
    A.S[0] += (double)x;
    A.S[1] += (double)y;
    A.S[2] += zd;
    A.S[3] += (double)(x * x);
    A.S[4] += (double)(y * y);
    A.S[5] += (double)(z * z);
    A.S[6] += (double)(x * y);
    A.S[7] += (double)(y * z);
    A.S[8] += (double)(x * z);
    return z; // the clamped depth (+0 if invalid): what the exactness guard looks at
}

// The same pixel in the streaming kernel, WITHOUT the validity test: a compare, a select and a carry add are three
// four-cycle instructions per pixel (profiles/r02_valu_rates.txt).  A pixel that is +0 adds +0 to every sum on its own, and
// a cell that holds anything else the reference would call invalid or that poisons a sum -- a negative depth, -0, NaN,
// +inf: every bit pattern above 0x7F7FFFFF -- is caught by the per-cell range guard (its largest pattern is tracked anyway)
// and redone by A2 with acc_px in the reference's order.  What is left to count is the non-zero pixels: v_min_u32 + v_add_u32.
// The count can only be too HIGH in a cell the guard rejects, never too low, so A2's "enough points for the in-order
// pass?" test errs on the side of redoing the cell.  Returns the bit pattern for the guard.
// (The six f32 products stay plain C: spelled as inline v_mul_f32_e32 they pin the schedule and the kernel spills -- 128 VGPRs +
// scratch.  What keeps the SLP vectoriser from pairing them into v_pk_mul_f32 + register-pair moves is -fno-slp-vectorize on
// cape_cell_moments.hip, see its header.)
__device__ __forceinline__ uint32_t acc_px_fast(float z, double a, double b, PxAcc& A)
{
    const uint32_t bits = __float_as_uint(z);
    A.n += min(bits, 1u);
    const double zd = (double)z;
    const float x = (float)(zd * a);
    const float y = (float)(zd * b);
    A.S[0] += (double)x;
    A.S[1] += (double)y;
    A.S[2] += zd;
    A.S[3] += (double)(x * x);
    A.S[4] += (double)(y * y);
    A.S[5] += (double)(z * z);
    A.S[6] += (double)(x * y);
    A.S[7] += (double)(y * z);
    A.S[8] += (double)(x * z);
    return bits;
}

__device__ __forceinline__ void acc_f4(const float4& v, double a0, double a1, double a2, double a3, double b, PxAcc& A)
{
    const uint32_t bx = acc_px_fast(v.x, a0, b, A);
    const uint32_t by = acc_px_fast(v.y, a1, b, A);
    const uint32_t bz = acc_px_fast(v.z, a2, b, A);
    const uint32_t bw = acc_px_fast(v.w, a3, b, A);
    // z range of the pixels (exactness guard) on the bit patterns: v_max3_u32 / v_min3_u32 + four v_sub_u32; a pattern with
    // the sign bit or above +inf's ends up in zmaxBits and fails the guard
    A.zmaxBits = max(max(A.zmaxBits, bx), max(by, max(bz, bw)));
    A.zminBits1 = min(min(A.zminBits1, bx - 1u), min(by - 1u, min(bz - 1u, bw - 1u)));
}

// plane_segment.cpp:44-60
__device__ __forceinline__ bool is_continuous(float pixelDepth, float& last)
{
    if (pixelDepth > 0)
    {
        if ((double)fabsf(pixelDepth - last) <= 4.0 * depth_quantization((double)pixelDepth))
        {
            last = pixelDepth;
            return true;
        }
        return false;
    }
    return true;
}

// The same step as straight-line code for a scan that runs every step on every lane: once a step has failed the scan's verdict
// is false whatever follows (the reference's `&&` stops calling the function there), so the later values of `last` are never
// observed and nothing needs to branch.
__device__ __forceinline__ bool is_continuous_flat(float pixelDepth, float& last)
{
    const bool positive = pixelDepth > 0; // NaN: not positive -> the step passes, like the reference's `if`
    const bool close = (double)fabsf(pixelDepth - last) <= 4.0 * depth_quantization((double)pixelDepth);
    last = (positive & close) ? pixelDepth : last;
    return !positive | close;
}

} // namespace cape
