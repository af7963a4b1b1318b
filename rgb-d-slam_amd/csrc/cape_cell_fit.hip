// Stage A -- fused back-projection + per-cell PCA ("cell fit") for gfx950.
//
// Replaces, per frame: Depth_Map_Transformation::get_organized_cloud_array (reference
// src/features/primitives/depth_map_transformation.cpp:89-142), Plane_Segment::init_plane_segment
// (plane_segment.cpp:102-168) for every cell, and the tolerance loop of
// Primitive_Detection::init_planar_cell_fitting (primitive_detection.cpp:187-237).  The 3.7 MB organised
// cloud of the reference is never materialised: the row-major depth image is read once.
//
// Mapping: one 320-thread workgroup = two "bands" (a band = 20 image rows x 640 pixels = 32 cells).  Thread t
// owns float4 column q = t % 160 of band t / 160 and walks the band's 20 rows, so every wave-level load is a
// contiguous 16 B/lane row segment of the row-major image.  A float4 never straddles a cell (20 = 5 float4).
// Per-thread partial moment sums (f64) meet in LDS; 5 partials make one cell.  The sums are exact in f64 for
// any summation order when the addends' exponent span is < 21 bits; a per-cell z-range guard
// decides whether that holds, otherwise the cell is redone in the reference's pixel order.  The first wave then
// runs the per-cell continuity scan + plane fit for the 64 cells of the workgroup.
#include <hip/hip_runtime.h>

#include "cape_device.h"
#include "cape_internal.h"

namespace cape {

constexpr int kThreadsA = 320;
constexpr int kBandThreads = 160;
constexpr int kPartStride = 11; // 10 f64 per thread, padded against LDS bank conflicts

__device__ __forceinline__ void acc_px(float zr, double a, double b, double (&S)[9], uint32_t& n, float& zmin, float& zmax)
{
    const bool valid = zr > 0.0f; // depth_map_transformation.cpp:124 / plane_segment.cpp:134
    const float z = valid ? zr : 0.0f;
    n += valid ? 1u : 0u;
    zmax = fmaxf(zmax, z);
    zmin = valid ? fminf(zmin, z) : zmin;
    const double zd = (double)z;
    // ScreenCoordinate::to_camera_coordinates (point_coordinates.cpp:150-167): x = z * fl(fl(k00*u) + k02), then the
    // cloud stores static_cast<float> (depth_map_transformation.cpp:133-135)
    const float x = (float)(zd * a);
    const float y = (float)(zd * b);
    // plane_segment.cpp:142-150 : float values / float products widened into double accumulators
    S[0] += (double)x;
    S[1] += (double)y;
    S[2] += zd;
    S[3] += (double)(x * x);
    S[4] += (double)(y * y);
    S[5] += (double)(z * z);
    S[6] += (double)(x * y);
    S[7] += (double)(y * z);
    S[8] += (double)(x * z);
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

__global__ __launch_bounds__(kThreadsA) void cape_cell_fit_kernel(StageAParams p)
{
    __shared__ double s_part[kThreadsA * kPartStride];
    __shared__ float s_zmin[kThreadsA];
    __shared__ float s_zmax[kThreadsA];
    __shared__ double s_sums[64 * 10];

    const int t = threadIdx.x;
    const int frame = blockIdx.x / p.pairsPerFrame;
    const int pair = blockIdx.x - frame * p.pairsPerFrame;
    const size_t frameOff = (size_t)frame * p.W * p.H;

    // ------------------------------------------------------------------ streaming accumulation
    {
        const int bsel = t / kBandThreads;
        const int q = t - bsel * kBandThreads;
        const int band = pair * 2 + bsel;
        const int cellRow = band / p.segsPerRow;
        const int seg = band - cellRow * p.segsPerRow;
        const int col0 = seg * 640 + q * 4;
        const bool active = (band < p.bandsPerFrame) && (col0 < p.W);

        double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t n = 0;
        float zmin = __builtin_huge_valf(), zmax = 0.0f;
        if (active)
        {
            const double a0 = p.acol[col0], a1 = p.acol[col0 + 1], a2 = p.acol[col0 + 2], a3 = p.acol[col0 + 3];
            const float* base = p.depth + frameOff + (size_t)(cellRow * kCell) * p.W + col0;
            const double* brow = p.brow + cellRow * kCell;
#pragma unroll 1
            for (int r0 = 0; r0 < kCell; r0 += 5)
            {
                float4 v[5];
                double b[5];
#pragma unroll
                for (int i = 0; i < 5; ++i)
                {
                    v[i] = *reinterpret_cast<const float4*>(base + (size_t)(r0 + i) * p.W);
                    b[i] = brow[r0 + i];
                }
#pragma unroll
                for (int i = 0; i < 5; ++i)
                {
                    acc_px(v[i].x, a0, b[i], S, n, zmin, zmax);
                    acc_px(v[i].y, a1, b[i], S, n, zmin, zmax);
                    acc_px(v[i].z, a2, b[i], S, n, zmin, zmax);
                    acc_px(v[i].w, a3, b[i], S, n, zmin, zmax);
                }
            }
        }
        double* dst = s_part + t * kPartStride;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            dst[k] = S[k];
        dst[9] = (double)n;
        s_zmin[t] = zmin;
        s_zmax[t] = zmax;
    }
    __syncthreads();

    // ------------------------------------------------------------------ 5 partials -> one cell (exact, any order)
#pragma unroll
    for (int e = t; e < 640; e += kThreadsA)
    {
        const int cell = e / 10;
        const int m = e - cell * 10;
        const int bsel = cell >> 5;
        const int cseg = cell & 31;
        const double* src = s_part + (bsel * kBandThreads + cseg * 5) * kPartStride + m;
        double acc = src[0];
        acc += src[kPartStride];
        acc += src[2 * kPartStride];
        acc += src[3 * kPartStride];
        acc += src[4 * kPartStride];
        s_sums[cell * 10 + m] = acc;
    }
    __syncthreads();

    // ------------------------------------------------------------------ per-cell fit: wave 0, one lane per cell
    if (t >= 64)
        return;
    const int bsel = t >> 5;
    const int cseg = t & 31;
    const int band = pair * 2 + bsel;
    if (band >= p.bandsPerFrame)
        return;
    const int cellRow = band / p.segsPerRow;
    const int seg = band - cellRow * p.segsPerRow;
    const int cellCol = seg * 32 + cseg;
    if (cellCol >= p.hCells)
        return;
    const int cell = cellRow * p.hCells + cellCol;
    const size_t gcell = (size_t)frame * p.cells + cell;

    const float* cellBase = p.depth + frameOff + (size_t)(cellRow * kCell) * p.W + cellCol * kCell;

    // is_cell_horizontal_continuous (plane_segment.cpp:82-100): local row 10, idx 200..219
    bool continuous = true;
    {
        const float* rowp = cellBase + (size_t)(kCell / 2) * p.W;
        float zr[kCell];
#pragma unroll
        for (int i = 0; i < kCell; i += 4)
        {
            const float4 v = *reinterpret_cast<const float4*>(rowp + i);
            zr[i] = v.x; zr[i + 1] = v.y; zr[i + 2] = v.z; zr[i + 3] = v.w;
        }
        float last = std_maxf(zr[0], zr[1]);
        if (last <= 0)
            continuous = false;
#pragma unroll
        for (int i = 1; i < kCell; ++i)
            continuous = continuous && is_continuous(zr[i], last);
    }
    // is_cell_vertical_continuous (:62-80): local col 10, idx 10, 30, ..., 370 (loop stops before 390)
    {
        const float* colp = cellBase + kCell / 2;
        float zc[kCell - 1];
#pragma unroll
        for (int i = 0; i < kCell - 1; ++i)
            zc[i] = colp[(size_t)i * p.W];
        float last = std_maxf(zc[0], zc[1]);
        if (last <= 0)
            continuous = false;
#pragma unroll
        for (int i = 1; i < kCell - 1; ++i)
            continuous = continuous && is_continuous(zc[i], last);
    }

    double S[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
        S[k] = s_sums[t * 10 + k];
    uint32_t n = (uint32_t)s_sums[t * 10 + 9];

    // exactness guard: all addends of every sum within 2^20 of each other (see header)
    float zmin = __builtin_huge_valf(), zmax = 0.0f;
    {
        const int pbase = bsel * kBandThreads + cseg * 5;
#pragma unroll
        for (int j = 0; j < 5; ++j)
        {
            zmin = fminf(zmin, s_zmin[pbase + j]);
            zmax = fmaxf(zmax, s_zmax[pbase + j]);
        }
    }
    const float rab = fmaxf(p.ratio_col[cellCol], p.ratio_row[cellRow]);
    const bool exact_ok = (n == 0) || (zmax * rab <= 512.0f * zmin);
    uint32_t inorder = 0;
    if (!exact_ok && continuous && n >= (uint32_t)(kPts / 2))
    {
        // in-order path: the reference's pixel order (plane_segment.cpp:131-152)
        inorder = 1;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            S[k] = 0.0;
        uint32_t nn = 0;
        float zmn = 0, zmx = 0;
        for (int r = 0; r < kCell; ++r)
        {
            const double b = p.brow[cellRow * kCell + r];
            for (int c = 0; c < kCell; ++c)
                acc_px(cellBase[(size_t)r * p.W + c], p.acol[cellCol * kCell + c], b, S, nn, zmn, zmx);
        }
        n = nn;
    }

    PlaneFit f;
    f.planar = false;
    f.nx = f.ny = f.nz = f.d = 0.0;
    f.cx = f.cy = f.cz = 0.0;
    f.mse = kDblMax;
    f.score = 0.0;
    bool planar = false;
    if (!continuous || n < (uint32_t)(kPts / 2))
    {
        // plane_segment.cpp:114-123: returns right after clear_plane_parameters()
#pragma unroll
        for (int k = 0; k < 9; ++k)
            S[k] = 0.0;
        n = 0;
    }
    else if (n >= (uint32_t)p.minZeroPointCount)
    {
        fit_plane(S, n, f);
        const double qz = depth_quantization(f.cz);
        planar = f.mse <= qz * qz; // plane_segment.cpp:167
    }

    // _cellDistanceTols (primitive_detection.cpp:201-220)
    float tol = 0.0f;
    if (planar)
    {
        const float z0 = cellBase[0];
        const float z1 = cellBase[(size_t)(kCell - 1) * p.W + (kCell - 1)];
        const int u0 = cellCol * kCell, v0 = cellRow * kCell;
        float x0 = 0, y0 = 0, zz0 = 0, x1 = 0, y1 = 0, zz1 = 0;
        if (z0 > 0)
        {
            x0 = (float)((double)z0 * p.acol[u0]);
            y0 = (float)((double)z0 * p.brow[v0]);
            zz0 = z0;
        }
        if (z1 > 0)
        {
            x1 = (float)((double)z1 * p.acol[u0 + kCell - 1]);
            y1 = (float)((double)z1 * p.brow[v0 + kCell - 1]);
            zz1 = z1;
        }
        const float dx = x1 - x0, dy = y1 - y0, dz = zz1 - zz0;
        const float diam = sqrtf(dx * dx + (dy * dy + dz * dz));
        tol = std_minf(50.0f, diam * p.sinMerge * sqrtf((float)n));
    }

    double* os = p.cell_sums + gcell * kSumStride;
#pragma unroll
    for (int k = 0; k < 9; ++k)
        os[k] = S[k];
    os[9] = (double)n;
    double* op = p.cell_plane + gcell * kPlaneStride;
    op[0] = f.nx; op[1] = f.ny; op[2] = f.nz; op[3] = f.d;
    op[4] = f.cx; op[5] = f.cy; op[6] = f.cz; op[7] = f.mse;
    p.cell_score[gcell] = f.score;
    p.cell_tol[gcell] = tol;
    p.cell_flags[gcell] = (n & kCountMask) | (inorder ? kFlagInorder : 0u) | (planar ? kFlagPlanar : 0u);
}

void launch_cell_fit(const StageAParams& p, int nFrames, hipStream_t stream)
{
    const int grid = nFrames * p.pairsPerFrame;
    hipLaunchKernelGGL(cape_cell_fit_kernel, dim3(grid), dim3(kThreadsA), 0, stream, p);
}

} // namespace cape
