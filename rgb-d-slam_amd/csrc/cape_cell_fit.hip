// Stage A -- fused back-projection + per-cell PCA for gfx950, as two kernels:
//
//   A1 cape_cell_moments_kernel : the HBM/FP64-issue bound streaming pass.  Reads the row-major depth image once,
//      back-projects on the fly and leaves, per cell, the nine moment sums + count, the result of the continuity
//      cross scan and the exactness verdict.  Replaces Depth_Map_Transformation::get_organized_cloud_array
//      (reference src/features/primitives/depth_map_transformation.cpp:89-142) and the scan + accumulation part of
//      Plane_Segment::init_plane_segment (plane_segment.cpp:44-152).  The 3.7 MB organised cloud is never built.
//   A2 cape_cell_plane_kernel   : one lane per cell, every lane busy: validity gates, Plane_Segment::fit_plane
//      (plane_segment.cpp:155-168, 205-284), the merge tolerance of init_planar_cell_fitting
//      (primitive_detection.cpp:201-220) and the histogram bin of init_histogram (primitive_detection.cpp:253-254 +
//      histogram.hpp:48-54).  Splitting it off keeps the long dependent f64 chains (eigen-solver, div, sqrt) out of
//      the streaming workgroups, whose LDS and wave slots are then released as soon as the band is summed.
//
// A1 mapping: one 320-thread workgroup = two "bands" (band = 20 image rows x 640 pixels = 32 cells).  Thread t owns
// float4 column q = t % 160 of band t / 160 and walks the band's 20 rows, so every wave-level load is a contiguous
// 16 B/lane segment of the row-major image; a float4 never straddles a cell (20 = 5 float4).  Rows are loaded two
// at a time (CAPE_A_GROUP), ping-pong buffered, one group ahead of the arithmetic (non-temporal: the image is read once).  Per-thread partial sums (f64) meet in LDS; 5 partials = one cell.
// The sums are exact in f64 for ANY summation order when the addends' exponent span is < 21 bits;
// a per-cell z-range guard decides whether that holds, otherwise A2 redoes the cell in the reference's pixel order.
// (A1 itself: cape_cell_moments.hip, compiled without the SLP vectoriser; the shared accumulators: cape_cell_acc.h.)
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cape_cell_acc.h"

namespace cape {

// ---------------------------------------------------------------------------------------------------------------
// A2: one lane per cell
// ---------------------------------------------------------------------------------------------------------------
#ifndef CAPE_A2_WAVES
#define CAPE_A2_WAVES 4
#endif
#ifndef CAPE_A2_ABLATE
#define CAPE_A2_ABLATE 0
#endif
// What stage A2 decides for ONE cell from its moment sums and stage A1's verdicts: the in-order redo of a cell whose sums are
// not provably exact, the validity gates, Plane_Segment::fit_plane, the merge tolerance and the histogram bin.  Shared by the
// throughput kernel below (one lane per cell of a tile) and the latency instance (cape_cell_strip_kernel).
struct CellFit
{
    PlaneFit f;
    bool planar;
    float tol;
    int bin;
    uint32_t nearEdge, inorder, n;
    bool rewrite; // S / n differ from what stage A1 summed (cleared, or redone in order): the caller stores them
};
__device__ __forceinline__ void cell_fit_and_bin(const StageAParams& p, int frame, int cellRow, int cellCol, const CellAux& aux, double (&S)[9],
                                                 CellFit& o)
{
    PlaneFit f;
    f.planar = false;
    f.nx = f.ny = f.nz = f.d = 0.0;
    f.cx = f.cy = f.cz = 0.0;
    f.mse = kDblMax;
    f.score = 0.0;
    bool planar = false;
    float tol = 0.0f;
    int bin = -1;
    uint32_t nearEdge = 0, inorder = 0;
    uint32_t n = aux.flags & kCountMask;
    const bool continuous = (aux.flags & kAuxContinuous) != 0;
    const bool exact_ok = (aux.flags & kAuxExact) != 0;
    bool rewrite = false;
    if (!exact_ok && continuous && n >= (uint32_t)(kPts / 2))
    {
        // in-order path: the reference's pixel order (plane_segment.cpp:131-152)
        inorder = 1;
        rewrite = true;
        const size_t cellOff = (size_t)frame * p.W * p.H + (size_t)(cellRow * kCell) * p.W + cellCol * kCell;
        PxAcc A;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            A.S[k] = 0.0;
        A.n = 0;
        A.zminBits1 = 0;
        A.zmaxBits = 0;
        for (int r = 0; r < kCell; ++r)
        {
            const double b = p.brow[cellRow * kCell + r];
            for (int c = 0; c < kCell; ++c)
            {
                const size_t o = cellOff + (size_t)r * p.W + c;
                const float zr = p.depth ? p.depth[o] : (float)p.depth_u16[o] * p.u16_scale;
                acc_px(zr, p.acol[cellCol * kCell + c], b, A);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k)
            S[k] = A.S[k];
        n = A.n;
    }

    if (!continuous || n < (uint32_t)(kPts / 2))
    {
        // plane_segment.cpp:114-123: returns right after clear_plane_parameters()
#pragma unroll
        for (int k = 0; k < 9; ++k)
            S[k] = 0.0;
        n = 0;
        rewrite = true;
    }
    // (CAPE_A2_ABLATE: timing experiments only -- profiles/a2_ablation.sh builds one library per bit and reports what
    //  each part of the kernel costs; results of such a build are wrong by construction.  1 = no plane fit,
    //  2 = no histogram bin (acos / atan2), 4 = no edge predicates, 8 = no tolerance)
    else if (n >= (uint32_t)p.minZeroPointCount && !(CAPE_A2_ABLATE & 1))
    {
        fit_plane(S, n, f);
        const double qz = depth_quantization(f.cz);
        planar = f.mse <= qz * qz; // plane_segment.cpp:167
    }

    // _cellDistanceTols (primitive_detection.cpp:201-220): first and last cloud rows of the cell (zeros if invalid)
    if ((CAPE_A2_ABLATE & 1) && n >= (uint32_t)p.minZeroPointCount)
    {
        planar = S[2] > 0.0; // keep the downstream work alive without the fit
        f.nx = S[0] * 1e-9, f.ny = S[1] * 1e-9, f.nz = -0.5, f.d = S[2] * 1e-6, f.cx = S[0] / n, f.cy = S[1] / n, f.cz = S[2] / n;
    }
    if (planar)
    {
        const float z0 = aux.z0, z1 = aux.z399;
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
        const float diam = (CAPE_A2_ABLATE & 8) ? dx : sqrtf(dx * dx + (dy * dy + dz * dz));
        tol = (CAPE_A2_ABLATE & 8) ? diam : std_minf(50.0f, diam * p.sinMerge * sqrtf((float)n));

        // init_histogram (primitive_detection.cpp:253-254) + Histogram::init_histogram (histogram.hpp:48-54)
        const double theta = (CAPE_A2_ABLATE & 2) ? -f.nz : acos(-f.nz);
        const double phi = (CAPE_A2_ABLATE & 2) ? f.nx : atan2(f.nx, f.ny);
        constexpr double kPi = 3.14159265358979323846;
        const double tx = 19.0 * (theta - 0.0) / kPi;
        const int xQ = (int)floor(tx);
        int yQ = 0;
        double ty = 0.5;
        if (xQ > 0)
        {
            ty = 19.0 * (phi - (-kPi)) / (kPi - (-kPi));
            yQ = (int)floor(ty);
        }
        bin = yQ * 20 + xQ;
        // libm tie guard: ocml vs glibc acos/atan2 may differ in the last ulp
        if (fabs(tx - rint(tx)) < 1e-9 || (xQ > 0 && fabs(ty - rint(ty)) < 1e-9))
            nearEdge = 1;
    }
    o.f = f;
    o.planar = planar;
    o.tol = tol;
    o.bin = bin;
    o.nearEdge = nearEdge;
    o.inorder = inorder;
    o.n = n;
    o.rewrite = rewrite;
}

// A workgroup owns a TILE of whole cell rows of one frame (THREADS cells at most, one lane per cell).  After the fit every
// cell publishes its plane (normal, d, centroid, merge tolerance) in LDS, and each lane evaluates region_growing's merge
// predicate for the directed edges to its left and upper neighbours -- the planes are still in registers here, whereas
// the grow kernel had to read all of them back (64 B per cell, a dozen exposed memory round trips per frame) to do the
// same.  Tiles are independent workgroups: the edges between the first row of a tile and the row above it (another
// workgroup's) are left to the grow kernel, which evaluates just those rows (2 of 23 row boundaries at 640x480).
// THREADS = 256 is the throughput instance (as many resident waves as before); THREADS = 1024 takes a 640x480 frame in a
// single tile and is launched for small batches, where the latency of one frame is what matters.
struct CellPub
{
    double nx, ny, nz, d, cx, cy, cz, tol;
};

template <int THREADS>
__global__ __launch_bounds__(THREADS, THREADS == 256 ? CAPE_A2_WAVES : 4) void cape_cell_plane_kernel(StageAParams p, int nFrames)
{
    __shared__ CellPub s_pub[THREADS];
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        if (p.clear0)
            *p.clear0 = 0u;
        if (p.clear1)
            *p.clear1 = 0u;
        if (p.clear2)
        {
            *p.clear2 = 0u;
            if (p.clear2Buckets)
                for (int c = 1; c <= kResumeClasses; ++c)
                    p.clear2[(size_t)c * p.clear2Buckets] = 0u;
        }
        if (p.clear3)
            *p.clear3 = 0u;
        if (p.clear4)
            p.clear4[0] = p.clear4[1] = 0u; // records handed out of the spill pool, frames through the general instance
    }
    const int HC = p.hCells, VC = p.vCells;
    const int rowsPerTile = THREADS / HC > 0 ? THREADS / HC : 1;
    const int tilesPerFrame = (VC + rowsPerTile - 1) / rowsPerTile;
    const int frame = blockIdx.x / tilesPerFrame;
    const int r0 = (blockIdx.x - frame * tilesPerFrame) * rowsPerTile;
    const int tid = threadIdx.x;
    const int tileCells = rowsPerTile * HC;
    const int lrow = tid / HC, cellCol = tid - lrow * HC;
    {
        const int cellRow = r0 + lrow;
        const bool valid = tid < tileCells && cellRow < VC;
        const int cell = cellRow * HC + cellCol;
        const size_t gcell = (size_t)frame * p.cells + (valid ? cell : 0);

        CellFit cf;
        cf.f.planar = false;
        cf.f.nx = cf.f.ny = cf.f.nz = cf.f.d = 0.0;
        cf.f.cx = cf.f.cy = cf.f.cz = 0.0;
        cf.f.mse = kDblMax;
        cf.f.score = 0.0;
        cf.planar = false;
        cf.tol = 0.0f;
        cf.bin = -1;
        cf.nearEdge = cf.inorder = cf.n = 0;
        cf.rewrite = false;
        if (valid)
        {
            const CellAux aux = p.cell_aux[gcell];
            double* os = p.cell_sums + gcell * kSumStride;
            double S[9];
#pragma unroll
            for (int k = 0; k < 9; ++k)
                S[k] = os[k];
            cell_fit_and_bin(p, frame, cellRow, cellCol, aux, S, cf);
            if (cf.rewrite)
            {
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    os[k] = S[k];
                os[9] = (double)cf.n;
            }
        }
        const PlaneFit& f = cf.f;
        const bool planar = cf.planar;
        const float tol = cf.tol;
        const int bin = cf.bin;
        const uint32_t nearEdge = cf.nearEdge, inorder = cf.inorder, n = cf.n;

        // ---- publish, then the merge predicate of the directed edges to the left and upper neighbours
        CellPub me;
        me.nx = f.nx; me.ny = f.ny; me.nz = f.nz; me.d = f.d;
        me.cx = f.cx; me.cy = f.cy; me.cz = f.cz; me.tol = (double)tol;
        s_pub[tid] = me;
        __syncthreads();
        uint32_t edges = 0;
        if (valid)
        {
            if (cellCol > 0 && !(CAPE_A2_ABLATE & 4))
            {
                const CellPub L = s_pub[tid - 1];
                if (can_be_merged(L.nx, L.ny, L.nz, L.d, me.nx, me.ny, me.nz, me.cx, me.cy, me.cz, me.tol, p.cosMergeA))
                    edges |= kFlagLeftToMe;
                if (can_be_merged(me.nx, me.ny, me.nz, me.d, L.nx, L.ny, L.nz, L.cx, L.cy, L.cz, L.tol, p.cosMergeA))
                    edges |= kFlagMeToLeft;
            }
            if (lrow > 0 && !(CAPE_A2_ABLATE & 4)) // the tile's first row: its upper neighbours belong to another workgroup (see above)
            {
                const CellPub Up = s_pub[tid - HC];
                if (can_be_merged(Up.nx, Up.ny, Up.nz, Up.d, me.nx, me.ny, me.nz, me.cx, me.cy, me.cz, me.tol, p.cosMergeA))
                    edges |= kFlagUpToMe;
                if (can_be_merged(me.nx, me.ny, me.nz, me.d, Up.nx, Up.ny, Up.nz, Up.cx, Up.cy, Up.cz, Up.tol, p.cosMergeA))
                    edges |= kFlagMeToUp;
            }
            double* op = p.cell_plane + gcell * kPlaneStride;
            op[0] = f.nx; op[1] = f.ny; op[2] = f.nz; op[3] = f.d;
            op[4] = f.cx; op[5] = f.cy; op[6] = f.cz; op[7] = f.mse;
            p.cell_mse[gcell] = f.mse;
            p.cell_score[gcell] = f.score;
            p.cell_tol[gcell] = tol;
            p.cell_bins[gcell] = bin;
            p.cell_flags[gcell] = (n & kCountMask) | edges | (nearEdge ? kFlagNearEdge : 0u) | (inorder ? kFlagInorder : 0u) |
                                  (planar ? kFlagPlanar : 0u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Latency instance of stage A (round 4): the reference calls the path with ONE frame (src/rgbd_slam.cpp:291-297).  With a
// dozen workgroups per frame the throughput kernels above leave the device idle and walk a band through ten dependent
// memory round trips (the input of a one-frame handle is read straight from pinned host memory, cape_host_alloc).  Here a
// frame is cut into STRIPS of 8 cells (160 x 20 pixels): 96 workgroups for 640x480, every load of a strip requested at once
// -- one round trip, then the link's bandwidth --, and the strip's workgroup carries on with what stage A2 does for its
// cells: cross scans, exactness verdict, cell_fit_and_bin.  The merge predicates of the directed cell edges need the planes
// of neighbouring strips: the workgroup that finishes a frame LAST (a counter per frame) evaluates all of them, so the
// whole of stage A is one launch and the grow kernel finds no tile-boundary rows to do (a2RowsPerTile = vCells).
// Same arithmetic per cell as A1 + A2: the partial sums meet in another order, which is exact under the guard (header), and a
// cell that fails the guard is redone in pixel order like everywhere else.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kStripCells = 8;
constexpr int kStripThreads = 256;
constexpr int kStripCols = kStripCells * 5;              // float4 columns of a strip row
constexpr int kStripRowGroups = 5;                       // x kStripRows rows = the cell height
constexpr int kStripRows = kCell / kStripRowGroups;      // rows per thread
constexpr int kStripActive = kStripCols * kStripRowGroups; // threads that stream pixels (200)
static_assert(kStripRows * kStripRowGroups == kCell && kStripActive <= kStripThreads, "strip geometry");

__device__ __forceinline__ float4 strip_to_f4(const float4& v, float) { return v; }
__device__ __forceinline__ float4 strip_to_f4(const ushort4& rw, float scale)
{
    // like the U16 variant of the streaming kernel: cv::Mat::convertTo(CV_32F, scale) = float(raw) * float(scale), one v_mul_f32 each
    auto mul1 = [](float a, float b) {
        float r;
        asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    };
    return make_float4(mul1((float)rw.x, scale), mul1((float)rw.y, scale), mul1((float)rw.z, scale), mul1((float)rw.w, scale));
}

template <bool U16> __global__ __launch_bounds__(kStripThreads) void cape_cell_strip_kernel(StageAParams p, uint32_t* frameCounters)
{
    __shared__ double s_part[kStripActive * kPartStride]; // [thread][9 sums, count, (zmin, zmax)]
    __shared__ float s_row[kStripCells * kCell];          // local row 10 of every cell
    __shared__ float s_col[kStripCells * kCell];          // local column 10 of every cell
    __shared__ float s_corner[kStripCells * 3];           // first, last and centre pixel of every cell
    __shared__ double s_sum[kStripCells * 10];            // the cells' sums + count
    __shared__ uint32_t s_range[kStripCells * 2];         // zminBits1, zmaxBits
    __shared__ uint32_t s_cont[2 * kStripCells];          // verdicts of the horizontal / vertical scans
    __shared__ uint32_t s_last;

    const int t = threadIdx.x;
    if (blockIdx.x == 0 && t == 0)
    {
        if (p.clear0)
            *p.clear0 = 0u;
        if (p.clear1)
            *p.clear1 = 0u;
        if (p.clear2)
        {
            *p.clear2 = 0u;
            if (p.clear2Buckets)
                for (int c = 1; c <= kResumeClasses; ++c)
                    p.clear2[(size_t)c * p.clear2Buckets] = 0u;
        }
        if (p.clear3)
            *p.clear3 = 0u;
        if (p.clear4)
            p.clear4[0] = p.clear4[1] = 0u; // records handed out of the spill pool, frames through the general instance
    }
    const int HC = p.hCells, VC = p.vCells;
    const int stripsPerRow = (HC + kStripCells - 1) / kStripCells;
    const int stripsPerFrame = stripsPerRow * VC;
    const int frame = blockIdx.x / stripsPerFrame;
    const int sidx = blockIdx.x - frame * stripsPerFrame;
    const int cellRow = sidx / stripsPerRow;
    const int strip = sidx - cellRow * stripsPerRow;
    const size_t frameOff = (size_t)frame * p.W * p.H;

    const int rg = t / kStripCols;          // row group 0..4 (5, 6: idle threads)
    const int q = t - rg * kStripCols;      // float4 column of the strip
    const int col0 = strip * (kStripCells * kCell) + q * 4;
    const bool active = t < kStripActive && col0 < p.W;
    const int cseg = q / 5, j = q - cseg * 5;

    // ------------------------------------------------------------------ streaming accumulation: four rows per thread
    if (t < kStripActive)
    {
        PxAcc A;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            A.S[k] = 0.0;
        A.n = 0;
        A.zminBits1 = 0xFFFFFFFFu;
        A.zmaxBits = 0u;
        if (active)
        {
            using Raw = typename std::conditional<U16, ushort4, float4>::type;
            const size_t pix = frameOff + (size_t)(cellRow * kCell + rg * kStripRows) * p.W + col0;
            Raw raw[kStripRows];
#pragma unroll
            for (int i = 0; i < kStripRows; ++i)
            {
                if constexpr (U16)
                    raw[i] = *reinterpret_cast<const ushort4*>(p.depth_u16 + pix + (size_t)i * p.W);
                else
                    raw[i] = *reinterpret_cast<const float4*>(p.depth + pix + (size_t)i * p.W);
            }
            const double a0 = p.acol[col0], a1 = p.acol[col0 + 1], a2 = p.acol[col0 + 2], a3 = p.acol[col0 + 3];
            const double* brow = p.brow + cellRow * kCell + rg * kStripRows;
#pragma unroll
            for (int i = 0; i < kStripRows; ++i)
            {
                const float4 v = strip_to_f4(raw[i], p.u16_scale);
                const int r = rg * kStripRows + i;
                acc_f4(v, a0, a1, a2, a3, brow[i], A);
                if (r == kCell / 2)
                    *reinterpret_cast<float4*>(&s_row[cseg * kCell + 4 * j]) = v;
                if (j == 2)
                    s_col[cseg * kCell + r] = v.z;
                if (r == 0 && j == 0)
                    s_corner[cseg * 3] = v.x;
                if (r == kCell - 1 && j == 4)
                    s_corner[cseg * 3 + 1] = v.w;
                if (r == kCell / 2 && j == 2)
                    s_corner[cseg * 3 + 2] = v.z;
            }
        }
        double* dst = s_part + t * kPartStride;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            dst[k] = A.S[k];
        dst[9] = (double)A.n;
        reinterpret_cast<uint2*>(dst + 10)[0] = make_uint2(A.zminBits1, A.zmaxBits);
    }
    __syncthreads();

    // ------------------------------------------------------------------ 25 partials -> one cell (exact, any order) on waves 0 and 1,
    //      the cells' cross scans on wave 3 meanwhile (lane c: horizontal, lane c + 8: vertical)
    if (t < kStripCells * 11)
    {
        const int cell = t / 11, m = t - cell * 11;
        if (m < 10)
        {
            double acc = 0.0;
#pragma unroll
            for (int g = 0; g < kStripRowGroups; ++g)
#pragma unroll
                for (int jj = 0; jj < 5; ++jj)
                    acc += s_part[(g * kStripCols + cell * 5 + jj) * kPartStride + m];
            s_sum[cell * 10 + m] = acc;
        }
        else
        {
            uint32_t zminBits1 = 0xFFFFFFFFu, zmaxBits = 0u;
#pragma unroll
            for (int g = 0; g < kStripRowGroups; ++g)
#pragma unroll
                for (int jj = 0; jj < 5; ++jj)
                {
                    const uint2 zr = reinterpret_cast<const uint2*>(s_part + (g * kStripCols + cell * 5 + jj) * kPartStride + 10)[0];
                    zminBits1 = min(zminBits1, zr.x);
                    zmaxBits = max(zmaxBits, zr.y);
                }
            s_range[cell * 2] = zminBits1;
            s_range[cell * 2 + 1] = zmaxBits;
        }
    }
    else if (t >= 192 && t < 192 + 4 * kStripCells)
    {
        // four lane-tasks per cell like in the streaming kernel (cape_cell_moments.hip): horizontal / vertical x first / second half of the
        // scan's steps, straight-line steps, the second half starting from the last positive depth of the first; the scans are the longest
        // chain between the two barriers (19 branching steps on 16 lanes through round 4)
        const int l = t - 192, c = l & (kStripCells - 1);
        const bool second = (l & kStripCells) != 0, vscan = (l & (2 * kStripCells)) != 0;
        // is_cell_horizontal_continuous (plane_segment.cpp:82-100): local row 10, idx 200..219, steps 1..19
        // is_cell_vertical_continuous (:62-80): local column 10, idx 10, 30, ..., 370: steps 1..18 (the loop stops before 390)
        const float* zs = (vscan ? s_col : s_row) + c * kCell;
        const float seed = std_maxf(zs[0], zs[1]);
        float last = seed;
#pragma unroll
        for (int i = 1; i <= 9; ++i)
        {
            const float z = zs[i];
            last = (second && z > 0) ? z : last;
        }
        bool continuous = second || !(seed <= 0);
        const float* zt = zs + (second ? 10 : 1);
        const int steps = second ? (vscan ? 9 : 10) : 9;
#pragma unroll
        for (int k = 0; k < 10; ++k)
        {
            const bool pass = is_continuous_flat(zt[k], last);
            continuous &= pass | (k >= steps);
        }
        continuous &= __shfl_xor((int)continuous, kStripCells) != 0;
        continuous &= __shfl_xor((int)continuous, 2 * kStripCells) != 0;
        if (l < 2 * kStripCells)
            s_cont[l] = continuous ? 1u : 0u;
    }
    __syncthreads();

    // ------------------------------------------------------------------ wave 0: one lane per cell
    if (t < 64)
    {
        const int c = t & (kStripCells - 1);
        const int cellCol = strip * kStripCells + c;
        const bool continuous = s_cont[c] != 0u && s_cont[c + kStripCells] != 0u;
        if (t < kStripCells && cellCol < HC)
        {
            const uint32_t n = (uint32_t)s_sum[c * 10 + 9];
            const uint32_t zminBits1 = s_range[c * 2], zmaxBits = s_range[c * 2 + 1];
            // back to depths: with n > 0 at least one pixel was valid, so the minimum is a real pattern - 1
            const float zmin = __uint_as_float(zminBits1 + 1u), zmax = __uint_as_float(zmaxBits);
            const float rab = fmaxf(p.ratio_col[cellCol], p.ratio_row[cellRow]);
            // all pixels +0 or positive and finite (acc_px_fast's precondition), and their range within the exactness bound
            const bool exact_ok = zmaxBits <= 0x7F7FFFFFu && ((n == 0) || (zmax * rab <= 512.0f * zmin));
            const size_t gcell = (size_t)frame * p.cells + cellRow * HC + cellCol;
            CellAux aux;
            aux.z0 = s_corner[c * 3];
            aux.z399 = s_corner[c * 3 + 1];
            aux.flags = (n & kCountMask) | (continuous ? kAuxContinuous : 0u) | (exact_ok ? kAuxExact : 0u);
            aux.zc = s_corner[c * 3 + 2];
            p.cell_aux[gcell] = aux;

            double S[9];
#pragma unroll
            for (int k = 0; k < 9; ++k)
                S[k] = s_sum[c * 10 + k];
            CellFit cf;
            cell_fit_and_bin(p, frame, cellRow, cellCol, aux, S, cf);
            double* os = p.cell_sums + gcell * kSumStride;
#pragma unroll
            for (int k = 0; k < 9; ++k)
                os[k] = S[k];
            os[9] = (double)cf.n;
            const PlaneFit& f = cf.f;
            double* op = p.cell_plane + gcell * kPlaneStride;
            op[0] = f.nx; op[1] = f.ny; op[2] = f.nz; op[3] = f.d;
            op[4] = f.cx; op[5] = f.cy; op[6] = f.cz; op[7] = f.mse;
            p.cell_mse[gcell] = f.mse;
            p.cell_score[gcell] = f.score;
            p.cell_tol[gcell] = cf.tol;
            p.cell_bins[gcell] = cf.bin;
            // (the edge bits are added by the frame's last workgroup, below)
            p.cell_flags[gcell] = (cf.n & kCountMask) | (cf.nearEdge ? kFlagNearEdge : 0u) | (cf.inorder ? kFlagInorder : 0u) |
                                  (cf.planar ? kFlagPlanar : 0u);
        }
        // this strip's cells are published: release, then count the strip
        __threadfence();
        if (t == 0)
        {
            const uint32_t old = atomicAdd(&frameCounters[frame], 1u);
            s_last = (old == (uint32_t)stripsPerFrame - 1u) ? 1u : 0u;
            if (s_last)
                atomicExch(&frameCounters[frame], 0u); // ready for the next call
        }
    }
    __syncthreads();
    if (!s_last)
        return;
    __threadfence(); // acquire: every other strip's planes

    // ------------------------------------------------------------------ last workgroup of the frame: region_growing's merge predicate
    //      (primitive_detection.cpp:802 with plane_segment.cpp:322-326) of every directed cell edge, as stage A2 evaluates it
    const size_t cellBase = (size_t)frame * p.cells;
    auto load_pub = [&](size_t gc) {
        const double2* pl = reinterpret_cast<const double2*>(p.cell_plane + gc * kPlaneStride);
        const double2 v0 = pl[0], v1 = pl[1], v2 = pl[2], v3 = pl[3];
        CellPub o;
        o.nx = v0.x; o.ny = v0.y; o.nz = v1.x; o.d = v1.y;
        o.cx = v2.x; o.cy = v2.y; o.cz = v3.x;
        o.tol = (double)p.cell_tol[gc];
        return o;
    };
    constexpr int kTrip = 3; // cells per thread and trip (a 640x480 frame is one trip): all their planes are requested together
    for (int c0 = 0; c0 < p.cells; c0 += kTrip * kStripThreads)
    {
        CellPub me[kTrip], lf[kTrip], up[kTrip];
#pragma unroll
        for (int u = 0; u < kTrip; ++u)
        {
            const int cell = c0 + u * kStripThreads + t;
            const int cl = cell < p.cells ? cell : p.cells - 1; // clamped: unconditional loads
            const int r = cl / HC, c = cl - r * HC;
            const size_t g = cellBase + cl;
            me[u] = load_pub(g);
            lf[u] = load_pub(c > 0 ? g - 1 : g);
            up[u] = load_pub(r > 0 ? g - HC : g);
        }
#pragma unroll
        for (int u = 0; u < kTrip; ++u)
        {
            const int cell = c0 + u * kStripThreads + t;
            if (cell >= p.cells)
                continue;
            const int r = cell / HC, c = cell - r * HC;
            const CellPub &M = me[u], &L = lf[u], &Up = up[u];
            uint32_t edges = 0;
            if (c > 0)
            {
                if (can_be_merged(L.nx, L.ny, L.nz, L.d, M.nx, M.ny, M.nz, M.cx, M.cy, M.cz, M.tol, p.cosMergeA))
                    edges |= kFlagLeftToMe;
                if (can_be_merged(M.nx, M.ny, M.nz, M.d, L.nx, L.ny, L.nz, L.cx, L.cy, L.cz, L.tol, p.cosMergeA))
                    edges |= kFlagMeToLeft;
            }
            if (r > 0)
            {
                if (can_be_merged(Up.nx, Up.ny, Up.nz, Up.d, M.nx, M.ny, M.nz, M.cx, M.cy, M.cz, M.tol, p.cosMergeA))
                    edges |= kFlagUpToMe;
                if (can_be_merged(M.nx, M.ny, M.nz, M.d, Up.nx, Up.ny, Up.nz, Up.cx, Up.cy, Up.cz, Up.tol, p.cosMergeA))
                    edges |= kFlagMeToUp;
            }
            if (edges)
                atomicOr(&p.cell_flags[cellBase + cell], edges);
        }
    }
}


// the latency instance: stage A in one launch (frames x strips workgroups); the grow kernel then has no tile-boundary rows to do
hipError_t launch_cell_strips(const StageAParams& p, int nFrames, uint32_t* frameCounters, hipStream_t stream)
{
    const int stripsPerRow = (p.hCells + kStripCells - 1) / kStripCells;
    const int grid = nFrames * stripsPerRow * p.vCells;
    if (p.depth)
        hipLaunchKernelGGL(cape_cell_strip_kernel<false>, dim3(grid), dim3(kStripThreads), 0, stream, p, frameCounters);
    else
        hipLaunchKernelGGL(cape_cell_strip_kernel<true>, dim3(grid), dim3(kStripThreads), 0, stream, p, frameCounters);
    return hipGetLastError();
}

// The 1024-thread instance (a 640x480 frame in ONE tile, no tile-boundary rows left to the grow kernel) exists for
// experiments only: its 12 busy waves share the four SIMDs of a single CU and the fits serialise (28 us per frame, measured),
// while three 256-thread tiles run on three CUs side by side (~9 us).  smallBatchFrames is 0 by default.
int cell_plane_threads(const StageAParams& p, int nFrames) { return nFrames <= p.smallBatchFrames ? 1024 : 256; }
// cell rows per workgroup of stage A2 = rows whose vertical edge predicates it evaluates itself (the grow kernel does
// the rows r = k * rowsPerTile, k >= 1)
int cell_plane_rows_per_tile(const StageAParams& p, int nFrames)
{
    const int t = cell_plane_threads(p, nFrames) / p.hCells;
    return t > 0 ? t : 1;
}

hipError_t launch_cell_plane(const StageAParams& p, int nFrames, hipStream_t stream)
{
    // a few frames: one wide workgroup per frame (a 640x480 grid is one tile: the fits of all its cells run at once);
    // a batch: 256-thread workgroups, four resident per CU, each walking its frame tile by tile
    const int threads = cell_plane_threads(p, nFrames);
    const int rowsPerTile = cell_plane_rows_per_tile(p, nFrames);
    const int tiles = (p.vCells + rowsPerTile - 1) / rowsPerTile;
    if (threads == 1024)
        hipLaunchKernelGGL(cape_cell_plane_kernel<1024>, dim3(nFrames * tiles), dim3(1024), 0, stream, p, nFrames);
    else
        hipLaunchKernelGGL(cape_cell_plane_kernel<256>, dim3(nFrames * tiles), dim3(256), 0, stream, p, nFrames);
    return hipGetLastError();
}

} // namespace cape
