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
//
// This file holds A1 only and is compiled with -fno-slp-vectorize (Makefile): left to itself the SLP vectoriser pairs the six
// f32 products of a pixel into v_pk_mul_f32, which costs as much as the two v_mul_f32_e32 it replaces on gfx950
// (profiles/r02_valu_rates.txt: 4.5 against 2 x 2.3 cycles) PLUS the v_mov_b32 / v_pk_mov_b32 that build its register pairs
// (36 moves per 16 pixels; 94 -> 84 VGPRs without them).  A2 and the strip kernel (cape_cell_fit.hip) keep the vectoriser:
// A2 measured 2 % slower without it.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cape_cell_acc.h"

namespace cape {

template <bool U16> __global__ __launch_bounds__(kThreadsA, CAPE_A_WAVES) void cape_cell_moments_kernel(StageAParams p)
{
    // LDS: per-thread partials (written behind the row loop) and the 64 cells' centre row / centre column samples (written in it)
    __shared__ double s_part[kThreadsA * kPartStride]; // [thread][9 sums, count, (zmin,zmax) packed in the pad slot]
    __shared__ float s_row[64 * kCell];  // local row 10 of every cell (idx 200..219)
    __shared__ float s_col[64 * kCell];  // local column 10 of every cell (idx 10, 30, ..., 390)
    __shared__ float s_corner[64 * 2];   // first and last pixel of every cell (the centre pixel is s_row[.. + 10])
    __shared__ float s_trash[kThreadsA + kCell]; // where the lanes that do not hold a cell's centre column send their row's store

    const int t = threadIdx.x;
    const int frame = blockIdx.x / p.pairsPerFrame;
    const int pair = blockIdx.x - frame * p.pairsPerFrame;
    const size_t frameOff = (size_t)frame * p.W * p.H;

    const int bsel = t / kBandThreads;
    const int q = t - bsel * kBandThreads;
    const int band = pair * 2 + bsel;
    const int cellRow = band / p.segsPerRow;
    const int seg = band - cellRow * p.segsPerRow;
    const int col0 = seg * 640 + q * 4;
    const bool active = (band < p.bandsPerFrame) && (col0 < p.W);
    const int cseg = q / 5;          // cell within the band segment
    const int j = q - cseg * 5;      // float4 within the cell row
    const int lcell = bsel * 32 + cseg;

    // ------------------------------------------------------------------ streaming accumulation
    PxAcc A;
#pragma unroll
    for (int k = 0; k < 9; ++k)
        A.S[k] = 0.0;
    A.n = 0;
    A.zminBits1 = 0xFFFFFFFFu;
    A.zmaxBits = 0u;
    if (active)
    {
        const double a0 = p.acol[col0], a1 = p.acol[col0 + 1], a2 = p.acol[col0 + 2], a3 = p.acol[col0 + 3];
        // addresses = one base per FRAME (uniform: it lives in scalar registers and advances row by row with scalar adds)
        // + one 32-bit element offset per thread (its column inside the frame): no 64-bit vector address arithmetic per load
        const uint32_t pixOff32 = (uint32_t)(cellRow * kCell) * (uint32_t)p.W + (uint32_t)col0;
        const float* frameBase = p.depth + frameOff;                 // float32 millimetres
        const uint16_t* frameBase16 = p.depth_u16 + frameOff;        // raw sensor units (U16 variant)
        const float scale16 = p.u16_scale;
        const double* brow = p.brow + cellRow * kCell;
        const size_t W = (size_t)p.W;

        // rows in groups of kGroup, ping-pong buffered: the loads of group g+1 are in flight while group g is summed
        constexpr int kGroup = CAPE_A_GROUP, kGroups = kCell / kGroup;
        static_assert(kCell % kGroup == 0 && kGroups % 2 == 0, "the ping-pong loop consumes two groups per trip");
        // the buffers hold what was LOADED (float4, or the raw ushort4 of the U16 variant): converting at load time would
        // make every load wait for its data on the spot and serialise the prefetch with the arithmetic
        using Raw = typename std::conditional<U16, ushort4, float4>::type;
        Raw bufA[kGroup], bufB[kGroup];
        auto load_group = [&](Raw (&buf)[kGroup], int g) {
#pragma unroll
            for (int i = 0; i < kGroup; ++i)
            {
                const size_t rowOff = (size_t)(kGroup * g + i) * W; // uniform
                // non-temporal: the image is read exactly once, nothing of it is worth a cache line (1 % by interleaved A/B runs)
                if constexpr (U16)
                {
                    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
                    const u16x4 v = __builtin_nontemporal_load(reinterpret_cast<const u16x4*>((frameBase16 + rowOff) + pixOff32));
                    buf[i] = make_ushort4(v.x, v.y, v.z, v.w);
                }
                else
                {
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>((frameBase + rowOff) + pixOff32));
                    buf[i] = make_float4(v.x, v.y, v.z, v.w);
                }
            }
        };
        auto to_f4 = [&](const Raw& rw) {
            if constexpr (U16)
            {
                // N4: the PNG payload of the TUM / CAPE datasets, converted exactly like
                // cv::Mat::convertTo(CV_32F, scale) does (examples/main_TUM.cpp:242): float(raw) * float(scale)
                // (one v_mul_f32_e32 each: this file is compiled without the SLP vectoriser, which used to pair them into
                // v_pk_mul_f32 + register-pair moves -- 2.00 ms per 4 096 frames packed against 1.54 ms; through round 4 they
                // were inline asm for that reason)
                return make_float4((float)rw.x * scale16, (float)rw.y * scale16, (float)rw.z * scale16, (float)rw.w * scale16);
            }
            else
            {
                return rw;
            }
        };
        // samples for the continuity cross scan and the tolerance corners.  The centre column is one store per row by EVERY
        // lane -- the lane that holds pixel column 10 of its cell (j == 2) into s_col, the others into a trash slot of their
        // own: a predicated store costs the wave an exec-mask round trip per row.  First / last pixel of the cell come from
        // the peeled first / last trip (compile-time rows), the centre row from the trip that holds row 10.
        float* const colp = (j == 2) ? &s_col[lcell * kCell] : &s_trash[t];
        auto sum_group = [&](const Raw (&rawbuf)[kGroup], int g, auto where) {
            constexpr int kWhere = decltype(where)::value; // 0: first trip, 1: loop, 2: last trip
            float4 buf[kGroup];
#pragma unroll
            for (int i = 0; i < kGroup; ++i)
                buf[i] = to_f4(rawbuf[i]);
#pragma unroll
            for (int i = 0; i < kGroup; ++i)
            {
                const int r = kGroup * g + i;
                acc_f4(buf[i], a0, a1, a2, a3, brow[r], A);
                colp[r] = buf[i].z; // pixel column 10 of the cell (lanes j == 2)
                if (kWhere == 1 && r == kCell / 2)
                    *reinterpret_cast<float4*>(&s_row[lcell * kCell + 4 * j]) = buf[i];
                if (kWhere == 0 && i == 0 && g == 0 && j == 0)
                    s_corner[lcell * 2] = buf[i].x;
                if (kWhere == 2 && r == kCell - 1 && j == 4)
                    s_corner[lcell * 2 + 1] = buf[i].w;
            }
        };
        using First = std::integral_constant<int, 0>;
        using Loop = std::integral_constant<int, 1>;
        using Last = std::integral_constant<int, 2>;
        static_assert(kGroups >= 6 && (kCell / 2) / kGroup >= 2 && (kCell / 2) / kGroup < kGroups - 2, "row 10 belongs to a trip of the loop");
        load_group(bufA, 0);
        // first and last trip peeled: compile-time rows for the corner samples, and no `if (g + 2 < kGroups)` inside the loop
        // (with it the loop carried both buffers through copies, eight v_mov_b64 per trip)
        load_group(bufB, 1);
        sum_group(bufA, 0, First{});
        load_group(bufA, 2);
        sum_group(bufB, 1, First{});
#pragma unroll 1
        for (int g = 2; g < kGroups - 2; g += 2)
        {
            load_group(bufB, g + 1);
            sum_group(bufA, g, Loop{});
            load_group(bufA, g + 2);
            sum_group(bufB, g + 1, Loop{});
        }
        load_group(bufB, kGroups - 1);
        sum_group(bufA, kGroups - 2, Last{});
        sum_group(bufB, kGroups - 1, Last{});
    }
    {
        double* dst = s_part + t * kPartStride;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            dst[k] = A.S[k];
        dst[9] = (double)A.n;
        reinterpret_cast<uint2*>(dst + 10)[0] = make_uint2(A.zminBits1, A.zmaxBits);
    }
    __syncthreads();

    // ------------------------------------------------------------------ behind the barrier: wave 4 sums, waves 0..3 scan
    // The workgroup's LDS and wave slots are held until its last wave leaves, so the work behind the last pixel is spread for
    // LATENCY: the 5-partials sums on one wave beside the scans, the scans cut in four (below) -- profiles/r04_a1_phases.txt.
    if (t >= 256)
    {
        // 5 partials -> one cell (exact, any order): 64 cells x 10 quantities = 640 tasks, ten per lane, consecutive lanes store
        // consecutive doubles (AoS [cell][10])
        const int l = t - 256;
        size_t base[2];
        int lim[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
        {
            const int bnd = pair * 2 + cb;
            const int cr = bnd / p.segsPerRow;
            const int sg = bnd - cr * p.segsPerRow;
            base[cb] = (size_t)frame * p.cells + cr * p.hCells + sg * 32;
            lim[cb] = bnd < p.bandsPerFrame ? p.hCells - sg * 32 : 0; // cells of the band that exist
        }
#pragma unroll
        for (int k = 0; k < 10; ++k)
        {
            const int e = l + 64 * k;
            const int cell = e / 10;
            const int m = e - cell * 10;
            const int cb = cell >> 5, cs = cell & 31;
            if (cs < (cb ? lim[1] : lim[0]))
            {
                const double* src = s_part + (cb * kBandThreads + cs * 5) * kPartStride + m;
                double acc = src[0];
                acc += src[kPartStride];
                acc += src[2 * kPartStride];
                acc += src[3 * kPartStride];
                acc += src[4 * kPartStride];
                p.cell_sums[((cb ? base[1] : base[0]) + cs) * kSumStride + m] = acc;
            }
        }
        return;
    }

    // Continuity cross scans, four lane-tasks per cell, sixteen cells per wave: lane = (task, cell) with task = horizontal /
    // vertical x first / second half of the scan's steps; the four verdicts meet through two cross-lane reads.
    //   is_cell_horizontal_continuous (plane_segment.cpp:82-100): local row 10, idx 200..219, steps 1..19
    //   is_cell_vertical_continuous (:62-80): local column 10, idx 10, 30, .., 370: steps 1..18 (the loop stops before idx 390)
    // A step only needs `last`, the latest depth that passed.  If the first half (seed test, steps 1..9) passes, every positive
    // depth in it has become `last` in turn, so the second half (steps 10..) starts from the last positive depth among steps
    // 1..9, else from the seed max(z[0], z[1]) -- and if the first half fails the cell is discontinuous whatever the second
    // half says.  The steps are straight-line code (is_continuous_flat), so all four tasks are one instruction stream of ten
    // steps.  (Through round 4 wave 0 ran both scans of all 64 cells one behind the other, 37 branching steps on a lone wave;
    // round 3's other extreme -- one lane per TEST on all five waves, a look-back loop per test -- measured slower.)
    const int w = t >> 6, l = t & 63;
    const int c64 = w * 16 + (l & 15);   // cell of the workgroup (= lcell of the lanes that streamed it)
    const bool vertical = (l & 32) != 0;
    const bool second = (l & 16) != 0;
    bool continuous;
    {
        const float* zs = (vertical ? s_col : s_row) + c64 * kCell;
        const float seed = std_maxf(zs[0], zs[1]);
        float last = seed;
#pragma unroll
        for (int i = 1; i <= 9; ++i)
        {
            const float z = zs[i];
            last = (second && z > 0) ? z : last;
        }
        continuous = second || !(seed <= 0);
        const float* zt = zs + (second ? 10 : 1);
        const int steps = second ? (vertical ? 9 : 10) : 9;
#pragma unroll
        for (int k = 0; k < 10; ++k)
        {
            const bool pass = is_continuous_flat(zt[k], last);
            continuous &= pass | (k >= steps);
        }
        continuous &= __shfl_xor((int)continuous, 16) != 0;
        continuous &= __shfl_xor((int)continuous, 32) != 0;
    }
    const int fb = c64 >> 5, fs = c64 & 31;
    const int fband = pair * 2 + fb;
    if ((l >> 4) != 0 || fband >= p.bandsPerFrame)
        return;
    const int fRow = fband / p.segsPerRow;
    const int fSeg = fband - fRow * p.segsPerRow;
    const int fCol = fSeg * 32 + fs;
    if (fCol >= p.hCells)
        return;
    // exactness guard: all addends of every sum within 2^20 of each other (see header)
    uint32_t zminBits1 = 0xFFFFFFFFu, zmaxBits = 0u;
    uint32_t n = 0;
    {
        const int pbase = fb * kBandThreads + fs * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k)
        {
            const uint2 zr = reinterpret_cast<const uint2*>(s_part + (pbase + k) * kPartStride + 10)[0];
            zminBits1 = min(zminBits1, zr.x);
            zmaxBits = max(zmaxBits, zr.y);
            n += (uint32_t)s_part[(pbase + k) * kPartStride + 9];
        }
    }
    // back to depths: with n > 0 at least one pixel was valid, so the minimum is a real pattern - 1
    const float zmin = __uint_as_float(zminBits1 + 1u), zmax = __uint_as_float(zmaxBits);
    const float rab = fmaxf(p.ratio_col[fCol], p.ratio_row[fRow]);
    // all pixels +0 or positive and finite (acc_px_fast's precondition), and their range within the exactness bound
    const bool exact_ok = zmaxBits <= 0x7F7FFFFFu && ((n == 0) || (zmax * rab <= 512.0f * zmin));

    const size_t gcell = (size_t)frame * p.cells + fRow * p.hCells + fCol;
    CellAux aux;
    aux.z0 = s_corner[c64 * 2];
    aux.z399 = s_corner[c64 * 2 + 1];
    aux.flags = (n & kCountMask) | (continuous ? kAuxContinuous : 0u) | (exact_ok ? kAuxExact : 0u);
    aux.zc = s_row[c64 * kCell + kCell / 2];
    p.cell_aux[gcell] = aux;
}

// every launch helper reports its own failure: hipGetLastError() right behind the launch (a later runtime call would
// overwrite the sticky-free error state on ROCm < 7)
hipError_t launch_cell_moments(const StageAParams& p, int nFrames, hipStream_t stream)
{
    const int grid = nFrames * p.pairsPerFrame;
    if (p.depth)
        hipLaunchKernelGGL(cape_cell_moments_kernel<false>, dim3(grid), dim3(kThreadsA), 0, stream, p);
    else
        hipLaunchKernelGGL(cape_cell_moments_kernel<true>, dim3(grid), dim3(kThreadsA), 0, stream, p);
    return hipGetLastError();
}

} // namespace cape
