// Cylinder RANSAC inside the grow wave (one wavefront per frame).
//
// Replaces Cylinder_Segment::Cylinder_Segment(planeGrid, isActivatedMask, cellActivatedCount) and run_ransac_loop
// (reference src/features/primitives/cylinder_segment.cpp:35-322) together with
// Primitive_Detection::cylinder_fitting / find_plane_segment_in_cylinder / add_cylinder_to_features
// (primitive_detection.cpp:413-501).
//
// What is parallel and what is not: distances of one RANSAC hypothesis to all remaining cells are evaluated by the
// 64 lanes; every floating-point SUM the reference accumulates in a loop (covariance GEMM, MSAC cost, LLS sums, MSE,
// merged moment sums) is not exact, so its order is observable and it is accumulated in ascending index order by one
// lane per quantity.  Random draws come from the precomputed mt19937(0) table (the reference's thread_local engine
// restarts on every frame because find_primitives runs on a fresh std::async thread, src/rgbd_slam.cpp:291).
#pragma once
#include <hip/hip_runtime.h>

#include "cape_device.h"
#include "cape_internal.h"
#include "cape_staged.h"

namespace cape {

// wave-local ordering point (see CAPE_WAVE_SYNC in cape_grow.hip): the waves of a workgroup are independent frames,
// so there is no s_barrier; drain all counters so that LDS / global scratch written by one lane is visible to the others
#define CAPE_CYL_SYNC()                                                                                       \
    do                                                                                                       \
    {                                                                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                               \
        __builtin_amdgcn_s_waitcnt(0);                                                                       \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                               \
    } while (0)

#ifdef CAPE_B_PROFILE
#define CAPE_CYL_TICK(k)                                                                                     \
    do                                                                                                       \
    {                                                                                                        \
        const unsigned long long _n = __builtin_amdgcn_s_memtime();                                          \
        if (lane == 0)                                                                                       \
            c.dbg[(k)] += _n - _ct;                                                                          \
        _ct = _n;                                                                                            \
    } while (0)
#define CAPE_CYL_TICK_INIT() unsigned long long _ct = __builtin_amdgcn_s_memtime()
#else
#define CAPE_CYL_TICK(k)
#define CAPE_CYL_TICK_INIT()
#endif

struct CylCtx
{
    const StageBParams* p;
    int lane;
    size_t cellBase;
    int C;
    const unsigned short* s_list; // activated cells, ascending (= _local2globalMap)
    int total;                    // _cellActivatedCount
    double* s_dist;               // N f64: MSAC cost of every remaining cell for the current hypothesis
    unsigned short* s_ids;        // idsLeft
    unsigned char* s_idmask;      // idsLeftMask
    unsigned char* s_cur;         // inliers of the current hypothesis
    unsigned char* s_best;        // finalInlierIndexes as flags
    double* scratch;              // [N][6] projected normals / projected centroids of this frame
    double* s_stage;              // kStageChunk x 10 f64 staging buffer (cape_staged.h)
    double* s_seg;
    unsigned char* s_lab;
    unsigned char* s_cyl;
    cape_frame_record* rec;
    unsigned long long* dbg;      // per-frame phase ticks (profiling builds)
};

__device__ __forceinline__ int cyl_wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o);
    return v;
}

// returns with nSeg / nCylLabels / rngPos / status updated; nCylFits is incremented by the caller
__device__ inline void cylinder_fitting(const CylCtx& c, int& nSeg, int& nCylLabels, int nCylFits, int& rngPos,
                                        uint32_t& status, bool& planeOverflow)
{
    const StageBParams& p = *c.p;
    const int lane = c.lane;
    const int N = c.total;
    const double* planeBase = p.cell_plane + c.cellBase * kPlaneStride;
    const double* sumsBase = p.cell_sums + c.cellBase * kSumStride;

    CAPE_CYL_TICK_INIT();
    // ---- cov = (M * M^T) / (cols - 1), M = [normals, -normals] (cylinder_segment.cpp:47-89): ascending column order
    double cov6;
    {
        // lane e -> (r,c) of the lower triangle: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2)
        const int e = lane < 6 ? lane : 0;
        const int r = (e == 0) ? 0 : (e <= 2 ? 1 : 2);
        const int cc = (e == 0 || e == 1 || e == 3) ? 0 : ((e == 2 || e == 4) ? 1 : 2);
        double acc = 0.0;
        for (int half = 0; half < 2; ++half)
        {
            // records: (nx, ny, nz, d) of every activated cell, staged through LDS (2 pieces of cell_plane)
            staged_for_each<2>(
                    N, planeBase, kPlaneStride, 0, [&](int e) { return (int)c.s_list[e]; }, c.s_stage, lane,
                    [&](int, const double* rec) { acc += rec[r] * rec[cc]; }); // (-a)*(-b) == a*b in the second half
        }
        cov6 = acc / (double)(2 * N - 1);
    }
    const double m00 = __shfl(cov6, 0), m10 = __shfl(cov6, 1), m11 = __shfl(cov6, 2);
    const double m20 = __shfl(cov6, 3), m21 = __shfl(cov6, 4), m22 = __shfl(cov6, 5);
    Eig3 eg;
    self_adjoint_eigen3(m00, m10, m11, m20, m21, m22, eg);
    const double score = eg.val[2] / eg.val[0];
    CAPE_CYL_TICK(12); // covariance + eigen
    if (score < (double)75.0f) // cylinderRansacMinimumScore, checkpoint 1 (:95-102)
        return;
    const double ax = eg.q[0][0], ay = eg.q[1][0], az = eg.q[2][0];

    // ---- projection on the plane orthogonal to the axis (:107-125)
    for (int j = lane; j < N; j += 64)
    {
        const double* pl = planeBase + (size_t)c.s_list[j] * kPlaneStride;
        const double nx = pl[0], ny = pl[1], nz = pl[2], cx = pl[4], cy = pl[5], cz = pl[6];
        const double cdt = dot3(ax, ay, az, cx, cy, cz);
        const double ndt = dot3(ax, ay, az, nx, ny, nz);
        const double px = nx - ndt * ax, py = ny - ndt * ay, pz = nz - ndt * az;
        const double nrm = sqrt((px * px + py * py) + pz * pz);
        double* o = c.scratch + (size_t)j * 6;
        o[0] = px / nrm;
        o[1] = py / nrm;
        o[2] = pz / nrm;
        o[3] = cx - cdt * ax;
        o[4] = cy - cdt * ay;
        o[5] = cz - cdt * az;
        c.s_ids[j] = (unsigned short)j;
        c.s_idmask[j] = 1;
    }
    CAPE_CYL_SYNC();

    CAPE_CYL_TICK(13); // projection
    int planeSegmentsLeft = N;
    int idsLeftCount = N;
    const float maxSqrtDistF = 0.04f; // cylinderRansacSqrtMaxDistance
    const double maxSqrtDist = (double)maxSqrtDistF;

    // ---- sequential RANSAC (:146-224)
    while (planeSegmentsLeft > p.minCellActivated && (double)planeSegmentsLeft > 0.1 * (double)N)
    {
        // ===== run_ransac_loop (:227-322)
        int bestCount = 0;
        if (idsLeftCount >= 3)
        {
            const int m = idsLeftCount;
            const unsigned inliersAccepted = (unsigned)floor(0.9 * (double)m);
            double minHyp = (double)(maxSqrtDistF * (float)m);
            int prevBestCount = 0; // size of the vector swapped out of finalInlierIndexes
            for (int j = lane; j < N; j += 64)
                c.s_best[j] = 0;
            for (int it = 0; it < p.ransacMaxIterations; ++it)
            {
                int id[3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
                {
                    double U = 0.0;
                    if (rngPos < p.rngCount)
                        U = p.rngTable[rngPos];
                    else
                        status |= CAPE_FRAME_RNG_EXHAUSTED;
                    ++rngPos;
                    id[q] = c.s_ids[(unsigned)floor(U * (double)(unsigned)m)];
                }
                const double* t1 = c.scratch + (size_t)id[0] * 6;
                const double* t2 = c.scratch + (size_t)id[1] * 6;
                const double* t3 = c.scratch + (size_t)id[2] * 6;
                const double n1x = t1[0], n1y = t1[1], n1z = t1[2], c1x = t1[3], c1y = t1[4], c1z = t1[5];
                const double n2x = t2[0], n2y = t2[1], n2z = t2[2], c2x = t2[3], c2y = t2[4], c2z = t2[5];
                const double n3x = t3[0], n3y = t3[1], n3z = t3[2], c3x = t3[3], c3y = t3[4], c3z = t3[5];
                const double sNx = (n1x + n2x) + n3x, sNy = (n1y + n2y) + n3y, sNz = (n1z + n2z) + n3z;
                const double sCx = (c1x + c2x) + c3x, sCy = (c1y + c2y) + c3y, sCz = (c1z + c2z) + c3z;
                const double a = 1.0 - ((sNx * sNx + sNy * sNy) + sNz * sNz) / 9.0;
                const double prx = (n1x * c1x + n2x * c2x) + n3x * c3x;
                const double pry = (n1y * c1y + n2y * c2y) + n3y * c3y;
                const double prz = (n1z * c1z + n2z * c2z) + n3z * c3z;
                const double b = ((prx + pry) + prz) / 3.0 - (dot3(sNx, sNy, sNz, sCx, sCy, sCz) / 9.0);
                const double radius = b / a;
                const double invR2 = 1.0 / (radius * radius);
                const double ctx = (sCx - radius * sNx) / 3.0;
                const double cty = (sCy - radius * sNy) / 3.0;
                const double ctz = (sCz - radius * sNz) / 3.0;

                // MSAC truncated distances of all remaining cells (parallel), cost summed in ascending order
                int curLocal = 0;
                for (int jj = lane; jj < m; jj += 64)
                {
                    const int i = c.s_ids[jj];
                    const double* t = c.scratch + (size_t)i * 6;
                    const double vx = (t[3] - radius * t[0]) - ctx;
                    const double vy = (t[4] - radius * t[1]) - cty;
                    const double vz = (t[5] - radius * t[2]) - ctz;
                    const double distance = ((vx * vx + vy * vy) + vz * vz) * invR2;
                    const bool inl = distance < maxSqrtDist;
                    c.s_cur[i] = inl ? 1 : 0;
                    c.s_dist[jj] = inl ? distance : maxSqrtDist;
                    curLocal += inl ? 1 : 0;
                }
                const int curCount = cyl_wave_sum(curLocal);
                CAPE_CYL_SYNC();
                double dist = 0.0;
                {
                    int jj = 0;
                    for (; jj + 4 <= m; jj += 4)
                    {
                        const double d0 = c.s_dist[jj], d1 = c.s_dist[jj + 1], d2 = c.s_dist[jj + 2], d3 = c.s_dist[jj + 3];
                        dist += d0;
                        dist += d1;
                        dist += d2;
                        dist += d3;
                    }
                    for (; jj < m; ++jj)
                        dist += c.s_dist[jj];
                }
                bool stop = false;
                if (dist < minHyp)
                {
                    minHyp = dist;
                    for (int jj = lane; jj < m; jj += 64)
                    {
                        const int i = c.s_ids[jj];
                        c.s_best[i] = c.s_cur[i];
                    }
                    prevBestCount = bestCount; // inlierIndexes now holds the previous best (swap)
                    bestCount = curCount;
                    // early-stop quirk (:308-312): tests the swapped-out vector
                    stop = (unsigned)prevBestCount > inliersAccepted;
                }
                CAPE_CYL_SYNC();
                if (stop)
                    break;
            }
        }
        CAPE_CYL_TICK(14); // RANSAC iterations
        // checkpoint 2
        if (bestCount < 6)
            break;
        const int maxInliers = bestCount;

        // ===== LLS over all inliers, ascending i (:157-186): lanes 0-2 sumN, 3-5 sumC, 6 b
        double chain = 0.0;
        staged_for_each<3>(
                N, c.scratch, 6, 0, [&](int e) { return e; }, c.s_stage, lane, [&](int i, const double* t) {
                    if (c.s_best[i])
                    {
                        // lanes 0..5 take component `lane`, lane 6 the n.c product -- branch-free
                        const double dotv = (t[0] * t[3] + t[1] * t[4]) + t[2] * t[5];
                        const double comp = t[lane < 6 ? lane : 0];
                        chain += (lane < 6) ? comp : dotv;
                    }
                });
        const double sNx = __shfl(chain, 0), sNy = __shfl(chain, 1), sNz = __shfl(chain, 2);
        const double sCx = __shfl(chain, 3), sCy = __shfl(chain, 4), sCz = __shfl(chain, 5);
        double b = __shfl(chain, 6);
        // remove the inliers from the remaining ids (:161-179)
        {
            int newCount = 0;
            for (int base = 0; base < N; base += 64)
            {
                const int i = base + lane;
                const bool inl = i < N && c.s_best[i];
                const bool keep = i < N && !inl && c.s_idmask[i];
                const unsigned long long kb = __ballot(keep);
                CAPE_CYL_SYNC(); // all reads of s_ids[...] of the previous pass are done before it is rewritten
                if (inl)
                    c.s_idmask[i] = 0;
                if (keep)
                    c.s_ids[newCount + __popcll(kb & ((1ull << lane) - 1ull))] = (unsigned short)i;
                newCount += __popcll(kb);
            }
            idsLeftCount = newCount;
            planeSegmentsLeft -= maxInliers;
        }
        CAPE_CYL_SYNC();

        const double kk = (double)((unsigned long long)maxInliers * (unsigned long long)maxInliers);
        const double oneOverSq = 1.0 / kk;
        const double a = 1 - ((sNx * sNx + sNy * sNy) + sNz * sNz) * oneOverSq;
        b /= (double)maxInliers;
        b -= dot3(sNx, sNy, sNz, sCx, sCy, sCz) * oneOverSq;
        double radius = b / a;
        const double ctx = (sCx - radius * sNx) / (double)maxInliers;
        const double cty = (sCy - radius * sNy) / (double)maxInliers;
        const double ctz = (sCz - radius * sNz) / (double)maxInliers;
        if (radius < 0)
            radius = -radius;

        // MSE of the inliers' (unprojected) centroids to the axis line (:198-218)
        const double P2x = ctx + ax, P2y = cty + ay, P2z = ctz + az;
        const double dx = P2x - ctx, dy = P2y - cty, dz = P2z - ctz;
        const double P1P2d = sqrt((dx * dx + dy * dy) + dz * dz);
        double mse = 0.0;
        {
            // the per-cell squared distances are independent: all lanes compute them (cx, cy, cz straight from
            // cell_plane) into LDS, then they are added in ascending order like the reference's loop
            for (int i = lane; i < N; i += 64)
            {
                double t2 = 0.0;
                if (c.s_best[i])
                {
                    const double* pl = planeBase + (size_t)c.s_list[i] * kPlaneStride;
                    const double wx = pl[4] - P2x, wy = pl[5] - P2y, wz = pl[6] - P2z;
                    const double crx = dy * wz - dz * wy;
                    const double cry = dz * wx - dx * wz;
                    const double crz = dx * wy - dy * wx;
                    const double t = sqrt((crx * crx + cry * cry) + crz * crz) / P1P2d - radius;
                    t2 = t * t;
                }
                c.s_dist[i] = t2;
            }
            CAPE_CYL_SYNC();
            for (int i = 0; i < N; ++i)
                if (c.s_best[i])
                    mse += c.s_dist[i];
            CAPE_CYL_SYNC();
        }
        mse /= (double)maxInliers;

        // ===== cylinder_fitting's per-segment work (primitive_detection.cpp:488-500): merged plane of the inlier cells
        const int ql = lane < 10 ? lane : 0;
        double acc = 0.0; // Plane_Segment newMergedPlane: cleared sums
        staged_for_each<5>(
                N, sumsBase, kSumStride, 0, [&](int e) { return (int)c.s_list[e]; }, c.s_stage, lane,
                [&](int i, const double* rec) {
                    if (c.s_best[i])
                        acc += rec[ql];
                });
        double S[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            S[k] = __shfl(acc, k);
        const double cnt = __shfl(acc, 9);
        PlaneFit f;
        fit_plane(S, (uint32_t)cnt, f);

        // ===== add_cylinder_to_features (:437-476): model selection on MSE
        if (f.mse < mse)
        {
            if (nSeg >= CAPE_MAX_PLANES)
            {
                planeOverflow = true;
                return;
            }
            if (lane == 0)
            {
                double* o = c.s_seg + nSeg * 20;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    o[k] = S[k];
                o[9] = cnt;
                double nx = f.nx, ny = f.ny, nz = f.nz;
                normalize3(nx, ny, nz); // _planeSegments.push_back copies the segment
                o[10] = nx; o[11] = ny; o[12] = nz; o[13] = f.d;
                o[14] = f.cx; o[15] = f.cy; o[16] = f.cz;
                o[17] = f.mse; o[18] = f.score; o[19] = f.planar ? 1.0 : 0.0;
            }
            ++nSeg;
            for (int i = lane; i < N; i += 64)
                if (c.s_best[i])
                    c.s_lab[c.s_list[i]] = (unsigned char)nSeg;
        }
        else
        {
            if (nCylLabels >= CAPE_MAX_CYLINDERS)
            {
                status |= CAPE_FRAME_CYL_OVERFLOW;
                return;
            }
            if (lane == 0)
            {
                cape_cylinder* o = &c.rec->cylinders[nCylLabels];
                o->axis[0] = ax; o->axis[1] = ay; o->axis[2] = az;
                o->radius = __builtin_nan(""); // shape_primitives.cpp:17-24 over a copy whose _segmentCount is 0
                o->kept = 0;
                o->region = (uint32_t)nCylFits;
            }
            ++nCylLabels;
            for (int i = lane; i < N; i += 64)
                if (c.s_best[i])
                    c.s_cyl[c.s_list[i]] = (unsigned char)nCylLabels;
        }
        CAPE_CYL_SYNC();
        CAPE_CYL_TICK(15); // LLS + MSE + merged plane + labels
    }
}

} // namespace cape
