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
#include "cape_wave.h"

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
            atomicAdd(&c.dbg[(k)], _n - _ct);                                                                \
        _ct = _n;                                                                                            \
    } while (0)
#define CAPE_CYL_TICK_INIT() unsigned long long _ct = __builtin_amdgcn_s_memtime()
#define CAPE_CYL_COUNT(k, v)                                                                                 \
    do                                                                                                       \
    {                                                                                                        \
        if (lane == 0)                                                                                       \
            atomicAdd(&c.dbg[(k)], (unsigned long long)(v));                                                 \
    } while (0)
#else
#define CAPE_CYL_TICK(k)
#define CAPE_CYL_TICK_INIT()
#define CAPE_CYL_COUNT(k, v)
#endif

// LabT: the type of a cell label in the instance's working grids -- a byte in the LDS-resident instances (<= 64 segments),
// 16 bits in the general instance (cape_grow_general.hip: any number of segments)
template <typename LabT> struct CylCtxT
{
    const StageBParams* p;
    int lane;
    size_t cellBase;
    int C;
    const unsigned short* s_list; // activated cells, ascending (= _local2globalMap)
    int total;                    // _cellActivatedCount
    double* s_dist;               // kStageChunk x 18 f64: staging of the combined LLS / merged-plane traversal
    unsigned short* s_ids;        // idsLeft
    unsigned char* s_idmask;      // idsLeftMask
    unsigned char* s_cur;         // inliers of the current hypothesis
    unsigned char* s_best;        // finalInlierIndexes as flags
    double* scratch;              // [N][kCylStride] projected normals / projected centroids / their dot product
    double* s_stage;              // kStageChunk x 10 f64 staging buffer (cape_staged.h)
    double* s_seg;
    LabT* s_lab;
    LabT* s_cyl;
    cape_cylinder* cylOut;        // where cylinder label k's record goes (the frame record's array; the general instance's scratch)
    int maxCylinders;             // ... and how many it holds
    int maxPlanes;                // segment slots of this kernel instance
    unsigned long long* dbg;      // per-frame phase ticks (profiling builds)
};
using CylCtx = CylCtxT<unsigned char>;

// The RANSAC distance pass.  The cells a lane evaluates (idsLeft[lane + 64 k]) do not change during one
// run_ransac_loop, so with up to kCylCacheRounds x 64 cells left (the whole 640x480 grid) their six projected coordinates
// are fetched ONCE into registers and every hypothesis is scored from there: the per-frame scratch (N x 64 B) would
// otherwise be streamed once per hypothesis by every resident wave, which overflows the 4 MB L2 of an XCD (measured:
// 48 % L2 misses, and more resident waves made the kernel slower).  The cylinder variant is built for one wave per
// SIMD (its LDS footprint admits no more), so the 144 cache registers cost no occupancy.  Larger grids stream the cells
// four rounds at a time, all twelve 16-byte loads of a trip requested before the first distance is computed.
constexpr int kCylCacheRounds = 12;
#define CAPE_CYL_ROUNDS(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11)
#define CAPE_CYL_TRIP(F) F(0) F(1) F(2) F(3)
#define CAPE_CYL_DECL(k) double2 cqa##k = make_double2(0, 0), cqb##k = cqa##k, cqc##k = cqa##k; int cqi##k = 0;
#define CAPE_CYL_FETCH(k)                                                                                    \
    if (j0 + 64 * (k) < m)                                                                                   \
    {                                                                                                        \
        const int jj_ = j0 + lane + 64 * (k);                                                                \
        cqi##k = c.s_ids[jj_ < m ? jj_ : 0];                                                                 \
        const double2* t_ = reinterpret_cast<const double2*>(c.scratch + (size_t)cqi##k * kCylStride);       \
        cqa##k = t_[0];                                                                                      \
        cqb##k = t_[1];                                                                                      \
        cqc##k = t_[2];                                                                                      \
    }
// One round = 64 cells.  The rounds are evaluated four at a time WITHOUT a branch between them (a cell past the end is
// computed and masked): a lone wave issues in order and stalls on every dependent f64 result (~9 cycles), so the only
// latency hiding there is comes from independent work in the same basic block -- the three other rounds of the group.
#define CAPE_CYL_SCORE1(k)                                                                                   \
    {                                                                                                        \
        bool inl_;                                                                                           \
        const double d_ = msac(cqa##k.x, cqa##k.y, cqb##k.x, cqb##k.y, cqc##k.x, cqc##k.y, inl_);            \
        const bool in_ = j0 + lane + 64 * (k) < m;                                                           \
        psum += in_ ? d_ : 0.0;                                                                              \
        curLocal += (in_ && inl_) ? 1 : 0;                                                                   \
        inlBits |= (in_ && inl_) ? (1u << (k)) : 0u;                                                         \
    }
#define CAPE_CYL_SCORE4(a, b, c_, d)                                                                         \
    if (j0 + 64 * (a) < m)                                                                                   \
    {                                                                                                        \
        CAPE_CYL_SCORE1(a) CAPE_CYL_SCORE1(b) CAPE_CYL_SCORE1(c_) CAPE_CYL_SCORE1(d)                         \
    }
#define CAPE_CYL_SCORE_ALL CAPE_CYL_SCORE4(0, 1, 2, 3) CAPE_CYL_SCORE4(4, 5, 6, 7) CAPE_CYL_SCORE4(8, 9, 10, 11)
// the distances of a hypothesis into LDS, for the ordered sum (only when the tree-order sum cannot decide, see below)
#define CAPE_CYL_DIST(k)                                                                                     \
    if (j0 + 64 * (k) < m)                                                                                   \
    {                                                                                                        \
        bool inl_;                                                                                           \
        const double d_ = msac(cqa##k.x, cqa##k.y, cqb##k.x, cqb##k.y, cqc##k.x, cqc##k.y, inl_);            \
        if (j0 + lane + 64 * (k) < m)                                                                        \
            c.scratch[(size_t)(j0 + lane + 64 * (k)) * kCylStride + 7] = d_;                                 \
    }
// the inlier flags of the winning hypothesis, from the bits the scoring pass left in inlBits (bit k = round k)
#define CAPE_CYL_FLAGS(k)                                                                                    \
    if (j0 + lane + 64 * (k) < m)                                                                            \
        c.s_best[cqi##k] = (unsigned char)((inlBits >> (k)) & 1u);

// streamed path (more cells than the register cache holds): the inlier flags of the hypothesis being scored go to s_cur, so
// that a winner copies them instead of fetching and scoring every cell a second time
#define CAPE_CYL_CURFLAGS(k)                                                                                 \
    if (j0 + lane + 64 * (k) < m)                                                                            \
        c.s_cur[cqi##k] = (unsigned char)((inlBits >> (k)) & 1u);

// relative distance between the tree-order sum and the ordered sum of m <= 4096 non-negative doubles: each is within
// (m - 1) * 2^-53 <= 2^-41 of the exact sum.  A test build widens it (-DCAPE_CYL_EPS=0.25) to drive the exact path.
#ifndef CAPE_CYL_PROJ_ROUNDS
#define CAPE_CYL_PROJ_ROUNDS 4
#endif
#ifndef CAPE_CYL_EPS
#define CAPE_CYL_EPS 0x1p-40
#endif

__device__ __forceinline__ int cyl_wave_sum(int v) { return wave_sum_i32(v); }

// init-less ordered sum s[0] + s[1] + ... + s[n-1] (ascending, one rounding per add, like the reference's loops) of
// NON-NEGATIVE addends parked in LDS.  The running sum of non-negative terms never decreases, so the scan may stop as
// soon as it reaches `limit`: the caller only needs to know that the total is >= limit then.
// Each lane fetches ONE element of a block of 64 (a single conflict-free ds_read_b64 per block, the next block already
// requested) and the chain takes its operands out of that register with v_readlane: two scalar moves per element in the
// shadow of the dependent add.  Reading every element with a wave-uniform LDS load cost an LDS instruction per two
// elements, and a wave gets one through only every ~16 cycles (profiles/r02_lds_rates.txt): 28 cycles per element.
// STRIDE: doubles between consecutive elements (the costs of the rare exact path are parked in the free eighth double of
// the per-cell cylinder scratch in HBM, not in LDS: a C-double LDS array cost 24 KB per wave on a 64x48 grid).
template <int STRIDE> __device__ __forceinline__ double ordered_sum_lds(const double* s, int n_, double limit, int lane)
{
    const int n = __builtin_amdgcn_readfirstlane(n_);
    double sum = 0.0;
    double cur = (lane < n) ? s[(size_t)lane * STRIDE] : 0.0;
    for (int j0 = 0; j0 < n; j0 += 64)
    {
        const int jn = j0 + 64 + lane;
        const double nxt = (jn < n) ? s[(size_t)jn * STRIDE] : 0.0; // x + (+0.0) == x for every x >= +0.0
        __builtin_amdgcn_sched_barrier(0);
        // the operand of element l + 1 is taken out while the add of element l waits for the one before it: issued right
        // in front of its own add, the scalar moves (and their way into the VALU) sat on the chain, 26 cycles per element
        double x = readlane_f64(cur, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
#pragma unroll
            for (int l = 0; l < 16; ++l)
            {
                const double xn = readlane_f64(cur, (16 * q + l + 1) & 63); // (the 64th wraps to lane 0 and is not used)
                __builtin_amdgcn_sched_barrier(0);
                sum += x;
                __builtin_amdgcn_sched_barrier(0);
                x = xn;
            }
            if (sum >= limit)
                return sum;
        }
        cur = nxt;
    }
    return sum;
}

// returns with nSeg / nCylLabels / rngPos / status updated; nCylFits is incremented by the caller
// planeOverflow: the instance is out of segment slots OR out of cylinder slots -- either way the frame goes to the next larger
// instance (the 64-segment one, then the general one, whose capacities are the grid's own bounds)
template <typename LabT>
__device__ inline void cylinder_fitting(const CylCtxT<LabT>& c, int& nSeg, int& nCylLabels, int nCylFits, int& rngPos,
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
        // lane e accumulates entry e of the lower triangle: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2), M(r,i) * M(c,i) per column.
        // The products do not depend on the running sums, so they are formed when a chunk is PARKED, by the lanes that
        // fetched it: the two pieces of a cell, (nx ny) and (nz d), sit in neighbouring lanes, one DPP swap apart, and the
        // chunk buffer holds six products per cell.  The chain is then one LDS operand and one add per column; with the
        // multiply and the second operand on it, it cost a lone wave half as much again.
        const int e = lane < 6 ? lane : 0;
        double acc = 0.0;
        for (int half = 0; half < 2; ++half)
        {
            staged_for_each_src<2, CAPE_STAGE_DEPTH_CYL>(
                    N, [&](int i, int) { return (int)c.s_list[i]; },
                    [&](int rec, int sub) { return reinterpret_cast<const double2*>(planeBase + (size_t)rec * kPlaneStride + 2 * sub); },
                    [](int) { return true; },
                    [&](double* buf, int piece, double2 v) {
                        const double ox = __longlong_as_double((long long)dpp_u64<kDppQuadXor1>((unsigned long long)__double_as_longlong(v.x)));
                        double* rec = buf + (piece >> 1) * 6;
                        if (!(piece & 1)) // (nx ny) here, nz next door
                        {
                            *reinterpret_cast<double2*>(rec) = make_double2(v.x * v.x, v.y * v.x);
                            *reinterpret_cast<double2*>(rec + 2) = make_double2(v.y * v.y, ox * v.x);
                            rec[4] = ox * v.y;
                        }
                        else
                            rec[5] = v.x * v.x;
                    },
                    c.s_stage, lane, [](int, int) {},
                    // (load() is handed the cell's place in the default layout, 4 doubles per cell)
                    [&](int, const double* rec) { return c.s_stage[((rec - c.s_stage) >> 2) * 6 + e]; },
                    [&](int, double prod) { acc += prod; }, // (-a)*(-b) == a*b in the second half
                    c.dbg);
            CAPE_CYL_TICK(28 + half); // covariance pass 1 / 2
        }
        cov6 = acc / (double)(2 * N - 1);
    }
    const double m00 = readlane_f64(cov6, 0), m10 = readlane_f64(cov6, 1), m11 = readlane_f64(cov6, 2);
    const double m20 = readlane_f64(cov6, 3), m21 = readlane_f64(cov6, 4), m22 = readlane_f64(cov6, 5);
    Eig3 eg;
    self_adjoint_eigen3(m00, m10, m11, m20, m21, m22, eg);
    const double score = eg.val[2] / eg.val[0];
    CAPE_CYL_TICK(12); // covariance + eigen
    if (score < (double)75.0f) // cylinderRansacMinimumScore, checkpoint 1 (:95-102)
        return;
    const double ax = eg.q[0][0], ay = eg.q[1][0], az = eg.q[2][0];

    // ---- projection on the plane orthogonal to the axis (:107-125); four rounds (256 cells) per trip with the sixteen
    //      16-byte loads of a trip requested before the first result is needed: one memory round trip per trip
    constexpr int kProjRounds = CAPE_CYL_PROJ_ROUNDS; // rounds of 64 cells whose loads are requested together
    for (int j0 = 0; j0 < N; j0 += 64 * kProjRounds)
    {
        double2 q0[kProjRounds], q1[kProjRounds], q2[kProjRounds], q3[kProjRounds]; // (nx ny) (nz d) (cx cy) (cz mse) of cell_plane
#pragma unroll
        for (int k = 0; k < kProjRounds; ++k)
        {
            const int j = j0 + lane + 64 * k;
            const double2* pl = reinterpret_cast<const double2*>(planeBase + (size_t)c.s_list[j < N ? j : 0] * kPlaneStride);
            q0[k] = pl[0];
            q1[k] = pl[1];
            q2[k] = pl[2];
            q3[k] = pl[3];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < kProjRounds; ++k)
        {
            const int j = j0 + lane + 64 * k;
            if (j < N)
            {
                const double nx = q0[k].x, ny = q0[k].y, nz = q1[k].x, cx = q2[k].x, cy = q2[k].y, cz = q3[k].x;
                const double cdt = dot3(ax, ay, az, cx, cy, cz);
                const double ndt = dot3(ax, ay, az, nx, ny, nz);
                const double px = nx - ndt * ax, py = ny - ndt * ay, pz = nz - ndt * az;
                const double nrm = sqrt((px * px + py * py) + pz * pz);
                const double o0 = px / nrm, o1 = py / nrm, o2 = pz / nrm;
                const double o3 = cx - cdt * ax, o4 = cy - cdt * ay, o5 = cz - cdt * az;
                double2* o = reinterpret_cast<double2*>(c.scratch + (size_t)j * kCylStride);
                o[0] = make_double2(o0, o1);
                o[1] = make_double2(o2, o3);
                o[2] = make_double2(o4, o5);
                // the LLS term b += n.dot(c) (cylinder_segment.cpp:171), ready for the ordered pass
                o[3] = make_double2((o0 * o3 + o1 * o4) + o2 * o5, 0.0);
                c.s_ids[j] = (unsigned short)j;
                c.s_idmask[j] = 1;
            }
        }
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
            // The reference keeps minHypothesisDist, the ORDERED sum of the MSAC costs of the best hypothesis so far, and only
            // ever compares against it (dist < minHypothesisDist, :296); nothing else of it is observable.  An ordered sum is
            // a serial chain over m cells, so the wave keeps an INTERVAL [minLo, minHi] that contains the best's ordered sum
            // (exact, lo == hi, until a hypothesis wins) and decides with the tree-order sum psum, whose distance to the
            // ordered sum is bounded (CAPE_CYL_EPS).  Only when the intervals overlap are the ordered sums computed -- of the
            // new hypothesis and, if it is still an interval, of the best (its parameters are kept for that).
            double minLo = (double)(maxSqrtDistF * (float)m), minHi = minLo;
            double bR = 0.0, bInvR2 = 0.0, bCx = 0.0, bCy = 0.0, bCz = 0.0; // the best hypothesis, for a late exact sum
            int prevBestCount = 0; // size of the vector swapped out of finalInlierIndexes
            for (int j = lane; j < N; j += 64)
                c.s_best[j] = 0;
            const bool cached = m <= 64 * kCylCacheRounds;
            CAPE_CYL_ROUNDS(CAPE_CYL_DECL)
            if (cached)
            {
                constexpr int j0 = 0;
                CAPE_CYL_ROUNDS(CAPE_CYL_FETCH)
            }
            // Every hypothesis of this loop is known before the loop starts: idsLeft does not change during one
            // run_ransac_loop and the draws come from the precomputed table, so lane `it` builds hypothesis `it` -- its three
            // draws, its three sample cells (one gathered memory round trip for ALL hypotheses instead of one per
            // hypothesis) and the reference's arithmetic for radius and centre (:262-283), operation for operation -- and the
            // sequential loop below only takes the five numbers out of lane `it`.  Hypotheses past an early stop are
            // computed and never looked at.  (Through round 2 the wave fetched and computed one hypothesis per iteration,
            // all lanes the same values: ~2 k cycles of exposed latency per hypothesis, which was all a small region cost.)
            const int rngBase = rngPos;
            if (p.ransacMaxIterations > 64)
                status |= CAPE_FRAME_RNG_EXHAUSTED; // cannot happen with the reference's constants (43 iterations)
            double hypR, hypInvR2, hypCx, hypCy, hypCz;
            {
                const int itL = lane < p.ransacMaxIterations ? lane : p.ransacMaxIterations - 1;
                const int last = p.rngCount - 1;
                double2 ta[3], tb[3], tc[3]; // (n.x n.y) (n.z c.x) (c.y c.z) of the three sample cells
#pragma unroll
                for (int q = 0; q < 3; ++q)
                {
                    const int di = rngBase + 3 * itL + q;
                    // table exhausted: U = 0, flagged when (if) the hypothesis is actually evaluated
                    const double u = di < p.rngCount ? p.rngTable[di < last ? di : last] : 0.0;
                    const int cell = (int)c.s_ids[(unsigned)floor(u * (double)(unsigned)m)];
                    const double2* t_ = reinterpret_cast<const double2*>(c.scratch + (size_t)cell * kCylStride);
                    ta[q] = t_[0];
                    tb[q] = t_[1];
                    tc[q] = t_[2];
                }
                const double n1x = ta[0].x, n1y = ta[0].y, n1z = tb[0].x, c1x = tb[0].y, c1y = tc[0].x, c1z = tc[0].y;
                const double n2x = ta[1].x, n2y = ta[1].y, n2z = tb[1].x, c2x = tb[1].y, c2y = tc[1].x, c2z = tc[1].y;
                const double n3x = ta[2].x, n3y = ta[2].y, n3z = tb[2].x, c3x = tb[2].y, c3y = tc[2].x, c3z = tc[2].y;
                const double sNx = (n1x + n2x) + n3x, sNy = (n1y + n2y) + n3y, sNz = (n1z + n2z) + n3z;
                const double sCx = (c1x + c2x) + c3x, sCy = (c1y + c2y) + c3y, sCz = (c1z + c2z) + c3z;
                const double a = 1.0 - ((sNx * sNx + sNy * sNy) + sNz * sNz) / 9.0;
                const double prx = (n1x * c1x + n2x * c2x) + n3x * c3x;
                const double pry = (n1y * c1y + n2y * c2y) + n3y * c3y;
                const double prz = (n1z * c1z + n2z * c2z) + n3z * c3z;
                const double b = ((prx + pry) + prz) / 3.0 - (dot3(sNx, sNy, sNz, sCx, sCy, sCz) / 9.0);
                hypR = b / a;
                hypInvR2 = 1.0 / (hypR * hypR);
                hypCx = (sCx - hypR * sNx) / 3.0;
                hypCy = (sCy - hypR * sNy) / 3.0;
                hypCz = (sCz - hypR * sNz) / 3.0;
            }
            // the hypothesis the distance macros evaluate (uniform)
            double radius = 0.0, invR2 = 0.0, ctx = 0.0, cty = 0.0, ctz = 0.0;
            auto msac = [&](double t0, double t1, double t2, double t3, double t4, double t5, bool& inl) {
                const double vx = (t3 - radius * t0) - ctx;
                const double vy = (t4 - radius * t1) - cty;
                const double vz = (t5 - radius * t2) - ctz;
                const double distance = ((vx * vx + vy * vy) + vz * vz) * invR2;
                inl = distance < maxSqrtDist;
                return inl ? distance : maxSqrtDist;
            };
            // ordered sum of the MSAC costs of the hypothesis in (radius, invR2, ctx, cty, ctz); may stop at `limit`
            auto ordered_cost = [&](double limit) {
                if (cached)
                {
                    constexpr int j0 = 0;
                    CAPE_CYL_ROUNDS(CAPE_CYL_DIST)
                }
                else
                {
                    for (int j0 = 0; j0 < m; j0 += 256)
                    {
                        CAPE_CYL_TRIP(CAPE_CYL_FETCH)
                        CAPE_CYL_TRIP(CAPE_CYL_DIST)
                    }
                }
                CAPE_CYL_SYNC();
                const double d = ordered_sum_lds<kCylStride>(c.scratch + 7, m, limit, lane);
                CAPE_CYL_SYNC(); // the parked costs are rewritten by the next call
                return d;
            };
            for (int it = 0; it < p.ransacMaxIterations; ++it)
            {
                if (rngBase + 3 * it + 2 >= p.rngCount)
                    status |= CAPE_FRAME_RNG_EXHAUSTED;
                rngPos = rngBase + 3 * (it + 1);
                const int itLane = it & 63;
                const double hR = readlane_f64(hypR, itLane), hInvR2 = readlane_f64(hypInvR2, itLane);
                const double hCx = readlane_f64(hypCx, itLane), hCy = readlane_f64(hypCy, itLane), hCz = readlane_f64(hypCz, itLane);
                radius = hR, invR2 = hInvR2, ctx = hCx, cty = hCy, ctz = hCz;

                // MSAC truncated distances of all remaining cells, in parallel, summed in tree order
                int curLocal = 0;
                double psum = 0.0;
                unsigned inlBits = 0; // cached path: bit k = the lane's cell of round k is an inlier of this hypothesis
                if (cached)
                {
                    constexpr int j0 = 0;
                    CAPE_CYL_SCORE_ALL
                }
                else
                {
                    // twelve rounds (768 cells) per trip, all 36 loads requested before the first distance: every trip exposes
                    // one memory round trip, and a lone wave has nothing else to hide it with
                    for (int j0 = 0; j0 < m; j0 += 64 * kCylCacheRounds)
                    {
                        CAPE_CYL_ROUNDS(CAPE_CYL_FETCH)
                        inlBits = 0;
                        CAPE_CYL_SCORE_ALL
                        CAPE_CYL_ROUNDS(CAPE_CYL_CURFLAGS)
                    }
                }
                const int curCount = cyl_wave_sum(curLocal);
                psum = wave_sum_f64_tree(psum);
                double lo = psum * (1.0 - CAPE_CYL_EPS), hi = psum * (1.0 + CAPE_CYL_EPS);
                bool wins;
                if (lo >= minHi)
                    wins = false; // ordered sum >= lo >= the best's
                else if (hi < minLo)
                    wins = true; // ordered sum <= hi < the best's
                else
                {
                    // cannot be told apart in tree order (never seen with the default bound outside the test build)
                    if (minLo != minHi)
                    {
                        radius = bR, invR2 = bInvR2, ctx = bCx, cty = bCy, ctz = bCz;
                        minLo = minHi = ordered_cost(__builtin_inf());
                        radius = hR, invR2 = hInvR2, ctx = hCx, cty = hCy, ctz = hCz;
                    }
                    lo = hi = ordered_cost(minHi); // stops once it reaches minHi: >= the best's, loses
                    wins = lo < minHi;
                }
                bool stop = false;
                if (wins)
                {
                    minLo = lo, minHi = hi;
                    bR = hR, bInvR2 = hInvR2, bCx = hCx, bCy = hCy, bCz = hCz;
                    if (cached)
                    {
                        constexpr int j0 = 0;
                        CAPE_CYL_ROUNDS(CAPE_CYL_FLAGS)
                    }
                    else
                    {
                        CAPE_CYL_SYNC(); // the flags the scoring pass left in s_cur
                        for (int jj = lane; jj < m; jj += 64)
                        {
                            const int i = c.s_ids[jj];
                            c.s_best[i] = c.s_cur[i];
                        }
                    }
                    prevBestCount = bestCount; // inlierIndexes now holds the previous best (swap)
                    bestCount = curCount;
                    // early-stop quirk (:308-312): tests the swapped-out vector
                    stop = (unsigned)prevBestCount > inliersAccepted;
                    CAPE_CYL_SYNC();
                }
                CAPE_CYL_COUNT(27, 1); // hypotheses evaluated
                if (stop)
                    break;
            }
        }
        CAPE_CYL_TICK(14); // RANSAC iterations
        // checkpoint 2
        if (bestCount < 6)
            break;
        const int maxInliers = bestCount;

        // ===== LLS over all inliers, ascending i (:157-186), and -- in the same chain -- the sums of cylinder_fitting's
        // merged plane of the inlier cells (primitive_detection.cpp:488-500).  Both walk the region's cells in order under
        // the same inlier mask and neither needs the other's result, so ONE traversal stages both records of a cell
        // (4 pieces of the projected scratch + 5 pieces of cell_sums) and one dependent add per cell serves both:
        //   lanes 0-2 sumN, 3-5 sumC, 6 b (the precomputed n.c product);  lanes 16-25 the merged plane's ten sums.
        // The 18-double records are staged in s_dist, idle between the RANSAC loop and the (rare) ordered MSE sum.
        double chain = 0.0;
        {
            const int slot = lane < 7 ? lane : ((lane >= 16 && lane < 26) ? 8 + (lane - 16) : 0); // the lane's double of a record
            staged_for_each_src<9, CAPE_STAGE_DEPTH_CYL>(
                    N, [&](int e, int sub) { return sub < 4 ? e : (int)c.s_list[e]; },
                    [&](int rec, int sub) {
                        return sub < 4 ? reinterpret_cast<const double2*>(c.scratch + (size_t)rec * kCylStride + 2 * sub)
                                       : reinterpret_cast<const double2*>(sumsBase + (size_t)rec * kSumStride + 2 * (sub - 4));
                    },
                    // a non-inlier is parked as +0.0, which leaves the running sums unchanged bit for bit (they are never -0.0)
                    [&](int e) { return c.s_best[e] != 0; }, stage_park_records<9>, c.s_dist, lane, [](int, int) {},
                    [&](int, const double* t) { return t[slot]; }, [&](int, double term) { chain += term; });
        }
        CAPE_CYL_TICK(16); // LLS ordered pass
        const double sNx = readlane_f64(chain, 0), sNy = readlane_f64(chain, 1), sNz = readlane_f64(chain, 2);
        const double sCx = readlane_f64(chain, 3), sCy = readlane_f64(chain, 4), sCz = readlane_f64(chain, 5);
        double b = readlane_f64(chain, 6);
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

        CAPE_CYL_TICK(17); // idsLeft compaction
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
        // The MSE only ever meets the merged plane's MSE in one comparison (primitive_detection.cpp:437-476), so -- like
        // the RANSAC costs above -- its ordered sum is bracketed by the tree-order sum of the same addends, and the serial
        // chain over N cells runs only if the bracket cannot decide.
        // The per-cell squared distances are independent: all lanes compute them (cx, cy, cz straight from cell_plane).
        auto mse_addends = [&](bool park) {
            double ps = 0.0;
            constexpr int kMseRounds = 2 * CAPE_CYL_PROJ_ROUNDS;
            for (int i0 = 0; i0 < N; i0 += 64 * kMseRounds)
            {
                double2 w0[kMseRounds], w1[kMseRounds]; // (cx cy) (cz mse) of cell_plane
#pragma unroll
                for (int k = 0; k < kMseRounds; ++k)
                {
                    const int i = i0 + lane + 64 * k;
                    const double2* pl = reinterpret_cast<const double2*>(planeBase + (size_t)c.s_list[i < N ? i : 0] * kPlaneStride);
                    w0[k] = pl[2];
                    w1[k] = pl[3];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < kMseRounds; ++k)
                {
                    const int i = i0 + lane + 64 * k;
                    if (i < N)
                    {
                        double t2 = 0.0;
                        if (c.s_best[i])
                        {
                            const double wx = w0[k].x - P2x, wy = w0[k].y - P2y, wz = w1[k].x - P2z;
                            const double crx = dy * wz - dz * wy;
                            const double cry = dz * wx - dx * wz;
                            const double crz = dx * wy - dy * wx;
                            const double t = sqrt((crx * crx + cry * cry) + crz * crz) / P1P2d - radius;
                            t2 = t * t;
                        }
                        if (park)
                            c.scratch[(size_t)i * kCylStride + 7] = t2; // non-inliers hold +0.0, and x + 0.0 == x for every x this sum can reach
                        ps += t2;
                    }
                }
            }
            return wave_sum_f64_tree(ps);
        };
        const double mseTree = mse_addends(false);
        CAPE_CYL_TICK(18); // MSE: parallel distances

        // ===== cylinder_fitting's per-segment work (primitive_detection.cpp:488-500): merged plane of the inlier cells
        //       (its sums came out of the LLS traversal above, lanes 16-25)
        CAPE_CYL_TICK(20); // merged plane sums, ordered pass
        double S[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            S[k] = readlane_f64(chain, 16 + k);
        const double cnt = readlane_f64(chain, 25);
        PlaneFit f;
        fit_plane(S, (uint32_t)cnt, f);
        if (!f.planar)
            status_count_not_planar(status); // "Plane segment is not planar after merge" (:497); the model selection still runs
        CAPE_CYL_TICK(21); // merged plane fit
        CAPE_CYL_COUNT(24, 1);          // RANSAC rounds (outer while)
        CAPE_CYL_COUNT(25, N);          // cells of the region
        CAPE_CYL_COUNT(26, maxInliers); // inliers removed

        // ===== add_cylinder_to_features (:437-476): model selection on MSE
        bool planeWins;
        {
            const double kInl = (double)maxInliers;
            const double mseLo = (mseTree * (1.0 - CAPE_CYL_EPS)) / kInl, mseHi = (mseTree * (1.0 + CAPE_CYL_EPS)) / kInl;
            if (f.mse < mseLo)
                planeWins = true;
            else if (!(f.mse < mseHi))
                planeWins = false;
            else
            {
                (void)mse_addends(true);
                CAPE_CYL_SYNC();
                const double mse = ordered_sum_lds<kCylStride>(c.scratch + 7, N, __builtin_inf(), lane) / kInl;
                CAPE_CYL_SYNC();
                planeWins = f.mse < mse;
            }
        }
        CAPE_CYL_TICK(19); // MSE: decision (ordered sum only when the bracket is not enough)
        if (planeWins)
        {
            if (nSeg >= c.maxPlanes)
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
                    c.s_lab[c.s_list[i]] = (LabT)nSeg;
        }
        else
        {
            if (nCylLabels >= c.maxCylinders)
            {
                planeOverflow = true;
                return;
            }
            if (lane == 0)
            {
                cape_cylinder* o = &c.cylOut[nCylLabels];
                o->axis[0] = ax; o->axis[1] = ay; o->axis[2] = az;
                o->radius = __builtin_nan(""); // shape_primitives.cpp:17-24 over a copy whose _segmentCount is 0
                o->kept = 0;
                o->region = (uint32_t)nCylFits;
            }
            ++nCylLabels;
            for (int i = lane; i < N; i += 64)
                if (c.s_best[i])
                    c.s_cyl[c.s_list[i]] = (LabT)nCylLabels;
        }
        CAPE_CYL_SYNC();
        CAPE_CYL_TICK(22); // model selection + labels
    }
}

} // namespace cape
