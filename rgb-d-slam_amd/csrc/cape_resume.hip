// Stage B, second pass with cylinders on: ONE WORKGROUP (four wavefronts) PER PARKED FRAME.
//
// A frame reaches this kernel when the plane-only grow pass (cape_grow.hip) met its first cylinder candidate after the
// seed loop had ended: its segments so far, the recorded regions with their plane fits, the cell lists and the label grid
// are parked in p.growState (GrowStateHeader, cape_grow_common.h).  The workgroup finishes the record -> segment
// conversion with Primitive_Detection::cylinder_fitting (reference src/features/primitives/primitive_detection.cpp:413-501)
// = Cylinder_Segment's constructor and run_ransac_loop (cylinder_segment.cpp:35-322); its wave 0 then runs the common
// tail (merge_planes, boundary candidates, records, cylinder morphology: grow_tail()).
//
// Why a workgroup: through round 2 one wavefront did all of this, 442 registers and one wave per SIMD, and a batch's second
// pass lasted as long as its slowest frame (a wall-sized candidate region: ~120 hypotheses over ~700 cells, three ordered
// passes).  What is parallel in cylinder_fitting is parallel over CELLS (projection, the MSAC cost of a hypothesis, the MSE
// addends, labels): 256 lanes take a quarter of the rounds each and the register cache of a lane shrinks from twelve cells
// to three.  What is NOT parallel are the sums whose rounding order is observable (the covariance GEMM, the LLS sums, the
// merged plane's moments): ascending index order, one add per cell on a dependent chain.  There wave 0 is a pure consumer
// -- one LDS operand and one v_add_f64 per cell -- while waves 1-3 fetch, mask and transform the next chunk into the other
// half of a double buffer (the lone wave paid ~35 cycles per cell for doing both; the chain itself is ~10).  Arithmetic is
// the lone-wave kernel's (cape_cylinder.h), expression for expression: both must match the oracle bit for bit, and
// tests/test_gpu_parity.py::test_cylinder_schedules_agree runs the two against each other.
#include <hip/hip_runtime.h>

#include "cape_grow_common.h"

namespace cape {

constexpr int kGroupThreads = 256;
constexpr int kProducers = kGroupThreads - 64;      // waves 1-3 stage the chunks of the ordered passes
constexpr int kCovChunk = kProducers;                // cells per chunk of the covariance passes: one cell per producer lane
constexpr int kLlsChunk = 64;                        // cells per chunk of the combined LLS / merged-plane pass (18 f64 each)
#ifndef CAPE_G_ROUNDS
#define CAPE_G_ROUNDS 6
#endif
constexpr int kCellRounds = CAPE_G_ROUNDS;           // RANSAC: cells a lane keeps in registers (12 VGPRs each)
// ballots a wave may park per hypothesis: a pass over the cells parks kCellRounds words, a region takes ceil(m / (64 * wph * kCellRounds))
// passes -- at most this many words for m <= cells (also on grids of fewer than 64 * kCellRounds cells, where ceil(cells / 64) is too few)
__host__ __device__ constexpr int ransac_ballot_words(int cells) { return kCellRounds * ((cells + 64 * kCellRounds - 1) / (64 * kCellRounds)); }
constexpr int kHypPerWave = 4;                       // RANSAC, regions beyond the register cache: hypotheses a wave scores per pass over the cells
constexpr int kXchDoubles = 64 + 2 * 32;             // exchange area: broadcasts + reductions (64), two halves of RANSAC batch partials (16 sums, 16 counts)
constexpr int kStageDoubles = kLlsChunk * 18;        // doubles per half of the double buffer (>= kCovChunk * 6)
static_assert(kCovChunk * 6 <= kStageDoubles, "a covariance chunk must fit one half of the staging buffer");
static_assert(kStageDoubles * 8 >= 1280 * 2, "the boundary phase's ring list borrows one half of the staging buffer");

#ifdef CAPE_B_PROFILE
#define CAPE_GTICK(k)                                                                                        \
    do                                                                                                       \
    {                                                                                                        \
        const unsigned long long _n = __builtin_amdgcn_s_memtime();                                          \
        if (tid == 0)                                                                                        \
            atomicAdd(&g.s_prof[(k)], _n - _gt);                                                             \
        _gt = _n;                                                                                            \
    } while (0)
#define CAPE_GTICK_INIT() unsigned long long _gt = __builtin_amdgcn_s_memtime()
#define CAPE_GCOUNT(k, v)                                                                                    \
    do                                                                                                       \
    {                                                                                                        \
        if (tid == 0)                                                                                        \
            atomicAdd(&g.s_prof[(k)], (unsigned long long)(v));                                              \
    } while (0)
#else
#define CAPE_GTICK(k)
#define CAPE_GTICK_INIT()
#define CAPE_GCOUNT(k, v)
#endif

struct GroupCtx
{
    const StageBParams* p;
    int tid, lane, wave;
    size_t cellBase;
    int C;
    const unsigned short* list; // activated cells of the region, ascending (= _local2globalMap)
    int total;                  // _cellActivatedCount
    unsigned short* s_ids;      // idsLeft
    unsigned char* s_idmask;    // idsLeftMask
    unsigned long long* s_inl;  // RANSAC: 2 batches x 4 waves x kHypPerWave x ransac_ballot_words(C) ballots, the inliers of the hypotheses being scored
    unsigned char* s_best;      // finalInlierIndexes as flags
    double* scratch;            // [N][kCylStride] projected normals / centroids / n.c / parked exact-path cost
    double* s_stage;            // 2 x kStageDoubles
    double* s_xch;              // kXchDoubles f64 exchange area (partials, broadcasts)
    double* s_seg;
    unsigned char* s_lab;
    unsigned char* s_cyl;
    cape_frame_record* rec;
    unsigned long long* s_prof;
};

// wave 0's side of an ordered pass: acc += buf[i * STRIDE + off] for i = 0 .. cn-1, one rounding per add, in order.  The
// operands of sixteen cells are requested before the sixteen dependent adds of the previous group run.  (A quantity-major
// chunk read with one 16-byte LDS instruction per two cells was built and measured: slower -- 42 k cycles against 22 k for
// the 565-cell LLS pass of a tunnel frame, with or without padded rows -- and it cost ~80 spilled registers.)
template <int STRIDE> __device__ __forceinline__ void chain_consume(const double* buf, int off, int cn, double& acc)
{
    const double* b = buf + off;
    int i = 0;
    constexpr int G = 16;
    if (cn >= G)
    {
        double va[G], vb[G];
#pragma unroll
        for (int u = 0; u < G; ++u)
            va[u] = b[u * STRIDE];
        for (; i + 2 * G <= cn; i += 2 * G)
        {
#pragma unroll
            for (int u = 0; u < G; ++u)
                vb[u] = b[(i + G + u) * STRIDE];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G; ++u)
                acc += va[u];
            __builtin_amdgcn_sched_barrier(0);
            if (i + 3 * G <= cn)
            {
#pragma unroll
                for (int u = 0; u < G; ++u)
                    va[u] = b[(i + 2 * G + u) * STRIDE];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < G; ++u)
                acc += vb[u];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (i + G <= cn)
        {
            // one full group left, already in va (either the first one, or requested by the last trip)
#pragma unroll
            for (int u = 0; u < G; ++u)
                acc += va[u];
            i += G;
        }
    }
    for (; i < cn; ++i)
        acc += b[i * STRIDE];
}

// The same chain over a chunk parked in PAIRS: element i of quantity q lives at buf[((i >> 1) * STRIDE + q) * 2 + (i & 1)], so one
// 16-byte LDS read brings the lane's operands of two consecutive cells -- half the LDS instructions per addend, the adds and
// their order unchanged.  Eight pairs in flight (round 3's attempt kept sixteen 16-byte operands twice over and spilled).
template <int STRIDE> __device__ __forceinline__ void chain_consume_pairs(const double* buf, int off, int cn, double& acc)
{
    const double2* b = reinterpret_cast<const double2*>(buf) + off; // pair j of this lane's quantity: b[j * STRIDE]
    const int pairs = cn >> 1;
    int j = 0;
    constexpr int G = 8;
    if (pairs >= G)
    {
        double2 va[G], vb[G];
#pragma unroll
        for (int u = 0; u < G; ++u)
            va[u] = b[u * STRIDE];
        for (; j + 2 * G <= pairs; j += 2 * G)
        {
#pragma unroll
            for (int u = 0; u < G; ++u)
                vb[u] = b[(j + G + u) * STRIDE];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G; ++u)
            {
                acc += va[u].x;
                acc += va[u].y;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (j + 3 * G <= pairs)
            {
#pragma unroll
                for (int u = 0; u < G; ++u)
                    va[u] = b[(j + 2 * G + u) * STRIDE];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < G; ++u)
            {
                acc += vb[u].x;
                acc += vb[u].y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j + G <= pairs)
        {
#pragma unroll
            for (int u = 0; u < G; ++u)
            {
                acc += va[u].x;
                acc += va[u].y;
            }
            j += G;
        }
    }
    for (; j < pairs; ++j)
    {
        const double2 v = b[j * STRIDE];
        acc += v.x;
        acc += v.y;
    }
    if (cn & 1)
        acc += b[pairs * STRIDE].x;
}

// returns with nSeg / nCylLabels / rngPos / status updated; nCylFits is incremented by the caller.  Every branch that
// decides what the workgroup does next is taken on values all 256 lanes hold identically.
template <typename MaskT>
__device__ inline void cylinder_fitting_group(const GroupCtx& g, int& nSeg, int& nCylLabels, int nCylFits, int& rngPos, uint32_t& status,
                                              bool& planeOverflow, int maxPlanes)
{
    const StageBParams& p = *g.p;
    const int tid = g.tid, lane = g.lane, wave = g.wave;
    const int N = g.total;
    const double* planeBase = p.cell_plane + g.cellBase * kPlaneStride;
    const double* sumsBase = p.cell_sums + g.cellBase * kSumStride;
    CAPE_GTICK_INIT();

    // ---- cov = (M * M^T) / (cols - 1), M = [normals, -normals] (cylinder_segment.cpp:47-89): ascending column order.  Lane e
    //      of wave 0 accumulates entry e of the lower triangle, (0,0) (1,0) (1,1) (2,0) (2,1) (2,2); the six products of a
    //      cell are formed by the producer lane that fetched its normal ((-a)*(-b) == a*b: the second half adds the same
    //      products again).
    {
        const int chunksPerHalf = (N + kCovChunk - 1) / kCovChunk;
        const int K = 2 * chunksPerHalf;
        // producers: the normal of chunk k + 2 is on its way while chunk k + 1 is parked and chunk k is summed
        double2 r0 = make_double2(0, 0), r1 = r0; // (nx ny) (nz d) of the producer lane's cell of the chunk in flight
        auto request = [&](int k) {
            if (wave == 0 || k >= K)
                return;
            const int i = (k % chunksPerHalf) * kCovChunk + (tid - 64);
            const double2* pl = reinterpret_cast<const double2*>(planeBase + (size_t)g.list[i < N ? i : N - 1] * kPlaneStride);
            r0 = pl[0];
            r1 = pl[1];
        };
        auto park = [&](int k) {
            if (wave == 0 || k >= K)
                return;
            // parked in pairs of cells (chain_consume_pairs): quantity e of cell c at ((c >> 1) * 6 + e) * 2 + (c & 1)
            const int c = tid - 64;
            double* rec = g.s_stage + (k & 1) * kStageDoubles + (size_t)(c >> 1) * 12 + (c & 1);
            rec[0] = r0.x * r0.x;
            rec[2] = r0.y * r0.x;
            rec[4] = r0.y * r0.y;
            rec[6] = r1.x * r0.x;
            rec[8] = r1.x * r0.y;
            rec[10] = r1.x * r1.x;
        };
        const int e = lane < 6 ? lane : 0;
        double acc = 0.0;
        request(0);
        park(0);
        request(1);
        __syncthreads();
        for (int k = 0; k < K; ++k)
        {
            park(k + 1);
            request(k + 2);
            if (wave == 0)
            {
                const int c0 = (k % chunksPerHalf) * kCovChunk;
                const int cn = (N - c0 < kCovChunk) ? (N - c0) : kCovChunk;
                chain_consume_pairs<6>(g.s_stage + (k & 1) * kStageDoubles, e, cn, acc);
            }
            __syncthreads();
        }
        if (wave == 0 && lane < 6)
            g.s_xch[lane] = acc / (double)(2 * N - 1);
        __syncthreads();
    }
    const double m00 = g.s_xch[0], m10 = g.s_xch[1], m11 = g.s_xch[2], m20 = g.s_xch[3], m21 = g.s_xch[4], m22 = g.s_xch[5];
    CAPE_GTICK(28); // covariance passes
    Eig3 eg;
    self_adjoint_eigen3(m00, m10, m11, m20, m21, m22, eg);
    const double score = eg.val[2] / eg.val[0];
    CAPE_GTICK(12); // eigen
    if (score < (double)75.0f) // cylinderRansacMinimumScore, checkpoint 1 (:95-102)
    {
        __syncthreads(); // s_xch is rewritten by the next region
        return;
    }
    const double ax = eg.q[0][0], ay = eg.q[1][0], az = eg.q[2][0];

    // ---- projection on the plane orthogonal to the axis (:107-125), one cell per lane and round
    // (the next cell's plane is requested before this one's arithmetic -- two divisions and a square root -- so that a region of
    //  nine rounds per lane pays one exposed memory round trip, not nine)
    double2 nq0 = make_double2(0, 0), nq1 = nq0, nq2 = nq0, nq3 = nq0;
    if (tid < N)
    {
        const double2* pl = reinterpret_cast<const double2*>(planeBase + (size_t)g.list[tid] * kPlaneStride);
        nq0 = pl[0], nq1 = pl[1], nq2 = pl[2], nq3 = pl[3];
    }
    for (int j = tid; j < N; j += kGroupThreads)
    {
        const double2 q0 = nq0, q1 = nq1, q2 = nq2, q3 = nq3;
        if (j + kGroupThreads < N)
        {
            const double2* pl = reinterpret_cast<const double2*>(planeBase + (size_t)g.list[j + kGroupThreads] * kPlaneStride);
            nq0 = pl[0], nq1 = pl[1], nq2 = pl[2], nq3 = pl[3];
        }
        const double nx = q0.x, ny = q0.y, nz = q1.x, cx = q2.x, cy = q2.y, cz = q3.x;
        const double cdt = dot3(ax, ay, az, cx, cy, cz);
        const double ndt = dot3(ax, ay, az, nx, ny, nz);
        const double px = nx - ndt * ax, py = ny - ndt * ay, pz = nz - ndt * az;
        const double nrm = sqrt((px * px + py * py) + pz * pz);
        const double o0 = px / nrm, o1 = py / nrm, o2 = pz / nrm;
        const double o3 = cx - cdt * ax, o4 = cy - cdt * ay, o5 = cz - cdt * az;
        double2* o = reinterpret_cast<double2*>(g.scratch + (size_t)j * kCylStride);
        o[0] = make_double2(o0, o1);
        o[1] = make_double2(o2, o3);
        o[2] = make_double2(o4, o5);
        // the LLS term b += n.dot(c) (cylinder_segment.cpp:171), ready for the ordered pass
        o[3] = make_double2((o0 * o3 + o1 * o4) + o2 * o5, 0.0);
        g.s_ids[j] = (unsigned short)j;
        g.s_idmask[j] = 1;
    }
    __syncthreads();
    CAPE_GTICK(13); // projection

    int planeSegmentsLeft = N;
    int idsLeftCount = N;
    const float maxSqrtDistF = 0.04f; // cylinderRansacSqrtMaxDistance
    const double maxSqrtDist = (double)maxSqrtDistF;
    int xchPar = 0;      // which half of the partial-sum exchange the next reduction uses
    int xchParBatch = 0; // ... and of the RANSAC batches' exchange (s_xch[64 ..))

    // (psum, count) over the workgroup: wave partials through LDS, combined in a fixed order by every lane
    auto group_reduce = [&](double ps, int cnt, double& psAll, int& cntAll) {
        ps = wave_sum_f64_tree(ps);
        cnt = wave_sum_i32(cnt);
        double* x = g.s_xch + 8 + xchPar * 8;
        if (lane == 0)
        {
            x[wave] = ps;
            x[4 + wave] = __longlong_as_double((long long)cnt);
        }
        __syncthreads();
        psAll = (x[0] + x[1]) + (x[2] + x[3]);
        cntAll = (int)__double_as_longlong(x[4]) + (int)__double_as_longlong(x[5]) + (int)__double_as_longlong(x[6]) +
                 (int)__double_as_longlong(x[7]);
        xchPar ^= 1; // the next reduction writes the other half: no lane can still be reading this one by then
    };

    // ---- sequential RANSAC (:146-224)
    while (planeSegmentsLeft > p.minCellActivated && (double)planeSegmentsLeft > 0.1 * (double)N)
    {
        // ===== run_ransac_loop (:227-322)
        int bestCount = 0;
        if (idsLeftCount >= 3)
        {
            const int m = idsLeftCount;
            const unsigned inliersAccepted = (unsigned)floor(0.9 * (double)m);
            // [minLo, minHi] brackets the ORDERED sum of the best hypothesis' MSAC costs (see cape_cylinder.h: the reference
            // only ever compares against it, and the tree-order sum of the same addends is within CAPE_CYL_EPS of it)
            double minLo = (double)(maxSqrtDistF * (float)m), minHi = minLo;
            int prevBestCount = 0;
            for (int j = tid; j < N; j += kGroupThreads)
                g.s_best[j] = 0;
            // How the four waves share a RANSAC loop (round 5).  Scoring a hypothesis does not depend on which hypotheses won before
            // it -- only the bookkeeping does (best so far, early stop) -- so hypotheses are scored in BATCHES between two barriers
            // and the reference's sequential bookkeeping is replayed over the batch by every lane on identical numbers; what is
            // scored past an early stop is wasted work, never observable.  Through round 4 every wave scored EIGHT hypotheses on
            // its quarter of the cells: eight wave-wide reductions per wave and batch, a 4-way combine per hypothesis, ~100
            // instructions of replay per hypothesis -- 1.9 k ticks per hypothesis, no faster than the lone wave.  Now a hypothesis
            // belongs to ONE wave (regions of up to 384 cells: four hypotheses per batch) or to a PAIR of waves (larger regions:
            // two per batch, each wave every other round of 64 cells): one reduction per wave and batch, the inlier sets travel as
            // wave ballots in LDS (no per-lane bit bookkeeping, no byte array of the streamed path), the replay carries the index
            // of the best hypothesis instead of its five parameters, and the winner's flags are written once per batch.
            const int wph = m > 64 * kCellRounds ? 2 : 1;        // waves per hypothesis
            const int sub = wave & (wph - 1), hloc = wave >> (wph - 1);
            const int chunkCells = 64 * wph * kCellRounds;       // cells of one pass over the register cache
            const bool cached = m <= chunkCells;
            // a region beyond the register cache (the 64 x 48 grid) is STREAMED: every pass over its cells serves kHypPerWave
            // hypotheses per wave, so that the loads are paid once per eight hypotheses (two pairs of waves), not once per two
            const int hpw = cached ? 1 : kHypPerWave;
            const int hpb = (4 / wph) * hpw;                     // hypotheses per batch: hypothesis h of a batch belongs to wave (pair) h / hpw
            const int nInlWords = ransac_ballot_words(g.C);      // ballots a wave may park per hypothesis
            double2 cqa[kCellRounds], cqb[kCellRounds], cqc[kCellRounds];
            auto fetch_cells = [&](int j0) {
#pragma unroll
                for (int k = 0; k < kCellRounds; ++k)
                {
                    const int jj = j0 + sub * 64 + lane + 64 * wph * k;
                    const int id = g.s_ids[jj < m ? jj : 0];
                    const double2* t_ = reinterpret_cast<const double2*>(g.scratch + (size_t)id * kCylStride);
                    cqa[k] = t_[0];
                    cqb[k] = t_[1];
                    cqc[k] = t_[2];
                }
            };
            if (cached)
                fetch_cells(0);
            // every hypothesis of this loop, lane `it` of each wave builds hypothesis `it` (cape_cylinder.h)
            const int rngBase = rngPos;
            if (p.ransacMaxIterations > 64)
                status |= CAPE_FRAME_RNG_EXHAUSTED;
            double hypR, hypInvR2, hypCx, hypCy, hypCz;
            {
                const int itL = lane < p.ransacMaxIterations ? lane : p.ransacMaxIterations - 1;
                const int last = p.rngCount - 1;
                double2 ta[3], tb[3], tc[3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
                {
                    const int di = rngBase + 3 * itL + q;
                    const double u = di < p.rngCount ? p.rngTable[di < last ? di : last] : 0.0;
                    const int cell = (int)g.s_ids[(unsigned)floor(u * (double)(unsigned)m)];
                    const double2* t_ = reinterpret_cast<const double2*>(g.scratch + (size_t)cell * kCylStride);
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
            __syncthreads(); // s_best is clear before the first winner flags its inliers

            // the hypothesis the distance code evaluates (uniform within a wave)
            double radius = 0.0, invR2 = 0.0, ctx = 0.0, cty = 0.0, ctz = 0.0;
            auto take_hypothesis = [&](int it) {
                const int itLane = (it < p.ransacMaxIterations ? it : p.ransacMaxIterations - 1) & 63;
                radius = readlane_f64(hypR, itLane), invR2 = readlane_f64(hypInvR2, itLane);
                ctx = readlane_f64(hypCx, itLane), cty = readlane_f64(hypCy, itLane), ctz = readlane_f64(hypCz, itLane);
            };
            auto msac = [&](const double2& qa, const double2& qb, const double2& qc, bool& inl) {
                const double vx = (qb.y - radius * qa.x) - ctx;
                const double vy = (qc.x - radius * qa.y) - cty;
                const double vz = (qc.y - radius * qb.x) - ctz;
                const double distance = ((vx * vx + vy * vy) + vz * vz) * invR2;
                inl = distance < maxSqrtDist;
                return inl ? distance : maxSqrtDist;
            };
            // ordered sum of the MSAC costs of hypothesis `it` (the never-taken exact path; a -DCAPE_CYL_EPS twin build takes it): the
            // costs are parked in the free eighth double of the scratch records, position by position, and every wave walks them
            // itself (same bits in all four)
            auto ordered_cost = [&](int it, double limit) {
                take_hypothesis(it);
                for (int jj = tid; jj < m; jj += kGroupThreads)
                {
                    const double2* t_ = reinterpret_cast<const double2*>(g.scratch + (size_t)g.s_ids[jj] * kCylStride);
                    bool inl_;
                    g.scratch[(size_t)jj * kCylStride + 7] = msac(t_[0], t_[1], t_[2], inl_);
                }
                __syncthreads();
                const double d = ordered_sum_lds<kCylStride>(g.scratch + 7, m, limit, lane);
                __syncthreads(); // the parked costs are rewritten by the next call
                return d;
            };
            bool stop = false;
            int bestIt = -1; // the hypothesis [minLo, minHi] belongs to
            for (int it0 = 0; it0 < p.ransacMaxIterations && !stop; it0 += hpb)
            {
                // ---- this wave's hypotheses of the batch over this wave's rounds of cells
                double ps[kHypPerWave];
                int cnt[kHypPerWave];
#pragma unroll
                for (int q = 0; q < kHypPerWave; ++q)
                    ps[q] = 0.0, cnt[q] = 0;
                unsigned long long* inl = g.s_inl + ((size_t)xchParBatch * 4 + wave) * kHypPerWave * nInlWords;
                for (int j0 = 0, w0 = 0; j0 < m; j0 += chunkCells, w0 += kCellRounds)
                {
                    if (!cached)
                        fetch_cells(j0);
#pragma unroll
                    for (int q = 0; q < kHypPerWave; ++q)
                    {
                        if (q >= hpw)
                            break;
                        take_hypothesis(it0 + hloc * hpw + q);
                        // the rounds are independent chains of ~8 dependent f64 operations each: all of them first (the scheduler
                        // interleaves them), the ballots parked by ONE masked block behind -- a masked store inside a round ends its
                        // basic block and the rounds then run one after the other at the latency of a lone wave's dependent issue
                        double dk[kCellRounds];
                        unsigned long long bal[kCellRounds];
#pragma unroll
                        for (int k = 0; k < kCellRounds; ++k)
                        {
                            bool inl_;
                            const double d_ = msac(cqa[k], cqb[k], cqc[k], inl_);
                            const bool in_ = j0 + sub * 64 + lane + 64 * wph * k < m;
                            dk[k] = in_ ? d_ : 0.0;
                            bal[k] = __ballot(in_ && inl_);
                        }
                        double pk = 0.0; // (any order: the comparison is made on a bracket that covers every order of these addends)
#pragma unroll
                        for (int k = 0; k + 1 < kCellRounds; k += 2)
                            pk += dk[k] + dk[k + 1];
                        if (kCellRounds & 1)
                            pk += dk[kCellRounds - 1];
                        ps[q] += pk;
#pragma unroll
                        for (int k = 0; k < kCellRounds; ++k)
                            cnt[q] += __popcll(bal[k]);
                        if (lane == 0)
                        {
#pragma unroll
                            for (int k = 0; k < kCellRounds; ++k)
                                inl[(size_t)q * nInlWords + w0 + k] = bal[k];
                        }
                    }
                }
                CAPE_GTICK(1); // RANSAC: scoring
                double* x = g.s_xch + 64 + xchParBatch * 32; // [0, 16): sums, [16, 32): counts; entry (hypothesis of the batch) * wph + wave of the pair
#pragma unroll
                for (int q = 0; q < kHypPerWave; ++q)
                {
                    if (q >= hpw)
                        break;
                    const double pw = wave_sum_f64_tree(ps[q]);
                    if (lane == 0)
                    {
                        x[(hloc * hpw + q) * wph + sub] = pw;
                        x[16 + (hloc * hpw + q) * wph + sub] = __longlong_as_double((long long)cnt[q]);
                    }
                }
                CAPE_GTICK(2); // RANSAC: wave reduction
                __syncthreads();
                CAPE_GTICK(3); // RANSAC: barrier
                // ---- the reference's bookkeeping, hypothesis by hypothesis (the batch's words in ONE LDS round trip)
                double psH[8];
                int cntH[8];
#pragma unroll
                for (int h = 0; h < 8; ++h)
                {
                    // (entries beyond hpb * wph hold what an earlier batch left: read, never used)
                    const int e0 = (h * wph) & 15, e1 = (h * wph + 1) & 15;
                    const double s0 = x[e0], s1 = x[e1];
                    const int c0 = (int)__double_as_longlong(x[16 + e0]), c1 = (int)__double_as_longlong(x[16 + e1]);
                    psH[h] = wph == 2 ? s0 + s1 : s0;
                    cntH[h] = wph == 2 ? c0 + c1 : c0;
                }
                int winner = -1;
#pragma unroll
                for (int h = 0; h < 8; ++h)
                {
                    const int it = it0 + h;
                    if (h >= hpb || it >= p.ransacMaxIterations)
                        break;
                    if (rngBase + 3 * it + 2 >= p.rngCount)
                        status |= CAPE_FRAME_RNG_EXHAUSTED;
                    rngPos = rngBase + 3 * (it + 1);
                    const double psAll = psH[h];
                    const int curCount = cntH[h];
                    double lo = psAll * (1.0 - CAPE_CYL_EPS), hi = psAll * (1.0 + CAPE_CYL_EPS);
                    bool wins;
                    if (lo >= minHi)
                        wins = false; // ordered sum >= lo >= the best's
                    else if (hi < minLo)
                        wins = true; // ordered sum <= hi < the best's
                    else
                    {
                        // cannot be told apart in any-order sums (never seen with the default bound outside the test build)
                        if (minLo != minHi)
                            minLo = minHi = ordered_cost(bestIt, __builtin_inf());
                        lo = hi = ordered_cost(it, minHi); // stops once it reaches minHi: >= the best's, loses
                        wins = lo < minHi;
                    }
                    if (wins)
                    {
                        minLo = lo, minHi = hi;
                        bestIt = it;
                        winner = h;
                        prevBestCount = bestCount; // inlierIndexes now holds the previous best (swap)
                        bestCount = curCount;
                        // early-stop quirk (:308-312): tests the swapped-out vector
                        stop = (unsigned)prevBestCount > inliersAccepted;
                    }
                    CAPE_GCOUNT(27, 1); // hypotheses evaluated
                    if (stop)
                        break;
                }
                if (winner >= 0)
                {
                    // finalInlierIndexes = the inliers of the batch's last winner: its wave(s) parked them as ballots, position by
                    // position of idsLeft (chunk, wave of the pair, round, lane)
                    const int wWave = (winner / hpw) * wph, wq = winner % hpw; // the (first) wave that scored it, and as which of its hypotheses
                    const unsigned long long* wi = g.s_inl + (((size_t)xchParBatch * 4 + wWave) * kHypPerWave + wq) * nInlWords;
                    for (int jj = tid; jj < m; jj += kGroupThreads)
                    {
                        const int c = cached ? 0 : jj / chunkCells;
                        const int r = jj - c * chunkCells, q = r >> 6;
                        const unsigned long long word = wi[(size_t)(q & (wph - 1)) * kHypPerWave * nInlWords + c * kCellRounds + (q >> (wph - 1))];
                        g.s_best[g.s_ids[jj]] = (unsigned char)((word >> (r & 63)) & 1ull);
                    }
                }
                xchParBatch ^= 1; // the next batch parks in the other half: nobody can still be reading this one when it is rewritten
                CAPE_GTICK(4); // RANSAC: replay
            }
            __syncthreads(); // the winner's flags are visible to every lane
        }
        CAPE_GTICK(14); // RANSAC iterations
        // checkpoint 2
        if (bestCount < 6)
            break;
        const int maxInliers = bestCount;

        // ===== LLS over all inliers, ascending i (:157-186), and -- on the same chain -- the sums of cylinder_fitting's merged
        // plane of the inlier cells (primitive_detection.cpp:488-500): an 18-double record per cell (8 of the projected
        // scratch, 10 of cell_sums), +0.0 throughout for a non-inlier (which leaves the running sums unchanged bit for bit).
        //   wave 0: lanes 0-2 sumN, 3-5 sumC, 6 b (the precomputed n.c product);  lanes 16-25 the merged plane's ten sums.
        {
            const int K = (N + kLlsChunk - 1) / kLlsChunk;
            constexpr int kPer = (kLlsChunk * 9 + kProducers - 1) / kProducers; // 16-byte pieces a producer lane moves per chunk
            double2 rq[kPer];
            bool rkeep[kPer]; // the inlier flag of the piece's cell: applied when the piece is PARKED -- a select right behind the load
                              // makes the producer wait for its data in front of the barrier, i.e. every chunk costs a memory round trip
                              // (70 k ticks of "wait for the producers" per 1280x960 tunnel frame, profiles/r05_cylinder_group.txt)
            auto request = [&](int k) {
                if (wave == 0 || k >= K)
                    return;
                const int c0 = k * kLlsChunk;
#pragma unroll
                for (int q = 0; q < kPer; ++q)
                {
                    const int piece = (tid - 64) + kProducers * q;
                    const int ce = piece / 9, sub = piece - ce * 9;
                    const int e = (c0 + ce) < N ? (c0 + ce) : N - 1; // clamped: unconditional loads
                    const double2 v = sub < 4 ? *reinterpret_cast<const double2*>(g.scratch + (size_t)e * kCylStride + 2 * sub)
                                              : *reinterpret_cast<const double2*>(sumsBase + (size_t)g.list[e] * kSumStride + 2 * (sub - 4));
                    rq[q] = v;
                    rkeep[q] = g.s_best[e] != 0;
                }
            };
            auto park = [&](int k) {
                if (wave == 0 || k >= K)
                    return;
                double* buf = g.s_stage + (k & 1) * kStageDoubles;
#pragma unroll
                for (int q = 0; q < kPer; ++q)
                {
                    const int piece = (tid - 64) + kProducers * q;
                    const int ce = piece / 9, sub = piece - ce * 9;
                    if (piece < kLlsChunk * 9)
                        *reinterpret_cast<double2*>(buf + ce * 18 + 2 * sub) = rkeep[q] ? rq[q] : make_double2(0.0, 0.0);
                }
            };
            const int slot = lane < 7 ? lane : ((lane >= 16 && lane < 26) ? 8 + (lane - 16) : 0); // the lane's double of a record
            double chain = 0.0;
            request(0);
            park(0);
            request(1);
            __syncthreads();
            for (int k = 0; k < K; ++k)
            {
                park(k + 1);
                request(k + 2);
                if (wave == 0)
                {
                    const int c0 = k * kLlsChunk;
                    const int cn = (N - c0 < kLlsChunk) ? (N - c0) : kLlsChunk;
                    chain_consume<18>(g.s_stage + (k & 1) * kStageDoubles, slot, cn, chain);
                }
                CAPE_GTICK(5); // LLS: wave 0 consumes
                __syncthreads();
                CAPE_GTICK(6); // LLS: wave 0 waits for the producers
            }
            if (wave == 0 && (lane < 7 || (lane >= 16 && lane < 26)))
                g.s_xch[24 + slot] = chain; // 24..30 LLS, 32..41 merged plane
            __syncthreads();
        }
        CAPE_GTICK(16); // LLS ordered pass
        const double sNx = g.s_xch[24], sNy = g.s_xch[25], sNz = g.s_xch[26];
        const double sCx = g.s_xch[27], sCy = g.s_xch[28], sCz = g.s_xch[29];
        double b = g.s_xch[30];
        double S[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            S[k] = g.s_xch[32 + k];
        const double cnt = g.s_xch[41];
        // remove the inliers from the remaining ids (:161-179): 256 cells per step, wave offsets through the exchange area
        {
            int newCount = 0;
            for (int base = 0; base < N; base += kGroupThreads)
            {
                const int i = base + tid;
                const bool inl = i < N && g.s_best[i];
                const bool keep = i < N && !inl && g.s_idmask[i];
                const unsigned long long kb = __ballot(keep);
                double* x = g.s_xch + 8 + xchPar * 8;
                if (lane == 0)
                    x[wave] = __longlong_as_double((long long)__popcll(kb));
                __syncthreads();
                const int w0 = (int)__double_as_longlong(x[0]), w1 = (int)__double_as_longlong(x[1]);
                const int w2 = (int)__double_as_longlong(x[2]), w3 = (int)__double_as_longlong(x[3]);
                xchPar ^= 1;
                const int before = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
                if (inl)
                    g.s_idmask[i] = 0;
                if (keep)
                    g.s_ids[newCount + before + __popcll(kb & ((1ull << lane) - 1ull))] = (unsigned short)i;
                newCount += (w0 + w1) + (w2 + w3);
            }
            idsLeftCount = newCount;
            planeSegmentsLeft -= maxInliers;
        }
        __syncthreads();
        CAPE_GTICK(17); // idsLeft compaction

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

        // MSE of the inliers' (unprojected) centroids to the axis line (:198-218); bracketed like the RANSAC costs
        const double P2x = ctx + ax, P2y = cty + ay, P2z = ctz + az;
        const double dx = P2x - ctx, dy = P2y - cty, dz = P2z - ctz;
        const double P1P2d = sqrt((dx * dx + dy * dy) + dz * dz);
        auto mse_addends = [&](bool park) {
            double ps = 0.0;
            for (int i = tid; i < N; i += kGroupThreads)
            {
                double t2 = 0.0;
                if (g.s_best[i])
                {
                    const double2* pl = reinterpret_cast<const double2*>(planeBase + (size_t)g.list[i] * kPlaneStride);
                    const double2 w0 = pl[2], w1 = pl[3]; // (cx cy) (cz mse)
                    const double wx = w0.x - P2x, wy = w0.y - P2y, wz = w1.x - P2z;
                    const double crx = dy * wz - dz * wy;
                    const double cry = dz * wx - dx * wz;
                    const double crz = dx * wy - dy * wx;
                    const double t = sqrt((crx * crx + cry * cry) + crz * crz) / P1P2d - radius;
                    t2 = t * t;
                }
                if (park)
                    g.scratch[(size_t)i * kCylStride + 7] = t2; // non-inliers hold +0.0
                ps += t2;
            }
            double all;
            int dummy;
            group_reduce(ps, 0, all, dummy);
            return all;
        };
        const double mseTree = mse_addends(false);
        CAPE_GTICK(18); // MSE: parallel distances

        // ===== cylinder_fitting's per-segment work (primitive_detection.cpp:488-500): merged plane of the inlier cells
        PlaneFit f;
        fit_plane(S, (uint32_t)cnt, f);
        if (!f.planar)
            status_count_not_planar(status); // "Plane segment is not planar after merge" (:497); the model selection still runs
        CAPE_GTICK(21); // merged plane fit
        CAPE_GCOUNT(24, 1);          // RANSAC rounds (outer while)
        CAPE_GCOUNT(25, N);          // cells of the region
        CAPE_GCOUNT(26, maxInliers); // inliers removed

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
                __syncthreads();
                const double mse = ordered_sum_lds<kCylStride>(g.scratch + 7, N, __builtin_inf(), lane) / kInl;
                __syncthreads();
                planeWins = f.mse < mse;
            }
        }
        CAPE_GTICK(19); // MSE: decision
        if (planeWins)
        {
            if (nSeg >= maxPlanes)
            {
                planeOverflow = true;
                return;
            }
            if (tid == 0)
            {
                double* o = g.s_seg + nSeg * kSegDoubles;
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
            for (int i = tid; i < N; i += kGroupThreads)
                if (g.s_best[i])
                    g.s_lab[g.list[i]] = (unsigned char)nSeg;
        }
        else
        {
            if (nCylLabels >= CAPE_MAX_CYLINDERS)
            {
                planeOverflow = true; // out of cylinder slots: like a full segment list, the frame goes to the next larger instance
                return;
            }
            if (tid == 0)
            {
                cape_cylinder* o = &g.rec->cylinders[nCylLabels];
                o->axis[0] = ax; o->axis[1] = ay; o->axis[2] = az;
                o->radius = __builtin_nan(""); // shape_primitives.cpp:17-24 over a copy whose _segmentCount is 0
                o->kept = 0;
                o->region = (uint32_t)nCylFits;
            }
            ++nCylLabels;
            for (int i = tid; i < N; i += kGroupThreads)
                if (g.s_best[i])
                    g.s_cyl[g.list[i]] = (unsigned char)nCylLabels;
        }
        __syncthreads();
        CAPE_GTICK(22); // model selection + labels
    }
}

size_t resume_group_lds_bytes(int cells)
{
    size_t b = 0;
    b += (size_t)(kFastPlanes + 1) * kSegDoubles * 8;   // s_seg
    b += (size_t)kPendResume * kSegDoubles * 8;         // s_pend
    b += (size_t)(kFastPlanes + 1) * 8;                 // s_adj
    b = (b + 15) & ~(size_t)15;
    b += kXchDoubles * 8;                               // s_xch
    b += (size_t)2 * kStageDoubles * 8;                 // s_stage (two halves)
    b += ((size_t)cells + 4) * 2;                       // s_list (+ pad entry)
    b = (b + 7) & ~(size_t)7;
    b += (size_t)cells + kFastPlanes + (size_t)cells;   // s_lab, s_mlab, s_cyl
    b = (b + 3) & ~(size_t)3;
    b += (size_t)cells * 4;                             // s_ids, s_idmask, s_best (the boundary phase's s_zc afterwards)
    b = (b + 7) & ~(size_t)7;
    b += (size_t)2 * 4 * kHypPerWave * (size_t)ransac_ballot_words(cells) * 8; // s_inl
#ifdef CAPE_B_PROFILE
    b = ((b + 15) & ~(size_t)15) + 8 * kProfileSlots;
#endif
    return (b + 15) & ~(size_t)15;
}

template <typename MaskT> __global__ __launch_bounds__(kGroupThreads, 2) void cape_resume_group_kernel(StageBParams p, int ldsBytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frame = resume_pick(p, (int)blockIdx.x); // the grid is sized for the worst case; the k-th workgroup takes the k-th parked frame
    if (frame < 0)
        return;
    const unsigned long long tPhase = p.phaseTicks ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull; // (see grow_frame_wave)
    constexpr int MAXP = kFastPlanes;
    const int C = p.cells;
    const size_t cellBase = (size_t)frame * C;

    // ---- LDS carve (resume_group_lds_bytes mirrors it)
    double* s_seg = reinterpret_cast<double*>(smem);
    double* s_pend = s_seg + (MAXP + 1) * kSegDoubles;
    unsigned long long* s_adj = reinterpret_cast<unsigned long long*>(s_pend + kPendResume * kSegDoubles);
    double* s_xch = reinterpret_cast<double*>(smem + (((size_t)(reinterpret_cast<unsigned char*>(s_adj + MAXP + 1) - smem) + 15) & ~(size_t)15));
    double* s_stage = s_xch + kXchDoubles;
    unsigned short* s_list = reinterpret_cast<unsigned short*>(s_stage + 2 * kStageDoubles);
    unsigned char* s_lab = smem + (((size_t)(reinterpret_cast<unsigned char*>(s_list + C + 4) - smem) + 7) & ~(size_t)7);
    unsigned char* s_mlab = s_lab + C;
    unsigned char* s_cyl = s_mlab + MAXP;
    unsigned short* s_ids = reinterpret_cast<unsigned short*>(smem + (((size_t)(s_cyl + C - smem) + 3) & ~(size_t)3));
    unsigned char* s_idmask = reinterpret_cast<unsigned char*>(s_ids + C);
    unsigned char* s_best = s_idmask + C;
    unsigned long long* s_inl = reinterpret_cast<unsigned long long*>(smem + (((size_t)(s_best + C - smem) + 7) & ~(size_t)7));
    float* s_zc = reinterpret_cast<float*>(s_ids);
#ifdef CAPE_B_PROFILE
    unsigned long long* s_prof = reinterpret_cast<unsigned long long*>(smem + ldsBytes - 8 * kProfileSlots);
    if (tid < kProfileSlots)
        s_prof[tid] = 0ull;
#else
    unsigned long long* s_prof = nullptr;
#endif

    // ---- pick the frame up where the plane-only pass parked it
    const unsigned char* st = p.growState + (size_t)frame * p.growStateStride;
    const GrowStateHeader hd = *reinterpret_cast<const GrowStateHeader*>(st);
    const double* gseg = reinterpret_cast<const double*>(st + grow_state_seg_off());
    const unsigned long long* gadj = reinterpret_cast<const unsigned long long*>(st + grow_state_adj_off());
    const unsigned short* glist = reinterpret_cast<const unsigned short*>(st + grow_state_list_off());
    const unsigned char* glab = st + grow_state_lab_off(C);
    int nSeg = hd.nSeg;
    const int nSeeds = hd.nSeeds, nPlanar = hd.nPlanar;
    uint32_t status = status_resume(hd.status, tid == 0);
    const int nRec = hd.pendCount - hd.pendFrom;
    for (int i = tid; i < nSeg * kSegDoubles; i += kGroupThreads)
        s_seg[i] = gseg[i];
    for (int i = tid; i < nRec * kSegDoubles; i += kGroupThreads)
        s_pend[i] = gseg[(hd.pendBaseSlot + hd.pendFrom) * kSegDoubles + i];
    for (int i = tid; i < MAXP + 1; i += kGroupThreads)
    {
        s_adj[i] = i < nRec ? gadj[hd.pendFrom + i] : 0ull;
        if (i < MAXP)
            s_mlab[i] = (unsigned char)i;
    }
    for (int i = tid; i < C + 4; i += kGroupThreads)
        s_list[i] = glist[i];
    for (int i = tid; i < C; i += kGroupThreads)
    {
        s_lab[i] = glab[i];
        s_cyl[i] = 0;
    }
    __syncthreads();

    int nCylLabels = 0, nCylFits = 0, rngPos = 0;
    bool handedOn = false;
    for (int j = 0; j < nRec; ++j)
    {
        SegRec ns;
        seg_load(s_pend + j * kSegDoubles, ns);
        const unsigned long long meta = s_adj[j];
        const int roff = (int)(unsigned)meta, total = (int)(meta >> 32);
        if (ns.planar == 0.0)
        {
            status_count_not_planar(status); // "Plane segment is not planar after merge" (:374)
            continue;
        }
        bool overflow = false;
        if (ns.score > 100)
        {
            if (nSeg >= MAXP)
                overflow = true;
            else
            {
                __syncthreads(); // every lane has read record j before slot nSeg (which may be its neighbour) is written
                if (tid == 0)
                    seg_store(s_seg + nSeg * kSegDoubles, ns);
                ++nSeg;
                for (int i = tid; i < total; i += kGroupThreads)
                    s_lab[s_list[1 + roff + i]] = (unsigned char)nSeg;
                __syncthreads();
            }
        }
        else if (total > 5)
        {
            GroupCtx g;
            g.p = &p;
            g.tid = tid;
            g.lane = lane;
            g.wave = wave;
            g.cellBase = cellBase;
            g.C = C;
            g.list = s_list + 1 + roff;
            g.total = total;
            g.s_ids = s_ids;
            g.s_idmask = s_idmask;
            g.s_inl = s_inl;
            g.s_best = s_best;
            g.scratch = p.cylScratch + cellBase * kCylStride;
            g.s_stage = s_stage;
            g.s_xch = s_xch;
            g.s_seg = s_seg;
            g.s_lab = s_lab;
            g.s_cyl = s_cyl;
            g.rec = p.records + frame;
            g.s_prof = s_prof;
            cylinder_fitting_group<MaskT>(g, nSeg, nCylLabels, nCylFits, rngPos, status, overflow, MAXP);
            ++nCylFits;
            __syncthreads();
        }
        if (overflow)
        {
            // out of LDS segment slots: the 64-segment instance of the grow kernel redoes this frame from the start
            if (p.redoList)
            {
                if (tid == 0)
                {
                    p.redoList[1 + atomicAdd(&p.redoList[0], 1u)] = (uint32_t)frame;
                    if (p.phaseTicks)
                        atomicAdd(&p.phaseTicks[(size_t)frame * 4], (unsigned long long)__builtin_amdgcn_s_memtime() - tPhase);
                }
                handedOn = true;
            }
            else
                status |= CAPE_FRAME_PLANE_OVERFLOW;
            break;
        }
    }
    if (handedOn)
        return;
    __syncthreads();
    // the record window borrowed s_adj for (list offset, length): back to zeros for merge_planes
    for (int i = tid; i < MAXP + 1; i += kGroupThreads)
        s_adj[i] = 0ull;
    // status bits the other waves hold are uniform ones (set by every lane alike), so wave 0's fold sees them all
    __syncthreads();
    if (wave != 0)
        return;
    GrowTailLds L;
    L.s_seg = s_seg;
    L.s_adj = s_adj;
    L.s_mlab = s_mlab;
    L.s_lab = s_lab;
    L.s_cyl = s_cyl;
    L.s_zc = s_zc;
    L.s_ring = reinterpret_cast<unsigned short*>(s_stage);
    L.s_prof = s_prof;
    grow_tail<MaskT, true, MAXP>(p, frame, lane, L, nSeg, nCylLabels, nSeeds, nPlanar, status, tPhase);
#ifdef CAPE_B_PROFILE
    CAPE_WAVE_SYNC();
    if (lane < kProfileSlots)
        p.debugCycles[(size_t)frame * kProfileSlots + lane] = s_prof[lane];
#endif
}

hipError_t launch_resume_group(const StageBParams& p, int nFrames, hipStream_t stream)
{
    const size_t lds = resume_group_lds_bytes(p.cells);
    if (lds > (size_t)p.ldsLimitBytes)
        return hipErrorInvalidConfiguration;
    if (p.hCells <= 32)
        hipLaunchKernelGGL(cape_resume_group_kernel<uint32_t>, dim3(nFrames), dim3(kGroupThreads), lds, stream, p, (int)lds);
    else if (p.hCells <= 64)
        hipLaunchKernelGGL(cape_resume_group_kernel<unsigned long long>, dim3(nFrames), dim3(kGroupThreads), lds, stream, p, (int)lds);
    else
        hipLaunchKernelGGL(cape_resume_group_kernel<Mask128>, dim3(nFrames), dim3(kGroupThreads), lds, stream, p, (int)lds);
    return hipGetLastError();
}

bool resume_group_fits(const StageBParams& p) { return resume_group_lds_bytes(p.cells) <= (size_t)p.ldsLimitBytes; }

} // namespace cape
