// Stage B -- histogram-seeded region growing, plane merge and boundary candidates: ONE WAVEFRONT PER FRAME.
//
// Replaces, per frame: Primitive_Detection::init_histogram, grow_planes_and_cylinders, grow_plane_segment_at_seed,
// region_growing, add_plane_segment_to_features, merge_planes, get_connected_components_matrix,
// add_planes_to_primitives, compute_plane_segment_boundary and add_cylinders_to_primitives
// (reference src/features/primitives/primitive_detection.cpp:239-776) plus Histogram<20> (histogram.hpp).
//
// The seed loop is sequential by construction (every seed consumes cells and histogram counts), so a frame gets
// one 64-lane wave and thousands of frames are in flight.  Inside the wave:
//   * the cell grid is held as bit rows: lane r owns row r of the 32x24 (u32), 64x48 (u64) or 96x54 (Mask128: two words) grid;
//   * the recursive DFS of region_growing is directed-graph reachability: the merge predicate of every directed
//     cell edge (parent plane vs child plane, child tolerance) is evaluated once per frame into four edge masks,
//     and growing a region is label propagation on bit rows (shift/and/or + lane shuffles) iterated with a
//     wave-wide vote until nothing changes;
//   * arg-max over the 400 histogram bins / arg-min MSE over the candidates are wave reductions with the
//     reference's first-index tie-breaks; moment sums of a region are accumulated in ascending cell order by nine
//     lanes (one per sum), because region sums are not exact and the order is observable.
// No workgroup barrier is needed: all cross-lane traffic is wave-synchronous (LDS + ballots).
#include <hip/hip_runtime.h>

#include "cape_grow_common.h"

namespace cape {

// MAXP: plane segments a frame may hold in this instance.  The two everyday instances keep kFastPlanes (32) segments in
// LDS; a frame that needs more is handed to the MAXP = CAPE_MAX_PLANES (64) instance through p.redoList, exactly like
// cylinder-branch frames are handed from the plane-only to the cylinder instance.
//
// RESUME (cylinder instance only): the wave does not grow its frame -- it takes a frame the plane-only pass parked at its
// first cylinder candidate (GrowStateHeader + arrays in p.growState: the segments made so far, the recorded regions with
// their fits, the cell lists and the label grid) and carries on from that region: the rest of the record -> segment
// conversion with cylinder_fitting, then merge_planes, boundaries and records like every other instance.  Without the
// histogram, the edge masks, the MSE registers and the seed loop this instance is compiled for two waves per SIMD.
// (the kernel's body; the __global__ function below adds the completion signal of the one-frame chain behind it)
template <typename MaskT, bool CYL, int MAXP, bool RESUME>
__device__ __forceinline__ void grow_frame_wave(const StageBParams& p, int nFrames, int ldsPerWave)
{
    static_assert(!RESUME || (CYL && MAXP == kFastPlanes), "only the 32-segment cylinder instance resumes parked frames");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int frame = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (frame >= nFrames)
        return; // whole wave leaves; there is no workgroup barrier in this kernel
    constexpr bool kRedo = MAXP > kFastPlanes;
    if (kRedo)
    {
        // third pass: wave k takes the k-th frame that ran out of segment slots in one of the 32-segment instances -- or, on the
        // one-frame chain (p.allFrames), frame k itself: this instance is then the only grow kernel of the call
        if (!p.allFrames)
        {
            if (frame >= (int)p.redoList[0])
                return;
            frame = (int)p.redoList[1 + frame];
        }
    }
    else if (RESUME)
    {
        // second pass, frames parked with their state: wave k takes the k-th of them
        frame = resume_pick(p, frame);
        if (frame < 0)
            return;
    }
    else if (CYL && p.twoPass)
    {
        // second pass of the two-pass schedule: wave k takes the k-th frame the plane-only pass gave up on
        if (frame >= (int)p.needCylinder[0])
            return;
        frame = (int)p.needCylinder[1 + frame];
    }
    unsigned char* smem = smem_all + (size_t)wave * ldsPerWave;
    const int C = p.cells, HC = p.hCells, VC = p.vCells;
    // timing on: the wave's ticks from here to grow_tail -- or to the point where it hands its frame on -- are the reference's
    // _growTime (grow_planes_and_cylinders); grow_tail books merge and refine
    const unsigned long long tPhase = p.phaseTicks ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull;
    auto book_grow_ticks = [&]() {
        if (p.phaseTicks && lane == 0)
            atomicAdd(&p.phaseTicks[(size_t)frame * 4], (unsigned long long)__builtin_amdgcn_s_memtime() - tPhase);
    };
    const size_t cellBase = (size_t)frame * C;

    // ---- LDS carve (every offset a multiple of 8; grow_lds_bytes() mirrors it).  The RESUME instance has no histogram, bins
    //      or separate staging chunk (its ordered passes stage through s_dist), and keeps the centre-pixel depths of the
    //      boundary phase where the RANSAC id arrays were: 17 KB per wave at 640x480.
    double* s_seg = reinterpret_cast<double*>(smem);                              // (MAXP + 1) x 20 f64 (one spare slot for the record window)
    double* s_afterSeg = s_seg + (MAXP + 1) * kSegDoubles;
    double* s_chunkOwn = s_afterSeg;                                               // kChunk x 10 f64 staging of cell sums
    unsigned long long* s_adj = reinterpret_cast<unsigned long long*>(RESUME ? s_afterSeg : s_chunkOwn + kChunkDoubles(C)); // MAXP + 1 u64
    int* s_hist = reinterpret_cast<int*>(s_adj + MAXP + 1);                        // 400 i32
    short* s_bins = reinterpret_cast<short*>(s_hist + (RESUME ? 0 : kHistBins));   // C i16
    unsigned short* s_list = reinterpret_cast<unsigned short*>(s_bins + (RESUME ? 0 : C)); // 1 pad + C u16 (rounded to 8 B)
    unsigned char* s_lab = reinterpret_cast<unsigned char*>(s_list + C + 4);      // C u8  plane labels
    unsigned char* s_mlab = s_lab + C;                                            // MAXP u8 merge labels
    // cylinder variant only (see grow_lds_bytes)
    unsigned char* s_cyl = s_mlab + MAXP;                                         // C u8  cylinder labels
    unsigned short* s_ids = reinterpret_cast<unsigned short*>(smem + (((size_t)(s_cyl + C - smem) + 3) & ~(size_t)3)); // C u16 idsLeft
    unsigned char* s_idmask = reinterpret_cast<unsigned char*>(s_ids + C);        // C u8
    unsigned char* s_best = s_idmask + C;                                         // C u8
    // inlier flags of the hypothesis being scored: only the streamed RANSAC path (regions beyond the register cache) uses them
    const bool needCur = !RESUME || C > 64 * kCylCacheRounds;
    unsigned char* s_cur = s_best + C;                                            // C u8 (absent when !needCur)
    // (aligned through the OFFSET, not through an integer cast of the pointer: the cast loses the LDS address space and every
    // access through s_dist / s_pendCyl becomes a flat_load that waits for vmcnt AND lgkmcnt)
    double* s_dist = reinterpret_cast<double*>(smem + (((size_t)(s_cur + (needCur ? C : 0) - smem) + 15) & ~(size_t)15)); // staging, 18 f64 x kChunk
    double* s_pendCyl = s_dist + cyl_dist_doubles(C);                             // kPendSlots region records (cylinder instances)
    double* s_chunk = RESUME ? s_dist : s_chunkOwn;
#ifdef CAPE_B_PROFILE
    unsigned long long* s_prof = reinterpret_cast<unsigned long long*>(smem + ldsPerWave - 8 * kProfileSlots);
    if (lane < kProfileSlots)
        s_prof[lane] = 0ull;
#endif
    // centre-pixel depths of the boundary phase: C f32 = exactly the bytes of s_bins + s_list (RESUME: of s_ids + s_idmask +
    // s_best), all dead after the seed loop / the last cylinder fit
    float* s_zc = RESUME ? reinterpret_cast<float*>(s_ids) : reinterpret_cast<float*>(s_bins);


    if constexpr (!RESUME)
        for (int i = lane; i < kHistBins; i += 64)
            s_hist[i] = 0;
    for (int i = lane; i < MAXP + 1; i += 64)
    {
        s_adj[i] = 0ull;
        if (i < MAXP)
            s_mlab[i] = (unsigned char)i;
    }
    CAPE_WAVE_SYNC();

    uint32_t status = 0;
    CAPE_TICK_INIT();
    MaskT U = 0, EL = 0, ER = 0, EU = 0, ED = 0; // bit rows, lane r <- grid row r (filled below)

    // =========================================================================================
    // init_histogram (primitive_detection.cpp:239-265, histogram.hpp:35-62) and the bit rows of the cell grid, lane r <- row r:
    // the unassigned mask and the four directed edge masks of region_growing's merge predicate (primitive_detection.cpp:802
    // with plane_segment.cpp:322-326), which stage A2 evaluated per cell and left in cell_flags:
    //   EL bit c : parent (r,c-1) -> child (r,c)      ER bit c : parent (r,c+1) -> child (r,c)
    //   EU bit c : parent (r-1,c) -> child (r,c)      ED bit c : parent (r+1,c) -> child (r,c)
    // One pass over flags + bins; the lanes take one grid row per ballot (two rows for grids up to 32 wide), and a batch of
    // steps is requested before the first is used, so the whole prologue costs about one memory round trip.
    // =========================================================================================
    int nPlanar = 0;
    if constexpr (!RESUME)
    {
        int nPlanarLocal = 0;
        if constexpr (sizeof(MaskT) == 16)
        {
            // rows of up to 128 cells (two mask words): two ballots per grid row, six rows requested together
            constexpr int kAheadW = 6;
            for (int t0 = 0; t0 < VC; t0 += kAheadW)
            {
                uint32_t fl[kAheadW][2];
                int bn[kAheadW][2];
#pragma unroll
                for (int k = 0; k < kAheadW; ++k)
#pragma unroll
                    for (int w = 0; w < 2; ++w)
                    {
                        const int r = t0 + k < VC ? t0 + k : VC - 1, c = 64 * w + lane;
                        const int ci = r * HC + (c < HC ? c : 0); // clamped: unconditional loads
                        fl[k][w] = p.cell_flags[cellBase + ci];
                        bn[k][w] = p.cell_bins[cellBase + ci];
                    }
#pragma unroll
                for (int k = 0; k < kAheadW; ++k)
                {
                    const int t = t0 + k;
                    if (t < VC)
                    {
                        unsigned long long bU[2], bL2M[2], bM2L[2], bU2M[2], bM2U[2];
#pragma unroll
                        for (int w = 0; w < 2; ++w)
                        {
                            const int c = 64 * w + lane;
                            const bool in = c < HC;
                            const uint32_t f = in ? fl[k][w] : 0u;
                            if (in)
                            {
                                const int ci = t * HC + c;
                                s_lab[ci] = 0;
                                if (CYL)
                                    s_cyl[ci] = 0;
                                s_bins[ci] = (short)bn[k][w];
                                if (f & kFlagPlanar)
                                {
                                    atomicAdd(&s_hist[bn[k][w]], 1);
                                    ++nPlanarLocal;
                                }
                            }
                            if (f & kFlagNearEdge)
                                status |= CAPE_FRAME_BIN_NEAR_EDGE;
                            if (f & kFlagInorder)
                                status |= CAPE_FRAME_INORDER_CELLS;
                            bU[w] = __ballot((f & kFlagPlanar) != 0);
                            bL2M[w] = __ballot((f & kFlagLeftToMe) != 0);
                            bM2L[w] = __ballot((f & kFlagMeToLeft) != 0);
                            bU2M[w] = __ballot((f & kFlagUpToMe) != 0);
                            bM2U[w] = __ballot((f & kFlagMeToUp) != 0);
                        }
                        if (lane == t)
                        {
                            U = MaskT(bU[0], bU[1]);
                            EL = MaskT(bL2M[0], bL2M[1]);
                            ER = MaskT(bM2L[0], bM2L[1]) >> 1;
                            EU = MaskT(bU2M[0], bU2M[1]);
                        }
                        if (lane == t - 1)
                            ED = MaskT(bM2U[0], bM2U[1]);
                    }
                }
            }
        }
        else
        {
            constexpr bool kTwoRows = sizeof(MaskT) == 4;
            const int h = kTwoRows ? (lane >> 5) : 0, col = kTwoRows ? (lane & 31) : lane;
            const bool colIn = col < HC;
            const int steps = kTwoRows ? (VC + 1) / 2 : VC;
            constexpr int kAhead = 12; // steps requested together (the 640x480 grid is 12 steps)
            for (int t0 = 0; t0 < steps; t0 += kAhead)
            {
                uint32_t fl[kAhead];
                int bn[kAhead];
    #pragma unroll
                for (int k = 0; k < kAhead; ++k)
                {
                    const int r = kTwoRows ? 2 * (t0 + k) + h : (t0 + k);
                    const int ci = (r < VC ? r : VC - 1) * HC + (colIn ? col : 0); // clamped: unconditional loads
                    fl[k] = p.cell_flags[cellBase + ci];
                    bn[k] = p.cell_bins[cellBase + ci];
                }
    #pragma unroll
                for (int k = 0; k < kAhead; ++k)
                {
                    const int t = t0 + k;
                    if (t < steps)
                    {
                        const int r = kTwoRows ? 2 * t + h : t;
                        const bool in = colIn && r < VC;
                        const uint32_t f = in ? fl[k] : 0u;
                        if (in)
                        {
                            const int ci = r * HC + col;
                            s_lab[ci] = 0;
                            if (CYL)
                                s_cyl[ci] = 0;
                            s_bins[ci] = (short)bn[k];
                            if (f & kFlagPlanar)
                            {
                                atomicAdd(&s_hist[bn[k]], 1);
                                ++nPlanarLocal;
                            }
                        }
                        if (f & kFlagNearEdge)
                            status |= CAPE_FRAME_BIN_NEAR_EDGE;
                        if (f & kFlagInorder)
                            status |= CAPE_FRAME_INORDER_CELLS;
                        const unsigned long long bU = __ballot((f & kFlagPlanar) != 0);
                        const unsigned long long bL2M = __ballot((f & kFlagLeftToMe) != 0);
                        const unsigned long long bM2L = __ballot((f & kFlagMeToLeft) != 0);
                        const unsigned long long bU2M = __ballot((f & kFlagUpToMe) != 0);
                        const unsigned long long bM2U = __ballot((f & kFlagMeToUp) != 0);
                        if (kTwoRows)
                        {
                            // low words: row 2t, high words: row 2t + 1 ; lane q keeps row q's masks (rows past the grid give zeros)
                            if (lane == 2 * t)
                            {
                                U = (MaskT)(uint32_t)bU;
                                EL = (MaskT)(uint32_t)bL2M;
                                ER = (MaskT)((uint32_t)bM2L >> 1);
                                EU = (MaskT)(uint32_t)bU2M;
                                ED = (MaskT)(uint32_t)(bM2U >> 32); // parent row 2t + 1 -> child row 2t
                            }
                            if (lane == 2 * t + 1)
                            {
                                U = (MaskT)(uint32_t)(bU >> 32);
                                EL = (MaskT)(uint32_t)(bL2M >> 32);
                                ER = (MaskT)((uint32_t)(bM2L >> 32) >> 1);
                                EU = (MaskT)(uint32_t)(bU2M >> 32);
                            }
                            if (lane == 2 * t - 1)
                                ED = (MaskT)(uint32_t)bM2U; // parent row 2t -> child row 2t - 1
                        }
                        else
                        {
                            if (lane == t)
                            {
                                U = (MaskT)bU;
                                EL = (MaskT)bL2M;
                                ER = (MaskT)(bM2L >> 1);
                                EU = (MaskT)bU2M;
                            }
                            if (lane == t - 1)
                                ED = (MaskT)bM2U;
                        }
                    }
                }
            }
        }
        // the vertical edges between the first cell row of a stage-A2 tile and the row above it (rows k * a2RowsPerTile):
        // both rows' planes are read back and the predicate is evaluated here, exactly as stage A2 does inside a tile
        if constexpr (sizeof(MaskT) == 16)
        {
            const int RPT = p.a2RowsPerTile;
            const int nB = (VC - 1) / RPT;
            for (int k = 1; k <= nB; ++k)
            {
                const int r = k * RPT;
                unsigned long long bU2M[2], bM2U[2];
                double2 m0[2], m1[2], m2[2], m3[2], u0[2], u1[2], u2[2], u3[2];
                float mt[2], ut[2];
#pragma unroll
                for (int w = 0; w < 2; ++w)
                {
                    const int c = 64 * w + lane;
                    const size_t ciMe = cellBase + (size_t)r * HC + (c < HC ? c : 0), ciUp = ciMe - HC;
                    const double2* pm = reinterpret_cast<const double2*>(p.cell_plane + ciMe * kPlaneStride);
                    const double2* pu = reinterpret_cast<const double2*>(p.cell_plane + ciUp * kPlaneStride);
                    m0[w] = pm[0]; m1[w] = pm[1]; m2[w] = pm[2]; m3[w] = pm[3];
                    u0[w] = pu[0]; u1[w] = pu[1]; u2[w] = pu[2]; u3[w] = pu[3];
                    mt[w] = p.cell_tol[ciMe];
                    ut[w] = p.cell_tol[ciUp];
                }
#pragma unroll
                for (int w = 0; w < 2; ++w)
                {
                    const bool on = 64 * w + lane < HC;
                    const bool u2m = on & can_be_merged(u0[w].x, u0[w].y, u1[w].x, u1[w].y, m0[w].x, m0[w].y, m1[w].x, m2[w].x, m2[w].y, m3[w].x, (double)mt[w], p.cosMerge);
                    const bool m2u = on & can_be_merged(m0[w].x, m0[w].y, m1[w].x, m1[w].y, u0[w].x, u0[w].y, u1[w].x, u2[w].x, u2[w].y, u3[w].x, (double)ut[w], p.cosMerge);
                    bU2M[w] = __ballot(u2m);
                    bM2U[w] = __ballot(m2u);
                }
                if (lane == r)
                    EU = MaskT(bU2M[0], bU2M[1]);
                if (lane == r - 1)
                    ED = MaskT(bM2U[0], bM2U[1]);
            }
        }
        else
        {
            constexpr bool kTwoRows = sizeof(MaskT) == 4;
            const int h = kTwoRows ? (lane >> 5) : 0, col = kTwoRows ? (lane & 31) : lane;
            const bool colIn = col < HC;
            const int RPT = p.a2RowsPerTile;
            const int nB = (VC - 1) / RPT; // boundaries at rows RPT, 2 RPT, ... < VC
            for (int s0 = 0; s0 < nB; s0 += (kTwoRows ? 2 : 1))
            {
                const int k = s0 + h + 1;
                const int r = k * RPT;
                const bool on = colIn && k <= nB;
                const int rr = (k <= nB ? r : RPT);
                const size_t ciMe = cellBase + (size_t)rr * HC + (colIn ? col : 0), ciUp = ciMe - HC;
                const double2* pm = reinterpret_cast<const double2*>(p.cell_plane + ciMe * kPlaneStride);
                const double2* pu = reinterpret_cast<const double2*>(p.cell_plane + ciUp * kPlaneStride);
                const double2 m0 = pm[0], m1 = pm[1], m2 = pm[2], m3 = pm[3];
                const double2 u0 = pu[0], u1 = pu[1], u2 = pu[2], u3 = pu[3];
                const double mtol = (double)p.cell_tol[ciMe], utol = (double)p.cell_tol[ciUp];
                const bool u2m = on & can_be_merged(u0.x, u0.y, u1.x, u1.y, m0.x, m0.y, m1.x, m2.x, m2.y, m3.x, mtol, p.cosMerge);
                const bool m2u = on & can_be_merged(m0.x, m0.y, m1.x, m1.y, u0.x, u0.y, u1.x, u2.x, u2.y, u3.x, utol, p.cosMerge);
                const unsigned long long bU2M = __ballot(u2m), bM2U = __ballot(m2u);
                if (kTwoRows)
                {
                    const int ra = (s0 + 1) * RPT, rb = (s0 + 2) * RPT; // boundary rows of the even / odd half
                    if (lane == ra)
                        EU = (MaskT)(uint32_t)bU2M;
                    if (lane == ra - 1)
                        ED = (MaskT)(uint32_t)bM2U;
                    if (s0 + 2 <= nB)
                    {
                        if (lane == rb)
                            EU = (MaskT)(uint32_t)(bU2M >> 32);
                        if (lane == rb - 1)
                            ED = (MaskT)(uint32_t)(bM2U >> 32);
                    }
                }
                else
                {
                    if (lane == r)
                        EU = (MaskT)bU2M;
                    if (lane == r - 1)
                        ED = (MaskT)bM2U;
                }
            }
        }
        nPlanar = wave_sum_i32(nPlanarLocal);
    }
    CAPE_WAVE_SYNC();
    CAPE_B_STOP(1);
    CAPE_TICK(0);

    CAPE_B_STOP(2);
    // =========================================================================================
    // grow_planes_and_cylinders (primitive_detection.cpp:267-310)
    // =========================================================================================
    int untried = nPlanar;
    int nSeg = 0;      // _planeSegments.size()
    int nCylLabels = 0; // cylinder2regionMap.size()
    int nCylFits = 0;   // _cylinderSegments.size()
    int rngPos = 0;     // draws consumed from the per-frame mt19937(0) stream
    int nSeeds = 0;
    const double* sumsBase = p.cell_sums + cellBase * kSumStride;
    const int maxSeedIters = 4 * C + 1024; // the loop provably terminates (every iteration burns a histogram count)

    // The cell MSEs never change: with up to 12 cells per lane (the 640x480 grid) they are fetched ONCE into registers
    // in exactly the lane <-> cell pattern of the candidate scan, so picking a seed costs no memory round trip.
    constexpr int kMseRegs = CAPE_B_MSE_REGS;
    const bool mseInRegs = C <= 64 * kMseRegs;
    const double* mseBase = p.cell_mse + cellBase;
    double mreg[kMseRegs];
    if constexpr (!RESUME)
    {
#pragma unroll
        for (int k = 0; k < kMseRegs; ++k)
        {
            const int i = lane + 64 * k;
            mreg[k] = mseBase[i < C ? i : 0];
        }
    }

    // What a seed's region is (label propagation), what it costs the histogram and the unassigned mask, and hence every
    // LATER seed, do not depend on the plane fitted to it (primitive_detection.cpp:332-389: the fit only decides whether
    // the region becomes a plane segment, goes to cylinder fitting or is dropped).  So the loop only RECORDS a region --
    // its ordered moment sums and its cell list -- and the fits of up to pendCap recorded regions run together, one lane
    // per region, instead of 64 lanes computing the same eigen-decomposition once per seed; the recorded regions are then
    // turned into segments in seed order.  Plane-only instances park the records in the free s_seg slots above nSeg (a
    // segment is never written above the record it comes from); the cylinder instance, whose cylinder_fitting appends
    // segments of its own, has a separate area.
    double* s_pend = CYL ? s_pendCyl : s_seg;     // base of the record window (plane-only: advanced to s_seg + nSeg at each flush)
    // (never more than 64 records in the window: their fits run one LANE each.  Through round 5 the 64-segment plane-only instance
    // let the window of an empty segment list grow to its 65 slots, and the 65th record -- a frame with that many regions before a
    // flush, which the round-6 fuzz sweep found -- was converted with whatever fit its slot held from before)
    constexpr int kWindowMax = 64;
    int pendCount = 0, pendCap = CYL ? kPendCyl : (MAXP + 1 < kWindowMax ? MAXP + 1 : kWindowMax);
    int listTop = 0;                               // bump pointer in s_list: recorded regions keep their cell lists
    bool moreSeeds = !RESUME;

    if constexpr (RESUME)
    {
        // ---- pick the frame up where the plane-only pass parked it
        const unsigned char* st = p.growState + (size_t)frame * p.growStateStride;
        const GrowStateHeader hd = *reinterpret_cast<const GrowStateHeader*>(st);
        const double* gseg = reinterpret_cast<const double*>(st + grow_state_seg_off());
        const unsigned long long* gadj = reinterpret_cast<const unsigned long long*>(st + grow_state_adj_off());
        const unsigned short* glist = reinterpret_cast<const unsigned short*>(st + grow_state_list_off());
        const unsigned char* glab = st + grow_state_lab_off(C);
        nSeg = hd.nSeg;
        nSeeds = hd.nSeeds;
        nPlanar = hd.nPlanar;
        untried = 0;
        status = status_resume(hd.status, lane == 0);
        const int nRec = hd.pendCount - hd.pendFrom;
        for (int i = lane; i < nSeg * kSegDoubles; i += 64)
            s_seg[i] = gseg[i];
        for (int i = lane; i < nRec * kSegDoubles; i += 64)
            s_pendCyl[i] = gseg[(hd.pendBaseSlot + hd.pendFrom) * kSegDoubles + i];
        if (lane < nRec)
            s_adj[lane] = gadj[hd.pendFrom + lane];
        for (int i = lane; i < C + 4; i += 64)
            s_list[i] = glist[i];
        for (int i = lane; i < C; i += 64)
        {
            s_lab[i] = glab[i];
            s_cyl[i] = 0;
        }
        pendCount = nRec;
        CAPE_WAVE_SYNC();
    }

    for (;;)
    {
        if (moreSeeds && !(untried > 0 && nSeeds < maxSeedIters))
            moreSeeds = false;
        if constexpr (!RESUME)
        if (moreSeeds)
            do
            {
                // ---- Histogram::get_points_from_most_frequent_bin (histogram.hpp:69-98): first index of the greatest count
                unsigned key = 0;
                for (int b = lane; b < kHistBins; b += 64)
                {
                    const int h = s_hist[b];
                    const unsigned k = ((unsigned)h << 16) | (unsigned)(0xFFFF - b);
                    key = (h > 0 && k > key) ? k : key;
                }
                key = wave_max_u32(key);
                if (key == 0)
                {
                    moreSeeds = false; // mostFrequentBin = -1 -> empty candidate list -> size < planeSeedCount
                    break;
                }
                const int bin = 0xFFFF - (int)(key & 0xFFFFu);
                CAPE_TICK(1); // histogram arg-max

                // ---- candidates = cells with _bins == bin ; seed = first strict minimum of MSE (:285-298)
                int candLocal = 0;
                unsigned long long bestLocal = ~0ull; // (mse bits) ; mse >= 0 so the bit pattern orders like the value
                int bestIdxLocal = 0x7FFFFFFF;
                if (mseInRegs)
                {
#pragma unroll
                    for (int k = 0; k < kMseRegs; ++k)
                    {
                        const int i = lane + 64 * k;
                        if (i < C && s_bins[i] == (short)bin)
                        {
                            ++candLocal;
                            const unsigned long long mb = (unsigned long long)__double_as_longlong(mreg[k]);
                            if (mb < bestLocal)
                            {
                                bestLocal = mb;
                                bestIdxLocal = i;
                            }
                        }
                    }
                }
                else
                {
                    constexpr int kBatch = 12; // 12 independent coalesced loads in flight per lane
                    for (int i0 = lane; i0 < C; i0 += 64 * kBatch)
                    {
                        double mv[kBatch];
#pragma unroll
                        for (int k = 0; k < kBatch; ++k)
                        {
                            const int i = i0 + 64 * k;
                            mv[k] = mseBase[i < C ? i : 0];
                        }
#pragma unroll
                        for (int k = 0; k < kBatch; ++k)
                        {
                            const int i = i0 + 64 * k;
                            if (i < C && s_bins[i] == (short)bin)
                            {
                                ++candLocal;
                                const unsigned long long mb = (unsigned long long)__double_as_longlong(mv[k]);
                                if (mb < bestLocal)
                                {
                                    bestLocal = mb;
                                    bestIdxLocal = i;
                                }
                            }
                        }
                    }
                }
                CAPE_TICK(2); // candidate scan
                const int cand = wave_sum_i32(candLocal);
                if (cand < p.planeSeedCount || cand == 0)
                {
                    moreSeeds = false;
                    break;
                }
                const unsigned long long bestAll = wave_min_u64(bestLocal);
                const unsigned idxKey = (bestLocal == bestAll) ? (unsigned)(0x7FFFFFFF - bestIdxLocal) : 0u;
                const int seed = 0x7FFFFFFF - (int)wave_max_u32(idxKey);
                if (__longlong_as_double((long long)bestAll) >= kDblMax)
                {
                    moreSeeds = false; // "invalid seed" (:299-304)
                    status |= CAPE_FRAME_INVALID_SEED;
                    break;
                }
                if (lane == 0 && p.seed_sequence && nSeeds < C)
                    p.seed_sequence[cellBase + nSeeds] = (uint16_t)seed; // debug / parity stream: seeds in the order they were tried
                ++nSeeds;

                // ---- grow_plane_segment_at_seed (:312-389).  The seed's own plane is requested now and looked at after the
                //      propagation, which does not need it: the memory round trip hides behind the label propagation.
                const int sy = seed / HC, sx = seed - sy * HC;
                const double2* spl = reinterpret_cast<const double2*>(p.cell_plane + (cellBase + seed) * kPlaneStride);
                const double2 sp0 = spl[0], sp1 = spl[1], sp2 = spl[2], sp3 = spl[3];
                const float stolf = p.cell_tol[cellBase + seed];
                const MaskT seedRowU = shfl_mask<MaskT>(U, sy);
                const bool seedUnassigned = test_bit<MaskT>(seedRowU, sx);

                CAPE_TICK(3); // seed pick
                // ---- region_growing (:778-818) as label propagation on bit rows
                MaskT act = 0;
                if (seedUnassigned)
                {
                    if (lane == sy)
                        act = (MaskT)1 << sx;
                    for (;;)
                    {
                        MaskT a = act;
                        for (;;)
                        {
                            const MaskT na = a | (U & (((MaskT)(a << 1) & EL) | ((MaskT)(a >> 1) & ER)));
                            if (na == a)
                                break;
                            a = na;
                        }
                        const MaskT up = Rows<MaskT>::up(a, lane);
                        const MaskT dn = Rows<MaskT>::dn(a, lane, VC);
                        a |= U & ((up & EU) | (dn & ED));
                        const bool changed = (a != act);
                        act = a;
                        if (!__any(changed))
                            break;
                    }
                }
                {
                    // the seed's own test: newPlaneSegment(planeToGrow) is a copy, and the copy re-normalises the normal
                    // (plane_coordinates.hpp:24-27) before can_be_merged compares it with the original
                    const double snx = sp0.x, sny = sp0.y, snz = sp1.x, sd = sp1.y, scx = sp2.x, scy = sp2.y, scz = sp3.x;
                    double pnx = snx, pny = sny, pnz = snz;
                    normalize3(pnx, pny, pnz);
                    const bool seedOK = can_be_merged(pnx, pny, pnz, sd, snx, sny, snz, scx, scy, scz, (double)stolf, p.cosMerge);
                    if (!seedOK)
                        act = 0;
                }

                CAPE_TICK(4); // label propagation + seed self test
                // ---- activated cell list in ascending cell index (row-major), appended at listTop
                const int rowCnt = popc<MaskT>(act);
                const int incl = wave_scan_i32(rowCnt);
                const int total = (int)readlane_u32((unsigned)incl, 63);
                unsigned short* rlist = s_list + 1 + listTop; // s_list[0] is a pad: the slot before a list parks the seed (below)
                {
                    int pos = incl - rowCnt;
                    MaskT m = act;
                    while (m)
                    {
                        const int c = ctz<MaskT>(m);
                        rlist[pos++] = (unsigned short)(lane * HC + c);
                        m = clear_lowest<MaskT>(m);
                    }
                }
                CAPE_WAVE_SYNC();

                // ---- expand_segment over activated cells in ascending order (:341-360): lanes 0..8 own one sum each, lane 9
                //      the point count.  The seed's own sums are counted twice (copy :325 + expand of the seed itself).
                CAPE_TICK(5); // list build
                const int ql = lane < 10 ? lane : 0;
                // element 0 is the seed itself (0.0 + x == x exactly), so its sums travel with the first staged chunk instead of
                // costing a memory round trip of their own.  The seed id is parked in the list slot just before the region's
                // cells for the duration of the pass (that slot is the last cell of the previous recorded region, or the pad):
                // the index function is then one LDS look-up with no branch in front of the loads.
                const unsigned short parkedOver = rlist[-1];
                CAPE_LDS_SYNC();
                if (lane == 0)
                    rlist[-1] = (unsigned short)seed;
                CAPE_LDS_SYNC();
                double acc = 0.0;
                staged_for_each<5, CAPE_STAGE_DEPTH_MAIN>(
                        total + 1, sumsBase, kSumStride, 0, [&](int e) { return (int)rlist[e - 1]; }, s_chunk, lane,
                        [&](int, const double* rec) { return rec[ql]; }, [&](int, double v) { acc += v; });
                if (lane == 0)
                    rlist[-1] = parkedOver;
                CAPE_LDS_SYNC();

                CAPE_TICK(6); // ordered accumulation
                // ---- Histogram::remove_point for every activated cell (histogram.hpp:103-113), _isUnassignedMask = false
                for (int i = lane; i < total; i += 64)
                {
                    const int cidx = rlist[i];
                    atomicSub(&s_hist[s_bins[cidx]], 1);
                    s_bins[cidx] = 1; // quirk: 1, not -1
                }
                CAPE_WAVE_SYNC();
                if (lane == 0 && s_hist[1] < 0)
                    s_hist[1] = 0; // "if != 0: -= 1" saturates; only bin 1 can be over-decremented
                U &= ~act;
                untried -= total;

                if (total == 0 || total < p.minCellActivated)
                {
                    if (lane == 0)
                    {
                        const int b = s_bins[seed];
                        if (s_hist[b] != 0)
                            s_hist[b] -= 1;
                        s_bins[seed] = 1;
                    }
                    CAPE_WAVE_SYNC();
                    break; // region dropped: nothing recorded, its list slots are reused
                }
                // ---- record the region: sums + count into the window slot, list kept
                if (lane < 10)
                    s_pend[pendCount * kSegDoubles + lane] = acc;
                if (lane == 0)
                    s_adj[pendCount] = (unsigned long long)(unsigned)listTop | ((unsigned long long)(unsigned)total << 32);
                listTop += total;
                ++pendCount;
                CAPE_WAVE_SYNC();
                CAPE_TICK(7); // histogram removal + record
            } while (0);

        if (moreSeeds && pendCount < pendCap)
            continue;
        if (pendCount > 0)
        {
            // ---- fit_plane (plane_segment.cpp:232-284) of every recorded region, one lane per region
            CAPE_TICK_RESTART();
            if (!RESUME && lane < pendCount) // (a parked frame's records carry their fits)
            {
                double* slot = s_pend + lane * kSegDoubles;
                double S[9];
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    S[k] = slot[k];
                PlaneFit f;
                fit_plane(S, (uint32_t)slot[9], f);
                // add_plane_segment_to_features (:391-411): push_back copies the segment (one more normalisation); only a
                // region that becomes a plane segment reads the normal, so the copy's normalisation is applied here
                double nx = f.nx, ny = f.ny, nz = f.nz;
                normalize3(nx, ny, nz);
                slot[10] = nx; slot[11] = ny; slot[12] = nz; slot[13] = f.d;
                slot[14] = f.cx; slot[15] = f.cy; slot[16] = f.cz;
                slot[17] = f.mse; slot[18] = f.score; slot[19] = f.planar ? 1.0 : 0.0;
            }
            CAPE_WAVE_SYNC();
            CAPE_TICK(8); // region plane fits (lane parallel)
            bool stopAll = false;
            for (int j = 0; j < pendCount && !stopAll; ++j)
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
                if (ns.score > 100)
                {
                    if (nSeg >= MAXP)
                    {
                        if (!kRedo && p.redoList)
                        {
                            // out of LDS segment slots: the 64-segment instance redoes this frame from the start
                            if (lane == 0)
                                p.redoList[1 + atomicAdd(&p.redoList[0], 1u)] = (uint32_t)frame;
                            book_grow_ticks();
                            return;
                        }
                        if (kRedo && p.spillList)
                        {
                            // more than a record holds: the general instance redoes the frame into a chain of records
                            if (lane == 0)
                                p.spillList[1 + atomicAdd(&p.spillList[0], 1u)] = (uint32_t)frame;
                            book_grow_ticks();
                            return;
                        }
                        status |= CAPE_FRAME_PLANE_OVERFLOW;
                        stopAll = true;
                        break;
                    }
                    if (lane == 0)
                        seg_store(s_seg + nSeg * kSegDoubles, ns); // at or below the record's own slot (plane-only window)
                    ++nSeg;
                    for (int i = lane; i < total; i += 64)
                        s_lab[s_list[1 + roff + i]] = (unsigned char)nSeg;
                    CAPE_WAVE_SYNC();
                }
                else if (!CYL && !kRedo && p.twoPass && total > 5)
                {
                    // first pass of the two-pass schedule: this region goes to cylinder_fitting, which the plane-only kernel does
                    // not carry.  If the seed loop is over (every region of the frame is recorded -- the usual case: the window
                    // holds 33 regions) the frame is PARKED: segments so far, the window with its fits, cell lists and labels go
                    // to p.growState and the RESUME instance of the cylinder kernel carries on from this record.  Otherwise the
                    // whole frame is handed to the cylinder kernel, which starts it over (nothing written so far counts).
                    const bool parked = MAXP == kFastPlanes && p.resumeList != nullptr && !moreSeeds && pendCount - j <= kPendResume;
                    if (parked)
                    {
                        unsigned char* st = p.growState + (size_t)frame * p.growStateStride;
                        const uint32_t statusAll = wave_or_u32(status);
                        const int baseSlot = (int)((s_pend - s_seg) / kSegDoubles);
                        if (lane == 0)
                        {
                            GrowStateHeader hd;
                            hd.nSeg = nSeg;
                            hd.pendFrom = j;
                            hd.pendCount = pendCount;
                            hd.pendBaseSlot = baseSlot;
                            hd.nSeeds = nSeeds;
                            hd.nPlanar = nPlanar;
                            hd.status = statusAll;
                            hd.pad = 0;
                            *reinterpret_cast<GrowStateHeader*>(st) = hd;
                        }
                        double* gseg = reinterpret_cast<double*>(st + grow_state_seg_off());
                        unsigned long long* gadj = reinterpret_cast<unsigned long long*>(st + grow_state_adj_off());
                        unsigned short* glist = reinterpret_cast<unsigned short*>(st + grow_state_list_off());
                        unsigned char* glab = st + grow_state_lab_off(C);
                        for (int i = lane; i < (baseSlot + pendCount) * kSegDoubles; i += 64)
                            gseg[i] = s_seg[i];
                        if (lane < pendCount)
                            gadj[lane] = s_adj[lane];
                        for (int i = lane; i < C + 4; i += 64)
                            glist[i] = s_list[i];
                        for (int i = lane; i < C; i += 64)
                            glab[i] = s_lab[i];
                        // cost class of what is left to do: the cells of the records that will go to cylinder_fitting
                        int candCells = 0;
                        if (lane < pendCount - j)
                        {
                            const double* rs = s_pend + (j + lane) * kSegDoubles;
                            const int cellsOf = (int)(s_adj[j + lane] >> 32);
                            if (rs[19] != 0.0 && !(rs[18] > 100) && cellsOf > 5)
                                candCells = cellsOf;
                        }
                        candCells = wave_sum_i32(candCells);
                        if (lane == 0)
                        {
                            p.resumeList[1 + atomicAdd(&p.resumeList[0], 1u)] = (uint32_t)frame;
                            if (p.resumeBucketStride)
                            {
                                uint32_t* bl = p.resumeList + (size_t)(1 + resume_cost_class(candCells, C)) * p.resumeBucketStride;
                                bl[1 + atomicAdd(&bl[0], 1u)] = (uint32_t)frame;
                            }
                        }
                    }
                    else if (lane == 0)
                        p.needCylinder[1 + atomicAdd(&p.needCylinder[0], 1u)] = (uint32_t)frame;
                    book_grow_ticks();
                    return;
                }
                else if (CYL && total > 5)
                {
                    // cylinder_fitting (:478-501) ; CYL == false is the "plane-only" mode (region dropped, cells stay consumed)
                    CylCtx cc;
                    cc.p = &p;
                    cc.lane = lane;
                    cc.cellBase = cellBase;
                    cc.C = C;
                    cc.s_list = s_list + 1 + roff;
                    cc.total = total;
                    cc.s_dist = s_dist;
                    cc.s_ids = s_ids;
                    cc.s_idmask = s_idmask;
                    cc.s_cur = s_cur;
                    cc.s_best = s_best;
                    cc.scratch = p.cylScratch + cellBase * kCylStride;
                    cc.s_stage = s_chunk;
                    cc.s_seg = s_seg;
                    cc.s_lab = s_lab;
                    cc.s_cyl = s_cyl;
                    cc.cylOut = p.records[frame].cylinders;
                    cc.maxCylinders = CAPE_MAX_CYLINDERS;
                    cc.maxPlanes = MAXP;
#ifdef CAPE_B_PROFILE
                    cc.dbg = s_prof;
#else
                    cc.dbg = nullptr;
#endif
                    bool planeOverflow = false;
                    cylinder_fitting(cc, nSeg, nCylLabels, nCylFits, rngPos, status, planeOverflow);
                    ++nCylFits;
                    CAPE_WAVE_SYNC();
                    if (planeOverflow)
                    {
                        if (!kRedo && p.redoList)
                        {
                            if (lane == 0)
                                p.redoList[1 + atomicAdd(&p.redoList[0], 1u)] = (uint32_t)frame;
                            book_grow_ticks();
                            return;
                        }
                        if (kRedo && p.spillList)
                        {
                            if (lane == 0)
                                p.spillList[1 + atomicAdd(&p.spillList[0], 1u)] = (uint32_t)frame;
                            book_grow_ticks();
                            return;
                        }
                        status |= CAPE_FRAME_PLANE_OVERFLOW;
                        stopAll = true;
                        break;
                    }
                }
            }
            CAPE_TICK_RESTART(); // the cylinder phases booked themselves in slots 12..15
            pendCount = 0;
            if (stopAll)
                break;
            if (!CYL)
            {
                // plane-only: the next window starts at the first free segment slot (one spare slot past MAXP keeps it non-empty)
                s_pend = s_seg + nSeg * kSegDoubles;
                pendCap = MAXP + 1 - nSeg < kWindowMax ? MAXP + 1 - nSeg : kWindowMax;
            }
        }
        if (!moreSeeds)
            break;
    }
    // the record window borrowed s_adj for (list offset, length): back to zeros for merge_planes
    for (int i = lane; i < MAXP + 1; i += 64)
        s_adj[i] = 0ull;
    CAPE_WAVE_SYNC();

    if (untried > 0 && nSeeds >= maxSeedIters)
        status |= CAPE_FRAME_SEED_LIMIT; // cannot happen (see maxSeedIters); says so if it ever does

    CAPE_B_STOP(3);
    CAPE_TICK(9);
    {
        GrowTailLds L;
        L.s_seg = s_seg;
        L.s_adj = s_adj;
        L.s_mlab = s_mlab;
        L.s_lab = s_lab;
        L.s_cyl = s_cyl;
        L.s_zc = s_zc;
        L.s_ring = reinterpret_cast<unsigned short*>(s_chunk); // kChunk * 10 * 8 / 2 = 1280 entries
#ifdef CAPE_B_PROFILE
        L.s_prof = s_prof;
#else
        L.s_prof = nullptr;
#endif
        grow_tail<MaskT, CYL, MAXP>(p, frame, lane, L, nSeg, nCylLabels, nSeeds, nPlanar, status, tPhase);
    }
#ifdef CAPE_B_PROFILE
    CAPE_WAVE_SYNC();
    if (lane < kProfileSlots)
        p.debugCycles[(size_t)frame * kProfileSlots + lane] = s_prof[lane];
#endif
}

template <typename MaskT, bool CYL, int MAXP, bool RESUME>
__global__ __launch_bounds__(64 * kWavesPerGroup, CYL ? (RESUME ? CAPE_B_RESUME_WAVES : 1) : CAPE_B_PLANE_WAVES) void cape_grow_kernel(StageBParams p, int nFrames, int ldsPerWave)
{
    grow_frame_wave<MaskT, CYL, MAXP, RESUME>(p, nFrames, ldsPerWave);
    // One-frame handles keep their results in pinned host memory and the caller spins on a sequence number there
    // (cape_host_results).  The 64-segment instance is the LAST kernel of such a chain -- with p.allFrames the only grow kernel --,
    // so its waves count themselves out and the last one stores the number: no one-thread kernel behind the chain.
    if constexpr (MAXP > kFastPlanes)
    {
        if (p.doneFlag && (threadIdx.x & 63) == 0)
        {
            __threadfence_system(); // this wave's records, label grids and boundary points first
            const unsigned total = gridDim.x * (blockDim.x >> 6);
            if (atomicAdd(p.doneCounter, 1u) == total - 1u)
            {
                atomicExch(p.doneCounter, 0u); // ready for the next chain
                // a frame handed to the general instance is not done yet: either that kernel is enqueued right behind and signals
                // instead, or (p.spillHost) the host is told how many frames wait for it and enqueues it itself
                const unsigned waiting = p.spillList ? atomicAdd(&p.spillList[0], 0u) : 0u;
                if (p.spillHost)
                    __hip_atomic_store(p.spillHost, waiting, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (p.spillHost || waiting == 0u)
                    __hip_atomic_store(p.doneFlag, p.doneSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

size_t grow_lds_bytes(int cells, bool cylinders, int maxPlanes, bool resume)
{
    size_t b = 0;
    b += (size_t)(maxPlanes + 1) * kSegDoubles * 8; // s_seg (+ one spare slot for the record window)
    if (!resume)
        b += (size_t)kChunkDoubles(cells) * 8;      // s_chunk (the RESUME instance stages through s_dist)
    b += (size_t)(maxPlanes + 1) * 8;               // s_adj
    if (!resume)
    {
        b += (size_t)kHistBins * 4;                 // s_hist
        b += (size_t)cells * 2;                     // s_bins  } after the seed loop these two hold s_zc
    }
    b += (size_t)cells * 2 + 8;                     // s_list  } (+ pad entry)
    b += (size_t)cells;                             // s_lab
    b += maxPlanes;                                 // s_mlab
    if (cylinders)
    {
        b += (size_t)cells;                         // s_cyl
        b = (b + 3) & ~(size_t)3;
        b += (size_t)cells * 2 + (size_t)cells * 2; // s_ids, s_idmask, s_best
        if (!resume || cells > 64 * kCylCacheRounds)
            b += (size_t)cells;                     // s_cur
        b = (b + 15) & ~(size_t)15;
        b += (size_t)cyl_dist_doubles(cells) * 8;   // s_dist
        b += (size_t)(resume ? kPendResume : kPendCyl) * kSegDoubles * 8; // s_pendCyl
    }
#ifdef CAPE_B_PROFILE
    b = ((b + 15) & ~(size_t)15) + 8 * kProfileSlots; // s_prof
#endif
    return (b + 15) & ~(size_t)15;
}
size_t grow_lds_bytes(int cells, bool cylinders, int maxPlanes) { return grow_lds_bytes(cells, cylinders, maxPlanes, false); }
size_t grow_state_bytes(int cells) { return grow_state_bytes_(cells); }

int grow_waves_per_group() { return kWavesPerGroup; }

// frame-waves of the grow kernel that one CU can hold at once (occupancy API; advisory, see MI355X_MICROARCH.md)
int grow_waves_per_cu(const StageBParams& p)
{
    const bool cyl = (p.flags & CAPE_FLAG_CYLINDERS) != 0;
    const int ldsPerWave = (int)grow_lds_bytes(p.cells, cyl, kFastPlanes);
    int wpg = kWavesPerGroup;
    while (wpg > 1 && (size_t)ldsPerWave * wpg > (size_t)p.ldsLimitBytes)
        --wpg;
    int blocks = 0;
    hipError_t e;
    if (p.hCells > 64)
        e = cyl ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cape_grow_kernel<Mask128, true, kFastPlanes, false>, 64 * wpg, (size_t)ldsPerWave * wpg)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cape_grow_kernel<Mask128, false, kFastPlanes, false>, 64 * wpg, (size_t)ldsPerWave * wpg);
    else if (p.hCells <= 32)
        e = cyl ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cape_grow_kernel<uint32_t, true, kFastPlanes, false>, 64 * wpg, (size_t)ldsPerWave * wpg)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cape_grow_kernel<uint32_t, false, kFastPlanes, false>, 64 * wpg, (size_t)ldsPerWave * wpg);
    else
        e = cyl ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cape_grow_kernel<unsigned long long, true, kFastPlanes, false>, 64 * wpg, (size_t)ldsPerWave * wpg)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cape_grow_kernel<unsigned long long, false, kFastPlanes, false>, 64 * wpg, (size_t)ldsPerWave * wpg);
    return e == hipSuccess ? blocks * wpg : 0;
}

hipError_t launch_resume_group(const StageBParams& p, int nFrames, hipStream_t stream); // cape_resume.hip
bool resume_group_fits(const StageBParams& p);
hipError_t launch_grow_general(const StageBParams& p, const GenParams& g, int nFrames, hipStream_t stream); // cape_grow_general.hip

namespace {

template <bool CYL, int MAXP, bool RESUME = false>
hipError_t launch_grow_variant(const StageBParams& p, int nFrames, hipStream_t stream)
{
    const int ldsPerWave = (int)grow_lds_bytes(p.cells, CYL, MAXP, RESUME);
    int wpg = kWavesPerGroup; // as many independent frame-waves per workgroup as the device's LDS admits (<= kWavesPerGroup)
    while (wpg > 1 && (size_t)ldsPerWave * wpg > (size_t)p.ldsLimitBytes)
        --wpg;
    const size_t lds = (size_t)ldsPerWave * wpg;
    if (lds > (size_t)p.ldsLimitBytes)
        return hipErrorInvalidConfiguration; // cape_create refuses such grids; belt and braces
    const dim3 grid((nFrames + wpg - 1) / wpg), block(64 * wpg);
    if (p.hCells <= 32)
        hipLaunchKernelGGL((cape_grow_kernel<uint32_t, CYL, MAXP, RESUME>), grid, block, lds, stream, p, nFrames, ldsPerWave);
    else if (p.hCells <= 64)
        hipLaunchKernelGGL((cape_grow_kernel<unsigned long long, CYL, MAXP, RESUME>), grid, block, lds, stream, p, nFrames, ldsPerWave);
    else // rows of up to 128 cells: two mask words per lane (1920 x 1080 = 96 x 54 cells)
        hipLaunchKernelGGL((cape_grow_kernel<Mask128, CYL, MAXP, RESUME>), grid, block, lds, stream, p, nFrames, ldsPerWave);
    return hipGetLastError();
}

} // namespace

// With cylinders enabled the kernel that carries cylinder_fitting holds one wave per SIMD (four frames per CU) against
// twelve for the plane-only kernel, and most frames of most scenes never take the cylinder branch.  So every frame is
// first grown by the plane-only kernel; a frame whose seed loop reaches a cylinder-candidate region (score <= 100 and
// more than 5 cells, primitive_detection.cpp:385-388) is abandoned there and appended to a list, and the cylinder kernel
// -- launched for the worst case, wave k takes the k-th listed frame, the others leave at once -- redoes exactly those
// frames from the start (densely packed: a workgroup of the cylinder kernel owns 60 % of a CU's LDS, so idle waves in
// it would cost real occupancy).  A frame that never
// takes the branch is bit-identical in both kernels (the branch is their only difference).
// `side` (optional): the SECOND pass -- everything behind the plane-only kernel, or the cylinder kernel alone -- is enqueued on
// this stream, forked from `stream` through `fork` and closed by `done`; `stream` itself does not wait for it (the caller of
// the next entry point on this handle does).  A second handle's streaming kernels then run under this one's slow tail.
// `gen`: the general instance (cape_grow_general.hip) -- with gen->allFrames it is the ONLY grow kernel of the handle (grids beyond
// 128 x 64 cells); otherwise it runs last, behind the 64-segment instance and on the same stream, over the frames that instance
// listed in p.spillList (more than one record holds; none, as a rule: its waves leave at once).
hipError_t launch_grow(const StageBParams& p, int nFrames, hipStream_t stream, hipStream_t side, hipEvent_t fork, hipEvent_t done,
                       const GenParams* gen)
{
    const bool cyl = (p.flags & CAPE_FLAG_CYLINDERS) != 0;
    if (gen && gen->allFrames)
        return launch_grow_general(p, *gen, nFrames, stream);
    if (!cyl)
        side = nullptr;
    hipStream_t second = side ? side : stream;
#define CAPE_LAUNCH_TRY(expr)              \
    do                                     \
    {                                      \
        const hipError_t e_ = (expr);      \
        if (e_ != hipSuccess)              \
            return e_;                     \
    } while (0)
    if (p.allFrames)
    {
        // the one-frame chain: the 64-segment instance grows every frame of the call itself (and signals the host, see the kernel)
        if (cyl)
            CAPE_LAUNCH_TRY((launch_grow_variant<true, CAPE_MAX_PLANES>(p, nFrames, stream)));
        else
            CAPE_LAUNCH_TRY((launch_grow_variant<false, CAPE_MAX_PLANES>(p, nFrames, stream)));
        if (gen && p.spillList && !p.spillHost)
            CAPE_LAUNCH_TRY(launch_grow_general(p, *gen, nFrames, stream));
        return hipSuccess;
    }
    // counters of the two hand-over lists ([0] = count, [1..] = frames)
    if (p.redoList && !p.countersCleared)
        CAPE_LAUNCH_TRY(hipMemsetAsync(p.redoList, 0, sizeof(uint32_t), stream));
    if (!cyl)
    {
        CAPE_LAUNCH_TRY((launch_grow_variant<false, kFastPlanes>(p, nFrames, stream)));
        if (p.redoList)
            CAPE_LAUNCH_TRY((launch_grow_variant<false, CAPE_MAX_PLANES>(p, nFrames, stream))); // frames with more than 32 segments (rare)
        if (gen && p.redoList && p.spillList)
            CAPE_LAUNCH_TRY(launch_grow_general(p, *gen, nFrames, stream));                      // ... with more than 64
        return hipSuccess;
    }
    if (!p.twoPass)
    {
        if (side)
        {
            CAPE_LAUNCH_TRY(hipEventRecord(fork, stream));
            CAPE_LAUNCH_TRY(hipStreamWaitEvent(side, fork, 0));
        }
        CAPE_LAUNCH_TRY((launch_grow_variant<true, kFastPlanes>(p, nFrames, second))); // nearly every frame needs it anyway: skip the plane-only pass
    }
    else
    {
        if (!p.countersCleared)
            CAPE_LAUNCH_TRY(hipMemsetAsync(p.needCylinder, 0, sizeof(uint32_t), stream));
        if (p.resumeList && !p.countersCleared)
            for (int c = 0; c <= (p.resumeBucketStride ? kResumeClasses : 0); ++c)
                CAPE_LAUNCH_TRY(hipMemsetAsync(p.resumeList + (size_t)c * p.resumeBucketStride, 0, sizeof(uint32_t), stream));
        CAPE_LAUNCH_TRY((launch_grow_variant<false, kFastPlanes>(p, nFrames, stream)));
        if (side)
        {
            CAPE_LAUNCH_TRY(hipEventRecord(fork, stream));
            CAPE_LAUNCH_TRY(hipStreamWaitEvent(side, fork, 0));
        }
        if (p.resumeList && p.resumeMode == 2)
            CAPE_LAUNCH_TRY(launch_resume_group(p, nFrames, second)); // frames parked with their state: one workgroup each
        else if (p.resumeList)
            CAPE_LAUNCH_TRY((launch_grow_variant<true, kFastPlanes, true>(p, nFrames, second))); // ... or one wavefront each
        CAPE_LAUNCH_TRY((launch_grow_variant<true, kFastPlanes>(p, nFrames, second)));           // frames that start over (rare once parking is on)
    }
    if (p.redoList)
        CAPE_LAUNCH_TRY((launch_grow_variant<true, CAPE_MAX_PLANES>(p, nFrames, second)));
    if (gen && p.redoList && p.spillList)
        CAPE_LAUNCH_TRY(launch_grow_general(p, *gen, nFrames, second));
    if (side)
        CAPE_LAUNCH_TRY(hipEventRecord(done, side));
    return hipSuccess;
#undef CAPE_LAUNCH_TRY
}

} // namespace cape
