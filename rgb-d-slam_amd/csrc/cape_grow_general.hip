// Stage B, the GENERAL instance: any cell grid, any number of plane segments -- one wavefront per frame.
//
// The reference's detector takes any image size (Primitive_Detection(width, height), primitive_detection.cpp:26-67) and keeps its
// plane segments in an unbounded std::vector (primitive_detection.hpp:206, pushed at primitive_detection.cpp:391-411 and :437-476).
// The everyday instances of the grow kernel (cape_grow.hip) buy their speed with two fixed shapes: a grid row is ONE mask of 32, 64
// or 128 bits held by the lane of that row (grids up to 128 x 64 cells: 1920 x 1080 is 96 x 54) and a frame's segments sit in 32 or
// 64 LDS slots, "lane j <- segment j" in every pass behind the seed loop.  This instance has neither limit and takes
//   * every frame of a handle whose grid has more than 64 rows or 128 columns (portrait 1080 x 1920, 2560 x 1440, 4K), and
//   * on the other handles, the frames the 64-segment instance ran out of record capacity on (StageBParams::spillList): a
//     checkerboard of small facets gives more than 64 plane segments, a field of pipes more than 64 cylinder labels.
// Same algorithm, statement for statement, as grow_frame_wave / grow_tail (the reference lines are cited there and again below);
// what changes is where the state lives:
//   * bit rows of the grid are arrays of 64-bit words in memory, [row][word]; region growing is the same label propagation, an
//     in-place monotone sweep over the (row, word) items until a sweep changes nothing (a carry bit crosses word boundaries);
//   * cell labels are 16 bits wide; segments, the P x P adjacency matrix of merge_planes (bit rows of ceil(P / 64) words) and the
//     cylinder records live in a scratch slot in HBM sized by the grid's own bound on P (a region needs max(1, uint(0.0065 cells))
//     cells, a plane inside a cylinder candidate six: general_slot_bytes);
//   * the per-cell arrays are carved out of LDS as far as the launch's LDS reaches and out of the scratch slot beyond -- the code
//     addresses both through flat pointers; only the staging buffers of the ordered sums, the histogram and the record window are
//     pinned to LDS (their hand-over relies on LDS executing in issue order);
//   * results go into a CHAIN of records: the frame's own, then records of the handle's spill pool linked by
//     cape_frame_header::next_record, 64 segments / cylinder labels and one boundary slab each.
// The wave loops over its share of the frames (a persistent grid: as many waves as there are scratch slots).
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "cape_grow_common.h"

namespace cape {

namespace {

using Lab = unsigned short;
using u64 = unsigned long long;
constexpr int kGenPend = 64; // recorded regions whose plane fits run together, one lane each

struct GenDims
{
    int C, HC, VC, Wd, items, capSeg, capCyl, chainMax;
    bool cyl;
};

// the working set of one frame (everything but the LDS-pinned buffers)
struct GenPtrs
{
    u64 *U, *L2M, *M2L, *U2M, *M2U; // bit rows of the grid: unassigned + planar; the four directed edge predicates as stage A2 left them
    u64 *act;                       // the region being grown; the merge group's mask / the cylinder's mask in the tail
    u64 *T0, *T1;                   // morphology temporaries of the tail
    Lab* lab;
    unsigned short* list;           // 1 pad + C: cell lists of the recorded regions
    short* bins;
    Lab* cyl;
    Lab* root;                      // tail: merge root of a cell's segment + 1 (0: none)
    unsigned short* ids;
    unsigned char *idmask, *best, *cur;
    float* zc;                      // centre-pixel depths (boundary phase); the prologue parks the cell flags here
    unsigned short* ring;
    Lab* mlab;                      // planeMergeLabels
    uint32_t* segOut;               // per segment: is_output, boundary offset, boundary count
    int *rmin, *rmax;               // per merge root: first / last grid row of its group
    int* recInfo;                   // per record of the chain: boundary points, output planes, kept cylinders, pad
    double* seg;                    // capSeg x 20 f64
    u64* adj;                       // P x ceil(P / 64) connectivity of merge_planes
    cape_cylinder* cylOut;
};

struct GenCarver
{
    unsigned char* lds;
    size_t ldsOff, ldsCap;
    unsigned char* glb;
    size_t glbOff;
    template <typename T> __host__ __device__ T* take(size_t count, bool mayLds)
    {
        const size_t bytes = (count * sizeof(T) + 15) & ~(size_t)15;
        if (mayLds && ldsOff + bytes <= ldsCap)
        {
            T* r = reinterpret_cast<T*>(lds + ldsOff);
            ldsOff += bytes;
            return r;
        }
        T* r = reinterpret_cast<T*>(glb + glbOff);
        glbOff += bytes;
        return r;
    }
};

__host__ __device__ inline size_t gen_fixed_lds_bytes()
{
    // s_stage, s_dist, s_pend, s_pmeta, s_hist
    return (size_t)kChunk * 10 * 8 + (size_t)kChunk * 18 * 8 + (size_t)kGenPend * kSegDoubles * 8 + (size_t)kGenPend * 8 + (size_t)kHistBins * 4;
}

// One walk decides where every array lives; the host runs it with null bases and an LDS capacity of zero to size the slot.
// Order = priority for LDS: the bit rows and the arrays of the seed loop first.
__host__ __device__ inline void gen_carve(GenCarver& cv, const GenDims& d, GenPtrs& P)
{
    const size_t it = (size_t)d.items, C = (size_t)d.C;
    P.U = cv.take<u64>(it, true);
    P.L2M = cv.take<u64>(it, true);
    P.M2L = cv.take<u64>(it, true);
    P.U2M = cv.take<u64>(it, true);
    P.M2U = cv.take<u64>(it, true);
    P.act = cv.take<u64>(it, true);
    P.lab = cv.take<Lab>(C, true);
    P.list = cv.take<unsigned short>(C + 4, true);
    P.bins = cv.take<short>(C, true);
    P.T0 = cv.take<u64>(it, true);
    P.T1 = cv.take<u64>(it, true);
    P.cyl = cv.take<Lab>(d.cyl ? C : 1, true);
    P.root = cv.take<Lab>(C, true);
    P.ids = cv.take<unsigned short>(d.cyl ? C : 1, true);
    P.idmask = cv.take<unsigned char>(d.cyl ? C : 1, true);
    P.best = cv.take<unsigned char>(d.cyl ? C : 1, true);
    P.cur = cv.take<unsigned char>(d.cyl ? C : 1, true);
    P.ring = cv.take<unsigned short>(C, true);
    P.zc = cv.take<float>(C, true);
    P.mlab = cv.take<Lab>((size_t)d.capSeg, true);
    P.rmin = cv.take<int>((size_t)d.capSeg, true);
    P.rmax = cv.take<int>((size_t)d.capSeg, true);
    P.segOut = cv.take<uint32_t>((size_t)d.capSeg * 3, false);
    P.recInfo = cv.take<int>((size_t)d.chainMax * 4, false);
    P.seg = cv.take<double>((size_t)d.capSeg * kSegDoubles, false);
    P.adj = cv.take<u64>((size_t)d.capSeg * (size_t)((d.capSeg + 63) / 64), false);
    P.cylOut = cv.take<cape_cylinder>((size_t)(d.cyl ? d.capCyl : 1), false);
}

__host__ __device__ inline GenDims gen_dims(int cells, int hCells, int vCells, bool cylinders, int capSeg, int capCyl)
{
    GenDims d;
    d.C = cells;
    d.HC = hCells;
    d.VC = vCells;
    d.Wd = (hCells + 63) / 64;
    d.items = d.Wd * vCells;
    d.capSeg = capSeg;
    d.capCyl = capCyl;
    const int m = capSeg > capCyl ? capSeg : capCyl;
    d.chainMax = (m + CAPE_MAX_PLANES - 1) / CAPE_MAX_PLANES + 1;
    d.cyl = cylinders;
    return d;
}

// ---- bit-row helpers: word `it` = (row r, word w) of a [VC][Wd] array; shifts carry across the words of a row
struct RowGeom
{
    int HC, VC, Wd;
    __device__ __forceinline__ u64 width_mask(int w) const // cells of the grid in word w
    {
        const int left = HC - 64 * w;
        return left >= 64 ? ~0ull : ((1ull << left) - 1ull);
    }
};
// (x << 1) of the row: bit c <- cell c - 1
__device__ __forceinline__ u64 row_shl(const u64* a, int it, int w)
{
    return (a[it] << 1) | (w > 0 ? (a[it - 1] >> 63) : 0ull);
}
// (x >> 1) of the row: bit c <- cell c + 1
__device__ __forceinline__ u64 row_shr(const u64* a, int it, int w, int Wd)
{
    return (a[it] >> 1) | (w + 1 < Wd ? (a[it + 1] << 63) : 0ull);
}

template <bool CYL>
__device__ void general_frame(const StageBParams& p, const GenParams& g, const GenDims& D, const GenPtrs& P, double* s_stage, double* s_dist,
                              double* s_pend, u64* s_pmeta, int* s_hist, const int frame, const int lane)
{
    const int C = D.C, HC = D.HC, VC = D.VC, Wd = D.Wd, items = D.items;
    const RowGeom G{HC, VC, Wd};
    const size_t cellBase = (size_t)frame * C;
    const u64 tPhase = p.phaseTicks ? (u64)__builtin_amdgcn_s_memtime() : 0ull;

    // =========================================================================================
    // init_histogram (primitive_detection.cpp:239-265, histogram.hpp:35-62) + the bit rows of the grid
    // =========================================================================================
    for (int i = lane; i < kHistBins; i += 64)
        s_hist[i] = 0;
    CAPE_WAVE_SYNC();
    uint32_t status = 0;
    int nPlanarLocal = 0;
    uint32_t* flagsTmp = reinterpret_cast<uint32_t*>(P.zc); // C words, free until the boundary phase
    {
        constexpr int kBatch = 8; // loads requested together: one memory round trip per 512 cells
        for (int i0 = lane; i0 < C; i0 += 64 * kBatch)
        {
            uint32_t fl[kBatch];
            int bn[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k)
            {
                const int i = i0 + 64 * k;
                fl[k] = p.cell_flags[cellBase + (i < C ? i : 0)];
                bn[k] = p.cell_bins[cellBase + (i < C ? i : 0)];
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k)
            {
                const int i = i0 + 64 * k;
                if (i >= C)
                    continue;
                const uint32_t f = fl[k];
                flagsTmp[i] = f;
                P.lab[i] = 0;
                if (CYL)
                    P.cyl[i] = 0;
                P.bins[i] = (short)bn[k];
                if (f & kFlagPlanar)
                {
                    atomicAdd(&s_hist[bn[k]], 1);
                    ++nPlanarLocal;
                }
                if (f & kFlagNearEdge)
                    status |= CAPE_FRAME_BIN_NEAR_EDGE;
                if (f & kFlagInorder)
                    status |= CAPE_FRAME_INORDER_CELLS;
            }
        }
    }
    CAPE_WAVE_SYNC();

    for (int it = 0; it < items; ++it)
    {
        const int r = it / Wd, w = it - r * Wd;
        const int c = 64 * w + lane;
        const uint32_t f = c < HC ? flagsTmp[r * HC + c] : 0u;
        const u64 bU = __ballot((f & kFlagPlanar) != 0);
        const u64 bL2M = __ballot((f & kFlagLeftToMe) != 0);
        const u64 bM2L = __ballot((f & kFlagMeToLeft) != 0);
        const u64 bU2M = __ballot((f & kFlagUpToMe) != 0);
        const u64 bM2U = __ballot((f & kFlagMeToUp) != 0);
        if (lane == 0)
        {
            P.U[it] = bU;     // unassigned mask (_isUnassignedMask starts as the planar flags)
            P.L2M[it] = bL2M; // parent (r, c-1) -> child (r, c)
            P.M2L[it] = bM2L; // parent (r, c)   -> child (r, c-1)
            P.U2M[it] = bU2M; // parent (r-1, c) -> child (r, c)
            P.M2U[it] = bM2U; // parent (r, c)   -> child (r-1, c)
        }
    }
    // the vertical edges across the tile boundaries of stage A2 (rows k * a2RowsPerTile): evaluated here, exactly as grow_frame_wave
    // does; two (row, word) items per trip so that their twenty loads are in flight together
    {
        const int RPT = p.a2RowsPerTile;
        const int nB = (VC - 1) / RPT;
        const int nQ = nB * Wd;
        constexpr int kPer = 2;
        for (int q0 = 0; q0 < nQ; q0 += kPer)
        {
            double2 m0[kPer], m1[kPer], m2[kPer], m3[kPer], u0[kPer], u1[kPer], u2[kPer], u3[kPer];
            float mt[kPer], ut[kPer];
#pragma unroll
            for (int u = 0; u < kPer; ++u)
            {
                const int q = q0 + u < nQ ? q0 + u : nQ - 1;
                const int r = (q / Wd + 1) * RPT, w = q % Wd;
                const int c = 64 * w + lane;
                const size_t ciMe = cellBase + (size_t)r * HC + (c < HC ? c : 0), ciUp = ciMe - HC;
                const double2* pm = reinterpret_cast<const double2*>(p.cell_plane + ciMe * kPlaneStride);
                const double2* pu = reinterpret_cast<const double2*>(p.cell_plane + ciUp * kPlaneStride);
                m0[u] = pm[0]; m1[u] = pm[1]; m2[u] = pm[2]; m3[u] = pm[3];
                u0[u] = pu[0]; u1[u] = pu[1]; u2[u] = pu[2]; u3[u] = pu[3];
                mt[u] = p.cell_tol[ciMe];
                ut[u] = p.cell_tol[ciUp];
            }
#pragma unroll
            for (int u = 0; u < kPer; ++u)
            {
                const int q = q0 + u;
                const int r = (q / Wd + 1) * RPT, w = q % Wd;
                const bool on = q < nQ && 64 * w + lane < HC;
                const bool u2m = on & can_be_merged(u0[u].x, u0[u].y, u1[u].x, u1[u].y, m0[u].x, m0[u].y, m1[u].x, m2[u].x, m2[u].y, m3[u].x,
                                                    (double)mt[u], p.cosMerge);
                const bool m2u = on & can_be_merged(m0[u].x, m0[u].y, m1[u].x, m1[u].y, u0[u].x, u0[u].y, u1[u].x, u2[u].x, u2[u].y, u3[u].x,
                                                    (double)ut[u], p.cosMerge);
                const u64 bU2M = __ballot(u2m), bM2U = __ballot(m2u);
                if (lane == 0 && q < nQ)
                {
                    P.U2M[r * Wd + w] = bU2M;
                    P.M2U[r * Wd + w] = bM2U;
                }
            }
        }
    }
    const int nPlanar = wave_sum_i32(nPlanarLocal);
    CAPE_WAVE_SYNC();

    // =========================================================================================
    // grow_planes_and_cylinders (primitive_detection.cpp:267-310)
    // =========================================================================================
    int untried = nPlanar;
    int nSeg = 0, nCylLabels = 0, nCylFits = 0, rngPos = 0, nSeeds = 0;
    const double* sumsBase = p.cell_sums + cellBase * kSumStride;
    const double* mseBase = p.cell_mse + cellBase;
    const int maxSeedIters = 4 * C + 1024;
    int pendCount = 0, listTop = 0;
    bool moreSeeds = true, stopAll = false;

    for (;;)
    {
        if (moreSeeds && !(untried > 0 && nSeeds < maxSeedIters))
            moreSeeds = false;
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
                    moreSeeds = false;
                    break;
                }
                const int bin = 0xFFFF - (int)(key & 0xFFFFu);

                // ---- candidates = cells with _bins == bin ; seed = first strict minimum of MSE (:285-298)
                int candLocal = 0;
                u64 bestLocal = ~0ull;
                int bestIdxLocal = 0x7FFFFFFF;
                {
                    constexpr int kBatch = 8;
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
                            if (i < C && P.bins[i] == (short)bin)
                            {
                                ++candLocal;
                                const u64 mb = (u64)__double_as_longlong(mv[k]);
                                if (mb < bestLocal)
                                {
                                    bestLocal = mb;
                                    bestIdxLocal = i;
                                }
                            }
                        }
                    }
                }
                const int cand = wave_sum_i32(candLocal);
                if (cand < p.planeSeedCount || cand == 0)
                {
                    moreSeeds = false;
                    break;
                }
                const u64 bestAll = wave_min_u64(bestLocal);
                const unsigned idxKey = (bestLocal == bestAll) ? (unsigned)(0x7FFFFFFF - bestIdxLocal) : 0u;
                const int seed = 0x7FFFFFFF - (int)wave_max_u32(idxKey);
                if (__longlong_as_double((long long)bestAll) >= kDblMax)
                {
                    moreSeeds = false; // "invalid seed" (:299-304)
                    status |= CAPE_FRAME_INVALID_SEED;
                    break;
                }
                if (lane == 0 && p.seed_sequence && nSeeds < C)
                    p.seed_sequence[cellBase + nSeeds] = (uint16_t)seed;
                ++nSeeds;

                // ---- grow_plane_segment_at_seed (:312-389)
                const int sy = seed / HC, sx = seed - sy * HC;
                const double2* spl = reinterpret_cast<const double2*>(p.cell_plane + (cellBase + seed) * kPlaneStride);
                const double2 sp0 = spl[0], sp1 = spl[1], sp2 = spl[2], sp3 = spl[3];
                const float stolf = p.cell_tol[cellBase + seed];
                const int seedItem = sy * Wd + (sx >> 6);
                const bool seedUnassigned = (P.U[seedItem] >> (sx & 63)) & 1ull;

                // ---- region_growing (:778-818) as label propagation on bit rows in memory
                for (int it = lane; it < items; it += 64)
                    P.act[it] = 0ull;
                CAPE_WAVE_SYNC();
                bool seedOK;
                {
                    // the seed's own test: newPlaneSegment(planeToGrow) is a copy, and the copy re-normalises the normal
                    // (plane_coordinates.hpp:24-27) before can_be_merged compares it with the original
                    const double snx = sp0.x, sny = sp0.y, snz = sp1.x, sd = sp1.y, scx = sp2.x, scy = sp2.y, scz = sp3.x;
                    double pnx = snx, pny = sny, pnz = snz;
                    normalize3(pnx, pny, pnz);
                    seedOK = can_be_merged(pnx, pny, pnz, sd, snx, sny, snz, scx, scy, scz, (double)stolf, p.cosMerge);
                }
                if (seedUnassigned && seedOK)
                {
                    if (lane == 0)
                        P.act[seedItem] = 1ull << (sx & 63);
                    CAPE_WAVE_SYNC();
                    for (;;)
                    {
                        bool changed = false;
                        for (int it = lane; it < items; it += 64)
                        {
                            const int r = it / Wd, w = it - r * Wd;
                            const u64 a0 = P.act[it];
                            u64 a = a0;
                            const u64 Uw = P.U[it], el = P.L2M[it], m2l = P.M2L[it];
                            const u64 cinL = w > 0 ? (P.act[it - 1] >> 63) : 0ull;                           // the cell left of bit 0
                            const u64 cinR = w + 1 < Wd ? ((P.act[it + 1] & P.M2L[it + 1]) & 1ull) : 0ull;    // the cell right of bit 63, reaching back
                            for (;;)
                            {
                                const u64 na = a | (Uw & ((((a << 1) | cinL) & el) | (((a & m2l) >> 1) | (cinR << 63))));
                                if (na == a)
                                    break;
                                a = na;
                            }
                            const u64 up = r > 0 ? P.act[it - Wd] : 0ull;
                            const u64 dn = r + 1 < VC ? (P.act[it + Wd] & P.M2U[it + Wd]) : 0ull;
                            a |= Uw & ((up & P.U2M[it]) | dn);
                            if (a != a0)
                            {
                                P.act[it] = a; // in place: the sweep is monotone, a neighbour sees this word now or next sweep
                                changed = true;
                            }
                        }
                        CAPE_WAVE_SYNC();
                        if (!__any(changed))
                            break;
                    }
                }

                // ---- activated cell list in ascending cell index (row-major), appended at listTop
                unsigned short* rlist = P.list + 1 + listTop;
                int total = 0;
                for (int it0 = 0; it0 < items; it0 += 64)
                {
                    const int it = it0 + lane;
                    const u64 m0 = it < items ? P.act[it] : 0ull;
                    const int cnt = __popcll(m0);
                    const int incl = wave_scan_i32(cnt);
                    const int tot = (int)readlane_u32((unsigned)incl, 63);
                    if (tot == 0)
                        continue;
                    const int r = it / Wd, w = it - r * Wd;
                    const int cell0 = r * HC + 64 * w;
                    int pos = total + incl - cnt;
                    u64 m = m0;
                    while (m)
                    {
                        const int c = __ffsll((long long)m) - 1;
                        rlist[pos++] = (unsigned short)(cell0 + c);
                        m &= m - 1;
                    }
                    total += tot;
                }
                CAPE_WAVE_SYNC();

                // ---- expand_segment over activated cells in ascending order (:341-360): lanes 0..8 own one sum each, lane 9 the
                //      point count.  The seed's own sums are counted twice (copy :325 + expand of the seed itself).
                const int ql = lane < 10 ? lane : 0;
                const unsigned short parkedOver = rlist[-1];
                CAPE_WAVE_SYNC();
                if (lane == 0)
                    rlist[-1] = (unsigned short)seed;
                CAPE_WAVE_SYNC();
                double acc = 0.0;
                staged_for_each<5, CAPE_STAGE_DEPTH_MAIN>(
                        total + 1, sumsBase, kSumStride, 0, [&](int e) { return (int)rlist[e - 1]; }, s_stage, lane,
                        [&](int, const double* rec) { return rec[ql]; }, [&](int, double v) { acc += v; });
                CAPE_WAVE_SYNC();
                if (lane == 0)
                    rlist[-1] = parkedOver;
                CAPE_WAVE_SYNC();

                // ---- Histogram::remove_point for every activated cell (histogram.hpp:103-113), _isUnassignedMask = false
                for (int i = lane; i < total; i += 64)
                {
                    const int cidx = rlist[i];
                    atomicSub(&s_hist[P.bins[cidx]], 1);
                    P.bins[cidx] = 1; // quirk: 1, not -1
                }
                for (int it = lane; it < items; it += 64)
                    P.U[it] &= ~P.act[it];
                CAPE_WAVE_SYNC();
                if (lane == 0 && s_hist[1] < 0)
                    s_hist[1] = 0; // "if != 0: -= 1" saturates; only bin 1 can be over-decremented
                untried -= total;

                if (total == 0 || total < p.minCellActivated)
                {
                    if (lane == 0)
                    {
                        const int b = P.bins[seed];
                        if (s_hist[b] != 0)
                            s_hist[b] -= 1;
                        P.bins[seed] = 1;
                    }
                    CAPE_WAVE_SYNC();
                    break; // region dropped: nothing recorded, its list slots are reused
                }
                // ---- record the region: sums + count into the window slot, list kept
                if (lane < 10)
                    s_pend[pendCount * kSegDoubles + lane] = acc;
                if (lane == 0)
                    s_pmeta[pendCount] = (u64)(unsigned)listTop | ((u64)(unsigned)total << 32);
                listTop += total;
                ++pendCount;
                CAPE_WAVE_SYNC();
            } while (0);

        if (moreSeeds && pendCount < kGenPend)
            continue;
        if (pendCount > 0)
        {
            // ---- fit_plane (plane_segment.cpp:232-284) of every recorded region, one lane per region
            if (lane < pendCount)
            {
                double* slot = s_pend + lane * kSegDoubles;
                double S[9];
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    S[k] = slot[k];
                PlaneFit f;
                fit_plane(S, (uint32_t)slot[9], f);
                // add_plane_segment_to_features (:391-411): push_back copies the segment (one more normalisation)
                double nx = f.nx, ny = f.ny, nz = f.nz;
                normalize3(nx, ny, nz);
                slot[10] = nx; slot[11] = ny; slot[12] = nz; slot[13] = f.d;
                slot[14] = f.cx; slot[15] = f.cy; slot[16] = f.cz;
                slot[17] = f.mse; slot[18] = f.score; slot[19] = f.planar ? 1.0 : 0.0;
            }
            CAPE_WAVE_SYNC();
            for (int j = 0; j < pendCount && !stopAll; ++j)
            {
                SegRec ns;
                seg_load(s_pend + j * kSegDoubles, ns);
                const u64 meta = s_pmeta[j];
                const int roff = (int)(unsigned)meta, total = (int)(meta >> 32);
                if (ns.planar == 0.0)
                {
                    status_count_not_planar(status); // "Plane segment is not planar after merge" (:374)
                    continue;
                }
                if (ns.score > 100)
                {
                    if (nSeg >= D.capSeg)
                    {
                        status |= CAPE_FRAME_PLANE_OVERFLOW; // cannot happen: capSeg is the grid's own bound (general_slot_bytes)
                        stopAll = true;
                        break;
                    }
                    if (lane == 0)
                        seg_store(P.seg + (size_t)nSeg * kSegDoubles, ns);
                    ++nSeg;
                    for (int i = lane; i < total; i += 64)
                        P.lab[P.list[1 + roff + i]] = (Lab)nSeg;
                    CAPE_WAVE_SYNC();
                }
                else if (CYL && total > 5)
                {
                    // cylinder_fitting (:478-501) ; CYL == false is the "plane-only" mode (region dropped, cells stay consumed)
                    CylCtxT<Lab> cc;
                    cc.p = &p;
                    cc.lane = lane;
                    cc.cellBase = cellBase;
                    cc.C = C;
                    cc.s_list = P.list + 1 + roff;
                    cc.total = total;
                    cc.s_dist = s_dist;
                    cc.s_ids = P.ids;
                    cc.s_idmask = P.idmask;
                    cc.s_cur = P.cur;
                    cc.s_best = P.best;
                    cc.scratch = p.cylScratch + cellBase * kCylStride;
                    cc.s_stage = s_stage;
                    cc.s_seg = P.seg;
                    cc.s_lab = P.lab;
                    cc.s_cyl = P.cyl;
                    cc.cylOut = P.cylOut;
                    cc.maxCylinders = D.capCyl;
                    cc.maxPlanes = D.capSeg;
                    cc.dbg = nullptr;
                    bool overflow = false;
                    cylinder_fitting(cc, nSeg, nCylLabels, nCylFits, rngPos, status, overflow);
                    ++nCylFits;
                    CAPE_WAVE_SYNC();
                    if (overflow)
                    {
                        status |= CAPE_FRAME_PLANE_OVERFLOW; // cannot happen (see above)
                        stopAll = true;
                        break;
                    }
                }
            }
            pendCount = 0;
            if (stopAll)
                break;
        }
        if (!moreSeeds)
            break;
    }
    if (untried > 0 && nSeeds >= maxSeedIters)
        status |= CAPE_FRAME_SEED_LIMIT;

    u64 tMerge = 0ull;
    if (p.phaseTicks)
    {
        tMerge = __builtin_amdgcn_s_memtime();
        if (lane == 0)
            atomicAdd(&p.phaseTicks[(size_t)frame * 4 + 0], tMerge - tPhase);
    }

    // =========================================================================================
    // the chain of records this frame needs: its own + spill records of the handle's pool
    // =========================================================================================
    const int recNeeded = [&]() {
        const int a = (nSeg + CAPE_MAX_PLANES - 1) / CAPE_MAX_PLANES, b = (nCylLabels + CAPE_MAX_CYLINDERS - 1) / CAPE_MAX_CYLINDERS;
        const int m = a > b ? a : b;
        return m > 1 ? m : 1;
    }();
    int recHave = 1;
    int firstSpill = 0;
    if (recNeeded > 1)
    {
        unsigned got = 0;
        if (lane == 0)
            got = atomicAdd(g.spillAlloc, (unsigned)(recNeeded - 1));
        got = readlane_u32(got, 0);
        firstSpill = (int)(got < (unsigned)g.poolCapacity ? got : (unsigned)g.poolCapacity);
        const int avail = g.poolCapacity - firstSpill;
        recHave = 1 + (recNeeded - 1 < avail ? recNeeded - 1 : avail);
    }
    const int segCap = recHave * CAPE_MAX_PLANES, cylCap = recHave * CAPE_MAX_CYLINDERS;
    const int nSegOut = nSeg < segCap ? nSeg : segCap;          // segments that reach the records
    const int nCylLabOut = nCylLabels < cylCap ? nCylLabels : cylCap;
    if (nSegOut < nSeg)
        status |= CAPE_FRAME_PLANE_OVERFLOW; // the pool ran out (cape_config.spill_records)
    if (nCylLabOut < nCylLabels)
        status |= CAPE_FRAME_CYL_OVERFLOW;
    auto record_of = [&](int k) -> cape_frame_record* { return k == 0 ? p.records + frame : g.poolRecords + (firstSpill + k - 1); };
    auto boundary_of = [&](int k) -> double* {
        return k == 0 ? p.boundary + (size_t)frame * p.boundaryCapacity * 3 : g.poolBoundary + (size_t)(firstSpill + k - 1) * p.boundaryCapacity * 3;
    };
    auto index_of = [&](int k) -> int { return g.poolBase + firstSpill + k - 1; };

    // =========================================================================================
    // merge_planes (:503-560) with get_connected_components_matrix (:736-776)
    // =========================================================================================
    const int adjWords = (nSeg + 63) / 64;
    for (int i = lane; i < nSeg * adjWords; i += 64)
        P.adj[i] = 0ull;
    for (int i = lane; i < nSeg; i += 64)
        P.mlab[i] = (Lab)i;
    CAPE_WAVE_SYNC();
    for (int i = lane; i < C; i += 64)
    {
        const int r = i / HC, c = i - r * HC;
        if (r >= VC - 1 || c >= HC - 1)
            continue; // last row / last column never act as sources
        const int a = P.lab[i];
        if (a <= 0)
            continue;
        const int b = P.lab[i + 1];
        const int dwn = P.lab[i + HC];
        if (b > 0 && a != b)
        {
            atomicOr(&P.adj[(size_t)(a - 1) * adjWords + ((b - 1) >> 6)], 1ull << ((b - 1) & 63));
            atomicOr(&P.adj[(size_t)(b - 1) * adjWords + ((a - 1) >> 6)], 1ull << ((a - 1) & 63));
        }
        if (dwn > 0 && a != dwn)
        {
            atomicOr(&P.adj[(size_t)(a - 1) * adjWords + ((dwn - 1) >> 6)], 1ull << ((dwn - 1) & 63));
            atomicOr(&P.adj[(size_t)(dwn - 1) * adjWords + ((a - 1) >> 6)], 1ull << ((a - 1) & 63));
        }
    }
    CAPE_WAVE_SYNC();
    for (int row = 0; row < nSeg; ++row)
    {
        const int planeId = P.mlab[row];
        SegRec A;
        seg_load(P.seg + (size_t)planeId * kSegDoubles, A);
        if (A.planar == 0.0)
            continue;
        bool expanded = false;
        for (int wq = (row + 1) >> 6; wq < adjWords; ++wq)
        {
            u64 conn = P.adj[(size_t)row * adjWords + wq];
            if (wq == ((row + 1) >> 6) && ((row + 1) & 63))
                conn &= ~0ull << ((row + 1) & 63); // columns row + 1 ... only
            while (conn)
            {
                const int col = 64 * wq + (__ffsll((long long)conn) - 1);
                conn &= conn - 1;
                if (col >= nSeg)
                    break;
                SegRec B;
                seg_load(P.seg + (size_t)col * kSegDoubles, B);
                if (B.planar == 0.0)
                    continue;
                // planeToExpand keeps its (stale) normal / d inside the row loop
                if (can_be_merged(A.nx, A.ny, A.nz, A.d, B.nx, B.ny, B.nz, B.cx, B.cy, B.cz, 50.0, p.cosMerge))
                {
                    A.S[0] += B.S[0]; A.S[1] += B.S[1]; A.S[2] += B.S[2];
                    A.S[3] += B.S[3]; A.S[4] += B.S[4]; A.S[5] += B.S[5];
                    A.S[6] += B.S[6]; A.S[7] += B.S[7]; A.S[8] += B.S[8];
                    A.n += B.n;
                    if (lane == 0)
                        P.mlab[col] = (Lab)planeId;
                    expanded = true;
                }
            }
        }
        if (expanded)
        {
            PlaneFit f;
            fit_plane(A.S, (uint32_t)A.n, f);
            A.cx = f.cx; A.cy = f.cy; A.cz = f.cz;
            A.planar = f.planar ? 1.0 : 0.0;
            if (f.planar) // on a degenerate refit fit_plane returns before touching normal / d / mse / score
            {
                A.nx = f.nx; A.ny = f.ny; A.nz = f.nz; A.d = f.d;
                A.mse = f.mse; A.score = f.score;
            }
            if (lane == 0)
                seg_store(P.seg + (size_t)planeId * kSegDoubles, A);
        }
        CAPE_WAVE_SYNC();
    }

    u64 tRefine = 0ull;
    if (p.phaseTicks)
    {
        tRefine = __builtin_amdgcn_s_memtime();
        if (lane == 0)
            atomicAdd(&p.phaseTicks[(size_t)frame * 4 + 1], tRefine - tMerge);
    }

    // =========================================================================================
    // add_planes_to_primitives (:562-648) + compute_plane_segment_boundary (:650-703)
    // =========================================================================================
    // per cell: the merge root of its segment (+ 1), per root: the grid rows its group spans; centre-pixel depths
    for (int i = lane; i < nSeg; i += 64)
    {
        P.rmin[i] = 0x7FFFFFFF;
        P.rmax[i] = -1;
    }
    CAPE_WAVE_SYNC();
    if (nSeg > 0)
    {
        constexpr int kBatch = 8;
        for (int i0 = 0; i0 < C; i0 += 64 * kBatch)
        {
            float z[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k)
            {
                const int i = i0 + lane + 64 * k;
                z[k] = p.cell_aux[cellBase + (i < C ? i : 0)].zc; // depthImage(centerY, centerX) of compute_plane_segment_boundary
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k)
            {
                const int i = i0 + lane + 64 * k;
                const bool in = i < C;
                const int l = in ? (int)P.lab[i] : 0;
                const int rt = l > 0 ? (int)P.mlab[l - 1] + 1 : 0;
                const int r = i / HC;
                if (in)
                {
                    P.root[i] = (Lab)rt;
                    P.zc[i] = z[k]; // (the parked cell flags are dead by now)
                }
                // one lane per run of equal roots books the row (runs are the rule: neighbouring cells share their plane)
                const int prevRt = __shfl_up(rt, 1), prevR = __shfl_up(r, 1);
                if (rt > 0 && (lane == 0 || prevRt != rt || prevR != r))
                {
                    atomicMin(&P.rmin[rt - 1], r);
                    atomicMax(&P.rmax[rt - 1], r);
                }
            }
        }
    }
    CAPE_WAVE_SYNC();

    int nBoundary = 0;   // points in the current record's slab
    int curRec = 0;
    double* bnd = boundary_of(0);
    for (int pi = 0; pi < nSegOut; ++pi)
    {
        const int k = pi / CAPE_MAX_PLANES;
        if (k != curRec)
        {
            if (lane == 0)
                P.recInfo[curRec * 4 + 0] = nBoundary < p.boundaryCapacity ? nBoundary : p.boundaryCapacity;
            curRec = k;
            bnd = boundary_of(k);
            nBoundary = 0;
        }
        const int mlabel = P.mlab[pi];
        const double* segp = P.seg + (size_t)pi * kSegDoubles;
        uint32_t isOutput = 0, bOff = (uint32_t)nBoundary, bCnt = 0;
        if (mlabel == pi && segp[19] != 0.0)
        {
            const double Anx = segp[10], Any = segp[11], Anz = segp[12], Ad = segp[13], Amse = segp[17];
            const int rLo = P.rmin[pi], rHi = P.rmax[pi];
            if (rHi >= rLo)
            {
                // rows [r0, r1] can hold ring cells: the group's rows and one more on either side (the 3x3 dilation)
                const int r0 = rLo > 0 ? rLo - 1 : 0, r1 = rHi + 1 < VC ? rHi + 1 : VC - 1;
                const int m0 = r0 > 0 ? r0 - 1 : 0, m1 = r1 + 1 < VC ? r1 + 1 : VC - 1; // mask rows the ring rows read
                // the group's mask M (= P.act): a cell belongs if its segment's merge root is pi (j >= pi follows: a merge label
                // never exceeds its own index)

                for (int it = m0 * Wd; it < (m1 + 1) * Wd; ++it)
                {
                    const int r = it / Wd, w = it - r * Wd;
                    const int c = 64 * w + lane;
                    const int rt = c < HC ? (int)P.root[r * HC + c] : 0;
                    const u64 bm = __ballot(rt == pi + 1);
                    if (lane == 0)
                        P.act[it] = bm;
                }
                CAPE_WAVE_SYNC();
                // erode, 3x3 cross, BORDER_CONSTANT 0 ; dilate, 3x3 square, border ignored ; ring = dilated - eroded -> P.T0
                for (int it = r0 * Wd + lane; it < (r1 + 1) * Wd; it += 64)
                {
                    const int r = it / Wd, w = it - r * Wd;
                    const u64 wm = G.width_mask(w);
                    const u64 M = P.act[it];
                    const bool hasUp = r > 0, hasDn = r + 1 < VC;
                    const u64 Mup = hasUp ? P.act[it - Wd] : 0ull, Mdn = hasDn ? P.act[it + Wd] : 0ull;
                    const u64 ero = M & row_shl(P.act, it, w) & row_shr(P.act, it, w, Wd) & Mup & Mdn;
                    u64 dil = (M | row_shl(P.act, it, w) | row_shr(P.act, it, w, Wd)) & wm;
                    if (hasUp)
                        dil |= (Mup | row_shl(P.act, it - Wd, w) | row_shr(P.act, it - Wd, w, Wd)) & wm;
                    if (hasDn)
                        dil |= (Mdn | row_shl(P.act, it + Wd, w) | row_shr(P.act, it + Wd, w, Wd)) & wm;
                    P.T0[it] = dil & ~ero;
                }
                CAPE_WAVE_SYNC();
                // ring cells in row-major order -> P.ring
                int R = 0;
                for (int it0 = r0 * Wd; it0 < (r1 + 1) * Wd; it0 += 64)
                {
                    const int it = it0 + lane;
                    const u64 rr = it < (r1 + 1) * Wd ? P.T0[it] : 0ull;
                    const int cnt = __popcll(rr);
                    const int incl = wave_scan_i32(cnt);
                    const int tot = (int)readlane_u32((unsigned)incl, 63);
                    if (tot == 0)
                        continue;
                    const int r = it / Wd, w = it - r * Wd;
                    const int cell0 = r * HC + 64 * w;
                    int pos = R + incl - cnt;
                    u64 m = rr;
                    while (m)
                    {
                        const int c = __ffsll((long long)m) - 1;
                        P.ring[pos++] = (unsigned short)(cell0 + c);
                        m &= m - 1;
                    }
                    R += tot;
                }
                CAPE_WAVE_SYNC();
                const double maxBoundaryDistance = 3 * sqrt(Amse);
                for (int j0 = 0; j0 < R; j0 += 64)
                {
                    const int j = j0 + lane;
                    bool hit = false;
                    double px = 0, py = 0, pz = 0;
                    const int cell = P.ring[j < R ? j : R - 1];
                    const int r = cell / HC, c = cell - r * HC;
                    const double ac = p.acol[c * kCell + kCell / 2], br = p.brow[r * kCell + kCell / 2];
                    const double dpt = (double)P.zc[cell]; // depthImage(centerY, centerX), staged by stage A
                    if (j < R && dpt > 0)
                    {
                        px = dpt * ac;
                        py = dpt * br;
                        pz = dpt;
                        const double dist = dot3(Anx, Any, Anz, px, py, pz) + Ad;
                        hit = fabs(dist) < maxBoundaryDistance;
                    }
                    const u64 hb = __ballot(hit);
                    if (hit)
                    {
                        const int pos = nBoundary + __popcll(hb & ((1ull << lane) - 1ull));
                        if (pos < p.boundaryCapacity)
                        {
                            bnd[(size_t)pos * 3 + 0] = px;
                            bnd[(size_t)pos * 3 + 1] = py;
                            bnd[(size_t)pos * 3 + 2] = pz;
                        }
                    }
                    nBoundary += __popcll(hb);
                    bCnt += (uint32_t)__popcll(hb);
                }
                CAPE_WAVE_SYNC();
            }
            if (nBoundary > p.boundaryCapacity)
                status |= CAPE_FRAME_BOUNDARY_OVERFLOW;
            if (bCnt >= 3)
                isOutput = 1;
            else
            {
                // rejected plane: its candidate points are dropped (reference: `continue` before emplace_back)
                nBoundary = (int)bOff;
                bCnt = 0;
            }
        }
        if (lane == 0)
        {
            P.segOut[pi * 3 + 0] = isOutput;
            P.segOut[pi * 3 + 1] = bOff;
            P.segOut[pi * 3 + 2] = bCnt;
        }
    }
    if (lane == 0)
        P.recInfo[curRec * 4 + 0] = nBoundary < p.boundaryCapacity ? nBoundary : p.boundaryCapacity;
    for (int k = lane; k < recHave; k += 64)
        if (k > curRec)
            P.recInfo[k * 4 + 0] = 0;
    CAPE_WAVE_SYNC();

    // the records of all segments, 64 at a time, lane j <- segment j0 + j: Plane::_parametrization's extra normalisation
    // (shape_primitives.cpp:49) and get_point_cloud_covariance (plane_segment.cpp:192-203) run lane parallel
    for (int j0 = 0; j0 < nSegOut; j0 += 64)
    {
        const int j = j0 + lane;
        uint32_t myOut = 0;
        if (j < nSegOut)
        {
            SegRec A;
            seg_load(P.seg + (size_t)j * kSegDoubles, A);
            myOut = P.segOut[j * 3 + 0];
            double onx = 0, ony = 0, onz = 0;
            double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (myOut)
            {
                onx = A.nx; ony = A.ny; onz = A.nz;
                normalize3(onx, ony, onz);
                inverse3_sym(A.S, cov);
            }
            cape_plane_segment* o = &record_of(j / CAPE_MAX_PLANES)->segments[j % CAPE_MAX_PLANES];
            o->normal[0] = A.nx; o->normal[1] = A.ny; o->normal[2] = A.nz;
            o->d = A.d;
            o->centroid[0] = A.cx; o->centroid[1] = A.cy; o->centroid[2] = A.cz;
            o->mse = A.mse;
            o->score = A.score;
#pragma unroll
            for (int k = 0; k < 9; ++k)
                o->sums[k] = A.S[k];
            o->out_normal[0] = onx; o->out_normal[1] = ony; o->out_normal[2] = onz;
#pragma unroll
            for (int k = 0; k < 9; ++k)
                o->cov[k] = cov[k];
            o->point_count = (uint32_t)A.n;
            o->merge_label = (uint32_t)P.mlab[j];
            o->planar = A.planar != 0.0 ? 1u : 0u;
            o->is_output = myOut;
            o->boundary_offset = P.segOut[j * 3 + 1];
            o->boundary_count = P.segOut[j * 3 + 2];
        }
        const int outs = __popcll(__ballot(myOut != 0)); // CAPE_MAX_PLANES = 64 = the wave: this pass is one record
        if (lane == 0)
            P.recInfo[(j0 / CAPE_MAX_PLANES) * 4 + 1] = outs;
    }
    for (int k = lane; k < recHave; k += 64)
    {
        if (k * CAPE_MAX_PLANES >= nSegOut)
            P.recInfo[k * 4 + 1] = 0;
        P.recInfo[k * 4 + 2] = 0;
    }
    CAPE_WAVE_SYNC();

    // =========================================================================================
    // add_cylinders_to_primitives (:705-734): open (dilate, erode) + erode with the 3x3 cross, default borders
    // =========================================================================================
    if (CYL)
        for (int ci = 0; ci < nCylLabOut; ++ci)
        {

            for (int it = 0; it < items; ++it)
            {
                const int r = it / Wd, w = it - r * Wd;
                const int c = 64 * w + lane;
                const int l = c < HC ? (int)P.cyl[r * HC + c] : 0;
                const u64 bm = __ballot(l == ci + 1);
                if (lane == 0)
                    P.act[it] = bm;
            }
            CAPE_WAVE_SYNC();
            // dilate: act -> T0
            for (int it = lane; it < items; it += 64)
            {
                const int r = it / Wd, w = it - r * Wd;
                const u64 x = P.act[it];
                const u64 up = r > 0 ? P.act[it - Wd] : 0ull, dn = r + 1 < VC ? P.act[it + Wd] : 0ull;
                P.T0[it] = (x | row_shl(P.act, it, w) | row_shr(P.act, it, w, Wd) | up | dn) & G.width_mask(w);
            }
            CAPE_WAVE_SYNC();
            // erode (outside the grid never erodes: morphologyDefaultBorderValue): src -> dst
            auto erode = [&](const u64* src, u64* dst) {
                for (int it = lane; it < items; it += 64)
                {
                    const int r = it / Wd, w = it - r * Wd;
                    const u64 wm = G.width_mask(w);
                    const u64 x = src[it];
                    const u64 up = r > 0 ? src[it - Wd] : wm, dn = r + 1 < VC ? src[it + Wd] : wm;
                    const u64 firstCol = w == 0 ? 1ull : 0ull;
                    const u64 lastCol = w == Wd - 1 ? (1ull << ((HC - 1) & 63)) : 0ull;
                    dst[it] = x & (row_shl(src, it, w) | firstCol) & (row_shr(src, it, w, Wd) | lastCol) & up & dn & wm;
                }
                CAPE_WAVE_SYNC();
            };
            erode(P.T0, P.T1);
            erode(P.T1, P.T0);
            int onesLocal = 0;
            for (int it = lane; it < items; it += 64)
                onesLocal += __popcll(P.T0[it]);
            const int ones = wave_sum_i32(onesLocal);
            const bool kept = ones > 0 && ones < C; // max > 0 and min < max
            if (lane == 0)
            {
                P.cylOut[ci].kept = kept ? 1u : 0u;
                if (kept)
                    P.recInfo[(ci / CAPE_MAX_CYLINDERS) * 4 + 2] += 1;
            }
            CAPE_WAVE_SYNC();
        }
    // cylinder records into the chain
    if (CYL)
        for (int ci = lane; ci < nCylLabOut; ci += 64)
            record_of(ci / CAPE_MAX_CYLINDERS)->cylinders[ci % CAPE_MAX_CYLINDERS] = P.cylOut[ci];

    // =========================================================================================
    // label grids + the headers of the chain
    // =========================================================================================
    for (int i = lane; i < C; i += 64)
    {
        p.plane_labels[cellBase + i] = (int32_t)P.lab[i];
        p.cyl_labels[cellBase + i] = CYL ? (int32_t)P.cyl[i] : 0;
    }
    status = wave_or_u32(status);
    CAPE_WAVE_SYNC();
    if (lane == 0)
    {
        int planesFromHere = 0, cylFromHere = 0;
        for (int k = recHave - 1; k >= 0; --k)
        {
            planesFromHere += P.recInfo[k * 4 + 1];
            cylFromHere += P.recInfo[k * 4 + 2];
            cape_frame_header& hd = record_of(k)->header;
            const int segLeft = nSegOut - k * CAPE_MAX_PLANES, cylLeft = nCylLabOut - k * CAPE_MAX_CYLINDERS;
            hd.n_plane_segments = segLeft > 0 ? segLeft : 0;
            hd.n_planes = planesFromHere;
            hd.n_cylinder_labels = cylLeft > 0 ? cylLeft : 0;
            hd.n_cylinders = cylFromHere;
            hd.n_boundary_points = P.recInfo[k * 4 + 0];
            hd.n_seeds = nSeeds;
            hd.status = status;
            hd.n_planar_cells = nPlanar;
            hd.next_record = k + 1 < recHave ? index_of(k + 1) : -1;
            hd.segment_base = k * CAPE_MAX_PLANES;
        }
        atomicAdd(g.genFrames, 1u);
        if (p.phaseTicks)
            atomicAdd(&p.phaseTicks[(size_t)frame * 4 + 2], (u64)__builtin_amdgcn_s_memtime() - tRefine);
    }
    CAPE_WAVE_SYNC();
}

} // namespace

template <bool CYL> __global__ __launch_bounds__(64, 1) void cape_grow_general_kernel(StageBParams p, GenParams g, int nFrames)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int work = g.allFrames ? nFrames : (int)p.spillList[0];
    if (work > 0)
    {
        // LDS-pinned buffers (their users order LDS traffic with compiler fences only)
        double* s_stage = reinterpret_cast<double*>(smem);
        double* s_dist = s_stage + kChunk * 10;
        double* s_pend = s_dist + kChunk * 18;
        u64* s_pmeta = reinterpret_cast<u64*>(s_pend + kGenPend * kSegDoubles);
        int* s_hist = reinterpret_cast<int*>(s_pmeta + kGenPend);
        const GenDims D = gen_dims(p.cells, p.hCells, p.vCells, CYL, g.capSeg, g.capCyl);
        GenCarver cv;
        cv.lds = smem;
        cv.ldsOff = (gen_fixed_lds_bytes() + 15) & ~(size_t)15;
        cv.ldsCap = (size_t)g.ldsBytes;
        cv.glb = g.scratch + (size_t)blockIdx.x * g.slotBytes;
        cv.glbOff = 0;
        GenPtrs P;
        gen_carve(cv, D, P);
        for (int k = (int)blockIdx.x; k < work; k += (int)gridDim.x)
        {
            const int frame = g.allFrames ? k : (int)p.spillList[1 + k];
            general_frame<CYL>(p, g, D, P, s_stage, s_dist, s_pend, s_pmeta, s_hist, frame, lane);
        }
    }
    // the one-frame chain (results in pinned host memory, the host spins on a sequence number): this kernel is the last of the call
    // whenever it had something to do, so its last wave stores the number (see cape_grow_kernel)
    if (p.doneFlag && work > 0 && lane == 0)
    {
        __threadfence_system();
        if (atomicAdd(p.doneCounter, 1u) == gridDim.x - 1u)
        {
            atomicExch(p.doneCounter, 0u);
            __hip_atomic_store(p.doneFlag, p.doneSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Most plane segments / cylinder labels a frame can hold: every segment consumes cells of its own -- a grown region at least
// max(1, minCellActivated) (primitive_detection.cpp:362-363), a plane found inside a cylinder candidate at least six inliers
// (cylinder_segment.cpp:154), a cylinder label likewise six.
size_t general_slot_bytes(int cells, int hCells, int vCells, bool cylinders, int minCellActivated, int* capSegOut, int* capCylOut)
{
    int per = minCellActivated > 1 ? minCellActivated : 1;
    if (cylinders && per > 6)
        per = 6;
    const int capSeg = cells / per + 1;
    const int capCyl = cylinders ? cells / 6 + 1 : 1;
    if (capSegOut)
        *capSegOut = capSeg;
    if (capCylOut)
        *capCylOut = capCyl;
    const GenDims d = gen_dims(cells, hCells, vCells, cylinders, capSeg, capCyl);
    GenCarver cv{nullptr, 0, 0, nullptr, 0};
    GenPtrs P;
    gen_carve(cv, d, P);
    return (cv.glbOff + 255) & ~(size_t)255;
}

// dynamic LDS of the launch: the pinned buffers + as much of the per-cell working set as fits under min(ldsLimit, 64 KB) -- two
// workgroups per CU
size_t general_lds_bytes(int cells, int hCells, int vCells, bool cylinders, int capSeg, int capCyl, int ldsLimit)
{
    const size_t fixed = (gen_fixed_lds_bytes() + 15) & ~(size_t)15;
    size_t cap = 64 * 1024;
    if (const char* e = std::getenv("CAPE_GENERAL_LDS"))
        cap = (size_t)std::atol(e);
    if (cap > (size_t)ldsLimit)
        cap = (size_t)ldsLimit;
    if (cap < fixed)
        return 0; // (the caller refuses)
    const GenDims d = gen_dims(cells, hCells, vCells, cylinders, capSeg, capCyl);
    GenCarver cv{nullptr, fixed, cap, nullptr, 0};
    GenPtrs P;
    gen_carve(cv, d, P);
    return cv.ldsOff;
}

hipError_t launch_grow_general(const StageBParams& p, const GenParams& g, int nFrames, hipStream_t stream)
{
    int waves = nFrames < g.scratchSlots ? nFrames : g.scratchSlots;
    if (waves < 1)
        waves = 1;
    if (p.flags & CAPE_FLAG_CYLINDERS)
        hipLaunchKernelGGL(cape_grow_general_kernel<true>, dim3(waves), dim3(64), (size_t)g.ldsBytes, stream, p, g, nFrames);
    else
        hipLaunchKernelGGL(cape_grow_general_kernel<false>, dim3(waves), dim3(64), (size_t)g.ldsBytes, stream, p, g, nFrames);
    return hipGetLastError();
}

} // namespace cape
