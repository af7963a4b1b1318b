// Shared pieces of the stage-B kernels (cape_grow.hip: one wavefront per frame; cape_resume.hip: one workgroup per parked
// frame): LDS record layout, wave-level helpers, the parked-frame state layout, and grow_tail() -- everything after the
// record -> segment conversion (merge_planes, boundary candidates, records, cylinder morphology, label grids, header).
#pragma once
#include <hip/hip_runtime.h>

#include "cape_cylinder.h"
#include "cape_staged.h"
#include "cape_wave.h"
#include "cape_device.h"
#include "cape_internal.h"

namespace cape {

constexpr int kHistBins = 400;
constexpr int kSegDoubles = 20; // LDS plane-segment record: sums[9], n, normal[3], d, centroid[3], mse, score, planar
constexpr int kChunk = kStageChunk; // cells staged per step of the ordered moment accumulation (cape_staged.h)
#ifndef CAPE_B_PLANE_WAVES
#define CAPE_B_PLANE_WAVES 2 // waves per SIMD the plane-only instances are compiled for: 256 registers, nothing spills (3 = 168 registers spills ~50)
#endif
#ifndef CAPE_B_RESUME_WAVES
#define CAPE_B_RESUME_WAVES 1 // waves per SIMD the RESUME instance of the cylinder kernel is compiled for (2: 256 registers, ~120 spilled -- measured slower)
#endif
#ifndef CAPE_B_MSE_REGS
#define CAPE_B_MSE_REGS 12 // cell MSEs a lane keeps in registers across the seed loop (x 64 lanes = cells covered)
#endif
#ifndef CAPE_B_WAVES_PER_GROUP
#define CAPE_B_WAVES_PER_GROUP 4
#endif
constexpr int kWavesPerGroup = CAPE_B_WAVES_PER_GROUP; // independent frames (waves) per workgroup
__host__ __device__ constexpr int kChunkDoubles(int) { return kChunk * kSumStride; }
// s_dist of the cylinder instance: staging of the 18-double records of the combined LLS / merged-plane traversal
// (cape_cylinder.h).  (Through round 2 it also held one MSAC cost per cell for the rare exact-sum path -- 24 KB per wave
// on a 64x48 grid; those live in the free eighth double of the per-cell cylinder scratch now.)
__host__ __device__ constexpr int cyl_dist_doubles(int) { return kChunk * 18; }

// kernel-phase ablation for profiling experiments: -DCAPE_B_STOP_AT=k makes the wave leave after phase k
#ifdef CAPE_B_STOP_AT
#define CAPE_B_STOP(k)                                                                                       \
    do                                                                                                       \
    {                                                                                                        \
        if ((k) == CAPE_B_STOP_AT)                                                                           \
        {                                                                                                    \
            if (lane == 0)                                                                                   \
                p.records[frame].header.n_seeds = (int)U + (int)EL + (int)ER + (int)EU + (int)ED;            \
            return;                                                                                          \
        }                                                                                                    \
    } while (0)
#else
#define CAPE_B_STOP(k)
#endif

// per-phase shader-clock accounting for profiling experiments (-DCAPE_B_PROFILE): slot k accumulates the ticks since
// the previous CAPE_TICK
#ifdef CAPE_B_PROFILE
#define CAPE_TICK(k)                                                                                         \
    do                                                                                                       \
    {                                                                                                        \
        const unsigned long long _now = __builtin_amdgcn_s_memtime();                                        \
        if (lane == 0)                                                                                       \
            atomicAdd(&s_prof[(k)], _now - _tick); /* LDS, no return value: the tick does not stall the wave */ \
        _tick = _now;                                                                                        \
    } while (0)
#define CAPE_TICK_INIT() unsigned long long _tick = __builtin_amdgcn_s_memtime()
#define CAPE_TICK_RESTART() _tick = __builtin_amdgcn_s_memtime()
#else
#define CAPE_TICK_RESTART()
#define CAPE_TICK(k)
#define CAPE_TICK_INIT()
#endif

// wave-synchronous ordering point for LDS traffic between lanes of the single wave of this workgroup
// A workgroup carries several INDEPENDENT waves (one frame each, own LDS slice): the hardware admits only ~8
// workgroups per CU, so single-wave workgroups would cap the CU at 8 frames in flight.  Nothing is ever exchanged
// between waves, hence no s_barrier anywhere: this macro is "all my earlier LDS and global accesses are complete and
// visible to the other lanes of MY wave" = drain every counter + compiler fence.
#define CAPE_WAVE_SYNC()                                                                                      \
    do                                                                                                       \
    {                                                                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                               \
        __builtin_amdgcn_s_waitcnt(0);                                                                       \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                               \
    } while (0)
// LDS-only ordering point.  The workgroup is ONE wave and LDS instructions of a wave execute in issue order, so a
// compiler-level fence is enough; unlike __syncthreads() it does not drain outstanding global loads (vmcnt), which
// is what lets the cell-sum prefetch below stay in flight across it.
#define CAPE_LDS_SYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

template <typename MaskT> __device__ __forceinline__ int popc(MaskT m);
template <> __device__ __forceinline__ int popc<uint32_t>(uint32_t m) { return __popc(m); }
template <> __device__ __forceinline__ int popc<unsigned long long>(unsigned long long m) { return __popcll(m); }
template <typename MaskT> __device__ __forceinline__ int ctz(MaskT m);
template <> __device__ __forceinline__ int ctz<uint32_t>(uint32_t m) { return __ffs(m) - 1; }
template <> __device__ __forceinline__ int ctz<unsigned long long>(unsigned long long m) { return __ffsll(m) - 1; }

template <> __device__ __forceinline__ int popc<Mask128>(Mask128 m) { return __popcll(m.lo) + __popcll(m.hi); }
template <> __device__ __forceinline__ int ctz<Mask128>(Mask128 m) { return m.lo ? __ffsll(m.lo) - 1 : 64 + __ffsll(m.hi) - 1; }

template <typename MaskT> __device__ __forceinline__ MaskT shfl_mask(MaskT v, int src)
{
    return (MaskT)__shfl((unsigned long long)v, src);
}
template <> __device__ __forceinline__ Mask128 shfl_mask<Mask128>(Mask128 v, int src) { return Mask128(__shfl(v.lo, src), __shfl(v.hi, src)); }
// the mask without its lowest set bit; bit c of a mask; the first HC cells of a row
template <typename MaskT> __device__ __forceinline__ MaskT clear_lowest(MaskT m) { return m & (MaskT)(m - 1); }
template <> __device__ __forceinline__ Mask128 clear_lowest<Mask128>(Mask128 m) { return m.lo ? Mask128(m.lo & (m.lo - 1ull), m.hi) : Mask128(0ull, m.hi & (m.hi - 1ull)); }
template <typename MaskT> __device__ __forceinline__ bool test_bit(MaskT m, int c) { return ((m >> c) & (MaskT)1) != (MaskT)0; }
template <typename MaskT> __device__ __forceinline__ MaskT width_mask(int hc)
{
    return (hc >= (int)(8 * sizeof(MaskT))) ? ~(MaskT)0 : (MaskT)(((MaskT)1 << hc) - 1);
}
template <> __device__ __forceinline__ Mask128 width_mask<Mask128>(int hc)
{
    return hc >= 128 ? Mask128(~0ull, ~0ull) : (hc >= 64 ? Mask128(~0ull, hc == 64 ? 0ull : ((1ull << (hc - 64)) - 1ull)) : Mask128((1ull << hc) - 1ull, 0ull));
}
constexpr bool kIsWide(size_t maskBytes) { return maskBytes == 16; }

// 3x3 morphology on bit rows.  up/dn are the neighbouring rows (0 outside the grid).
template <typename MaskT> __device__ __forceinline__ MaskT row3(MaskT x, MaskT widthMask) { return (x | (x << 1) | (x >> 1)) & widthMask; }

template <typename MaskT> struct Rows
{
    // neighbour rows of a per-lane row mask: one DPP wave shift (zero past the ends of the wave)
    static __device__ __forceinline__ MaskT up(MaskT v, int) { return wave_from_lane_below(v); } // row r-1
    static __device__ __forceinline__ MaskT dn(MaskT v, int lane, int vCells)                     // row r+1
    {
        const MaskT w = wave_from_lane_above(v);
        return (lane + 1 < vCells) ? w : (MaskT)0;
    }
};

struct SegRec // uniform (all lanes hold the same values)
{
    double S[9];
    double n; // point count (exact integer in f64)
    double nx, ny, nz, d;
    double cx, cy, cz;
    double mse, score;
    double planar;
};

__device__ __forceinline__ void seg_store(double* lds, const SegRec& s)
{
#pragma unroll
    for (int k = 0; k < 9; ++k)
        lds[k] = s.S[k];
    lds[9] = s.n;
    lds[10] = s.nx; lds[11] = s.ny; lds[12] = s.nz; lds[13] = s.d;
    lds[14] = s.cx; lds[15] = s.cy; lds[16] = s.cz;
    lds[17] = s.mse; lds[18] = s.score; lds[19] = s.planar;
}
__device__ __forceinline__ void seg_load(const double* lds, SegRec& s)
{
#pragma unroll
    for (int k = 0; k < 9; ++k)
        s.S[k] = lds[k];
    s.n = lds[9];
    s.nx = lds[10]; s.ny = lds[11]; s.nz = lds[12]; s.d = lds[13];
    s.cx = lds[14]; s.cy = lds[15]; s.cz = lds[16];
    s.mse = lds[17]; s.score = lds[18]; s.planar = lds[19];
}

// Matrix3d::inverse, cofactor method -> Plane_Segment::get_point_cloud_covariance
__device__ inline void inverse3_sym(const double (&S)[9], double (&r)[9])
{
    // hessian {{Sxs,Sxy,Szx},{Sxy,Sys,Syz},{Szx,Syz,Szs}}
    const double m[3][3] = {{S[3], S[6], S[8]}, {S[6], S[4], S[7]}, {S[8], S[7], S[5]}};
    auto cof = [&](int i, int j) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
    };
    const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const double det = (c00 * m[0][0] + c10 * m[1][0]) + c20 * m[2][0];
    const double invdet = 1.0 / det;
    r[0] = c00 * invdet;
    r[1] = c10 * invdet;
    r[2] = c20 * invdet;
    r[3] = cof(0, 1) * invdet;
    r[4] = cof(1, 1) * invdet;
    r[5] = cof(2, 1) * invdet;
    r[6] = cof(0, 2) * invdet;
    r[7] = cof(1, 2) * invdet;
    r[8] = cof(2, 2) * invdet;
}

// ---- state of a frame parked by the plane-only pass for the RESUME instance (p.growState, grow_state_bytes() per frame)
struct GrowStateHeader
{
    int32_t nSeg;        // plane segments made so far (s_seg[0, nSeg))
    int32_t pendFrom;    // first record that has not been converted: the cylinder candidate
    int32_t pendCount;   // records in the window
    int32_t pendBaseSlot; // s_seg slot of record 0 of the window
    int32_t nSeeds, nPlanar;
    uint32_t status;
    int32_t pad;
};
constexpr int kPendCyl = 16;   // recorded regions the cylinder instance keeps at a time
constexpr int kPendResume = 8; // ... and the RESUME instance: a frame parked with more unconverted regions takes the full redo
__host__ __device__ constexpr size_t grow_state_seg_off() { return sizeof(GrowStateHeader); }
__host__ __device__ constexpr size_t grow_state_adj_off() { return grow_state_seg_off() + (size_t)(kFastPlanes + 1) * kSegDoubles * 8; }
__host__ __device__ constexpr size_t grow_state_list_off() { return grow_state_adj_off() + (size_t)(kFastPlanes + 1) * 8; }
__host__ __device__ inline size_t grow_state_lab_off(int cells) { return grow_state_list_off() + (((size_t)cells + 4) * 2 + 7) / 8 * 8; }
__host__ __device__ inline size_t grow_state_bytes_(int cells) { return (grow_state_lab_off(cells) + (size_t)cells + 15) / 16 * 16; }

// ---- the parked frames by cost class (StageBParams::resumeBucketStride): the k-th frame of the finisher, dearest class first
__device__ __forceinline__ int resume_cost_class(int candidateCells, int cells)
{
    // cells that will go through cylinder_fitting (RANSAC + ordered sums), in eighths of the grid: whole walls ... a patch
    const int c = candidateCells * kResumeClasses / (cells > 0 ? cells : 1);
    return c < kResumeClasses ? c : kResumeClasses - 1;
}
__device__ __forceinline__ int resume_pick(const StageBParams& p, int k)
{
    if (!p.resumeBucketStride)
        return k < (int)p.resumeList[0] ? (int)p.resumeList[1 + k] : -1;
    for (int c = kResumeClasses; c >= 1; --c)
    {
        const uint32_t* bl = p.resumeList + (size_t)c * p.resumeBucketStride;
        const int n = (int)bl[0];
        if (k < n)
            return (int)bl[1 + k];
        k -= n;
    }
    return -1;
}

// Everything after the seed loop / the record -> segment conversion, for ONE wave (lane r owns grid row r): merge_planes,
// boundary candidates, the segment records, the cylinder morphology, label grids and the frame header.  Shared by the grow
// kernel and by the cylinder group kernel (cape_resume.hip), whose wave 0 runs it.
struct GrowTailLds
{
    double* s_seg;
    unsigned long long* s_adj;   // MAXP + 1 u64, all zero on entry
    unsigned char* s_mlab;       // MAXP u8, identity on entry
    unsigned char* s_lab;
    unsigned char* s_cyl;
    float* s_zc;                 // C f32, free on entry
    unsigned short* s_ring;      // >= 1280 u16, free on entry
    unsigned long long* s_prof;  // phase ticks (profiling builds)
};

template <typename MaskT, bool CYL, int MAXP>
__device__ __forceinline__ void grow_tail(const StageBParams& p, const int frame, const int lane, const GrowTailLds& L, const int nSeg,
                                          const int nCylLabels, const int nSeeds, const int nPlanar, uint32_t status,
                                          const unsigned long long tPhase = 0ull)
{
    // stage buckets of the reference (timing on): everything up to here was grow_planes_and_cylinders
    unsigned long long tMerge = 0ull;
    if (p.phaseTicks)
    {
        tMerge = __builtin_amdgcn_s_memtime();
        if (lane == 0)
            atomicAdd(&p.phaseTicks[(size_t)frame * 4 + 0], tMerge - tPhase);
    }
    double* const s_seg = L.s_seg;
    unsigned long long* const s_adj = L.s_adj;
    unsigned char* const s_mlab = L.s_mlab;
    unsigned char* const s_lab = L.s_lab;
    unsigned char* const s_cyl = L.s_cyl;
    float* const s_zc = L.s_zc;
    unsigned short* const s_ring = L.s_ring;
#ifdef CAPE_B_PROFILE
    unsigned long long* const s_prof = L.s_prof;
#endif
    const int C = p.cells, HC = p.hCells, VC = p.vCells;
    const size_t cellBase = (size_t)frame * C;
    const MaskT widthMask = width_mask<MaskT>(HC);
    CAPE_TICK_INIT();
    // =========================================================================================
    // merge_planes (:503-560) with get_connected_components_matrix (:736-776)
    // =========================================================================================
    for (int i = lane; i < C; i += 64)
    {
        const int r = i / HC, c = i - r * HC;
        if (r >= VC - 1 || c >= HC - 1)
            continue; // last row / last column never act as sources
        const int a = s_lab[i];
        if (a <= 0)
            continue;
        const int b = s_lab[i + 1];
        const int dwn = s_lab[i + HC];
        if (b > 0 && a != b)
        {
            atomicOr(&s_adj[a - 1], 1ull << (b - 1));
            atomicOr(&s_adj[b - 1], 1ull << (a - 1));
        }
        if (dwn > 0 && a != dwn)
        {
            atomicOr(&s_adj[a - 1], 1ull << (dwn - 1));
            atomicOr(&s_adj[dwn - 1], 1ull << (a - 1));
        }
    }
    CAPE_WAVE_SYNC();

    for (int row = 0; row < nSeg; ++row)
    {
        const int planeId = s_mlab[row];
        SegRec A;
        seg_load(s_seg + planeId * kSegDoubles, A);
        if (A.planar == 0.0)
            continue;
        bool expanded = false;
        const unsigned long long conn = s_adj[row];
        for (int col = row + 1; col < nSeg; ++col)
        {
            if (!((conn >> col) & 1ull))
                continue;
            SegRec B;
            seg_load(s_seg + col * kSegDoubles, B);
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
                    s_mlab[col] = (unsigned char)planeId;
                expanded = true;
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
                seg_store(s_seg + planeId * kSegDoubles, A);
        }
        CAPE_WAVE_SYNC();
    }

    CAPE_TICK(10);
    unsigned long long tRefine = 0ull;
    if (p.phaseTicks)
    {
        tRefine = __builtin_amdgcn_s_memtime();
        if (lane == 0)
            atomicAdd(&p.phaseTicks[(size_t)frame * 4 + 1], tRefine - tMerge);
    }
    // =========================================================================================
    // add_planes_to_primitives (:562-648) + compute_plane_segment_boundary (:650-703)
    // =========================================================================================
    // centre-pixel depth of every cell (depthImage(centerY, centerX) of compute_plane_segment_boundary) into LDS
    if (nSeg > 0)
    {
        for (int i = lane; i < C; i += 64)
            s_zc[i] = p.cell_aux[cellBase + i].zc;
    }
    CAPE_WAVE_SYNC();
    cape_frame_record* rec = p.records + frame;
    double* bnd = p.boundary + (size_t)frame * p.boundaryCapacity * 3;
    const double browCenterOfLane = p.brow[(lane < VC ? lane : 0) * kCell + kCell / 2]; // lane r: row r's centre ordinate
    const double acolCenterOfLane = p.acol[(lane < HC ? lane : 0) * kCell + kCell / 2]; // lane c: column c's centre abscissa
    int nBoundary = 0;
    // lane j keeps what the loop decides for plane segment j (MAXP <= 64 = the wave width)
    const int myMlab = lane < nSeg ? (int)s_mlab[lane] : -1;
    uint32_t myOut = 0, myOff = 0, myCnt = 0;
    // Boundary candidates of one plane = the cells of its ring (dilate(square) minus erode(cross)) whose centre pixel lies
    // within 3 sigma of the plane, in row-major order.  The ring cells are enumerated with a prefix sum over the rows into a
    // cell list (the staging chunk is free here), so that 64 lanes test 64 ring cells at a time instead of one grid row at
    // a time; a band of rows that surely fits the list is handled per pass (the whole grid for 640x480).
    constexpr int kBandRows = sizeof(MaskT) == 4 ? 32 : (sizeof(MaskT) == 8 ? 16 : 8); // band rows x grid width <= 1024 entries
    for (int pi = 0; pi < nSeg; ++pi)
    {
        const int mlabel = s_mlab[pi];
        const double* segp = s_seg + pi * kSegDoubles;
        uint32_t isOutput = 0, bOff = (uint32_t)nBoundary, bCnt = 0;
        if (mlabel == pi && segp[19] != 0.0)
        {
            const double Anx = segp[10], Any = segp[11], Anz = segp[12], Ad = segp[13], Amse = segp[17];
            // label set of the merge group: j >= pi with planeMergeLabels[j] == pi
            const unsigned long long group = __ballot(lane >= pi && myMlab == mlabel);
            // row masks by ballot: the lanes take the cells of one grid row (two rows for grids up to 32 wide)
            MaskT M = 0;
            if (sizeof(MaskT) == 4)
            {
                const int h = lane >> 5, col = lane & 31;
                for (int t = 0; 2 * t < VC; ++t)
                {
                    const int r = 2 * t + h;
                    const bool in = col < HC && r < VC;
                    const int l = in ? (int)s_lab[r * HC + col] : 0;
                    const unsigned long long bm = __ballot(l > 0 && ((group >> (l - 1)) & 1ull));
                    if (lane == 2 * t)
                        M = (MaskT)(uint32_t)bm;
                    if (lane == 2 * t + 1)
                        M = (MaskT)(uint32_t)(bm >> 32);
                }
            }
            else if constexpr (sizeof(MaskT) == 16)
            {
                // rows of up to 128 cells: two ballots per row
                for (int r = 0; r < VC; ++r)
                {
                    const int l0 = lane < HC ? (int)s_lab[r * HC + lane] : 0, l1 = 64 + lane < HC ? (int)s_lab[r * HC + 64 + lane] : 0;
                    const unsigned long long b0 = __ballot(l0 > 0 && ((group >> (l0 - 1)) & 1ull));
                    const unsigned long long b1 = __ballot(l1 > 0 && ((group >> (l1 - 1)) & 1ull));
                    if (lane == r)
                        M = MaskT(b0, b1);
                }
            }
            else
            {
                for (int r = 0; r < VC; ++r)
                {
                    const int l = lane < HC ? (int)s_lab[r * HC + lane] : 0;
                    const unsigned long long bm = __ballot(l > 0 && ((group >> (l - 1)) & 1ull));
                    if (lane == r)
                        M = (MaskT)bm;
                }
            }
            const MaskT Mup = Rows<MaskT>::up(M, lane), Mdn = Rows<MaskT>::dn(M, lane, VC);
            // erode, 3x3 cross, BORDER_CONSTANT 0 ; dilate, 3x3 square, border ignored
            const MaskT ero = M & (MaskT)(M << 1) & (MaskT)(M >> 1) & Mup & Mdn;
            const MaskT dil = row3<MaskT>(M, widthMask) | row3<MaskT>(Mup, widthMask) | row3<MaskT>(Mdn, widthMask);
            const MaskT ring = dil & ~ero;

            const double maxBoundaryDistance = 3 * sqrt(Amse);
            for (int r0 = 0; r0 < VC; r0 += kBandRows)
            {
                // ring cells of rows [r0, r0 + kBandRows) in row-major order -> s_ring
                const bool mine = lane >= r0 && lane < r0 + kBandRows && lane < VC;
                const MaskT rr = mine ? ring : (MaskT)0;
                const int cnt = popc<MaskT>(rr);
                const int incl = wave_scan_i32(cnt);
                const int R = (int)readlane_u32((unsigned)incl, 63);
                if (R == 0)
                    continue;
                {
                    int pos = incl - cnt;
                    MaskT m = rr;
                    while (m)
                    {
                        const int c = ctz<MaskT>(m);
                        s_ring[pos++] = (unsigned short)(lane * HC + c);
                        m = clear_lowest<MaskT>(m);
                    }
                }
                CAPE_LDS_SYNC();
                for (int j0 = 0; j0 < R; j0 += 64)
                {
                    const int j = j0 + lane;
                    bool hit = false;
                    double px = 0, py = 0, pz = 0;
                    // every lane takes part in the two lane look-ups (a permute reads nothing from an inactive lane)
                    const int cell = s_ring[j < R ? j : R - 1];
                    const int r = cell / HC, c = cell - r * HC;
                    // (column c's centre abscissa lives in lane c -- on the grids of up to 64 columns; a wider row reads it from memory)
                    const double ac = sizeof(MaskT) == 16 ? p.acol[c * kCell + kCell / 2] : __shfl(acolCenterOfLane, c), br = __shfl(browCenterOfLane, r);
                    const double dpt = (double)s_zc[cell]; // depthImage(centerY, centerX), staged by stage A
                    if (j < R && dpt > 0)
                    {
                        px = dpt * ac;
                        py = dpt * br;
                        pz = dpt;
                        const double dist = dot3(Anx, Any, Anz, px, py, pz) + Ad;
                        hit = fabs(dist) < maxBoundaryDistance;
                    }
                    const unsigned long long hb = __ballot(hit);
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
                CAPE_LDS_SYNC();
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
        if (lane == pi)
        {
            myOut = isOutput;
            myOff = bOff;
            myCnt = bCnt;
        }
    }
    const int nPlanesOut = __popcll(__ballot(myOut != 0));
    // the records of all segments at once, lane j <- segment j: Plane::_parametrization's extra normalisation
    // (shape_primitives.cpp:49) and get_point_cloud_covariance (plane_segment.cpp:192-203) run lane parallel
    if (lane < nSeg)
    {
        SegRec A;
        seg_load(s_seg + lane * kSegDoubles, A);
        double onx = 0, ony = 0, onz = 0;
        double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (myOut)
        {
            onx = A.nx; ony = A.ny; onz = A.nz;
            normalize3(onx, ony, onz);
            inverse3_sym(A.S, cov);
        }
        cape_plane_segment* o = &rec->segments[lane];
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
        o->merge_label = (uint32_t)myMlab;
        o->planar = A.planar != 0.0 ? 1u : 0u;
        o->is_output = myOut;
        o->boundary_offset = myOff;
        o->boundary_count = myCnt;
    }

    CAPE_TICK(11);
    // =========================================================================================
    // add_cylinders_to_primitives (:705-734): open (dilate, erode) + erode with the 3x3 cross, default borders
    // =========================================================================================
    int nCylOut = 0;
    for (int ci = 0; ci < nCylLabels; ++ci)
    {
        MaskT M = 0;
        if (lane < VC)
            for (int c = 0; c < HC; ++c)
                if (s_cyl[lane * HC + c] == ci + 1)
                    M |= (MaskT)1 << c;
        const MaskT allRow = (lane < VC) ? widthMask : (MaskT)0;
        const MaskT firstCol = (MaskT)1, lastCol = (MaskT)1 << (HC - 1);
        auto dilate = [&](MaskT x) {
            const MaskT up = Rows<MaskT>::up(x, lane), dn = Rows<MaskT>::dn(x, lane, VC);
            return (MaskT)((x | (MaskT)(x << 1) | (MaskT)(x >> 1) | up | dn) & allRow);
        };
        auto erode = [&](MaskT x) { // outside the grid never erodes (morphologyDefaultBorderValue)
            const MaskT up = Rows<MaskT>::up(x, lane) | (lane == 0 ? widthMask : (MaskT)0);
            const MaskT dn = Rows<MaskT>::dn(x, lane, VC) | (lane == VC - 1 ? widthMask : (MaskT)0);
            return (MaskT)(x & ((MaskT)(x << 1) | firstCol) & ((MaskT)(x >> 1) | lastCol) & up & dn & allRow);
        };
        const MaskT e2 = erode(erode(dilate(M)));
        const int ones = wave_sum_i32(popc<MaskT>(e2));
        const bool kept = ones > 0 && ones < C; // max > 0 and min < max
        if (kept)
        {
            if (lane == 0)
                rec->cylinders[ci].kept = 1;
            ++nCylOut;
        }
    }

    // =========================================================================================
    // label grids + header
    // =========================================================================================
    for (int i = lane; i < C; i += 64)
    {
        p.plane_labels[cellBase + i] = (int32_t)s_lab[i];
        p.cyl_labels[cellBase + i] = CYL ? (int32_t)s_cyl[i] : 0;
    }
    // fold the per-lane status bits
    status = wave_or_u32(status);
    if (lane == 0)
    {
        rec->header.n_plane_segments = nSeg;
        rec->header.n_planes = nPlanesOut;
        rec->header.n_cylinder_labels = nCylLabels;
        rec->header.n_cylinders = nCylOut;
        rec->header.n_boundary_points = nBoundary < p.boundaryCapacity ? nBoundary : p.boundaryCapacity;
        rec->header.n_seeds = nSeeds;
        rec->header.status = status;
        rec->header.n_planar_cells = nPlanar;
        rec->header.next_record = -1; // a frame of these instances fits its record (more than 64 segments: the general instance)
        rec->header.segment_base = 0;
        if (p.phaseTicks)
            atomicAdd(&p.phaseTicks[(size_t)frame * 4 + 2], (unsigned long long)__builtin_amdgcn_s_memtime() - tRefine);
    }
}

} // namespace cape
