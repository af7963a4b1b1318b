// Internal device/host shared layouts of libcape_hip (not part of the C ABI).
#pragma once
#include <stdint.h>

#include "../../include/cape_hip.h"

namespace cape {

// Per-cell scratch written by stage A (cell fit) and read by stage B (grow).  HBM layout, per frame:
//   cell_sums  [cells][10] f64 : Sx Sy Sz Sxs Sys Szs Sxy Syz Szx, point count (as f64) -- 80 B/cell, AoS so that
//                                the ordered region accumulation reads one cell with a single 80-byte request
//   cell_plane [cells][8]  f64 : nx ny nz d cx cy cz mse                               -- 64 B/cell
//   cell_score [cells]     f64
//   cell_tol   [cells]     f32 : _cellDistanceTols
//   cell_flags [cells]     u32 : point count | four directed-edge merge predicates << 24 | near-edge << 29 | inorder << 30 | planar << 31
//   cell_bins  [cells]     i32 : histogram bin (-1 if not planar)
//   cell_aux   [cells]   16 B  : A1 -> A2 hand-over (corner depths, count, continuity / exactness verdicts) + centre depth
//   cell_mse   [cells]     f64 : copy of the cell MSE, compact so the seed selection reads it coalesced
constexpr int kSumStride = 10;
constexpr int kFastPlanes = 32;     // plane segments the everyday grow-kernel instances hold in LDS (CAPE_MAX_PLANES = 64 in the redo instance)
constexpr int kProfileSlots = 32;   // phase counters per frame of a -DCAPE_B_PROFILE build (cape_debug_cycles)
constexpr int kResumeClasses = 8;  // cost classes of the parked frames (StageBParams::resumeBucketStride)
constexpr int kCylStride = 8;       // doubles per cell of the cylinder scratch: projected normal[3], projected centroid[3], their dot product, pad
constexpr int kPlaneStride = 8;
constexpr uint32_t kFlagPlanar = 1u << 31;
constexpr uint32_t kFlagInorder = 1u << 30;
constexpr uint32_t kFlagNearEdge = 1u << 29;
constexpr uint32_t kCountMask = (1u << 24) - 1;
// region_growing's merge predicate (primitive_detection.cpp:802 with plane_segment.cpp:322-326) of the four directed cell
// edges that end or start at this cell, evaluated by stage A2 where the cell planes are still in registers:
constexpr uint32_t kFlagLeftToMe = 1u << 24; // parent (r, c-1) -> child (r, c), child's tolerance   = EL bit c of row r
constexpr uint32_t kFlagMeToLeft = 1u << 25; // parent (r, c) -> child (r, c-1), left cell's tolerance = ER bit c-1 of row r
constexpr uint32_t kFlagUpToMe = 1u << 26;   // parent (r-1, c) -> child (r, c)                        = EU bit c of row r
constexpr uint32_t kFlagMeToUp = 1u << 27;   // parent (r, c) -> child (r-1, c)                        = ED bit c of row r-1
constexpr uint32_t kAuxContinuous = 1u << 31;
constexpr uint32_t kAuxExact = 1u << 30;

struct CellAux
{
    float z0, z399;  // depth of the cell's first / last pixel (cloud rows offset and offset+399)
    uint32_t flags;  // valid pixel count | kAuxExact | kAuxContinuous
    float zc;        // depth of the cell's centre pixel (col*20+10, row*20+10): boundary candidate test
};

struct StageAParams
{
    const float* depth; // frames x H x W float32 millimetres ...
    const uint16_t* depth_u16; // ... or raw uint16 sensor units (then depth == nullptr): z = (float)raw * u16_scale
    float u16_scale;
    int W, H, hCells, vCells, cells;
    int segsPerRow;     // ceil(hCells / 32): 640-pixel wide band segments per cell row
    int bandsPerFrame;  // vCells * segsPerRow
    int pairsPerFrame;  // ceil(bandsPerFrame / 2): workgroups per frame
    const double* acol; // [W]  fl(fl(k00*u) + k02)
    const double* brow; // [H]  fl(fl(k11*v) + k12)
    const float* ratio_col; // [hCells] max|a| / min nonzero |a| over the cell's columns (exactness guard)
    const float* ratio_row; // [vCells]
    double* cell_sums;
    double* cell_plane;
    double* cell_score;
    float* cell_tol;
    uint32_t* cell_flags;
    int32_t* cell_bins;
    CellAux* cell_aux;
    double* cell_mse;
    float sinMerge; // sinf((float)(18 * pi / 180)), primitive_detection.cpp:189-190
    uint32_t* clear0; // hand-over counters of the grow kernel (redo list, cylinder list, resume list): zeroed by one thread of
    uint32_t* clear1; // stage A2, which always runs right before it on the same stream -- instead of memset nodes per call
    uint32_t* clear2;
    uint32_t clear2Buckets; // words between the resume list and each of its kResumeClasses cost-class lists (0: none), zeroed with clear2
    uint32_t* clear3; // the spill list of the general grow instance: [0] frames listed
    uint32_t* clear4; // ... and the allocation counter of the spill record pool (nullptr on a sub-batch: cleared once per call)
    double cosMergeA; // cos(18 * pi / 180), plane_segment.cpp:324 (the edge predicates of stage A2)
    int smallBatchFrames; // host side: batches up to this many frames run the latency-oriented kernel instances
    int minZeroPointCount; // floor(400 * 0.7f) = 280, plane_segment.hpp:33-34
};

struct StageBParams
{
    int W, H, hCells, vCells, cells;
    const double* acol;
    const double* brow;
    const double* cell_sums;
    const double* cell_plane;
    const double* cell_score;
    const float* cell_tol;
    const uint32_t* cell_flags;
    const int32_t* cell_bins; // Histogram::_bins right after init_histogram (written by stage A2)
    const CellAux* cell_aux;
    const double* cell_mse;
    cape_frame_record* records;
    int32_t* plane_labels;
    int32_t* cyl_labels;
    double* boundary;
    int boundaryCapacity;
    uint32_t flags;
    double cosMerge;       // cos(18 * pi / 180), plane_segment.cpp:324
    int planeSeedCount;    // uint(0.008 * cells), primitive_detection.cpp:278-279
    int minCellActivated;  // uint(0.0065 * cells), primitive_detection.cpp:362-363
    const double* rngTable; // first rngCount doubles of uniform_real_distribution(mt19937(0)), random.hpp:17-30
    int rngCount;
    int ransacMaxIterations; // 43
    double* cylScratch;      // [frames][cells][kCylStride] projected normals / centroids / n.c (only with CAPE_FLAG_CYLINDERS)
    // Two-pass scheduling of the cylinder variant (see launch_grow): the plane-only kernel runs first on every frame and
    // appends a frame to the list needCylinder[1..] (count in [0]) -- abandoning it -- when one of its regions takes the
    // cylinder branch; the cylinder kernel then redoes exactly the listed frames.  nullptr: single pass.
    uint32_t* needCylinder;
    uint32_t* redoList;      // same layout: frames that need more than kFastPlanes segment slots; nullptr = truncate + flag
    uint32_t* spillList;     // same layout: frames the 64-segment instance ran out of record capacity on (more than 64 plane segments or
                             // cylinder labels): the general instance (cape_grow_general.hip) redoes them into a chain of records
    // Hand-over WITH state (round 3): when the plane-only pass reaches a cylinder candidate after its seed loop has ended, it
    // parks what it has -- segments so far, the recorded regions with their fits, cell lists, labels -- in growState and
    // appends the frame to resumeList; the RESUME instance of the cylinder kernel picks the frame up at that region instead
    // of growing it again from the first seed (grow_state_bytes() per frame; nullptr: always the full redo).
    uint32_t* resumeList;
    // the parked frames once more, by cost class (cells of their cylinder candidates): list c starts (c + 1) * resumeBucketStride words
    // behind resumeList, same layout.  The finisher takes the dearest class first -- its pass lasts as long as its slowest frame, so
    // the long ones must not start last (0: no class lists, frames in the order they were parked)
    uint32_t resumeBucketStride;
    unsigned char* growState;
    uint32_t growStateStride;
    int resumeMode;          // how parked frames are finished: 2 = one workgroup per frame (cape_resume.hip), 1 = one wavefront
                             // per frame (the RESUME instance of the grow kernel; kept for A/B runs: CAPE_RESUME=wave)
    int twoPass;             // 0: the cylinder kernel grows every frame itself (chosen when most frames were handed over)
    unsigned long long* debugCycles; // [frames][kProfileSlots] shader-clock ticks per phase (only in -DCAPE_B_PROFILE builds)
    // The reference's stage buckets inside stage B (primitive_detection.cpp:140-160): shader-clock ticks every frame's wave spent
    // in [0] grow_planes_and_cylinders, [1] merge_planes, [2] add_planes / add_cylinders_to_primitives, summed over the timed calls
    // -- four u64 PER FRAME SLOT of the batch (three fire-and-forget atomics per frame, each on the frame's own words: 4 096 waves
    // adding to three shared words cost the kernel 70 us); nullptr unless the handle's timing is on.  cape_get_timings sums the
    // slots and splits the grow kernels' event time in these proportions.
    unsigned long long* phaseTicks;
    int countersCleared;     // 1: stage A2 zeroed redoList[0] / needCylinder[0] (StageAParams::clear0/1); 0: launch_grow does
    int a2RowsPerTile;       // cell rows per workgroup of stage A2: the vertical edges into rows k * a2RowsPerTile are evaluated here
    uint16_t* seed_sequence; // [frames][cells] seed cells in the order the seed loop tried them (first n_seeds entries valid)
    int ldsLimitBytes;       // host side only: LDS one workgroup may ask for on the handle's device (queried at cape_create)
    // the one-frame chain (handles of max_batch <= 8, results in pinned host memory): ONE grow kernel -- the 64-segment instance on
    // every frame of the call (allFrames) -- whose last wave stores the call's sequence number into the pinned word the host spins on
    int allFrames;
    uint32_t* doneFlag;      // pinned, device-mapped ; nullptr: nobody to signal
    uint32_t* doneCounter;   // device: waves of the signalling kernel that are through
    uint32_t doneSeq;
    // one-frame chain on a grid the fast kernels serve: the general instance is NOT enqueued behind the 64-segment one (a launch on
    // the latency path for a frame in ten thousand) -- the signalling wave leaves the number of frames on spillList in this pinned
    // word, and whoever reads the results launches the general kernel then, if it is not zero (cape_api.hip: wait_results)
    uint32_t* spillHost;
};

// The general grow instance (cape_grow_general.hip): any grid size, any number of plane segments.  One wavefront per frame, its
// working set carved out of LDS as far as that reaches and out of a scratch slot in HBM beyond; results go into a chain of
// records (cape_frame_header::next_record) taken from the handle's spill pool.
struct GenParams
{
    uint32_t* spillAlloc;            // records handed out of the pool by the call in flight (one counter per handle)
    uint32_t* genFrames;             // frames that went through this instance in the call in flight (cape_spill_info)
    cape_frame_record* poolRecords;  // the pool: record index poolBase + k
    double* poolBoundary;            // k-th boundary slab of the pool (boundaryCapacity x 3 doubles each)
    int poolCapacity, poolBase;
    unsigned char* scratch;          // scratchSlots x slotBytes
    size_t slotBytes;
    int scratchSlots;
    int ldsBytes;                    // dynamic LDS of the launch
    int capSeg, capCyl;              // most plane segments / cylinder labels a frame of this grid can hold
    int rowWords;                    // 64-bit words per bit row of the cell grid
    int allFrames;                   // 1: wave k takes frames k, k + waves, ... of the call; 0: the frames on StageBParams::spillList
};
size_t general_slot_bytes(int cells, int hCells, int vCells, bool cylinders, int minCellActivated, int* capSeg, int* capCyl);
size_t general_lds_bytes(int cells, int hCells, int vCells, bool cylinders, int capSeg, int capCyl, int ldsLimit);

// N3: Depth_Map_Transformation::rectify_depth
struct RectifyParams
{
    const float* in;  // frames x H x W depth of camera 2
    float* out;       // frames x H x W depth registered to camera 1
    unsigned* frameFlag; // [frames] set by the tile kernel when a pixel of the frame lands outside the scanning bands' reach
    unsigned* flagged;   // [1 + frames] count, then the flagged frames: redone by the general kernels
    int bandRows;        // 0: chosen by launch_rectify (CAPE_RECTIFY_BAND overrides it)
    int shiftLo, shiftHi; // predicted range of (target row - source row) for this rig, margin included (cape_rectify_depth)
    int ldsLimitBytes = 0; // LDS one workgroup may use on the handle's device (0: unknown, 64 KB assumed)
    int W, H;
    const float* xpre; // [W] static_cast<float>(acol), ypre [H]
    const float* ypre;
    double T[12];      // first three rows of the 4x4 camera2 -> camera1 matrix, row-major
    double fx, fy, cx, cy; // camera 1 intrinsics
};

// N2 (device part): cell-mask plane matching between consecutive frames
struct MatchParams
{
    const cape_frame_record* records;
    const int32_t* plane_labels;
    cape_frame_match* matches;
    int cells;
    uint32_t flags;
    double minCosAngle;     // abs(cos(20 * pi / 180)), shape_primitives.cpp:72-73
    double maxDistance;     // 100 mm, shape_primitives.cpp:84
    double minOverlap;      // (double)0.4f, halved for the advanced search (map_primitive.cpp:106-107)
};

// N2 with polygon areas (cape_match_polygon.hip)
struct MatchPolygonParams
{
    const cape_frame_record* records;
    const cape_polygon* polygons; // frames x CAPE_MAX_PLANES
    const double2* vertices;      // frames x boundaryCapacity
    cape_frame_match_exact* matches;
    unsigned* listCounts; // pairs on the work list of each capacity tier of the intersection kernel
    unsigned* pairLists;  // 4 lists of pairCapacity entries: (frame << 8) | (j << 4) | i
    size_t pairCapacity;  // max_batch x 256
    int boundaryCapacity;
    int computeUnits;
    uint32_t flags;
    double minCosAngle, maxDistance, minOverlap; // as MatchParams
    int ldsLimitBytes = 65536;     // LDS one workgroup may use on the handle's device
    const double* poses = nullptr; // cape_match_polygons_pose: frames x 16, row-major [R t; 0 0 0 1] from camera f-1 to camera f; null = identity
};

// multi-GPU gather: device-side packing of the ragged primitive lists (cape_gather.hip)
struct PackParams
{
    const cape_frame_record* records;
    const cape_frame_record* recordsBase; // the handle's record array (what cape_frame_header::next_record indexes)
    int poolBase;                         // first record index of the spill pool (= max_batch)
    const int32_t* planeLabelsIn;
    const int32_t* cylLabelsIn;
    cape_packed_header* header;
    cape_packed_frame* frames;
    cape_packed_plane* planes;
    cape_packed_cylinder* cylinders;
    uint8_t* planeLabels8;
    uint8_t* cylLabels8;
    int nFrames, firstFrame, framesCapacity, planesCapacity, cylindersCapacity, cells;
    uint32_t flags;
};

// N1 on the device: boundary polygons (cape_polygon.hip)
constexpr int kPolyListHeader = 40; // words in front of a polygon work list: [0] entries, [1] head / spare, [2] next entry, [3] planes to finish,
                                    // [4, 20) planes per size bucket, [20, 36) placed per bucket (static list of the task kernel)
constexpr int kPolyMaxPoints = 1024; // boundary candidates of one plane the device hull takes (more: CAPE_POLY_OVERFLOW, host class)
struct PolygonParams
{
    const cape_frame_record* records;
    const double* boundary;  // frames x boundaryCapacity x 3
    cape_polygon* polygons;  // frames x CAPE_MAX_PLANES
    double2* vertices;       // frames x boundaryCapacity plane-frame vertices (a plane's ring starts at its boundary_offset)
    int boundaryCapacity;
    uint32_t* lists;          // two work lists of listStride words (kPolyListHeader words, then entries): the planes of up to 256
    uint32_t listStride;      // candidates, the planes of 257 .. 1 024 candidates
    uint32_t* queue;          // the task kernel's queue of spawned (plane, rung) tasks: kPolyListHeader words ([0] tail, [1] head), then
    uint32_t queueCapacity;   // queueCapacity slots = polygon_queue_slots(frames the scratch was sized for)
    uint32_t* state;          // frames x CAPE_MAX_PLANES state words of the task kernel (done mask | hull mask << 8 | finalised << 16)
    unsigned short* park;     // frames x 6 rungs x parkStride: hulls waiting for the verdict of lower rungs
    uint32_t parkStride;      // boundaryCapacity + 2 * CAPE_MAX_PLANES (a plane's hull: length + at most count + 1 indices)
    int computeUnits;
    int originInCentroid;     // cape_debug_polygon only: the polygon's origin is read from the record's centroid field
    // spill records (frames of more than 64 plane segments, cape_frame_header::next_record): record poolBase + k, k < min(*poolUsed,
    // poolCapacity), is handled like one more frame -- every array above is indexed by the RECORD index and sized for the pool too
    int poolBase = 0, poolCapacity = 0;
    const uint32_t* poolUsed = nullptr;
    unsigned long long* prof; // [frames][kProfileSlots] phase ticks of a -DCAPE_POLY_PROFILE build (cape_debug_cycles), else unused
};

size_t polygon_queue_slots(size_t frames);
size_t polygon_scratch_bytes(size_t frames, int boundaryCapacity);
void polygon_bind_scratch(PolygonParams& p, void* base, size_t frames, int boundaryCapacity);

struct RcclUniqueId
{
    char internal[CAPE_COMM_ID_BYTES]; // ncclUniqueId
};

} // namespace cape
