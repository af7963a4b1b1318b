// C ABI of libcape_hip (see include/cape_hip.h).  Host side only: buffer ownership, constant tables, launches.
// There is no CPU fallback: without a HIP device cape_create fails with CAPE_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstring>
#include <new>
#include <random>
#include <string>
#include <vector>

#include "cape_internal.h"

namespace cape {
// every launcher returns the error of its own launch(es) (hipGetLastError right behind hipLaunchKernelGGL)
hipError_t launch_cell_moments(const StageAParams& p, int nFrames, hipStream_t stream);
hipError_t launch_cell_plane(const StageAParams& p, int nFrames, hipStream_t stream);
hipError_t launch_cell_strips(const StageAParams& p, int nFrames, uint32_t* frameCounters, hipStream_t stream);
int cell_plane_rows_per_tile(const StageAParams& p, int nFrames);
hipError_t launch_grow(const StageBParams& p, int nFrames, hipStream_t stream, hipStream_t side, hipEvent_t fork, hipEvent_t done,
                       const GenParams* gen);
hipError_t launch_grow_general(const StageBParams& p, const GenParams& g, int nFrames, hipStream_t stream);
size_t grow_lds_bytes(int cells, bool cylinders, int maxPlanes);
size_t grow_state_bytes(int cells);
bool resume_group_fits(const StageBParams& p);
hipError_t launch_rectify(const RectifyParams& p, int nFrames, int computeUnits, hipStream_t stream);
hipError_t launch_match(const MatchParams& p, int nFrames, hipStream_t stream);
hipError_t launch_pack(const PackParams& p, hipStream_t stream);
hipError_t launch_polygons(const PolygonParams& p, int nFrames, hipStream_t stream);
hipError_t launch_match_polygons(const MatchPolygonParams& p, int nFrames, hipStream_t stream);
const char* rccl_load(); // nullptr on success, else the reason
int rccl_unique_id(RcclUniqueId* id);
int rccl_comm_init(void** comm, int world, const RcclUniqueId& id, int rank);
int rccl_comm_destroy(void* comm);
int rccl_all_gather_bytes(const void* send, void* recv, size_t bytes, void* comm, hipStream_t stream);
const char* rccl_error_string(int code);
void rccl_comm_query(void* comm, int* count, int* rank, int* device);
bool rccl_has_gather();
int rccl_gather_bytes(const void* send, void* recv, size_t bytes, int root, void* comm, hipStream_t stream);
hipError_t launch_count_primitives(const cape_frame_record* records, int nFrames, int32_t* out, hipStream_t stream);
int grow_waves_per_group();
int grow_waves_per_cu(const StageBParams& p);
} // namespace cape

namespace {

thread_local std::string g_lastError;

int fail(int code, const std::string& msg)
{
    g_lastError = msg;
    return code;
}

#define CAPE_HIP_TRY(expr)                                                                                   \
    do                                                                                                       \
    {                                                                                                        \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            return fail(CAPE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                    \
    } while (0)

// completion signal of a chain whose results live in pinned host memory (see wait_results)
__global__ void cape_signal_kernel(uint32_t* flag, uint32_t seq)
{
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

constexpr int kHostResultFrames = 8; // see cape_handle_s::resultsOnHost
// RANSAC draws a frame can ask for (the table of the first draws of mt19937(seed) the handle keeps on the device): a run_ransac_loop
// takes at most 43 x 3 draws and every loop but a region's last removes at least six cells (cylinder_segment.cpp:154), so a frame of
// C cells runs at most C / 6 + (regions <= C / 6) loops: 43 C draws bound it.  40 000 covers the 640 x 480 grid's 33 024 as before.
constexpr int kRngTableMin = 40000;
int rng_table_size(int cells) { const int need = 43 * cells + 129; return need > kRngTableMin ? need : kRngTableMin; }

// Every entry point that allocates, copies, launches or synchronises runs with the HANDLE's device current, whatever
// the calling thread had selected (one process may drive several GPUs, torch may leave another device current), and
// gives the caller its device back on the way out.
class DeviceGuard
{
  public:
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&_prev) != hipSuccess)
            _prev = -1;
        _err = (_prev == device) ? hipSuccess : hipSetDevice(device);
        _restore = (_err == hipSuccess) && _prev >= 0 && _prev != device;
    }
    ~DeviceGuard()
    {
        if (_restore)
            (void)hipSetDevice(_prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
    hipError_t error() const { return _err; }

  private:
    int _prev = -1;
    hipError_t _err = hipSuccess;
    bool _restore = false;
};

#define CAPE_ON_DEVICE(h)                                                                                    \
    DeviceGuard _deviceGuard((h)->cfg.device);                                                               \
    if (_deviceGuard.error() != hipSuccess)                                                                  \
    return fail(CAPE_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(_deviceGuard.error()))

// A one-frame chain may have left the general grow instance to the first reader of its results (cape_handle_s::lazySpillArmed):
// an entry point that is about to enqueue work on those results settles that first (a host wait -- such handles serve one frame
// at a time, their callers wait for every call anyway).
#define CAPE_SETTLE_RESULTS(h)                                                                               \
    do                                                                                                       \
    {                                                                                                        \
        if ((h)->lazySpillArmed)                                                                             \
            if (const int rc_ = wait_results(h); rc_ != CAPE_OK)                                             \
                return rc_;                                                                                  \
    } while (0)

} // namespace

struct cape_handle_s
{
    cape_config cfg{};
    int hCells = 0, vCells = 0, cells = 0, boundaryCap = 0;
    // constants
    double* acol = nullptr;
    double* brow = nullptr;
    float* ratioCol = nullptr;
    float* ratioRow = nullptr;
    double* rng = nullptr;
    double* cylScratch = nullptr;
    uint32_t* needCylinder = nullptr; // [0] count, [1..] frames the plane-only pass handed to the cylinder kernel
    uint32_t* redoList = nullptr;     // [0] count, [1..] frames that need more than 32 plane-segment slots
    // the general grow instance (cape_grow_general.hip): frames of more than 64 plane segments / cylinder labels, every frame of a grid
    // beyond 64 x 64 cells
    uint32_t* spillList = nullptr;    // [0] count, [1..] frames the 64-segment instance handed on
    uint32_t* spillCounters = nullptr; // [0] records handed out of the pool by the call in flight, [1] frames through the general instance
    unsigned char* genScratch = nullptr;
    int spillRecords = 0;             // records of the pool = record indices [max_batch, max_batch + spillRecords)
    bool generalAll = false;          // the grid is beyond the fast kernels' 64 x 64 cells: the general instance grows every frame
    int rngCount = 0;                 // doubles of the RANSAC draw table (rng_table_size)
    cape::GenParams gen{};
    // one-frame handles (max_batch <= kHostResultFrames): stage A runs as ONE launch of strip workgroups (cape_cell_strip_kernel);
    // a counter per frame tells the strip that finishes last.  nullptr: the two throughput kernels (debug knob CAPE_STAGE_A=bands)
    uint32_t* stripCounters = nullptr;
    bool stripsAlways = false; // CAPE_STAGE_A=strips: also when the frame is read over the link (see launch_chain)
    bool inputOverLink = false; // the call in flight reads its frames straight from pinned host memory (cape_extract_host)
    bool pinnedByDma = false;   // debug knob CAPE_PINNED_INPUT=dma: pinned frames take the staging copy as well (A/B of the two routes)
    uint32_t* resumeList = nullptr;   // [0] count, [1..] frames handed to the cylinder kernel WITH their recorded regions
    unsigned char* growState = nullptr; // max_batch x grow_state_bytes(): the parked state of those frames
    // schedule feedback: the count of the last two-pass call is copied to pinned host memory behind the kernels and read
    // (never waited for) before the next call.  The cylinder kernel works in rounds of cylSlots resident frames; the
    // plane-only first pass pays off when it saves at least one such round (see launch_chain)
    uint32_t* handedOverHost = nullptr;
    hipEvent_t handedOverReady = nullptr;
    int handedOverFrames = 0;  // frames of the call the pending count belongs to (0: nothing pending)
    double handedOverFraction = 0.0; // last measured share of frames that took the cylinder branch
    int cylSlots = 1024;             // frames the cylinder kernel keeps resident on the device (occupancy x CUs)
    bool singlePass = false;
    int callsSinceProbe = 0;
    int forcedSchedule = 0; // debug knob CAPE_SCHEDULE=two|single (read at create): 1 = always two-pass, 2 = always single
    unsigned long long* debugCycles = nullptr;
    // rectify_depth (N3): float copies of the back-projection factors + collision keys (allocated on first use)
    float* xpre = nullptr;
    float* ypre = nullptr;
    unsigned* rectFlags = nullptr; // rectify_depth: a flag per frame, then the list of flagged frames
    size_t rectFlagFrames = 0;
    // plane matching between consecutive frames (N2): max_batch x cape_frame_match, allocated on first use
    cape_frame_match* matches = nullptr;
    // per-frame scratch (stage A -> stage B)
    double* cellSums = nullptr;
    double* cellPlane = nullptr;
    double* cellScore = nullptr;
    float* cellTol = nullptr;
    uint32_t* cellFlags = nullptr;
    int32_t* cellBins = nullptr;
    cape::CellAux* cellAux = nullptr;
    double* cellMse = nullptr;
    uint16_t* seedSeq = nullptr;
    // results
    cape_frame_record* records = nullptr;
    int32_t* planeLabels = nullptr;
    int32_t* cylLabels = nullptr;
    double* boundary = nullptr;
    // handles for a few frames at a time (max_batch <= kHostResultFrames) keep records / label grids / boundary points in
    // pinned, device-mapped HOST memory: the grow kernel's stores go straight over PCIe (posted writes), and reading the
    // results is a stream synchronisation + a host memcpy instead of three device-to-host copies
    bool resultsOnHost = false;
    // resultsOnHost: a one-thread kernel behind every chain stores a sequence number into this pinned word; whoever reads
    // the results spins on it instead of going through the runtime's stream synchronisation (wait_results)
    uint32_t* doneFlag = nullptr;
    uint32_t* doneCounter = nullptr; // device: the one-frame chain's grow kernel counts its waves out and stores the number itself
    uint32_t doneSeq = 0;      // sequence number of the last chain enqueued
    // one-frame chain: the general grow instance is enqueued by whoever waits for the results, and only if the chain's last wave
    // reported frames on the spill list (doneFlag[1], StageBParams::spillHost); the chain's parameters are kept for that launch
    bool lazySpillArmed = false;
    int lazySpillFrames = 0;
    cape::StageBParams lazySpillParams{};
    bool doneArmed = false;    // a chain with a signal behind it is (or was) in flight
    // host staging for cape_extract_host
    float* depthStage = nullptr;
    // timing: one event triple per timed cape_extract, folded lazily by cape_get_timings
    struct EvTriple
    {
        hipEvent_t e[4];
        hipEvent_t e2b = nullptr; // start of the A2 kernel when it runs on another stream than A1 (pipelined mode)
        int frames;
        bool split = false;
    };
    std::vector<EvTriple> evPool;   // created on demand, reused
    size_t evPending = 0;           // triples [0, evPending) hold unread measurements
    bool timing = false;
    cape_timings tm{};
    unsigned long long* phaseTicks = nullptr; // device, max_batch x 4 u64: ticks in grow / merge / refine of the timed calls (StageBParams::phaseTicks)
    // cape_set_log_callback
    cape_log_fn logFn = nullptr;
    void* logUser = nullptr;
    bool logPending = false; // a batch has been extracted whose records have not been through the callback yet
    int logDone = 0;         // ... frames [0, logDone) of it have (a cape_copy_results of fewer frames than the batch delivers the rest later)
    // sub-batch pipelining (cfg.sub_batches > 1)
    hipStream_t pipeStream[2] = {nullptr, nullptr};
    hipEvent_t pipeFork = nullptr;
    hipEvent_t pipeJoin[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> pipeStage;
    int lastFrames = 0;
    // Per-handle scratch (depth staging, rectify keys, hand-over feedback, result buffers) is reused from call to call
    // without per-buffer events: ONE stream is in flight per handle.  Every enqueueing call leaves a handle-owned event
    // behind its work (StreamScope); a call that arrives on ANOTHER stream makes that stream wait for the event
    // (hipStreamWaitEvent: no host block, and the previous stream's handle is never touched again -- the caller may have
    // destroyed it).  lastStream is only compared, never dereferenced.
    hipStream_t lastStream = nullptr;
    bool hasLastStream = false;
    int scopeDepth = 0;             // nesting of entry points (StreamScope)
    // CAPE_FLAG_ASYNC_SECOND_PASS: the cylinder second pass runs on the handle's own stream; whoever touches the handle next
    // (any entry point, cape_destroy) first waits for sideDone
    hipStream_t sideStream = nullptr;
    hipEvent_t sideFork = nullptr, sideDone = nullptr;
    bool sidePending = false;
    hipEvent_t workDone = nullptr;  // recorded behind the last enqueued work of this handle
    bool workRecorded = false;
    // multi-GPU gather: two packed staging slots, the RCCL communicator and its stream
    cape_gather_config gatherCfg{};
    cape_gather_layout gatherLayout{};
    unsigned char* packed[2] = {nullptr, nullptr};
    hipEvent_t packedFree[2] = {nullptr, nullptr}; // recorded behind the gather that read the slot
    bool packedBusy[2] = {false, false};
    int packSlot = 1;                              // slot filled by the last cape_pack_primitives
    hipEvent_t packReady = nullptr;
    void* comm = nullptr;
    int commRank = 0, commWorld = 0;
    hipStream_t commStream = nullptr;
    hipEvent_t gatherDone = nullptr;
    bool gatherPending = false;
    int32_t* countScratch = nullptr; // cape_count_primitives
    // N1 on the device: polygons of the last batch (allocated on first use)
    cape_polygon* polygons = nullptr;
    double* polyVertices = nullptr;
    uint32_t* polyLadder = nullptr; // scratch of the polygon kernels: work lists, state words, parking area (polygon_scratch_bytes)
    int polygonFrames = 0;          // frames of the last cape_build_polygons (0: none for the current batch)
    int matchExactFrames = 0;       // frames of the last cape_match_polygons (0: none for the current batch)
    double* matchPoses = nullptr;   // cape_match_polygons_pose: max_batch x 16 doubles, allocated on first use
    double* matchPosesStage = nullptr; // pinned twin the caller's poses are copied into before the call returns
    hipEvent_t matchPosesFree = nullptr; // recorded behind the H2D copy out of the twin: its next writer waits for it
    bool matchPosesBusy = false;         // ... once it has been recorded
    cape_frame_match_exact* matchesExact = nullptr;
    unsigned* matchLists = nullptr; // counters (padded to 64 entries) + 4 lists of max_batch x 256 pairs
    int computeUnits = 0;           // CUs of the handle's device (queried on first use)
    int ldsLimit = 0; // LDS bytes one workgroup may use on this device (hipDeviceAttributeMaxSharedMemoryPerBlock)
    cape::StageAParams pa{};
    cape::StageBParams pb{};
};

namespace {

// Matrix3d::inverse as Eigen evaluates it (cofactor method); for K = [[fx,0,cx],[0,fy,cy],[0,0,1]] this yields
// k00 = fy*invdet, k02 = -(cx*fy)*invdet, k11 = fx*invdet, k12 = -(fx*cy)*invdet with invdet = 1/(fx*fy)
// (reference src/coordinates/point_coordinates.cpp:79-83, Parameters::get_camera_1_intrinsics parameters.hpp:144-149)
void inverse_intrinsics(double fx, double fy, double cx, double cy, double& k00, double& k02, double& k11, double& k12)
{
    const double m[3][3] = {{fx, 0.0, cx}, {0.0, fy, cy}, {0.0, 0.0, 1.0}};
    auto cof = [&](int i, int j) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
    };
    const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const double det = (c00 * m[0][0] + c10 * m[1][0]) + c20 * m[2][0];
    const double invdet = 1.0 / det;
    k00 = c00 * invdet;
    k02 = c20 * invdet;
    k11 = cof(1, 1) * invdet;
    k12 = cof(2, 1) * invdet;
}

template <typename T> hipError_t dalloc(T*& p, size_t n) { return hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)); }

void free_all(cape_handle_s* h)
{
    (void)hipFree(h->acol);
    (void)hipFree(h->brow);
    (void)hipFree(h->ratioCol);
    (void)hipFree(h->ratioRow);
    (void)hipFree(h->rng);
    (void)hipFree(h->cylScratch);
    (void)hipFree(h->needCylinder);
    (void)hipFree(h->redoList);
    (void)hipFree(h->spillList);
    (void)hipFree(h->spillCounters);
    (void)hipFree(h->genScratch);
    (void)hipFree(h->stripCounters);
    (void)hipFree(h->doneCounter);
    (void)hipFree(h->resumeList);
    (void)hipFree(h->growState);
    if (h->handedOverHost)
        (void)hipHostFree(h->handedOverHost);
    if (h->handedOverReady)
        (void)hipEventDestroy(h->handedOverReady);
    (void)hipFree(h->debugCycles);
    (void)hipFree(h->phaseTicks);
    (void)hipFree(h->countScratch);
    (void)hipFree(h->polyLadder);
    (void)hipFree(h->matchesExact);
    (void)hipFree(h->matchLists);
    (void)hipFree(h->matchPoses);
    if (h->matchPosesStage)
        (void)hipHostFree(h->matchPosesStage);
    if (h->matchPosesFree)
        (void)hipEventDestroy(h->matchPosesFree);
    if (h->resultsOnHost)
    {
        if (h->polygons)
            (void)hipHostFree(h->polygons);
        if (h->polyVertices)
            (void)hipHostFree(h->polyVertices);
    }
    else
    {
        (void)hipFree(h->polygons);
        (void)hipFree(h->polyVertices);
    }
    (void)hipFree(h->xpre);
    (void)hipFree(h->ypre);
    (void)hipFree(h->rectFlags);
    (void)hipFree(h->matches);
    (void)hipFree(h->cellSums);
    (void)hipFree(h->cellPlane);
    (void)hipFree(h->cellScore);
    (void)hipFree(h->cellTol);
    (void)hipFree(h->cellFlags);
    (void)hipFree(h->cellBins);
    (void)hipFree(h->cellAux);
    (void)hipFree(h->cellMse);
    (void)hipFree(h->seedSeq);
    if (h->resultsOnHost)
    {
        (void)hipHostFree(h->records);
        (void)hipHostFree(h->planeLabels);
        (void)hipHostFree(h->cylLabels);
        (void)hipHostFree(h->boundary);
        (void)hipHostFree(h->doneFlag);
    }
    else
    {
        (void)hipFree(h->records);
        (void)hipFree(h->planeLabels);
        (void)hipFree(h->cylLabels);
        (void)hipFree(h->boundary);
    }
    (void)hipFree(h->depthStage);
    if (h->comm)
        (void)cape::rccl_comm_destroy(h->comm);
    h->comm = nullptr;
    for (int k = 0; k < 2; ++k)
    {
        (void)hipFree(h->packed[k]);
        if (h->packedFree[k])
            (void)hipEventDestroy(h->packedFree[k]);
    }
    if (h->packReady)
        (void)hipEventDestroy(h->packReady);
    if (h->gatherDone)
        (void)hipEventDestroy(h->gatherDone);
    if (h->commStream)
        (void)hipStreamDestroy(h->commStream);
    for (auto& t : h->evPool)
    {
        for (auto& e : t.e)
            if (e)
                (void)hipEventDestroy(e);
        if (t.e2b)
            (void)hipEventDestroy(t.e2b);
    }
    for (auto& e : h->pipeStage)
        if (e)
            (void)hipEventDestroy(e);
    for (auto& st : h->pipeStream)
        if (st)
            (void)hipStreamDestroy(st);
    if (h->pipeFork)
        (void)hipEventDestroy(h->pipeFork);
    if (h->workDone)
        (void)hipEventDestroy(h->workDone);
    if (h->sideStream)
        (void)hipStreamDestroy(h->sideStream);
    if (h->sideFork)
        (void)hipEventDestroy(h->sideFork);
    if (h->sideDone)
        (void)hipEventDestroy(h->sideDone);
    for (auto& e : h->pipeJoin)
        if (e)
            (void)hipEventDestroy(e);
}

// parameter blocks of a sub-batch that starts at frame f0
void offset_params(const cape_handle_s* h, int f0, cape::StageAParams& a, cape::StageBParams& b)
{
    a = h->pa;
    b = h->pb;
    const size_t C = (size_t)h->cells, F = (size_t)f0;
    const size_t px = F * (size_t)h->cfg.width * h->cfg.height;
    if (a.depth)
        a.depth += px;
    if (a.depth_u16)
        a.depth_u16 += px;
    a.cell_sums += F * C * cape::kSumStride;
    a.cell_plane += F * C * cape::kPlaneStride;
    a.cell_score += F * C;
    a.cell_tol += F * C;
    a.cell_flags += F * C;
    a.cell_bins += F * C;
    a.cell_aux += F * C;
    a.cell_mse += F * C;
    b.cell_aux = a.cell_aux;
    b.cell_mse = a.cell_mse;
    b.cell_sums = a.cell_sums;
    b.cell_plane = a.cell_plane;
    b.cell_score = a.cell_score;
    b.cell_tol = a.cell_tol;
    b.cell_flags = a.cell_flags;
    b.cell_bins = a.cell_bins;
    b.records += F;
    b.plane_labels += F * C;
    b.cyl_labels += F * C;
    b.boundary += F * (size_t)h->boundaryCap * 3;
    if (b.cylScratch)
        b.cylScratch += F * C * cape::kCylStride;
    if (b.needCylinder)
        b.needCylinder += 2 * F; // a sub-batch of n frames uses 1 + n entries of its own
    if (b.resumeList)
    {
        b.resumeList += 2 * F;
        b.growState += F * (size_t)b.growStateStride;
    }
    b.redoList += 2 * F;
    b.spillList += 2 * F;
    b.seed_sequence += F * C;
    b.debugCycles += F * cape::kProfileSlots;
}

int fold_timings(cape_handle_s* h);
int wait_results(cape_handle_s* h);

// utils::Random (src/utils/random.hpp:17-30, :59-64): the first kRngTable doubles of mt19937(seed) + uniform_real_distribution(0, 1)
// (libstdc++ on the host = the reference's own generator)
std::vector<double> rng_table(uint32_t seed, int count)
{
    std::vector<double> rng((size_t)count);
    std::mt19937 engine(seed);
    std::uniform_real_distribution<double> dist(0.0, 1.0);
    for (auto& v : rng)
        v = dist(engine);
    return rng;
}

// next free event set for one timed kernel chain (creates / recycles on demand)
int acquire_events(cape_handle_s* h, int frames, cape_handle_s::EvTriple** out)
{
    *out = nullptr;
    if (!h->timing)
        return CAPE_OK;
    if (h->evPending == h->evPool.size())
    {
        if (h->evPool.size() >= 4096)
        {
            const int rc = fold_timings(h); // synchronises; keeps the pool bounded
            if (rc != CAPE_OK)
                return rc;
        }
        else
        {
            cape_handle_s::EvTriple nt{};
            for (auto& e : nt.e)
                CAPE_HIP_TRY(hipEventCreate(&e));
            CAPE_HIP_TRY(hipEventCreate(&nt.e2b));
            h->evPool.push_back(nt);
        }
    }
    *out = &h->evPool[h->evPending];
    (*out)->frames = frames;
    (*out)->split = h->cfg.sub_batches > 1;
    h->evPending += 1;
    return CAPE_OK;
}

// one stream in flight per handle (see cape_handle_s::lastStream): a call on another stream is ordered behind the
// handle's previous work on the device, through the handle's own event
int enter_stream(cape_handle_s* h, hipStream_t st)
{
    const bool other = h->hasLastStream && h->lastStream != st;
    h->lastStream = st;
    h->hasLastStream = true;
    if (other && h->workRecorded)
        CAPE_HIP_TRY(hipStreamWaitEvent(st, h->workDone, 0));
    if (h->sidePending)
    {
        // the previous call's second pass is still on the handle's side stream: this call (and the caller's stream from here
        // on) is ordered behind it
        CAPE_HIP_TRY(hipStreamWaitEvent(st, h->sideDone, 0));
        h->sidePending = false;
    }
    return CAPE_OK;
}

// everything this handle has enqueued so far is done (host side)
hipError_t drain_handle(cape_handle_s* h)
{
    if (h->sidePending)
    {
        if (const hipError_t e = hipEventSynchronize(h->sideDone); e != hipSuccess)
            return e;
        h->sidePending = false;
    }
    if (h->workRecorded)
        return hipEventSynchronize(h->workDone);
    return hipSuccess;
}

// Brackets the enqueueing part of an entry point: orders the call behind the handle's earlier work (enter_stream) and,
// on the way out -- whether or not a launch in between failed -- records the handle's event behind what was enqueued.
class StreamScope
{
  public:
    StreamScope(cape_handle_s* h, hipStream_t st) : _h(h), _st(st)
    {
        _outer = h->scopeDepth++ == 0; // an entry point that calls another one (cape_extract_host -> cape_extract) records once
        _rc = _outer ? enter_stream(h, st) : CAPE_OK;
    }
    ~StreamScope()
    {
        --_h->scopeDepth;
        if (_outer && _rc == CAPE_OK && _h->workDone && hipEventRecord(_h->workDone, _st) == hipSuccess)
            _h->workRecorded = true;
    }
    StreamScope(const StreamScope&) = delete;
    StreamScope& operator=(const StreamScope&) = delete;
    int rc() const { return _rc; }

  private:
    cape_handle_s* _h;
    hipStream_t _st;
    int _rc;
    bool _outer;
};

// one kernel chain (A1 -> A2 -> B) on `st`, optionally bracketed by timing events
int launch_chain(cape_handle_s* h, const cape::StageAParams& a, const cape::StageBParams& b, int frames, hipStream_t st)
{
    cape_handle_s::EvTriple* t = nullptr;
    const int rc = acquire_events(h, frames, &t);
    if (rc != CAPE_OK)
        return rc;
    if (t)
        CAPE_HIP_TRY(hipEventRecord(t->e[0], st));
    cape::StageAParams a2 = a;
    a2.clear0 = b.redoList;
    a2.clear1 = b.needCylinder;
    a2.clear2 = b.resumeList;
    a2.clear2Buckets = b.resumeList ? b.resumeBucketStride : 0u;
    a2.clear3 = b.spillList;
    a2.clear4 = h->spillCounters;
    // A frame read straight from pinned host memory arrives at the link's pace (~34 us for 1.2 MB): the band kernel streams it in
    // and the plane kernel's 17 us follow; a strip's tail behind its last pixel is as long, so nothing is gained there (measured,
    // profiles/r04_single_frame_latency.txt).  With the frame in HBM the one-launch form is 5-10 us faster.
    const bool strips = h->stripCounters != nullptr && frames <= kHostResultFrames && (!h->inputOverLink || h->stripsAlways);
    if (strips)
    {
        // the latency instance: all of stage A in one launch (timing: booked as the moments kernel, the plane kernel reads 0)
        CAPE_HIP_TRY(cape::launch_cell_strips(a2, frames, h->stripCounters, st));
        if (t)
        {
            CAPE_HIP_TRY(hipEventRecord(t->e[1], st));
            CAPE_HIP_TRY(hipEventRecord(t->e[2], st));
        }
    }
    else
    {
        CAPE_HIP_TRY(cape::launch_cell_moments(a, frames, st));
        if (t)
            CAPE_HIP_TRY(hipEventRecord(t->e[1], st));
        CAPE_HIP_TRY(cape::launch_cell_plane(a2, frames, st));
        if (t)
            CAPE_HIP_TRY(hipEventRecord(t->e[2], st));
    }
    cape::StageBParams bb = b;
    bb.phaseTicks = t ? h->phaseTicks : nullptr; // the reference's grow / merge / refine buckets, only while timing is on
    bb.a2RowsPerTile = strips ? a.vCells : cape::cell_plane_rows_per_tile(a, frames);
    bb.countersCleared = 1;
    // The one-frame chain: stage A, then ONE grow kernel -- the 64-segment instance on every frame of the call, no
    // 32-segment pass in front, no redo pass and no one-thread signal kernel behind: its last wave stores the sequence number the
    // host spins on.  Two launches instead of four or five on the path the reference calls (CAPE_STAGE_A=bands: the classic chain).
    const bool oneFrameChain = h->resultsOnHost && h->doneFlag && h->doneCounter && frames <= kHostResultFrames;
    if (oneFrameChain)
    {
        bb.allFrames = h->generalAll ? 0 : 1;
        bb.doneFlag = h->doneFlag;
        bb.doneCounter = h->doneCounter;
        bb.doneSeq = ++h->doneSeq;
        bb.spillHost = h->generalAll ? nullptr : h->doneFlag + 1;
    }
    if (bb.needCylinder)
    {
        // Cost model, in rounds of the cylinder kernel (one round = cylSlots resident frames, ~0.25 ms at 640x480):
        //   cylinder kernel alone      ceil(frames / slots)
        //   plane-only pass first      kPlanePassPerRound * frames / slots  +  ceil(handed_over / slots)
        // kPlanePassPerRound = 0.29 is the measured cost of growing one round's worth of frames with the plane-only
        // kernel (profiles/schedule_crossover.py).  A single handed-over frame is cheaper alone; a batch that hands
        // over half of its frames usually saves a round.
        constexpr double kPlanePassPerRound = 0.29;
        constexpr int kProbeEvery = 32; // a single-pass handle re-measures with a two-pass call now and then
        if (h->handedOverFrames > 0 && hipEventQuery(h->handedOverReady) == hipSuccess)
        {
            h->handedOverFraction = (double)(h->handedOverHost[0] + h->handedOverHost[1]) / (double)h->handedOverFrames; // redone + parked
            h->handedOverFrames = 0;
        }
        {
            const double slots = (double)(h->cylSlots > 0 ? h->cylSlots : 1024);
            const double alone = std::ceil((double)frames / slots);
            // a parked frame is finished, not grown again: its round of the second pass is shorter (measured, 640x480: 0.12 ms
            // per 1 024 tunnel frames by the lone-wave RESUME instance against 0.15 ms for the full kernel; the workgroup
            // kernel of the wide grids: 0.5 ms against 0.97 ms per 1 024 frames of 1280x960)
            const double secondPassPerRound = !bb.resumeList ? 1.0 : (bb.resumeMode == 2 ? 0.5 : 0.8);
            const double twoPass = kPlanePassPerRound * (double)frames / slots +
                                   secondPassPerRound * std::ceil(h->handedOverFraction * (double)frames / slots);
            h->singlePass = twoPass >= alone;
            // a handful of frames (the reference's one-frame call pattern): what counts is the number of launches on the
            // latency path, and the cylinder kernel alone is one launch instead of four
            if (frames <= kHostResultFrames)
                h->singlePass = true;
        }
        const bool probe = h->singlePass && frames > kHostResultFrames && ++h->callsSinceProbe >= kProbeEvery;
        bb.twoPass = (!h->singlePass || probe) ? 1 : 0;
        if (h->forcedSchedule)
            bb.twoPass = h->forcedSchedule == 1 ? 1 : 0;
        if (probe)
            h->callsSinceProbe = 0;
    }
    CAPE_HIP_TRY(cape::launch_grow(bb, frames, st, h->generalAll ? nullptr : h->sideStream, h->sideFork, h->sideDone, &h->gen));
    if (h->sideStream && bb.needCylinder && !h->generalAll)
        h->sidePending = true;
    if (bb.needCylinder && bb.twoPass && h->handedOverFrames == 0 && !h->generalAll)
    {
        CAPE_HIP_TRY(hipMemcpyAsync(h->handedOverHost, bb.needCylinder, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        if (bb.resumeList)
            CAPE_HIP_TRY(hipMemcpyAsync(h->handedOverHost + 1, bb.resumeList, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        CAPE_HIP_TRY(hipEventRecord(h->handedOverReady, st));
        h->handedOverFrames = frames;
    }
    if (t)
        CAPE_HIP_TRY(hipEventRecord(t->e[3], h->sidePending ? h->sideStream : st));
    if (h->resultsOnHost && h->doneFlag)
    {
        if (!oneFrameChain)
        {
            hipLaunchKernelGGL(cape_signal_kernel, dim3(1), dim3(1), 0, st, h->doneFlag, ++h->doneSeq);
            CAPE_HIP_TRY(hipGetLastError());
        }
        h->doneArmed = true;
    }
    h->lazySpillArmed = bb.spillHost != nullptr;
    if (h->lazySpillArmed)
    {
        h->lazySpillParams = bb;
        h->lazySpillFrames = frames;
    }
    return CAPE_OK;
}

// Results in pinned host memory: wait for the signal word of the last chain.  The runtime's hipStreamSynchronize costs
// ~10 us of wake-up on top of the kernels when the whole call is ~130 us; a spin on a pinned word costs a PCIe write.
// If the word does not arrive in time (a faulted kernel, a descheduled process) the stream synchronisation takes over
// and reports whatever went wrong.
int wait_results_once(cape_handle_s* h);
int wait_results(cape_handle_s* h)
{
    if (const int rc = wait_results_once(h); rc != CAPE_OK)
        return rc;
    if (h->lazySpillArmed)
    {
        // the one-frame chain left the general grow instance to us: a frame of more than 64 plane segments / cylinder labels is on
        // the spill list (the word next to the completion word says how many) -- enqueue that kernel now and wait for its signal
        h->lazySpillArmed = false;
        if (h->doneFlag && h->doneFlag[1] != 0u)
        {
            cape::StageBParams p = h->lazySpillParams;
            p.spillHost = nullptr;
            p.allFrames = 0;
            p.doneSeq = ++h->doneSeq;
            CAPE_HIP_TRY(cape::launch_grow_general(p, h->gen, h->lazySpillFrames, h->lastStream));
            if (h->workDone && hipEventRecord(h->workDone, h->lastStream) == hipSuccess)
                h->workRecorded = true;
            h->doneArmed = true;
            return wait_results_once(h);
        }
    }
    return CAPE_OK;
}

int wait_results_once(cape_handle_s* h)
{
    if (h->doneArmed && h->doneFlag)
    {
        volatile const uint32_t* flag = h->doneFlag;
        const auto t0 = std::chrono::steady_clock::now();
        for (int spin = 0;; ++spin)
        {
            if (*flag == h->doneSeq)
            {
                std::atomic_thread_fence(std::memory_order_acquire); // the results are read after the word
                return CAPE_OK;
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#elif defined(__aarch64__)
            __asm__ __volatile__("yield");
#endif
            if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5))
                break;
        }
    }
    if (h->workRecorded)
        CAPE_HIP_TRY(hipEventSynchronize(h->workDone));
    else
        CAPE_HIP_TRY(hipDeviceSynchronize());
    return CAPE_OK;
}

int fold_timings(cape_handle_s* h)
{
    if (h->evPending == 0 && h->tm.calls > 0)
        return CAPE_OK; // nothing timed since the last fold: the sums (and the tick slots behind the split) are unchanged
    for (size_t i = 0; i < h->evPending; ++i)
    {
        auto& t = h->evPool[i];
        CAPE_HIP_TRY(hipEventSynchronize(t.e[3]));
        float a1 = 0, a2 = 0, b = 0;
        CAPE_HIP_TRY(hipEventElapsedTime(&a1, t.e[0], t.e[1]));
        CAPE_HIP_TRY(hipEventElapsedTime(&a2, t.split ? t.e2b : t.e[1], t.e[2]));
        CAPE_HIP_TRY(hipEventElapsedTime(&b, t.e[2], t.e[3]));
        h->tm.cell_moments_s += a1 * 1e-3;
        h->tm.cell_plane_s += a2 * 1e-3;
        h->tm.cell_fit_s += (a1 + a2) * 1e-3;
        h->tm.grow_s += b * 1e-3;
        h->tm.total_s += (a1 + a2 + b) * 1e-3;
        h->tm.frames += (uint64_t)t.frames;
        h->tm.calls += 1;
    }
    h->evPending = 0;
    // the reference's buckets: stage B's event time split by the ticks its waves booked (all timed calls since the last reset)
    h->tm.reset_s = 0.0;
    h->tm.init_s = h->tm.cell_fit_s;
    h->tm.grow_phase_s = h->tm.grow_s;
    h->tm.merge_s = h->tm.refine_s = 0.0;
    if (h->phaseTicks && h->tm.calls > 0)
    {
        unsigned long long ticks[4] = {0, 0, 0, 0};
        std::vector<unsigned long long> slots((size_t)h->cfg.max_batch * 4);
        CAPE_HIP_TRY(hipMemcpy(slots.data(), h->phaseTicks, slots.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < slots.size(); ++i)
            ticks[i & 3] += slots[i];
        const double all = (double)ticks[0] + (double)ticks[1] + (double)ticks[2];
        if (all > 0)
        {
            h->tm.merge_s = h->tm.grow_s * ((double)ticks[1] / all);
            h->tm.refine_s = h->tm.grow_s * ((double)ticks[2] / all);
            h->tm.grow_phase_s = h->tm.grow_s - h->tm.merge_s - h->tm.refine_s;
        }
    }
    return CAPE_OK;
}

// the reference's hot-path log lines of one frame, from its record (cape_set_log_callback).  `next`: where the records a frame
// continues in (cape_frame_header::next_record) are found -- null if the caller has none of them at hand
template <typename NextFn> int log_frame(cape_log_fn fn, void* user, const cape_frame_record& r, int frame, NextFn next)
{
    int lines = 0;
    const uint32_t st = r.header.status;
    // grow_planes_and_cylinders comes first in the reference (the seed loop's lines, in the order they fall; only their NUMBER
    // travels in the record), add_planes_to_primitives behind it
    for (uint32_t k = 0; k < CAPE_FRAME_NOT_PLANAR_COUNT(st); ++k, ++lines)
        fn(0, "Plane segment is not planar after merge", frame, user);
    if (st & CAPE_FRAME_INVALID_SEED) // ends the seed loop: the last line of the grow step
        fn(1, "Could not find a single plane segment: invalid seed", frame, user), ++lines;
    // (a chain only ever points FORWARD in the record array -- a spill record's index is beyond the batch's and its successor's beyond
    // its own --, so a zero-filled or damaged link cannot loop)
    int at = frame;
    for (const cape_frame_record* q = &r; q;)
    {
        const int n = q->header.n_plane_segments < CAPE_MAX_PLANES ? q->header.n_plane_segments : CAPE_MAX_PLANES;
        for (int i = 0; i < n; ++i)
        {
            const cape_plane_segment& s = q->segments[i];
            if (s.merge_label == (uint32_t)(i + q->header.segment_base) && s.planar && s.boundary_count < 3)
                fn(1, "Could not find a correct boundary polygon, rejecting plane segment", frame, user), ++lines;
        }
        const int nxt = q->header.next_record;
        q = nxt > at ? next(nxt) : nullptr;
        at = nxt;
    }
    if (st & (CAPE_FRAME_PLANE_OVERFLOW | CAPE_FRAME_CYL_OVERFLOW | CAPE_FRAME_BOUNDARY_OVERFLOW))
        fn(1, "find_primitives: per-frame capacity exceeded, primitive list truncated", frame, user), ++lines;
    return lines;
}
void log_batch(cape_handle_s* h, const cape_frame_record* records, int n)
{
    if (!h->logFn || !h->logPending || !records)
        return;
    const int upTo = n < h->lastFrames ? n : h->lastFrames;
    // spill records (frames of more than 64 plane segments) are fetched from the handle's pool when a frame points at one
    cape_frame_record spill;
    auto next = [&](int idx) -> const cape_frame_record* {
        if (idx < h->cfg.max_batch || idx >= h->cfg.max_batch + h->spillRecords)
            return nullptr;
        if (h->resultsOnHost)
            return h->records + idx;
        return hipMemcpy(&spill, h->records + idx, sizeof(spill), hipMemcpyDeviceToHost) == hipSuccess ? &spill : nullptr;
    };
    for (int f = h->logDone; f < upTo; ++f) // every frame of the batch once, whichever copy brings it to the host first
        (void)log_frame(h->logFn, h->logUser, records[f], f, next);
    if (upTo > h->logDone)
        h->logDone = upTo;
    if (h->logDone >= h->lastFrames)
        h->logPending = false;
}

} // namespace

extern "C" {

const char* cape_last_error(void) { return g_lastError.c_str(); }
const char* cape_version(void) { return "cape_hip 0.2 (gfx950)"; }
int32_t cape_abi_version(void) { return CAPE_ABI_VERSION; }

int cape_device_count(int32_t* count_out)
{
    if (!count_out)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    int ndev = 0;
    *count_out = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(CAPE_ERR_NO_DEVICE, "no HIP device: libcape_hip has no CPU fallback");
    *count_out = ndev;
    return CAPE_OK;
}

int cape_create(const cape_config* cfg, cape_handle* out)
{
    if (!cfg || !out)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    // any image the reference's constructor takes (primitive_detection.cpp:26-67: whole cells of 20 px) up to the widths of the
    // index types used here: 16-bit cell numbers (65 535 cells), one stage-A2 tile row per workgroup (256 cells = 5120 px wide)
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->width % CAPE_CELL_SIZE || cfg->height % CAPE_CELL_SIZE || cfg->max_batch <= 0 ||
        cfg->spill_records < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "width/height must be positive multiples of 20; max_batch > 0; spill_records >= 0");
    if (cfg->width / CAPE_CELL_SIZE > 256 || (long long)(cfg->width / CAPE_CELL_SIZE) * (cfg->height / CAPE_CELL_SIZE) > 65535)
        return fail(CAPE_ERR_UNSUPPORTED, "cell grid beyond 256 cells wide (5120 px) or 65 535 cells");
    if (!(cfg->fx > 0) || !(cfg->fy > 0))
        return fail(CAPE_ERR_INVALID_ARGUMENT, "focal lengths must be positive");
    if (cfg->flags & ~(uint32_t)(CAPE_FLAG_CYLINDERS | CAPE_FLAG_ASYNC_SECOND_PASS))
        return fail(CAPE_ERR_INVALID_ARGUMENT, "unknown CAPE_FLAG_* bit in cape_config.flags");
    // the debug knobs are read here, once per handle, and a value they do not know is an error, not a silent default
    if (const char* e = std::getenv("CAPE_RESUME"))
        if (std::string(e) != "off" && std::string(e) != "wave" && std::string(e) != "group")
            return fail(CAPE_ERR_INVALID_ARGUMENT, "CAPE_RESUME must be off, wave or group");
    if (const char* e = std::getenv("CAPE_SCHEDULE"))
        if (std::string(e) != "two" && std::string(e) != "single")
            return fail(CAPE_ERR_INVALID_ARGUMENT, "CAPE_SCHEDULE must be two or single");
    if (const char* e = std::getenv("CAPE_STAGE_A"))
        if (std::string(e) != "strips" && std::string(e) != "bands")
            return fail(CAPE_ERR_INVALID_ARGUMENT, "CAPE_STAGE_A must be strips or bands");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(CAPE_ERR_NO_DEVICE, "no HIP device: libcape_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    DeviceGuard deviceGuard(cfg->device);
    if (deviceGuard.error() != hipSuccess)
        return fail(CAPE_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(deviceGuard.error()));
    int ldsLimit = 0;
    CAPE_HIP_TRY(hipDeviceGetAttribute(&ldsLimit, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device));

    cape_handle_s* h = new (std::nothrow) cape_handle_s();
    if (!h)
        return fail(CAPE_ERR_HIP, "out of host memory");
    h->cfg = *cfg;
    h->ldsLimit = ldsLimit;
    h->hCells = cfg->width / CAPE_CELL_SIZE;
    h->vCells = cfg->height / CAPE_CELL_SIZE;
    h->cells = h->hCells * h->vCells;
    h->boundaryCap = cfg->boundary_capacity > 0 ? cfg->boundary_capacity : 2 * h->cells;
    const size_t B = (size_t)cfg->max_batch, C = (size_t)h->cells;
    h->generalAll = h->hCells > 128 || h->vCells > 64; // beyond "one lane per grid row, two mask words per row"
    if (const char* e = std::getenv("CAPE_GROW")) // debug knob: "general" sends every frame through the general instance
    {
        if (std::string(e) == "general")
            h->generalAll = true;
        else if (std::string(e) != "fast")
        {
            delete h;
            return fail(CAPE_ERR_INVALID_ARGUMENT, "CAPE_GROW must be general or fast");
        }
    }
    h->spillRecords = cfg->spill_records > 0 ? cfg->spill_records : std::max(8, cfg->max_batch / 8);
    h->rngCount = rng_table_size(h->cells);
    const size_t R = B + (size_t)h->spillRecords; // records / boundary slabs: the batch's, then the spill pool

#define CAPE_ALLOC(expr)                                                                                     \
    do                                                                                                       \
    {                                                                                                        \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
        {                                                                                                    \
            free_all(h);                                                                                     \
            delete h;                                                                                        \
            return fail(CAPE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                    \
        }                                                                                                    \
    } while (0)

    CAPE_ALLOC(dalloc(h->acol, cfg->width));
    CAPE_ALLOC(dalloc(h->brow, cfg->height));
    CAPE_ALLOC(dalloc(h->ratioCol, h->hCells));
    CAPE_ALLOC(dalloc(h->ratioRow, h->vCells));
    CAPE_ALLOC(dalloc(h->rng, (size_t)h->rngCount));
    CAPE_ALLOC(dalloc(h->cellSums, B * C * cape::kSumStride));
    CAPE_ALLOC(dalloc(h->cellPlane, B * C * cape::kPlaneStride));
    CAPE_ALLOC(dalloc(h->cellScore, B * C));
    CAPE_ALLOC(dalloc(h->cellTol, B * C));
    CAPE_ALLOC(dalloc(h->cellFlags, B * C));
    CAPE_ALLOC(dalloc(h->cellBins, B * C));
    CAPE_ALLOC(dalloc(h->cellAux, B * C));
    CAPE_ALLOC(dalloc(h->cellMse, B * C));
    CAPE_ALLOC(dalloc(h->seedSeq, B * C));
    if (cfg->flags & CAPE_FLAG_CYLINDERS)
    {
        CAPE_ALLOC(dalloc(h->cylScratch, B * C * cape::kCylStride));
        CAPE_ALLOC(dalloc(h->needCylinder, 2 * B + 2));
        // debug knob CAPE_RESUME=off: the round-2 schedule (every handed-over frame is grown again from scratch)
        const char* resumeEnv = std::getenv("CAPE_RESUME");
        if (!(resumeEnv && std::string(resumeEnv) == "off") && !std::getenv("CAPE_NO_RESUME"))
        {
            // the list + its cost-class lists (StageBParams::resumeBucketStride), one stride apart
            CAPE_ALLOC(dalloc(h->resumeList, (1 + cape::kResumeClasses) * (2 * B + 2)));
            CAPE_ALLOC(hipMemset(h->resumeList, 0, (1 + cape::kResumeClasses) * (2 * B + 2) * sizeof(uint32_t)));
            CAPE_ALLOC(dalloc(h->growState, B * cape::grow_state_bytes(h->cells)));
        }
        CAPE_ALLOC(hipHostMalloc(reinterpret_cast<void**>(&h->handedOverHost), 2 * sizeof(uint32_t)));
        h->handedOverHost[0] = h->handedOverHost[1] = 0;
        CAPE_ALLOC(hipEventCreateWithFlags(&h->handedOverReady, hipEventDisableTiming));
    }
    CAPE_ALLOC(dalloc(h->redoList, 2 * B + 2));
    CAPE_ALLOC(dalloc(h->spillList, 2 * B + 2));
    CAPE_ALLOC(hipMemset(h->spillList, 0, (2 * B + 2) * sizeof(uint32_t)));
    CAPE_ALLOC(dalloc(h->spillCounters, 2));
    CAPE_ALLOC(hipMemset(h->spillCounters, 0, 2 * sizeof(uint32_t)));
    {
        const char* stageA = std::getenv("CAPE_STAGE_A");
        if (cfg->max_batch <= kHostResultFrames && cfg->sub_batches <= 1 && !(stageA && std::string(stageA) == "bands"))
        {
            h->stripsAlways = stageA != nullptr; // (validated above: "strips")
            const char* pinnedInput = std::getenv("CAPE_PINNED_INPUT");
            h->pinnedByDma = pinnedInput && std::string(pinnedInput) == "dma";
            CAPE_ALLOC(dalloc(h->doneCounter, 1));
            CAPE_ALLOC(hipMemset(h->doneCounter, 0, sizeof(uint32_t)));
            CAPE_ALLOC(dalloc(h->stripCounters, (size_t)kHostResultFrames));
            CAPE_ALLOC(hipMemset(h->stripCounters, 0, kHostResultFrames * sizeof(uint32_t)));
        }
    }
    CAPE_ALLOC(hipEventCreateWithFlags(&h->workDone, hipEventDisableTiming));
    if ((cfg->flags & CAPE_FLAG_ASYNC_SECOND_PASS) && (cfg->flags & CAPE_FLAG_CYLINDERS) && cfg->max_batch > kHostResultFrames &&
        cfg->sub_batches <= 1)
    {
        CAPE_ALLOC(hipStreamCreateWithFlags(&h->sideStream, hipStreamNonBlocking));
        CAPE_ALLOC(hipEventCreateWithFlags(&h->sideFork, hipEventDisableTiming));
        CAPE_ALLOC(hipEventCreateWithFlags(&h->sideDone, hipEventDisableTiming));
    }
    CAPE_ALLOC(dalloc(h->debugCycles, B * cape::kProfileSlots));
    CAPE_ALLOC(hipMemset(h->debugCycles, 0, B * cape::kProfileSlots * 8));
    CAPE_ALLOC(dalloc(h->phaseTicks, B * 4));
    CAPE_ALLOC(hipMemset(h->phaseTicks, 0, B * 4 * 8));
    h->resultsOnHost = cfg->max_batch <= kHostResultFrames;
    if (h->resultsOnHost)
    {
        CAPE_ALLOC(hipHostMalloc(reinterpret_cast<void**>(&h->records), R * sizeof(cape_frame_record), hipHostMallocMapped | hipHostMallocCoherent));
        CAPE_ALLOC(hipHostMalloc(reinterpret_cast<void**>(&h->planeLabels), B * C * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
        CAPE_ALLOC(hipHostMalloc(reinterpret_cast<void**>(&h->cylLabels), B * C * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
        CAPE_ALLOC(hipHostMalloc(reinterpret_cast<void**>(&h->boundary), R * (size_t)h->boundaryCap * 3 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        CAPE_ALLOC(hipHostMalloc(reinterpret_cast<void**>(&h->doneFlag), 64, hipHostMallocMapped | hipHostMallocCoherent));
        *h->doneFlag = 0;
    }
    else
    {
        CAPE_ALLOC(dalloc(h->records, R));
        CAPE_ALLOC(dalloc(h->planeLabels, B * C));
        CAPE_ALLOC(dalloc(h->cylLabels, B * C));
        CAPE_ALLOC(dalloc(h->boundary, R * (size_t)h->boundaryCap * 3));
    }

    // ---- constant tables
    double k00, k02, k11, k12;
    inverse_intrinsics(cfg->fx, cfg->fy, cfg->cx, cfg->cy, k00, k02, k11, k12);
    std::vector<double> acol(cfg->width), brow(cfg->height);
    // transform_screen_to_camera: K^-1[:, :2] * [u v] + K^-1[:, 2] with k01 = k10 = 0
    for (int u = 0; u < cfg->width; ++u)
        acol[u] = (k00 * (double)u + 0.0) + k02;
    for (int v = 0; v < cfg->height; ++v)
        brow[v] = (0.0 + k11 * (double)v) + k12;
    auto ratios = [](const std::vector<double>& t, int ncell) {
        std::vector<float> r(ncell, 1.0f);
        for (int c = 0; c < ncell; ++c)
        {
            double mx = 0.0, mn = 0.0;
            for (int i = 0; i < CAPE_CELL_SIZE; ++i)
            {
                const double a = std::fabs(t[c * CAPE_CELL_SIZE + i]);
                if (a > 0.0)
                {
                    mx = (a > mx) ? a : mx;
                    mn = (mn == 0.0 || a < mn) ? a : mn;
                }
            }
            // rounded up so the device-side guard stays conservative
            r[c] = (mn > 0.0) ? std::nextafter((float)(mx / mn), INFINITY) : 1.0f;
        }
        return r;
    };
    const std::vector<float> rc = ratios(acol, h->hCells), rr = ratios(brow, h->vCells);
    const std::vector<double> rng = rng_table(0u, h->rngCount); // MAKE_DETERMINISTIC's seed; cape_set_rng_seed changes it
    {
        // _Xpre / _Ypre of Depth_Map_Transformation::init_matrices (depth_map_transformation.cpp:156-161)
        std::vector<float> xp(acol.begin(), acol.end()), yp(brow.begin(), brow.end());
        CAPE_ALLOC(dalloc(h->xpre, xp.size()));
        CAPE_ALLOC(dalloc(h->ypre, yp.size()));
        CAPE_ALLOC(hipMemcpy(h->xpre, xp.data(), xp.size() * sizeof(float), hipMemcpyHostToDevice));
        CAPE_ALLOC(hipMemcpy(h->ypre, yp.data(), yp.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    CAPE_ALLOC(hipMemcpy(h->acol, acol.data(), acol.size() * sizeof(double), hipMemcpyHostToDevice));
    CAPE_ALLOC(hipMemcpy(h->brow, brow.data(), brow.size() * sizeof(double), hipMemcpyHostToDevice));
    CAPE_ALLOC(hipMemcpy(h->ratioCol, rc.data(), rc.size() * sizeof(float), hipMemcpyHostToDevice));
    CAPE_ALLOC(hipMemcpy(h->ratioRow, rr.data(), rr.size() * sizeof(float), hipMemcpyHostToDevice));
    CAPE_ALLOC(hipMemcpy(h->rng, rng.data(), rng.size() * sizeof(double), hipMemcpyHostToDevice));
    if (h->resultsOnHost)
        std::memset(h->records, 0, R * sizeof(cape_frame_record));
    else
        CAPE_ALLOC(hipMemset(h->records, 0, R * sizeof(cape_frame_record)));

    // ---- kernel parameter blocks
    cape::StageAParams& a = h->pa;
    a.W = cfg->width;
    a.H = cfg->height;
    a.hCells = h->hCells;
    a.vCells = h->vCells;
    a.cells = h->cells;
    a.segsPerRow = (h->hCells + 31) / 32;
    a.bandsPerFrame = h->vCells * a.segsPerRow;
    a.pairsPerFrame = (a.bandsPerFrame + 1) / 2;
    a.acol = h->acol;
    a.brow = h->brow;
    a.ratio_col = h->ratioCol;
    a.ratio_row = h->ratioRow;
    a.cell_sums = h->cellSums;
    a.cell_plane = h->cellPlane;
    a.cell_score = h->cellScore;
    a.cell_tol = h->cellTol;
    a.cell_flags = h->cellFlags;
    a.cell_bins = h->cellBins;
    a.cell_aux = h->cellAux;
    a.cell_mse = h->cellMse;
    // primitive_detection.cpp:189-190 ; parameters.hpp:75 maximumPlaneAngleForMerge_d = 18.0f
    a.sinMerge = sinf(static_cast<float>(18.0f * M_PI / 180.0));
    a.cosMergeA = std::cos(static_cast<double>(18.0f) * M_PI / 180.0); // plane_segment.cpp:324
    a.smallBatchFrames = 0; // see cell_plane_threads(); CAPE_A2_WIDE_BATCH=n selects the one-tile instance for batches <= n
    if (const char* wide = std::getenv("CAPE_A2_WIDE_BATCH"))
        a.smallBatchFrames = std::atoi(wide);
    // plane_segment.hpp:33-34 ; parameters.hpp:72 minimumZeroDepthProportion = 0.7f
    a.minZeroPointCount = static_cast<int>(std::floor(static_cast<float>(400) * 0.7f));

    cape::StageBParams& b = h->pb;
    b.W = cfg->width;
    b.H = cfg->height;
    b.hCells = h->hCells;
    b.vCells = h->vCells;
    b.cells = h->cells;
    b.acol = h->acol;
    b.brow = h->brow;
    b.cell_sums = h->cellSums;
    b.cell_plane = h->cellPlane;
    b.cell_score = h->cellScore;
    b.cell_tol = h->cellTol;
    b.cell_flags = h->cellFlags;
    b.cell_bins = h->cellBins;
    b.cell_aux = h->cellAux;
    b.cell_mse = h->cellMse;
    b.records = h->records;
    b.plane_labels = h->planeLabels;
    b.cyl_labels = h->cylLabels;
    b.boundary = h->boundary;
    b.boundaryCapacity = h->boundaryCap;
    b.flags = cfg->flags;
    b.cosMerge = std::cos(static_cast<double>(18.0f) * M_PI / 180.0); // plane_segment.cpp:324
    b.planeSeedCount = static_cast<int>(static_cast<unsigned>((0.8 / 100.0) * h->cells));
    b.minCellActivated = static_cast<int>(static_cast<unsigned>((0.65 / 100.0) * h->cells));
    b.cylScratch = h->cylScratch;
    b.needCylinder = h->needCylinder;
    b.redoList = h->redoList;
    b.spillList = h->spillList;
    b.resumeList = h->resumeList;
    b.resumeBucketStride = h->resumeList ? (uint32_t)(2 * B + 2) : 0u;
    b.growState = h->growState;
    b.growStateStride = (uint32_t)cape::grow_state_bytes(h->cells);
    b.ldsLimitBytes = h->ldsLimit;
    {
        // Parked frames are finished by one WORKGROUP each on the wide grids (1280x960: 0.64 ms against 0.74 ms per 1 024 tunnel
        // frames) and by one WAVEFRONT each on grids up to 32 cells wide, where the lean lone-wave instance keeps four times as
        // many frames in flight and wins (640x480: 0.33 against 0.44 ms per 2 048 tunnel frames, 0.52 against 0.60 ms per 4 096
        // room frames; profiles/r03_cylinder_schedules.txt).  CAPE_RESUME=wave|group forces one (A/B runs, tests).
        const char* resumeEnv = std::getenv("CAPE_RESUME");
        const std::string forced = resumeEnv ? resumeEnv : "";
        b.resumeMode = (h->hCells > 32 && forced != "wave") || forced == "group" ? 2 : 1;
        if (!cape::resume_group_fits(b))
            b.resumeMode = 1;
    }
    b.twoPass = h->needCylinder ? 1 : 0;
    b.ldsLimitBytes = h->ldsLimit;
    if (!h->generalAll && cape::grow_lds_bytes(h->cells, (cfg->flags & CAPE_FLAG_CYLINDERS) != 0, CAPE_MAX_PLANES) > (size_t)h->ldsLimit)
        h->generalAll = true; // (a device with less LDS than the 64-segment instance wants: the general instance needs 20 KB)
    {
        // the general instance: a persistent grid of waves, one scratch slot each (two workgroups per CU hold its 64 KB of LDS; a handle
        // whose frames only reach it through the spill list gets by with fewer)
        const bool cyl = (cfg->flags & CAPE_FLAG_CYLINDERS) != 0;
        cape::GenParams& g = h->gen;
        g.slotBytes = cape::general_slot_bytes(h->cells, h->hCells, h->vCells, cyl, b.minCellActivated, &g.capSeg, &g.capCyl);
        g.ldsBytes = (int)cape::general_lds_bytes(h->cells, h->hCells, h->vCells, cyl, g.capSeg, g.capCyl, h->ldsLimit);
        if (g.ldsBytes <= 0)
        {
            free_all(h);
            delete h;
            return fail(CAPE_ERR_UNSUPPORTED, "this device offers too little LDS per workgroup for the grow kernels (" + std::to_string(ldsLimit) + " bytes)");
        }
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        size_t slots = h->generalAll ? (size_t)2 * cus : (size_t)std::min(cus, 64);
        if (const char* e = std::getenv("CAPE_GENERAL_SLOTS"))
            slots = (size_t)std::max(1, std::atoi(e));
        slots = std::min(slots, B);
        while (slots > 1 && slots * g.slotBytes > ((size_t)2 << 30)) // keep the scratch under 2 GB whatever the grid
            slots /= 2;
        g.scratchSlots = (int)slots;
        CAPE_ALLOC(hipMalloc(reinterpret_cast<void**>(&h->genScratch), slots * g.slotBytes));
        g.scratch = h->genScratch;
        g.spillAlloc = h->spillCounters;
        g.genFrames = h->spillCounters + 1;
        g.poolRecords = h->records + B;
        g.poolBoundary = h->boundary + B * (size_t)h->boundaryCap * 3;
        g.poolCapacity = h->spillRecords;
        g.poolBase = cfg->max_batch;
        g.rowWords = (h->hCells + 63) / 64;
        g.allFrames = h->generalAll ? 1 : 0;
    }
    if (h->needCylinder && !h->generalAll)
    {
        hipDeviceProp_t prop;
        const int perCu = cape::grow_waves_per_cu(h->pb);
        if (perCu > 0 && hipGetDeviceProperties(&prop, h->cfg.device) == hipSuccess)
            h->cylSlots = perCu * prop.multiProcessorCount;
    }
    if (const char* sched = std::getenv("CAPE_SCHEDULE"))
        h->forcedSchedule = std::string(sched) == "two" ? 1 : (std::string(sched) == "single" ? 2 : 0);
    b.debugCycles = h->debugCycles;
    b.phaseTicks = nullptr; // set per launch (launch_chain) while timing is on
    b.seed_sequence = h->seedSeq;
    b.rngTable = h->rng;
    b.rngCount = h->rngCount;
    // cylinder_segment.cpp:132
    b.ransacMaxIterations = static_cast<int>(static_cast<unsigned>(logf(1.0f - 0.8f) / logf(1.0f - powf(0.33f, 3.0f))));

    if (cfg->sub_batches > 1)
    {
        for (auto& st : h->pipeStream)
            CAPE_ALLOC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        CAPE_ALLOC(hipEventCreateWithFlags(&h->pipeFork, hipEventDisableTiming));
        for (auto& e : h->pipeJoin)
            CAPE_ALLOC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    *out = h;
    return CAPE_OK;
}

void cape_destroy(cape_handle h)
{
    if (!h)
        return;
    {
        DeviceGuard deviceGuard(h->cfg.device);
        (void)drain_handle(h); // nothing of the handle's may still be running on its buffers
        free_all(h);
    }
    delete h;
}

int cape_get_layout(cape_handle h, cape_layout* out)
{
    if (!h || !out)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    CAPE_ON_DEVICE(h);
    {
        hipDeviceProp_t prop;
        out->compute_units = (hipGetDeviceProperties(&prop, h->cfg.device) == hipSuccess) ? prop.multiProcessorCount : 0;
        out->grow_frames_per_cu = cape::grow_waves_per_cu(h->pb);
    }
    out->h_cells = h->hCells;
    out->v_cells = h->vCells;
    out->cells = h->cells;
    out->boundary_capacity = h->boundaryCap;
    out->frame_record_bytes = sizeof(cape_frame_record);
    out->effective_flags = h->cfg.flags & ~(uint32_t)(h->sideStream ? 0u : CAPE_FLAG_ASYNC_SECOND_PASS);
    out->reserved = 0;
    return CAPE_OK;
}

static int extract_impl(cape_handle h, const float* depth_dev, const uint16_t* depth_u16, float scale, int32_t n_frames,
                        void* stream_);

int cape_extract(cape_handle h, const float* depth_dev, int32_t n_frames, void* stream_)
{
    return extract_impl(h, depth_dev, nullptr, 0.0f, n_frames, stream_);
}

int cape_extract_u16(cape_handle h, const uint16_t* depth_dev, float scale, int32_t n_frames, void* stream_)
{
    if (!(scale > 0.0f))
        return fail(CAPE_ERR_INVALID_ARGUMENT, "scale must be positive");
    return extract_impl(h, nullptr, depth_dev, scale, n_frames, stream_);
}

static int extract_impl(cape_handle h, const float* depth_dev, const uint16_t* depth_u16, float scale, int32_t n_frames,
                        void* stream_)
{
    if (!h || (!depth_dev && !depth_u16) || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle/depth or negative frame count");
    if (n_frames > h->cfg.max_batch)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds max_batch");
    // the streaming kernel reads four pixels per lane with one vector load
    if ((depth_dev && reinterpret_cast<uintptr_t>(depth_dev) % 16 != 0) || (depth_u16 && reinterpret_cast<uintptr_t>(depth_u16) % 8 != 0))
        return fail(CAPE_ERR_INVALID_ARGUMENT, "depth must be aligned to four pixels (16 bytes of float32, 8 bytes of uint16)");
    h->lastFrames = n_frames;
    h->logPending = n_frames > 0; // cape_set_log_callback: this batch's records have not reached the host yet
    h->logDone = 0;
    h->polygonFrames = 0; // the polygons on the device belong to the previous batch
    h->matchExactFrames = 0; // and so do the polygon matches
    if (n_frames == 0)
        return CAPE_OK;
    CAPE_ON_DEVICE(h); // the handle's device, whatever the calling thread had current
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StreamScope streamScope(h, stream);
    if (streamScope.rc() != CAPE_OK)
        return streamScope.rc();
    h->pa.depth = depth_dev;
    h->pa.depth_u16 = depth_u16;
    h->pa.u16_scale = scale;
    h->doneArmed = false; // only launch_chain puts a signal behind the work; every other path drains the stream
    h->lazySpillArmed = false; // (a one-frame chain nobody read: its results are about to be overwritten)
    if (h->cfg.sub_batches > 1 && n_frames >= 2 * h->cfg.sub_batches)
    {
        // fork: both internal streams wait for everything already enqueued on the caller's stream
        CAPE_HIP_TRY(hipMemsetAsync(h->spillCounters, 0, 2 * sizeof(uint32_t), stream));
        CAPE_HIP_TRY(hipEventRecord(h->pipeFork, stream));
        for (auto& st : h->pipeStream)
            CAPE_HIP_TRY(hipStreamWaitEvent(st, h->pipeFork, 0));
        // stream 0 runs the streaming kernel of every sub-batch back to back; stream 1 runs the per-cell fit and the
        // grow kernel of sub-batch i as soon as its moments are done, i.e. underneath the moments of sub-batch i+1
        const int k = h->cfg.sub_batches;
        if ((int)h->pipeStage.size() < k)
        {
            const size_t old = h->pipeStage.size();
            h->pipeStage.resize(k, nullptr);
            for (size_t q = old; q < h->pipeStage.size(); ++q)
                CAPE_HIP_TRY(hipEventCreateWithFlags(&h->pipeStage[q], hipEventDisableTiming));
        }
        for (int i = 0; i < k; ++i)
        {
            const int f0 = (int)((long long)n_frames * i / k), f1 = (int)((long long)n_frames * (i + 1) / k);
            cape::StageAParams a;
            cape::StageBParams b;
            offset_params(h, f0, a, b);
            cape_handle_s::EvTriple* t = nullptr;
            const int rc = acquire_events(h, f1 - f0, &t);
            if (rc != CAPE_OK)
                return rc;
            if (t)
                CAPE_HIP_TRY(hipEventRecord(t->e[0], h->pipeStream[0]));
            CAPE_HIP_TRY(cape::launch_cell_moments(a, f1 - f0, h->pipeStream[0]));
            if (t)
                CAPE_HIP_TRY(hipEventRecord(t->e[1], h->pipeStream[0]));
            CAPE_HIP_TRY(hipEventRecord(h->pipeStage[i], h->pipeStream[0]));
            CAPE_HIP_TRY(hipStreamWaitEvent(h->pipeStream[1], h->pipeStage[i], 0));
            if (t)
                CAPE_HIP_TRY(hipEventRecord(t->e2b, h->pipeStream[1]));
            a.clear0 = b.redoList;
            a.clear1 = b.needCylinder;
            a.clear2 = b.resumeList;
            a.clear2Buckets = b.resumeList ? b.resumeBucketStride : 0u;
            a.clear3 = b.spillList;
            a.clear4 = nullptr; // (the pool's counters are shared by the sub-batches: cleared once, in front of the fork)
            CAPE_HIP_TRY(cape::launch_cell_plane(a, f1 - f0, h->pipeStream[1]));
            if (t)
                CAPE_HIP_TRY(hipEventRecord(t->e[2], h->pipeStream[1]));
            b.a2RowsPerTile = cape::cell_plane_rows_per_tile(a, f1 - f0);
            b.countersCleared = 1;
            b.phaseTicks = t ? h->phaseTicks : nullptr;
            CAPE_HIP_TRY(cape::launch_grow(b, f1 - f0, h->pipeStream[1], nullptr, nullptr, nullptr, &h->gen));
            if (t)
                CAPE_HIP_TRY(hipEventRecord(t->e[3], h->pipeStream[1]));
        }
        // join
        for (int i = 0; i < 2; ++i)
        {
            CAPE_HIP_TRY(hipEventRecord(h->pipeJoin[i], h->pipeStream[i]));
            CAPE_HIP_TRY(hipStreamWaitEvent(stream, h->pipeJoin[i], 0));
        }
        return CAPE_OK;
    }
    return launch_chain(h, h->pa, h->pb, n_frames, stream);
}

int cape_extract_host(cape_handle h, const float* depth_host, int32_t n_frames, void* stream_)
{
    if (!h || !depth_host || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle/depth or negative frame count");
    if (n_frames > h->cfg.max_batch)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds max_batch");
    CAPE_ON_DEVICE(h);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StreamScope streamScope(h, stream);
    if (streamScope.rc() != CAPE_OK)
        return streamScope.rc();
    // Pinned input (cape_host_alloc / cape_host_register, or any hipHostMalloc'ed / registered buffer): a few frames are
    // read by the streaming kernel straight from host memory -- the image is read exactly once, so the PCIe transfer IS the
    // kernel's input stream and no staging copy precedes it; larger batches take one DMA from the pinned pages.  Pageable
    // input goes through the runtime's staged copy.
    const size_t frameBytes = (size_t)h->cfg.width * h->cfg.height * sizeof(float);
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, depth_host) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer;
    if (!pinned)
        (void)hipGetLastError(); // an unregistered pointer is not an error here
    if (pinned && !h->pinnedByDma && n_frames <= kHostResultFrames && reinterpret_cast<uintptr_t>(attr.devicePointer) % 16 == 0)
    {
        h->inputOverLink = true;
        const int rc = cape_extract(h, static_cast<const float*>(attr.devicePointer), n_frames, stream_);
        h->inputOverLink = false;
        return rc;
    }
    const size_t bytes = (size_t)h->cfg.max_batch * frameBytes;
    if (!h->depthStage)
        CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->depthStage), bytes));
    CAPE_HIP_TRY(hipMemcpyAsync(h->depthStage, depth_host, (size_t)n_frames * frameBytes, hipMemcpyHostToDevice, stream));
    return cape_extract(h, h->depthStage, n_frames, stream_);
}

int cape_stream_create(cape_handle h, void** stream_out)
{
    if (!h || !stream_out)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    CAPE_ON_DEVICE(h);
    hipStream_t s = nullptr;
    CAPE_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = s;
    return CAPE_OK;
}

int cape_stream_destroy(cape_handle h, void* stream)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    if (!stream)
        return CAPE_OK;
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(drain_handle(h)); // (the handle's event may still refer to work on it)
    CAPE_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    CAPE_HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return CAPE_OK;
}

int cape_extract_u16_host(cape_handle h, const uint16_t* depth_host, float scale, int32_t n_frames, void* stream_)
{
    if (!h || !depth_host || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle/depth or negative frame count");
    if (n_frames > h->cfg.max_batch)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds max_batch");
    CAPE_ON_DEVICE(h);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StreamScope streamScope(h, stream);
    if (streamScope.rc() != CAPE_OK)
        return streamScope.rc();
    // like cape_extract_host: a few pinned frames are read in place, everything else is staged with one copy
    const size_t frameBytes = (size_t)h->cfg.width * h->cfg.height * sizeof(uint16_t);
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, depth_host) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer;
    if (!pinned)
        (void)hipGetLastError();
    if (pinned && !h->pinnedByDma && n_frames <= kHostResultFrames && reinterpret_cast<uintptr_t>(attr.devicePointer) % 8 == 0)
    {
        h->inputOverLink = true;
        const int rc = cape_extract_u16(h, static_cast<const uint16_t*>(attr.devicePointer), scale, n_frames, stream_);
        h->inputOverLink = false;
        return rc;
    }
    if (!h->depthStage) // (sized for float32 frames: cape_extract_host shares it)
        CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->depthStage), (size_t)h->cfg.max_batch * h->cfg.width * h->cfg.height * sizeof(float)));
    CAPE_HIP_TRY(hipMemcpyAsync(h->depthStage, depth_host, (size_t)n_frames * frameBytes, hipMemcpyHostToDevice, stream));
    return cape_extract_u16(h, reinterpret_cast<const uint16_t*>(h->depthStage), scale, n_frames, stream_);
}

int cape_device_results(cape_handle h, void** records, int32_t** plane_labels, int32_t** cyl_labels, double** boundary)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    if (records)
        *records = h->records;
    if (plane_labels)
        *plane_labels = h->planeLabels;
    if (cyl_labels)
        *cyl_labels = h->cylLabels;
    if (boundary)
        *boundary = h->boundary;
    return CAPE_OK;
}

int cape_sync_results(cape_handle h, void* stream_)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    CAPE_ON_DEVICE(h);
    StreamScope streamScope(h, static_cast<hipStream_t>(stream_)); // enter_stream does the waiting
    return streamScope.rc();
}

int cape_copy_results(cape_handle h, int32_t n_frames, cape_frame_record* records, int32_t* plane_labels,
                      int32_t* cyl_labels, double* boundary)
{
    if (!h || n_frames < 0 || n_frames > h->cfg.max_batch)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad handle / frame count");
    CAPE_ON_DEVICE(h);
    const size_t n = (size_t)n_frames, C = (size_t)h->cells;
    if (h->resultsOnHost)
    {
        // the kernels wrote into pinned host memory: once the chain's signal word has arrived the data is simply there
        if (const int rc = wait_results(h); rc != CAPE_OK)
            return rc;
        if (records)
            std::memcpy(records, h->records, n * sizeof(cape_frame_record));
        log_batch(h, records ? h->records : nullptr, n_frames);
        if (plane_labels)
            std::memcpy(plane_labels, h->planeLabels, n * C * sizeof(int32_t));
        if (cyl_labels)
            std::memcpy(cyl_labels, h->cylLabels, n * C * sizeof(int32_t));
        if (boundary)
            std::memcpy(boundary, h->boundary, n * (size_t)h->boundaryCap * 3 * sizeof(double));
        return CAPE_OK;
    }
    // behind THIS handle's work only (its event, its side stream): several handles driven from several host threads -- the overlay's
    // shards -- must not wait for each other's kernels here (through round 5 this was a hipDeviceSynchronize)
    if (h->workRecorded || h->sidePending)
        CAPE_HIP_TRY(drain_handle(h));
    else
        CAPE_HIP_TRY(hipDeviceSynchronize());
    if (records)
        CAPE_HIP_TRY(hipMemcpy(records, h->records, n * sizeof(cape_frame_record), hipMemcpyDeviceToHost));
    log_batch(h, records, n_frames);
    if (plane_labels)
        CAPE_HIP_TRY(hipMemcpy(plane_labels, h->planeLabels, n * C * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (cyl_labels)
        CAPE_HIP_TRY(hipMemcpy(cyl_labels, h->cylLabels, n * C * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (boundary)
        CAPE_HIP_TRY(hipMemcpy(boundary, h->boundary, n * (size_t)h->boundaryCap * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return CAPE_OK;
}

int cape_host_results(cape_handle h, const cape_frame_record** records, const int32_t** plane_labels, const int32_t** cyl_labels,
                      const double** boundary)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    if (!h->resultsOnHost)
        return fail(CAPE_ERR_UNSUPPORTED, "results live in device memory for this handle (max_batch > 8): use cape_copy_results");
    CAPE_ON_DEVICE(h);
    if (const int rc = wait_results(h); rc != CAPE_OK)
        return rc;
    log_batch(h, h->records, h->lastFrames);
    if (records)
        *records = h->records;
    if (plane_labels)
        *plane_labels = h->planeLabels;
    if (cyl_labels)
        *cyl_labels = h->cylLabels;
    if (boundary)
        *boundary = h->boundary;
    return CAPE_OK;
}

int cape_host_alloc(cape_handle h, uint64_t bytes, void** out)
{
    if (!h || !out || bytes == 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument or zero size");
    CAPE_ON_DEVICE(h);
    *out = nullptr;
    CAPE_HIP_TRY(hipHostMalloc(out, (size_t)bytes, hipHostMallocMapped));
    return CAPE_OK;
}

int cape_host_free(cape_handle h, void* p)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    if (!p)
        return CAPE_OK;
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(drain_handle(h)); // a kernel of this handle may still be reading the buffer
    CAPE_HIP_TRY(hipHostFree(p));
    return CAPE_OK;
}

int cape_host_register(cape_handle h, void* p, uint64_t bytes)
{
    if (!h || !p || bytes == 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument or zero size");
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(hipHostRegister(p, (size_t)bytes, hipHostRegisterMapped));
    return CAPE_OK;
}

int cape_host_unregister(cape_handle h, void* p)
{
    if (!h || !p)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(drain_handle(h));
    CAPE_HIP_TRY(hipHostUnregister(p));
    return CAPE_OK;
}

int cape_copy_cell_stats(cape_handle h, int32_t frame, cape_cell_stats* out)
{
    if (!h || !out || frame < 0 || frame >= h->cfg.max_batch)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad handle / frame");
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(hipDeviceSynchronize());
    const size_t C = (size_t)h->cells, off = (size_t)frame * C;
    std::vector<double> sums(C * cape::kSumStride), plane(C * cape::kPlaneStride), score(C);
    std::vector<float> tol(C);
    std::vector<uint32_t> flags(C);
    std::vector<int32_t> bins(C);
    CAPE_HIP_TRY(hipMemcpy(sums.data(), h->cellSums + off * cape::kSumStride, sums.size() * 8, hipMemcpyDeviceToHost));
    CAPE_HIP_TRY(hipMemcpy(plane.data(), h->cellPlane + off * cape::kPlaneStride, plane.size() * 8, hipMemcpyDeviceToHost));
    CAPE_HIP_TRY(hipMemcpy(score.data(), h->cellScore + off, C * 8, hipMemcpyDeviceToHost));
    CAPE_HIP_TRY(hipMemcpy(tol.data(), h->cellTol + off, C * 4, hipMemcpyDeviceToHost));
    CAPE_HIP_TRY(hipMemcpy(flags.data(), h->cellFlags + off, C * 4, hipMemcpyDeviceToHost));
    CAPE_HIP_TRY(hipMemcpy(bins.data(), h->cellBins + off, C * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < C; ++i)
    {
        cape_cell_stats& o = out[i];
        for (int k = 0; k < 9; ++k)
            o.sums[k] = sums[i * cape::kSumStride + k];
        const double* p = &plane[i * cape::kPlaneStride];
        o.normal[0] = p[0]; o.normal[1] = p[1]; o.normal[2] = p[2];
        o.d = p[3];
        o.centroid[0] = p[4]; o.centroid[1] = p[5]; o.centroid[2] = p[6];
        o.mse = p[7];
        o.score = score[i];
        o.tol = tol[i];
        o.point_count = flags[i] & cape::kCountMask;
        o.bin = bins[i];
        o.planar = (flags[i] & cape::kFlagPlanar) ? 1u : 0u;
        o.inorder = (flags[i] & cape::kFlagInorder) ? 1u : 0u;
        o.pad = 0;
    }
    return CAPE_OK;
}

int cape_copy_seed_sequence(cape_handle h, int32_t frame, int32_t* seeds_out, int32_t capacity, int32_t* n_out)
{
    if (!h || !n_out || frame < 0 || frame >= h->cfg.max_batch || capacity < 0 || (capacity > 0 && !seeds_out))
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad handle / frame / buffer");
    CAPE_ON_DEVICE(h);
    CAPE_SETTLE_RESULTS(h);
    CAPE_HIP_TRY(hipDeviceSynchronize());
    cape_frame_header hdr;
    CAPE_HIP_TRY(hipMemcpy(&hdr, &h->records[frame].header, sizeof(hdr), hipMemcpyDeviceToHost));
    *n_out = hdr.n_seeds;
    int n = hdr.n_seeds < h->cells ? hdr.n_seeds : h->cells; // the buffer keeps one entry per cell
    n = n < capacity ? n : capacity;
    std::vector<uint16_t> tmp((size_t)(n > 0 ? n : 0));
    if (n > 0)
        CAPE_HIP_TRY(hipMemcpy(tmp.data(), h->seedSeq + (size_t)frame * h->cells, (size_t)n * sizeof(uint16_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i)
        seeds_out[i] = (int32_t)tmp[(size_t)i];
    return CAPE_OK;
}

int cape_rectify_depth(cape_handle h, const float* depth_dev, float* rectified_dev, int32_t n_frames,
                       const double* cam2_to_cam1, void* stream_)
{
    if (!h || !depth_dev || !rectified_dev || !cam2_to_cam1 || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument or negative frame count");
    if (n_frames == 0)
        return CAPE_OK;
    if (depth_dev == rectified_dev)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "rectify_depth is not in-place");
    CAPE_ON_DEVICE(h);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StreamScope streamScope(h, stream);
    if (streamScope.rc() != CAPE_OK)
        return streamScope.rc();
    if (h->rectFlagFrames < (size_t)n_frames)
    {
        CAPE_HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(h->rectFlags);
        h->rectFlags = nullptr;
        h->rectFlagFrames = 0;
        CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->rectFlags), (2 * (size_t)n_frames + 1) * sizeof(unsigned)));
        h->rectFlagFrames = (size_t)n_frames;
    }
    if (h->computeUnits <= 0)
    {
        hipDeviceProp_t prop;
        h->computeUnits = hipGetDeviceProperties(&prop, h->cfg.device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    cape::RectifyParams p;
    p.in = depth_dev;
    p.out = rectified_dev;
    p.ldsLimitBytes = h->ldsLimit;
    p.frameFlag = h->rectFlags;
    p.flagged = h->rectFlags + h->rectFlagFrames;
    {
        const char* eb = std::getenv("CAPE_RECTIFY_BAND");
        p.bandRows = eb ? std::atoi(eb) : 0;
    }
    p.W = h->cfg.width;
    p.H = h->cfg.height;
    p.xpre = h->xpre;
    p.ypre = h->ypre;
    for (int i = 0; i < 12; ++i)
        p.T[i] = cam2_to_cam1[i];
    p.fx = h->cfg.fx;
    p.fy = h->cfg.fy;
    p.cx = h->cfg.cx;
    p.cy = h->cfg.cy;
    {
        // Which source rows can land in a band of target rows?  The row displacement of this rig, sampled over the image and over
        // depths from 0.3 m to 10 m (plain doubles: a prediction, the kernel checks every pixel and flags what escapes it).
        double lo = 0.0, hi = 0.0;
        bool any = false;
        const double zs[] = {300.0, 600.0, 1200.0, 2500.0, 5000.0, 10000.0};
        for (int ry = 0; ry <= 4; ++ry)
            for (int rx = 0; rx <= 4; ++rx)
                for (double z : zs)
                {
                    const double row = (h->cfg.height - 1) * ry / 4.0, col = (h->cfg.width - 1) * rx / 4.0;
                    const double x = (col - h->cfg.cx) / h->cfg.fx * z, y = (row - h->cfg.cy) / h->cfg.fy * z;
                    const double q1 = p.T[4] * x + p.T[5] * y + p.T[6] * z + p.T[7], q2 = p.T[8] * x + p.T[9] * y + p.T[10] * z + p.T[11];
                    if (!(q2 > 0))
                        continue;
                    const double d = (h->cfg.fy * q1 / q2 + h->cfg.cy) - row;
                    lo = any ? std::min(lo, d) : d;
                    hi = any ? std::max(hi, d) : d;
                    any = true;
                }
        const char* em = std::getenv("CAPE_RECTIFY_MARGIN");
        const int margin = em ? std::atoi(em) : 2;
        const double cap = 4.0 * h->cfg.height; // (a degenerate rig: everything escapes, the general kernels take over)
        p.shiftLo = (int)std::floor(std::max(-cap, std::min(cap, lo))) - margin;
        p.shiftHi = (int)std::ceil(std::max(-cap, std::min(cap, hi))) + margin;
    }
    CAPE_HIP_TRY(cape::launch_rectify(p, n_frames, h->computeUnits, stream));
    return CAPE_OK;
}

int cape_rectify_depth_host(cape_handle h, const float* depth_host, float* rectified_host, int32_t n_frames,
                            const double* cam2_to_cam1)
{
    if (!h || !depth_host || !rectified_host || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument or negative frame count");
    CAPE_ON_DEVICE(h);
    const size_t bytes = (size_t)n_frames * h->cfg.width * h->cfg.height * sizeof(float);
    float *din = nullptr, *dout = nullptr;
    CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&din), bytes));
    if (hipMalloc(reinterpret_cast<void**>(&dout), bytes) != hipSuccess)
    {
        (void)hipFree(din);
        return fail(CAPE_ERR_HIP, "hipMalloc failed");
    }
    int rc = CAPE_OK;
    if (hipMemcpy(din, depth_host, bytes, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(CAPE_ERR_HIP, "H2D copy failed");
    if (rc == CAPE_OK)
        rc = cape_rectify_depth(h, din, dout, n_frames, cam2_to_cam1, nullptr);
    if (rc == CAPE_OK && hipStreamSynchronize(nullptr) != hipSuccess)
        rc = fail(CAPE_ERR_HIP, "rectify kernels failed");
    if (rc == CAPE_OK && hipMemcpy(rectified_host, dout, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(CAPE_ERR_HIP, "D2H copy failed");
    (void)hipFree(din);
    (void)hipFree(dout);
    return rc;
}

int cape_match_consecutive(cape_handle h, int32_t n_frames, uint32_t flags, void* stream_)
{
    if (!h || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle or negative frame count");
    if (n_frames > h->lastFrames)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds the last cape_extract batch");
    if (flags & ~(uint32_t)(CAPE_MATCH_ADVANCED | CAPE_MATCH_ALLOW_INDEX0))
        return fail(CAPE_ERR_INVALID_ARGUMENT, "unknown match flag");
    if (n_frames == 0)
        return CAPE_OK;
    CAPE_ON_DEVICE(h);
    CAPE_SETTLE_RESULTS(h);
    StreamScope streamScope(h, static_cast<hipStream_t>(stream_));
    if (streamScope.rc() != CAPE_OK)
        return streamScope.rc();
    if (!h->matches)
        CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->matches), (size_t)h->cfg.max_batch * sizeof(cape_frame_match)));
    cape::MatchParams p;
    p.records = h->records;
    p.plane_labels = h->planeLabels;
    p.matches = h->matches;
    p.cells = h->cells;
    p.flags = flags;
    // parameters::matching (src/parameters.hpp:89-95), evaluated on the host like the reference's function-local statics
    p.minCosAngle = std::abs(std::cos(20.0 * M_PI / 180.0));
    p.maxDistance = 100.0;
    const double planeMinimalOverlap = static_cast<double>(0.4f);
    p.minOverlap = (flags & CAPE_MATCH_ADVANCED) ? planeMinimalOverlap / 2 : planeMinimalOverlap;
    CAPE_HIP_TRY(cape::launch_match(p, n_frames, static_cast<hipStream_t>(stream_)));
    return CAPE_OK;
}

int cape_device_matches(cape_handle h, void** matches)
{
    if (!h || !matches)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    if (!h->matches)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "cape_match_consecutive has not run");
    *matches = h->matches;
    return CAPE_OK;
}

int cape_copy_matches(cape_handle h, int32_t n_frames, cape_frame_match* out)
{
    if (!h || !out || n_frames < 0 || n_frames > h->cfg.max_batch)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad argument");
    if (!h->matches)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "cape_match_consecutive has not run");
    CAPE_ON_DEVICE(h);
    if (h->workRecorded || h->sidePending)
        CAPE_HIP_TRY(drain_handle(h));
    else
        CAPE_HIP_TRY(hipDeviceSynchronize());
    CAPE_HIP_TRY(hipMemcpy(out, h->matches, (size_t)n_frames * sizeof(cape_frame_match), hipMemcpyDeviceToHost));
    return CAPE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// multi-GPU gather of the packed primitive lists (see cape_gather.hip)
// ---------------------------------------------------------------------------------------------------------------
namespace {
size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

void fill_layout(const cape_handle_s* h, const cape_gather_config& c, cape_gather_layout& L)
{
    L = cape_gather_layout{};
    L.frames_capacity = c.frames_capacity;
    L.planes_capacity = c.frames_capacity * c.planes_per_frame;
    L.cylinders_capacity = c.frames_capacity * c.cylinders_per_frame;
    L.cells = h->cells;
    size_t off = align16(sizeof(cape_packed_header));
    L.frames_offset = off;
    off = align16(off + (size_t)L.frames_capacity * sizeof(cape_packed_frame));
    L.planes_offset = off;
    off = align16(off + (size_t)L.planes_capacity * sizeof(cape_packed_plane));
    L.cylinders_offset = off;
    off = align16(off + (size_t)L.cylinders_capacity * sizeof(cape_packed_cylinder));
    if (c.flags & CAPE_GATHER_LABELS)
    {
        L.plane_labels_offset = off;
        off = align16(off + (size_t)L.frames_capacity * h->cells);
        L.cyl_labels_offset = off;
        off = align16(off + (size_t)L.frames_capacity * h->cells);
    }
    L.bytes_per_rank = off;
}

// default capacities the first time a pack / gather is asked for without cape_gather_configure
int ensure_gather_configured(cape_handle_s* h)
{
    if (h->packed[0])
        return CAPE_OK;
    cape_gather_config c{};
    c.frames_capacity = h->cfg.max_batch;
    return cape_gather_configure(h, &c, nullptr);
}

int pack_into_next_slot(cape_handle_s* h, int n_frames, int first_frame, hipStream_t stream)
{
    const int slot = h->packSlot ^ 1;
    // the slot may still be read by the all-gather of two batches ago
    if (h->packedBusy[slot])
    {
        CAPE_HIP_TRY(hipStreamWaitEvent(stream, h->packedFree[slot], 0));
        h->packedBusy[slot] = false;
    }
    const cape_gather_layout& L = h->gatherLayout;
    unsigned char* base = h->packed[slot];
    cape::PackParams p{};
    p.records = h->records;
    p.recordsBase = h->records;
    p.poolBase = h->cfg.max_batch;
    p.planeLabelsIn = h->planeLabels;
    p.cylLabelsIn = h->cylLabels;
    p.header = reinterpret_cast<cape_packed_header*>(base);
    p.frames = reinterpret_cast<cape_packed_frame*>(base + L.frames_offset);
    p.planes = reinterpret_cast<cape_packed_plane*>(base + L.planes_offset);
    p.cylinders = reinterpret_cast<cape_packed_cylinder*>(base + L.cylinders_offset);
    p.planeLabels8 = L.plane_labels_offset ? base + L.plane_labels_offset : nullptr;
    p.cylLabels8 = L.cyl_labels_offset ? base + L.cyl_labels_offset : nullptr;
    p.nFrames = n_frames;
    p.firstFrame = first_frame;
    p.framesCapacity = L.frames_capacity;
    p.planesCapacity = L.planes_capacity;
    p.cylindersCapacity = L.cylinders_capacity;
    p.cells = h->cells;
    p.flags = h->gatherCfg.flags;
    CAPE_HIP_TRY(cape::launch_pack(p, stream));
    h->packSlot = slot;
    return CAPE_OK;
}
} // namespace

int cape_gather_configure(cape_handle h, const cape_gather_config* cfg, cape_gather_layout* layout_out)
{
    if (!h || !cfg)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    cape_gather_config c = *cfg;
    if (c.planes_per_frame == 0)
        c.planes_per_frame = 16;
    if (c.cylinders_per_frame == 0)
        c.cylinders_per_frame = 8;
    if (c.frames_capacity <= 0 || c.frames_capacity > h->cfg.max_batch || c.planes_per_frame < 0 ||
        c.planes_per_frame > 4096 || c.cylinders_per_frame < 0 || c.cylinders_per_frame > 4096 ||
        (c.flags & ~(uint32_t)CAPE_GATHER_LABELS))
        return fail(CAPE_ERR_INVALID_ARGUMENT, "frames_capacity in [1, max_batch], planes/cylinders per frame in [1, 4096], known flags");
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(hipDeviceSynchronize()); // nothing may still read the old slots
    cape_gather_layout L;
    fill_layout(h, c, L);
    for (int k = 0; k < 2; ++k)
    {
        (void)hipFree(h->packed[k]);
        h->packed[k] = nullptr;
        h->packedBusy[k] = false;
    }
    h->gatherPending = false;
    for (int k = 0; k < 2; ++k)
    {
        CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->packed[k]), L.bytes_per_rank));
        CAPE_HIP_TRY(hipMemset(h->packed[k], 0, L.bytes_per_rank));
        if (!h->packedFree[k])
            CAPE_HIP_TRY(hipEventCreateWithFlags(&h->packedFree[k], hipEventDisableTiming));
    }
    if (!h->packReady)
        CAPE_HIP_TRY(hipEventCreateWithFlags(&h->packReady, hipEventDisableTiming));
    if (!h->gatherDone)
        CAPE_HIP_TRY(hipEventCreateWithFlags(&h->gatherDone, hipEventDisableTiming));
    h->gatherCfg = c;
    h->gatherLayout = L;
    if (layout_out)
        *layout_out = L;
    return CAPE_OK;
}

int cape_pack_primitives(cape_handle h, int32_t n_frames, int32_t first_frame, void** packed_dev, void* stream_)
{
    if (!h || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle or negative frame count");
    if (n_frames > h->lastFrames)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds the last cape_extract batch");
    CAPE_ON_DEVICE(h);
    CAPE_SETTLE_RESULTS(h);
    if (const int rc = ensure_gather_configured(h); rc != CAPE_OK)
        return rc;
    if (n_frames > h->gatherLayout.frames_capacity)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds cape_gather_config.frames_capacity");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StreamScope streamScope(h, stream);
    if (streamScope.rc() != CAPE_OK)
        return streamScope.rc();
    if (const int rc = pack_into_next_slot(h, n_frames, first_frame, stream); rc != CAPE_OK)
        return rc;
    if (packed_dev)
        *packed_dev = h->packed[h->packSlot];
    return CAPE_OK;
}

int cape_copy_packed(cape_handle h, void* packed_host)
{
    if (!h || !packed_host)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    if (!h->packed[0])
        return fail(CAPE_ERR_INVALID_ARGUMENT, "nothing has been packed yet");
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(hipDeviceSynchronize());
    CAPE_HIP_TRY(hipMemcpy(packed_host, h->packed[h->packSlot], h->gatherLayout.bytes_per_rank, hipMemcpyDeviceToHost));
    return CAPE_OK;
}

int cape_comm_unique_id(void* id_out)
{
    if (!id_out)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    if (const char* why = cape::rccl_load())
        return fail(CAPE_ERR_UNSUPPORTED, why);
    cape::RcclUniqueId id;
    if (const int rc = cape::rccl_unique_id(&id); rc != 0)
        return fail(CAPE_ERR_HIP, std::string("ncclGetUniqueId: ") + cape::rccl_error_string(rc));
    std::memcpy(id_out, id.internal, CAPE_COMM_ID_BYTES);
    return CAPE_OK;
}

int cape_comm_init(cape_handle h, const void* id_, int32_t rank, int32_t world)
{
    if (!h || !id_ || world <= 0 || rank < 0 || rank >= world)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument or rank outside [0, world)");
    if (h->comm)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "the handle already has a communicator (cape_comm_destroy first)");
    if (const char* why = cape::rccl_load())
        return fail(CAPE_ERR_UNSUPPORTED, why);
    CAPE_ON_DEVICE(h);
    cape::RcclUniqueId id;
    std::memcpy(id.internal, id_, CAPE_COMM_ID_BYTES);
    if (!h->commStream)
        CAPE_HIP_TRY(hipStreamCreateWithFlags(&h->commStream, hipStreamNonBlocking));
    if (const int rc = cape::rccl_comm_init(&h->comm, world, id, rank); rc != 0)
    {
        h->comm = nullptr;
        return fail(CAPE_ERR_HIP, std::string("ncclCommInitRank: ") + cape::rccl_error_string(rc));
    }
    h->commRank = rank;
    h->commWorld = world;
    return CAPE_OK;
}

int cape_comm_info(cape_handle h, cape_comm_info_t* out)
{
    if (!h || !out)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    std::memset(out, 0, sizeof(*out));
    out->nranks = out->rank = out->device = -1;
    out->handle_device = h->cfg.device;
    if (!h->comm)
        return CAPE_OK; // no communicator: has_comm = 0
    out->has_comm = 1;
    out->has_gather = cape::rccl_has_gather() ? 1 : 0;
    CAPE_ON_DEVICE(h);
    int count = -1, rank = -1, device = -1;
    cape::rccl_comm_query(h->comm, &count, &rank, &device);
    out->nranks = count;
    out->rank = rank;
    out->device = device;
    out->init_nranks = h->commWorld;
    out->init_rank = h->commRank;
    return CAPE_OK;
}

int cape_comm_destroy(cape_handle h)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    if (!h->comm)
        return CAPE_OK;
    CAPE_ON_DEVICE(h);
    if (h->commStream)
        CAPE_HIP_TRY(hipStreamSynchronize(h->commStream));
    const int rc = cape::rccl_comm_destroy(h->comm);
    h->comm = nullptr;
    h->gatherPending = false;
    if (rc != 0)
        return fail(CAPE_ERR_HIP, std::string("ncclCommDestroy: ") + cape::rccl_error_string(rc));
    return CAPE_OK;
}

// root < 0: ncclAllGather (every rank receives); root >= 0: ncclGather to that rank (recv_dev is read on the root only)
static int gather_impl(cape_handle h, int32_t n_frames, int32_t first_frame, int32_t root, void* recv_dev, void* stream_)
{
    if (!h || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle or negative frame count");
    if (!h->comm)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "no communicator: call cape_comm_init first");
    if (root >= h->commWorld)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "root outside [0, world)");
    if (!recv_dev && (root < 0 || root == h->commRank))
        return fail(CAPE_ERR_INVALID_ARGUMENT, "recv_dev is null on a receiving rank");
    if (root >= 0 && !cape::rccl_has_gather())
        return fail(CAPE_ERR_UNSUPPORTED, "this librccl.so has no ncclGather: use cape_gather_primitives");
    if (n_frames > h->lastFrames)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds the last cape_extract batch");
    CAPE_ON_DEVICE(h);
    CAPE_SETTLE_RESULTS(h);
    if (const int rc = ensure_gather_configured(h); rc != CAPE_OK)
        return rc;
    if (n_frames > h->gatherLayout.frames_capacity)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds cape_gather_config.frames_capacity");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StreamScope streamScope(h, stream);
    if (streamScope.rc() != CAPE_OK)
        return streamScope.rc();
    if (const int rc = pack_into_next_slot(h, n_frames, first_frame, stream); rc != CAPE_OK)
        return rc;
    const int slot = h->packSlot;
    // the collective runs on the handle's own stream, behind the pack kernels: the caller's stream is free for the
    // kernels of the next batch
    CAPE_HIP_TRY(hipEventRecord(h->packReady, stream));
    CAPE_HIP_TRY(hipStreamWaitEvent(h->commStream, h->packReady, 0));
    if (root < 0)
    {
        if (const int rc = cape::rccl_all_gather_bytes(h->packed[slot], recv_dev, h->gatherLayout.bytes_per_rank, h->comm, h->commStream);
            rc != 0)
            return fail(CAPE_ERR_HIP, std::string("ncclAllGather: ") + cape::rccl_error_string(rc));
    }
    else if (const int rc = cape::rccl_gather_bytes(h->packed[slot], recv_dev, h->gatherLayout.bytes_per_rank, root, h->comm, h->commStream);
             rc != 0)
        return fail(CAPE_ERR_HIP, std::string("ncclGather: ") + cape::rccl_error_string(rc));
    CAPE_HIP_TRY(hipEventRecord(h->packedFree[slot], h->commStream));
    h->packedBusy[slot] = true;
    CAPE_HIP_TRY(hipEventRecord(h->gatherDone, h->commStream));
    h->gatherPending = true;
    return CAPE_OK;
}

int cape_gather_primitives(cape_handle h, int32_t n_frames, int32_t first_frame, void* recv_dev, void* stream_)
{
    return gather_impl(h, n_frames, first_frame, -1, recv_dev, stream_);
}

int cape_gather_primitives_root(cape_handle h, int32_t n_frames, int32_t first_frame, int32_t root, void* recv_dev, void* stream_)
{
    if (root < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "root outside [0, world)");
    return gather_impl(h, n_frames, first_frame, root, recv_dev, stream_);
}

int cape_count_primitives(cape_handle h, int32_t n_frames, int32_t* n_planes, int32_t* n_cylinders, int32_t* max_planes_per_frame)
{
    if (!h || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle or negative frame count");
    if (n_frames > h->lastFrames)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds the last cape_extract batch");
    CAPE_ON_DEVICE(h);
    CAPE_SETTLE_RESULTS(h);
    int32_t tot[4] = {0, 0, 0, 0};
    if (n_frames > 0)
    {
        if (!h->countScratch)
            CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->countScratch), 4 * sizeof(int32_t)));
        // behind the batch, wherever it was enqueued (the caller's stream AND the handle's side stream of an asynchronous
        // second pass), without touching those streams
        if (h->workRecorded || h->sidePending)
            CAPE_HIP_TRY(drain_handle(h));
        else
            CAPE_HIP_TRY(hipDeviceSynchronize());
        hipStream_t st = nullptr;
        CAPE_HIP_TRY(cape::launch_count_primitives(h->records, n_frames, h->countScratch, st));
        CAPE_HIP_TRY(hipMemcpyAsync(tot, h->countScratch, sizeof(tot), hipMemcpyDeviceToHost, st));
        CAPE_HIP_TRY(hipStreamSynchronize(st));
    }
    if (n_planes)
        *n_planes = tot[0];
    if (n_cylinders)
        *n_cylinders = tot[1];
    if (max_planes_per_frame)
        *max_planes_per_frame = tot[2];
    return CAPE_OK;
}

int cape_gather_wait(cape_handle h, void* stream_, int32_t host_sync)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    if (!h->gatherPending)
        return CAPE_OK;
    CAPE_ON_DEVICE(h);
    if (host_sync)
    {
        CAPE_HIP_TRY(hipEventSynchronize(h->gatherDone));
        h->gatherPending = false;
    }
    else
    {
        CAPE_HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream_), h->gatherDone, 0));
    }
    return CAPE_OK;
}

int cape_build_polygons(cape_handle h, int32_t n_frames, void* stream_)
{
    if (!h || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle or negative frame count");
    if (n_frames > h->lastFrames)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds the last cape_extract batch");
    if (n_frames == 0)
        return CAPE_OK;
    CAPE_ON_DEVICE(h);
    CAPE_SETTLE_RESULTS(h);
    const size_t B = (size_t)h->cfg.max_batch + (size_t)h->spillRecords; // a polygon row / vertex slab per record, spill pool included
    if (!h->polygons)
    {
        if (h->resultsOnHost)
        {
            // the records and boundary points of a few-frame handle live in pinned host memory (the kernel reads them over
            // PCIe); the polygons follow them there
            CAPE_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->polygons), B * CAPE_MAX_PLANES * sizeof(cape_polygon),
                                       hipHostMallocMapped | hipHostMallocCoherent));
            CAPE_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->polyVertices), B * (size_t)h->boundaryCap * 2 * sizeof(double),
                                       hipHostMallocMapped | hipHostMallocCoherent));
        }
        else
        {
            CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->polygons), B * CAPE_MAX_PLANES * sizeof(cape_polygon)));
            CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->polyVertices), B * (size_t)h->boundaryCap * 2 * sizeof(double)));
        }
    }
    if (!h->polyLadder)
        CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->polyLadder), cape::polygon_scratch_bytes(B, h->boundaryCap)));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StreamScope streamScope(h, stream);
    if (streamScope.rc() != CAPE_OK)
        return streamScope.rc();
    cape::PolygonParams p;
    cape::polygon_bind_scratch(p, h->polyLadder, B, h->boundaryCap);
    if (h->computeUnits <= 0)
    {
        hipDeviceProp_t prop;
        h->computeUnits = hipGetDeviceProperties(&prop, h->cfg.device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    p.computeUnits = h->computeUnits;
    p.originInCentroid = 0;
    p.records = h->records;
    p.boundary = h->boundary;
    p.polygons = h->polygons;
    p.vertices = reinterpret_cast<double2*>(h->polyVertices);
    p.boundaryCapacity = h->boundaryCap;
    p.prof = h->debugCycles;
    p.poolBase = h->cfg.max_batch;
    p.poolCapacity = h->spillRecords;
    p.poolUsed = h->spillCounters;
#ifdef CAPE_POLY_PROFILE
    CAPE_HIP_TRY(hipMemsetAsync(h->debugCycles, 0, (size_t)n_frames * cape::kProfileSlots * 8, stream));
    CAPE_HIP_TRY(hipMemsetAsync(h->debugCycles + 6, 0xFF, 2 * 8, stream)); // the two minima of the task kernel's timeline
#endif
    h->doneArmed = false; // the chain's completion word was written before this kernel: results are waited for the slow way
    CAPE_HIP_TRY(cape::launch_polygons(p, n_frames, stream));
    h->polygonFrames = n_frames;
    return CAPE_OK;
}

static int match_polygons_impl(cape_handle h, int32_t n_frames, const double* prev_to_cur, uint32_t flags, void* stream_);

int cape_match_polygons(cape_handle h, int32_t n_frames, uint32_t flags, void* stream_)
{
    return match_polygons_impl(h, n_frames, nullptr, flags, stream_);
}

int cape_match_polygons_pose(cape_handle h, int32_t n_frames, const double* prev_to_cur, uint32_t flags, void* stream_)
{
    return match_polygons_impl(h, n_frames, prev_to_cur, flags, stream_);
}

static int match_polygons_impl(cape_handle h, int32_t n_frames, const double* prev_to_cur, uint32_t flags, void* stream_)
{
    if (!h || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle or negative frame count");
    if (n_frames > h->polygonFrames)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds the frames of the last cape_build_polygons (build the polygons of the batch first)");
    if (flags & ~(uint32_t)(CAPE_MATCH_ADVANCED | CAPE_MATCH_ALLOW_INDEX0))
        return fail(CAPE_ERR_INVALID_ARGUMENT, "unknown match flag");
    if (n_frames == 0)
        return CAPE_OK;
    CAPE_ON_DEVICE(h);
    const size_t pairCapacity = (size_t)h->cfg.max_batch * CAPE_MATCH_MAX_PLANES * CAPE_MATCH_MAX_PLANES;
    if (!h->matchesExact)
        CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->matchesExact), (size_t)h->cfg.max_batch * sizeof(cape_frame_match_exact)));
    if (!h->matchLists)
        CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->matchLists), (64 + 4 * pairCapacity) * sizeof(unsigned)));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StreamScope streamScope(h, stream);
    if (streamScope.rc() != CAPE_OK)
        return streamScope.rc();
    cape::MatchPolygonParams p;
    if (prev_to_cur)
    {
        // the poses travel to the device in the caller's memory order: n_frames x 16 doubles (entry 0 is never read)
        // The header promises that prev_to_cur is read before the call returns.  A hipMemcpyAsync straight from the caller's
        // memory keeps that promise only for pageable memory (the runtime then blocks -- behind everything queued on the stream);
        // from pinned memory it is truly asynchronous.  So: memcpy into a pinned twin of the handle (waiting for the H2D copy of
        // the previous call to have left it), then the asynchronous copy from there -- any host pointer, no implicit stream sync.
        const size_t poseBytes = (size_t)n_frames * 16 * sizeof(double);
        if (!h->matchPoses)
            CAPE_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->matchPoses), (size_t)h->cfg.max_batch * 16 * sizeof(double)));
        if (!h->matchPosesStage)
            CAPE_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->matchPosesStage), (size_t)h->cfg.max_batch * 16 * sizeof(double), hipHostMallocDefault));
        if (!h->matchPosesFree) // (its own guard: a failed creation must not leave the stage without its event for good)
            CAPE_HIP_TRY(hipEventCreateWithFlags(&h->matchPosesFree, hipEventDisableTiming));
        if (h->matchPosesBusy)
            CAPE_HIP_TRY(hipEventSynchronize(h->matchPosesFree));
        std::memcpy(h->matchPosesStage, prev_to_cur, poseBytes);
        CAPE_HIP_TRY(hipMemcpyAsync(h->matchPoses, h->matchPosesStage, poseBytes, hipMemcpyHostToDevice, stream));
        CAPE_HIP_TRY(hipEventRecord(h->matchPosesFree, stream));
        h->matchPosesBusy = true;
        p.poses = h->matchPoses;
    }
    p.records = h->records;
    p.polygons = h->polygons;
    p.vertices = reinterpret_cast<const double2*>(h->polyVertices);
    p.matches = h->matchesExact;
    p.listCounts = h->matchLists;
    p.pairLists = h->matchLists + 64;
    p.pairCapacity = pairCapacity;
    p.computeUnits = h->computeUnits;
    p.ldsLimitBytes = h->ldsLimit;
    p.boundaryCapacity = h->boundaryCap;
    p.flags = flags;
    p.minCosAngle = std::abs(std::cos(20.0 * M_PI / 180.0));
    p.maxDistance = 100.0;
    const double planeMinimalOverlap = static_cast<double>(0.4f);
    p.minOverlap = (flags & CAPE_MATCH_ADVANCED) ? planeMinimalOverlap / 2 : planeMinimalOverlap;
    CAPE_HIP_TRY(cape::launch_match_polygons(p, n_frames, stream));
    h->matchExactFrames = n_frames;
    return CAPE_OK;
}

int cape_copy_polygon_matches(cape_handle h, int32_t n_frames, cape_frame_match_exact* out)
{
    if (!h || !out || n_frames < 0 || n_frames > h->cfg.max_batch)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad argument");
    if (!h->matchesExact)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "cape_match_polygons has not run");
    if (n_frames > h->matchExactFrames)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds the frames of the last cape_match_polygons of the current batch");
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(drain_handle(h));
    CAPE_HIP_TRY(hipMemcpy(out, h->matchesExact, (size_t)n_frames * sizeof(cape_frame_match_exact), hipMemcpyDeviceToHost));
    return CAPE_OK;
}

int cape_device_polygons(cape_handle h, cape_polygon** polygons, double** vertices)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    if (h->polygonFrames <= 0)
        return fail(CAPE_ERR_CAPACITY, "no polygons of the current batch: cape_build_polygons has not run since the last cape_extract");
    if (polygons)
        *polygons = h->polygons;
    if (vertices)
        *vertices = h->polyVertices;
    return CAPE_OK;
}

int cape_copy_polygons(cape_handle h, int32_t n_frames, cape_polygon* polygons, double* vertices)
{
    if (!h || n_frames < 0 || n_frames > h->cfg.max_batch)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad handle / frame count");
    if (!h->polygons)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "no polygons have been built yet");
    if (n_frames > h->polygonFrames)
        return fail(CAPE_ERR_CAPACITY, "n_frames exceeds the frames of the last cape_build_polygons of the current batch");
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(drain_handle(h));
    const size_t n = (size_t)n_frames;
    const hipMemcpyKind kind = h->resultsOnHost ? hipMemcpyHostToHost : hipMemcpyDeviceToHost;
    if (polygons)
        CAPE_HIP_TRY(hipMemcpy(polygons, h->polygons, n * CAPE_MAX_PLANES * sizeof(cape_polygon), kind));
    if (vertices)
        CAPE_HIP_TRY(hipMemcpy(vertices, h->polyVertices, n * (size_t)h->boundaryCap * 2 * sizeof(double), kind));
    return CAPE_OK;
}

int cape_spill_info(cape_handle h, int32_t* used, int32_t* capacity, int32_t* frames)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    CAPE_ON_DEVICE(h);
    uint32_t c[2] = {0u, 0u};
    if (h->lastFrames > 0)
    {
        if (h->resultsOnHost)
        {
            if (const int rc = wait_results(h); rc != CAPE_OK)
                return rc;
        }
        else
            CAPE_HIP_TRY(drain_handle(h));
        CAPE_HIP_TRY(hipMemcpy(c, h->spillCounters, sizeof(c), hipMemcpyDeviceToHost));
    }
    if (used)
        *used = (int32_t)std::min<uint32_t>(c[0], (uint32_t)h->spillRecords);
    if (capacity)
        *capacity = h->spillRecords;
    if (frames)
        *frames = (int32_t)c[1];
    return CAPE_OK;
}

int cape_copy_spill(cape_handle h, int32_t first, int32_t count, cape_frame_record* records, double* boundary)
{
    if (!h || first < 0 || count < 0 || first > h->spillRecords || count > h->spillRecords - first)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad handle / spill record range");
    CAPE_ON_DEVICE(h);
    if (h->resultsOnHost)
    {
        if (const int rc = wait_results(h); rc != CAPE_OK)
            return rc;
    }
    else
        CAPE_HIP_TRY(drain_handle(h));
    const size_t at = (size_t)h->cfg.max_batch + (size_t)first, n = (size_t)count;
    const hipMemcpyKind kind = h->resultsOnHost ? hipMemcpyHostToHost : hipMemcpyDeviceToHost;
    if (records && n)
        CAPE_HIP_TRY(hipMemcpy(records, h->records + at, n * sizeof(cape_frame_record), kind));
    if (boundary && n)
        CAPE_HIP_TRY(hipMemcpy(boundary, h->boundary + at * (size_t)h->boundaryCap * 3, n * (size_t)h->boundaryCap * 3 * sizeof(double), kind));
    return CAPE_OK;
}

int cape_copy_spill_polygons(cape_handle h, int32_t first, int32_t count, cape_polygon* polygons, double* vertices)
{
    if (!h || first < 0 || count < 0 || first > h->spillRecords || count > h->spillRecords - first)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad handle / spill record range");
    if (!h->polygons || h->polygonFrames <= 0)
        return fail(CAPE_ERR_CAPACITY, "no cape_build_polygons has run on the current batch");
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(drain_handle(h));
    const size_t at = (size_t)h->cfg.max_batch + (size_t)first, n = (size_t)count;
    const hipMemcpyKind kind = h->resultsOnHost ? hipMemcpyHostToHost : hipMemcpyDeviceToHost;
    if (polygons && n)
        CAPE_HIP_TRY(hipMemcpy(polygons, h->polygons + at * CAPE_MAX_PLANES, n * CAPE_MAX_PLANES * sizeof(cape_polygon), kind));
    if (vertices && n)
        CAPE_HIP_TRY(hipMemcpy(vertices, h->polyVertices + at * (size_t)h->boundaryCap * 2, n * (size_t)h->boundaryCap * 2 * sizeof(double), kind));
    return CAPE_OK;
}

int cape_debug_polygon(cape_handle h, const double* points3, int32_t n, const double* normal, const double* center,
                       cape_polygon* polygon_out, double* vertices_out)
{
    if (!h || !points3 || !normal || !center || !polygon_out || n < 0 || n > h->boundaryCap)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument or more points than boundary_capacity");
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(drain_handle(h));
    // a one-plane frame of its own: record, boundary points, polygon and vertex arrays (freed on the way out)
    cape_frame_record* rec = nullptr;
    double* bnd = nullptr;
    cape_polygon* poly = nullptr;
    double* verts = nullptr;
    uint32_t* ladder = nullptr;
    cape_frame_record* hostRec = new (std::nothrow) cape_frame_record();
    if (!hostRec)
        return fail(CAPE_ERR_HIP, "out of host memory");
    std::memset(hostRec, 0, sizeof(*hostRec));
    hostRec->header.n_plane_segments = 1;
    hostRec->header.n_planes = 1;
    cape_plane_segment& s = hostRec->segments[0];
    for (int k = 0; k < 3; ++k)
    {
        s.normal[k] = normal[k];
        s.centroid[k] = center[k];
    }
    s.is_output = 1;
    s.planar = 1;
    s.boundary_offset = 0;
    s.boundary_count = (uint32_t)n;
    int rc = CAPE_OK;
    auto step = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == CAPE_OK)
            rc = fail(CAPE_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
        return e == hipSuccess;
    };
    const size_t cap = (size_t)h->boundaryCap;
    if (step(hipMalloc(reinterpret_cast<void**>(&rec), sizeof(cape_frame_record)), "hipMalloc") &&
        step(hipMalloc(reinterpret_cast<void**>(&bnd), cap * 3 * sizeof(double)), "hipMalloc") &&
        step(hipMalloc(reinterpret_cast<void**>(&poly), CAPE_MAX_PLANES * sizeof(cape_polygon)), "hipMalloc") &&
        step(hipMalloc(reinterpret_cast<void**>(&verts), cap * 2 * sizeof(double)), "hipMalloc") &&
        step(hipMalloc(reinterpret_cast<void**>(&ladder), cape::polygon_scratch_bytes(1, h->boundaryCap)), "hipMalloc") &&
        step(hipMemcpy(rec, hostRec, sizeof(cape_frame_record), hipMemcpyHostToDevice), "hipMemcpy") &&
        step(n ? hipMemcpy(bnd, points3, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice) : hipSuccess, "hipMemcpy"))
    {
        cape::PolygonParams p;
        p.records = rec;
        p.boundary = bnd;
        p.polygons = poly;
        p.vertices = reinterpret_cast<double2*>(verts);
        p.boundaryCapacity = h->boundaryCap;
        p.prof = nullptr;
        cape::polygon_bind_scratch(p, ladder, 1, h->boundaryCap);
        p.computeUnits = 4;
        p.originInCentroid = 1; // an arbitrary origin, as the caller asked
        if (step(cape::launch_polygons(p, 1, nullptr), "launch") && step(hipDeviceSynchronize(), "hipDeviceSynchronize") &&
            step(hipMemcpy(polygon_out, poly, sizeof(cape_polygon), hipMemcpyDeviceToHost), "hipMemcpy"))
        {
            if (vertices_out && polygon_out->vertex_count)
                step(hipMemcpy(vertices_out, verts, (size_t)polygon_out->vertex_count * 2 * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");
        }
    }
    (void)hipFree(rec);
    (void)hipFree(bnd);
    (void)hipFree(poly);
    (void)hipFree(verts);
    (void)hipFree(ladder);
    delete hostRec;
    return rc;
}

int cape_debug_cycles(cape_handle h, int32_t n_frames, unsigned long long* out)
{
    if (!h || !out || n_frames < 0 || n_frames > h->cfg.max_batch)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad argument");
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(hipDeviceSynchronize());
    CAPE_HIP_TRY(hipMemcpy(out, h->debugCycles, (size_t)n_frames * cape::kProfileSlots * 8, hipMemcpyDeviceToHost));
    return CAPE_OK;
}

int cape_debug_rectify_flagged(cape_handle h, int32_t* count)
{
    if (!h || !count)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad argument");
    *count = 0;
    if (!h->rectFlags)
        return CAPE_OK;
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(hipDeviceSynchronize());
    unsigned n = 0;
    CAPE_HIP_TRY(hipMemcpy(&n, h->rectFlags + h->rectFlagFrames, sizeof n, hipMemcpyDeviceToHost));
    *count = (int32_t)n;
    return CAPE_OK;
}

int cape_debug_polygon_queue(cape_handle h, uint32_t* reserved, uint32_t* tickets, uint32_t* slots)
{
    if (!h || !reserved || !tickets || !slots)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad argument");
    *reserved = *tickets = *slots = 0;
    if (!h->polyLadder || h->polygonFrames <= 0)
        return CAPE_OK;
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(hipDeviceSynchronize());
    cape::PolygonParams p;
    cape::polygon_bind_scratch(p, h->polyLadder, (size_t)h->cfg.max_batch + (size_t)h->spillRecords, h->boundaryCap);
    uint32_t hd[2] = {0, 0};
    CAPE_HIP_TRY(hipMemcpy(hd, p.queue, sizeof hd, hipMemcpyDeviceToHost));
    *reserved = hd[0];
    *tickets = hd[1];
    const size_t wanted = cape::polygon_queue_slots((size_t)h->polygonFrames + (size_t)h->spillRecords); // (the batch + the spill pool)
    *slots = (uint32_t)(wanted < (size_t)p.queueCapacity ? wanted : (size_t)p.queueCapacity);
    return CAPE_OK;
}

int cape_log_records(const cape_frame_record* records, int32_t n_frames, cape_log_fn fn, void* user)
{
    if (!records || !fn || n_frames < 0)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null records / callback or negative frame count");
    int lines = 0;
    // (a chain is followed as far as the caller's array reaches: records + pool copied together hold all of it)
    auto next = [&](int idx) -> const cape_frame_record* { return idx >= 0 && idx < n_frames ? records + idx : nullptr; };
    for (int f = 0; f < n_frames; ++f)
        lines += log_frame(fn, user, records[f], f, next);
    return lines;
}

int cape_debug_match_lists(cape_handle h, uint32_t* words32)
{
    if (!h || !words32)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "bad argument");
    std::memset(words32, 0, 32 * sizeof(uint32_t));
    if (!h->matchLists)
        return CAPE_OK;
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(hipDeviceSynchronize());
    CAPE_HIP_TRY(hipMemcpy(words32, h->matchLists, 32 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CAPE_OK;
}

int cape_set_rng_seed(cape_handle h, uint32_t seed)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    CAPE_ON_DEVICE(h);
    CAPE_HIP_TRY(drain_handle(h)); // a grow kernel in flight may still be drawing from the table
    const std::vector<double> rng = rng_table(seed, h->rngCount);
    CAPE_HIP_TRY(hipMemcpy(h->rng, rng.data(), rng.size() * sizeof(double), hipMemcpyHostToDevice));
    return CAPE_OK;
}

int cape_set_log_callback(cape_handle h, cape_log_fn fn, void* user)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    h->logFn = fn;
    h->logUser = user;
    return CAPE_OK;
}

int cape_enable_timing(cape_handle h, int32_t enable)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    h->timing = enable != 0;
    return CAPE_OK;
}

int cape_get_timings(cape_handle h, cape_timings* out)
{
    if (!h || !out)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    CAPE_ON_DEVICE(h);
    const int rc = fold_timings(h);
    if (rc != CAPE_OK)
        return rc;
    *out = h->tm;
    return CAPE_OK;
}

int cape_get_timings_sized(cape_handle h, void* out, uint64_t out_bytes)
{
    if (!h || !out)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null argument");
    CAPE_ON_DEVICE(h);
    const int rc = fold_timings(h);
    if (rc != CAPE_OK)
        return rc;
    std::memcpy(out, &h->tm, (size_t)std::min<uint64_t>(out_bytes, sizeof(cape_timings)));
    return CAPE_OK;
}

int cape_reset_timings(cape_handle h)
{
    if (!h)
        return fail(CAPE_ERR_INVALID_ARGUMENT, "null handle");
    CAPE_ON_DEVICE(h);
    const int rc = fold_timings(h);
    h->tm = cape_timings{};
    if (h->phaseTicks)
        CAPE_HIP_TRY(hipMemset(h->phaseTicks, 0, (size_t)h->cfg.max_batch * 4 * 8));
    return rc;
}

} // extern "C"
