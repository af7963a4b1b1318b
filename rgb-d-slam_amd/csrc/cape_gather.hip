// Multi-GPU exchange of per-frame primitive lists (BASELINE.json configs[3], [4]).
//
// Frames are independent, so a batch shards by contiguous frame blocks, one GPU each, with no collective on the data
// path.  What IS exchanged, once per batch, is what plane_container / cylinder_container hold after find_primitives
// (reference src/features/primitives/shape_primitives.hpp:129-130) -- ragged lists.  A fixed slot per frame either
// truncates (round 1: 16 planes) or ships mostly padding (64 planes x 152 B), so the lists are PACKED on the device:
//
//   [cape_packed_header][frames_capacity x cape_packed_frame][planes_capacity x cape_packed_plane]
//   [cylinders_capacity x cape_packed_cylinder][frames_capacity x cells u8 plane labels][same, cylinder labels]
//
// with the capacities sized per BATCH (frames_capacity x planes_per_frame), so a frame with 40 planes costs nothing as
// long as the batch average stays under planes_per_frame; planes_per_frame = CAPE_MAX_PLANES can never overflow, and an
// overflow is reported in the header, never silent.  Every rank sends the same byte count, which is what one
// ncclAllGather needs.  Two kernels: an exclusive scan of the per-frame counts (one workgroup), then one wavefront per
// frame that copies its primitives -- CAPE_MAX_PLANES = CAPE_MAX_CYLINDERS = 64 = the wave width, so the rank of a
// kept primitive among its frame's is a ballot + popcount.
//
// The collective is RCCL, called from this layer (north_star: "host code stays C++ and calls into a thin extern-C HIP
// layer ... RCCL-over-xGMI gather"): librccl is resolved with dlopen at the first cape_comm_* call, so single-GPU users
// never load it and a process that already holds torch's copy binds to that one.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <dlfcn.h>

#include <string>

#include "cape_internal.h"

namespace cape {

constexpr int kScanThreads = 1024;

// exclusive scan of (n_planes, n_cylinders) over the frames of the shard; writes the frame table and the header
__global__ __launch_bounds__(kScanThreads) void cape_pack_scan_kernel(PackParams p)
{
    __shared__ int s_p[kScanThreads], s_c[kScanThreads];
    __shared__ unsigned s_status;
    const int t = threadIdx.x;
    const int per = (p.nFrames + kScanThreads - 1) / kScanThreads;
    const int f0 = t * per, f1 = min(f0 + per, p.nFrames);
    if (t == 0)
        s_status = 0u;
    int sp = 0, sc = 0;
    unsigned st = 0;
    for (int f = f0; f < f1; ++f)
    {
        const cape_frame_header& h = p.records[f].header; // (a frame's first record holds the whole frame's counts)
        sp += h.n_planes;
        sc += h.n_cylinders;
        st |= h.status & 0xFFu; // the flag bits; bits 8..15 are a per-frame COUNT (not-planar-after-merge), which an OR would garble
        if (h.n_plane_segments > 255 || h.n_cylinder_labels > 255)
            st |= 1u << 31; // label grids travel as bytes: folded into hd.overflow below
    }
    s_p[t] = sp;
    s_c[t] = sc;
    __syncthreads();
    if (st)
        atomicOr(&s_status, st);
    // Hillis-Steele inclusive scan over the 1024 partials
    for (int o = 1; o < kScanThreads; o <<= 1)
    {
        const int ap = (t >= o) ? s_p[t - o] : 0, ac = (t >= o) ? s_c[t - o] : 0;
        __syncthreads();
        s_p[t] += ap;
        s_c[t] += ac;
        __syncthreads();
    }
    int op = s_p[t] - sp, oc = s_c[t] - sc; // exclusive offsets of this thread's first frame
    for (int f = f0; f < f1; ++f)
    {
        const cape_frame_header& h = p.records[f].header;
        cape_packed_frame pf;
        pf.plane_offset = op;
        pf.n_planes = h.n_planes;
        pf.cylinder_offset = oc;
        pf.n_cylinders = h.n_cylinders;
        pf.status = h.status;
        pf.n_plane_segments = h.n_plane_segments;
        p.frames[f] = pf;
        op += h.n_planes;
        oc += h.n_cylinders;
    }
    // the table entries past the shard's own frames stay zero so that equal inputs give equal bytes on the wire
    for (int f = p.nFrames + t; f < p.framesCapacity; f += kScanThreads)
    {
        cape_packed_frame z = {0, 0, 0, 0, 0u, 0};
        p.frames[f] = z;
    }
    if (t == kScanThreads - 1)
    {
        cape_packed_header hd;
        hd.magic = CAPE_PACKED_MAGIC;
        hd.n_frames = p.nFrames;
        hd.first_frame = p.firstFrame;
        hd.n_planes_total = s_p[t];
        hd.n_cylinders_total = s_c[t];
        hd.planes_capacity = p.planesCapacity;
        hd.cylinders_capacity = p.cylindersCapacity;
        hd.overflow = (s_p[t] > p.planesCapacity ? CAPE_PACKED_PLANES_DROPPED : 0u) |
                      (s_c[t] > p.cylindersCapacity ? CAPE_PACKED_CYLINDERS_DROPPED : 0u) |
                      (((s_status >> 31) && (p.flags & CAPE_GATHER_LABELS)) ? CAPE_PACKED_LABELS_CLIPPED : 0u);
        hd.status_or = s_status & 0xFFu;
        hd.cells = p.cells;
        hd.frames_capacity = p.framesCapacity;
        hd.flags = p.flags;
        *p.header = hd;
    }
}

// one wavefront per frame: lane i looks at plane segment i / cylinder label i of the frame
__global__ __launch_bounds__(256) void cape_pack_copy_kernel(PackParams p)
{
    const int lane = threadIdx.x & 63;
    const int frame = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (frame >= p.framesCapacity)
        return;
    const int C = p.cells;
    if (frame >= p.nFrames)
    {
        // label rows of the unused frame slots are zeroed (deterministic wire bytes)
        if (p.flags & CAPE_GATHER_LABELS)
            for (int i = lane; i < C; i += 64)
            {
                p.planeLabels8[(size_t)frame * C + i] = 0;
                p.cylLabels8[(size_t)frame * C + i] = 0;
            }
        return;
    }
    const cape_packed_frame pf = p.frames[frame];
    // a frame of more than 64 plane segments / cylinder labels is a chain of records (cape_frame_header::next_record, an index into
    // the handle's record array): the planes and cylinders of every record of the chain, in order
    int planesBefore = 0, cylsBefore = 0;
    for (const cape_frame_record* recp = &p.records[frame]; recp; recp = recp->header.next_record >= p.poolBase ? p.recordsBase + recp->header.next_record : nullptr)
    {
    const cape_frame_record& rec = *recp;
    // planes: k-th kept segment (is_output) -> planes[plane_offset + k]
    {
        const bool keep = lane < rec.header.n_plane_segments && rec.segments[lane].is_output != 0;
        const unsigned long long m = __ballot(keep);
        const int k = planesBefore + __popcll(m & ((1ull << lane) - 1ull));
        planesBefore += __popcll(m);
        const int dst = pf.plane_offset + k;
        if (keep && dst < p.planesCapacity)
        {
            const cape_plane_segment& s = rec.segments[lane];
            cape_packed_plane o;
            o.normal[0] = s.out_normal[0]; o.normal[1] = s.out_normal[1]; o.normal[2] = s.out_normal[2];
            o.d = s.d;
            o.centroid[0] = s.centroid[0]; o.centroid[1] = s.centroid[1]; o.centroid[2] = s.centroid[2];
            o.mse = s.mse;
            o.score = s.score;
#pragma unroll
            for (int q = 0; q < 9; ++q)
                o.sums[q] = s.sums[q];
            o.point_count = s.point_count;
            o.segment = (uint32_t)(rec.header.segment_base + lane);
            p.planes[dst] = o;
        }
    }
    // cylinders: k-th kept label -> cylinders[cylinder_offset + k]
    {
        const bool keep = lane < rec.header.n_cylinder_labels && rec.cylinders[lane].kept != 0;
        const unsigned long long m = __ballot(keep);
        const int k = cylsBefore + __popcll(m & ((1ull << lane) - 1ull));
        cylsBefore += __popcll(m);
        const int dst = pf.cylinder_offset + k;
        if (keep && dst < p.cylindersCapacity)
        {
            cape_packed_cylinder o;
            o.axis[0] = rec.cylinders[lane].axis[0];
            o.axis[1] = rec.cylinders[lane].axis[1];
            o.axis[2] = rec.cylinders[lane].axis[2];
            o.radius = rec.cylinders[lane].radius;
            p.cylinders[dst] = o;
        }
    }
    }
    if (p.flags & CAPE_GATHER_LABELS)
    {
        // _gridPlaneSegmentMap / _gridCylinderSegMap (primitive_detection.hpp:212-214) travel as bytes; a label beyond 255 (a frame of
        // that many segments) is clipped to 255 and the header says so (CAPE_PACKED_LABELS_CLIPPED)
        for (int i = lane; i < C; i += 64)
        {
            const int32_t a = p.planeLabelsIn[(size_t)frame * C + i], b = p.cylLabelsIn[(size_t)frame * C + i];
            p.planeLabels8[(size_t)frame * C + i] = (uint8_t)(a > 255 ? 255 : a);
            p.cylLabels8[(size_t)frame * C + i] = (uint8_t)(b > 255 ? 255 : b);
        }
    }
}

// primitives slots past the batch totals are zeroed (same reason as above); grid-stride over 8-byte words
__global__ __launch_bounds__(256) void cape_pack_clear_tail_kernel(PackParams p)
{
    const cape_packed_header hd = *p.header;
    const size_t pUsed = (size_t)min(hd.n_planes_total, p.planesCapacity) * (sizeof(cape_packed_plane) / 8);
    const size_t pAll = (size_t)p.planesCapacity * (sizeof(cape_packed_plane) / 8);
    const size_t cUsed = (size_t)min(hd.n_cylinders_total, p.cylindersCapacity) * (sizeof(cape_packed_cylinder) / 8);
    const size_t cAll = (size_t)p.cylindersCapacity * (sizeof(cape_packed_cylinder) / 8);
    unsigned long long* pw = reinterpret_cast<unsigned long long*>(p.planes);
    unsigned long long* cw = reinterpret_cast<unsigned long long*>(p.cylinders);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = pUsed + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pAll; i += stride)
        pw[i] = 0ull;
    for (size_t i = cUsed + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cAll; i += stride)
        cw[i] = 0ull;
}

// totals of a batch (what sizes a tight payload budget): one wave-reduced atomic per workgroup
__global__ __launch_bounds__(256) void cape_count_primitives_kernel(const cape_frame_record* records, int nFrames, int32_t* out)
{
    int planes = 0, cyls = 0, maxPlanes = 0;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nFrames; f += gridDim.x * blockDim.x)
    {
        const int np = records[f].header.n_planes;
        planes += np;
        cyls += records[f].header.n_cylinders;
        maxPlanes = max(maxPlanes, np);
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        planes += __shfl_xor(planes, o);
        cyls += __shfl_xor(cyls, o);
        maxPlanes = max(maxPlanes, __shfl_xor(maxPlanes, o));
    }
    if ((threadIdx.x & 63) == 0)
    {
        atomicAdd(&out[0], planes);
        atomicAdd(&out[1], cyls);
        atomicMax(&out[2], maxPlanes);
    }
}

hipError_t launch_count_primitives(const cape_frame_record* records, int nFrames, int32_t* out, hipStream_t stream)
{
    if (const hipError_t e = hipMemsetAsync(out, 0, 4 * sizeof(int32_t), stream); e != hipSuccess)
        return e;
    const int blocks = max(1, min(64, (nFrames + 255) / 256));
    hipLaunchKernelGGL(cape_count_primitives_kernel, dim3(blocks), dim3(256), 0, stream, records, nFrames, out);
    return hipGetLastError();
}

hipError_t launch_pack(const PackParams& p, hipStream_t stream)
{
    hipLaunchKernelGGL(cape_pack_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, p);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(cape_pack_copy_kernel, dim3((p.framesCapacity + 3) / 4), dim3(256), 0, stream, p);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(cape_pack_clear_tail_kernel, dim3(256), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// RCCL, resolved at run time
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct RcclApi
{
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*Gather)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr; // RCCL extension; optional
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;    // what the communicator itself reports (cape_comm_info); optional
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*CommCuDevice)(void*, int*) = nullptr;
    std::string error;
};
RcclApi g_rccl;
} // namespace

const char* rccl_load()
{
    if (g_rccl.lib)
        return nullptr;
    if (!g_rccl.error.empty())
        return g_rccl.error.c_str();
    // RTLD_NOLOAD first: a process that already holds an RCCL (torch ships its own librccl.so) must keep ONE copy
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* lib = nullptr;
    if (const char* forced = std::getenv("CAPE_RCCL_LIB")) // an explicit library (deployments with several ROCm trees)
        lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    else
    {
        for (const char* n : names)
            if (!lib)
                lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char* n : names)
            if (!lib)
                lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!lib)
            lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    }
    if (!lib)
    {
        const char* why = dlerror(); // (a second call would return null: the first one clears the message)
        g_rccl.error = std::string("librccl.so not found: ") + (why ? why : "");
        return g_rccl.error.c_str();
    }
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    g_rccl.Gather = reinterpret_cast<decltype(g_rccl.Gather)>(sym("ncclGather"));
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(sym("ncclCommCount"));
    g_rccl.CommUserRank = reinterpret_cast<decltype(g_rccl.CommUserRank)>(sym("ncclCommUserRank"));
    g_rccl.CommCuDevice = reinterpret_cast<decltype(g_rccl.CommCuDevice)>(sym("ncclCommCuDevice"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString)
    {
        g_rccl.error = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather";
        return g_rccl.error.c_str();
    }
    g_rccl.lib = lib;
    return nullptr;
}

int rccl_unique_id(RcclUniqueId* id) { return g_rccl.GetUniqueId(id); }
int rccl_comm_init(void** comm, int world, const RcclUniqueId& id, int rank) { return g_rccl.CommInitRank(comm, world, id, rank); }
int rccl_comm_destroy(void* comm) { return g_rccl.CommDestroy(comm); }
int rccl_all_gather_bytes(const void* send, void* recv, size_t bytes, void* comm, hipStream_t stream)
{
    return g_rccl.AllGather(send, recv, bytes, /* ncclChar */ 0, comm, stream);
}
bool rccl_has_gather() { return g_rccl.Gather != nullptr; }
int rccl_gather_bytes(const void* send, void* recv, size_t bytes, int root, void* comm, hipStream_t stream)
{
    return g_rccl.Gather(send, recv, bytes, /* ncclChar */ 0, root, comm, stream);
}
// what RCCL says about a communicator: -1 where the loaded library lacks the query
void rccl_comm_query(void* comm, int* count, int* rank, int* device)
{
    *count = *rank = *device = -1;
    if (g_rccl.CommCount && g_rccl.CommCount(comm, count) != 0)
        *count = -1;
    if (g_rccl.CommUserRank && g_rccl.CommUserRank(comm, rank) != 0)
        *rank = -1;
    if (g_rccl.CommCuDevice && g_rccl.CommCuDevice(comm, device) != 0)
        *device = -1;
}
const char* rccl_error_string(int code) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "rccl not loaded"; }

} // namespace cape
