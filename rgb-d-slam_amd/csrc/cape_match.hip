// N2 ("next" row of SURVEY.md 8f), device part: cell-mask pre-filter of MapPlane::find_matches
// (reference src/map_management/map_features/map_primitive.cpp:91-161, driven by feature_map.hpp:651-669) between
// consecutive frames of a batch.  One wavefront per frame f: the label grids of f-1 and f are streamed once (2 x cells x
// 4 B, coalesced), cell counts of every (previous plane, plane) pair are accumulated with LDS atomics -- integer, so
// the order is free -- and the selection loop runs with one lane per detected plane and a packed-key wave maximum per
// previous plane.  HBM-bound in principle (6 KB + two record headers per frame), microseconds per batch in practice.
#include <hip/hip_runtime.h>

#include "cape_internal.h"

namespace cape {

namespace {

constexpr int kWavesPerGroup = 2;
constexpr int P = CAPE_MAX_PLANES;

// label -> output plane index table of one frame, in LDS: map[0] = -1 (no plane), map[k + 1] = index among the
// is_output segments of the merge root of segment k (the mask loop of primitive_detection.cpp:586-594 adds segment j
// to root r's mask iff planeMergeLabels[j] == r and j >= r), -1 if that root is not an output plane.
// s_segOf[i] receives the segment index of output plane i.  Returns the number of output planes.
__device__ __forceinline__ int build_label_map(const cape_frame_record& rec, int lane, int* s_map, int* s_segOf)
{
    int nSeg = rec.header.n_plane_segments;
    nSeg = nSeg < 0 ? 0 : (nSeg > P ? P : nSeg);
    const bool mine = lane < nSeg;
    const uint32_t root = mine ? rec.segments[lane].merge_label : 0u;
    const bool isOut = mine && rec.segments[lane].is_output != 0;
    const unsigned long long outMask = __ballot(isOut);
    const int myIdx = isOut ? __popcll(outMask & ((1ull << lane) - 1ull)) : -1;
    const int rootIdx = __shfl(myIdx, (int)(root & 63u));
    if (lane == 0)
        s_map[0] = -1;
    if (lane < P)
        s_map[lane + 1] = (mine && root < (uint32_t)nSeg && root <= (uint32_t)lane) ? rootIdx : -1;
    if (isOut)
        s_segOf[myIdx] = lane;
    return __popcll(outMask);
}

} // namespace

__global__ __launch_bounds__(64 * kWavesPerGroup) void cape_match_kernel(MatchParams p, int nFrames)
{
    __shared__ unsigned int s_inter[kWavesPerGroup][P * P];
    __shared__ unsigned int s_area[kWavesPerGroup][2 * P];
    __shared__ int s_map[kWavesPerGroup][2 * (P + 1)];
    __shared__ int s_seg[kWavesPerGroup][2 * P];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int frame = blockIdx.x * kWavesPerGroup + wave;
    if (frame >= nFrames)
        return;
    cape_frame_match& out = p.matches[frame];
    unsigned int* inter = s_inter[wave];
    unsigned int* area = s_area[wave]; // [0,P) previous, [P,2P) current
    int* mapPrev = s_map[wave];
    int* mapCur = s_map[wave] + (P + 1);

    for (int k = lane; k < P * P; k += 64)
        inter[k] = 0;
    for (int k = lane; k < 2 * P; k += 64)
        area[k] = 0;

    int nPrev = 0;
    const int nCur = build_label_map(p.records[frame], lane, mapCur, s_seg[wave] + P);
    if (frame > 0)
        nPrev = build_label_map(p.records[frame - 1], lane, mapPrev, s_seg[wave]);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int segCur = (lane < nCur) ? s_seg[wave][P + lane] : -1;
    const int segPrev = (lane < nPrev) ? s_seg[wave][lane] : -1;

    const int32_t* labCur = p.plane_labels + (size_t)frame * p.cells;
    const int32_t* labPrev = labCur - p.cells;
    for (int c = lane; c < p.cells; c += 64)
    {
        const int lc = labCur[c];
        const int a = (lc >= 0 && lc <= P) ? mapCur[lc] : -1;
        if (a >= 0)
            atomicAdd(&area[P + a], 1u);
        if (frame > 0)
        {
            const int lp = labPrev[c];
            const int b = (lp >= 0 && lp <= P) ? mapPrev[lp] : -1;
            if (b >= 0)
            {
                atomicAdd(&area[b], 1u);
                if (a >= 0)
                    atomicAdd(&inter[b * P + a], 1u);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

    // lane i < nCur : detected plane i of this frame
    double n0 = 0, n1 = 0, n2 = 0, d = 0;
    if (lane < nCur && segCur >= 0)
    {
        const cape_plane_segment& s = p.records[frame].segments[segCur];
        n0 = s.out_normal[0], n1 = s.out_normal[1], n2 = s.out_normal[2], d = s.d;
    }
    const double myArea = (lane < nCur) ? (double)area[P + lane] : 0.0;
    bool matched = false; // _isDetectedFeatureMatched[lane]
    int myMatch = -1;     // lane j < nPrev: result for previous plane j
    for (int j = 0; j < nPrev; ++j)
    {
        const int sj = __shfl(segPrev, j);
        const cape_plane_segment& q = p.records[frame - 1].segments[sj < 0 ? 0 : sj];
        const double q0 = q.out_normal[0], q1 = q.out_normal[1], q2 = q.out_normal[2], qd = q.d;
        const unsigned int projectedArea = area[j];
        unsigned long long key = 0;
        if (lane < nCur && !matched && projectedArea > 0)
        {
            const double cosAngle = (n0 * q0 + n1 * q1) + n2 * q2; // PlaneCoordinates::get_cos_angle = Eigen dot
            const bool distanceSimilar = fabs(d - qd) < p.maxDistance;
            const bool normalSimilar = fabs(cosAngle) > p.minCosAngle;
            const unsigned int ia = inter[j * P + lane];
            // `interArea > greatestSimilarity` with greatestSimilarity starting at 0.0 : ia > 0 ; ascending scan with a
            // strict comparison = the lowest index among the largest inter areas
            if (distanceSimilar && normalSimilar && ia > 0 && (double)ia / myArea >= p.minOverlap)
                key = ((unsigned long long)ia << 8) | (unsigned long long)(63 - lane);
        }
        // wave maximum of the packed key
        for (int off = 32; off > 0; off >>= 1)
        {
            const unsigned long long o = __shfl_xor(key, off);
            key = o > key ? o : key;
        }
        int selected = key ? 63 - (int)(key & 0xffull) : -1;
        if (!(p.flags & CAPE_MATCH_ALLOW_INDEX0) && selected <= 0) // map_primitive.cpp:146
            selected = -1;
        if (selected >= 0 && lane == selected)
            matched = true;
        if (lane == j)
            myMatch = selected;
    }

    if (lane == 0)
    {
        out.n_prev = nPrev;
        out.n_cur = nCur;
    }
    if (lane < P)
    {
        out.match[lane] = (lane < nPrev) ? myMatch : -1;
        out.area_prev[lane] = (uint16_t)area[lane];
        out.area_cur[lane] = (uint16_t)area[P + lane];
    }
    for (int k = lane; k < P * P; k += 64)
        (&out.inter[0][0])[k] = (uint16_t)inter[k];
}

hipError_t launch_match(const MatchParams& p, int nFrames, hipStream_t stream)
{
    const int groups = (nFrames + kWavesPerGroup - 1) / kWavesPerGroup;
    hipLaunchKernelGGL(cape_match_kernel, dim3(groups), dim3(64 * kWavesPerGroup), 0, stream, p, nFrames);
    return hipGetLastError();
}

} // namespace cape
