// Ordered traversal of per-cell records that live in HBM, for ONE wave.
//
// Every floating-point sum the reference accumulates in a loop is order-sensitive, so the grow wave has to visit the
// records in ascending index order -- a dependent chain.  Reading the records straight from global memory inside the
// chain would expose the full memory latency (~2-3 k cycles) per element.  staged_for_each() instead lets all 64
// lanes fetch a chunk of kStageChunk records as coalesced 16-byte pieces into registers, parks them in a small LDS
// buffer, and runs the caller's body over the chunk from LDS while the NEXT chunk's pieces are already in flight.
// Only compiler-level fences separate the LDS write from the LDS reads (single wave, LDS executes in issue order), so
// the outstanding global loads are not drained at the hand-over.
#pragma once
#include <hip/hip_runtime.h>

namespace cape {

#ifndef CAPE_STAGE_CHUNK
#define CAPE_STAGE_CHUNK 32
#endif
constexpr int kStageChunk = CAPE_STAGE_CHUNK; // records per chunk; the LDS buffer must hold kStageChunk * 2 * PIECES doubles

#define CAPE_STAGE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

// PIECES  : 16-byte pieces (double2) per record, 1..5
// index(e): record number of element e (e.g. the activated-cell list), e in [0, N)
// body(e, rec): called for e = 0..N-1 in order by ALL lanes (uniform); rec points at the 2*PIECES doubles in LDS
template <int PIECES, typename IndexFn, typename Body>
__device__ __forceinline__ void staged_for_each(int N, const double* base, int strideDoubles, int firstPiece, IndexFn index,
                                                double* s_buf, int lane, Body body)
{
    constexpr int kPiecesPerChunk = kStageChunk * PIECES;
    constexpr int kPerLane = (kPiecesPerChunk + 63) / 64; // pieces each lane moves per chunk
    static_assert(PIECES >= 1 && PIECES <= 5 && kPerLane <= 5, "record / chunk too large for the staging registers");
    double2 r0 = make_double2(0, 0), r1 = r0, r2 = r0, r3 = r0, r4 = r0;

#define CAPE_STAGE_SRC(q, c0_) \
    (base + (size_t)index((c0_) + ((lane + 64 * (q)) / PIECES)) * strideDoubles + 2 * (firstPiece + ((lane + 64 * (q)) % PIECES)))
#define CAPE_STAGE_DST(q) (s_buf + ((lane + 64 * (q)) / PIECES) * 2 * PIECES + 2 * ((lane + 64 * (q)) % PIECES))
    auto issue = [&](int c0) {
        const int np = ((N - c0 < kStageChunk) ? (N - c0) : kStageChunk) * PIECES;
        if (lane < np)
            r0 = *reinterpret_cast<const double2*>(CAPE_STAGE_SRC(0, c0));
        if (kPerLane > 1 && lane + 64 < np)
            r1 = *reinterpret_cast<const double2*>(CAPE_STAGE_SRC(1, c0));
        if (kPerLane > 2 && lane + 128 < np)
            r2 = *reinterpret_cast<const double2*>(CAPE_STAGE_SRC(2, c0));
        if (kPerLane > 3 && lane + 192 < np)
            r3 = *reinterpret_cast<const double2*>(CAPE_STAGE_SRC(3, c0));
        if (kPerLane > 4 && lane + 256 < np)
            r4 = *reinterpret_cast<const double2*>(CAPE_STAGE_SRC(4, c0));
    };

    if (N > 0)
        issue(0);
    for (int c0 = 0; c0 < N; c0 += kStageChunk)
    {
        const int cn = (N - c0 < kStageChunk) ? (N - c0) : kStageChunk;
        const int np = cn * PIECES;
        if (lane < np)
            *reinterpret_cast<double2*>(CAPE_STAGE_DST(0)) = r0;
        if (kPerLane > 1 && lane + 64 < np)
            *reinterpret_cast<double2*>(CAPE_STAGE_DST(1)) = r1;
        if (kPerLane > 2 && lane + 128 < np)
            *reinterpret_cast<double2*>(CAPE_STAGE_DST(2)) = r2;
        if (kPerLane > 3 && lane + 192 < np)
            *reinterpret_cast<double2*>(CAPE_STAGE_DST(3)) = r3;
        if (kPerLane > 4 && lane + 256 < np)
            *reinterpret_cast<double2*>(CAPE_STAGE_DST(4)) = r4;
        if (c0 + kStageChunk < N)
            issue(c0 + kStageChunk);
        CAPE_STAGE_FENCE();
        if (cn == kStageChunk)
        {
#pragma unroll
            for (int ci = 0; ci < kStageChunk; ++ci)
                body(c0 + ci, s_buf + ci * 2 * PIECES);
        }
        else
        {
            for (int ci = 0; ci < cn; ++ci)
                body(c0 + ci, s_buf + ci * 2 * PIECES);
        }
        CAPE_STAGE_FENCE();
    }
#undef CAPE_STAGE_SRC
#undef CAPE_STAGE_DST
}

} // namespace cape
