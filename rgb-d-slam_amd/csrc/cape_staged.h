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

// chunks in flight: the plane-only grow kernel runs three waves per SIMD and has ~20 registers to spare, the cylinder
// variant one wave per SIMD and plenty
#ifndef CAPE_STAGE_DEPTH_MAIN
#define CAPE_STAGE_DEPTH_MAIN 2
#endif
#ifndef CAPE_STAGE_DEPTH_CYL
#define CAPE_STAGE_DEPTH_CYL 2
#endif
#define CAPE_STAGE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

// default park(): the chunk buffer holds the records as they are in memory, [element][2 * PIECES] doubles
template <int PIECES> __device__ __forceinline__ void stage_park_records(double* buf, int piece, double2 v)
{
    *reinterpret_cast<double2*>(buf + (piece / PIECES) * 2 * PIECES + 2 * (piece % PIECES)) = v;
}

// PIECES  : 16-byte pieces (double2) per record, 1..5
// index(e): record number of element e (e.g. the activated-cell list), e in [0, N)
// load(e, rec) -> value  : the lane's operand(s) of element e, plain reads of the record's 2*PIECES doubles in LDS (no
//                          arithmetic, no side effects; called for a whole group of elements before any is folded)
// fold(e, v)             : the order-sensitive step (e.g. acc += v), called for e = 0..N-1 in order by ALL lanes
// The split is what lets the LDS reads of eight elements be requested back to back while the dependent adds of the
// previous eight run; a single body(e, rec) left that to the scheduler, which emitted read - wait - use per element
// (one LDS round trip each) -- and it does the same inside load() if load() computes on what it reads.
//
// Every lane moves its pieces UNCONDITIONALLY (indices clamped to the last element / last piece, so the surplus lanes
// re-write a valid record into a slot nobody reads): predicated loads would put each one behind a divergent branch, and
// the compiler's wait-count insertion then serialises them with `s_waitcnt vmcnt(0)` -- and would also wait for the
// prefetched chunk before the first add of the current one.  For the same reason the caller's accumulators must not be
// the destination of a still-pending global load: the entry drains the counters once.
//
// prep(c0, cn): called by all lanes right before the cn bodies of the chunk that starts at element c0 (e.g. to ballot a
// per-element flag into a uniform mask, so that the bodies stay free of LDS look-ups and branches)
// DEPTH (2..4): chunks requested ahead of the one being consumed; each costs ceil(32 * PIECES / 64) double2 registers.
//
// staged_for_each_src is the general form: the pieces of a record may come from more than one array (two traversals over
// the same elements folded into one chain, e.g. different quantities in different lanes):
// index(e, sub) -> record number that holds piece `sub` of element e;  addr(rec, sub) -> its 16 bytes;
// keep(e) -> false parks +0.0 in place of element e's record (a masked sum: the select is paid once per piece by the lanes
// that stage the chunk, not once per element on the chain; looked up when the chunk is requested).
// park(buf, piece, v)    : how a fetched piece (number `piece` = element-in-chunk * PIECES + sub, value v) is written to the
//                          chunk buffer; called by all 64 lanes together, so it may exchange pieces between lanes first (the
//                          cylinder covariance parks the six products of a cell instead of its normal).  The default
//                          layout is [element][2 * PIECES] doubles, which is what load() is handed a pointer into.
template <int PIECES, int DEPTH, typename IndexFn, typename AddrFn, typename Keep, typename Park, typename Prep, typename Load,
          typename Fold>
__device__ __forceinline__ void staged_for_each_src(int N_, IndexFn index, AddrFn addr, Keep keep, Park park, double* s_buf, int lane,
                                                    Prep prep, Load load, Fold fold, unsigned long long* prof = nullptr)
{
    constexpr int kPiecesPerChunk = kStageChunk * PIECES;
    constexpr int kPerLane = (kPiecesPerChunk + 63) / 64; // pieces each lane moves per chunk
    static_assert(PIECES >= 1 && PIECES <= 10 && kPerLane <= 5, "record / chunk too large for the staging registers");
    static_assert(DEPTH >= 2 && DEPTH <= 4, "two to four chunks in flight");
    // the element count is wave-uniform by contract; telling the compiler makes the loop control scalar, so the two
    // exits below are real branches instead of exec-mask updates that funnel through one latch block
    const int N = __builtin_amdgcn_readfirstlane(N_);
    if (N <= 0)
        return;
    // named registers on purpose: an array captured by the lambda is not promoted out of scratch memory.
    // DEPTH register sets (a, b, c, d) = DEPTH chunks in flight while one more is being consumed from LDS.
#define CAPE_STAGE_PIECE(q) ((lane + 64 * (q)) < kPiecesPerChunk ? (lane + 64 * (q)) : kPiecesPerChunk - 1)
    // the record number of piece q of the chunk that starts at element c0_ (clamped to the last element), and its load
#define CAPE_STAGE_INDEX(q, c0_) \
    index(((c0_) + CAPE_STAGE_PIECE(q) / PIECES) < N ? ((c0_) + CAPE_STAGE_PIECE(q) / PIECES) : N - 1, CAPE_STAGE_PIECE(q) % PIECES)
#define CAPE_STAGE_LOAD(q, rec_) (*addr((rec_), CAPE_STAGE_PIECE(q) % PIECES))
#define CAPE_STAGE_KEEP(q, c0_) \
    (keep(((c0_) + CAPE_STAGE_PIECE(q) / PIECES) < N ? ((c0_) + CAPE_STAGE_PIECE(q) / PIECES) : N - 1) ? (1u << (q)) : 0u)
    double2 a0 = make_double2(0, 0), a1 = a0, a2 = a0, a3 = a0, a4 = a0;
    double2 b0 = a0, b1 = a0, b2 = a0, b3 = a0, b4 = a0;
    double2 g0 = a0, g1 = a0, g2 = a0, g3 = a0, g4 = a0;
    double2 d0 = a0, d1 = a0, d2 = a0, d3 = a0, d4 = a0;
    unsigned ka = 0, kb = 0, kg = 0, kd = 0; // bit q: piece q of the set belongs to a kept element
    // unconditional (clamped) on purpose, see above; at most two surplus chunks are fetched at the end of a call.
    // The scheduling barriers keep the loads of one set together: the in-order vmcnt counter can only wait for "all
    // but the k youngest", so interleaving the two sets would make every hand-over wait for both.
    // index() usually reads a cell list in LDS: all look-ups of a set are requested before the first address is formed --
    // written per piece, every load waited for its own LDS round trip (three exposed LDS latencies per chunk)
#define CAPE_STAGE_ISSUE(S, at_)                                            \
    do                                                                      \
    {                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                  \
        const int r0_ = CAPE_STAGE_INDEX(0, at_);                           \
        const int r1_ = kPerLane > 1 ? CAPE_STAGE_INDEX(1, at_) : 0;        \
        const int r2_ = kPerLane > 2 ? CAPE_STAGE_INDEX(2, at_) : 0;        \
        const int r3_ = kPerLane > 3 ? CAPE_STAGE_INDEX(3, at_) : 0;        \
        const int r4_ = kPerLane > 4 ? CAPE_STAGE_INDEX(4, at_) : 0;        \
        k##S = CAPE_STAGE_KEEP(0, at_) | (kPerLane > 1 ? CAPE_STAGE_KEEP(1, at_) : 0u) |                        \
               (kPerLane > 2 ? CAPE_STAGE_KEEP(2, at_) : 0u) | (kPerLane > 3 ? CAPE_STAGE_KEEP(3, at_) : 0u) |  \
               (kPerLane > 4 ? CAPE_STAGE_KEEP(4, at_) : 0u);                                                 \
        __builtin_amdgcn_sched_barrier(0);                                  \
        S##0 = CAPE_STAGE_LOAD(0, r0_);                                     \
        if (kPerLane > 1)                                                   \
            S##1 = CAPE_STAGE_LOAD(1, r1_);                                 \
        if (kPerLane > 2)                                                   \
            S##2 = CAPE_STAGE_LOAD(2, r2_);                                 \
        if (kPerLane > 3)                                                   \
            S##3 = CAPE_STAGE_LOAD(3, r3_);                                 \
        if (kPerLane > 4)                                                   \
            S##4 = CAPE_STAGE_LOAD(4, r4_);                                 \
        __builtin_amdgcn_sched_barrier(0);                                  \
    } while (0)
#define CAPE_STAGE_STORE1(S, q) park(s_buf, CAPE_STAGE_PIECE(q), ((k##S >> (q)) & 1u) ? S##q : make_double2(0.0, 0.0))
#define CAPE_STAGE_STORE(S)                   \
    do                                        \
    {                                         \
        CAPE_STAGE_STORE1(S, 0);              \
        if (kPerLane > 1)                     \
            CAPE_STAGE_STORE1(S, 1);          \
        if (kPerLane > 2)                     \
            CAPE_STAGE_STORE1(S, 2);          \
        if (kPerLane > 3)                     \
            CAPE_STAGE_STORE1(S, 3);          \
        if (kPerLane > 4)                     \
            CAPE_STAGE_STORE1(S, 4);          \
    } while (0)
    auto consume = [&](int c0) {
        const int cn = (N - c0 < kStageChunk) ? (N - c0) : kStageChunk;
        CAPE_STAGE_FENCE();
        if (cn > 0)
            prep(c0, cn);
        if (cn == kStageChunk)
        {
            // groups of kGroup elements: the LDS reads of group g + 1 are all requested (scheduling barrier) before the
            // dependent folds of group g run, so one LDS round trip overlaps kGroup chain steps
            constexpr int kGroup = 8;
            constexpr int kGroups = kStageChunk / kGroup;
            static_assert(kStageChunk % (2 * kGroup) == 0, "chunk = even number of groups");
            using Value = decltype(load(0, s_buf));
            Value va[kGroup], vb[kGroup];
#pragma unroll
            for (int u = 0; u < kGroup; ++u)
                va[u] = load(c0 + u, s_buf + u * 2 * PIECES);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < kGroups; g += 2)
            {
#pragma unroll
                for (int u = 0; u < kGroup; ++u)
                    vb[u] = load(c0 + (g + 1) * kGroup + u, s_buf + ((g + 1) * kGroup + u) * 2 * PIECES);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < kGroup; ++u)
                    fold(c0 + g * kGroup + u, va[u]);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < kGroups)
                {
#pragma unroll
                    for (int u = 0; u < kGroup; ++u)
                        va[u] = load(c0 + (g + 2) * kGroup + u, s_buf + ((g + 2) * kGroup + u) * 2 * PIECES);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int u = 0; u < kGroup; ++u)
                    fold(c0 + (g + 1) * kGroup + u, vb[u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        else
        {
            int ci = 0;
            for (; ci + 4 <= cn; ci += 4)
            {
                decltype(load(0, s_buf)) v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    v[u] = load(c0 + ci + u, s_buf + (ci + u) * 2 * PIECES);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    fold(c0 + ci + u, v[u]);
            }
            for (; ci < cn; ++ci)
                fold(c0 + ci, load(c0 + ci, s_buf + ci * 2 * PIECES));
        }
        CAPE_STAGE_FENCE();
    };

    __builtin_amdgcn_s_waitcnt(0);
    CAPE_STAGE_ISSUE(a, 0);
    CAPE_STAGE_ISSUE(b, kStageChunk);
    if (DEPTH >= 3)
        CAPE_STAGE_ISSUE(g, 2 * kStageChunk);
    if (DEPTH >= 4)
        CAPE_STAGE_ISSUE(d, 3 * kStageChunk);
    // one exit only, and no branch around the later sets (consume() does nothing for an empty chunk): every path to the
    // top of the loop carries the sets in the order a, b, g, d, which is what the wait counts are derived from
#ifdef CAPE_B_PROFILE
#define CAPE_STAGE_T(k)                                                   \
    if (prof)                                                             \
    {                                                                     \
        __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) only */         \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();       \
        if (lane == 0)                                                    \
            atomicAdd(&prof[k], n_ - pt_);                                \
        pt_ = n_;                                                         \
    }
    unsigned long long pt_ = __builtin_amdgcn_s_memtime();
#else
#define CAPE_STAGE_T(k)
#endif
    for (int c0 = 0; c0 < N; c0 += DEPTH * kStageChunk)
    {
        CAPE_STAGE_T(23)
        CAPE_STAGE_STORE(a);
        CAPE_STAGE_T(30)
        CAPE_STAGE_ISSUE(a, c0 + DEPTH * kStageChunk);
        CAPE_STAGE_T(31)
        consume(c0);
        CAPE_STAGE_STORE(b);
        CAPE_STAGE_ISSUE(b, c0 + (DEPTH + 1) * kStageChunk);
        consume(c0 + kStageChunk);
        if (DEPTH >= 3)
        {
            CAPE_STAGE_STORE(g);
            CAPE_STAGE_ISSUE(g, c0 + (DEPTH + 2) * kStageChunk);
            consume(c0 + 2 * kStageChunk);
        }
        if (DEPTH >= 4)
        {
            CAPE_STAGE_STORE(d);
            CAPE_STAGE_ISSUE(d, c0 + (DEPTH + 3) * kStageChunk);
            consume(c0 + 3 * kStageChunk);
        }
    }
#undef CAPE_STAGE_T
#undef CAPE_STAGE_PIECE
#undef CAPE_STAGE_LOAD
#undef CAPE_STAGE_INDEX
#undef CAPE_STAGE_ISSUE
#undef CAPE_STAGE_STORE
#undef CAPE_STAGE_STORE1
#undef CAPE_STAGE_KEEP
}

// all pieces of a record from one array: piece `sub` of record r = base + r * strideDoubles + 2 * (firstPiece + sub)
template <int PIECES, int DEPTH, typename IndexFn, typename Prep, typename Load, typename Fold>
__device__ __forceinline__ void staged_for_each(int N, const double* base, int strideDoubles, int firstPiece, IndexFn index,
                                                double* s_buf, int lane, Prep prep, Load load, Fold fold,
                                                unsigned long long* prof = nullptr)
{
    staged_for_each_src<PIECES, DEPTH>(
            N, [&](int e, int) { return index(e); },
            [&](int rec, int sub) { return reinterpret_cast<const double2*>(base + (size_t)rec * strideDoubles + 2 * (firstPiece + sub)); },
            [](int) { return true; }, stage_park_records<PIECES>, s_buf, lane, prep, load, fold, prof);
}

template <int PIECES, int DEPTH, typename IndexFn, typename Load, typename Fold>
__device__ __forceinline__ void staged_for_each(int N, const double* base, int strideDoubles, int firstPiece, IndexFn index,
                                                double* s_buf, int lane, Load load, Fold fold)
{
    staged_for_each<PIECES, DEPTH>(N, base, strideDoubles, firstPiece, index, s_buf, lane, [](int, int) {}, load, fold);
}

} // namespace cape
