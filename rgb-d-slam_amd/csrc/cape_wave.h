// Cross-lane primitives of the one-wavefront-per-frame kernels, on the DPP data path of gfx950.
//
// __shfl_xor / __shfl_up compile to ds_bpermute_b32: a round trip through the LDS crossbar (~70 cycles, and the grow
// wave is latency bound: a 6-step butterfly on a 64-bit key was ~850 cycles of pure waiting, several times per seed).
// DPP operands move data between lanes inside the VALU (a v_mov with a lane pattern, ~8 cycles): quad permutes and row
// mirrors reduce 16 lanes in four steps, v_readlane joins the four rows; row_shr + row_bcast give the inclusive scan;
// wave_shr / wave_shl shift the whole wave by one lane (GFX9 DPP controls, present on gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cape {

constexpr int kDppQuadXor1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowMirror = 0x140;
constexpr int kDppRowShr = 0x110;       // + n
constexpr int kDppRowBcast15 = 0x142;
constexpr int kDppRowBcast31 = 0x143;
constexpr int kDppWaveShl1 = 0x130;     // lane i <- lane i + 1
constexpr int kDppWaveShr1 = 0x138;     // lane i <- lane i - 1

template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v)
{
    const unsigned lo = dpp_u32<CTRL>((unsigned)v), hi = dpp_u32<CTRL>((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// after the four steps every lane of a 16-lane row holds the row's reduction
#define CAPE_ROW_REDUCE(T, DPP, OP)                 \
    {                                               \
        T w_;                                       \
        w_ = DPP<kDppQuadXor1>(v);                  \
        v = OP(v, w_);                              \
        w_ = DPP<kDppQuadXor2>(v);                  \
        v = OP(v, w_);                              \
        w_ = DPP<kDppRowHalfMirror>(v);             \
        v = OP(v, w_);                              \
        w_ = DPP<kDppRowMirror>(v);                 \
        v = OP(v, w_);                              \
    }

__device__ __forceinline__ unsigned op_max_u32(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned op_or_u32(unsigned a, unsigned b) { return a | b; }
__device__ __forceinline__ unsigned op_add_u32(unsigned a, unsigned b) { return a + b; }
__device__ __forceinline__ unsigned long long op_min_u64(unsigned long long a, unsigned long long b) { return a < b ? a : b; }

__device__ __forceinline__ unsigned readlane_u32(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l)
{
    return ((unsigned long long)readlane_u32((unsigned)(v >> 32), l) << 32) | readlane_u32((unsigned)v, l);
}

// wave-wide reductions; the result is uniform (it comes out of v_readlane, i.e. it lives in scalar registers)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    CAPE_ROW_REDUCE(unsigned, dpp_u32, op_max_u32)
    return op_max_u32(op_max_u32(readlane_u32(v, 0), readlane_u32(v, 16)), op_max_u32(readlane_u32(v, 32), readlane_u32(v, 48)));
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v)
{
    CAPE_ROW_REDUCE(unsigned, dpp_u32, op_or_u32)
    return (readlane_u32(v, 0) | readlane_u32(v, 16)) | (readlane_u32(v, 32) | readlane_u32(v, 48));
}
__device__ __forceinline__ int wave_sum_i32(int vi)
{
    unsigned v = (unsigned)vi;
    CAPE_ROW_REDUCE(unsigned, dpp_u32, op_add_u32)
    return (int)((readlane_u32(v, 0) + readlane_u32(v, 16)) + (readlane_u32(v, 32) + readlane_u32(v, 48)));
}
__device__ __forceinline__ unsigned op_min_u32(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    CAPE_ROW_REDUCE(unsigned, dpp_u32, op_min_u32)
    return op_min_u32(op_min_u32(readlane_u32(v, 0), readlane_u32(v, 16)), op_min_u32(readlane_u32(v, 32), readlane_u32(v, 48)));
}
// lexicographic in (high word, low word): two 32-bit reductions (one v_min_u32 per DPP step) instead of one 64-bit one
// (a 64-bit compare and two selects per step)
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned mh = wave_min_u32(hi);
    const unsigned ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return ((unsigned long long)mh << 32) | ml;
}

// the same minimum for keys that nearly always differ in their high words (a squared distance's leading bits): when exactly one
// lane holds the smallest high word -- one ballot tells -- its low word is taken out with one v_readlane instead of a second
// reduction; ties take the second reduction.  Same result as wave_min_u64 for every input.
__device__ __forceinline__ unsigned long long wave_min_u64_lead(unsigned long long v)
{
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned mh = wave_min_u32(hi);
    const unsigned long long tie = __ballot(hi == mh);
    unsigned ml;
    if (__popcll(tie) == 1)
        ml = readlane_u32(lo, __ffsll((long long)tie) - 1);
    else
        ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return ((unsigned long long)mh << 32) | ml;
}

// ---- the 64 lanes' keys in ascending order, lane i <- the i-th smallest: a bitonic network of 21 compare-exchange steps.
// The partner lane ^ STRIDE comes through DPP (strides 1, 2), ds_swizzle (4, 8, 16: within each half) or ds_bpermute (32).
template <int STRIDE> __device__ __forceinline__ unsigned xor_lane_u32(unsigned v, int lane)
{
    if (STRIDE == 1)
        return dpp_u32<kDppQuadXor1>(v);
    if (STRIDE == 2)
        return dpp_u32<kDppQuadXor2>(v);
    if (STRIDE == 32)
        return (unsigned)__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, (int)v);
    // BITMODE swizzle: offset = and_mask | or_mask << 5 | xor_mask << 10 over the lane's five low bits
    return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (STRIDE << 10));
}
template <int SIZE, int STRIDE> __device__ __forceinline__ unsigned long long sort_step_u64(unsigned long long v, int lane)
{
    const unsigned olo = xor_lane_u32<STRIDE>((unsigned)v, lane), ohi = xor_lane_u32<STRIDE>((unsigned)(v >> 32), lane);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    // the lower lane of the pair keeps the minimum where the block of SIZE sorts upwards, the maximum where it sorts downwards
    const bool keepMin = ((lane & STRIDE) == 0) == ((lane & SIZE) == 0); // (SIZE = 64: every lane sorts upwards)
    return (keepMin == (o < v)) ? o : v;
}
__device__ __forceinline__ unsigned long long wave_sort_u64(unsigned long long v, int lane)
{
#define CAPE_SORT_STEP(SIZE, STRIDE) v = sort_step_u64<SIZE, STRIDE>(v, lane);
    CAPE_SORT_STEP(2, 1)
    CAPE_SORT_STEP(4, 2) CAPE_SORT_STEP(4, 1)
    CAPE_SORT_STEP(8, 4) CAPE_SORT_STEP(8, 2) CAPE_SORT_STEP(8, 1)
    CAPE_SORT_STEP(16, 8) CAPE_SORT_STEP(16, 4) CAPE_SORT_STEP(16, 2) CAPE_SORT_STEP(16, 1)
    CAPE_SORT_STEP(32, 16) CAPE_SORT_STEP(32, 8) CAPE_SORT_STEP(32, 4) CAPE_SORT_STEP(32, 2) CAPE_SORT_STEP(32, 1)
    CAPE_SORT_STEP(64, 32) CAPE_SORT_STEP(64, 16) CAPE_SORT_STEP(64, 8) CAPE_SORT_STEP(64, 4) CAPE_SORT_STEP(64, 2) CAPE_SORT_STEP(64, 1)
#undef CAPE_SORT_STEP
    return v;
}

// sum of the 64 lanes' doubles in TREE order (not the order of any reference loop: only for quantities whose rounding is
// not observable, e.g. a conservative bound); uniform result
__device__ __forceinline__ double op_add_f64_bits(unsigned long long a, unsigned long long b)
{
    return __longlong_as_double((long long)a) + __longlong_as_double((long long)b);
}
__device__ __forceinline__ double wave_sum_f64_tree(double x)
{
    unsigned long long v = (unsigned long long)__double_as_longlong(x);
#define CAPE_F64_STEP(CTRL) v = (unsigned long long)__double_as_longlong(op_add_f64_bits(v, dpp_u64<CTRL>(v)));
    CAPE_F64_STEP(kDppQuadXor1)
    CAPE_F64_STEP(kDppQuadXor2)
    CAPE_F64_STEP(kDppRowHalfMirror)
    CAPE_F64_STEP(kDppRowMirror)
#undef CAPE_F64_STEP
    return (op_add_f64_bits(readlane_u64(v, 0), readlane_u64(v, 16))) + (op_add_f64_bits(readlane_u64(v, 32), readlane_u64(v, 48)));
}
// the double held by lane l (l uniform): v_readlane instead of a trip through the LDS crossbar
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    return __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(v), l));
}

// inclusive prefix sum over the 64 lanes: Kogge-Stone inside each row of 16 (row_shr shifts zeros in), then the row
// totals travel down with row_bcast:15 (rows 1 and 3 take lane 15 of the row before) and row_bcast:31 (rows 2 and 3)
__device__ __forceinline__ int wave_scan_i32(int vi)
{
    int v = vi;
    v += __builtin_amdgcn_update_dpp(0, v, kDppRowShr + 1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, kDppRowShr + 2, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, kDppRowShr + 4, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, kDppRowShr + 8, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, kDppRowBcast15, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, kDppRowBcast31, 0xC, 0xF, false);
    return v;
}

// A grid row of up to 128 cells as two 64-bit words (round 6: 1920 x 1080 = 96 x 54 cells in the fast grow kernels): the operators
// the bit-row code uses on its 32- / 64-bit masks, with the carry between the words.
struct Mask128
{
    unsigned long long lo, hi;
    __device__ __forceinline__ Mask128() {}
    __device__ __forceinline__ Mask128(unsigned long long l, unsigned long long h = 0ull) : lo(l), hi(h) {}
    __device__ __forceinline__ Mask128 operator<<(int n) const
    {
        if (n == 0)
            return *this;
        if (n >= 64)
            return Mask128(0ull, n >= 128 ? 0ull : lo << (n - 64));
        return Mask128(lo << n, (hi << n) | (lo >> (64 - n)));
    }
    __device__ __forceinline__ Mask128 operator>>(int n) const
    {
        if (n == 0)
            return *this;
        if (n >= 64)
            return Mask128(n >= 128 ? 0ull : hi >> (n - 64), 0ull);
        return Mask128((lo >> n) | (hi << (64 - n)), hi >> n);
    }
    __device__ __forceinline__ Mask128 operator&(const Mask128& o) const { return Mask128(lo & o.lo, hi & o.hi); }
    __device__ __forceinline__ Mask128 operator|(const Mask128& o) const { return Mask128(lo | o.lo, hi | o.hi); }
    __device__ __forceinline__ Mask128 operator~() const { return Mask128(~lo, ~hi); }
    __device__ __forceinline__ Mask128& operator&=(const Mask128& o) { lo &= o.lo; hi &= o.hi; return *this; }
    __device__ __forceinline__ Mask128& operator|=(const Mask128& o) { lo |= o.lo; hi |= o.hi; return *this; }
    __device__ __forceinline__ bool operator==(const Mask128& o) const { return lo == o.lo && hi == o.hi; }
    __device__ __forceinline__ bool operator!=(const Mask128& o) const { return lo != o.lo || hi != o.hi; }
    __device__ __forceinline__ explicit operator bool() const { return (lo | hi) != 0ull; }
    __device__ __forceinline__ explicit operator int() const { return (int)lo; } // (ablation builds only)
};

// the value of the lane below / above (zero past the ends of the wave)
__device__ __forceinline__ unsigned wave_from_lane_below(unsigned v) { return dpp_u32<kDppWaveShr1>(v); }       // lane i <- i - 1
__device__ __forceinline__ unsigned wave_from_lane_above(unsigned v) { return dpp_u32<kDppWaveShl1>(v); }       // lane i <- i + 1
__device__ __forceinline__ unsigned long long wave_from_lane_below(unsigned long long v) { return dpp_u64<kDppWaveShr1>(v); }
__device__ __forceinline__ unsigned long long wave_from_lane_above(unsigned long long v) { return dpp_u64<kDppWaveShl1>(v); }
__device__ __forceinline__ Mask128 wave_from_lane_below(const Mask128& v) { return Mask128(dpp_u64<kDppWaveShr1>(v.lo), dpp_u64<kDppWaveShr1>(v.hi)); }
__device__ __forceinline__ Mask128 wave_from_lane_above(const Mask128& v) { return Mask128(dpp_u64<kDppWaveShl1>(v.lo), dpp_u64<kDppWaveShl1>(v.hi)); }
__device__ __forceinline__ double wave_from_lane_below(double v)
{
    return __longlong_as_double((long long)dpp_u64<kDppWaveShr1>((unsigned long long)__double_as_longlong(v)));
}

} // namespace cape
