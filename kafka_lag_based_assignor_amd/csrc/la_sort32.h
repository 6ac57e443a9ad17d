// la_sort32.h -- the same direction-free bitonic networks as la_sort64.h, on 32-bit keys.
//
// Used by the wave-tile kernel's fast sort: a key is (top bits of the packed record) | (slot index), so
// the network moves one VGPR per record instead of two, and a compare-exchange is
//
//   across lanes (2 VALU):   v_mov_b32_dpp  t, x <lane^J>            ; the partner's key
//                            v_med3_u32     x, x, t, DIR            ; DIR = 0 on the lanes that keep the smaller key,
//                                                                   ; all ones on the others: med3(a, b, 0) = min(a, b),
//                                                                   ; med3(a, b, ~0) = max(a, b) -- min / max / select in one
//   inside a lane (2 VALU):  v_min_u32 / v_max_u32
//
// against 4 VALU + 1 SALU and 6 VALU for 64-bit records.  No VCC, no data-dependent masks.  DIR is one VGPR per lane
// distance (dir_vec<J>: bit J of the lane id, smeared), loop-invariant.  (Until round 3's second session a step was
// v_min_u32_dpp + v_max_u32_dpp + v_cndmask_b32 against an SGPR mask, 3 VALU; -DLA_SORT32_MED3=0 still builds that form.)
// Hazard rules and the register-order argument are those of la_sort64.h (2 wait states between a VALU
// write of a VGPR and a DPP / v_permlane*_swap read of it).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "la_device.h"
#include "la_sort64.h"

#ifndef LA_SORT32_MED3
#define LA_SORT32_MED3 1
#endif

namespace la {

// all ones on the lanes whose lane-id bit J is set (they keep the LARGER key of a pair), 0 on the others
template <int J>
__device__ __forceinline__ uint32_t dir_vec() {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return (lane & (uint32_t)J) ? 0xFFFFFFFFu : 0u;
}

#if LA_SORT32_MED3
// ---- 2-VALU forms ------------------------------------------------------------------------------------------------
#define LA32_SAME_ASM(PADSTR, CTRL)                                             \
    asm volatile(PADSTR                                                         \
                 "v_mov_b32_dpp %1, %0 " CTRL LA_DPP_TAIL "\n\t"                \
                 "v_med3_u32 %0, %0, %1, %2"                                    \
                 : "+v"(x), "=&v"(t1)                                           \
                 : "v"(dir))

#define LA32_SAME_PADS(CTRL)                                      \
    do {                                                          \
        if constexpr (PAD == 0) LA32_SAME_ASM(LA_PAD0, CTRL);      \
        else if constexpr (PAD == 1) LA32_SAME_ASM(LA_PAD1, CTRL); \
        else LA32_SAME_ASM(LA_PAD2, CTRL);                         \
    } while (0)

// x <- min or max of (x, lane^J's x); min where bit J of the lane id is clear
template <int J, int PAD>
__device__ __forceinline__ void cmpx32_same_xor(uint32_t& x) {
    const uint32_t dir = dir_vec<J>();
    uint32_t t1;
    if constexpr (J == 1) LA32_SAME_PADS("quad_perm:[1,0,3,2]");
    else if constexpr (J == 2) LA32_SAME_PADS("quad_perm:[2,3,0,1]");
    else if constexpr (J == 8) LA32_SAME_PADS("row_ror:8");
    else if constexpr (J == 4) {
        // lane ^ 4 = half-mirror, then quad reverse
        uint32_t h;
        asm volatile(LA_PAD2
                     "v_mov_b32_dpp %2, %0 row_half_mirror" LA_DPP_TAIL "\n\t"
                     "s_nop 1\n\t"
                     "v_mov_b32_dpp %1, %2 quad_perm:[3,2,1,0]" LA_DPP_TAIL "\n\t"
                     "v_med3_u32 %0, %0, %1, %3"
                     : "+v"(x), "=&v"(t1), "=&v"(h)
                     : "v"(dir));
    } else {
        // lane ^ 16 / ^ 32: after the swap both lanes of a pair hold (A, B) = (lower's, upper's) key
        uint32_t b;
        if constexpr (J == 16)
            asm volatile(LA_PAD2 "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\t"
                         "v_med3_u32 %0, %0, %1, %2"
                         : "+v"(x), "=&v"(b) : "v"(dir));
        else
            asm volatile(LA_PAD2 "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\t"
                         "v_med3_u32 %0, %0, %1, %2"
                         : "+v"(x), "=&v"(b) : "v"(dir));
    }
}

// one record per lane, mirror step
template <int M, int PAD>
__device__ __forceinline__ void cmpx32_same_mirror(uint32_t& x) {
    if constexpr (M == 2) {
        cmpx32_same_xor<1, PAD>(x);
    } else if constexpr (M == 4 || M == 8 || M == 16) {
        const uint32_t dir = dir_vec<M / 2>();
        uint32_t t1;
        if constexpr (M == 4) LA32_SAME_PADS("quad_perm:[3,2,1,0]");
        else if constexpr (M == 8) LA32_SAME_PADS("row_half_mirror");
        else LA32_SAME_PADS("row_mirror");
    } else {
        asm volatile("s_nop 1" : "+v"(x));        // compiler code reads x through DPP next (see la_sort64.h)
        const uint32_t o = shfl_mirror<M>(x);
        const bool keep_min = (KeepMin<M / 2>::value >> __lane_id()) & 1;
        const uint32_t lo = o < x ? o : x, hi = o < x ? x : o;
        x = keep_min ? lo : hi;
        asm volatile("s_nop 1" : "+v"(x));        // ... and the next block reads x through DPP
    }
}

// mirror step between registers: my r <-> partner's q and my q <-> partner's r
#define LA32_CROSS_ASM(PADSTR, CTRL)                                            \
    asm volatile(PADSTR                                                         \
                 "v_mov_b32_dpp %2, %1 " CTRL LA_DPP_TAIL "\n\t"                \
                 "v_mov_b32_dpp %3, %0 " CTRL LA_DPP_TAIL "\n\t"                \
                 "v_med3_u32 %0, %0, %2, %4\n\t"                               \
                 "v_med3_u32 %1, %1, %3, %4"                                    \
                 : "+v"(r), "+v"(q), "=&v"(t1), "=&v"(t2)                        \
                 : "v"(dir))

#define LA32_CROSS_PADS(CTRL)                                      \
    do {                                                           \
        if constexpr (PAD == 0) LA32_CROSS_ASM(LA_PAD0, CTRL);      \
        else if constexpr (PAD == 1) LA32_CROSS_ASM(LA_PAD1, CTRL); \
        else LA32_CROSS_ASM(LA_PAD2, CTRL);                         \
    } while (0)

template <int M, int PAD>
__device__ __forceinline__ void cmpx32_cross_mirror(uint32_t& r, uint32_t& q) {
    if constexpr (M <= 16) {
        const uint32_t dir = dir_vec<M / 2>();
        uint32_t t1, t2;
        if constexpr (M == 2) LA32_CROSS_PADS("quad_perm:[1,0,3,2]");
        else if constexpr (M == 4) LA32_CROSS_PADS("quad_perm:[3,2,1,0]");
        else if constexpr (M == 8) LA32_CROSS_PADS("row_half_mirror");
        else LA32_CROSS_PADS("row_mirror");
    } else {
        const uint64_t keep = KeepMin<M / 2>::value;
        asm volatile("s_nop 1" : "+v"(r), "+v"(q));
        const uint32_t oq = shfl_mirror<M>(q), orr = shfl_mirror<M>(r);
        const bool keep_min = (keep >> __lane_id()) & 1;
        const uint32_t rl = oq < r ? oq : r, rh = oq < r ? r : oq;
        const uint32_t ql = orr < q ? orr : q, qh = orr < q ? q : orr;
        r = keep_min ? rl : rh;
        q = keep_min ? ql : qh;
        asm volatile("s_nop 1" : "+v"(r), "+v"(q));   // the next block may read either through DPP
    }
}

#else   // ---- the 3-VALU forms (v_min_dpp, v_max_dpp, v_cndmask against an SGPR lane mask) ----------------------------

#define LA32_SAME_ASM(PADSTR, CTRL)                                             \
    asm volatile(PADSTR                                                         \
                 "v_min_u32_dpp %1, %0, %0 " CTRL LA_DPP_TAIL "\n\t"              \
                 "v_max_u32_dpp %2, %0, %0 " CTRL LA_DPP_TAIL "\n\t"              \
                 "v_cndmask_b32_e64 %0, %2, %1, %3"                             \
                 : "+v"(x), "=&v"(t1), "=&v"(t2)                                 \
                 : "s"(keep))

#define LA32_SAME_PADS(CTRL)                                      \
    do {                                                          \
        if constexpr (PAD == 0) LA32_SAME_ASM(LA_PAD0, CTRL);      \
        else if constexpr (PAD == 1) LA32_SAME_ASM(LA_PAD1, CTRL); \
        else LA32_SAME_ASM(LA_PAD2, CTRL);                         \
    } while (0)

// x <- min or max of (x, lane^J's x); min where bit J of the lane id is clear
template <int J, int PAD>
__device__ __forceinline__ void cmpx32_same_xor(uint32_t& x) {
    const uint64_t keep = KeepMin<J>::value;
    uint32_t t1, t2;
    if constexpr (J == 1) LA32_SAME_PADS("quad_perm:[1,0,3,2]");
    else if constexpr (J == 2) LA32_SAME_PADS("quad_perm:[2,3,0,1]");
    else if constexpr (J == 8) LA32_SAME_PADS("row_ror:8");
    else if constexpr (J == 4) {
        // lane ^ 4 = half-mirror, then quad reverse as the DPP source of the min / max
        uint32_t h;
        asm volatile(LA_PAD2
                     "v_mov_b32_dpp %3, %0 row_half_mirror" LA_DPP_TAIL "\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %1, %3, %0 quad_perm:[3,2,1,0]" LA_DPP_TAIL "\n\t"
                     "v_max_u32_dpp %2, %3, %0 quad_perm:[3,2,1,0]" LA_DPP_TAIL "\n\t"
                     "v_cndmask_b32_e64 %0, %2, %1, %4"
                     : "+v"(x), "=&v"(t1), "=&v"(t2), "=&v"(h)
                     : "s"(keep));
    } else {
        // lane ^ 16 / ^ 32: after the swap both lanes of a pair hold (A, B) = (lower's, upper's) key
        uint32_t b;
        if constexpr (J == 16)
            asm volatile(LA_PAD2 "v_mov_b32 %3, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %3\n\t"
                         "v_min_u32 %1, %0, %3\n\tv_max_u32 %2, %0, %3\n\tv_cndmask_b32_e64 %0, %2, %1, %4"
                         : "+v"(x), "=&v"(t1), "=&v"(t2), "=&v"(b) : "s"(keep));
        else
            asm volatile(LA_PAD2 "v_mov_b32 %3, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %3\n\t"
                         "v_min_u32 %1, %0, %3\n\tv_max_u32 %2, %0, %3\n\tv_cndmask_b32_e64 %0, %2, %1, %4"
                         : "+v"(x), "=&v"(t1), "=&v"(t2), "=&v"(b) : "s"(keep));
    }
}

// one record per lane, mirror step
template <int M, int PAD>
__device__ __forceinline__ void cmpx32_same_mirror(uint32_t& x) {
    if constexpr (M == 2) {
        cmpx32_same_xor<1, PAD>(x);
    } else if constexpr (M == 4 || M == 8 || M == 16) {
        const uint64_t keep = KeepMin<M / 2>::value;
        uint32_t t1, t2;
        if constexpr (M == 4) LA32_SAME_PADS("quad_perm:[3,2,1,0]");
        else if constexpr (M == 8) LA32_SAME_PADS("row_half_mirror");
        else LA32_SAME_PADS("row_mirror");
    } else {
        asm volatile("s_nop 1" : "+v"(x));        // compiler code reads x through DPP next (see la_sort64.h)
        const uint32_t o = shfl_mirror<M>(x);
        const bool keep_min = (KeepMin<M / 2>::value >> __lane_id()) & 1;
        const uint32_t lo = o < x ? o : x, hi = o < x ? x : o;
        x = keep_min ? lo : hi;
        asm volatile("s_nop 1" : "+v"(x));        // ... and the next block reads x through DPP
    }
}

// mirror step between registers: my r <-> partner's q and my q <-> partner's r
#define LA32_CROSS_ASM(PADSTR, CTRL)                                            \
    asm volatile(PADSTR                                                         \
                 "v_min_u32_dpp %2, %1, %0 " CTRL LA_DPP_TAIL "\n\t"              \
                 "v_max_u32_dpp %3, %1, %0 " CTRL LA_DPP_TAIL "\n\t"              \
                 "v_min_u32_dpp %4, %0, %1 " CTRL LA_DPP_TAIL "\n\t"              \
                 "v_max_u32_dpp %5, %0, %1 " CTRL LA_DPP_TAIL "\n\t"              \
                 "v_cndmask_b32_e64 %0, %3, %2, %6\n\t"                         \
                 "v_cndmask_b32_e64 %1, %5, %4, %6"                             \
                 : "+v"(r), "+v"(q), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4)  \
                 : "s"(keep))

#define LA32_CROSS_PADS(CTRL)                                      \
    do {                                                           \
        if constexpr (PAD == 0) LA32_CROSS_ASM(LA_PAD0, CTRL);      \
        else if constexpr (PAD == 1) LA32_CROSS_ASM(LA_PAD1, CTRL); \
        else LA32_CROSS_ASM(LA_PAD2, CTRL);                         \
    } while (0)

template <int M, int PAD>
__device__ __forceinline__ void cmpx32_cross_mirror(uint32_t& r, uint32_t& q) {
    const uint64_t keep = KeepMin<M / 2>::value;
    if constexpr (M <= 16) {
        uint32_t t1, t2, t3, t4;
        if constexpr (M == 2) LA32_CROSS_PADS("quad_perm:[1,0,3,2]");
        else if constexpr (M == 4) LA32_CROSS_PADS("quad_perm:[3,2,1,0]");
        else if constexpr (M == 8) LA32_CROSS_PADS("row_half_mirror");
        else LA32_CROSS_PADS("row_mirror");
    } else {
        asm volatile("s_nop 1" : "+v"(r), "+v"(q));
        const uint32_t oq = shfl_mirror<M>(q), orr = shfl_mirror<M>(r);
        const bool keep_min = (keep >> __lane_id()) & 1;
        const uint32_t rl = oq < r ? oq : r, rh = oq < r ? r : oq;
        const uint32_t ql = orr < q ? orr : q, qh = orr < q ? q : orr;
        r = keep_min ? rl : rh;
        q = keep_min ? ql : qh;
        asm volatile("s_nop 1" : "+v"(r), "+v"(q));   // the next block may read either through DPP
    }
}

#endif  // LA_SORT32_MED3

__device__ __forceinline__ void cmpx32_regs(uint32_t& a, uint32_t& b) {   // a <= b afterwards
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}

template <int L, int E, int J>
__device__ __forceinline__ void clean32(uint32_t (&key)[E]) {
    if constexpr (J >= 1) {
        if constexpr (J >= E) {
            constexpr int PADN = (E <= 2) ? 2 : 0;
#pragma unroll
            for (int r = 0; r < E; ++r) cmpx32_same_xor<J / E, PADN>(key[r]);
        } else {
#pragma unroll
            for (int r = 0; r < E; ++r)
                if ((r & J) == 0) cmpx32_regs(key[r], key[r | J]);
        }
        clean32<L, E, J / 2>(key);
    }
}

template <int L, int E, int K>
__device__ __forceinline__ void merge32(uint32_t (&key)[E]) {
    if constexpr (K <= L * E) {
        if constexpr (K <= E) {
#pragma unroll
            for (int r = 0; r < E; ++r)
                if ((r & (K >> 1)) == 0) cmpx32_regs(key[r], key[r ^ (K - 1)]);
        } else {
            constexpr int M = K / E;
            // in-register stages are compiler code: a lane stage that follows one pads (2 wait states)
            if constexpr (E == 1) {
                cmpx32_same_mirror<M, 2>(key[0]);
            } else {
#pragma unroll
                for (int r = E / 2 - 1; r >= 0; --r) cmpx32_cross_mirror<M, 2>(key[r], key[E - 1 - r]);
            }
        }
        clean32<L, E, K / 4>(key);
        merge32<L, E, K * 2>(key);
    }
}

// L lanes x E registers, element index i = gl*E + r, ascending on exit
template <int L, int E>
__device__ __forceinline__ void bitonic_sort_tile_u32(uint32_t (&key)[E]) {
    merge32<L, E, 2>(key);
}

}  // namespace la
