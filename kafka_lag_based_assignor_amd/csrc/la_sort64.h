// la_sort64.h -- compare-exchange networks on packed 64-bit records, written at the instruction level.
//
// A record is one 64-bit unsigned word kept as two VGPRs (P64).  Everything here is the inner loop of
// the wave-tile kernel: the sort of a topic's partitions (Main.java:228-235) and, per greedy round, the
// sort of its consumer bins (the comparator of Main.java:243-261 minus the count, which the round
// structure removes).  hipcc's own code for `(o < mine) == keep_min ? o : mine` on uint64 is
// 2 v_mov_dpp + v_cmp_lt_u64 + s_xor + s_nop + 2 v_cndmask (and TWO 64-bit compares for an in-register
// exchange, because it canonicalises to umin/umax).  The forms below fold the lane move into the
// arithmetic (DPP source operands), use the borrow of a 64-bit subtract as the compare, and keep the
// mask in VCC:
//
//     v_sub_co_u32_dpp   t, vcc, lo, lo  <lane^J>        ; t = lo[lane^J] - lo
//     v_subb_co_u32_dpp  t, vcc, hi, hi, vcc <lane^J>    ; vcc = rec[lane^J] < rec
//     s_xor_b64          vcc, vcc, KEEP_MIN              ; vcc = "keep my own record"
//     v_cndmask_b32_dpp  lo, lo, lo, vcc <lane^J>        ; lo = vcc ? lo : lo[lane^J]
//     v_cndmask_b32_dpp  hi, hi, hi, vcc <lane^J>
//
// 4 full-rate VALU + 1 SALU per compare-exchange.  KEEP_MIN is a compile-time 64-bit lane mask
// (direction-free bitonic network: the lower index always keeps the smaller record, and "lower" is
// one bit of the lane id).
//
// Hazards (the compiler does not look inside an asm statement; rules from the CDNA4 guides, §5.7 / T21):
//   * a VALU write of a VGPR needs 2 wait states before a DPP or v_permlane*_swap read of it.
//     Inside a block the instruction order provides them; between blocks the callers' register
//     order does (see the notes at bitonic_sort_tile_p64), and blocks that may directly follow a write
//     of their DPP source take PAD = 1 (s_nop 0) or 2 (s_nop 1).
//   * VCC: v_sub_co -> v_subb_co is a carry chain (no wait); SALU write of VCC -> VALU read needs none.
//   * s_xor_b64 writes SCC: every statement that contains one lists "scc" as clobbered (the compiler keeps
//     s_cmp / s_cselect pairs live across asm statements otherwise -- found the hard way).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "la_device.h"

namespace la {

struct P64 {
    uint32_t lo, hi;
};

__device__ __forceinline__ P64 p64_from(uint64_t x) { P64 p; p.lo = (uint32_t)x; p.hi = (uint32_t)(x >> 32); return p; }
__device__ __forceinline__ uint64_t p64_value(const P64& p) { return ((uint64_t)p.hi << 32) | p.lo; }

// keep-min lane masks: lanes whose bit `J` of the lane id is clear
template <int J>
struct KeepMin {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "bad lane bit");
    static constexpr uint64_t value = J == 1 ? 0x5555555555555555ull : J == 2 ? 0x3333333333333333ull
                                    : J == 4 ? 0x0F0F0F0F0F0F0F0Full : J == 8 ? 0x00FF00FF00FF00FFull
                                    : J == 16 ? 0x0000FFFF0000FFFFull : 0x00000000FFFFFFFFull;
};

#define LA_DPP_TAIL " row_mask:0xf bank_mask:0xf"
// "keep my own record" from the borrow of partner - mine, two ways:
//   SALU form  s_xor_b64 vcc, vcc, KEEP_MIN (one issue slot).  Right for kernels with several wavefronts per SIMD (the
//              wave-tile kernels, the workgroup sorts): the ~17 cycles a VALU -> SALU -> VALU hop through VCC stalls its
//              wavefront are filled by the others.
//   VALU form  (VO = true) for a chain ONE wavefront runs alone (the block path's greedy rounds): records below 2^63, so
//              the borrow is the sign of the difference's high word T; T ^= (all-ones on keep-min lanes); vcc = T < 0.
//              Two slots, no hop (measured: 200 x 8 000 x 16, 99 -> 74 us of greedy rounds).
#define LA_FIX_S(K) "s_xor_b64 vcc, vcc, " K "\n\t"
#define LA_FIX_V(T, K) "v_xor_b32 " T ", " K ", " T "\n\t" "v_cmp_gt_i32 vcc, 0, " T "\n\t"

// all-ones on the lanes whose lane-id bit J is clear (the keep-min lanes of KeepMin<J>), as a VGPR
template <int J>
__device__ __forceinline__ uint32_t keep_vec() {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return (lane & (uint32_t)J) ? 0u : 0xFFFFFFFFu;
}

#define LA_PAD0 ""
#define LA_PAD1 "s_nop 0\n\t"
#define LA_PAD2 "s_nop 1\n\t"

// ---- (a) rec <-> the same register of lane^(pattern), single-DPP patterns -------------------------------
#define LA_SAME_ASM(PADSTR, CTRL, FIX, KC, KV)                                             \
    asm volatile(PADSTR                                                                    \
                 "v_sub_co_u32_dpp %2, vcc, %0, %0 " CTRL LA_DPP_TAIL "\n\t"                 \
                 "v_subb_co_u32_dpp %2, vcc, %1, %1, vcc " CTRL LA_DPP_TAIL "\n\t"           \
                 FIX                                                                       \
                 "v_cndmask_b32_dpp %0, %0, %0, vcc " CTRL LA_DPP_TAIL "\n\t"                \
                 "v_cndmask_b32_dpp %1, %1, %1, vcc " CTRL LA_DPP_TAIL                       \
                 : "+v"(r.lo), "+v"(r.hi), "=&v"(t)                                         \
                 : KC(KV)                                                                  \
                 : "vcc", "scc")
#define LA_SAME_VO(PADSTR, CTRL)                                                           \
    do {                                                                                   \
        if constexpr (VO) LA_SAME_ASM(PADSTR, CTRL, LA_FIX_V("%2", "%3"), "v", kvec);      \
        else LA_SAME_ASM(PADSTR, CTRL, LA_FIX_S("%3"), "s", keep);                         \
    } while (0)

#define LA_SAME_PADS(CTRL)                                  \
    do {                                                    \
        if constexpr (PAD == 0) LA_SAME_VO(LA_PAD0, CTRL);   \
        else if constexpr (PAD == 1) LA_SAME_VO(LA_PAD1, CTRL); \
        else LA_SAME_VO(LA_PAD2, CTRL);                      \
    } while (0)

// lane ^ 4: no single DPP control; half-mirror into temporaries, then quad reverse as the DPP source
#define LA_SAME_X4_ASM(PADSTR, FIX, KC, KV)                                                    \
    asm volatile(PADSTR                                                                        \
                 "v_mov_b32_dpp %2, %0 row_half_mirror" LA_DPP_TAIL "\n\t"                       \
                 "v_mov_b32_dpp %3, %1 row_half_mirror" LA_DPP_TAIL "\n\t"                       \
                 "s_nop 0\n\t"                                                                 \
                 "v_sub_co_u32_dpp %4, vcc, %2, %0 quad_perm:[3,2,1,0]" LA_DPP_TAIL "\n\t"       \
                 "v_subb_co_u32_dpp %4, vcc, %3, %1, vcc quad_perm:[3,2,1,0]" LA_DPP_TAIL "\n\t" \
                 FIX                                                                           \
                 "v_cndmask_b32_dpp %0, %2, %0, vcc quad_perm:[3,2,1,0]" LA_DPP_TAIL "\n\t"      \
                 "v_cndmask_b32_dpp %1, %3, %1, vcc quad_perm:[3,2,1,0]" LA_DPP_TAIL             \
                 : "+v"(r.lo), "+v"(r.hi), "=&v"(t), "=&v"(u), "=&v"(w)                          \
                 : KC(KV)                                                                      \
                 : "vcc", "scc")
#define LA_SAME_X4_VO(PADSTR)                                                              \
    do {                                                                                   \
        if constexpr (VO) LA_SAME_X4_ASM(PADSTR, LA_FIX_V("%4", "%5"), "v", kvec);         \
        else LA_SAME_X4_ASM(PADSTR, LA_FIX_S("%5"), "s", keep);                            \
    } while (0)

// lane ^ 16 / lane ^ 32: v_permlane*_swap gives both lanes of a pair (A, B) = (lower's, upper's)
// record; each keeps A or B.  keep-min lanes keep A iff A < B.
#define LA_SAME_SWAP_ASM(PADSTR, SWAP, FIX, KC, KV)          \
    asm volatile(PADSTR                                      \
                 "v_mov_b32 %2, %0\n\t"                      \
                 "v_mov_b32 %3, %1\n\t"                      \
                 "s_nop 1\n\t"                               \
                 SWAP " %0, %2\n\t"                          \
                 SWAP " %1, %3\n\t"                          \
                 "v_sub_co_u32 %4, vcc, %0, %2\n\t"          \
                 "v_subb_co_u32 %4, vcc, %1, %3, vcc\n\t"    \
                 FIX                                         \
                 "v_cndmask_b32 %0, %0, %2, vcc\n\t"         \
                 "v_cndmask_b32 %1, %1, %3, vcc"             \
                 : "+v"(r.lo), "+v"(r.hi), "=&v"(t), "=&v"(u), "=&v"(w) \
                 : KC(KV)                                    \
                 : "vcc", "scc")
#define LA_SAME_SWAP_VO(PADSTR, SWAP)                                                      \
    do {                                                                                   \
        if constexpr (VO) LA_SAME_SWAP_ASM(PADSTR, SWAP, LA_FIX_V("%4", "%5"), "v", kvec); \
        else LA_SAME_SWAP_ASM(PADSTR, SWAP, LA_FIX_S("%5"), "s", keep);                    \
    } while (0)

// rec <- min or max of (rec, record of lane ^ J), min where bit J of the lane id is clear
template <int J, int PAD, bool VO = false>
__device__ __forceinline__ void cmpx_same_xor(P64& r) {
    [[maybe_unused]] const uint64_t keep = KeepMin<J>::value;
    [[maybe_unused]] const uint32_t kvec = keep_vec<J>();
    uint32_t t;
    if constexpr (J == 1) LA_SAME_PADS("quad_perm:[1,0,3,2]");
    else if constexpr (J == 2) LA_SAME_PADS("quad_perm:[2,3,0,1]");
    else if constexpr (J == 8) LA_SAME_PADS("row_ror:8");
    else if constexpr (J == 4) {
        uint32_t u, w;
        if constexpr (PAD == 0) LA_SAME_X4_VO(LA_PAD0);
        else if constexpr (PAD == 1) LA_SAME_X4_VO(LA_PAD1);
        else LA_SAME_X4_VO(LA_PAD2);
    } else {
        uint32_t u, w;
        // after the swaps: %0/%1 = A (lower lane's record), %2/%3 = B; vcc = (A<B) ^ keepmin; 1 -> B
        if constexpr (J == 16) LA_SAME_SWAP_VO(LA_PAD2, "v_permlane16_swap_b32");
        else LA_SAME_SWAP_VO(LA_PAD2, "v_permlane32_swap_b32");
    }
}

// plain (already moved) select: r <- keep-min lanes (lane-id bit J clear) min(r, o), others max(r, o)
#define LA_SELECT_ASM(FIX, KC, KV)                               \
    asm volatile("v_sub_co_u32 %2, vcc, %3, %0\n\t"              \
                 "v_subb_co_u32 %2, vcc, %4, %1, vcc\n\t"        \
                 FIX                                             \
                 "v_cndmask_b32 %0, %3, %0, vcc\n\t"             \
                 "v_cndmask_b32 %1, %4, %1, vcc"                 \
                 : "+v"(r.lo), "+v"(r.hi), "=&v"(t)              \
                 : "v"(o.lo), "v"(o.hi), KC(KV)                  \
                 : "vcc", "scc")
template <int J, bool VO = false>
__device__ __forceinline__ void select_minmax(P64& r, const P64& o) {
    uint32_t t;
    if constexpr (VO) {
        const uint32_t kvec = keep_vec<J>();
        LA_SELECT_ASM(LA_FIX_V("%2", "%5"), "v", kvec);
    } else {
        const uint64_t keep = KeepMin<J>::value;
        LA_SELECT_ASM(LA_FIX_S("%5"), "s", keep);
    }
}

// rec <- min or max of (rec, record of lane ^ (M-1)), min where bit M/2 of the lane id is clear.
// Single-register form (one record per lane: the consumer bins).
template <int M, int PAD, bool VO = false>
__device__ __forceinline__ void cmpx_same_mirror(P64& r) {
    static_assert(M == 2 || M == 4 || M == 8 || M == 16 || M == 32 || M == 64, "bad mirror width");
    if constexpr (M == 2) {
        cmpx_same_xor<1, PAD, VO>(r);
    } else if constexpr (M == 4 || M == 8 || M == 16) {
        [[maybe_unused]] const uint64_t keep = KeepMin<M / 2>::value;
        [[maybe_unused]] const uint32_t kvec = keep_vec<M / 2>();
        uint32_t t;
        if constexpr (M == 4) LA_SAME_PADS("quad_perm:[3,2,1,0]");
        else if constexpr (M == 8) LA_SAME_PADS("row_half_mirror");
        else LA_SAME_PADS("row_mirror");
    } else {
        // lane ^ 31 = row_mirror then lane ^ 16; lane ^ 63 additionally lane ^ 32.  Rare (once per sort):
        // move first, then a plain (non-DPP) select.  The move is compiler code whose first instruction
        // reads r through DPP: r may have been written by the previous block's last instruction, and the
        // compiler pads only one wait state after an asm statement.
        asm volatile("s_nop 1" : "+v"(r.lo), "+v"(r.hi));
        P64 o;
        o.lo = shfl_mirror<M>(r.lo);
        o.hi = shfl_mirror<M>(r.hi);
        select_minmax<M / 2, VO>(r, o);
    }
}

// ---- (b) mirror step between registers: my r <-> partner's q and my q <-> partner's r ---------------------
#define LA_CROSS_ASM(PADSTR, CTRL, FIX, KC, KV)                                             \
    asm volatile(PADSTR                                                                     \
                 "v_sub_co_u32_dpp %2, vcc, %0, %5 " CTRL LA_DPP_TAIL "\n\t"                  \
                 "v_subb_co_u32_dpp %2, vcc, %1, %6, vcc " CTRL LA_DPP_TAIL "\n\t"            \
                 FIX                                                                        \
                 "v_cndmask_b32_dpp %3, %0, %5, vcc " CTRL LA_DPP_TAIL "\n\t"                 \
                 "v_cndmask_b32_dpp %4, %1, %6, vcc " CTRL LA_DPP_TAIL "\n\t"                 \
                 "v_sub_co_u32_dpp %2, vcc, %5, %0 " CTRL LA_DPP_TAIL "\n\t"                  \
                 "v_subb_co_u32_dpp %2, vcc, %6, %1, vcc " CTRL LA_DPP_TAIL "\n\t"            \
                 FIX                                                                        \
                 "v_cndmask_b32_dpp %0, %5, %0, vcc " CTRL LA_DPP_TAIL "\n\t"                 \
                 "v_cndmask_b32_dpp %1, %6, %1, vcc " CTRL LA_DPP_TAIL                        \
                 : "+v"(q.lo), "+v"(q.hi), "=&v"(t), "=&v"(n.lo), "=&v"(n.hi)                       \
                 : "v"(r.lo), "v"(r.hi), KC(KV)                                             \
                 : "vcc", "scc")
#define LA_CROSS_VO(PADSTR, CTRL)                                                          \
    do {                                                                                   \
        if constexpr (VO) LA_CROSS_ASM(PADSTR, CTRL, LA_FIX_V("%2", "%7"), "v", kvec);     \
        else LA_CROSS_ASM(PADSTR, CTRL, LA_FIX_S("%7"), "s", keep);                        \
    } while (0)

#define LA_CROSS_PADS(CTRL)                                   \
    do {                                                      \
        if constexpr (PAD == 0) LA_CROSS_VO(LA_PAD0, CTRL);    \
        else if constexpr (PAD == 1) LA_CROSS_VO(LA_PAD1, CTRL); \
        else LA_CROSS_VO(LA_PAD2, CTRL);                       \
    } while (0)

template <int M, int PAD, bool VO = false>
__device__ __forceinline__ void cmpx_cross_mirror(P64& r, P64& q) {
    static_assert(M == 2 || M == 4 || M == 8 || M == 16 || M == 32 || M == 64, "bad mirror width");
    [[maybe_unused]] const uint64_t keep = KeepMin<M / 2>::value;
    [[maybe_unused]] const uint32_t kvec = keep_vec<M / 2>();
    if constexpr (M <= 16) {
        uint32_t t;
        P64 n;
        if constexpr (M == 2) LA_CROSS_PADS("quad_perm:[1,0,3,2]");
        else if constexpr (M == 4) LA_CROSS_PADS("quad_perm:[3,2,1,0]");
        else if constexpr (M == 8) LA_CROSS_PADS("row_half_mirror");
        else LA_CROSS_PADS("row_mirror");
        r = n;
    } else {
        asm volatile("s_nop 1" : "+v"(r.lo), "+v"(r.hi), "+v"(q.lo), "+v"(q.hi));   // as in cmpx_same_mirror
        P64 oq, orr;
        oq.lo = shfl_mirror<M>(q.lo); oq.hi = shfl_mirror<M>(q.hi);
        orr.lo = shfl_mirror<M>(r.lo); orr.hi = shfl_mirror<M>(r.hi);
        select_minmax<M / 2, VO>(r, oq);
        select_minmax<M / 2, VO>(q, orr);
    }
}

// ---- (c) two registers of one lane: a <= b afterwards ---------------------------------------------------------
__device__ __forceinline__ void cmpx_regs_p64(P64& a, P64& b) {
    uint32_t t;
    P64 n;
    asm volatile("v_sub_co_u32 %2, vcc, %0, %5\n\t"             // b - a
                 "v_subb_co_u32 %2, vcc, %1, %6, vcc\n\t"       // vcc = b < a
                 "v_cndmask_b32 %3, %5, %0, vcc\n\t"            // n = vcc ? b : a   (min)
                 "v_cndmask_b32 %4, %6, %1, vcc\n\t"
                 "v_cndmask_b32 %0, %0, %5, vcc\n\t"            // b = vcc ? a : b   (max)
                 "v_cndmask_b32 %1, %1, %6, vcc"
                 : "+v"(b.lo), "+v"(b.hi), "=&v"(t), "=&v"(n.lo), "=&v"(n.hi)
                 : "v"(a.lo), "v"(a.hi)
                 : "vcc", "scc");
    a = n;
}

// ---- networks ---------------------------------------------------------------------------------------------------
// Element index i = gl*E + r (L lanes per group, E registers per lane); ascending on exit.
//
// Register order between blocks (the 2-wait-state DPP rule).  With E >= 4 a lane stage walks r = 0..E-1,
// so a register is touched again E-1 blocks later; an in-register stage is followed by the mirror stage
// of the next merge, whose first block must not DPP-read the register the last in-register block just
// wrote: the mirror walks pairs from the middle outwards ((E/2-1, E/2) first), and the in-register stage
// finishes on (E-2, E-1).  E <= 2 cannot be ordered that way and pads every DPP block.

template <int L, int E, int J, bool FIRST, bool VO = false>
__device__ __forceinline__ void clean_p64(P64 (&rec)[E]) {
    if constexpr (J >= 1) {
        if constexpr (J >= E) {
            constexpr int PADN = (E <= 2) ? 1 : 0;
#pragma unroll
            for (int r = 0; r < E; ++r) cmpx_same_xor<J / E, PADN, VO>(rec[r]);
        } else {
#pragma unroll
            for (int r = 0; r < E; ++r)
                if ((r & J) == 0) cmpx_regs_p64(rec[r], rec[r | J]);
        }
        clean_p64<L, E, J / 2, false, VO>(rec);
    }
}

template <int L, int E, int K, bool FIRST, bool VO = false>
__device__ __forceinline__ void merge_p64(P64 (&rec)[E]) {
    if constexpr (K <= L * E) {
        if constexpr (K <= E) {
#pragma unroll
            for (int r = 0; r < E; ++r)
                if ((r & (K >> 1)) == 0) cmpx_regs_p64(rec[r], rec[r ^ (K - 1)]);
        } else {
            constexpr int M = K / E;
            if constexpr (E == 1) {
                cmpx_same_mirror<M, FIRST ? 2 : 1, VO>(rec[0]);
            } else {
                constexpr int PADN = FIRST ? 2 : ((E <= 2) ? 1 : 0);
#pragma unroll
                for (int r = E / 2 - 1; r >= 0; --r) cmpx_cross_mirror<M, PADN, VO>(rec[r], rec[E - 1 - r]);
            }
        }
        clean_p64<L, E, K / 4, false, VO>(rec);
        merge_p64<L, E, K * 2, false, VO>(rec);
    }
}

template <int L, int E, bool VO = false>
__device__ __forceinline__ void bitonic_sort_tile_p64(P64 (&rec)[E]) {
    // the first DPP block may directly follow compiler-generated writes of its sources: PAD 2
    if constexpr (E == 1) merge_p64<L, E, 2, true, VO>(rec);
    else merge_p64<L, E, 2, false, VO>(rec);
}

// ---- bins that are in order already ------------------------------------------------------------------------------
// Between greedy rounds the bins are (ascending totals) + (descending lags): often still ascending (few consumers, a Zipf
// tail, equal lags).  `gl` = lane inside its group, one record per lane; lanes beyond the live bins hold equal all-ones
// sentinels, which compare as "in order".  Used by the wave-tile kernel, where a round's sort is 15 stages over 32 lanes
// and the check pays (-2 % VALU, -1..2.5 % time on the target).  The block path's one-wavefront greedy was measured with
// the same check plus one or two odd-even transposition steps for nearly ordered bins, with and without a back-off for
// topics that never have the property: slower in every case (200 x 8 000 x 16 Zipf: 0.236 against 0.214 ms; random lags:
// 0.238 / 0.216) -- the instruction-level networks are cheaper than compiler-scheduled DPP shifts, 64-bit compares and
// branches.  Dropped there.

// the record of lane - 1 (wave shift: crosses the 16-lane rows; lane 0 reads zero)
__device__ __forceinline__ uint64_t p64_of_prev_lane(const P64& r) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r.lo, 0x138, 0xF, 0xF, false);   // wave_shr:1
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r.hi, 0x138, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}

// every group of the wavefront ascending already?  (wavefront-uniform answer)
__device__ __forceinline__ bool lanes_in_order_p64(const P64& r, int gl) {
    // The shift is a cross-lane operation: EVERY lane executes it, before any lane-dependent condition.  (Written as
    // `gl > 0 && p64_of_prev_lane(r) > ...` the DPP moves run only in lanes with gl > 0, and a lane whose source -- the
    // first lane of a group -- is masked off reads `old` = 0: "in order" whatever the bins are.  Fifty-two parity tests
    // said so.)
    const uint64_t prev = p64_of_prev_lane(r);
    const bool bad = (gl > 0) & (prev > p64_value(r));
    return __builtin_amdgcn_ballot_w64(bad) == 0;
}

// One record per lane (consumer bins), ascending over each group of L lanes.
template <int L, bool VO = false>
__device__ __forceinline__ void bitonic_sort_lanes_p64(P64& rec) {
    P64 r1[1] = {rec};
    merge_p64<L, 1, 2, true, VO>(r1);
    rec = r1[0];
}

// ---- (d) one greedy round of a one-wavefront topic, as ONE statement ---------------------------------------------------
// The block path's greedy for up to 64 consumers is a chain of ceil(P / C) dependent rounds run by ONE wavefront on a CU that
// has nothing else to issue.  Measured on 200 topics x 8 000 partitions x 16 consumers (tools/block_probe.py with a
// -DLA_BLOCK_CLOCKS build): a wavefront issues one instruction every 4 cycles whatever its kind (s_nop and SALU
// included), a VALU result feeds the next VALU instruction without a bubble -- and a VALU -> SALU -> VALU hop through
// VCC (the s_xor_b64 of the steps above) stalls ~17 cycles.  The round took ~530 cycles: 10 steps x (6 issue slots + that
// stall) + ~55 slots of compiler-scheduled addressing, predication and loop control.  Here a whole round is one asm
// statement: bins += (the round's lags << idx_bits) after the sort of the bins, and
//   * the direction of a step stays in the VALU: bins and sentinels are below 2^63, so the borrow of partner - mine is the
//     sign of the difference's high word; that word XOR a per-lane constant (all-ones on keep-min lanes), compared with
//     zero, is the "keep my own record" mask -- two VALU instead of one SALU, and no hop;
//   * the wait states a DPP read needs after the previous step's v_cndmask do the round's own bookkeeping instead of
//     s_nop where there is any (next slot address, its clamp, the LDS read of the next round's add values);
//   * the add values come from LDS pre-shifted (8 B per sorted position, written by the whole workgroup), and the winner's
//     consumer position goes back into the low word of the slot just consumed -- member ranks and the global stores are
//     done by all wavefronts after the last round, coalesced;
//   * lanes without a live bin and slots past the topic are not predicated: their slot address clamps to a slot that
//     holds zero (v_min_u32), their winner word is masked to zero.
// Slots: byte addresses in LDS.  `sb` = this lane's slot for the NEXT round (advanced by `stride` here), `ab_cur` = the
// clamped slot of THIS round (winner goes there), `ab_next` <- the clamped slot of the next round, `nxt` <- its add value
// (complete when the statement ends: s_waitcnt inside), `cur` = this round's add value.  kv[0..3]: all-ones on the lanes
// whose lane-id bit 1 / 2 / 4 / 8 is clear (round_keep_vectors).
#define LA_R_STEP(CTRL, KEEP)                                                         \
    "v_sub_co_u32_dpp %[t], vcc, %[lo], %[lo] " CTRL LA_DPP_TAIL "\n\t"               \
    "v_subb_co_u32_dpp %[t], vcc, %[hi], %[hi], vcc " CTRL LA_DPP_TAIL "\n\t"         \
    "v_xor_b32 %[t], %[" KEEP "], %[t]\n\t"                                          \
    "v_cmp_gt_i32 vcc, 0, %[t]\n\t"                                                  \
    "v_cndmask_b32_dpp %[lo], %[lo], %[lo], vcc " CTRL LA_DPP_TAIL "\n\t"             \
    "v_cndmask_b32_dpp %[hi], %[hi], %[hi], vcc " CTRL LA_DPP_TAIL "\n\t"
#define LA_R_STEP_X4(KEEP)                                                                     \
    "v_mov_b32_dpp %[u], %[lo] row_half_mirror" LA_DPP_TAIL "\n\t"                               \
    "v_mov_b32_dpp %[w], %[hi] row_half_mirror" LA_DPP_TAIL "\n\t"                               \
    "s_nop 0\n\t"                                                                              \
    "v_sub_co_u32_dpp %[t], vcc, %[u], %[lo] quad_perm:[3,2,1,0]" LA_DPP_TAIL "\n\t"             \
    "v_subb_co_u32_dpp %[t], vcc, %[w], %[hi], vcc quad_perm:[3,2,1,0]" LA_DPP_TAIL "\n\t"       \
    "v_xor_b32 %[t], %[" KEEP "], %[t]\n\t"                                                   \
    "v_cmp_gt_i32 vcc, 0, %[t]\n\t"                                                           \
    "v_cndmask_b32_dpp %[lo], %[u], %[lo], vcc quad_perm:[3,2,1,0]" LA_DPP_TAIL "\n\t"           \
    "v_cndmask_b32_dpp %[hi], %[w], %[hi], vcc quad_perm:[3,2,1,0]" LA_DPP_TAIL "\n\t"
#define LA_R_X1 "quad_perm:[1,0,3,2]"
#define LA_R_X2 "quad_perm:[2,3,0,1]"
#define LA_R_M4 "quad_perm:[3,2,1,0]"
#define LA_R_NOP "s_nop 0\n\t"
// the bookkeeping that fills the first three wait-state slots
#define LA_R_ADV "v_add_u32 %[sb], %[stride], %[sb]\n\t"
#define LA_R_CLAMP "v_min_u32 %[abn], %[zb], %[sb]\n\t"
#define LA_R_READ "ds_read_b64 %[nxt], %[abn]\n\t"
#define LA_R_TAIL                                            \
    "s_waitcnt lgkmcnt(0)\n\t"                               \
    "v_add_co_u32 %[lo], vcc, %[clo], %[lo]\n\t"             \
    "v_addc_co_u32 %[hi], vcc, %[chi], %[hi], vcc\n\t"       \
    "v_and_b32 %[won], %[mk], %[lo]\n\t"                     \
    "ds_write_b32 %[abc], %[won]"
// the sorting networks of merge_p64<L, 1, ...>, step by step (direction-free bitonic: mirror, then lane ^ j)
#define LA_R_NET2 LA_R_STEP(LA_R_X1, "k1")
#define LA_R_NET4_REST LA_R_STEP(LA_R_M4, "k2") LA_R_CLAMP LA_R_STEP(LA_R_X1, "k1")
#define LA_R_NET8_REST LA_R_STEP("row_half_mirror", "k4") LA_R_NOP LA_R_STEP(LA_R_X2, "k2") LA_R_NOP LA_R_STEP(LA_R_X1, "k1")
#define LA_R_NET16_REST \
    LA_R_STEP("row_mirror", "k8") LA_R_NOP LA_R_STEP_X4("k4") LA_R_NOP LA_R_STEP(LA_R_X2, "k2") LA_R_NOP LA_R_STEP(LA_R_X1, "k1")
#define LA_ROUND_ASM(BODY)                                                                                             \
    asm volatile(BODY LA_R_TAIL                                                                                        \
                 : [lo] "+v"(bin.lo), [hi] "+v"(bin.hi), [sb] "+v"(sb), [abn] "=&v"(ab_next), [nxt] "=&v"(nxt),         \
                   [t] "=&v"(t), [u] "=&v"(u), [w] "=&v"(w), [won] "=&v"(won)                                           \
                 : [abc] "v"(ab_cur), [clo] "v"((uint32_t)cur), [chi] "v"((uint32_t)(cur >> 32)), [stride] "s"(stride), \
                   [zb] "s"(zb), [mk] "v"(lane_mask), [k1] "v"(kv[0]), [k2] "v"(kv[1]), [k4] "v"(kv[2]), [k8] "v"(kv[3]) \
                 : "vcc", "memory")

constexpr uint64_t kRoundSentinel = 0x7FFFFFFFFFFFFFFFull;   // an idle lane's bin: above every real bin, below 2^63

__device__ __forceinline__ void round_keep_vectors(int lane, uint32_t (&kv)[4]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) kv[b] = (lane & (1 << b)) ? 0u : 0xFFFFFFFFu;
}

template <int L>
__device__ __forceinline__ void greedy_round_p64(P64& bin, uint32_t& sb, uint32_t ab_cur, uint32_t& ab_next, uint64_t cur,
                                                 uint64_t& nxt, uint32_t stride, uint32_t zb, uint32_t lane_mask,
                                                 const uint32_t (&kv)[4]) {
    static_assert(L == 2 || L == 4 || L == 8 || L == 16, "one statement per width up to a DPP row");
    uint32_t t, u, w, won;
    // wait states: a step ends on v_cndmask lo, v_cndmask hi; the next one DPP-reads lo first -> one more instruction between
    if constexpr (L == 2) LA_ROUND_ASM(LA_R_ADV LA_R_CLAMP LA_R_READ LA_R_NET2);
    else if constexpr (L == 4) LA_ROUND_ASM(LA_R_ADV LA_R_NET2 LA_R_CLAMP LA_R_STEP(LA_R_M4, "k2") LA_R_READ LA_R_STEP(LA_R_X1, "k1"));
    else if constexpr (L == 8) LA_ROUND_ASM(LA_R_NET2 LA_R_ADV LA_R_NET4_REST LA_R_READ LA_R_NET8_REST);
    else LA_ROUND_ASM(LA_R_NET2 LA_R_ADV LA_R_NET4_REST LA_R_READ LA_R_NET8_REST LA_R_NOP LA_R_NET16_REST);
}

}  // namespace la
