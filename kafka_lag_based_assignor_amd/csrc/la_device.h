// la_device.h -- CDNA4 (gfx950) device primitives shared by the assignor kernels.
//
// Everything on the hot path is integer compare/select work on 96-bit records:
//
//   Rec = (hi:lo = 64-bit unsigned key, tb = 32-bit unsigned tie-break), ascending.
//
//   partitions : key = (uint64)lag ^ 0x7FFF'FFFF'FFFF'FFFF   -> ascending == lag DESCENDING
//                                                               under Java's signed compare
//                tb  = (uint32)partition ^ 0x8000'0000        -> ascending == id ascending
//                (the sort comparator of Main.java:228-235)
//   consumers  : key = (uint64)totalLag + 2^63 (biased)       -> ascending == signed ascending
//                tb  = position in the topic's rank-sorted consumer list
//                (comparator levels 2 and 3 of Main.java:253-259; level 1, the count,
//                 is what the round structure removes -- see la_wave_tile.hip)
//
// Cross-lane movement uses DPP / v_permlane*_swap (VALU-rate) instead of ds_bpermute
// (LDS-pipe rate, shared by the CU's four SIMDs).  Wavefront = 64 lanes, always.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace la {

constexpr int kWave = 64;

struct Rec {
    uint32_t hi, lo, tb;
};

__device__ __forceinline__ bool rec_less(const Rec& a, const Rec& b) {
    const uint64_t ka = ((uint64_t)a.hi << 32) | a.lo;
    const uint64_t kb = ((uint64_t)b.hi << 32) | b.lo;
    return (ka < kb) | ((ka == kb) & (a.tb < b.tb));
}

// ---- lag arithmetic: computePartitionLag, Main.java:376-404 -----------------------
// committed < 0 == "no committed offset"; wrapping subtract, signed max with 0.
__device__ __forceinline__ int64_t partition_lag(int64_t begin, int64_t end, int64_t committed,
                                                 bool reset_latest) {
    const int64_t next = committed >= 0 ? committed : (reset_latest ? end : begin);
    const int64_t d = (int64_t)((uint64_t)end - (uint64_t)next);
    return d > 0 ? d : 0;
}

constexpr uint64_t kLagKeyFlip = 0x7FFFFFFFFFFFFFFFull;   // signed-desc -> unsigned-asc
constexpr uint64_t kTotalBias = 0x8000000000000000ull;    // signed-asc  -> unsigned-asc
constexpr uint32_t kPidBias = 0x80000000u;

// ---- lane ^ J shuffles ---------------------------------------------------------------
// J in {1,2,4,8}: DPP inside a row of 16 lanes.  J = 16 / 32: gfx950 v_permlane{16,32}_swap.
template <int J>
__device__ __forceinline__ uint32_t shfl_xor(uint32_t x) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "bad xor distance");
    if constexpr (J == 1) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    } else if constexpr (J == 2) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    } else if constexpr (J == 4) {
        // banks 0,2 (lanes 0-3, 8-11) read lane+4; banks 1,3 read lane-4
        int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xF, 0x5, false);    // row_shl:4
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xF, 0xA, false);  // row_shr:4
    } else if constexpr (J == 8) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, false);  // row_ror:8
    } else if constexpr (J == 16) {
        // swap(vdst rows 1,3 <-> src rows 0,2): r[0] = {x.r0,x.r0,x.r2,x.r2}, r[1] = {x.r1,x.r1,x.r3,x.r3}
        auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (__lane_id() & 16) ? r[0] : r[1];
    } else {
        auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return (__lane_id() & 32) ? r[0] : r[1];
    }
}

template <int J>
__device__ __forceinline__ Rec shfl_xor(const Rec& r) {
    Rec o;
    o.hi = shfl_xor<J>(r.hi);
    o.lo = shfl_xor<J>(r.lo);
    o.tb = shfl_xor<J>(r.tb);
    return o;
}

// wave-uniform runtime distance
__device__ __forceinline__ Rec shfl_xor_dyn(const Rec& r, int j) {
    switch (j) {
        case 1: return shfl_xor<1>(r);
        case 2: return shfl_xor<2>(r);
        case 4: return shfl_xor<4>(r);
        case 8: return shfl_xor<8>(r);
        case 16: return shfl_xor<16>(r);
        default: return shfl_xor<32>(r);
    }
}

// One compare-exchange of `mine` with the record held by lane^J.  keep_min lanes end up
// with the smaller record.  Equal records carry identical bits, so "take the other one"
// on equality is harmless and one compare serves both directions.
template <int J>
__device__ __forceinline__ void cmpx_lanes(Rec& mine, bool keep_min) {
    const Rec o = shfl_xor<J>(mine);
    const bool take = (rec_less(o, mine) == keep_min);
    mine.hi = take ? o.hi : mine.hi;
    mine.lo = take ? o.lo : mine.lo;
    mine.tb = take ? o.tb : mine.tb;
}

__device__ __forceinline__ void cmpx_lanes_dyn(Rec& mine, int j, bool keep_min) {
    const Rec o = shfl_xor_dyn(mine, j);
    const bool take = (rec_less(o, mine) == keep_min);
    mine.hi = take ? o.hi : mine.hi;
    mine.lo = take ? o.lo : mine.lo;
    mine.tb = take ? o.tb : mine.tb;
}

// In-register compare-exchange: afterwards a <= b when asc, a >= b otherwise.
__device__ __forceinline__ void cmpx_regs(Rec& a, Rec& b, bool asc) {
    const bool sw = (rec_less(b, a) == asc);
    const Rec ta = a, tb_ = b;
    a.hi = sw ? tb_.hi : ta.hi;  a.lo = sw ? tb_.lo : ta.lo;  a.tb = sw ? tb_.tb : ta.tb;
    b.hi = sw ? ta.hi : tb_.hi;  b.lo = sw ? ta.lo : tb_.lo;  b.tb = sw ? ta.tb : tb_.tb;
}

// ---- bitonic sort of L*E records: L lanes (a sub-wave group), E registers per lane ----
// Element index i = lane_in_group * E + r  (blocked), so the frequent small strides stay
// in registers.  Sorted ascending on exit: position s lives in lane s / E, register s % E.
template <int L, int E, int K, int J>
__device__ __forceinline__ void bitonic_stage(Rec (&rec)[E], int gl) {
    constexpr int N = L * E;
    if constexpr (J >= E) {
        constexpr int JL = J / E;                       // lane distance
        const bool lower = (gl & JL) == 0;
        const bool asc = (K == N) ? true : ((gl & (K / E)) == 0);
        const bool keep_min = (lower == asc);
#pragma unroll
        for (int r = 0; r < E; ++r) cmpx_lanes<JL>(rec[r], keep_min);
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            if ((r & J) == 0) {
                bool asc;
                if constexpr (K == N) asc = true;
                else if constexpr (K < E) asc = ((r & K) == 0);
                else asc = ((gl & (K / E)) == 0);
                cmpx_regs(rec[r], rec[r | J], asc);
            }
        }
    }
}

template <int L, int E, int K, int J>
__device__ __forceinline__ void bitonic_merge(Rec (&rec)[E], int gl) {
    bitonic_stage<L, E, K, J>(rec, gl);
    if constexpr (J > 1) bitonic_merge<L, E, K, J / 2>(rec, gl);
}

template <int L, int E, int K = 2>
__device__ __forceinline__ void bitonic_sort_tile(Rec (&rec)[E], int gl) {
    if constexpr (L * E >= 2) {
        bitonic_merge<L, E, K, K / 2>(rec, gl);
        if constexpr (K < L * E) bitonic_sort_tile<L, E, K * 2>(rec, gl);
    }
}

// One record per lane, sorted ascending over the first n lanes of each group
// (n = wave-uniform power of two, n <= group width).
__device__ __forceinline__ void bitonic_sort_lanes(Rec& rec, int gl, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool keep_min = (((gl & j) == 0) == ((gl & k) == 0));
            cmpx_lanes_dyn(rec, j, keep_min);
        }
    }
}

// ---- lane ^ (M-1) moves ------------------------------------------------------------------------
// The networks of la_sort32.h / la_sort64.h are the direction-free form of the bitonic sorter: the first
// step of a merge of size K pairs element i with i ^ (K-1) (mirror), every later step pairs i with i ^ j,
// and the lower index always keeps the smaller record.  quad_perm / row_half_mirror / row_mirror are
// single DPP moves; wider mirrors add v_permlane*_swap steps.

__device__ __forceinline__ uint32_t dpp_row_mirror(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xF, 0xF, false);
}

template <int M>
__device__ __forceinline__ uint32_t shfl_mirror(uint32_t x) {
    static_assert(M == 2 || M == 4 || M == 8 || M == 16 || M == 32 || M == 64, "bad mirror width");
    if constexpr (M == 2) return shfl_xor<1>(x);
    else if constexpr (M == 4) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x1B, 0xF, 0xF, false);   // quad_perm [3,2,1,0]
    else if constexpr (M == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, false);  // row_half_mirror
    else if constexpr (M == 16) return dpp_row_mirror(x);
    else if constexpr (M == 32) return shfl_xor<16>(dpp_row_mirror(x));
    else return shfl_xor<32>(shfl_xor<16>(dpp_row_mirror(x)));
}

// OR over the whole wavefront, result in every lane (idempotent, so a butterfly is enough).
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    v |= shfl_xor<1>(v);
    v |= shfl_xor<2>(v);
    v |= shfl_xor<4>(v);
    v |= shfl_xor<8>(v);
    v |= shfl_xor<16>(v);
    v |= shfl_xor<32>(v);
    return v;
}

// LDS accesses made by one wave are executed in order; these only stop the compiler from
// moving them across the point where lanes exchange data through LDS.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is a workgroup-scope fence + s_barrier, and
// the fence makes every wavefront wait for ALL its outstanding memory operations (s_waitcnt vmcnt(0)) -- including global
// loads issued only to be used much later (the next round's lags) and fire-and-forget global stores.  Here only the
// LDS traffic must have landed: lgkmcnt(0), then the barrier; global accesses stay in flight across it.  The "memory"
// clobber keeps the compiler from moving LDS accesses over the barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// inclusive scan over the wavefront through DPP (row shifts, then the row broadcasts of GFX9)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}

}  // namespace la
