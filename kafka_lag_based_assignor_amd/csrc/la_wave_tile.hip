// la_wave_tile.hip -- fused lag + sort + greedy for topics that fit one sub-wave tile.
//
// Replaces, per topic, computePartitionLag (Main.java:376-404) + the sort
// (Main.java:228-235) + the greedy select/update loop (Main.java:237-266).
//
// Mapping.  A *group* of L lanes (L = 8/16/32/64) owns one topic; a 64-lane wavefront
// carries 64/L topics, a 256-thread workgroup 4x that.  Each lane holds E partition
// records in registers (L*E >= partitions of the topic, L >= consumers of the topic).
//
//   1. load      begin/end/committed/partition id, coalesced (lane-contiguous), compute the
//                lag in registers, build 96-bit sort records.                  [28 B/partition]
//   2. sort      bitonic network over the L*E records: strides < E in registers, larger
//                strides via DPP / v_permlane swaps.  No LDS, no HBM.
//   3. transpose sorted lags + ids through the group's LDS slice (padded, conflict-free).
//   4. greedy    ROUND-STRUCTURED: the count is the comparator's first key (Main.java:246-250),
//                so assignment proceeds in rounds of C partitions; in a round the k-th
//                partition goes to the k-th consumer in (total lag, memberId) order as of
//                the round start.  One round = one C-element bitonic sort of the consumer
//                bins (one bin per lane, in registers) + one add.  ceil(P/C) dependent
//                steps instead of P.  LA_ALGO_ARGMIN keeps the literal per-partition
//                wavefront argmin for cross-checking.
//   5. store     partition ids in assignment order + chosen member rank, coalesced. [8 B/partition]
//
// HBM traffic is exactly the algorithmic 36 B/partition (+ ~2% descriptors): every input
// byte is read once, every output byte written once, nothing spills to HBM in between.
#include "la_kernels.h"
#include "la_device.h"

namespace la {

template <int L, int E>
struct TileCfg {
    static constexpr int kGroupsPerWave = kWave / L;
    static constexpr int kWavesPerBlock = 4;
    static constexpr int kThreads = kWave * kWavesPerBlock;
    static constexpr int kTopicsPerBlock = kGroupsPerWave * kWavesPerBlock;
    static constexpr int kCap = L * E;                          // partitions per tile
    // 8-byte slots, one pad slot per 8: lane stride of E slots becomes bank-conflict-free
    static constexpr int kSlots = kCap + (kCap >> 3) + 1;
};

__device__ __forceinline__ int slot_of(int s) { return s + (s >> 3); }

template <int L, int E, bool ARGMIN>
__global__ __launch_bounds__(256) void wave_tile_assign_kernel(TileArgs a) {
    using Cfg = TileCfg<L, E>;
    __shared__ uint64_t lds[Cfg::kTopicsPerBlock * Cfg::kSlots];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int gl = lane & (L - 1);            // lane within group
    const int grp = lane / L;                 // group within wave
    const int64_t topic = ((int64_t)blockIdx.x * Cfg::kWavesPerBlock + wave) * Cfg::kGroupsPerWave + grp;
    uint64_t* slice = lds + (wave * Cfg::kGroupsPerWave + grp) * Cfg::kSlots;

    // ---- topic descriptor ------------------------------------------------------------
    int64_t p0 = 0, c0 = 0;
    int P = 0, C = 0;
    if (topic < a.n_topics) {
        p0 = a.part_off[topic];
        c0 = a.cons_off[topic];
        const int64_t Pl = a.part_off[topic + 1] - p0, Cl = a.cons_off[topic + 1] - c0;
        if (Pl > Cfg::kCap || Cl > L || Pl < 0 || Cl < 0) {
            if (gl == 0) atomicOr(a.status, kStatusShape);   // hint was wrong; leave outputs alone
        } else {
            P = (int)Pl;
            C = (int)Cl;
        }
    }

    // ---- 1. load + lag + records (element e = v*L + gl: lane-contiguous) ---------------
    Rec rec[E];
    const bool latest = a.reset_latest != 0;
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const int e = v * L + gl;
        rec[v].hi = rec[v].lo = rec[v].tb = 0xFFFFFFFFu;             // sentinel: sorts last
        if (e < P) {
            const int64_t g = p0 + e;
            int64_t lag;
            if (a.lag) {
                lag = a.lag[g];
            } else {
                const int64_t en = a.end[g], cm = a.committed[g];
                const int64_t bg = a.begin ? a.begin[g] : 0;
                lag = partition_lag(bg, en, cm, latest);
            }
            const uint64_t key = (uint64_t)lag ^ kLagKeyFlip;
            rec[v].hi = (uint32_t)(key >> 32);
            rec[v].lo = (uint32_t)key;
            rec[v].tb = (uint32_t)a.pid[g] ^ kPidBias;
        }
    }

    // ---- 2. sort (lag desc, partition asc) ---------------------------------------------
    bitonic_sort_tile<L, E>(rec, gl);

    // ---- 3. sorted position s = gl*E + r.  ids out (striped through LDS), lags into LDS --
#pragma unroll
    for (int r = 0; r < E; ++r) slice[slot_of(gl * E + r)] = rec[r].tb ^ kPidBias;
    wave_lds_fence();
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const int s = v * L + gl;
        if (s < P) a.out_pid[p0 + s] = (int32_t)(uint32_t)slice[slot_of(s)];
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint64_t key = ((uint64_t)rec[r].hi << 32) | rec[r].lo;
        slice[slot_of(gl * E + r)] = key ^ kLagKeyFlip;              // the lag itself
    }
    wave_lds_fence();

    // ---- 4. greedy -----------------------------------------------------------------------
    // bin of consumer `gl` (position in the rank-sorted list): biased total + index
    Rec bin;
    uint64_t total = kTotalBias;                                       // biased 0
    bin.tb = (gl < C) ? (uint32_t)gl : 0xFFFFFFFFu;
    if constexpr (!ARGMIN) {
        const int rounds = (C > 0) ? (P + C - 1) / C : 0;
        const int max_rounds = __builtin_amdgcn_readfirstlane(wave_max_i32(rounds));
        bin.hi = (gl < C) ? (uint32_t)(total >> 32) : 0xFFFFFFFFu;
        bin.lo = (gl < C) ? (uint32_t)total : 0xFFFFFFFFu;
        for (int q = 0; q < max_rounds; ++q) {
            // round 0 starts sorted: all totals 0, indices ascending
            if (q > 0) bitonic_sort_lanes(bin, gl, a.lc);
            const int s = q * C + gl;
            if (gl < C && s < P) {
                const uint64_t lag = slice[slot_of(s)];
                uint64_t t = (((uint64_t)bin.hi << 32) | bin.lo) + lag;     // Main.java:265
                bin.hi = (uint32_t)(t >> 32);
                bin.lo = (uint32_t)t;
                slice[slot_of(s)] = bin.tb;                                // chosen consumer
            }
        }
        total = ((uint64_t)bin.hi << 32) | bin.lo;
    } else {
        // literal form: P dependent wavefront argmins over (count, total, index)
        const int maxP = __builtin_amdgcn_readfirstlane(wave_max_i32(C > 0 ? P : 0));
        uint32_t count = (gl < C) ? 0u : 0xFFFFFFFFu;
        for (int s = 0; s < maxP; ++s) {
            uint32_t bc = count, bh = (uint32_t)(total >> 32), bl = (uint32_t)total, bi = bin.tb;
            for (int j = 1; j < L; j <<= 1) {                  // butterfly argmin inside the group
                Rec o; o.hi = bh; o.lo = bl; o.tb = bi;
                o = shfl_xor_dyn(o, j);
                Rec m; m.hi = bh; m.lo = bl; m.tb = bi;
                const uint32_t oc = (uint32_t)__shfl_xor((int)bc, j);
                const bool take = (oc < bc) | ((oc == bc) & rec_less(o, m));
                bc = take ? oc : bc; bh = take ? o.hi : bh; bl = take ? o.lo : bl; bi = take ? o.tb : bi;
            }
            if (s < P && C > 0 && bi == (uint32_t)gl) {
                total += slice[slot_of(s)];
                count += 1;
                slice[slot_of(s)] = (uint64_t)gl;
            }
        }
    }
    wave_lds_fence();

    // ---- 5. outputs --------------------------------------------------------------------------
    if (a.out_total && bin.tb < (uint32_t)C)
        a.out_total[c0 + bin.tb] = (int64_t)(total ^ kTotalBias);
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const int s = v * L + gl;
        if (s < P) {
            int32_t m = -1;
            if (C > 0) m = a.cons_rank[c0 + (uint32_t)slice[slot_of(s)]];
            a.out_rank[p0 + s] = m;
        }
    }
}

// ---- launcher ---------------------------------------------------------------------------------
template <int L, int E>
static hipError_t launch_one(const TileArgs& a, bool argmin, hipStream_t stream) {
    using Cfg = TileCfg<L, E>;
    const int64_t blocks = (a.n_topics + Cfg::kTopicsPerBlock - 1) / Cfg::kTopicsPerBlock;
    if (blocks <= 0) return hipSuccess;
    if (argmin)
        hipLaunchKernelGGL((wave_tile_assign_kernel<L, E, true>), dim3((unsigned)blocks), dim3(Cfg::kThreads), 0, stream, a);
    else
        hipLaunchKernelGGL((wave_tile_assign_kernel<L, E, false>), dim3((unsigned)blocks), dim3(Cfg::kThreads), 0, stream, a);
    return hipGetLastError();
}

template <int L>
static hipError_t launch_l(int e, const TileArgs& a, bool argmin, hipStream_t stream) {
    switch (e) {
        case 1: return launch_one<L, 1>(a, argmin, stream);
        case 2: return launch_one<L, 2>(a, argmin, stream);
        case 4: return launch_one<L, 4>(a, argmin, stream);
        case 8: return launch_one<L, 8>(a, argmin, stream);
        default: return launch_one<L, 16>(a, argmin, stream);
    }
}

static int pow2ceil(int64_t x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

bool wave_tile_fits(int64_t max_p, int64_t max_c) {
    return max_p <= kTileMaxPartitions && max_c <= kTileMaxConsumers;
}

// Picks the tile: the narrowest group that holds the consumers (more topics per wave, more
// of the sort in registers), at most 16 records per lane.
void wave_tile_pick(int64_t max_p, int64_t max_c, int* L, int* E) {
    int l = pow2ceil(max_c > 1 ? max_c : 1);
    if (l < 8) l = 8;
    while ((int64_t)l * 16 < max_p) l <<= 1;
    int e = pow2ceil((max_p + l - 1) / l);
    if (e < 1) e = 1;
    *L = l;
    *E = e;
}

hipError_t wave_tile_launch(TileArgs a, int64_t max_p, int64_t max_c, bool argmin, hipStream_t stream) {
    int L, E;
    wave_tile_pick(max_p, max_c, &L, &E);
    a.lc = pow2ceil(max_c > 1 ? max_c : 1);
    switch (L) {
        case 8: return launch_l<8>(E, a, argmin, stream);
        case 16: return launch_l<16>(E, a, argmin, stream);
        case 32: return launch_l<32>(E, a, argmin, stream);
        default: return launch_l<64>(E, a, argmin, stream);
    }
}

}  // namespace la
