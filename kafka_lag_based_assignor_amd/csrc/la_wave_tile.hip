// la_wave_tile.hip -- fused lag + sort + greedy for topics that fit one sub-wave tile.
//
// Replaces, per topic, computePartitionLag (Main.java:376-404) + the sort
// (Main.java:228-235) + the greedy select/update loop (Main.java:237-266).
//
// Mapping.  A *group* of L lanes (L = 8/16/32/64) owns one topic; a 64-lane wavefront
// carries 64/L topics, a 256-thread workgroup 4x that.  Each lane holds E partition
// records in registers (L*E >= partitions of the topic, L >= consumers of the topic).
//
//   1. load      begin/end/committed/partition id, 16 B per lane per array where the tile
//                allows it (the order records enter the sorter is irrelevant, so a lane takes
//                pairs of neighbours); lag in registers.                     [28 B/partition]
//   2. sort      bitonic network over the L*E records: strides < E in registers, larger
//                strides via DPP / v_permlane swaps.  No LDS, no HBM.
//   3. transpose sorted records through the group's LDS slice (padded, conflict-free writes).
//   4. greedy    ROUND-STRUCTURED: the count is the comparator's first key (Main.java:246-250),
//                so assignment proceeds in rounds of C partitions; in a round the k-th
//                partition goes to the k-th consumer in (total lag, memberId) order as of
//                the round start.  One round = one L-lane bitonic sort of the consumer
//                bins (one bin per lane, in registers) + one add.  ceil(P/C) dependent
//                steps instead of P.  LA_ALGO_ARGMIN keeps the literal per-partition
//                wavefront argmin for cross-checking.
//   5. store     partition ids in assignment order + chosen member rank.     [8 B/partition]
//
// Two record formats, chosen per wavefront at run time (wave-uniform branch):
//
//   packed   one 64-bit word per record.  With sh = bits needed by the wave's largest
//            partition id:   rec = ((2^(63-sh)-1 - lag) << sh) | id     ascending == (lag desc, id asc)
//            and consumer bins  bin = (total << 6) | index.  Taken when every lag of the wave
//            satisfies 0 <= lag < 2^min(63-sh, 57-log2(L*E)) and no id is negative; then no
//            total can reach 2^57, nothing wraps, and the packed order is exactly the
//            reference's.  One 64-bit compare + two selects per compare-exchange.
//   wide     (key64, tie-break32) records, biased so that unsigned order == Java's signed
//            order; totals wrap like Java's long.  Any int64 lag, any int32 id.
//            LA_ALGO_ROUNDS_WIDE forces it (tests run both on the same inputs).
//
// HBM traffic is exactly the algorithmic 36 B/partition (+ ~2% descriptors): every input
// byte is read once, every output byte written once, nothing spills to HBM in between.
#include "la_kernels.h"
#include "la_device.h"

namespace la {

enum : int { kModeAuto = 0, kModeWide = 1, kModeArgmin = 2 };

template <int L, int E>
struct TileCfg {
    static constexpr int kGroupsPerWave = kWave / L;
    static constexpr int kWavesPerBlock = 4;
    static constexpr int kThreads = kWave * kWavesPerBlock;
    static constexpr int kTopicsPerBlock = kGroupsPerWave * kWavesPerBlock;
    static constexpr int kCap = L * E;                          // partitions per tile
    // 8-byte slots, one pad slot per 8: lane stride of E slots becomes bank-conflict-free
    static constexpr int kSlots = kCap + (kCap >> 3) + 1;
    static constexpr int kLog2Cap = (kCap <= 1) ? 0 : (31 - __builtin_clz(kCap - 1)) + 1;
};

__device__ __forceinline__ int slot_of(int s) { return s + (s >> 3); }

// 16-byte / 8-byte loads from arrays that are only element-aligned (a topic may start anywhere)
struct __attribute__((aligned(8))) I64x2 { int64_t x, y; };
struct __attribute__((aligned(4))) I32x2 { int32_t x, y; };

// position of the v-th record a lane loads: pairs of neighbours, so int64 arrays move 16 B per lane
template <int L, int E>
__device__ __forceinline__ int load_index(int v, int gl) {
    if constexpr (E >= 2) return (v >> 1) * (2 * L) + 2 * gl + (v & 1);
    else return gl;
}

// ---- 1. load + lag ------------------------------------------------------------------------------
template <int L, int E>
__device__ __forceinline__ void load_lags(const TileArgs& a, int64_t p0, int P, int gl, int64_t (&lag)[E],
                                          int32_t (&pid)[E]) {
    const bool latest = a.reset_latest != 0;
    if constexpr (E >= 2) {
#pragma unroll
        for (int v = 0; v < E; v += 2) {
            const int e = load_index<L, E>(v, gl);
            lag[v] = lag[v + 1] = 0;
            pid[v] = pid[v + 1] = 0;
            if (e + 1 < P) {
                const int64_t g = p0 + e;
                const I32x2 id = *reinterpret_cast<const I32x2*>(a.pid + g);
                pid[v] = id.x; pid[v + 1] = id.y;
                if (a.lag) {
                    const I64x2 l = *reinterpret_cast<const I64x2*>(a.lag + g);
                    lag[v] = l.x; lag[v + 1] = l.y;
                } else {
                    const I64x2 en = *reinterpret_cast<const I64x2*>(a.end + g);
                    const I64x2 cm = *reinterpret_cast<const I64x2*>(a.committed + g);
                    I64x2 bg; bg.x = bg.y = 0;
                    if (!latest && a.begin) bg = *reinterpret_cast<const I64x2*>(a.begin + g);
                    lag[v] = partition_lag(bg.x, en.x, cm.x, latest);
                    lag[v + 1] = partition_lag(bg.y, en.y, cm.y, latest);
                }
            } else if (e < P) {
                const int64_t g = p0 + e;
                pid[v] = a.pid[g];
                if (a.lag) lag[v] = a.lag[g];
                else lag[v] = partition_lag((!latest && a.begin) ? a.begin[g] : 0, a.end[g], a.committed[g], latest);
            }
        }
    } else {
        lag[0] = 0; pid[0] = 0;
        if (gl < P) {
            const int64_t g = p0 + gl;
            pid[0] = a.pid[g];
            if (a.lag) lag[0] = a.lag[g];
            else lag[0] = partition_lag((!latest && a.begin) ? a.begin[g] : 0, a.end[g], a.committed[g], latest);
        }
    }
}

// ---- packed path ------------------------------------------------------------------------------------
template <int L, int E>
__device__ __forceinline__ void assign_packed(const TileArgs& a, uint64_t* slice, int32_t* rank_tab, int64_t p0,
                                              int64_t c0, int P, int C, int gl, const int64_t (&lag)[E],
                                              const int32_t (&pid)[E], int sh) {
    const uint64_t lag_max = (~0ull >> 1) >> sh;                    // 2^(63-sh) - 1
    const uint32_t pid_mask = (uint32_t)((1ull << sh) - 1);

    // ---- records; empty slots sort last ---------------------------------------------------------
    uint64_t rec[E];
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const int e = load_index<L, E>(v, gl);
        rec[v] = (e < P) ? (((lag_max - (uint64_t)lag[v]) << sh) | (uint32_t)pid[v]) : ~0ull;
    }

    // ---- 2. sort (lag desc, partition asc) -------------------------------------------------------
    bitonic_sort_tile64<L, E>(rec, gl);

    // ---- 3. sorted position s = gl*E + r -> LDS -----------------------------------------------------
#pragma unroll
    for (int r = 0; r < E; ++r) slice[slot_of(gl * E + r)] = rec[r];
    if (gl < C) rank_tab[gl] = a.cons_rank[c0 + gl];
    wave_lds_fence();

    // ---- 4. greedy rounds: bin = (total << 6) | index in the rank-sorted consumer list ----------------
    uint64_t bin = (gl < C) ? (uint64_t)gl : ~0ull;
    const int rounds = (C > 0) ? (P + C - 1) / C : 0;
    const int max_rounds = __builtin_amdgcn_readfirstlane(wave_max_i32(rounds));
    for (int q = 0; q < max_rounds; ++q) {
        // round 0 starts sorted: all totals 0, indices ascending
        if (q > 0) bitonic_sort_lanes64<L>(bin, gl);
        const int s = q * C + gl;
        if (gl < C && s < P) {
            const uint64_t r = slice[slot_of(s)];
            bin += (lag_max - (r >> sh)) << 6;                                   // Main.java:265
            slice[slot_of(s)] = ((uint64_t)((uint32_t)bin & 63u) << 32) | ((uint32_t)r & pid_mask);
        }
    }
    wave_lds_fence();

    // ---- 5. outputs ------------------------------------------------------------------------------------
    if (a.out_total && gl < C && bin != ~0ull) a.out_total[c0 + ((uint32_t)bin & 63u)] = (int64_t)(bin >> 6);
    if constexpr (E >= 4) {
        // four consecutive positions per lane: 16-byte stores
#pragma unroll
        for (int k = 0; k < E / 4; ++k) {
            const int s0 = k * 4 * L + 4 * gl;
            int32_t op[4], om[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint64_t w = slice[slot_of(s0 + i)];
                op[i] = (int32_t)((uint32_t)w & pid_mask);
                om[i] = (C > 0) ? rank_tab[(uint32_t)(w >> 32) & 63u] : -1;
            }
            if (s0 + 3 < P) {
                struct __attribute__((aligned(4))) I32x4 { int32_t x, y, z, w; };
                I32x4 vp; vp.x = op[0]; vp.y = op[1]; vp.z = op[2]; vp.w = op[3];
                I32x4 vm; vm.x = om[0]; vm.y = om[1]; vm.z = om[2]; vm.w = om[3];
                *reinterpret_cast<I32x4*>(a.out_pid + p0 + s0) = vp;
                *reinterpret_cast<I32x4*>(a.out_rank + p0 + s0) = vm;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (s0 + i < P) { a.out_pid[p0 + s0 + i] = op[i]; a.out_rank[p0 + s0 + i] = om[i]; }
            }
        }
    } else {
#pragma unroll
        for (int v = 0; v < E; ++v) {
            const int s = v * L + gl;
            if (s < P) {
                const uint64_t w = slice[slot_of(s)];
                a.out_pid[p0 + s] = (int32_t)((uint32_t)w & pid_mask);
                a.out_rank[p0 + s] = (C > 0) ? rank_tab[(uint32_t)(w >> 32) & 63u] : -1;
            }
        }
    }
}

// ---- wide path (any int64 lag, any int32 id; also hosts the literal argmin form) ---------------------
template <int L, int E, bool ARGMIN>
__device__ __forceinline__ void assign_wide(const TileArgs& a, uint64_t* slice, int64_t p0, int64_t c0, int P, int C,
                                            int gl, const int64_t (&lag)[E], const int32_t (&pid)[E]) {
    Rec rec[E];
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const int e = load_index<L, E>(v, gl);
        rec[v].hi = rec[v].lo = rec[v].tb = 0xFFFFFFFFu;             // sentinel: sorts last
        if (e < P) {
            const uint64_t key = (uint64_t)lag[v] ^ kLagKeyFlip;
            rec[v].hi = (uint32_t)(key >> 32);
            rec[v].lo = (uint32_t)key;
            rec[v].tb = (uint32_t)pid[v] ^ kPidBias;
        }
    }

    bitonic_sort_tile<L, E>(rec, gl);

    // sorted position s = gl*E + r.  ids out (striped through LDS), lags into LDS
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
            if (q > 0) bitonic_sort_lanes(bin, gl, a.lc);
            const int s = q * C + gl;
            if (gl < C && s < P) {
                const uint64_t lg = slice[slot_of(s)];
                uint64_t t = (((uint64_t)bin.hi << 32) | bin.lo) + lg;       // Main.java:265
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

template <int L, int E, int MODE>
__global__ __launch_bounds__(256) void wave_tile_assign_kernel(TileArgs a) {
    using Cfg = TileCfg<L, E>;
    __shared__ uint64_t lds[Cfg::kTopicsPerBlock * Cfg::kSlots];
    __shared__ int32_t rank_lds[Cfg::kTopicsPerBlock * L];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int gl = lane & (L - 1);            // lane within group
    const int grp = lane / L;                 // group within wave
    const int64_t topic = ((int64_t)blockIdx.x * Cfg::kWavesPerBlock + wave) * Cfg::kGroupsPerWave + grp;
    uint64_t* slice = lds + (wave * Cfg::kGroupsPerWave + grp) * Cfg::kSlots;
    int32_t* rank_tab = rank_lds + (wave * Cfg::kGroupsPerWave + grp) * L;

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

    int64_t lag[E];
    int32_t pid[E];
    load_lags<L, E>(a, p0, P, gl, lag, pid);

    if constexpr (MODE == kModeAuto) {
        // can this wavefront's records be packed into 64 bits?  (empty slots hold lag 0, id 0)
        uint32_t id_or = 0;
        uint64_t lag_or = 0;
#pragma unroll
        for (int v = 0; v < E; ++v) { id_or |= (uint32_t)pid[v]; lag_or |= (uint64_t)lag[v]; }
        id_or = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_or_u32(id_or));
        const int sh = 32 - __builtin_clz(id_or | 1u);                   // 1..32 (32: a negative id)
        int lb = 63 - sh;
        if (lb > 57 - Cfg::kLog2Cap) lb = 57 - Cfg::kLog2Cap;
        const bool fits = (sh < 32) && ((lag_or >> lb) == 0);
        if (__builtin_amdgcn_ballot_w64(!fits) == 0) {
            assign_packed<L, E>(a, slice, rank_tab, p0, c0, P, C, gl, lag, pid, sh);
            return;
        }
    }
    assign_wide<L, E, MODE == kModeArgmin>(a, slice, p0, c0, P, C, gl, lag, pid);
}

// ---- launcher ---------------------------------------------------------------------------------
template <int L, int E>
static hipError_t launch_one(const TileArgs& a, int mode, hipStream_t stream) {
    using Cfg = TileCfg<L, E>;
    const int64_t blocks = (a.n_topics + Cfg::kTopicsPerBlock - 1) / Cfg::kTopicsPerBlock;
    if (blocks <= 0) return hipSuccess;
    const dim3 g((unsigned)blocks), b(Cfg::kThreads);
    if (mode == kModeArgmin)
        hipLaunchKernelGGL((wave_tile_assign_kernel<L, E, kModeArgmin>), g, b, 0, stream, a);
    else if (mode == kModeWide)
        hipLaunchKernelGGL((wave_tile_assign_kernel<L, E, kModeWide>), g, b, 0, stream, a);
    else
        hipLaunchKernelGGL((wave_tile_assign_kernel<L, E, kModeAuto>), g, b, 0, stream, a);
    return hipGetLastError();
}

template <int L>
static hipError_t launch_l(int e, const TileArgs& a, int mode, hipStream_t stream) {
    switch (e) {
        case 1: return launch_one<L, 1>(a, mode, stream);
        case 2: return launch_one<L, 2>(a, mode, stream);
        case 4: return launch_one<L, 4>(a, mode, stream);
        case 8: return launch_one<L, 8>(a, mode, stream);
        default: return launch_one<L, 16>(a, mode, stream);
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

hipError_t wave_tile_launch(TileArgs a, int64_t max_p, int64_t max_c, int mode, hipStream_t stream) {
    int L, E;
    wave_tile_pick(max_p, max_c, &L, &E);
    a.lc = pow2ceil(max_c > 1 ? max_c : 1);
    switch (L) {
        case 8: return launch_l<8>(E, a, mode, stream);
        case 16: return launch_l<16>(E, a, mode, stream);
        case 32: return launch_l<32>(E, a, mode, stream);
        default: return launch_l<64>(E, a, mode, stream);
    }
}

}  // namespace la
