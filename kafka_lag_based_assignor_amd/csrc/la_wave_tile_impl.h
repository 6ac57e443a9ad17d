// la_wave_tile_impl.h -- fused lag + sort + greedy for topics that fit one sub-wave tile (device code and
// the per-tile-shape launchers; included by the la_wave_tile_l*.hip translation units, one per group width,
// so they compile in parallel).
//
// Replaces, per topic, computePartitionLag (Main.java:376-404) + the sort
// (Main.java:228-235) + the greedy select/update loop (Main.java:237-266).
//
// Mapping.  A *group* of L lanes (L = 8/16/32/64) owns one topic; a 64-lane wavefront
// carries 64/L topics, a 256-thread workgroup 4x that.  Each lane holds E partition
// records in registers (L*E >= partitions of the topic, L >= consumers of the topic).
//
//   1. load      committed / end / partition id, 16 B per lane per array, unconditional and clamped, all
//                issued back to back; `begin` only where there is no committed offset; lag in registers.
//   2. format    per wavefront: packed 64-bit records if ids and lags are narrow enough (kernel 1), else
//                the tile is deferred to the wide-record kernel (kernel 2).
//   3. sort      32-bit keys (top bits of the record, id as tie-break) through a bitonic network whose steps across
//                lanes are a DPP move + v_med3_u32 (la_sort32.h); records fetched from the LDS slice by the key's
//                index and CHECKED to be strictly ascending; the full 64-bit network only if the check fails.  No HBM.
//   4. greedy    ROUND-STRUCTURED: the count is the comparator's first key (Main.java:246-250),
//                so assignment proceeds in rounds of C partitions; in a round the k-th
//                partition goes to the k-th consumer in (total lag, memberId) order as of
//                the round start.  One round = one L-lane bitonic sort of the consumer
//                bins (one bin per lane, in registers) + one add.  ceil(P/C) dependent
//                steps instead of P.  LA_ALGO_ARGMIN keeps the literal per-partition
//                wavefront argmin for cross-checking.
//   5. store     partition ids in assignment order + chosen member rank, 16 B per lane.  [8 B/partition]
//
// Record formats:
//
//   packed   one 64-bit word per record.  With sh = bits of the wavefront's largest partition id and
//            lbw = bits of its largest lag:   rec = ((2^lbw - 1 - lag) << sh) | id     ascending == (lag desc, id asc)
//            and consumer bins  bin = (total << 6) | index.  Taken when no id or lag is negative and
//            lbw <= min(63-sh, 57-log2(L*E)); then no total can reach 2^57, nothing wraps, and the
//            packed order is exactly the reference's.
//   wide     (key64, tie-break32) records, biased so that unsigned order == Java's signed
//            order; totals wrap like Java's long.  Any int64 lag, any int32 id.
//            LA_ALGO_ROUNDS_WIDE forces it (tests run both on the same inputs).
//
// Steps 1-5 exist in two forms: a wavefront whose topics all fill their tile exactly (P == L*E) runs the one without index
// clamps, validity selects and empty-slot sentinels (template flag FULL, decided by one ballot on the descriptors).
//
// HBM traffic is at most the algorithmic 36 B/partition (+ ~2% descriptors): every input byte is read at
// most once (`begin` usually not at all), every output byte written once, nothing spills in between.
#pragma once
#include "la_kernels.h"
#include "la_device.h"
#include "la_sort64.h"
#include "la_sort32.h"
#include "la_group_small.h"
#ifdef LA_LAB
#include <cstdio>
#include <cstdlib>
#endif

namespace la {

enum : int { kModeAuto = 0, kModeWide = 1, kModeArgmin = 2 };

// Ablation hooks for tools/tile_lab.hip (phase timing on the GPU); always 0 in the library.
//   1: no sort, no greedy (memory only)   2: no global loads / stores (compute only)
//   3: sort but no greedy                 4: greedy but no sort       10+n: at most n greedy rounds
#ifndef LA_ABLATE
#define LA_ABLATE 0
#endif
constexpr int kAblate = LA_ABLATE;
#ifndef LA_SKIP_SORTED_ROUNDS
#define LA_SKIP_SORTED_ROUNDS 1
#endif
constexpr bool kSkipSortedRounds = LA_SKIP_SORTED_ROUNDS != 0;   // lab switch for the A/B of bins_in_order()
#ifndef LA_WPB
#define LA_WPB 4          // wavefronts per workgroup (1, 2 and 8 measured: see DESIGN.md)
#endif

template <int L, int E>
struct TileCfg {
    static constexpr int kGroupsPerWave = kWave / L;
    static constexpr int kWavesPerBlock = LA_WPB;
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
// The loads of a tile are issued first, all of them, and used later; Raw holds what is in flight.
//
// Every load is unconditional: lanes beyond their topic's partitions (and groups beyond the last topic)
// read a CLAMPED in-bounds element and ignore it.  Branches around loads make hipcc wait for each load
// before issuing the next (one HBM round trip per branch); straight-line loads all go out back to back.
template <int E>
struct Raw {
    I64x2 en[(E + 1) / 2];      // end offsets (or precomputed lags)
    I64x2 cm[(E + 1) / 2];      // committed offsets; after stage 2 (earliest mode): the offset to subtract
    I32x2 id[(E + 1) / 2];      // partition ids
};

// IDX = uint32_t when every element index and byte offset of the batch fits 32 bits (the launcher
// checks): clamps become one v_min_u32 and addresses SGPR base + 32-bit VGPR offset instead of 64-bit
// VALU arithmetic.  int64_t otherwise.
template <typename IDX>
struct TopicDescT {
    IDX p0, c0;
    int P, C;
};
using TopicDesc = TopicDescT<int64_t>;

// element `idx` of `base`, read as V (a 1- or 2-element vector of T).  With 32-bit indexing the byte offset is
// formed in 32 bits, which lets the load use the SGPR-base + VGPR-offset form (no 64-bit VALU address math).
// The tile's inputs are read ONCE: non-temporal loads (round 6) keep them from displacing anything in the caches on their way
// through -- same-box A/B on the 25.6 M-partition target, three alternations: 0.1440 / 0.1434 / 0.1442 ms per step with plain
// loads, 0.1387 / 0.1387 / 0.1385 with these (frac 0.79 -> 0.82; profiles/r06_ab_nt_loads.txt).  -DLA_NT_LOADS=0: plain loads.
#ifndef LA_NT_LOADS
#define LA_NT_LOADS 1
#endif
template <typename V, typename T, typename IDX>
__device__ __forceinline__ V load_at(const T* base, IDX idx) {
    const V* p;
    if constexpr (sizeof(IDX) == 4)
        p = reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + (uint32_t)(idx * (uint32_t)sizeof(T)));
    else
        p = reinterpret_cast<const V*>(base + idx);
#if LA_NT_LOADS
    if constexpr (sizeof(V) == 16) {
        typedef long long LL2 __attribute__((ext_vector_type(2), aligned(8)));
        const LL2 v = __builtin_nontemporal_load(reinterpret_cast<const LL2*>(p));
        V out;
        __builtin_memcpy(&out, &v, 16);
        return out;
    } else if constexpr (sizeof(V) == 8) {
        typedef int I2 __attribute__((ext_vector_type(2), aligned(4)));
        const I2 v = __builtin_nontemporal_load(reinterpret_cast<const I2*>(p));
        V out;
        __builtin_memcpy(&out, &v, 8);
        return out;
    } else {
        return *p;
    }
#else
    return *p;
#endif
}

// element index of a lane's v-th pair, clamped so that a 2-element load stays inside [0, n_total).
// FULL (here and below): every topic of the wavefront fills its tile exactly (P == L * E) -- every slot holds a
// partition, no index leaves its topic, and the clamps, validity selects and empty-slot sentinels of the general
// form are dead code.  The kernel decides per wavefront (one ballot on the descriptors) and runs one of the two forms;
// partition counts that are a power of two from 8 to 1 024 -- the usual choice -- are full tiles of some shape.
template <int L, int E, bool FULL = false, typename D>
__device__ __forceinline__ auto clamped_index(const TileArgs& a, const D& d, int v, int gl) {
    using IDX = decltype(d.p0);
    const IDX g = d.p0 + (IDX)load_index<L, E>(v, gl);
    if constexpr (FULL) return g;
    const IDX hi = (IDX)(a.n_total - (E >= 2 ? 2 : 1));
    return g < hi ? g : hi;
}

// stage 1: everything that does not depend on data.  Committed offsets first: stage 2 needs only them.
template <int L, int E, bool FULL = false, typename D>
__device__ __forceinline__ void issue_loads(const TileArgs& a, const D& d, int gl, Raw<E>& raw) {
    if constexpr (kAblate == 2) return;
    constexpr int NP = (E + 1) / 2;
    const int64_t* src_en = a.lag ? a.lag : a.end;
    if (!a.lag) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const auto g = clamped_index<L, E, FULL>(a, d, 2 * k, gl);
            if constexpr (E >= 2) raw.cm[k] = load_at<I64x2>(a.committed, g);
            else { raw.cm[k].x = load_at<int64_t>(a.committed, g); raw.cm[k].y = 0; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) raw.cm[k].x = raw.cm[k].y = 0;
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const auto g = clamped_index<L, E, FULL>(a, d, 2 * k, gl);
        if constexpr (E >= 2) {
            raw.en[k] = load_at<I64x2>(src_en, g);
            raw.id[k] = load_at<I32x2>(a.pid, g);
        } else {
            raw.en[k].x = load_at<int64_t>(src_en, g); raw.en[k].y = 0;
            raw.id[k].x = load_at<int32_t>(a.pid, g); raw.id[k].y = 0;
        }
    }
}

// stage 2: the beginning offset is needed only where there is no committed offset and auto.offset.reset is
// not "latest" (Main.java:384-396).  Lanes that need it read `begin`; every other lane reads begin[0] (one cache
// line for all of them, no HBM traffic to speak of) and keeps what it has, so the loads stay unconditional and
// batched; afterwards cm is the "next offset" of Main.java:386-396 itself.
template <int L, int E>
__device__ __forceinline__ bool second_stage_needed(const TileArgs& a) {
    return !a.lag && !a.reset_latest && a.begin;                        // wave-uniform
}

template <int L, int E, bool FULL = false, typename D>
__device__ __forceinline__ void issue_begin_loads(const TileArgs& a, const D& d, int gl, Raw<E>& raw) {
    if constexpr (kAblate == 2) return;
    constexpr int NP = (E + 1) / 2;
    if (!second_stage_needed<L, E>(a)) return;
    // A wavefront in which NO partition lacks its committed offset -- every topic of an established consumer group -- skips the
    // stage altogether (round 6): ONE wave-uniform branch around all of its loads (they still go out back to back inside it).
    // For a resident batch that is a handful of cached reads of begin[0]; for a zero-copy small call, whose arrays live in host
    // memory, it is a whole PCIe round trip (~1.3 us of a ~20 us rebalance) for a value nobody uses.
    {
        bool any = false;
#pragma unroll
        for (int k = 0; k < NP; ++k) any |= (raw.cm[k].x < 0) | (E >= 2 && raw.cm[k].y < 0);
        if (__builtin_amdgcn_ballot_w64(any) == 0 && !(a.flags & kTileAlwaysStage2)) return;
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const auto g = clamped_index<L, E, FULL>(a, d, 2 * k, gl);
        // a lane whose element has a committed offset needs nothing: all such lanes read ONE fixed word (a single
        // cache line per wavefront instead of a second pass over their own lines through L2) and keep their value
        using G = decltype(g);
        const bool nx = raw.cm[k].x < 0;
        if constexpr (E >= 2) {
            // the pair's two begin offsets in ONE 16-byte load (the form of stage 1) when either is needed: a group
            // that has no committed offsets at all -- a new consumer group, Main.java:393-396 -- then streams `begin`
            // exactly as it streams `end` and `committed` (two 8-byte loads per pair until round 6)
            const bool ny = raw.cm[k].y < 0;
            const I64x2 bv = load_at<I64x2>(a.begin, (nx | ny) ? g : (G)0);
            raw.cm[k].x = nx ? bv.x : raw.cm[k].x;
            raw.cm[k].y = ny ? bv.y : raw.cm[k].y;
        } else {
            const int64_t vx = a.begin[nx ? g : (G)0];
            raw.cm[k].x = nx ? vx : raw.cm[k].x;
        }
    }
}

// computePartitionLag (Main.java:376-404) on what the two stages fetched
template <int L, int E, bool FULL = false, typename D>
__device__ __forceinline__ void finish_lags(const TileArgs& a, const D& d, int gl, const Raw<E>& raw,
                                            int64_t (&lag)[E], int32_t (&pid)[E]) {
    const bool latest = a.reset_latest != 0;
    if constexpr (kAblate == 2) {
#pragma unroll
        for (int v = 0; v < E; ++v) {
            const uint32_t h = ((uint32_t)d.p0 + (uint32_t)(v * L + gl)) * 2654435761u;
            lag[v] = (int64_t)(h >> 2) + a.reset_latest;
            pid[v] = (int32_t)((v * L + gl) * 77 + 13) & (L * E - 1);
        }
        return;
    }
    constexpr int NP = (E + 1) / 2;
    // after stage 2 the cm registers hold the offset to subtract; otherwise "none" (< 0) means `end`
    // (latest) or 0 (earliest without a begin array)
    const bool cm_is_next = second_stage_needed<L, E>(a);
    auto lag_of = [&](int64_t en, int64_t cm) -> int64_t {
        if (a.lag) return en;
        if (cm_is_next) { const int64_t dlt = (int64_t)((uint64_t)en - (uint64_t)cm); return dlt > 0 ? dlt : 0; }
        return partition_lag(0, en, cm, latest);
    };
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int e = load_index<L, E>(2 * k, gl);
        // a pair clamped back by one element holds this lane's first element in .y (only the very last
        // element of the batch can be in that position)
        const bool shifted = !FULL && (E >= 2) && (d.p0 + (decltype(d.p0))e == (decltype(d.p0))(a.n_total - 1));
        const int64_t en_x = shifted ? raw.en[k].y : raw.en[k].x;
        const int64_t cm_x = shifted ? raw.cm[k].y : raw.cm[k].x;
        pid[2 * k] = shifted ? raw.id[k].y : raw.id[k].x;
        lag[2 * k] = lag_of(en_x, cm_x);
        if constexpr (E >= 2) {
            pid[2 * k + 1] = raw.id[k].y;
            lag[2 * k + 1] = lag_of(raw.en[k].y, raw.cm[k].y);
        }
    }
    // slots past the topic's partitions: lag 0, id 0 (they must not influence the format decision)
    if constexpr (FULL) return;
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const bool valid = load_index<L, E>(v, gl) < d.P;
        lag[v] = valid ? lag[v] : 0;
        pid[v] = valid ? pid[v] : 0;
    }
}

// ---- packed path ------------------------------------------------------------------------------------
// records of the packed format:  ((2^lbw - 1 - lag) << sh) | id,  empty slots all ones (sort last)
template <int L, int E, bool FULL = false>
__device__ __forceinline__ void pack_records(int P, int gl, const int64_t (&lag)[E], const int32_t (&pid)[E], int sh,
                                             uint64_t lag_max, P64 (&rec)[E]) {
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const int e = load_index<L, E>(v, gl);
        rec[v] = p64_from((FULL || e < P) ? (((lag_max - (uint64_t)lag[v]) << sh) | (uint32_t)pid[v]) : ~0ull);
    }
}

// ---- 2. sort (lag desc, partition asc), leaving the sorted records in the LDS slice ---------------------
// Fast form: sort 32-bit keys that are a monotone function of the record "almost always", fetch each
// record from the slice through the index carried in the key's low bits, then CHECK that the fetched
// records are strictly ascending.  If any neighbour pair in the wavefront is not, the wavefront re-sorts
// the full 64-bit records (la_sort64.h).  Two key layouts (wave-uniform choice):
//   ids dense (every id < tile capacity):  key = (lag part, low bits dropped to fit) << sh | id, record stored
//        at slot `id`.  Equal lags -- the common tie, e.g. many partitions with lag 0 -- are ordered by id
//        inside the key itself, exactly as Main.java:231-234 orders them.
//   otherwise:  key = (top bits of the record) << idx_bits | slot, record stored at its load slot.
// Dropped low bits (and duplicate ids, which would collide in a slot) can only make the check fail, never
// pass wrongly: the check is on the full records.
template <int L, int E, bool FULL = false>
__device__ __forceinline__ void sort_into_slice(uint64_t* slice, int gl, P64 (&rec)[E], int lbw, int sh) {
    using Cfg = TileCfg<L, E>;
    if constexpr (kAblate == 1 || kAblate == 4) {
#pragma unroll
        for (int r = 0; r < E; ++r) slice[slot_of(gl * E + r)] = p64_value(rec[r]);
        return;
    }
#ifdef LA_TILE_SORT64   // lab build (round 3): the full 64-bit network always -- no scatter by id, no gather by key
    bitonic_sort_tile_p64<L, E>(rec);
#pragma unroll
    for (int r = 0; r < E; ++r) slice[slot_of(gl * E + r)] = p64_value(rec[r]);
    return;
#endif
    const bool by_id = sh <= Cfg::kLog2Cap;                               // wave-uniform
    const int idx_bits = by_id ? sh : Cfg::kLog2Cap;
    const uint32_t idx_mask = (1u << idx_bits) - 1;
    const int top_bits = by_id ? lbw : lbw + sh;                          // bits above the index field's source
    const int keep = 31 - idx_bits;                                       // key < 2^31: all-ones stays largest
    const int drop = (by_id ? sh : 0) + (top_bits > keep ? top_bits - keep : 0);
    uint32_t key[E];
    if (by_id && drop == sh) {
        // nothing dropped: the key IS the record (it is shorter than 31 bits); no fetch, no check
#pragma unroll
        for (int v = 0; v < E; ++v) key[v] = (!FULL && rec[v].hi == 0xFFFFFFFFu) ? 0xFFFFFFFFu : rec[v].lo;
        bitonic_sort_tile_u32<L, E>(key);
#pragma unroll
        for (int r = 0; r < E; ++r)
            slice[slot_of(gl * E + r)] = (!FULL && key[r] == 0xFFFFFFFFu) ? ~0ull : (uint64_t)key[r];
        return;
    }
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const uint64_t r = p64_value(rec[v]);
        const bool valid = FULL || r != ~0ull;
        const uint32_t idx = by_id ? ((uint32_t)r & idx_mask) : (uint32_t)load_index<L, E>(v, gl);
        if (valid) slice[slot_of((int)idx)] = r;
        key[v] = valid ? (((uint32_t)(r >> drop) << idx_bits) | idx) : 0xFFFFFFFFu;
    }
    wave_lds_fence();
    bitonic_sort_tile_u32<L, E>(key);

    P64 got[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint64_t x = slice[slot_of((int)(key[r] & idx_mask))];
        got[r] = p64_from((!FULL && key[r] == 0xFFFFFFFFu) ? ~0ull : x);
    }
    // strictly ascending?  position s = gl*E + r; all-ones records (empty slots) are all at the end
    bool bad = false;
#pragma unroll
    for (int r = 0; r + 1 < E; ++r) {
        const uint64_t x = p64_value(got[r]), y = p64_value(got[r + 1]);
        bad |= (x >= y) && (FULL || y != ~0ull);
    }
    {
        const uint64_t x = p64_value(got[E - 1]);
        const uint64_t y = ((uint64_t)(uint32_t)__shfl_down((int)got[0].hi, 1) << 32) | (uint32_t)__shfl_down((int)got[0].lo, 1);
        bad |= (gl != L - 1) && (x >= y) && (FULL || y != ~0ull);
    }
    wave_lds_fence();
    if (__builtin_amdgcn_ballot_w64(bad) != 0) {
        bitonic_sort_tile_p64<L, E>(rec);                                 // full records, full network
#pragma unroll
        for (int r = 0; r < E; ++r) slice[slot_of(gl * E + r)] = p64_value(rec[r]);
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) slice[slot_of(gl * E + r)] = p64_value(got[r]);
    }
}

// ---- 4. greedy rounds: bin = (total << 6) | index in the rank-sorted consumer list ---------------------------
// LC = lanes the bins' network spans: the group width L, or less when every topic of the wavefront has at most LC
// consumers (a 1 000-partition topic with 3 consumers sits in a 64-lane group; its 334 rounds then sort 4 lanes,
// 3 steps, instead of 64 lanes, 21 steps).  Lanes >= C hold the all-ones sentinel, so any LC >= C sorts the same.
// Rounds whose bins are in ascending order already skip their sort (lanes_in_order_p64, la_sort64.h: one DPP wave shift
// per dword and one 64-bit compare): all-zero lags, a Zipf tail (the second round of the 100 000 x 256 x 32 target),
// topics with few consumers or few distinct lags.
template <int L, int LC>
__device__ __forceinline__ void greedy_rounds_tile(P64& bin, uint64_t* slice, int P, int C, int gl, int sh,
                                                   uint64_t lag_max, uint32_t pid_mask, int max_rounds) {
    for (int q = 0; q < max_rounds; ++q) {
        // round 0 starts sorted: all totals 0, indices ascending
        if (q >= 1 && kSkipSortedRounds && lanes_in_order_p64(bin, gl)) {
            // nothing to sort
        } else if (q == 1) {
            // After round 0 consumer k holds the k-th largest lag: if those lags are STRICTLY descending over
            // a full group, ascending (total, index) order is simply the reverse -- one mirror instead of a
            // sort.  Equal lags (index order must win) or a partly filled group take the sort.
            bool mirrored = false;
            if constexpr (LC == L) {
                const uint64_t mine = p64_value(bin);
                const uint64_t prev = ((uint64_t)(uint32_t)__shfl_up((int)bin.hi, 1) << 32) | (uint32_t)__shfl_up((int)bin.lo, 1);
                const bool bad = (C != L) || (gl > 0 && !((prev >> 6) > (mine >> 6)));
                if (__builtin_amdgcn_ballot_w64(bad) == 0) {
                    bin.lo = shfl_mirror<L>(bin.lo);
                    bin.hi = shfl_mirror<L>(bin.hi);
                    mirrored = true;
                }
            }
            if (!mirrored) bitonic_sort_lanes_p64<LC>(bin);
        } else if (q > 1) {
            bitonic_sort_lanes_p64<LC>(bin);
        }
        const int s = q * C + gl;
        if (gl < C && s < P) {
            const uint64_t r = slice[slot_of(s)];
            const uint64_t nb = p64_value(bin) + ((lag_max - (r >> sh)) << 6);                  // Main.java:265
            bin = p64_from(nb);
            slice[slot_of(s)] = ((uint64_t)(bin.lo & 63u) << 32) | ((uint32_t)r & pid_mask);
        }
    }
}

// ---- 3..5: greedy rounds over the sorted slice, outputs ------------------------------------------------------
// One wire element of the all-gather's narrow format (la_wire.hip): ((member rank + 1) << id_bits) | partition id.
__device__ __forceinline__ uint32_t wire_element(int32_t id, int32_t rank, int id_bits, uint32_t limit, bool& bad) {
    const uint32_t r1 = (uint32_t)rank + 1u;                                // -1 (no consumer) -> 0
    const uint32_t w = (r1 << id_bits) | (uint32_t)id;
    bad |= ((uint32_t)id >> id_bits) != 0 || r1 >= limit;
    return w;
}

template <int L, int E, bool FULL = false, bool WIRE = false, typename IDX>
__device__ __forceinline__ void assign_packed(const TileArgs& a, uint64_t* slice, int32_t* rank_tab, IDX p0,
                                              IDX c0, int P, int C, int gl, int sh, uint64_t lag_max,
                                              int32_t my_rank) {
    const uint32_t pid_mask = (uint32_t)((1ull << sh) - 1);

    if (gl < C) rank_tab[gl] = my_rank;
    wave_lds_fence();

    P64 bin = p64_from((gl < C) ? (uint64_t)gl : ~0ull);
    const int rounds = (C > 0) ? (P + C - 1) / C : 0;
    int max_rounds = __builtin_amdgcn_readfirstlane(wave_max_i32(rounds));
    if constexpr (kAblate == 1 || kAblate == 3) max_rounds = 0;
    if constexpr (kAblate >= 10) max_rounds = max_rounds < kAblate - 10 ? max_rounds : kAblate - 10;   // lab: cap the rounds
    // the widest consumer list among this wavefront's topics picks the network (wavefront-uniform)
    const int c_max = __builtin_amdgcn_readfirstlane(wave_max_i32(C));
    bool done = false;
    if constexpr (L > 4) if (!done && c_max <= 4) { greedy_rounds_tile<L, 4>(bin, slice, P, C, gl, sh, lag_max, pid_mask, max_rounds); done = true; }
    if constexpr (L > 8) if (!done && c_max <= 8) { greedy_rounds_tile<L, 8>(bin, slice, P, C, gl, sh, lag_max, pid_mask, max_rounds); done = true; }
    if constexpr (L > 16) if (!done && c_max <= 16) { greedy_rounds_tile<L, 16>(bin, slice, P, C, gl, sh, lag_max, pid_mask, max_rounds); done = true; }
    if constexpr (L > 32) if (!done && c_max <= 32) { greedy_rounds_tile<L, 32>(bin, slice, P, C, gl, sh, lag_max, pid_mask, max_rounds); done = true; }
    if (!done) greedy_rounds_tile<L, L>(bin, slice, P, C, gl, sh, lag_max, pid_mask, max_rounds);
    wave_lds_fence();

    // ---- 5. outputs ------------------------------------------------------------------------------------
    if (a.out_total && gl < C) a.out_total[c0 + (bin.lo & 63u)] = (int64_t)(p64_value(bin) >> 6);
    if constexpr (WIRE) {
        // LA_FLAG_WIRE_OUT: the (partition, member) pairs leave as wire elements -- 2 (or 4) bytes per partition instead of 8
        const int id_bits = a.wire_id_bits;
        const bool two = a.wire_bytes == 2;                                  // (kernel-uniform)
        const int rank_bits = (two ? 16 : 32) - id_bits;                      // bits of an element above the id: rank + 1 < 2^rank_bits
        const uint32_t limit = rank_bits >= 32 ? 0xFFFFFFFFu : (1u << rank_bits);
        bool bad = false;
        if constexpr (E >= 4) {
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                const int s0 = k * 4 * L + 4 * gl;
                uint32_t wv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint64_t w = slice[slot_of(s0 + i)];
                    bool b1 = false;
                    wv[i] = wire_element((int32_t)((uint32_t)w & pid_mask), (C > 0) ? rank_tab[(uint32_t)(w >> 32) & 63u] : -1, id_bits, limit,
                                         b1);
                    bad |= b1 && (FULL || s0 + i < P);                       // (slots past the topic hold no pair)
                }
                if (FULL || s0 + 3 < P) {
                    if (two) {
                        typedef uint32_t U32x2w __attribute__((ext_vector_type(2), aligned(2)));
                        const U32x2w v = {wv[0] | (wv[1] << 16), wv[2] | (wv[3] << 16)};
                        __builtin_nontemporal_store(v, reinterpret_cast<U32x2w*>(reinterpret_cast<uint16_t*>(a.out_wire) + p0 + s0));
                    } else {
                        typedef uint32_t U32x4w __attribute__((ext_vector_type(4), aligned(4)));
                        const U32x4w v = {wv[0], wv[1], wv[2], wv[3]};
                        __builtin_nontemporal_store(v, reinterpret_cast<U32x4w*>(reinterpret_cast<uint32_t*>(a.out_wire) + p0 + s0));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (s0 + i < P) {
                            if (two) reinterpret_cast<uint16_t*>(a.out_wire)[p0 + s0 + i] = (uint16_t)wv[i];
                            else reinterpret_cast<uint32_t*>(a.out_wire)[p0 + s0 + i] = wv[i];
                        }
                }
            }
        } else {
#pragma unroll
            for (int v = 0; v < E; ++v) {
                const int s = v * L + gl;
                if (s < P) {
                    const uint64_t w = slice[slot_of(s)];
                    const uint32_t x = wire_element((int32_t)((uint32_t)w & pid_mask), (C > 0) ? rank_tab[(uint32_t)(w >> 32) & 63u] : -1, id_bits,
                                                    limit, bad);
                    if (two) reinterpret_cast<uint16_t*>(a.out_wire)[p0 + s] = (uint16_t)x;
                    else reinterpret_cast<uint32_t*>(a.out_wire)[p0 + s] = x;
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, kStatusWire);
        return;
    }
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
            if (kAblate == 2 && (op[0] ^ om[1] ^ op[2] ^ om[3]) != 0x7FFFFFF1) continue;
            if (FULL || s0 + 3 < P) {
                // element-aligned 16-byte stores, non-temporal: the results are not read again by this launch
                // (measured: -1 % on the target batch; the same hint on the loads changes nothing)
                typedef int I32x4 __attribute__((ext_vector_type(4), aligned(4)));
                const I32x4 vp = {op[0], op[1], op[2], op[3]}, vm = {om[0], om[1], om[2], om[3]};
                __builtin_nontemporal_store(vp, reinterpret_cast<I32x4*>(a.out_pid + p0 + s0));
                __builtin_nontemporal_store(vm, reinterpret_cast<I32x4*>(a.out_rank + p0 + s0));
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
        // the network spans as many lanes as the widest consumer list among this wavefront's topics needs
        const int c_max = __builtin_amdgcn_readfirstlane(wave_max_i32(C));
        const int lc_w = c_max <= 1 ? 1 : 1 << (32 - __builtin_clz((unsigned)(c_max - 1)));
        for (int q = 0; q < max_rounds; ++q) {
            if (q > 0) bitonic_sort_lanes(bin, gl, lc_w);
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

// Tile -> topic descriptor of this lane's group.
// Tile -> topic descriptor of this lane's group.  Unconditional loads (clamped topic index); the words are
// fetched one step (fetch_desc) and interpreted another (make_desc), so a prefetched descriptor is not
// waited for where it is issued.
struct DescWords {
    int64_t p0, p1, c0, c1;
    bool exists;
};

template <int L, int E>
__device__ __forceinline__ DescWords fetch_desc(const TileArgs& a, int64_t t, int64_t n_tiles, int grp) {
    using Cfg = TileCfg<L, E>;
    const int64_t topic = t * Cfg::kGroupsPerWave + grp;
    DescWords w;
    w.exists = t < n_tiles && topic < a.n_topics;
    int64_t tc = w.exists ? topic : a.n_topics - 1;
    if (a.topic_list) tc = a.topic_list[tc];                           // uniform branch: null for plain batches
    w.p0 = a.part_off[tc]; w.p1 = a.part_off[tc + 1];
    w.c0 = a.cons_off[tc]; w.c1 = a.cons_off[tc + 1];
    return w;
}

template <int L, int E, typename IDX = int64_t>
__device__ __forceinline__ TopicDescT<IDX> make_desc(const TileArgs& a, const DescWords& w, int gl) {
    using Cfg = TileCfg<L, E>;
    TopicDescT<IDX> d;
    d.p0 = (IDX)w.p0;
    d.c0 = (IDX)w.c0;
    const int64_t Pl = w.p1 - w.p0, Cl = w.c1 - w.c0;
    const bool bad = Pl > Cfg::kCap || Cl > L || Pl < 0 || Cl < 0;
    // hint was wrong: leave outputs alone (kTileSkipOversize: the topic belongs to the block / large path)
    if (w.exists && bad && gl == 0 && !(a.flags & kTileSkipOversize)) atomicOr(a.status, kStatusShape);
    d.P = (w.exists && !bad) ? (int)Pl : 0;
    d.C = (w.exists && !bad) ? (int)Cl : 0;
    return d;
}

template <int L, int E, typename IDX = int64_t>
__device__ __forceinline__ TopicDescT<IDX> load_desc(const TileArgs& a, int64_t t, int64_t n_tiles, int grp, int gl) {
    return make_desc<L, E, IDX>(a, fetch_desc<L, E>(a, t, n_tiles, grp), gl);
}

// One tile of kernel 1, from the descriptor on (see the kernel below).
template <int L, int E, typename IDX, bool INLINE_WIDE, bool FULL, bool WIRE = false>
__device__ __forceinline__ void packed_tile(const TileArgs& a, const TopicDescT<IDX>& cur, uint64_t* slice, int32_t* rank_tab,
                                            int64_t tile, int gl, int lane) {
    using Cfg = TileCfg<L, E>;
    Raw<E> raw;
    issue_loads<L, E, FULL>(a, cur, gl, raw);
    // consumer ranks of the topic, used only at the very end: fetched with everything else
    int32_t my_rank = 0;
    if constexpr (kAblate != 2) {
        const IDX want = cur.c0 + (IDX)gl, last = (IDX)(a.k_total > 0 ? a.k_total - 1 : 0);
        if (a.k_total > 0) my_rank = load_at<int32_t>(a.cons_rank, want < last ? want : last);
    }
    issue_begin_loads<L, E, FULL>(a, cur, gl, raw);

    P64 rec[E];
    int sh, lbw;
    bool fits;
    uint64_t lag_max;
    int64_t lag[E];
    int32_t pid[E];
    {
        finish_lags<L, E, FULL>(a, cur, gl, raw, lag, pid);
        // can this wavefront's records be packed into 64 bits?  (empty slots hold lag 0, id 0)
        uint32_t id_or = 0;
        uint64_t lag_or = 0;
#pragma unroll
        for (int v = 0; v < E; ++v) { id_or |= (uint32_t)pid[v]; lag_or |= (uint64_t)lag[v]; }
        id_or = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_or_u32(id_or));
        sh = 32 - __builtin_clz(id_or | 1u);                             // 1..32 (32: a negative id)
        // bits of the wavefront's largest lag (64: a negative lag)
        const uint32_t hi = (uint32_t)(lag_or >> 32), lo = (uint32_t)lag_or;
        const int my_bits = hi ? 64 - __builtin_clz(hi) : (lo ? 32 - __builtin_clz(lo) : 0);
        lbw = __builtin_amdgcn_readfirstlane(wave_max_i32(my_bits));
        int lim = 63 - sh;
        if (lim > 57 - Cfg::kLog2Cap) lim = 57 - Cfg::kLog2Cap;
        fits = sh < 32 && lbw <= lim;                                    // wave-uniform
        lag_max = lbw >= 64 ? ~0ull : (((uint64_t)1 << lbw) - 1);
        if (fits) pack_records<L, E, FULL>(cur.P, gl, lag, pid, sh, lag_max, rec);
    }
    if (fits) {
        sort_into_slice<L, E, FULL>(slice, gl, rec, lbw, sh);
        assign_packed<L, E, FULL, WIRE>(a, slice, rank_tab, cur.p0, cur.c0, cur.P, cur.C, gl, sh, lag_max, my_rank);
    } else if constexpr (INLINE_WIDE) {
        assign_wide<L, E, false>(a, slice, (int64_t)cur.p0, (int64_t)cur.c0, cur.P, cur.C, gl, lag, pid);
    } else if (lane == 0) {
        if (a.flags & kTileNoDefer) atomicOr(a.status, kStatusBounds);      // the caller's bounds said this could not happen
        else a.defer_list[atomicAdd(a.defer_count, 1)] = (int32_t)tile;
    }
}

// ---- the tail of a small rebalance's one launch (TileTail, la_kernels.h) --------------------------------------------------
// Called by every thread of every workgroup of the single-launch form once its tiles' results are stored.  Release (the
// results, device- and host-visible), count the workgroup done; the LAST one acquires, builds every member's list from all
// workgroups' results (one workgroup's job below kSmallGroupN entries) and stores `done | status` into the host's word.
__device__ __forceinline__ void tile_tail(const TileTail& t, const uint32_t* status) {
    __shared__ uint32_t s_start[kTailGroupM];
    __shared__ int32_t s_in[7 * kSmallGroupN];                // ranks, ids, topics of the entries; the lists (group_small_body_staged)
    __shared__ uint32_t s_wsum[LA_WPB];
    __shared__ uint32_t s_turn, s_last;
    __threadfence_system();                                   // this thread's result stores (to HBM or into the host's arrays)
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t done = __hip_atomic_fetch_add(t.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = done == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;                                      // (workgroup-uniform)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // the other workgroups' results
    if (threadIdx.x == 0) __hip_atomic_store(t.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next call's count
    if (t.member_off)
        group_small_body_staged<kWave * LA_WPB, kTailGroupM>(t.n, t.n_members, t.n_topics, t.part_off, t.out_pid, t.out_rank,
                                                             t.member_off, t.grouped_topic, t.grouped_partition, nullptr, s_start,
                                                             s_wsum, &s_turn, s_in, s_in + kSmallGroupN, s_in + 2 * kSmallGroupN,
                                                             s_in + 3 * kSmallGroupN, s_in + 4 * kSmallGroupN,
                                                             reinterpret_cast<uint32_t*>(s_in + 5 * kSmallGroupN),
                                                             reinterpret_cast<uint32_t*>(s_in + 6 * kSmallGroupN));
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(t.fin_flag, 0x80000000u | st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- kernel 1: packed records -------------------------------------------------------------------------
// One tile per wavefront.  Loads (all issued back to back) -> lags -> format decision -> 32-bit key sort ->
// greedy rounds -> stores.  A tile whose records do not fit the packed format is appended to the deferred
// list and left to kernel 2 -- except in the INLINE_WIDE build, used when the whole batch is one round of
// resident workgroups: there occupancy does not matter, the wide code sits in the same kernel and the second
// launch (the larger part of a small batch's latency) disappears.
#ifdef LA_WPE   // lab hook: pin the packed kernel's wavefronts per SIMD (5..8 measured)
#define LA_WPE_ATTR __attribute__((amdgpu_waves_per_eu(LA_WPE, LA_WPE)))
#else
#define LA_WPE_ATTR
#endif
template <int L, int E, typename IDX, bool INLINE_WIDE, bool WIRE = false>
__global__ __launch_bounds__(256) LA_WPE_ATTR void wave_tile_packed_kernel(TileArgs a) {
    using Cfg = TileCfg<L, E>;
    __shared__ uint64_t lds[Cfg::kTopicsPerBlock * Cfg::kSlots];
    __shared__ int32_t rank_lds[Cfg::kTopicsPerBlock * L];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int gl = lane & (L - 1);            // lane within group
    const int grp = lane / L;                 // group within wave
    uint64_t* slice = lds + (wave * Cfg::kGroupsPerWave + grp) * Cfg::kSlots;
    int32_t* rank_tab = rank_lds + (wave * Cfg::kGroupsPerWave + grp) * L;

    const int64_t n_tiles = (a.n_topics + Cfg::kGroupsPerWave - 1) / Cfg::kGroupsPerWave;
    // one tile per wavefront: the grid covers all tiles (a resident-sized grid looping over tiles, with or
    // without the next tile's loads prefetched into registers, measured 10-25 % slower: more live
    // registers, fewer wavefronts per SIMD)
    const int64_t tile = (int64_t)blockIdx.x * Cfg::kWavesPerBlock + wave;
    // The single-launch form has no wide kernel behind it to zero the counter of the NEXT launch (the pair
    // alternates per launch, la_api.hip): do it here, or a later deferring launch would start counting at
    // whatever an earlier one left there and re-run a stale list.  Idle by stream order, like in the wide kernel.
    if constexpr (INLINE_WIDE) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *a.defer_count_next = 0;
        // the single-launch form may carry a tail (a small rebalance's lists and completion word): every wavefront stays to the end
        if (tile < n_tiles) {
            const TopicDescT<IDX> cur = load_desc<L, E, IDX>(a, tile, n_tiles, grp, gl);
            packed_tile<L, E, IDX, true, false>(a, cur, slice, rank_tab, tile, gl, lane);
        }
        if (a.tail.enabled) tile_tail(a.tail, a.status);                 // (kernel-uniform)
        return;
    }
    if (tile >= n_tiles) return;
    const TopicDescT<IDX> cur = load_desc<L, E, IDX>(a, tile, n_tiles, grp, gl);
    // every topic of this wavefront fills its tile exactly: the form without clamps, validity selects and sentinels
    // (the single-launch form of small batches keeps to the general one: it carries the wide code already)
    if constexpr (!INLINE_WIDE) {
        // (and lies inside the batch: the general form's clamps are also what keeps a bogus descriptor's loads in bounds)
        if (__builtin_amdgcn_ballot_w64(cur.P != Cfg::kCap || (int64_t)cur.p0 + Cfg::kCap > a.n_total) == 0) {
            packed_tile<L, E, IDX, false, true, WIRE>(a, cur, slice, rank_tab, tile, gl, lane);
            return;
        }
    }
    packed_tile<L, E, IDX, INLINE_WIDE, false, WIRE>(a, cur, slice, rank_tab, tile, gl, lane);
}

// ---- kernel 2: wide records (and the literal argmin form) ------------------------------------------------------
// Tiles come from the deferred list of kernel 1 (LA_ALGO_AUTO) or are all tiles (forced wide / argmin).
template <int L, int E, bool ARGMIN>
__global__ __launch_bounds__(256) void wave_tile_wide_kernel(TileArgs a, int from_list) {
    using Cfg = TileCfg<L, E>;
    __shared__ uint64_t lds[Cfg::kTopicsPerBlock * Cfg::kSlots];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int gl = lane & (L - 1);
    const int grp = lane / L;
    uint64_t* slice = lds + (wave * Cfg::kGroupsPerWave + grp) * Cfg::kSlots;

    const int64_t n_tiles = (a.n_topics + Cfg::kGroupsPerWave - 1) / Cfg::kGroupsPerWave;
    const int64_t n_waves = (int64_t)gridDim.x * Cfg::kWavesPerBlock;
    const int64_t count = from_list ? (int64_t)*a.defer_count : n_tiles;
    // the other counter of the pair is the next launch's: it is idle now (stream order), reset it here
    if (from_list && blockIdx.x == 0 && threadIdx.x == 0) *a.defer_count_next = 0;
    for (int64_t i = (int64_t)blockIdx.x * Cfg::kWavesPerBlock + wave; i < count; i += n_waves) {
        const int64_t tile = from_list ? (int64_t)a.defer_list[i] : i;
        const TopicDesc d = load_desc<L, E>(a, tile, n_tiles, grp, gl);
        Raw<E> raw;
        issue_loads<L, E>(a, d, gl, raw);
        issue_begin_loads<L, E>(a, d, gl, raw);
        int64_t lag[E];
        int32_t pid[E];
        finish_lags<L, E>(a, d, gl, raw, lag, pid);
        assign_wide<L, E, ARGMIN>(a, slice, d.p0, d.c0, d.P, d.C, gl, lag, pid);
        wave_lds_fence();
    }
}

// Grid = what is resident: CUs x (workgroups per CU the kernel's registers / LDS admit), at most one
// workgroup per four tiles.  Nothing depends on co-residency (no inter-workgroup communication); a
// smaller or larger grid only changes speed.
template <typename K>
static hipError_t resident_blocks(K kernel, int threads, int* out) {
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0)) != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    *out = cus * per_cu;
    return hipSuccess;
}

template <int L, int E>
static hipError_t launch_one(const TileArgs& a_in, int mode, hipStream_t stream, bool* tail_done) {
    TileArgs a = a_in;
    const bool want_tail = a.tail.enabled != 0;
    a.tail.enabled = 0;                                          // only the single-launch form below turns it back on
    using Cfg = TileCfg<L, E>;
    const int64_t blocks = (a.n_topics + Cfg::kTopicsPerBlock - 1) / Cfg::kTopicsPerBlock;
    if (blocks <= 0) return hipSuccess;
    // Resident workgroups per instantiation AND per device: every device this library accepts is gfx950, but two of them
    // need not expose the same number of CUs (partition modes).  Atomics: lanes of several shards may launch concurrently.
    struct Res { std::atomic<int> inl{0}, wide{0}, argmin{0}; };
    static Res s_res[33];                                        // device ids 0 .. 31; [32]: any other id, never cached
    hipError_t e;
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    const bool cached = dev >= 0 && dev < 32;
    Res& rs = s_res[cached ? dev : 32];
    std::atomic<int>&s_inline = rs.inl, &s_wide = rs.wide, &s_argmin = rs.argmin;
    int res_inline = cached ? s_inline.load(std::memory_order_acquire) : 0, res_wide = s_wide.load(std::memory_order_relaxed),
        res_argmin = s_argmin.load(std::memory_order_relaxed);
    if (res_inline == 0) {
        if ((e = resident_blocks(wave_tile_packed_kernel<L, E, uint32_t, true>, Cfg::kThreads, &res_inline)) != hipSuccess) return e;
        if ((e = resident_blocks(wave_tile_wide_kernel<L, E, false>, Cfg::kThreads, &res_wide)) != hipSuccess) return e;
        if ((e = resident_blocks(wave_tile_wide_kernel<L, E, true>, Cfg::kThreads, &res_argmin)) != hipSuccess) return e;
        if (cached) {
            s_wide.store(res_wide, std::memory_order_relaxed);
            s_argmin.store(res_argmin, std::memory_order_relaxed);
            s_inline.store(res_inline, std::memory_order_release);
        }
#ifdef LA_LAB
        printf("resident blocks: inline %d wide %d argmin %d (needed %lld)\n", res_inline, res_wide, res_argmin, (long long)blocks);
#endif
    }
    const dim3 b(Cfg::kThreads);
    auto grid = [&](int resident) { return dim3((unsigned)(blocks < resident ? blocks : resident)); };
    if (mode == kModeArgmin) {
        LA_LAUNCH((wave_tile_wide_kernel<L, E, true>), grid(res_argmin), b, 0, stream, a, 0);
    } else if (mode == kModeWide) {
        LA_LAUNCH((wave_tile_wide_kernel<L, E, false>), grid(res_wide), b, 0, stream, a, 0);
    } else {
        if (!a.defer_count || !a.defer_count_next || !a.defer_list) return hipErrorInvalidValue;
        // 32-bit indexing when every byte offset (8-byte arrays) fits 32 bits
        const bool idx32 = a.n_total < ((int64_t)1 << 29) && a.k_total < ((int64_t)1 << 30) && !(a.flags & 1);
        if (idx32 && blocks <= res_inline && !(a.flags & 2) && !(a.flags & kTileWireOut)) {
            // the whole batch is resident at once: one kernel with the wide code inline, no second launch
            if (want_tail && tail_done) {
                a.tail.enabled = 1;
                *tail_done = true;
            }
            LA_LAUNCH((wave_tile_packed_kernel<L, E, uint32_t, true>), dim3((unsigned)blocks), b, 0, stream, a);
            return hipGetLastError();
        }
        if (a.flags & kTileWireOut) {
            // results in the wire format: only the form whose bounds prove that every tile packs (no wide-record code has the
            // wire stores) and whose indices fit 32 bits -- la_api.hip checks both before it sets the flag
            if (!idx32 || !(a.flags & kTileNoDefer) || !a.out_wire) return hipErrorInvalidValue;
            LA_LAUNCH((wave_tile_packed_kernel<L, E, uint32_t, false, true>), dim3((unsigned)blocks), b, 0, stream, a);
            return hipGetLastError();
        }
        if (idx32)
            LA_LAUNCH((wave_tile_packed_kernel<L, E, uint32_t, false>), dim3((unsigned)blocks), b, 0, stream, a);
        else
            LA_LAUNCH((wave_tile_packed_kernel<L, E, int64_t, false>), dim3((unsigned)blocks), b, 0, stream, a);
        if (a.flags & kTileNoDefer) return hipGetLastError();      // proven: nothing can be deferred, no second launch
        // usually nothing was deferred: every wavefront reads the count and leaves
#ifdef LA_LAB
        if (getenv("LA_NO_WIDE")) return hipGetLastError();
#endif
        LA_LAUNCH((wave_tile_wide_kernel<L, E, false>), grid(res_wide), b, 0, stream, a, 1);
    }
    return hipGetLastError();
}

template <int L>
static inline hipError_t launch_l(int e, const TileArgs& a, int mode, hipStream_t stream, bool* tail_done) {
    switch (e) {
        case 1: return launch_one<L, 1>(a, mode, stream, tail_done);
        case 2: return launch_one<L, 2>(a, mode, stream, tail_done);
        case 4: return launch_one<L, 4>(a, mode, stream, tail_done);
        case 8: return launch_one<L, 8>(a, mode, stream, tail_done);
        default: return launch_one<L, 16>(a, mode, stream, tail_done);
    }
}

}  // namespace la
