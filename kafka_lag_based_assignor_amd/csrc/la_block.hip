// la_block.hip -- the block path: one workgroup per topic, for topics beyond a wave tile (more than 1 024
// partitions or more than 64 consumers) up to 8 192 partitions x 2 048 consumers, or 16 384 x 1 024.  All such
// topics of a batch run side by side (one launch per size class over a topic list), where the large path
// (la_large.hip) would take them one after another through a dozen device-wide kernels each.
//
// Per topic:
//   1. loads, lag fused in (computePartitionLag, Main.java:376-404), straight into the sort's registers;
//   2. bitonic sort by (lag descending, partition ascending) (Main.java:228-235): E records per thread; distances
//      below E in registers, inside a wavefront through DPP / v_permlane*_swap, across wavefronts through LDS.
//      Records that fit one 64-bit word -- ((lag_max - lag) << sh) | id, decided per topic by a workgroup-wide OR
//      -- run through the instruction-level networks of la_sort64.h (block_sort_packed), the rest as (key64, id32)
//      records (block_sort_regs);
//   3. sorted keys to LDS by position, partition ids to out_partition from the registers;
//   4. greedy in rounds of C partitions: the bins are sorted ascending by (total, position in the rank-sorted
//      consumer list) and the k-th partition of the round goes to the k-th bin (Main.java:237-266; the
//      comparator's first key, the assigned count, is what makes rounds -- SURVEY.md section 8a note 5).
//      Up to 256 consumers: ONE wavefront, bins in registers, packed 64-bit words when no total can overflow
//      them (greedy_one_wave_packed) else 96-bit records (greedy_one_wave).  More consumers: bins in LDS, sorted
//      by the direction-free network (first step of a merge is the mirror i <-> i ^ (K-1), the rest i <-> i ^ j,
//      the lower slot always keeps the smaller record), which has two useful properties:
//        * slots >= the live count behave as +inf that never moves (a compare-exchange never lowers the larger
//          record), so they are neither stored nor visited: no padding, any C;
//        * a step whose pairs stay inside 128-slot spans is executed by the span's owner wavefront, so runs of
//          such steps need only a wavefront-level fence; the workgroup barrier is paid by the few wider steps.
#include "la_device.h"
#include "la_kernels.h"
#include "la_sort64.h"
#include "la_sort32.h"
#include "la_sort32_net.h"

namespace la {

namespace {

#ifdef LA_BLOCK_CLOCKS   // development build: thread 0 of workgroup 0 accumulates the time of every stage (tools/block_probe.py --clocks)
__device__ unsigned long long g_block_clocks[16];     // 0-5 stages, 6-7 key32 rounds, 8-12 phases of a digit pass
#define LA_BCLK(i)                                                         \
    do {                                                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                         \
            const unsigned long long now_ = wall_clock64();                \
            g_block_clocks[i] += now_ - bclk_;                             \
            bclk_ = now_;                                                  \
        }                                                                  \
    } while (0)
#define LA_BCLK_START unsigned long long bclk_ = wall_clock64()
#define LA_RCLK(i)                                                         \
    do {                                                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                         \
            const unsigned long long now_ = wall_clock64();                \
            g_block_clocks[i] += now_ - rclk_;                             \
            rclk_ = now_;                                                  \
        }                                                                  \
    } while (0)
#define LA_RCLK_START unsigned long long rclk_ = wall_clock64()
#else
#define LA_BCLK(i) do {} while (0)
#define LA_BCLK_START do {} while (0)
#define LA_RCLK(i) do {} while (0)
#define LA_RCLK_START do {} while (0)
#endif

constexpr int kSpan = 128;      // slots a wavefront owns: 64 pairs

// n: power of two >= live.  cmpx(i, p) with i < p leaves the smaller record at i.
// entry_fence_only: the data was last written by the span owners (so an in-span first step needs no barrier).
template <typename F>
__device__ __forceinline__ void lds_bitonic_sort(int n, int live, int tid, int nt, bool entry_fence_only, F cmpx) {
    bool prev_cross = !entry_fence_only;
    const int half = n >> 1;
    for (int K = 2; K <= n; K <<= 1) {
        for (int j = K >> 1; j >= 1; j >>= 1) {
            const bool mirror = (j == (K >> 1));
            const bool cross = 2 * j > kSpan;
            if (prev_cross || cross) __syncthreads(); else wave_lds_fence();
            const int flip = mirror ? (K - 1) : j;
            for (int idx = tid; idx < half; idx += nt) {
                const int off = idx & (j - 1);
                const int i = ((idx - off) << 1) | off;
                const int p = i ^ flip;
                if (p < live) cmpx(i, p);
            }
            prev_cross = cross;
        }
    }
}

__device__ __forceinline__ int pow2ceil_dev(int x) {
    return x <= 1 ? 1 : 1 << (32 - __builtin_clz((unsigned)(x - 1)));
}

// ---- partition sort: E records per thread in registers (8, or 16 in the largest class), slot = tid*E + r --------
// Classic bitonic network over n_eff = pow2ceil(P) slots (slots >= P hold an all-ones sentinel; a record equal
// to it carries the same bits, so which of the two lands where does not matter).  Distances below E stay in
// registers, distances inside a wavefront go through DPP / permlane moves, only distances >= 64*E cross
// wavefronts through LDS (10 of the 91 steps at 8 192 slots).  Wavefronts wholly beyond n_eff only keep the
// barriers company.
constexpr int kZeroSlotBytes = 16;    // region A's tail: the explicit zero slot s_key[np_cap] (see block_topic_kernel)
constexpr int kXchg = 8;              // records per thread that cross wavefronts in one LDS exchange

template <int JL, int E>
__device__ __forceinline__ void sort_lane_step(Rec (&rec)[E], int tid, int k) {
    if (JL * E < k) {
        const bool keep_min = (((tid & JL) == 0) == ((tid & (k / E)) == 0));
#pragma unroll
        for (int r = 0; r < E; ++r) cmpx_lanes<JL>(rec[r], keep_min);
    }
}

template <int J, int E>
__device__ __forceinline__ void sort_reg_step(Rec (&rec)[E], int tid, int k) {
    if constexpr (J >= 1 && J < E) {
        if (J < k) {
#pragma unroll
            for (int r = 0; r < E; ++r) {
                if ((r & J) == 0) {
                    const bool asc = k < E ? ((r & k) == 0) : ((tid & (k / E)) == 0);
                    cmpx_regs(rec[r], rec[r | J], asc);
                }
            }
        }
    }
}

// x_key / x_id: exchange area of kXchg * blockDim slots (register-major: slot r of thread t at r * blockDim + t,
// so consecutive lanes touch consecutive words); with 16 records per thread the exchange runs in two halves.
template <int E>
__device__ __forceinline__ void block_sort_regs(Rec (&rec)[E], int n_eff, int tid, int nt, uint64_t* x_key,
                                                uint32_t* x_id) {
    const bool active = (tid & ~(kWave - 1)) * E < n_eff;                // wavefront-uniform
    for (int k = 2; k <= n_eff; k <<= 1) {
        for (int j = k >> 1; j >= kWave * E; j >>= 1) {                   // across wavefronts: LDS
            const int pt = tid ^ (j / E);
            const bool keep_min = (((tid & (j / E)) == 0) == ((tid & (k / E)) == 0));
#pragma unroll
            for (int h = 0; h < E; h += kXchg) {
                if (active) {
#pragma unroll
                    for (int r = 0; r < kXchg; ++r) {
                        x_key[r * nt + tid] = ((uint64_t)rec[h + r].hi << 32) | rec[h + r].lo;
                        x_id[r * nt + tid] = rec[h + r].tb;
                    }
                }
                __syncthreads();
                if (active) {
#pragma unroll
                    for (int r = 0; r < kXchg; ++r) {
                        const uint64_t ok = x_key[r * nt + pt];
                        Rec o;
                        o.hi = (uint32_t)(ok >> 32); o.lo = (uint32_t)ok; o.tb = x_id[r * nt + pt];
                        Rec& mine = rec[h + r];
                        const bool take = (rec_less(o, mine) == keep_min);
                        mine.hi = take ? o.hi : mine.hi;
                        mine.lo = take ? o.lo : mine.lo;
                        mine.tb = take ? o.tb : mine.tb;
                    }
                }
                __syncthreads();
            }
        }
        if (active) {
            sort_lane_step<32, E>(rec, tid, k);
            sort_lane_step<16, E>(rec, tid, k);
            sort_lane_step<8, E>(rec, tid, k);
            sort_lane_step<4, E>(rec, tid, k);
            sort_lane_step<2, E>(rec, tid, k);
            sort_lane_step<1, E>(rec, tid, k);
            sort_reg_step<8, E>(rec, tid, k);
            sort_reg_step<4, E>(rec, tid, k);
            sort_reg_step<2, E>(rec, tid, k);
            sort_reg_step<1, E>(rec, tid, k);
        }
    }
}

// ---- the same sort on packed records ((lag_max - lag) << sh) | id -- one 64-bit word, ascending == (lag desc,
// id asc) -- through the instruction-level networks of la_sort64.h: 4 VALU per compare-exchange step instead of
// ~14.  Direction-free form (every block comes out ascending), so stopping at n_eff slots works here too.  The
// workgroup decides per topic whether its records fit (no negative lag or id, lag bits + id bits <= 63).
template <int E>
__device__ __forceinline__ void block_sort_packed(P64 (&rec)[E], int n_eff, int live, int tid, int nt, uint64_t* x_key) {
    // `live`: slots at or beyond it hold sentinels (all equal, larger than any record), and keep holding them: a block of
    // slots that starts there is in order whatever is done to it, and a merge whose upper half starts there changes nothing.
    constexpr int kSpanSlots = kWave * E;                                // slots of one wavefront
    const int w0 = (tid & ~(kWave - 1)) * E;                             // my wavefront's first slot
    const bool in_range = w0 < n_eff;                                    // wavefront-uniform
    bool active = in_range && w0 < live;
    if (active) {
#pragma unroll
        for (int r = 0; r < E; ++r) asm volatile("s_nop 1" : "+v"(rec[r].lo), "+v"(rec[r].hi));
        bitonic_sort_tile_p64<kWave, E>(rec);
    }
    for (int K = 2 * kSpanSlots; K <= n_eff; K <<= 1) {
        active = in_range && (w0 & ~(K - 1)) + (K >> 1) < live;          // (all wavefronts of a block of K slots agree)
        for (int j = K >> 1; j >= kSpanSlots; j >>= 1) {
            // first step of a merge: i <-> i ^ (K-1) (mirror); the rest: i <-> i ^ j; the lower slot keeps the min
            const int mask = (j == (K >> 1)) ? (K - 1) : j;
            if (active) {
#pragma unroll
                for (int r = 0; r < E; ++r) x_key[r * nt + tid] = p64_value(rec[r]);
            }
            __syncthreads();
            if (active) {
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const int i = tid * E + r;
                    const int pi = i ^ mask;
                    const uint64_t o = x_key[(pi % E) * nt + pi / E], x = p64_value(rec[r]);
                    const bool keep_min = (i & j) == 0;
                    const uint64_t lo = o < x ? o : x, hi = o < x ? x : o;
                    rec[r] = p64_from(keep_min ? lo : hi);
                }
            }
            __syncthreads();
        }
        if (active) {
#pragma unroll
            for (int r = 0; r < E; ++r) asm volatile("s_nop 1" : "+v"(rec[r].lo), "+v"(rec[r].hi));
            clean_p64<kWave, E, kSpanSlots / 2, false>(rec);              // i <-> i ^ j for every j inside the wavefront
        }
    }
}

// ---- the packed records once more, by digits: an LSD radix sort inside the workgroup -------------------------------------
// The network above costs 8 (16) records x ~5 VALU x 91 (105) steps per thread whatever the keys are; the records of a topic
// have lbw + sh significant bits -- 47 for 34-bit lags and 8 191 ids -- i.e. six 8-bit digits.  Per digit, with the records in
// registers: every wavefront counts its digits in its own 256 counters with RETURNING LDS atomics, whose old values are the
// record's rank among the wavefront's equal digits (lanes of one instruction are served in lane order: the property
// la_create tests, LA_FEATURE_ATOMIC_RANK; a wavefront issues its E instructions in order) -> a barrier -> thread d turns the
// counters of digit d into the wavefronts' first places inside the digit and leaves the digit's total -> a barrier -> one
// wavefront scans the 256 totals -> a barrier -> every record goes to (digit's first place + its wavefront's + its rank) in
// region A -> a barrier -> the wavefronts read the result back blocked by wavefront, register-major (consecutive lanes,
// consecutive words; the order the next digit's ranks are taken in).  The last digit's records stay in region A, by position:
// the caller reads record i and writes the key of position i into the same word.
// Only the wavefronts below `live` slots take part (the others hold sentinels: they would sort behind everything); the
// sentinels inside them (all ones) have digit 255 in every pass.  hist: [live / (64 E)][256] counters (the bins' area, unused
// until the sort is done), aux: [2][256].
template <int E>
__device__ __forceinline__ void block_sort_radix(P64 (&rec)[E], int live, int bits, int tid, int nt, uint64_t* buf,
                                                 uint32_t* hist, uint32_t* aux, int first_shift = 0) {
    constexpr int kSpanSlots = kWave * E;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = (live + kSpanSlots - 1) / kSpanSlots;                 // wavefronts that hold records (a small topic: part of one)
    const bool act = wave < nw;                                          // wavefront-uniform
    uint32_t* mine = hist + wave * 256;
    uint32_t* total = aux;                                               // [256] records per digit, then first place of the digit
    if (act) {
#pragma unroll
        for (int k = 0; k < 4; ++k) mine[k * kWave + lane] = 0;
    }
    wave_lds_fence();
    LA_RCLK_START;
    for (int shift = first_shift; shift < bits; shift += 8) {
        uint32_t old[E];
        if (act) {
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint32_t d = (uint32_t)(p64_value(rec[r]) >> shift) & 255u;
                // the lanes that share lane 0's digit take their ranks from ONE atomic (skewed lags: the high digits of nearly
                // every record are equal, and 64 lanes on one counter are 64 turns); the others as before, after it
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                const uint64_t grp = __builtin_amdgcn_ballot_w64(d == d0);
                const int cnt = __builtin_popcountll(grp);
                if (cnt >= 8) {                                              // (wavefront-uniform)
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&mine[d0], (uint32_t)cnt);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    const uint32_t before = (uint32_t)__builtin_popcountll(grp & ((1ull << lane) - 1ull));
                    uint32_t o = base + before;
                    if (d != d0) o = atomicAdd(&mine[d], 1u);
                    old[r] = o;
                } else {
                    old[r] = atomicAdd(&mine[d], 1u);
                }
            }
        }
        __syncthreads();
        LA_RCLK(8);                                                      // counts (+ the barrier)
        for (int d = tid; d < 256; d += nt) {                            // digit d: first places of the wavefronts inside it
            uint32_t run = 0;
            for (int w = 0; w < nw; ++w) {
                const uint32_t c = hist[w * 256 + d];
                hist[w * 256 + d] = run;
                run += c;
            }
            total[d] = run;
        }
        __syncthreads();
        LA_RCLK(9);                                                      // per-digit prefix over the wavefronts
        if (wave == 0) {                                                 // exclusive scan of the 256 totals, four per lane
            const uint4 t = reinterpret_cast<const uint4*>(total)[lane];
            const uint32_t s = t.x + t.y + t.z + t.w;
            const uint32_t excl = wave_incl_scan_u32(s) - s;
            __builtin_amdgcn_wave_barrier();
            reinterpret_cast<uint4*>(total)[lane] = make_uint4(excl, excl + t.x, excl + t.x + t.y, excl + t.x + t.y + t.z);
        }
        __syncthreads();
        LA_RCLK(10);                                                     // scan of the totals
        if (act) {
            // this wavefront's first place of every digit = the digit's first place + the wavefront's inside it: folded into the
            // wavefront's own table once (four entries per lane), so that a record's place costs ONE table read, not two
            {
                const uint4 t = reinterpret_cast<const uint4*>(total)[lane];
                uint4 m = reinterpret_cast<uint4*>(mine)[lane];
                m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
                reinterpret_cast<uint4*>(mine)[lane] = m;
            }
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint32_t d = (uint32_t)(p64_value(rec[r]) >> shift) & 255u;
                buf[mine[d] + old[r]] = p64_value(rec[r]);
            }
        }
        __syncthreads();
        LA_RCLK(11);                                                     // scatter
        if (act && shift + 8 < bits) {
#pragma unroll
            for (int k = 0; k < 4; ++k) mine[k * kWave + lane] = 0;       // (read above, before the barrier)
#pragma unroll
            for (int r = 0; r < E; ++r) rec[r] = p64_from(buf[wave * kSpanSlots + r * kWave + lane]);
            // (the next scatter into buf comes three barriers later)
        }
        LA_RCLK(12);                                                     // read back
    }
}

// Greedy rounds for up to 256 consumers: one wavefront, bins in registers (EC per lane, slot = lane*EC + r),
// sorted by the DPP / permlane networks of la_device.h -- no LDS traffic and no barrier between rounds.
// Slots >= C hold an all-ones sentinel (a real bin's index is < C, so it never equals it).  L = lanes in use
// (EC == 1: the network is unrolled for that width).
template <int EC, int L>
__device__ __forceinline__ void greedy_one_wave(const BlockArgs& a, const uint64_t* s_key, const int32_t* s_rank,
                                                int64_t p0, int64_t c0, int P, int C, int lane) {
    Rec bin[EC];
#pragma unroll
    for (int r = 0; r < EC; ++r) {
        const int e = lane * EC + r;
        if (e < C) { bin[r].hi = (uint32_t)(kTotalBias >> 32); bin[r].lo = 0; bin[r].tb = (uint32_t)e; }
        else bin[r].hi = bin[r].lo = bin[r].tb = 0xFFFFFFFFu;
    }
    const int rounds = (P + C - 1) / C;
    uint64_t lag[EC];                                    // this round's lags, read before the bins are sorted
#pragma unroll
    for (int r = 0; r < EC; ++r) {
        const int e = lane * EC + r;
        lag[r] = (e < C && e < P) ? (s_key[e] ^ kLagKeyFlip) : 0;
    }
    for (int q = 0; q < rounds; ++q) {
        uint64_t next[EC];
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const int e = lane * EC + r;
            const int s = (q + 1) * C + e;
            next[r] = (e < C && s < P) ? (s_key[s] ^ kLagKeyFlip) : 0;
        }
        if (q > 0) {
            bitonic_sort_tile<L, EC>(bin, lane & (L - 1));      // lanes >= L sort sentinels among themselves
        }
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const int e = lane * EC + r;
            const int s = q * C + e;
            if (e < C && s < P) {
                const uint64_t t = (((uint64_t)bin[r].hi << 32) | bin[r].lo) + lag[r];   // Main.java:265
                bin[r].hi = (uint32_t)(t >> 32);
                bin[r].lo = (uint32_t)t;
                a.out_rank[p0 + s] = s_rank[bin[r].tb];
            }
            lag[r] = next[r];
        }
    }
    if (a.out_total) {
#pragma unroll
        for (int r = 0; r < EC; ++r)
            if (bin[r].tb < (uint32_t)C)
                a.out_total[c0 + bin[r].tb] = (int64_t)((((uint64_t)bin[r].hi << 32) | bin[r].lo) ^ kTotalBias);
    }
}

// The same with packed bins (total << idx_bits) | index -- one 64-bit word, sorted by the instruction-level
// networks of la_sort64.h (4 VALU per compare-exchange step instead of ~14 for the 96-bit records).  Usable when
// no lag is negative and no total can reach 2^62; the caller checks that on the sorted keys.
template <int EC, int L>
__device__ __forceinline__ void greedy_one_wave_packed(const BlockArgs& a, const uint64_t* s_key, const int32_t* s_rank,
                                                       int64_t p0, int64_t c0, int P, int C, int idx_bits, int lane) {
    // The chain of rounds is: sort the bins -> add the round's lags -> sort ...  Everything else is kept OFF that chain:
    //  * LDS reads are unconditional (index clamped; a branch around a read makes the wavefront wait for it on the
    //    spot) and issued a round ahead: next round's lags, and the member ranks of LAST round's winners;
    //  * a round's ranks are stored one round late, after the sort that hid their lookup.
    const uint32_t idx_mask = (1u << idx_bits) - 1;
    P64 bin[EC];
    uint64_t lag[EC];                                    // raw sorted keys of the round's partitions
    uint32_t won[EC];                                    // consumer position that took slot r's partition last round
    bool had[EC];
    const int last = P > 0 ? P - 1 : 0;                 // clamp for the reads that run past the topic
#pragma unroll
    for (int r = 0; r < EC; ++r) {
        const int e = lane * EC + r;
        bin[r] = p64_from(e < C ? (uint64_t)e : kRoundSentinel);    // below 2^63: the steps' VALU form (la_sort64.h)
        lag[r] = s_key[e < P ? e : last];
        won[r] = 0;
        had[r] = false;
    }
    const int rounds = (P + C - 1) / C;
    for (int q = 0; q < rounds; ++q) {
        uint64_t next[EC];
        int32_t rk[EC];
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const int s = (q + 1) * C + lane * EC + r;
            next[r] = s_key[s < P ? s : last];
            rk[r] = s_rank[won[r]];
        }
        if (q > 0) {
            // the networks read these registers through DPP: 2 wait states after their last compiler-generated write
#pragma unroll
            for (int r = 0; r < EC; ++r) asm volatile("s_nop 1" : "+v"(bin[r].lo), "+v"(bin[r].hi));
            bitonic_sort_tile_p64<L, EC, true>(bin);
        }
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const int e = lane * EC + r;
            const int s = q * C + e;
            if (had[r]) a.out_rank[p0 + s - C] = rk[r];                                    // last round's assignment
            had[r] = e < C && s < P;
            if (had[r]) {
                const uint64_t nb = p64_value(bin[r]) + ((lag[r] ^ kLagKeyFlip) << idx_bits);   // Main.java:265
                bin[r] = p64_from(nb);
                won[r] = (uint32_t)nb & idx_mask;
            }
            lag[r] = next[r];
        }
    }
#pragma unroll
    for (int r = 0; r < EC; ++r)
        if (had[r]) a.out_rank[p0 + (rounds - 1) * C + lane * EC + r] = s_rank[won[r]];
    if (a.out_total) {
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const uint64_t v = p64_value(bin[r]);
            if (v != kRoundSentinel) a.out_total[c0 + ((uint32_t)v & idx_mask)] = (int64_t)(v >> idx_bits);
        }
    }
}

// ---- greedy rounds for 65 .. 256 consumers on ONE wavefront, the bins ordered through 32-bit keys -----------------------------
// The chain of ceil(P / C) dependent rounds is what a single-topic call waits for (BASELINE config 2: 10 000 x 128, 79 rounds),
// and a round of the forms above is a full network over 64-bit bins: 28 steps x ~8 issue slots for 128 bins, on two wavefronts
// with an LDS exchange and two barriers, 0.93 us.  Here the bins do not move at all:
//   * every bin has a HOME -- consumer position e in register e % EC of lane e / EC -- where its total lives and where its key
//         key = (((total - floor) >> drop) << idx_bits) | e
//     is formed; the keys (any order going in) are sorted by ONE generated asm statement of 2-VALU steps (a DPP move +
//     v_med3_u32; la_sort32_net.h, tools/gen_sort32_net.py: 155 issue slots for 128 keys, 313 for 256 -- strung together from
//     la_sort32.h's per-step blocks the same sort came out at ~300 / ~450 with the pads each block carries for its worst case,
//     and the chain of rounds of a lone wavefront is a count of issue slots);
//   * sorted position i knows which bin stands there (the key's low bits): it hands that bin the lag of the round's i-th
//     partition through LDS -- a PERMUTATION write (every slot of `hand` written exactly once: idle positions and idle bins
//     pair up with lag 0), read back linearly by the homes -- and leaves the winner's position in the low word of the
//     partition's slot (member ranks and the global stores are the workgroup's business after the last round, coalesced);
//   * home: total += lag (Main.java:265), next key.  (The first form of this function kept the totals in LDS, gathered and
//     scattered by the sorted positions: two random 64-bit accesses and a read -> add -> write chain per round -- 0.43 us of a
//     0.71 us round, 1.9 us with the idle bins of a 200-consumer topic all hitting one slot.)
// Why 32 bits are enough.  Lags are non-negative here (the packed condition), every bin takes exactly one partition per full
// round, and the rounds are consecutive ranges of the DESCENDING lag list: with floor_q = the sum of the smallest lag of every
// full round so far (the round's last partition; wavefront-uniform, one broadcast LDS read per round off the chain),
//     0 <= total_e - floor_q = sum over rounds (lag_e,r - min_r) <= sum over rounds (max_r - min_r) <= lag_first - lag_last
// i.e. lag_bits bits, of which the key keeps the top 31 - idx_bits.  The truncation is monotone: when no two neighbours of the
// sorted keys share their truncated total, the order IS the order of (total, e) -- (assigned lag, memberId) of Main.java:253-259.
// When some do (ties of the full totals included) and bits were dropped, that one round is ordered again by the exact 64-bit
// network on (total << idx_bits) | e; with drop == 0 the key is exact and e breaks the ties as the reference does.  Uniform
// 40-bit lags over 128 bins: a shared truncated total about once in 8 000 rounds.
template <int EC>
__device__ __forceinline__ void greedy_one_wave_key32(const BlockArgs& a, uint64_t* s_key, uint64_t* hand, uint32_t* scratch, int64_t c0,
                                                      int P_in, int C_in, int idx_bits_in, int lag_bits_in, int lane) {
    constexpr uint32_t kIdle = 0x80000000u;                                            // an idle bin's key: | e, above every real key
    // wavefront-uniform by construction; said so to the compiler (values that came through LDS look divergent to it, and a
    // "divergent" shift count or loop bound turns into exec-mask branches around every use)
    const int P = __builtin_amdgcn_readfirstlane(P_in), C = __builtin_amdgcn_readfirstlane(C_in);
    const int idx_bits = __builtin_amdgcn_readfirstlane(idx_bits_in), lag_bits = __builtin_amdgcn_readfirstlane(lag_bits_in);
    const uint32_t idx_mask = (1u << idx_bits) - 1;
    const int keep = 31 - idx_bits;
    const int drop = lag_bits > keep ? lag_bits - keep : 0;
    const uint32_t tie_lim = 1u << idx_bits;
    const int rounds = (P + C - 1) / C;
    uint32_t dirs[6];
    sort_net_dirs(lane, dirs);
    uint32_t key[EC];
    uint64_t tot[EC];                                                                  // of the bins whose home this lane is
    bool mine[EC];
#pragma unroll
    for (int r = 0; r < EC; ++r) {
        const int e = lane * EC + r;
        mine[r] = e < C;
        key[r] = (mine[r] ? 0u : kIdle) | (uint32_t)e;                                 // round 0: totals 0, positions ascending
        tot[r] = 0;
    }
    uint64_t floor_sum = 0;                                                            // (the same value in every lane)
    // slot P holds "lag 0" for idle positions and rounds past the topic; the winner words nobody reads go to scratch[i], a
    // word per position (one shared dummy word would make every idle position of a round hit the same LDS address)
    char* const key_bytes = reinterpret_cast<char*>(s_key);
    uint64_t lag[EC];
    uint32_t won_at[EC], idle_at[EC];                                                  // byte offsets from s_key
#pragma unroll
    for (int r = 0; r < EC; ++r) {
        const int i = lane * EC + r;
        const bool live = i < C && i < P;
        idle_at[r] = (uint32_t)(reinterpret_cast<char*>(scratch + i) - key_bytes);
        lag[r] = s_key[live ? i : P] ^ kLagKeyFlip;
        won_at[r] = live ? (uint32_t)i * 8u : idle_at[r];
    }
    uint64_t round_min = s_key[C <= P ? C - 1 : P] ^ kLagKeyFlip;                      // a partial round adds nothing to the floor
    for (int q = 0; q < rounds; ++q) {
        uint64_t next[EC];
        uint32_t next_at[EC];
#pragma unroll
        for (int r = 0; r < EC; ++r) {                                                 // next round's lags: off the chain
            const int i = lane * EC + r, g = (q + 1) * C + i;
            const bool live = i < C && g < P;
            next[r] = s_key[live ? g : P] ^ kLagKeyFlip;
            next_at[r] = live ? (uint32_t)g * 8u : idle_at[r];
        }
        const int last_next = (q + 2) * C - 1;
        const uint64_t next_min = s_key[last_next < P ? last_next : P] ^ kLagKeyFlip;  // (slot P: "lag 0"; one broadcast read)
#ifdef LA_BLOCK_CLOCKS
        const unsigned long long kclk0 = wall_clock64();
#endif
        if (q > 0) {
            if constexpr (EC == 2) sort_net_u32_e2(key, dirs);
            else sort_net_u32_e4(key, dirs);
        }
#ifdef LA_BLOCK_CLOCKS
        asm volatile("" : "+v"(key[0]));
        const unsigned long long kclk1 = wall_clock64();
#endif
        uint32_t e[EC];
#pragma unroll
        for (int r = 0; r < EC; ++r) e[r] = key[r] & idx_mask;
        // sorted position i = lane * EC + r: its partition's lag to the bin that stands there, the bin's position to the partition.
        // Written BEFORE the order is known to be exact: every slot of `hand` and every live winner word is written by every
        // round's hand-over, so the rare re-ordered round simply writes them again -- and the tie check below (a ballot and a
        // branch: a VALU -> SALU hop) runs while these stores are on their way instead of in front of them.
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            hand[e[r]] = lag[r];
            *reinterpret_cast<uint32_t*>(key_bytes + won_at[r]) = e[r];
        }
        if (q > 0 && drop > 0) {
            // neighbours that share their truncated total: the keys cannot tell which of the two bins is smaller.  (Idle bins
            // differ from every real key in the top bit and from each other nowhere above the position: they sit behind
            // the C real keys, where a "tie" among them is of no consequence -- but must not be reported: compare real keys only.)
            bool tie = false;
#pragma unroll
            for (int r = 0; r + 1 < EC; ++r) tie |= (key[r] ^ key[r + 1]) < tie_lim && (int32_t)key[r + 1] >= 0;
            asm volatile("s_nop 1" : "+v"(key[0]));                    // (written by the network's last block: a DPP read follows)
            const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp((int)kIdle, (int)key[0], 0x130, 0xF, 0xF, false);   // wave_shl:1
            tie |= (key[EC - 1] ^ up) < tie_lim && (int32_t)up >= 0;
            if (__builtin_amdgcn_ballot_w64(tie) != 0) {                               // rare: this round by the exact network
                P64 bin[EC];
#pragma unroll
                for (int r = 0; r < EC; ++r)
                    bin[r] = p64_from(mine[r] ? ((tot[r] << idx_bits) | (uint32_t)(lane * EC + r)) : (kRoundSentinel - idx_mask + (uint32_t)(lane * EC + r)));
#pragma unroll
                for (int r = 0; r < EC; ++r) asm volatile("s_nop 1" : "+v"(bin[r].lo), "+v"(bin[r].hi));
                bitonic_sort_tile_p64<kWave, EC, true>(bin);
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < EC; ++r) {
                    e[r] = bin[r].lo & idx_mask;
                    hand[e[r]] = lag[r];
                    *reinterpret_cast<uint32_t*>(key_bytes + won_at[r]) = e[r];
                }
            }
        }
        wave_lds_fence();
        floor_sum += round_min;
        uint64_t got[EC];
#pragma unroll
        for (int r = 0; r < EC; ++r) got[r] = hand[lane * EC + r];                     // (all reads out before the first use)
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            tot[r] += got[r];                                                          // Main.java:265 (idle bins and late rounds: + 0)
            const uint32_t field = (uint32_t)((tot[r] - floor_sum) >> drop);
            key[r] = (mine[r] ? (field << idx_bits) : kIdle) | (uint32_t)(lane * EC + r);     // (the keys start every round at home)
            lag[r] = next[r];
            won_at[r] = next_at[r];
        }
        round_min = next_min;
        wave_lds_fence();
#ifdef LA_BLOCK_CLOCKS
        if (lane == 0 && blockIdx.x == 0) {
            asm volatile("" : "+v"(key[0]));
            const unsigned long long kclk2 = wall_clock64();
            g_block_clocks[6] += kclk1 - kclk0;                                        // the network
            g_block_clocks[7] += kclk2 - kclk1;                                        // the rest of the round
        }
#endif
    }
    if (a.out_total) {
#pragma unroll
        for (int r = 0; r < EC; ++r)
            if (mine[r]) a.out_total[c0 + lane * EC + r] = (int64_t)tot[r];
    }
}

// The rounds behind a topic's last lag (block path, the two one-wavefront greedy forms that leave their winners in the slots).  The
// sorted lags descend: from the first round whose FIRST lag is zero on, nothing is added to any bin, the order of the bins is final
// and every further round repeats that round's winners -- a consumer group that has caught up on most of a topic's partitions runs
// 2 of its 500 rounds.  zero = the word a lag of zero stands as (a slot: 0; a sorted key: kLagKeyFlip).  Out: *P_run = the partitions
// the greedy has to run over (whole rounds, the first all-zero round included), *tail_from = that round's first position: position
// i >= *P_run has the winner of position *tail_from + i % C.  Ends with a barrier (every thread of the workgroup calls it).
__device__ __forceinline__ void zero_tail_rounds(const uint64_t* s_key, const uint64_t zero, uint32_t* s_word, const int P, const int C,
                                                 const int tid, const int nt, int* P_run, int* tail_from) {
    *P_run = P;
    *tail_from = 0;
    if (tid == 0) *s_word = (uint32_t)P;
    lds_barrier();
    if (P <= 0 || s_key[P - 1] != zero) return;                         // (workgroup-uniform) no lag of zero at all
    for (int i = tid; i < P; i += nt)
        if (s_key[i] == zero && (i == 0 || s_key[i - 1] != zero)) *s_word = (uint32_t)i;       // one writer: the first zero
    lds_barrier();
    const int z = (int)*s_word;
    const int fz = (z + C - 1) / C;                                     // the first round that hands out zeros only
    if ((fz + 1) * C < P) { *P_run = (fz + 1) * C; *tail_from = fz * C; }
}

// LDS byte address of a __shared__ object (what ds_* instructions take)
__device__ __forceinline__ uint32_t lds_address(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ uint64_t lds_read64(uint32_t addr) {
    return *(const __attribute__((address_space(3))) uint64_t*)(uintptr_t)addr;
}
__device__ __forceinline__ void lds_write32(uint32_t addr, uint32_t v) {
    *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)addr = v;
}

// Greedy rounds for up to 64 consumers with packed bins, one bin per lane, the chain of rounds cut down to its
// instruction count (la_sort64.h (d)).  `slot` holds, per sorted position, the partition's lag << idx_bits (written by
// the whole workgroup) and slot[P] = 0; a round reads its slots, and writes the winner's consumer position back into
// the low word of the slot.  Member ranks and global stores are the workgroup's business after the last round.
template <int L>
__device__ __forceinline__ void greedy_one_wave_slots(const BlockArgs& a, uint64_t* slot, int64_t c0, int P, int C,
                                                      int idx_bits, int lane) {
    const uint32_t idx_mask = (1u << idx_bits) - 1;
    const bool own = lane < C;
    const uint32_t base = lds_address(slot);
    const uint32_t zb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(base + (uint32_t)P * 8u));
    const uint32_t stride = (uint32_t)__builtin_amdgcn_readfirstlane(C * 8);
    const int rounds = __builtin_amdgcn_readfirstlane((P + C - 1) / C);
    const uint32_t lane_mask = own ? idx_mask : 0u;       // idle lanes write zero into the zero slot
    P64 bin = p64_from(own ? (uint64_t)lane : kRoundSentinel);
    uint32_t kv[4];
    round_keep_vectors(lane, kv);
    uint32_t sb = own ? base + (uint32_t)lane * 8u : 0x40000000u;   // idle lanes: always clamped
    uint32_t ab_a = sb < zb ? sb : zb, ab_b;
    uint64_t cur_a = lds_read64(ab_a), cur_b;
    // round 0: the bins are in order as they stand (totals 0, positions ascending)
    sb += stride;
    ab_b = sb < zb ? sb : zb;
    cur_b = lds_read64(ab_b);
    {
        const uint64_t nb = p64_value(bin) + cur_a;                                        // Main.java:265
        bin = p64_from(nb);
        lds_write32(ab_a, (uint32_t)nb & lane_mask);
    }
    if constexpr (L >= 2 && L <= 16) {
        asm volatile("s_nop 1" : "+v"(bin.lo), "+v"(bin.hi));          // compiler-written bins -> a DPP read
        int q = 1;
        for (; q + 1 < rounds; q += 2) {
            greedy_round_p64<L>(bin, sb, ab_b, ab_a, cur_b, cur_a, stride, zb, lane_mask, kv);
            greedy_round_p64<L>(bin, sb, ab_a, ab_b, cur_a, cur_b, stride, zb, lane_mask, kv);
        }
        if (q < rounds) greedy_round_p64<L>(bin, sb, ab_b, ab_a, cur_b, cur_a, stride, zb, lane_mask, kv);
    } else {
        for (int q = 1; q < rounds; ++q) {
            sb += stride;
            ab_a = sb < zb ? sb : zb;
            cur_a = lds_read64(ab_a);
            if constexpr (L > 1) {
                asm volatile("s_nop 1" : "+v"(bin.lo), "+v"(bin.hi));
                bitonic_sort_lanes_p64<L, true>(bin);      // one wavefront alone: the steps without the SALU hop
            }
            const uint64_t nb = p64_value(bin) + cur_b;
            bin = p64_from(nb);
            lds_write32(ab_b, (uint32_t)nb & lane_mask);
            ab_b = ab_a;
            cur_b = cur_a;
        }
    }
    if (a.out_total && own) {
        const uint64_t v = p64_value(bin);
        a.out_total[c0 + ((uint32_t)v & idx_mask)] = (int64_t)(v >> idx_bits);
    }
}

// Greedy rounds for 257 .. 2 048 consumers with packed bins: the bins stay in registers, EC per thread over
// n_c / EC threads (slot = tid*EC + r); a round's sort runs inside each wavefront through the networks of
// la_sort64.h and across wavefronts through LDS exchanges (x: n_c words, register-major) -- the scheme of the
// large path's one-workgroup greedy, fed from LDS.  Threads beyond the bins only keep the barriers company.
template <int EC>
__device__ __forceinline__ void greedy_multi_wave_packed(const BlockArgs& a, const uint64_t* s_key, const int32_t* s_rank,
                                                         uint64_t* x, int64_t p0, int64_t c0, int P, int C, int n_c,
                                                         int idx_bits, int tid) {
    constexpr int kSpanSlots = kWave * EC;
    constexpr bool kVo = true;                                           // the steps' VALU form (few wavefronts per SIMD)
    const int ntu = n_c / EC;                                            // threads that hold bins (multiple of 64)
    const bool active = tid < ntu;                                       // wavefront-uniform
    if (!active) return;                                                 // (s_barrier counts the wavefronts that are left)
    const uint32_t idx_mask = (1u << idx_bits) - 1;
    P64 bin[EC];
    uint64_t lag[EC];
#pragma unroll
    for (int r = 0; r < EC; ++r) {
        const int e = tid * EC + r;
        bin[r] = p64_from((active && e < C) ? (uint64_t)e : kRoundSentinel);
        lag[r] = (active && e < C && e < P) ? (s_key[e] ^ kLagKeyFlip) : 0;
    }
    const int rounds = (P + C - 1) / C;
    for (int q = 0; q < rounds; ++q) {
        uint64_t next[EC];
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const int e = tid * EC + r;
            const int s = (q + 1) * C + e;
            next[r] = (active && e < C && s < P) ? (s_key[s] ^ kLagKeyFlip) : 0;
        }
        if (q > 0) {
            if (active) {
#pragma unroll
                for (int r = 0; r < EC; ++r) asm volatile("s_nop 1" : "+v"(bin[r].lo), "+v"(bin[r].hi));
                bitonic_sort_tile_p64<kWave, EC, kVo>(bin);
            }
            for (int K = 2 * kSpanSlots; K <= n_c; K <<= 1) {
                for (int j = K >> 1; j >= kSpanSlots; j >>= 1) {
                    const int mask = (j == (K >> 1)) ? (K - 1) : j;       // mirror first, then i <-> i ^ j
                    if (active) {
#pragma unroll
                        for (int r = 0; r < EC; ++r) x[r * ntu + tid] = p64_value(bin[r]);
                    }
                    lds_barrier();        // (LDS traffic only: the round's global stores stay in flight, la_device.h)
                    if (active) {
#pragma unroll
                        for (int r = 0; r < EC; ++r) {
                            const int i = tid * EC + r;
                            const int pi = i ^ mask;
                            const uint64_t o = x[(pi % EC) * ntu + pi / EC], v = p64_value(bin[r]);
                            const bool keep_min = (i & j) == 0;
                            const uint64_t lo = o < v ? o : v, hi = o < v ? v : o;
                            bin[r] = p64_from(keep_min ? lo : hi);
                        }
                    }
                    lds_barrier();        // (LDS traffic only: the round's global stores stay in flight, la_device.h)
                }
                if (active) {
#pragma unroll
                    for (int r = 0; r < EC; ++r) asm volatile("s_nop 1" : "+v"(bin[r].lo), "+v"(bin[r].hi));
                    clean_p64<kWave, EC, kSpanSlots / 2, false, kVo>(bin);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const int e = tid * EC + r;
            const int s = q * C + e;
            if (active && e < C && s < P) {
                const uint64_t nb = p64_value(bin[r]) + (lag[r] << idx_bits);              // Main.java:265
                bin[r] = p64_from(nb);
                a.out_rank[p0 + s] = s_rank[(uint32_t)nb & idx_mask];
            }
            lag[r] = next[r];
        }
    }
    if (a.out_total && active) {
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const uint64_t v = p64_value(bin[r]);
            if (v != kRoundSentinel) a.out_total[c0 + ((uint32_t)v & idx_mask)] = (int64_t)(v >> idx_bits);
        }
    }
}

// LDS: [region A][bins].  Region A is the sort's exchange area (kXchg * blockDim keys + ids) and, once the sort is
// done, the sorted keys by position (E * blockDim words); the ids leave through the registers.
template <int E>
__global__ __launch_bounds__(1024) void block_topic_kernel(BlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t smem64[];
    // bytes; + one 16-byte slot behind the np_cap keys: s_key[P] is the zero slot of the one-wavefront slots greedy, and with
    // P == np_cap (E = 16: region A is exactly np_cap words) it would otherwise be s_tot[0]
    const int region_a = (E * 8 > kXchg * 12 ? E * 8 : kXchg * 12) * (int)blockDim.x + kZeroSlotBytes;
    static_assert(kZeroSlotBytes % 16 == 0 && kZeroSlotBytes >= 8, "the bins behind region A stay 16-byte aligned");
    uint64_t* s_key = smem64;                                         // sorted partition keys by position
    uint64_t* x_key = smem64;                                         // exchange area of the sort
    uint32_t* x_id = reinterpret_cast<uint32_t*>(x_key + kXchg * blockDim.x);
    uint64_t* s_tot = smem64 + region_a / 8;                          // [nc_cap] biased bin totals
    uint32_t* s_idx = reinterpret_cast<uint32_t*>(s_tot + a.nc_cap);  // [nc_cap] bin -> consumer position
    int32_t* s_rank = reinterpret_cast<int32_t*>(s_idx + a.nc_cap);   // [nc_cap] consumer position -> member rank

    const int tid = threadIdx.x, nt = blockDim.x;
    // the topic's segment bounds are the same in every lane: said so (v_readfirstlane -> SGPRs).  As four VGPR pairs they were
    // the first thing a 128-register workgroup spilled, and the reload in front of every record's store (with the s_waitcnt
    // vmcnt(0) that comes with it) made the sixteen stores of a thread wait for each other: 7 us of cfg2b's 100.
    auto uniform64 = [](int64_t v) {
        return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v));
    };
    const int64_t topic = uniform64(a.list ? a.list[blockIdx.x] : a.inline_list[blockIdx.x & 7]);
    const int64_t p0 = uniform64(a.part_off[topic]), c0 = uniform64(a.cons_off[topic]);
    const int64_t Pl = uniform64(a.part_off[topic + 1]) - p0, Cl = uniform64(a.cons_off[topic + 1]) - c0;
    if (Pl < 0 || Cl < 0 || Pl > a.np_cap || Cl > a.nc_cap) {           // the host's lists disagree with the
        if (tid == 0) atomicOr(a.status, kStatusShape);                 // device's offsets: leave outputs alone
        return;
    }
    const int P = (int)Pl, C = (int)Cl;
    const bool latest = a.reset_latest != 0;
    LA_BCLK_START;

    // ---- records: coalesced loads straight into the sort's registers (the sort does not care where a record
    // starts), lag fused in --------------------------------------------------------------------------------
    const int n_eff = pow2ceil_dev(P);
    const int nt_eff = n_eff > E ? n_eff / E : 1;                       // threads holding slots < n_eff
    // The records go to as FEW wavefronts as hold them (record r * nt_ld + tid in register r of thread tid < nt_ld): a topic of
    // 10 000 partitions in the 16 384 class fills 10 wavefronts and leaves 6 with sentinels only, and the packed sort skips
    // every merge whose upper half is all sentinels -- those 6 idle through 91 of the 105 steps and the 10 share the SIMDs.
    const int nt_ld = min(nt_eff, ((P + E - 1) / E + kWave - 1) & ~(kWave - 1));
    int64_t lag[E];
    int32_t pid[E];
    uint64_t lag_or = 0;
    uint32_t id_or = 0;
#pragma unroll
    for (int r = 0; r < E; ++r) { lag[r] = 0; pid[r] = 0; }
    if (P > 0) {
        // UNCONDITIONAL loads, index clamped, all of a kind issued back to back: a branch around a load makes hipcc
        // wait for it before issuing the next one, and with `if (valid) { committed; begin?; end; id }` per record
        // that was ~4 dependent memory round trips for each of the E records of a thread -- tens of microseconds on
        // the critical path of a workgroup that has nothing else to run.  Now: one round trip for committed / end /
        // id (or the lags), one more for `begin` where there is no committed offset (lanes that do not need it all
        // read the topic's first word: one cache line per wavefront, Main.java:384-396).
        // (32-bit record offsets, the 64-bit address formed at each load: sixteen 64-bit addresses held across the whole
        //  stage were sixteen registers more than the 128 a 1 024-thread workgroup gets, spilled and reloaded around every load)
        int32_t g[E];
        bool valid[E];
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int src = r * nt_ld + tid;
            valid[r] = tid < nt_ld && src < P;
            g[r] = valid[r] ? src : 0;
        }
        int32_t idv[E];
#pragma unroll
        for (int r = 0; r < E; ++r) idv[r] = a.pid[p0 + g[r]];
        if (a.lag) {
            int64_t lv[E];
#pragma unroll
            for (int r = 0; r < E; ++r) lv[r] = a.lag[p0 + g[r]];
#pragma unroll
            for (int r = 0; r < E; ++r) lag[r] = valid[r] ? lv[r] : 0;
        } else {
            // computePartitionLag (Main.java:376-404; la_device.h partition_lag) in two stages, so that the committed and end
            // offsets are dead before the begin offsets arrive: sixteen records x three 64-bit arrays held at once were 96
            // registers of the 128 a 1 024-thread workgroup gets (E = 16: 30 spilled registers -> 12; cfg2b 0.0885 -> 0.0879 ms.
            // The offsets form still runs ~6 us behind the lags form, profiles/r05_ag_block_kernel_durations.txt: three arrays'
            // loads and a dependent second stage against one array's).
            //   committed >= 0:            lag = max(end - committed, 0)          (done after stage 1)
            //   none, reset latest:        lag = max(end - end, 0) = 0            (done after stage 1)
            //   none, reset earliest:      lag = max(end - begin, 0)              (lag[r] holds `end` until stage 2)
            uint32_t need = 0;                                          // bit r: record r waits for its begin offset
            {
                int64_t cm[E], en[E];
#pragma unroll
                for (int r = 0; r < E; ++r) cm[r] = a.committed[p0 + g[r]];
#pragma unroll
                for (int r = 0; r < E; ++r) en[r] = a.end[p0 + g[r]];
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const bool none = cm[r] < 0;
                    const int64_t d = (int64_t)((uint64_t)en[r] - (uint64_t)cm[r]);
                    lag[r] = none ? (latest ? 0 : en[r]) : (d > 0 ? d : 0);
                    if (none && !latest && valid[r]) need |= 1u << r;
                }
            }
            if (!latest) {
                int64_t bg[E];
                if (a.begin) {
#pragma unroll
                    for (int r = 0; r < E; ++r) bg[r] = a.begin[p0 + (((need >> r) & 1u) ? g[r] : 0)];
                } else {
#pragma unroll
                    for (int r = 0; r < E; ++r) bg[r] = 0;
                }
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const int64_t d = (int64_t)((uint64_t)lag[r] - (uint64_t)bg[r]);      // (lag[r] is `end` here)
                    if ((need >> r) & 1u) lag[r] = d > 0 ? d : 0;
                }
            }
#pragma unroll
            for (int r = 0; r < E; ++r) lag[r] = valid[r] ? lag[r] : 0;
        }
#pragma unroll
        for (int r = 0; r < E; ++r) {
            pid[r] = valid[r] ? idv[r] : 0;
            lag_or |= (uint64_t)lag[r];
            id_or |= (uint32_t)pid[r];
        }
    }
    // the topic's member ranks, fetched with the records: their LDS copies are written after the sort, BEHIND the stores of the
    // sorted partition ids -- a load issued there would have to wait for those stores (one in-order vmcnt), and so would the
    // barrier in front of the greedy rounds
    constexpr int kRankPre = E == 16 ? 1 : 4;                           // nc_cap / blockDim of the size classes (block_launch)
    int32_t rank_pre[kRankPre];
#pragma unroll
    for (int k = 0; k < kRankPre; ++k) {
        const int i = tid + k * nt;
        rank_pre[k] = a.cons_rank[c0 + (i < C ? i : 0)];
    }
    LA_BCLK(0);                                                         // record loads issued and consumed
    // do the topic's records fit one 64-bit word?  (workgroup-wide OR of the lags and the ids)
    uint32_t* s_or = reinterpret_cast<uint32_t*>(s_rank + a.nc_cap);   // [3] (4 allotted) workgroup-wide ORs
    if (tid < 3) s_or[tid] = 0;
    __syncthreads();
    {
        const uint32_t a0 = wave_or_u32((uint32_t)lag_or), a1 = wave_or_u32((uint32_t)(lag_or >> 32)), a2 = wave_or_u32(id_or);
        if ((tid & (kWave - 1)) == 0) {
            if (a0) atomicOr(&s_or[0], a0);
            if (a1) atomicOr(&s_or[1], a1);
            if (a2) atomicOr(&s_or[2], a2);
        }
    }
    __syncthreads();
    const uint64_t all_lag = ((uint64_t)s_or[1] << 32) | s_or[0];
    const uint32_t all_id = s_or[2];
    const int lbw = all_lag ? 64 - __builtin_clzll((unsigned long long)all_lag) : 0;    // 64: a negative lag
    const int sh = all_id ? 32 - __builtin_clz(all_id) : 0;                             // 32: a negative id
    const bool fits = sh < 32 && lbw + sh <= 63;                                        // workgroup-uniform
    // Up to 64 consumers and bins that pack (no negative lag: lbw < 64; no total near 2^62: the top bit of the OR of the
    // lags IS the top bit of the largest lag): the sorted positions become the slots of greedy_one_wave_slots.
    const int n_c = pow2ceil_dev(C);
    const int idx_bits = 31 - __builtin_clz((unsigned)(n_c | 1));
    const bool slots = C > 0 && P > 0 && n_c <= kWave && lbw < 64 &&
                       lbw + (32 - __builtin_clz((unsigned)((P + C - 1) / C) | 1u)) + idx_bits <= 62;

    LA_BCLK(1);
    // ---- sort by (lag desc, partition asc); the ids leave from the registers, the keys go to LDS by position --
    if (fits) {
        const uint64_t lag_max = lbw ? (~0ull >> (64 - lbw)) : 0;
        const uint32_t id_mask = sh ? (0xFFFFFFFFu >> (32 - sh)) : 0;
        P64 rec[E];
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const bool valid = tid < nt_ld && r * nt_ld + tid < P;
            rec[r] = p64_from(valid ? (((lag_max - (uint64_t)lag[r]) << sh) | (uint32_t)pid[r]) : ~0ull);
        }
        // (below ~700 partitions the network's few steps are cheaper than six digits of four barriers each: 10 000 x 200 x 100
        //  0.105 ms against 0.150, 5 000 x 512 x 100 and 4 000 x 700 x 100 even, 3 000 x 1 000 x 100 0.104 against 0.095)
        if (a.radix_sort && lbw + sh > 0 && (a.radix_sort >= 2 || P >= 768)) {
            uint32_t* aux = reinterpret_cast<uint32_t*>(s_rank + a.nc_cap) + 4;          // [512] behind the OR words
            // Partition ids of a Kafka topic are 0 .. P-1: a permutation.  Then "sorted by id" is ONE scatter -- record with id i to
            // place i -- instead of the id's digit passes, and the stable passes over the lag's digits alone finish the order
            // (lag desc, id asc): cfg2b (40-bit lags, 14-bit ids) runs 5 digits instead of 7.  Checked, not assumed: every id below
            // P and a bit of an LDS bitmap per id set exactly once (returning atomic OR); anything else sorts all digits as before.
            int first_shift = 0;
            if (a.dense_ids && sh > 0 && (lbw + sh + 7) / 8 - (lbw + 7) / 8 >= 2) {
                uint32_t* bitmap = reinterpret_cast<uint32_t*>(s_tot);                     // [(P + 31) / 32] words (the sort's counters: not in use yet)
                for (int k = tid; k < (P + 31) / 32; k += nt) bitmap[k] = 0;
                if (tid == 0) s_or[3] = 0;
                __syncthreads();
                bool bad = false;
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const bool valid = tid < nt_ld && r * nt_ld + tid < P;
                    const uint32_t id = rec[r].lo & id_mask;                               // (sh <= 31: the id sits in the low word)
                    if (valid) {
                        if (id >= (uint32_t)P) bad = true;
                        else bad |= (atomicOr(&bitmap[id >> 5], 1u << (id & 31)) >> (id & 31)) & 1u;
                    }
                }
                if (__builtin_amdgcn_ballot_w64(bad) != 0 && (tid & (kWave - 1)) == 0) atomicOr(&s_or[3], 1u);
                __syncthreads();
                if (s_or[3] == 0) {                                                        // workgroup-uniform
#pragma unroll
                    for (int r = 0; r < E; ++r) {
                        const bool valid = tid < nt_ld && r * nt_ld + tid < P;
                        if (valid) x_key[rec[r].lo & id_mask] = p64_value(rec[r]);
                    }
                    __syncthreads();
                    // back into the registers in the order the passes rank in: blocked by wavefront, register-major
                    const int w0 = (tid & ~(kWave - 1)) * E, ln = tid & (kWave - 1);
#pragma unroll
                    for (int r = 0; r < E; ++r) {
                        const int pos = w0 + r * kWave + ln;
                        rec[r] = p64_from(pos < P ? x_key[pos] : ~0ull);
                    }
                    __syncthreads();                                                       // (the first pass scatters into x_key again)
                    first_shift = sh;
                }
            }
            block_sort_radix<E>(rec, nt_ld * E, lbw + sh, tid, nt, x_key, reinterpret_cast<uint32_t*>(s_tot), aux, first_shift);
            LA_BCLK(2);
            // the sorted records lie in region A by position: record i becomes the key of position i, in place.  On the way
            // every record is compared with the one before it: the ranks of the sort rest on a property of the LDS atomics
            // that the device was tested for once (la_create), so an order that is not one is an error of the call
            // (kStatusOrder -> LA_EHIP), never a silently different assignment -- what emit_ids_kernel does for the large path.
            // A padding sentinel (all ones) and a real record whose lbw + sh bits are all ones (lag 0, id 2^sh - 1) have the
            // same digits when lbw + sh is a multiple of 8, and the stable sort keeps such ties in (wavefront, register, lane)
            // order -- a sentinel of wavefront 0 may then stand before that record of wavefront 1 and land at a position < P.
            // Every position < P belongs to a real record, and the only real records a sentinel can displace are equal to
            // 2^(lbw+sh) - 1: reading positions < P through the mask of the examined bits puts exactly that value there
            // (real records have no bit above it).
            const uint64_t rec_mask = ~0ull >> (64 - (lbw + sh));           // (1 <= lbw + sh <= 63 here)
            uint64_t sorted[E];
            bool bad = false;
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const int i = r * nt + tid;
                sorted[r] = x_key[i < P ? i : 0] & rec_mask;
                bad |= i > 0 && i < P && (x_key[i < P ? i - 1 : 0] & rec_mask) > sorted[r];
            }
            __syncthreads();                                            // every record is in a register: the keys go over them
            if (__builtin_amdgcn_ballot_w64(bad) != 0 && (tid & (kWave - 1)) == 0) atomicOr(a.status, kStatusOrder);
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const int i = r * nt + tid;
                if (i < P) {
                    const uint64_t v = sorted[r];
                    const uint64_t lv = lag_max - (v >> sh);
                    s_key[i] = slots ? (lv << idx_bits) : (lv ^ kLagKeyFlip);
                    a.out_pid[p0 + i] = (int32_t)((uint32_t)v & id_mask);
                    if (C == 0) a.out_rank[p0 + i] = -1;                // Main.java:211-214: nobody to assign to
                }
            }
        } else {
        block_sort_packed<E>(rec, n_eff, nt_ld * E, tid, nt, x_key);
        LA_BCLK(2);
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int i = tid * E + r;
            if (i < P) {
                const uint64_t v = p64_value(rec[r]);
                const uint64_t lv = lag_max - (v >> sh);
                s_key[i] = slots ? (lv << idx_bits) : (lv ^ kLagKeyFlip);
                a.out_pid[p0 + i] = (int32_t)((uint32_t)v & id_mask);
                if (C == 0) a.out_rank[p0 + i] = -1;                    // Main.java:211-214: nobody to assign to
            }
        }
        }
    } else {
        Rec rec[E];
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const bool valid = tid < nt_ld && r * nt_ld + tid < P;
            const uint64_t key = (uint64_t)lag[r] ^ kLagKeyFlip;
            rec[r].hi = valid ? (uint32_t)(key >> 32) : 0xFFFFFFFFu;
            rec[r].lo = valid ? (uint32_t)key : 0xFFFFFFFFu;
            rec[r].tb = valid ? ((uint32_t)pid[r] ^ kPidBias) : 0xFFFFFFFFu;
        }
        block_sort_regs<E>(rec, n_eff, tid, nt, x_key, x_id);
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int i = tid * E + r;
            if (i < P) {
                const uint64_t key = ((uint64_t)rec[r].hi << 32) | rec[r].lo;
                s_key[i] = slots ? ((key ^ kLagKeyFlip) << idx_bits) : key;
                a.out_pid[p0 + i] = (int32_t)(rec[r].tb ^ kPidBias);
                if (C == 0) a.out_rank[p0 + i] = -1;                    // Main.java:211-214: nobody to assign to
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kRankPre; ++k) {
        const int i = tid + k * nt;
        if (i < C) {
            s_tot[i] = kTotalBias;                                      // total 0
            s_idx[i] = (uint32_t)i;
            s_rank[i] = rank_pre[k];
        }
    }
    if (slots && tid == 0) s_key[P] = 0;                                // the slot idle lanes and rounds past P read
    lds_barrier();                                                      // (LDS only: the stores of the sorted ids stay in flight)
    LA_BCLK(3);
    if (C == 0) return;

    // ---- greedy rounds --------------------------------------------------------------------------------
    if (slots) {
        int P_run, tail_from;
        zero_tail_rounds(s_key, 0ull, s_or + 3, P, C, tid, nt, &P_run, &tail_from);
        if (tid < kWave) {
            switch (n_c) {
                case 1: greedy_one_wave_slots<1>(a, s_key, c0, P_run, C, idx_bits, tid); break;
                case 2: greedy_one_wave_slots<2>(a, s_key, c0, P_run, C, idx_bits, tid); break;
                case 4: greedy_one_wave_slots<4>(a, s_key, c0, P_run, C, idx_bits, tid); break;
                case 8: greedy_one_wave_slots<8>(a, s_key, c0, P_run, C, idx_bits, tid); break;
                case 16: greedy_one_wave_slots<16>(a, s_key, c0, P_run, C, idx_bits, tid); break;
                case 32: greedy_one_wave_slots<32>(a, s_key, c0, P_run, C, idx_bits, tid); break;
                default: greedy_one_wave_slots<64>(a, s_key, c0, P_run, C, idx_bits, tid); break;
            }
        }
        lds_barrier();
        LA_BCLK(4);
        // the winners (consumer positions, low word of each slot) -> member ranks, every wavefront, coalesced; the rounds behind
        // the last lag repeat the last round that ran
        const uint32_t* won = reinterpret_cast<const uint32_t*>(s_key);
        for (int i = tid; i < P; i += nt) a.out_rank[p0 + i] = s_rank[won[2 * (i < P_run ? i : tail_from + i % C)]];
        LA_BCLK(5);
        return;
    }
    // Slots 2*idx and 2*idx+1 are updated by the thread that owns pair idx in the network's in-span steps,
    // so the update and the next round's sort are separated by a wavefront fence only.
    if (n_c <= 4 * kWave) {
        // packed bins when nothing can overflow: the keys are sorted, the first carries the largest lag and
        // the last the smallest
        const int64_t lmax = P > 0 ? (int64_t)(s_key[0] ^ kLagKeyFlip) : 0;
        const int64_t lmin = P > 0 ? (int64_t)(s_key[P - 1] ^ kLagKeyFlip) : 0;
        const int lag_bits = lmax > 0 ? 64 - __builtin_clzll((unsigned long long)lmax) : 0;
        const int round_bits = 32 - __builtin_clz((unsigned)((P + C - 1) / C) | 1u);
        const bool packed = lmin >= 0 && lag_bits + round_bits + idx_bits <= 62;
        // (256 bins: four keys per lane on one wavefront lose to one bin per lane on four -- 1 x 8 000 x 256 0.095 against 0.086 ms,
        //  1 x 16 000 x 200 0.225 against 0.185; 128 bins: 1 x 10 000 x 128 0.109 against 0.130.  LA_BLOCK_KEY32=2 runs both here)
        if (packed && n_c > kWave && a.key32_greedy && (n_c == 2 * kWave || a.key32_greedy >= 2)) {
            // 128 / 256 bins: ONE wavefront, the bins stay where they are (LDS), the order comes from 32-bit keys
            if (tid == 0) s_key[P] = kLagKeyFlip;                       // "lag 0" for idle slots and rounds past the topic (P + 1: scratch)
            int P_run, tail_from;
            zero_tail_rounds(s_key, kLagKeyFlip, s_or + 3, P, C, tid, nt, &P_run, &tail_from);       // (with the barrier the sentinel needs)
            if (tid < kWave) {
                if (n_c == 2 * kWave) greedy_one_wave_key32<2>(a, s_key, s_tot, s_idx, c0, P_run, C, idx_bits, lag_bits, tid);
                else greedy_one_wave_key32<4>(a, s_key, s_tot, s_idx, c0, P_run, C, idx_bits, lag_bits, tid);
            }
            lds_barrier();
            LA_BCLK(4);
            // the winners (consumer positions, low word of each slot) -> member ranks, every wavefront, coalesced
            const uint32_t* won = reinterpret_cast<const uint32_t*>(s_key);
            for (int i = tid; i < P; i += nt) a.out_rank[p0 + i] = s_rank[won[2 * (i < P_run ? i : tail_from + i % C)]];
            LA_BCLK(5);
            return;
        }
        if (packed && n_c > kWave && n_c <= nt && (n_c > 2 * kWave || gridDim.x <= 512)) {
            // 128 / 256 bins: one per lane on 2 / 4 wavefronts (an exchange through LDS per merge level) instead of 2 / 4 per
            // lane on one; the other wavefronts leave (greedy_multi_wave_packed), so a round's barriers are among those 2 / 4.
            // 1 x 8 000 x 256: 0.121 -> 0.101 ms, 1 x 16 000 x 200: 0.268 -> 0.218, 2 000 x 1 000 x 200: 0.086 -> 0.078; with
            // 128 bins the gain is small (1 x 10 000 x 128: 0.167 -> 0.160) and a launch that fills the CUs loses 4 %, so
            // there only up to 512 topics.
            greedy_multi_wave_packed<1>(a, s_key, s_rank, s_tot, p0, c0, P, C, n_c, idx_bits, tid);
            LA_BCLK(4);
            return;
        }
        if (tid < kWave) {
#define LA_ONE_WAVE(EC, L)                                                                              \
    if (packed) greedy_one_wave_packed<EC, L>(a, s_key, s_rank, p0, c0, P, C, idx_bits, tid);           \
    else greedy_one_wave<EC, L>(a, s_key, s_rank, p0, c0, P, C, tid)
            switch (n_c) {                               // a fully unrolled network per width
                case 1: LA_ONE_WAVE(1, 1); break;
                case 2: LA_ONE_WAVE(1, 2); break;
                case 4: LA_ONE_WAVE(1, 4); break;
                case 8: LA_ONE_WAVE(1, 8); break;
                case 16: LA_ONE_WAVE(1, 16); break;
                case 32: LA_ONE_WAVE(1, 32); break;
                case 64: LA_ONE_WAVE(1, 64); break;
                case 128: LA_ONE_WAVE(2, 64); break;
                default: LA_ONE_WAVE(4, 64); break;
            }
#undef LA_ONE_WAVE
        }
        return;
    }
    // more bins than one wavefront holds comfortably.  Packed bins (no negative lag, no total near 2^62): registers
    // across wavefronts.  Otherwise: bins in LDS, same network as the classic sort, 96-bit comparisons.
    {
        const int64_t lmax = P > 0 ? (int64_t)(s_key[0] ^ kLagKeyFlip) : 0;
        const int64_t lmin = P > 0 ? (int64_t)(s_key[P - 1] ^ kLagKeyFlip) : 0;
        const int lag_bits = lmax > 0 ? 64 - __builtin_clzll((unsigned long long)lmax) : 0;
        const int round_bits = 32 - __builtin_clz((unsigned)((P + C - 1) / C) | 1u);
        if (lmin >= 0 && lag_bits + round_bits + idx_bits <= 62) {       // workgroup-uniform
            if (n_c <= nt) greedy_multi_wave_packed<1>(a, s_key, s_rank, s_tot, p0, c0, P, C, n_c, idx_bits, tid);
            else greedy_multi_wave_packed<2>(a, s_key, s_rank, s_tot, p0, c0, P, C, n_c, idx_bits, tid);
            return;
        }
    }
    const int upd = n_c >> 1;
    const int rounds = (P + C - 1) / C;
    for (int q = 0; q < rounds; ++q) {
        if (q > 0) {
            lds_bitonic_sort(n_c, C, tid, nt, true, [&](int i, int p) {
                const uint64_t ta = s_tot[i], tb = s_tot[p];
                const uint32_t ia = s_idx[i], ib = s_idx[p];
                if ((tb < ta) | ((tb == ta) & (ib < ia))) {             // Main.java:253-259
                    s_tot[i] = tb; s_tot[p] = ta;
                    s_idx[i] = ib; s_idx[p] = ia;
                }
            });
            wave_lds_fence();
        }
        const int base = q * C;
        for (int idx = tid; idx < upd; idx += nt) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * idx + h;
                const int s = base + e;
                if (e < C && s < P) {
                    s_tot[e] += s_key[s] ^ kLagKeyFlip;                 // Main.java:265, wrapping like a long
                    a.out_rank[p0 + s] = s_rank[s_idx[e]];
                }
            }
        }
    }
    if (a.out_total) {
        wave_lds_fence();
        for (int idx = tid; idx < upd; idx += nt) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * idx + h;
                if (e < C) a.out_total[c0 + s_idx[e]] = (int64_t)(s_tot[e] ^ kTotalBias);
            }
        }
    }
}

}  // namespace

#ifdef LA_BLOCK_CLOCKS
extern "C" __attribute__((visibility("default"))) int la_debug_block_clocks(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_clocks), sizeof(g_block_clocks));
    if (e == hipSuccess && reset) {
        unsigned long long zero[16] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_block_clocks), zero, sizeof zero);
    }
    return e == hipSuccess ? 0 : -3;
}
#endif

hipError_t block_launch(BlockArgs a, int cls, hipStream_t stream) {
    // workgroup size = np_cap / records per thread; the small classes leave room for many workgroups per CU
    static const int kThreads[kBlockClasses] = {64, 256, 512, 1024, 1024};
    static const int kRecs[kBlockClasses] = {8, 8, 8, 8, 16};
    static const int kNc[kBlockClasses] = {256, 256, 1024, (int)kBlockMaxConsumers, 1024};
    if (a.n_list <= 0) return hipSuccess;
    if (cls < 0 || cls >= kBlockClasses) return hipErrorInvalidValue;
    const int nt = kThreads[cls], e = kRecs[cls];
    a.np_cap = nt * e;
    a.nc_cap = kNc[cls];
    const size_t region_a = (size_t)(e * 8 > kXchg * 12 ? e * 8 : kXchg * 12) * nt + kZeroSlotBytes;
    const size_t lds = region_a + (size_t)16 * a.nc_cap + 16 + 2048;     // + [2][256] words of the radix sort behind the OR words
    // which topics sort by digits (LA_BLOCK_RADIX: 0 never -- the network, the form of rounds 1-3 --, 1 only the three largest
    // classes, 2 every class: the default; topics of fewer than 768 partitions keep the network; needs the atomic ranks).
    // Same box, ms per call, network / digits: 1 x 10 000 x 128 0.164 / 0.126, 1 x 16 000 x 200 0.226 / 0.177, 200 x 8 000 x 16
    // 0.147 / 0.124, 64 x 8 192 x 2 048 0.098 / 0.075, 1 000 x 2 000 x 100 0.089 / 0.072, 1 000 x 4 000 x 100 0.190 / 0.153
    // (profiles/archive/r04_block_radix.txt)
    // (3: every topic whatever its size -- the test hook that drives small topics through the digits)
    static const int radix_mode = [] { const char* e = getenv("LA_BLOCK_RADIX"); return e ? atoi(e) : 2; }();
    a.radix_sort = (large_atomic_rank_supported() && (radix_mode >= 2 || (radix_mode == 1 && cls >= 2))) ? (radix_mode >= 3 ? 2 : 1) : 0;
    // LA_BLOCK_KEY32=0: the 65 .. 256-bin greedy of rounds 3-4 (64-bit bins through the networks) instead of the 32-bit-key form (A/B, tests)
    static const int key32_mode = [] { const char* e = getenv("LA_BLOCK_KEY32"); return e ? atoi(e) : 1; }();
    a.key32_greedy = key32_mode;
    // LA_BLOCK_DENSE_IDS=0: never the one-scatter placement by id in front of the digit passes (A/B, tests)
    static const int dense_mode = [] { const char* e = getenv("LA_BLOCK_DENSE_IDS"); return e ? atoi(e) : 1; }();
    a.dense_ids = dense_mode;
    static PerDeviceOnce lds_opt_in;
    hipError_t err = lds_opt_in.run([] {
        hipError_t e2 = hipFuncSetAttribute((const void*)block_topic_kernel<8>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e2 != hipSuccess) return e2;
        return hipFuncSetAttribute((const void*)block_topic_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024);
    });
    if (err != hipSuccess) return err;
    if (e == 8) LA_LAUNCH(block_topic_kernel<8>, dim3((unsigned)a.n_list), dim3(nt), lds, stream, a);
    else LA_LAUNCH(block_topic_kernel<16>, dim3((unsigned)a.n_list), dim3(nt), lds, stream, a);
    return hipGetLastError();
}

}  // namespace la
