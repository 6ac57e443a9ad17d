// la_group_small.h -- every member's list for what a real rebalance is: up to a few thousand entries, ONE workgroup.
//
// The wrap step of the reference (every member's list is created at Main.java:171-174 and appended to at :264, topic by topic,
// inside a topic in assignment order) as a stable counting sort in LDS, LINEAR in n.  Device code shared by
//   * group_small_kernel (la_large.hip): the one-workgroup form of la_group_by_member, 1 024 threads;
//   * the tail of the single-launch tile kernel (la_wave_tile_impl.h): the LAST workgroup of a small rebalance's one launch
//     builds the lists and finishes the call, 256 threads.
//
// member_keys + plan + one or two radix passes + emit are five dependent launches (~25 us, 84 us for 2 000 entries) for a job one
// workgroup does in a few microseconds (round 3's form placed an entry by walking all entries before it: n^2 / 2 compares, hence
// its 1 024-entry limit):
//   1. count the entries of every group (group = member rank + 1; 0 = topics without consumers), exclusive scan -> cursors;
//   2. chunks of 64 consecutive entries, chunk c to wavefront c % (NT / 64): inside a chunk every lane finds its peers (the lanes
//      with the same group: one ballot per group-id bit) -- its rank among them and, for the first of them, their number;
//   3. the chunks take their places IN ORDER: wavefront-ordered hand-over -- a wavefront's turn waits until `turn` says its
//      predecessor has advanced the cursors, its group leaders advance them by their peers' counts (one LDS atomic per chunk,
//      back to back), it passes the turn on.  Turn t - 1 belongs to another wavefront of the same workgroup that waits for
//      nothing later: no deadlock; the ordered section is a few atomic instructions per chunk.
// Stable by construction (chunk order, then lane order), no reliance on how colliding lanes of an atomic are served.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "la_device.h"

namespace la {

#ifdef LA_GROUP_CLOCKS   // lab build of la_large.hip: thread 0 accumulates the time of every phase of group_small_body_staged (100 MHz)
__device__ unsigned long long g_group_clocks[12];
#define LA_GCLK(i)                                                  \
    do {                                                            \
        if (threadIdx.x == 0) {                                     \
            const unsigned long long now_ = wall_clock64();         \
            g_group_clocks[i] += now_ - gclk_;                      \
            gclk_ = now_;                                           \
        }                                                           \
    } while (0)
#define LA_GCLK_START unsigned long long gclk_ = wall_clock64()
#else
#define LA_GCLK(i) do {} while (0)
#define LA_GCLK_START do {} while (0)
#endif

constexpr int kSmallGroupN = 2560;       // entries.  Measured on one box, device-resident, back to back (tools/group_probe.py,
                                         // profiles/archive/r04_group_probe.txt): this kernel 4.0 us at 100 entries, 9 at 1 000, 14.8 at 2 000, 26 at
                                         // 4 096 (its chunks' global loads and topic searches are dependent round trips, 16 chunks deep per
                                         // wavefront at 16 384: 90-100 us), the radix form 17-23 us whatever the size: beyond ~2 500 entries
                                         // the five launches win
constexpr int kSmallGroupM = 8192;       // groups (members + 2) of the 1 024-thread kernel
constexpr int kSmallGroupBits = 13;      // bits of a group id
constexpr int kTailMaxEntries = 1024;    // entries up to which a zero-copy call fuses its lists into the tile kernel (la_api.hip, assign_small_zc)
constexpr int kTailGroupM = 2048;        // groups (members + 2) the tile kernel's tail holds (8 KB of LDS beside the tiles' slices)

// A TURN is kSub consecutive chunks of one wavefront: their ranks are found first, side by side; inside the turn the wavefront's
// cursor atomics go out back to back (LDS executes one wavefront's operations in order), so the ordered hand-over -- ~0.4 us per
// turn -- is paid once per kSub * 64 entries.  kSub = 1 is what runs (every wavefront gets work); 4 was measured 2-3 us slower at
// every size the kernel is used for (lab builds with -DLA_GROUP_SUB=4).
// topic_of (may be null): the topic of every entry, already worked out (LDS); otherwise a binary search over part_off per entry.
template <int kSub, int NT>
__device__ __forceinline__ void group_small_place(int n, uint32_t G, int64_t n_topics, const int64_t* part_off, const int32_t* out_partition,
                                                  const int32_t* member_rank, int32_t* grouped_topic, int32_t* grouped_partition,
                                                  int32_t* grouped_entry, uint32_t* start, uint32_t* turn_p, int lane, int wave,
                                                  const int32_t* topic_of = nullptr) {
    uint32_t& turn = *turn_p;
    const uint64_t below = ((uint64_t)1 << lane) - 1;
    const int n_turns = (n + kSub * kWave - 1) / (kSub * kWave);
    for (int turn_i = wave; turn_i < n_turns; turn_i += NT / kWave) {
        uint32_t gi[kSub], rank[kSub], cnt[kSub], first[kSub];
        int leader[kSub];
        int32_t part[kSub], topic[kSub];
        bool valid[kSub];
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
            const int i = (turn_i * kSub + u) * kWave + lane;
            valid[u] = i < n;
            gi[u] = 0;
            part[u] = 0;
            if (valid[u]) {
                gi[u] = (uint32_t)(member_rank[i] + 1);
                gi[u] = gi[u] < G ? gi[u] : G;
                part[u] = out_partition ? out_partition[i] : 0;
            }
        }
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
            uint64_t peers = __ballot(valid[u]);
#pragma unroll
            for (int bit = 0; bit < kSmallGroupBits + 1; ++bit) {       // (G itself may need one bit more than G - 1)
                const bool one = (gi[u] >> bit) & 1u;
                const uint64_t bal = __ballot(one);
                peers &= one ? bal : ~bal;
            }
            rank[u] = (uint32_t)__popcll(peers & below);
            cnt[u] = (uint32_t)__popcll(peers);
            leader[u] = __ffsll((unsigned long long)peers) - 1;
            topic[u] = 0;
            if (valid[u] && grouped_topic && topic_of) {
                topic[u] = topic_of[(turn_i * kSub + u) * kWave + lane];
            } else if (valid[u] && grouped_topic) {
                const int64_t i = (int64_t)(turn_i * kSub + u) * kWave + lane;
                int64_t lo = 0, hi = n_topics;                         // largest t with part_off[t] <= i
                while (hi - lo > 1) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (part_off[mid] <= i) lo = mid; else hi = mid;
                }
                topic[u] = (int32_t)lo;
            }
        }
        // the ordered section: wait for the turn before this one, advance the cursors, pass the turn on
        while (__hip_atomic_load(&turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (uint32_t)turn_i) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
            first[u] = 0;
            if (valid[u] && lane == leader[u]) first[u] = atomicAdd(&start[gi[u]], cnt[u]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&turn, (uint32_t)turn_i + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
            const uint32_t f = (uint32_t)__shfl((int)first[u], leader[u] < 0 ? 0 : leader[u]);
            if (valid[u]) {
                const int i = (turn_i * kSub + u) * kWave + lane;
                const uint32_t pos = f + rank[u];
                if (grouped_entry) grouped_entry[pos] = i;
                if (grouped_partition) grouped_partition[pos] = part[u];
                if (grouped_topic) grouped_topic[pos] = topic[u];
            }
        }
    }
}

// The whole grouping by one workgroup of NT threads (every thread of it calls this).  start: [M] words of LDS, wsum: [NT / 64],
// turn: one word.  Needs n_members + 2 <= M and n_members + 1 < 2^(kSmallGroupBits + 1).
template <int NT, int M, int kSub = 1>
__device__ __forceinline__ void group_small_body(int n, int32_t n_members, int64_t n_topics, const int64_t* part_off,
                                                 const int32_t* out_partition, const int32_t* member_rank, int64_t* member_off,
                                                 int32_t* grouped_topic, int32_t* grouped_partition, int32_t* grouped_entry,
                                                 uint32_t* start, uint32_t* wsum, uint32_t* turn) {
    static_assert(M % NT == 0 && NT % kWave == 0, "M counters over NT threads");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t G = (uint32_t)n_members + 1;                      // a rank >= n_members (out of contract) sorts behind every
                                                                       // member, as in member_emit_kernel: member_off[n_members]
    for (int k = tid; k < M; k += NT) start[k] = 0;                   // is then where such entries start
    if (tid == 0) *turn = 0;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        uint32_t gi = (uint32_t)(member_rank[i] + 1);
        gi = gi < G ? gi : G;
        atomicAdd(&start[gi], 1u);
    }
    __syncthreads();
    // exclusive scan over the M counts: M / NT per thread, a wavefront scan, the wavefronts' sums
    constexpr int PER = M / NT;
    uint32_t c[PER], run = 0;
#pragma unroll
    for (int r = 0; r < PER; ++r) { c[r] = start[PER * tid + r]; run += c[r]; }
    uint32_t incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = incl - run;
    for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
    for (int r = 0; r < PER; ++r) { start[PER * tid + r] = base; base += c[r]; }
    __syncthreads();
    // member r's list starts where the groups 0 .. r end; positions before member_off[0] belong to topics without consumers
    for (int k = tid; k <= n_members; k += NT) member_off[k] = (int64_t)start[k + 1];
    __syncthreads();                                                    // (the cursors move from here on)
    group_small_place<kSub, NT>(n, G, n_topics, part_off, out_partition, member_rank, grouped_topic, grouped_partition, grouped_entry,
                                start, turn, lane, wave);
}

// ---- the same with the inputs staged in LDS first ---------------------------------------------------------------------------
// The chunks of the placement above are DEPENDENT round trips per wavefront: the loads of a chunk's ranks and partition ids
// (one trip to HBM each) and a binary search over part_off per entry -- log2(T) trips, to HOST memory over PCIe (~1.5 us each)
// in a zero-copy call -- eight chunks deep per wavefront in the 256-thread tail: 20 us for 2 000 entries.  Here every global
// input is read ONCE, coalesced, all loads of a thread issued back to back: ranks and ids into LDS, and the topic of every
// entry from ONE pass over part_off -- +1 at every topic's first position, an inclusive scan (topic of entry i = the number of
// topic starts at or before i, minus one: the largest t with part_off[t] <= i, empty topics included).  The placement then
// runs on LDS.  s_rank / s_part / s_topic / o_part / o_topic / lead / first: [cap] words each, n <= cap.
template <int NT, int M>
__device__ __forceinline__ void group_small_body_staged(int n, int32_t n_members, int64_t n_topics, const int64_t* part_off,
                                                        const int32_t* out_partition, const int32_t* member_rank, int64_t* member_off,
                                                        int32_t* grouped_topic, int32_t* grouped_partition, int32_t* grouped_entry,
                                                        uint32_t* start, uint32_t* wsum, uint32_t* turn, int32_t* s_rank,
                                                        int32_t* s_part, int32_t* s_topic, int32_t* o_part, int32_t* o_topic,
                                                        uint32_t* lead, uint32_t* first) {
    static_assert(M % NT == 0 && NT % kWave == 0, "M counters over NT threads");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t G = (uint32_t)n_members + 1;
    LA_GCLK_START;
    for (int k = tid; k < M; k += NT) start[k] = 0;
    for (int i = tid; i < n; i += NT) s_topic[i] = 0;
    if (tid == 0) *turn = 0;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {                               // (independent loads: they go out back to back)
        s_rank[i] = member_rank[i];
        s_part[i] = out_partition ? out_partition[i] : 0;
    }
    if (grouped_topic)
        for (int64_t t = tid; t < n_topics; t += NT) {                // topic t starts at part_off[t]; one at or beyond n holds nothing
            const int64_t at = part_off[t];
            if (at >= 0 && at < n) atomicAdd((uint32_t*)&s_topic[at], 1u);
        }
    __syncthreads();
    LA_GCLK(0);                                                            // zero + loads + topic heads
    // A (all wavefronts, chunks of 64 consecutive entries in parallel): ranks and peer counts by ballots; the FIRST lane of every
    // group present in a chunk leaves (group, count) in lead[chunk][lane] and adds the count to its group's counter -- one LDS
    // atomic per (chunk, group) instead of one per entry: with a handful of members every entry of a rebalance lands on the same
    // few counters, and 2 000 same-address atomics were 5 us of a 13 us grouping.
    constexpr int W = NT / kWave;
    const int nch = (n + kWave - 1) / kWave;
    const uint64_t below = ((uint64_t)1 << lane) - 1;
    for (int c = wave; c < nch; c += W) {
        const int i = c * kWave + lane;
        const bool valid = i < n;
        uint32_t gi = valid ? (uint32_t)(s_rank[i] + 1) : 0u;
        gi = gi < G ? gi : G;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < kSmallGroupBits + 1; ++bit) {               // (G itself may need one bit more than G - 1)
            const bool one = (gi >> bit) & 1u;
            const uint64_t bal = __ballot(one);
            peers &= one ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & below), cnt = (uint32_t)__popcll(peers);
        const int leader = __ffsll((unsigned long long)peers) - 1;
        if (valid) {
            s_rank[i] = (int32_t)(((uint32_t)leader << 16) | rank);         // (the group id is not needed again)
            lead[i] = lane == leader ? ((gi << 8) | cnt) : 0u;              // cnt in 1 .. 64: seven bits
            if (lane == leader) atomicAdd(&start[gi], cnt);
        }
    }
    LA_GCLK(1);                                                            // A: ballots, leader counts
    // inclusive scan of the topic starts over the n entries: PERT consecutive entries per thread, a wavefront scan of the
    // threads' sums, the wavefronts' sums (the form of the counts' scan below; blocks of NT entries with two barriers and a
    // serial walk over the wavefronts' sums each were 3.7 us of a 12 us grouping at 2 000 entries)
    if (grouped_topic) {
        constexpr int PERT = (kSmallGroupN + NT - 1) / NT;
        uint32_t tv[PERT], trun = 0;
#pragma unroll
        for (int r = 0; r < PERT; ++r) {
            const int i = PERT * tid + r;
            tv[r] = i < n ? (uint32_t)s_topic[i] : 0u;
            trun += tv[r];
        }
        uint32_t tincl = trun;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(tincl, o);
            if (lane >= o) tincl += y;
        }
        if (lane == 63) wsum[wave] = tincl;
        __syncthreads();
        uint32_t tbase = tincl - trun;
        for (int w = 0; w < wave; ++w) tbase += wsum[w];
#pragma unroll
        for (int r = 0; r < PERT; ++r) {
            const int i = PERT * tid + r;
            tbase += tv[r];
            if (i < n) s_topic[i] = (int32_t)tbase - 1;
        }
    }
    __syncthreads();                                                    // (wsum is used again; the leaders' counts are in)
    LA_GCLK(2);                                                            // topic scan
    // exclusive scan over the M counts: M / NT per thread, a wavefront scan, the wavefronts' sums
    constexpr int PER = M / NT;
    uint32_t c[PER], run = 0;
#pragma unroll
    for (int r = 0; r < PER; ++r) { c[r] = start[PER * tid + r]; run += c[r]; }
    uint32_t incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = incl - run;
    for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
    for (int r = 0; r < PER; ++r) { start[PER * tid + r] = base; base += c[r]; }
    __syncthreads();
    for (int k = tid; k <= n_members; k += NT) member_off[k] = (int64_t)start[k + 1];
    __syncthreads();                                                    // (the cursors move from here on)
    // Places.  An entry's place is (its group's cursor when its chunk's turn comes) + (its rank among the chunk's entries of that
    // group, found in A).  The form above passes the turn from chunk to chunk through a spin on an LDS word: ~0.4 us per chunk,
    // one after another.  Here the serial part is cut down to what is serial:
    //   B (every wavefront walks ALL chunks in order, for the groups g with g % wavefronts == its number): cursor read, cursor +
    //     count in one returning LDS atomic, the old cursor left in first[chunk][lane] (NOT in lead: the other wavefronts, which
    //     may be chunks behind or ahead, still read this chunk's (group, count) words).  Groups of different wavefronts are disjoint,
    //     the groups inside a chunk distinct, and one wavefront's LDS operations execute in the order it issues them: no spinning;
    //   C (chunks in parallel): place = first[chunk][first lane of my group] + my rank; the lists are built in LDS (o_part /
    //     o_topic) and leave in ONE coalesced pass -- a scattered 4-byte store into the host's memory (a zero-copy call's lists
    //     go straight there) is a PCIe write transaction of its own.
    // lead, first: [cap] words each.  Stable by construction (chunk order, then lane order).
    __syncthreads();
    LA_GCLK(3);                                                            // counts scan + member_off
    {
        // (returning atomics of ONE wavefront execute in the order they are issued, chunk by chunk; the groups of a chunk are
        //  distinct and the groups of different wavefronts disjoint: the old value IS the cursor at the chunk's turn.  Four chunks
        //  in flight: a read -> add -> write chain per chunk was 0.13 us x 32 chunks.)
        constexpr int kAhead = 4;
        for (int c0 = 0; c0 < nch; c0 += kAhead) {
            uint32_t v[kAhead], f[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const int i = (c0 + u) * kWave + lane;
                v[u] = (c0 + u < nch && i < n) ? lead[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const uint32_t cnt = v[u] & 0xFFu, g = v[u] >> 8;
                f[u] = 0;
                if (cnt != 0 && (g % W) == (uint32_t)wave) f[u] = atomicAdd(&start[g], cnt);
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const uint32_t cnt = v[u] & 0xFFu, g = v[u] >> 8;
                if (cnt != 0 && (g % W) == (uint32_t)wave) first[(c0 + u) * kWave + lane] = f[u];
            }
        }
    }
    __syncthreads();
    LA_GCLK(4);                                                            // B: cursors
    for (int c = wave; c < nch; c += W) {
        const int i = c * kWave + lane;
        if (i < n) {
            const uint32_t pk = (uint32_t)s_rank[i];
            const uint32_t pos = first[c * kWave + (int)(pk >> 16)] + (pk & 0xFFFFu);
            o_part[pos] = s_part[i];
            if (grouped_topic) o_topic[pos] = s_topic[i];
            if (grouped_entry) grouped_entry[pos] = i;
        }
    }
    __syncthreads();
    LA_GCLK(5);                                                            // C: places
    for (int i = tid; i < n; i += NT) {
        grouped_partition[i] = o_part[i];
        if (grouped_topic) grouped_topic[i] = o_topic[i];
    }
    LA_GCLK(6);                                                            // lists out
}

// ---- the same lists for a few thousand to 65 536 entries and FEW members: two launches -----------------------------------------
// Between the one-workgroup form (<= kSmallGroupN entries) and the radix form (five dependent launches, 17-25 us whatever the
// size) sits what a mid-size rebalance is: 3 000 - 65 000 partitions, a handful of members (10 000 entries: 85 -> 80 us per call).  With G = members + 2 <= kMidGroupM
// groups a table of per-block counts is small, and the stable counting sort splits over workgroups:
//   launch 1 (group_mid_count): block b = entries [b * kMidBlock, ...): the chunks' group leaders (ballots, as above) add their
//            peer counts into LDS counters; the block's G counts go to cnt[b][*];
//   launch 2 (group_mid_place): every block reads the whole table (blocks x G words), sums it per group (the groups' starts, an
//            exclusive scan) and over the blocks before it (its own first places), then places its entries exactly like the
//            one-workgroup form does -- ranks inside a chunk by ballots, cursors advanced chunk by chunk by in-order returning
//            LDS atomics -- and stores them straight to their places.  The topic of every entry: the block's first topic by a
//            256-ary search over part_off (two rounds up to 65 536 topics), the rest from the topic starts inside the block (+1
//            at a topic's first position, an inclusive scan).
// Stable by construction: blocks in order, chunks in order, lanes in order.
constexpr int kMidBlock = 2048;          // entries per workgroup
constexpr int kMidThreads = 256;
constexpr int kMidGroupM = 512;          // groups (members + 2) the table form holds
constexpr int kMidMaxBlocks = 32;        // 65 536 entries: every block sums the whole table (blocks x groups words); 125 blocks made it slower than the radix form

#ifdef LA_GROUP_MID_KERNELS   // the two kernels are compiled by ONE translation unit (la_large.hip defines this before the include)
__device__ __forceinline__ void group_mid_chunk_ranks(int nb, uint32_t G, const int32_t* s_rank_in, int32_t* s_pk, uint32_t* lead,
                                                      uint32_t* counts, int lane, int wave) {
    constexpr int W = kMidThreads / kWave;
    const int nch = (nb + kWave - 1) / kWave;
    const uint64_t below = ((uint64_t)1 << lane) - 1;
    for (int c = wave; c < nch; c += W) {
        const int i = c * kWave + lane;
        const bool valid = i < nb;
        uint32_t gi = valid ? (uint32_t)(s_rank_in[i] + 1) : 0u;
        gi = gi < G ? gi : G;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 10; ++bit) {                                // G <= kMidGroupM - 1 < 2^9 (+ the clamp value G itself)
            const bool one = (gi >> bit) & 1u;
            const uint64_t bal = __ballot(one);
            peers &= one ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & below), cnt = (uint32_t)__popcll(peers);
        const int leader = __ffsll((unsigned long long)peers) - 1;
        if (valid) {
            if (s_pk) s_pk[i] = (int32_t)(((uint32_t)leader << 16) | rank);
            if (lead) lead[i] = lane == leader ? ((gi << 8) | cnt) : 0u;
            if (lane == leader) atomicAdd(&counts[gi], cnt);
        }
    }
}

__global__ __launch_bounds__(kMidThreads) void group_mid_count_kernel(int n, int32_t n_members, const int32_t* member_rank, uint32_t* cnt) {
    __shared__ uint32_t counts[kMidGroupM];
    __shared__ int32_t s_rank[kMidBlock];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e0 = blockIdx.x * kMidBlock, nb = min(kMidBlock, n - e0);
    const uint32_t G = (uint32_t)n_members + 1;
    for (int k = tid; k < kMidGroupM; k += kMidThreads) counts[k] = 0;
    for (int i = tid; i < nb; i += kMidThreads) s_rank[i] = member_rank[e0 + i];
    __syncthreads();
    group_mid_chunk_ranks(nb, G, s_rank, nullptr, nullptr, counts, lane, wave);
    __syncthreads();
    for (uint32_t g = tid; g <= G; g += kMidThreads) cnt[(size_t)blockIdx.x * kMidGroupM + g] = counts[g];
    if (blockIdx.x == 0 && tid == 0) cnt[(size_t)gridDim.x * kMidGroupM] = 0;      // the place kernel's "blocks done" counter
}

__global__ __launch_bounds__(kMidThreads) void group_mid_place_kernel(int n, int32_t n_members, int64_t n_topics, const int64_t* part_off,
                                                                      const int32_t* out_partition, const int32_t* member_rank,
                                                                      uint32_t* cnt, int64_t* member_off, int32_t* grouped_topic,
                                                                      int32_t* grouped_partition, int32_t* grouped_entry,
                                                                      const uint32_t* status, uint32_t* fin_flag) {
    __shared__ uint32_t cursor[kMidGroupM];           // this block's first place of every group, then its cursors
    __shared__ uint32_t counts[kMidGroupM];           // (scratch of the chunk ranks: the block's own counts again)
    __shared__ uint32_t wsum[kMidThreads / kWave];
    __shared__ int32_t s_rank[kMidBlock], s_pk[kMidBlock], s_topic[kMidBlock];
    __shared__ uint32_t lead[kMidBlock], first[kMidBlock];
    __shared__ int s_t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int W = kMidThreads / kWave;
    const int nblocks = gridDim.x;
    const int e0 = blockIdx.x * kMidBlock, nb = min(kMidBlock, n - e0), e1 = e0 + nb;
    const uint32_t G = (uint32_t)n_members + 1;
    // inputs (independent loads first), the table's column sums
    for (int i = tid; i < nb; i += kMidThreads) { s_rank[i] = member_rank[e0 + i]; s_topic[i] = 0; }
    for (int k = tid; k < kMidGroupM; k += kMidThreads) counts[k] = 0;
    uint32_t tot[kMidGroupM / kMidThreads], mine[kMidGroupM / kMidThreads];     // groups tid, tid + 256
#pragma unroll
    for (int r = 0; r < kMidGroupM / kMidThreads; ++r) {
        const uint32_t g = (uint32_t)(tid + r * kMidThreads);
        uint32_t all = 0, before = 0;
        if (g <= G)
            for (int b = 0; b < nblocks; ++b) {
                const uint32_t c = cnt[(size_t)b * kMidGroupM + g];
                all += c;
                before += b < (int)blockIdx.x ? c : 0u;
            }
        tot[r] = all;
        mine[r] = before;
    }
    // exclusive scan of the groups' totals over g (group g = tid + r * 256: scan the r = 0 half, then the r = 1 half behind it)
    uint32_t base_of[kMidGroupM / kMidThreads];
    uint32_t carry = 0;
#pragma unroll
    for (int r = 0; r < kMidGroupM / kMidThreads; ++r) {
        uint32_t incl = tot[r];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = carry, all = 0;
        for (int w = 0; w < W; ++w) { before += w < wave ? wsum[w] : 0u; all += wsum[w]; }
        base_of[r] = before + incl - tot[r];
        carry += all;
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < kMidGroupM / kMidThreads; ++r) {
        const uint32_t g = (uint32_t)(tid + r * kMidThreads);
        cursor[g] = base_of[r] + mine[r];
        // member m's list starts where the groups 0 .. m end: member_off[m] = start of group m + 1
        if (blockIdx.x == 0 && g >= 1 && g <= (uint32_t)n_members + 1) member_off[g - 1] = (int64_t)base_of[r];
    }
    // the block's first topic: the largest t with part_off[t] <= e0, by a 256-ary search
    if (grouped_topic) {
        int64_t lo = 0, hi = n_topics;                                       // invariant: part_off[lo] <= e0 (part_off[0] = 0), answer in [lo, hi)
        while (hi - lo > 1) {
            const int64_t step = (hi - lo + kMidThreads - 1) / kMidThreads;
            if (tid == 0) s_t0 = 0;
            __syncthreads();
            const int64_t at = lo + (int64_t)tid * step;
            if (at < hi && part_off[at] <= (int64_t)e0) atomicMax(&s_t0, tid);
            __syncthreads();
            const int64_t nlo = lo + (int64_t)s_t0 * step;
            hi = nlo + step < hi ? nlo + step : hi;
            lo = nlo;
            __syncthreads();
        }
        const int64_t t0 = lo;
        for (int64_t t = t0 + 1 + tid; t < n_topics; t += kMidThreads) {   // topic starts inside the block
            const int64_t at = part_off[t];
            if (at >= (int64_t)e1) break;
            atomicAdd((uint32_t*)&s_topic[at - e0], 1u);
        }
        __syncthreads();
        constexpr int PERT = kMidBlock / kMidThreads;
        uint32_t tv[PERT], trun = 0;
#pragma unroll
        for (int r = 0; r < PERT; ++r) {
            const int i = PERT * tid + r;
            tv[r] = i < nb ? (uint32_t)s_topic[i] : 0u;
            trun += tv[r];
        }
        uint32_t tincl = trun;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(tincl, o);
            if (lane >= o) tincl += y;
        }
        if (lane == 63) wsum[wave] = tincl;
        __syncthreads();
        uint32_t tbase = tincl - trun;
        for (int w = 0; w < wave; ++w) tbase += wsum[w];
#pragma unroll
        for (int r = 0; r < PERT; ++r) {
            const int i = PERT * tid + r;
            tbase += tv[r];
            if (i < nb) s_topic[i] = (int32_t)(t0 + tbase);
        }
    }
    __syncthreads();
    group_mid_chunk_ranks(nb, G, s_rank, s_pk, lead, counts, lane, wave);
    __syncthreads();
    {
        const int nch = (nb + kWave - 1) / kWave;
        constexpr int kAhead = 4;
        for (int c0 = 0; c0 < nch; c0 += kAhead) {
            uint32_t v[kAhead], f[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const int i = (c0 + u) * kWave + lane;
                v[u] = (c0 + u < nch && i < nb) ? lead[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const uint32_t c = v[u] & 0xFFu, g = v[u] >> 8;
                f[u] = 0;
                if (c != 0 && (g % W) == (uint32_t)wave) f[u] = atomicAdd(&cursor[g], c);
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const uint32_t c = v[u] & 0xFFu, g = v[u] >> 8;
                if (c != 0 && (g % W) == (uint32_t)wave) first[(c0 + u) * kWave + lane] = f[u];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nb; i += kMidThreads) {
        const uint32_t pk = (uint32_t)s_pk[i];
        const uint32_t pos = first[(i & ~(kWave - 1)) + (int)(pk >> 16)] + (pk & 0xFFFFu);
        grouped_partition[pos] = out_partition ? out_partition[e0 + i] : 0;
        if (grouped_topic) grouped_topic[pos] = s_topic[i];
        if (grouped_entry) grouped_entry[pos] = e0 + i;
    }
    if (fin_flag) {
        // a zero-copy call ends here (as in tile_tail): every block releases what it wrote -- the lists sit in host memory -- and
        // counts itself done; the last one stores `done | status` where the calling thread spins
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            uint32_t* done = cnt + (size_t)gridDim.x * kMidGroupM;
            if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
                const uint32_t st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                __hip_atomic_store(fin_flag, 0x80000000u | st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
#endif  // LA_GROUP_MID_KERNELS

}  // namespace la
