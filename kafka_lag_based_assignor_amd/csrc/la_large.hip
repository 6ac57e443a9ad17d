// la_large.hip -- topics that do not fit one wave tile (> 1024 partitions or > 64 consumers).
//
// Three phases per topic, all on the device, no host round trip in between:
//
//   kernel 1  build_keys     lag from offsets (Main.java:376-404), 64-bit sort key + 32-bit id,
//                            and the 12 digit histograms of the 96-bit composite in one read.
//   kernel 2  radix sort     LSD, 8 bits per pass, stable, over (key64, id32):
//                            id digits first, then key digits -> (lag desc, id asc), the
//                            comparator of Main.java:228-235.  A pass whose digit is constant
//                            over the whole topic is skipped on the device (plan kernel), so
//                            lags < 2^40 with ids < 2^24 cost 8 passes, not 12; input that is
//                            already in id order skips the 4 id passes.
//                            per pass ONE kernel: stable scatter with decoupled look-back (every tile
//                            publishes its digit counts as {tag, count} granules and walks back over its
//                            predecessors'); ranks from returning LDS atomics.  The four-kernel form (tile
//                            digit counts -> per-digit scan over tiles -> scatter) stays as a test hook.
//   kernel 3  greedy         ONE workgroup (the chain of rounds is serial): consumer bins live in
//                            registers, E per thread; each round bitonic-sorts the bins by
//                            (total lag, member) -- in registers, across lanes with DPP, across
//                            waves through LDS -- and hands the round's C partitions out in
//                            that order.  ceil(P/C) rounds instead of P argmins.  A round whose bins are a
//                            few ascending runs (flat tails of lags) merges the runs instead of sorting.
//                            LA_ALGO_ARGMIN keeps the literal form: bins (count, total) in LDS,
//                            one wavefront-argmin + LDS combine per partition.
#include "la_kernels.h"
#include "la_device.h"
#include "la_sort64.h"
#define LA_GROUP_MID_KERNELS
#include "la_group_small.h"

#include <algorithm>
#include <type_traits>
#include <vector>

namespace la {

namespace {

// Arrays that a kernel reads ONCE (the caller's inputs in build_keys_kernel, a pass's source buffers) as non-temporal loads
// (round 6, after the tile kernel's A/B: profiles/r06_ab_nt_loads.txt).  -DLA_LARGE_NT_LOADS=0: plain loads.
#ifndef LA_LARGE_NT_LOADS
#define LA_LARGE_NT_LOADS 1
#endif
template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
#if LA_LARGE_NT_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// Round 1 of the greedy turns the bins around instead of sorting them from scratch (greedy_rounds_packed).  -DLA_ROUND1_REVERSE=0: round 5's form.
#ifndef LA_ROUND1_REVERSE
#define LA_ROUND1_REVERSE 1
#endif
// The moved-bins sort takes up to twice its narrow capacity in a wide form (moved_sort_bins).  -DLA_WIDE_MOVED=0: narrow only.
// LA_WIDE_MOVED_LIMIT x (bins per thread of the round) = the most bins the wide form is tried on (512 = all that fit).
#ifndef LA_WIDE_MOVED
#define LA_WIDE_MOVED 1
#endif
// The rounds kernel's lags as 32-bit words and its results as 16-bit consumer indices where they fit (rounds_io).  -DLA_ROUNDS_NARROW_IO=0: off.
#ifndef LA_ROUNDS_NARROW_IO
#define LA_ROUNDS_NARROW_IO 1
#endif
#ifndef LA_ROUNDS_NARROW_EC
#define LA_ROUNDS_NARROW_EC 8          // from how many bins per thread on (8: more than 4 096 consumers; at 4 bins per thread it lost 3 %)
#endif
#ifndef LA_WIDE_MOVED_LIMIT
#define LA_WIDE_MOVED_LIMIT 512
#endif

constexpr int kDigits = 12;            // 4 id digits + 8 key digits
constexpr int kRadix = 256;
constexpr int kSortThreads = 256;
constexpr int kSortWaves = kSortThreads / kWave;
constexpr int kItems = 16;
constexpr int kTile = kSortThreads * kItems;     // 4096 elements per workgroup
constexpr int kScanRows = 32;                    // tiles per workgroup of the offset scan
constexpr uint32_t kSampleHeavy = 8;             // 1 in 256 keys is sampled: a counter at 8 ~ a lag shared by ~2 000 partitions
constexpr int kRepairThreads = 1024, kRepairEC = 8;        // tie_repair_kernel
constexpr int kRepairCap = kRepairThreads * kRepairEC;     // 8 192 positions a workgroup sorts at once
constexpr int kRepairWindow = kRepairCap / 2;              // so a run of up to 4 096 starting anywhere in a window fits

// Pass slots: 0 .. kDigits-1 are the sort's passes (digit = slot); kDigits .. 2*kDigits-1 are the same digits again, for the one
// case in which a keys-first sort has to be redone in full (see tie_repair_kernel); they are no-ops otherwise.
constexpr int kSlots = 2 * kDigits;
constexpr int kFinal = kSlots;         // ctl->cur[kFinal]: the buffer that holds the sorted data

struct SortCtl {
    uint32_t skip[kSlots];
    uint32_t cur[kSlots + 1];          // which buffer (0/1) holds the data before pass slot s; [kFinal]: after the last
    uint32_t unsorted_ids;             // != 0: input ids are not ascending
    uint32_t tie_heavy;                // build_keys: a key came up often in the sample (or filled a whole wavefront): many equal lags
    uint32_t keys_first;               // plan: the id passes are skipped, ties are put in id order by tie_repair_kernel
    uint32_t redo;                     // tie_repair: a run of equal keys did not fit its workgroup -> the redo slots sort in full
    uint32_t redo_arrived;             // onesweep_redo_kernel's grid barrier: workgroups that finished a redo pass (zeroed with ctl)
    uint32_t pad[10];
};

struct SortBufs {
    SortCtl* ctl;
    uint32_t* hist;                    // [kDigits][256]
    uint64_t* key[2];
    uint32_t* val[2];
    uint32_t* tile_off;                // [n_tiles][256] (a tile's 256 digit counts / offsets are one 1 KB row)
    uint32_t* group_sum;               // [n_groups][256] digit counts of kScanRows consecutive tiles
    // single-kernel passes (decoupled look-back): zeroed per sort together with ctl and hist
    uint32_t* ticket;                  // [kSlots] arrival counter of every pass slot
    uint32_t* gbase;                   // [kDigits][256] exclusive scan of hist: global first position of every digit
    unsigned long long* tile_state;    // [n_tiles][256] granules {tag = (pass + 1) << 2 | status, count}; null: multi-kernel passes
    int64_t n;
    int n_tiles;
    int n_groups;
    int atomic_rank;                   // ranks from returning LDS atomics (the device passed lds_atomic_order_test_kernel)
    int sweep_threads;                 // workgroup size of the single-kernel passes: 512 (tiles of 8 192) or 256 (4 096)
    // keys-first sorts (large n, shuffled ids): a hash table of SAMPLED keys (zeroed with ctl) tells whether some lag is so
    // frequent that its run could not be repaired in one workgroup; null: this sort never goes keys first
    uint32_t* samp;
    uint32_t samp_bits;                // table of 2^samp_bits counters
    uint32_t* tie_flag;                // [ceil(n / kRepairWindow)] != 0: a run of more than four equal keys starts in the window
    int keys_first_force;              // test hook: keys first whatever the sample says (long runs then take the redo slots)
};

// One large topic of a batched launch (large_topics_launch): kernels launched over SEVERAL topics at once read where their
// topic's partitions / consumers are and where its sort lives from a device array -- blockIdx.y (or an order list) picks the
// item -- instead of from the kernel arguments.  `items == nullptr` is the single-topic form: the arguments are the kernel's own.
// An item holds NO pointers: the per-partition arrays are the batch's (the kernel's own LargeArgs, shared by every item) and
// the sort's buffers are byte offsets into the scratch block, whose base is a kernel argument.  A pointer LOADED from memory
// is a generic pointer to the compiler -- every access through it a flat_* instruction (tests/test_isa_hazards.py) -- while
// one derived from a kernel argument is known to be global.
struct LargeItem {
    int64_t p0, n_part, c0, n_cons;
    int64_t n;
    int32_t n_tiles, n_groups;
    uint64_t o_ctl, o_hist, o_ticket, o_state, o_gbase, o_k0, o_k1, o_v0, o_v1;
    uint64_t o_samp;                   // 0: no sample table (this topic never sorts keys first)
    uint64_t o_flag;
    uint32_t samp_bits, pad;
};

// b.key[x] / b.val[x] with a run-time x, as a select: indexing the pointer pair of a LOCAL SortBufs dynamically would put the
// struct in scratch memory and bring the pointers back generic (flat_* accesses)
__device__ __forceinline__ uint64_t* key_buf(const SortBufs& b, uint32_t x) { return x ? b.key[1] : b.key[0]; }
__device__ __forceinline__ uint32_t* val_buf(const SortBufs& b, uint32_t x) { return x ? b.val[1] : b.val[0]; }

__device__ __forceinline__ void bind_item(LargeArgs& a, SortBufs& b, const LargeItem& it, char* scratch) {
    a.p0 = it.p0; a.n_part = it.n_part; a.c0 = it.c0; a.n_cons = it.n_cons;
    b.ctl = (SortCtl*)(scratch + it.o_ctl);
    b.hist = (uint32_t*)(scratch + it.o_hist);
    b.ticket = (uint32_t*)(scratch + it.o_ticket);
    b.tile_state = (unsigned long long*)(scratch + it.o_state);
    b.gbase = (uint32_t*)(scratch + it.o_gbase);
    b.key[0] = (uint64_t*)(scratch + it.o_k0);
    b.key[1] = (uint64_t*)(scratch + it.o_k1);
    b.val[0] = (uint32_t*)(scratch + it.o_v0);
    b.val[1] = (uint32_t*)(scratch + it.o_v1);
    b.samp = it.o_samp ? (uint32_t*)(scratch + it.o_samp) : nullptr;
    b.tie_flag = (uint32_t*)(scratch + it.o_flag);
    b.samp_bits = it.samp_bits;
    b.n = it.n;
    b.n_tiles = it.n_tiles;
    b.n_groups = it.n_groups;
}

// (a0 / b0 of a batched launch: the batch's arrays and flags; atomic_rank / sweep_threads of the launch's tile class)
#define LA_PICK_ITEM(a, b, a0, b0, items, scratch, which)  \
    LargeArgs a = a0;                                       \
    SortBufs b = b0;                                        \
    if (items) bind_item(a, b, (items)[which], scratch);

// bijection [0, n) -> [0, n): workgroups with equal (w % 8) get consecutive results
__device__ __forceinline__ int xcd_contiguous(int w, int n) {
    const int per = n / 8, rem = n % 8;                  // XCD x owns per (+1 if x < rem) values
    const int x = w % 8, k = w / 8;
    return x * per + (x < rem ? x : rem) + k;
}

__device__ __forceinline__ uint32_t digit_of(int pass, uint64_t key, uint32_t val) {
    return pass < 4 ? (val >> (8 * pass)) & 0xFFu : (uint32_t)(key >> (8 * (pass - 4))) & 0xFFu;
}

// grid of build_keys_kernel.  Every block ends by adding its non-zero digit counters (up to 12 x 256) to the global histograms with
// atomics, so beyond 512 blocks a block takes 2 048 partitions instead of 256 (grid stride), up to 1 024 blocks (2 048 from 16 M
// partitions on, where the flush is noise and the loads want every CU several times over).  1 M partitions (cfg5): 0.057 ->
// 0.040 ms; 4 M: 0.079 -> 0.074 ms (profiles/r05_aa_keys_grid.txt).
static int keys_grid(int64_t n) {
    static const int forced = [] { const char* e = getenv("LA_KEYS_GRID"); return e ? atoi(e) : 0; }();   // (lab)
    const int64_t fine = (n + 255) / 256;
    if (forced > 0) return (int)(fine > forced ? forced : fine);
    if (fine <= 512) return (int)fine;
    const int64_t cap = n >= ((int64_t)16 << 20) ? 2048 : 1024;
    const int64_t coarse = (n + 2047) / 2048;
    const int64_t g = coarse > cap ? cap : coarse;
    return (int)(g < 512 ? 512 : g);
}

// ---- kernel 1: keys + all digit histograms ---------------------------------------------------
__global__ __launch_bounds__(256) void build_keys_kernel(LargeArgs a0, SortBufs b0, const LargeItem* items, char* scratch) {
    LA_PICK_ITEM(a, b, a0, b0, items, scratch, blockIdx.y)
    if ((int64_t)blockIdx.x * blockDim.x >= b.n) return;              // (a grid sized for the largest topic of the launch)
    __shared__ uint32_t h[kDigits * kRadix];
    for (int i = threadIdx.x; i < kDigits * kRadix; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const bool latest = a.reset_latest != 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint64_t last_sample = 0;                                          // (lane 0 of a sampling wavefront)
    bool have_sample = false, sampling = true;
    bool out_of_bounds = false;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < b.n; base += stride) {
        const int64_t i = base + threadIdx.x;
        const bool valid = i < b.n;
        uint64_t key = 0;
        uint32_t val = 0;
        if (valid) {
            const int64_t g = a.p0 + i;
            int64_t lag;
            if (a.lag) lag = stream_load(a.lag + g);
            else lag = partition_lag(a.begin ? stream_load(a.begin + g) : 0, stream_load(a.end + g), stream_load(a.committed + g), latest);
            key = (uint64_t)lag ^ kLagKeyFlip;
            const int32_t id = a.pid[g];
            val = (uint32_t)id ^ kPidBias;
            b.key[0][i] = key;
            b.val[0][i] = val;
            // the caller's bounds decided which passes were launched at all (bounds_pass_mask): a partition outside them is an
            // error of the call (LA_EINVAL), never a differently sorted topic
            if (a.max_lag_hint >= 0 && ((uint64_t)lag > (uint64_t)a.max_lag_hint || (uint64_t)(int64_t)id > (uint64_t)a.max_id_hint))
                out_of_bounds = true;
            if (i + 1 < b.n && a.pid[g + 1] < id) b.ctl->unsorted_ids = 1;
        }
        const uint64_t vmask = __ballot(valid);
        if (b.samp) {
            // How frequent is the most frequent lag?  (keys-first sorts repair runs of equal keys inside one workgroup: a run
            // must fit.)  64 equal neighbours say "very"; otherwise every fourth wavefront counts its first key in a hash
            // table of n / 32 counters (n / 256 samples) with a FIRE-AND-FORGET atomic -- waiting for the counter's old value
            // put a ~2 us round trip into every fourth iteration, 70 us of a 300 us kernel -- and sample_scan_kernel looks for a
            // counter at kSampleHeavy afterwards (~ a lag that some thousand partitions share).  A wavefront that draws the same
            // key twice in a row says "frequent" at once and stops sampling: a hot counter sees a few atomics per wavefront, not
            // one per sample (reading the shared flag instead cost an L2 round trip per sample: 60 us of this kernel).
            const uint64_t k0 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(key >> 32)) << 32) |
                                __builtin_amdgcn_readfirstlane((uint32_t)key);
            if (vmask == ~0ull && __ballot(key == k0) == ~0ull) {
                if (__lane_id() == 0) b.ctl->tie_heavy = 1;
            } else if (((base + (threadIdx.x & ~63)) >> 6) % 4 == 0 && (vmask & 1ull) && __lane_id() == 0 && sampling) {
                if (have_sample && k0 == last_sample) {
                    b.ctl->tie_heavy = 1;
                    sampling = false;                                  // (this wavefront has said what it had to say)
                } else {
                    const uint32_t h = (uint32_t)((k0 * 0x9E3779B97F4A7C15ull) >> (64 - b.samp_bits));
                    (void)__hip_atomic_fetch_add(&b.samp[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                last_sample = k0;
                have_sample = true;
            }
        }
#pragma unroll
        for (int p = 0; p < kDigits; ++p) {
            const uint32_t d = digit_of(p, key, val);
            // common case for the high digits: the whole wave agrees -> one add
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
            const uint64_t same = __ballot(valid && d == d0);
            if (same == vmask) {
                if (valid && (int)__lane_id() == __ffsll((unsigned long long)vmask) - 1)
                    atomicAdd(&h[p * kRadix + d0], (uint32_t)__popcll(vmask));
            } else if (valid) {
                atomicAdd(&h[p * kRadix + d], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kDigits * kRadix; i += blockDim.x)
        if (h[i]) atomicAdd(&b.hist[i], h[i]);
    if (__any(out_of_bounds) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, kStatusBounds);
}

// ---- the sample of the lags: did any counter reach kSampleHeavy? ----------------------------------------------------------
__global__ __launch_bounds__(256) void sample_scan_kernel(SortBufs b0, const LargeItem* items, char* scratch) {
    LargeArgs unused{};
    SortBufs b = b0;
    if (items) bind_item(unused, b, items[blockIdx.y], scratch);
    if (!b.samp) return;
    const uint4* t = reinterpret_cast<const uint4*>(b.samp);
    const int64_t words = ((int64_t)1 << b.samp_bits) / 4, stride = (int64_t)gridDim.x * blockDim.x;
    bool heavy = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) {
        const uint4 v = t[i];
        heavy |= v.x >= kSampleHeavy || v.y >= kSampleHeavy || v.z >= kSampleHeavy || v.w >= kSampleHeavy;
    }
    if (__any(heavy) && (threadIdx.x & (kWave - 1)) == 0) b.ctl->tie_heavy = 1;
}

// ---- plan: which passes are no-ops, where the data lives before each pass --------------------
__global__ __launch_bounds__(kRadix) void plan_kernel(SortBufs b0, const LargeItem* items, char* scratch) {
    LargeArgs unused{};
    SortBufs b = b0;
    if (items) bind_item(unused, b, items[blockIdx.y], scratch);
    // block p: global first position of every digit of pass p (exclusive scan of its histogram): the single-kernel passes
    // add a tile's look-back result to it
    __shared__ uint32_t wsum[kRadix / kWave];
    {
        const int p = blockIdx.x, d = threadIdx.x, lane = d & 63, wave = d >> 6;
        const uint32_t h = b.hist[p * kRadix + d];
        uint32_t incl = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - h;
        for (int w = 0; w < wave; ++w) run += wsum[w];
        b.gbase[p * kRadix + d] = run;
    }
    if (blockIdx.x != 0) return;
    // block 0: which passes are no-ops, where the data lives before each pass
    __shared__ uint32_t skip[kDigits];
    if (threadIdx.x < kDigits) skip[threadIdx.x] = 0;
    __syncthreads();
    for (int p = 0; p < kDigits; ++p)
        if (b.hist[p * kRadix + threadIdx.x] == (uint32_t)b.n) skip[p] = 1;   // one bin holds everything
    __syncthreads();
    if (threadIdx.x == 0) {
        // Keys first: the id passes only order partitions with EQUAL lags.  When the ids arrive shuffled and no lag is
        // frequent, sort by the key digits alone (stable: equal keys stay in input order) and let tie_repair_kernel put every
        // run of equal keys in id order afterwards -- for lags without long runs that is 5 scatter passes + one read where
        // the full LSD sort takes 9.  Frequent lags (the sample) keep the full order of passes: nothing changes for them.
        bool key_digits = false;
        for (int p = 4; p < kDigits; ++p) key_digits |= skip[p] == 0;
        const bool keys_first = b.samp != nullptr && b.ctl->unsorted_ids != 0 && key_digits &&
                                (b.keys_first_force || b.ctl->tie_heavy == 0);
        b.ctl->keys_first = keys_first ? 1u : 0u;
        uint32_t cur = 0;
        for (int p = 0; p < kDigits; ++p) {
            uint32_t s = skip[p];
            if (p < 4 && (b.ctl->unsorted_ids == 0 || keys_first)) s = 1;     // stable passes keep the input's id order
            b.ctl->skip[p] = s;
            b.ctl->cur[p] = cur;
            cur ^= (s ? 0u : 1u);
        }
        for (int p = kDigits; p < kSlots; ++p) {              // the redo slots: idle unless replan_kernel arms them
            b.ctl->skip[p] = 1;
            b.ctl->cur[p] = cur;
        }
        b.ctl->cur[kFinal] = cur;
    }
}

// After tie_repair_kernel: a run of equal keys that did not fit its workgroup (ctl->redo) sends the sort through the redo
// slots -- the full order of passes, ids first, from wherever the data stands; any permutation is a valid LSD input.
__global__ void replan_kernel(SortBufs b0, const LargeItem* items, char* scratch) {
    LargeArgs unused{};
    SortBufs b = b0;
    if (items) bind_item(unused, b, items[blockIdx.x], scratch);
    if (threadIdx.x != 0 || !b.ctl->keys_first || !b.ctl->redo) return;
    uint32_t cur = b.ctl->cur[kDigits];
    for (int p = 0; p < kDigits; ++p) {
        bool constant = false;
        for (int d = 0; d < kRadix; ++d) constant |= b.hist[p * kRadix + d] == (uint32_t)b.n;
        const uint32_t s = constant ? 1u : 0u;                 // (the ids are not ascending: keys first was chosen)
        b.ctl->skip[kDigits + p] = s;
        b.ctl->cur[kDigits + p] = cur;
        cur ^= (s ? 0u : 1u);
    }
    b.ctl->cur[kFinal] = cur;
}

// ---- per pass: tile digit counts ------------------------------------------------------------------
// A resident-sized grid walks the tiles; the next tile's loads (16 B per lane, whole tile) are issued before
// the current tile's histogram updates, one private histogram per wavefront (LDS atomics contend only inside
// a wavefront).  Only the array that carries the pass's digit is read: 8 B (key passes) or 4 B (id passes)
// per element.
template <bool IDS>
__device__ __forceinline__ void count_tiles(const SortBufs& b, int pass, uint32_t (*h)[kRadix]) {
    using Vec = typename std::conditional<IDS, uint4, ulonglong2>::type;
    constexpr int PER = IDS ? 4 : 2;                              // elements per 16-byte load
    constexpr int N = kTile / (PER * kSortThreads);
    const int wave = threadIdx.x >> 6;
    const uint32_t cur = b.ctl->cur[pass];
    const char* src = IDS ? (const char*)b.val[cur] : (const char*)b.key[cur];
    const int esz = IDS ? 4 : 8;
    const int sh = IDS ? 8 * pass : 8 * (pass - 4);
    auto digit = [&](const Vec& v, int k) -> uint32_t {
        if constexpr (IDS) return ((k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w) >> sh) & 0xFFu;
        else return (uint32_t)((k == 0 ? v.x : v.y) >> sh) & 0xFFu;
    };
    Vec v[N];
    auto load_tile = [&](int tile) {
        const int64_t t0 = (int64_t)tile * kTile;
        if (t0 + kTile <= b.n) {
#pragma unroll
            for (int k = 0; k < N; ++k)
                v[k] = *reinterpret_cast<const Vec*>(src + (t0 + PER * (k * kSortThreads + (int)threadIdx.x)) * esz);
        }
    };
    int tile = blockIdx.x;
    if (tile < b.n_tiles) load_tile(tile);
    for (; tile < b.n_tiles; tile += gridDim.x) {
        for (int i = threadIdx.x; i < kSortWaves * kRadix; i += kSortThreads) (&h[0][0])[i] = 0;
        __syncthreads();
        const int64_t t0 = (int64_t)tile * kTile;
        if (t0 + kTile <= b.n) {
            Vec w[N];
#pragma unroll
            for (int k = 0; k < N; ++k) w[k] = v[k];
            if (tile + (int)gridDim.x < b.n_tiles) load_tile(tile + gridDim.x);
#pragma unroll
            for (int k = 0; k < N; ++k)
#pragma unroll
                for (int e = 0; e < PER; ++e) atomicAdd(&h[wave][digit(w[k], e)], 1u);
        } else {
            for (int64_t i = t0 + threadIdx.x; i < b.n; i += kSortThreads) {
                const uint32_t d = IDS ? digit_of(pass, 0, b.val[cur][i]) : digit_of(pass, b.key[cur][i], 0);
                atomicAdd(&h[wave][d], 1u);
            }
        }
        __syncthreads();
        uint32_t c = 0;
#pragma unroll
        for (int w2 = 0; w2 < kSortWaves; ++w2) c += h[w2][threadIdx.x];
        b.tile_off[(int64_t)tile * kRadix + threadIdx.x] = c;
        __syncthreads();
    }
}

__global__ __launch_bounds__(kSortThreads) void tile_count_kernel(SortBufs b, int pass) {
    if (b.ctl->skip[pass]) return;
    __shared__ uint32_t h[kSortWaves][kRadix];
    if (pass < 4) count_tiles<true>(b, pass, h);
    else count_tiles<false>(b, pass, h);
}

// ---- per pass: exclusive scan over the tiles of every digit's count, plus the digit's global base --------
// tile_off is [tile][digit]: the count kernel writes and the scatter kernel reads one contiguous 1 KB row
// per tile.  The scan walks columns: thread d owns digit d, a workgroup owns kScanRows consecutive tiles
// (every row access is a coalesced 1 KB).  Kernel A: the group's column sums.  Kernel B: every group adds up
// the sums of the groups before it (L2-resident, <= 256 KB per group) and rewrites its rows as offsets.
__global__ __launch_bounds__(kRadix) void scan_group_sums_kernel(SortBufs b, int pass) {
    if (b.ctl->skip[pass]) return;
    const int r0 = blockIdx.x * kScanRows;
    const int r1 = r0 + kScanRows < b.n_tiles ? r0 + kScanRows : b.n_tiles;
    uint32_t sum = 0;
    for (int r = r0; r < r1; ++r) sum += b.tile_off[(int64_t)r * kRadix + threadIdx.x];
    b.group_sum[(int64_t)blockIdx.x * kRadix + threadIdx.x] = sum;
}

__global__ __launch_bounds__(kRadix) void scan_offsets_kernel(SortBufs b, int pass) {
    if (b.ctl->skip[pass]) return;
    __shared__ uint32_t wsum[kRadix / kWave];
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6;
    // global base of digit d = number of elements with a smaller digit (exclusive scan of the histogram)
    const uint32_t hcount = b.hist[pass * kRadix + d];
    uint32_t incl = hcount;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - hcount;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    // + this digit's elements in the groups before this one (independent loads: keep many in flight)
#pragma unroll 16
    for (int g = 0; g < (int)blockIdx.x; ++g) run += b.group_sum[(int64_t)g * kRadix + d];
    const int r0 = blockIdx.x * kScanRows;
    const int r1 = r0 + kScanRows < b.n_tiles ? r0 + kScanRows : b.n_tiles;
    for (int r = r0; r < r1; ++r) {
        uint32_t* p = b.tile_off + (int64_t)r * kRadix + d;
        const uint32_t x = *p;
        *p = run;
        run += x;
    }
}

// ---- rank of an element among the equal digits of its wavefront's share of a tile --------------------------------
// Order inside a tile = (wave, item, lane).  Two forms, same result:
//   * ATOMIC: ONE returning LDS atomic per element, `ds_add_rtn_u32 cnt[wave][digit], 1`.  The lanes of one DS instruction
//     that hit the same address are served in ascending lane order on gfx950, so the returned pre-values ARE the stable
//     ranks.  That order is what the hardware does, not something the ISA manual promises: la::large_init_device() checks
//     it once per device with lds_atomic_order_test_kernel (thousands of address patterns, several wavefronts at once) and
//     the launchers fall back to the other form if it ever fails.
//   * match: the wavefront finds every lane's peers (equal digits) with 8 ballots, the group's first lane advances the
//     counter.  ~100 VALU instructions per element more than the atomic form: with it the scatter kernels spend 2 190
//     VALU instructions per wavefront and tile and sit at 66 % VALU utilisation (profiles/archive/r03_sort_pmc_*.json).
template <bool ATOMIC>
__device__ __forceinline__ void rank_in_wave(const uint64_t (&key)[kItems], const uint32_t (&val)[kItems],
                                             uint32_t (&loc)[kItems], uint32_t* cnt_wave, int pass, int64_t w0, int64_t n,
                                             int lane) {
    if constexpr (ATOMIC) {
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const bool valid = w0 + it * kWave + lane < n;
            const uint32_t d = digit_of(pass, key[it], val[it]);
            uint32_t r = 0;
            if (valid) r = atomicAdd(&cnt_wave[d], 1u);
            loc[it] = (d << 16) | r;
        }
    } else {
        const uint64_t lt = ((uint64_t)1 << lane) - 1;
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const bool valid = w0 + it * kWave + lane < n;
            const uint32_t d = digit_of(pass, key[it], val[it]);
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool one = (d >> bit) & 1;
                const uint64_t bal = __ballot(one);
                peers &= one ? bal : ~bal;
            }
            uint32_t old = 0;
            if (valid) old = cnt_wave[d];                        // same address for all peers: broadcast
            wave_lds_fence();
            if (valid && (peers & lt) == 0) cnt_wave[d] = old + (uint32_t)__popcll(peers);   // group leader
            wave_lds_fence();
            loc[it] = (d << 16) | (old + (uint32_t)__popcll(peers & lt));
        }
    }
}

// One-off hardware check behind the ATOMIC form above: do the lanes of a returning LDS atomic that collide on an address
// get their pre-values in lane order?  4 wavefronts at once, each on its own counters, address sets of 1 .. 256 words,
// some lanes switched off, several atomics back to back as in the real loop.  *bad != 0: they do not.
__global__ __launch_bounds__(256) void lds_atomic_order_test_kernel(uint32_t* bad) {
    __shared__ uint32_t c[4][kRadix];              // counters the atomics advance
    __shared__ uint32_t shadow[4][kRadix];         // the same counters advanced by the match form: what a stable rank is
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    uint32_t wrong = 0;
    for (uint32_t pattern = 0; pattern < 384; ++pattern) {
        for (int i = lane; i < kRadix; i += kWave) { c[wave][i] = 0; shadow[wave][i] = 0; }
        wave_lds_fence();
        const uint32_t span = 1u << ((pattern + wave) % 9);                       // 1, 2, 4 .. 256 distinct addresses
        uint32_t got[4], want[4], addr[4];
        bool on[4];
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            uint32_t h = (uint32_t)lane * 2654435761u + pattern * 40503u + (uint32_t)rep * 97u + (uint32_t)wave * 7919u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            addr[rep] = h & (span - 1);
            on[rep] = ((h >> 20) & 7u) != 0;
        }
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            uint64_t peers = __ballot(on[rep]);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool one = (addr[rep] >> bit) & 1;
                const uint64_t bal = __ballot(one);
                peers &= one ? bal : ~bal;
            }
            uint32_t old = 0;
            if (on[rep]) old = shadow[wave][addr[rep]];
            wave_lds_fence();
            if (on[rep] && (peers & lt) == 0) shadow[wave][addr[rep]] = old + (uint32_t)__popcll(peers);
            wave_lds_fence();
            want[rep] = old + (uint32_t)__popcll(peers & lt);
        }
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {                  // back to back, as in rank_in_wave
            got[rep] = 0;
            if (on[rep]) got[rep] = atomicAdd(&c[wave][addr[rep]], 1u);
        }
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) wrong |= (on[rep] && got[rep] != want[rep]) ? 1u : 0u;
        wave_lds_fence();
    }
    if (wrong) atomicOr(bad, 1u);
}

// ---- per pass: stable scatter ----------------------------------------------------------------------
// Order inside a tile = (wave, item, lane); elements are loaded wave-striped so every load is a
// 64-element contiguous run.  Rank among equal digits: wave-level match (8 ballots), running
// per-wave digit counters in LDS, then an exclusive scan of those counters over the waves (stable; no
// atomics on the data path).  The tile is then REORDERED IN LDS by digit, so that consecutive lanes write
// consecutive addresses of a digit's output run (a tile holds ~16 elements per digit: 128 B runs of keys,
// 64 B of ids) instead of 64 unrelated 8-byte writes per wavefront.
template <bool ATOMIC_RANK>
__global__ __launch_bounds__(kSortThreads) void tile_scatter_kernel(SortBufs b, int pass) {
    if (b.ctl->skip[pass]) return;
    static_assert(kSortThreads == kRadix, "one thread per digit in the offset phase");
    __shared__ uint32_t cnt[kSortWaves][kRadix];
    __shared__ uint32_t bin_start[kRadix];            // tile-local position of the digit's first element
    __shared__ uint32_t bin_base[kRadix];             // global position of it, minus bin_start
    __shared__ uint32_t wsum[kSortWaves];
    __shared__ uint64_t s_stage[kTile];               // the tile reordered by digit: one array at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kSortWaves * kRadix; i += kSortThreads) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t cur = b.ctl->cur[pass];
    const uint64_t* kin = b.key[cur];
    const uint32_t* vin = b.val[cur];
    uint64_t* kout = b.key[cur ^ 1];
    uint32_t* vout = b.val[cur ^ 1];
    // Workgroup w runs on XCD w % 8 (observed dispatch order; only speed depends on it).  Neighbouring tiles
    // write neighbouring runs of every digit's output: give each XCD a CONTIGUOUS range of tiles so that the
    // partial cache lines at the run boundaries meet in one L2 instead of being written back twice.
    const int tile = xcd_contiguous(blockIdx.x, gridDim.x);
    const int64_t t0 = (int64_t)tile * kTile;
    const int64_t w0 = t0 + (int64_t)wave * kItems * kWave;
    const int n_here = (int)((b.n - t0) < kTile ? (b.n - t0) : kTile);

    uint64_t key[kItems];
    uint32_t val[kItems];
    uint32_t loc[kItems];       // digit << 16 | rank inside (wave, digit)   (rank < 1024)
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        const int64_t i = w0 + it * kWave + lane;
        const bool valid = i < b.n;
        key[it] = valid ? stream_load(kin + i) : 0;
        val[it] = valid ? stream_load(vin + i) : 0;
    }
    rank_in_wave<ATOMIC_RANK>(key, val, loc, cnt[wave], pass, w0, b.n, lane);
    __syncthreads();
    {   // thread d: exclusive scan of digit d's counters over the waves, then of the tile's digit counts
        const int d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            const uint32_t c = cnt[w][d];
            cnt[w][d] = run;
            run += c;
        }
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const uint32_t start = woff + incl - run;
        bin_start[d] = start;
        bin_base[d] = b.tile_off[(int64_t)tile * kRadix + d] - start;
    }
    __syncthreads();
    // tile-local sorted position of every element
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        const uint32_t d = loc[it] >> 16;
        loc[it] = bin_start[d] + cnt[wave][d] + (loc[it] & 0xFFFFu);
    }
    // The array that carries the pass's digit goes through the staging buffer first: the global position of
    // tile-local position j is bin_base[digit of the element at j] + j.
    uint32_t* s_val = reinterpret_cast<uint32_t*>(s_stage);
    uint32_t gpos[kItems];
    if (pass < 4) {
#pragma unroll
        for (int it = 0; it < kItems; ++it)
            if (w0 + it * kWave + lane < b.n) s_val[loc[it]] = val[it];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int j = it * kSortThreads + threadIdx.x;
            if (j < n_here) {
                const uint32_t v = s_val[j];
                gpos[it] = bin_base[digit_of(pass, 0, v)] + (uint32_t)j;
                vout[gpos[it]] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kItems; ++it)
            if (w0 + it * kWave + lane < b.n) s_stage[loc[it]] = key[it];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int j = it * kSortThreads + threadIdx.x;
            if (j < n_here) kout[gpos[it]] = s_stage[j];
        }
    } else {
#pragma unroll
        for (int it = 0; it < kItems; ++it)
            if (w0 + it * kWave + lane < b.n) s_stage[loc[it]] = key[it];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int j = it * kSortThreads + threadIdx.x;
            if (j < n_here) {
                const uint64_t k = s_stage[j];
                gpos[it] = bin_base[digit_of(pass, k, 0)] + (uint32_t)j;
                kout[gpos[it]] = k;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kItems; ++it)
            if (w0 + it * kWave + lane < b.n) s_val[loc[it]] = val[it];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int j = it * kSortThreads + threadIdx.x;
            if (j < n_here) vout[gpos[it]] = s_val[j];
        }
    }
}

// ---- per pass, single-kernel form: stable scatter with decoupled look-back ----------------------------------------
// One launch per pass instead of four (count, two scans, scatter): the count kernel's re-read of the digit array
// (4-8 B per element) and the offset matrix are gone.  The global digit histograms come from build_keys_kernel
// (b.gbase = their exclusive scans, plan_kernel); what a tile still needs is how many elements of each digit the tiles
// BEFORE it hold.  Every tile publishes its 256 digit counts as granules {tag, count} -- ONE naturally aligned 8-byte
// relaxed agent-scope store each, the data is its own flag (cdna_hip_programming.md section 6, Guideline 16, form R2:
// the per-XCD L2s are not coherent, a plain store/load pair would go stale) -- first as an AGGREGATE (its own count),
// then, after walking back over its predecessors until it meets an INCLUSIVE one, as the inclusive prefix.  tag =
// (pass + 1) << 2 | status, so the state array is zeroed once per sort, not per pass.
//
// Which tile a workgroup takes is decided by an arrival ticket, never by blockIdx: tile = arrival number, so a tile only
// ever waits for tiles whose workgroups are already running, and the walk cannot deadlock whatever the dispatch order is
// (a bounded spin turns anything unforeseen into kStatusInternal instead of a hung device).  Tried and measured without
// effect on this kernel (profiles/archive/r03_sort_*): giving every XCD 4-32 consecutive tiles at a time (arrival- or
// blockIdx-based) -- what made the four-kernel scatter 20 % faster -- and polling 4, 8 or 16 predecessors per hop.
#ifndef LA_LOOK_WINDOW
#define LA_LOOK_WINDOW 2
#endif
#ifdef LA_LOOKBACK_STATS   // development build: how far the walks go (tools/lookback_probe.py)
__device__ unsigned long long g_lookback_stats[8];      // walks, hops, polls that found nothing, longest walk, cycles in walks
#endif
#ifdef LA_SWEEP_CLOCKS     // development build: thread 0 of every tile adds the time of each phase (tools/lookback_probe.py)
__device__ unsigned long long g_sweep_clocks[12];
#define LA_SCLK(i)                                                                      \
    do {                                                                                \
        if (threadIdx.x == 0) {                                                         \
            const unsigned long long now_ = wall_clock64();                             \
            atomicAdd(&g_sweep_clocks[i], now_ - sclk_);                                \
            sclk_ = now_;                                                               \
        }                                                                               \
    } while (0)
#define LA_SCLK_START unsigned long long sclk_ = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&g_sweep_clocks[11], 1ull)
#else
#define LA_SCLK(i) do {} while (0)
#define LA_SCLK_START do {} while (0)
#endif
constexpr uint32_t kStateAggregate = 1u, kStateInclusive = 2u;
constexpr int kLookWindow = LA_LOOK_WINDOW;           // predecessors a walk polls at once
constexpr uint32_t kLookbackSpinLimit = 1u << 22;     // polls of one granule (~1 us each with the sleep) before giving up

// One tile of one pass: the body of onesweep_pass_kernel (one tile per workgroup, taken by arrival) and of onesweep_redo_kernel
// (round 6: workgroups that loop over tickets and passes).  s_ticket: the tile's number, written by thread 0 of the caller BEFORE
// the call and read here after the first barrier.
template <bool ATOMIC_RANK, int THREADS>
__device__ __forceinline__ void onesweep_tile(const SortBufs& b, const int slot, uint32_t* status, const uint32_t* s_ticket) {
    const int pass = slot >= kDigits ? slot - kDigits : slot;   // the digit the slot sorts by (the redo slots repeat the digits)
    static_assert(THREADS >= kRadix && THREADS % kWave == 0, "one thread per digit in the look-back");
    constexpr int WAVES = THREADS / kWave, TILE = THREADS * kItems;
    __shared__ uint32_t cnt[WAVES][kRadix];
    __shared__ uint32_t bin_start[kRadix];            // tile-local position of the digit's first element
    __shared__ uint32_t bin_base[kRadix];             // global position of it, minus bin_start
    __shared__ uint32_t wsum[kRadix / kWave];
    __shared__ uint64_t s_stage[TILE];                // the tile reordered by digit: one array at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool digit_thread = threadIdx.x < kRadix;   // (whole wavefronts: 0 .. 3)
    LA_SCLK_START;
    for (int i = threadIdx.x; i < WAVES * kRadix; i += THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t cur = b.ctl->cur[slot];
    const uint64_t* kin = key_buf(b, cur);
    const uint32_t* vin = val_buf(b, cur);
    uint64_t* kout = key_buf(b, cur ^ 1u);
    uint32_t* vout = val_buf(b, cur ^ 1u);
    const int tile = (int)*s_ticket;
    const int64_t t0 = (int64_t)tile * TILE;
    const int64_t w0 = t0 + (int64_t)wave * kItems * kWave;
    const int n_here = (int)((b.n - t0) < TILE ? (b.n - t0) : TILE);
    const uint32_t epoch = (uint32_t)(slot + 1) << 2;
    unsigned long long* my_state = b.tile_state + (int64_t)tile * kRadix + (threadIdx.x & (kRadix - 1));

    uint64_t key[kItems];
    uint32_t val[kItems];
    uint32_t loc[kItems];       // digit << 16 | rank inside (wave, digit)   (rank < 1024)
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        const int64_t i = w0 + it * kWave + lane;
        const bool valid = i < b.n;
        key[it] = valid ? stream_load(kin + i) : 0;
        val[it] = valid ? stream_load(vin + i) : 0;
    }
    LA_SCLK(0);                                       // ticket, counters, loads issued
    rank_in_wave<ATOMIC_RANK>(key, val, loc, cnt[wave], pass, w0, b.n, lane);
    LA_SCLK(1);                                       // loads landed + ranks (this wavefront)
    __syncthreads();
    LA_SCLK(2);                                       // ... of the slowest wavefront
    // thread d: digit d's count in this tile; published at once, so that later tiles never wait for this tile's walk
    uint32_t count = 0, incl = 0;
    unsigned long long win[kLookWindow];
    if (digit_thread) {
        const int d = threadIdx.x;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = cnt[w][d];
            cnt[w][d] = count;
            count += c;
        }
        __hip_atomic_store(my_state, ((unsigned long long)(epoch | (tile == 0 ? kStateInclusive : kStateAggregate)) << 32) | count,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the first window of the walk is in flight while the tile is staged below
#pragma unroll
        for (int i = 0; i < kLookWindow; ++i)
            win[i] = i < tile ? __hip_atomic_load(my_state - (i + 1) * kRadix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        incl = count;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
    }
    __syncthreads();
    if (digit_thread) {
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        bin_start[threadIdx.x] = woff + incl - count;
    }
    __syncthreads();
    // tile-local sorted position of every element; the array that carries the pass's digit goes to the staging buffer
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        const uint32_t d = loc[it] >> 16;
        loc[it] = bin_start[d] + cnt[wave][d] + (loc[it] & 0xFFFFu);
    }
    uint32_t* s_val = reinterpret_cast<uint32_t*>(s_stage);
    if (pass < 4) {
#pragma unroll
        for (int it = 0; it < kItems; ++it)
            if (w0 + it * kWave + lane < b.n) s_val[loc[it]] = val[it];
    } else {
#pragma unroll
        for (int it = 0; it < kItems; ++it)
            if (w0 + it * kWave + lane < b.n) s_stage[loc[it]] = key[it];
    }
    LA_SCLK(3);                                       // counts published, bin starts, first array staged
    if (digit_thread)
    {   // The walk: thread d adds up digit d's counts of the tiles before this one, nearest first, until it meets an
        // inclusive prefix.  A hop costs a trip to the fabric (the granules are write-through, the per-XCD L2s are not
        // coherent) behind this CU's own streaming loads and stores -- about 1 us -- and a walk passes ~28 tiles that have
        // published only their aggregate (measured: tools/lookback_probe.py), so kLookWindow predecessors are polled at once.
        const int d = threadIdx.x;
        uint32_t excl = 0;
        if (tile > 0) {
            int next = tile - 1;                                 // the nearest tile not yet added
            uint32_t spins = 0;
#ifdef LA_LOOKBACK_STATS
            unsigned long long hops = 0, empty = 0;
            const unsigned long long clk0 = clock64();
#endif
            for (;;) {
                bool done = false;
                int used = 0;
#pragma unroll
                for (int i = 0; i < kLookWindow; ++i) {
                    const uint32_t tag = (uint32_t)(win[i] >> 32);
                    const bool ready = (tag & ~3u) == epoch && i < next + 1 && used == i && !done;
                    if (ready) {
                        excl += (uint32_t)win[i];
                        used = i + 1;
                        done = (tag & 3u) == kStateInclusive;
                    }
                }
#ifdef LA_LOOKBACK_STATS
                hops += (unsigned long long)used;
                empty += used == 0 ? 1 : 0;
#endif
                if (done) break;
                next -= used;                                    // tile 0 is always inclusive: next never drops below 0 here
                if (used == 0) {
                    if (++spins > kLookbackSpinLimit) { atomicOr(status, kStatusInternal); break; }
                    __builtin_amdgcn_s_sleep(2);
                } else {
                    spins = 0;
                }
                const unsigned long long* look = b.tile_state + (int64_t)next * kRadix + d;
#pragma unroll
                for (int i = 0; i < kLookWindow; ++i)
                    win[i] = i <= next ? __hip_atomic_load(look - i * kRadix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
#ifdef LA_LOOKBACK_STATS
            if (d == 0) {
                atomicAdd(&g_lookback_stats[0], 1ull);
                atomicAdd(&g_lookback_stats[1], hops);
                atomicAdd(&g_lookback_stats[2], empty);
                atomicMax(&g_lookback_stats[3], hops);
                atomicAdd(&g_lookback_stats[4], (unsigned long long)(clock64() - clk0));
            }
#endif
            __hip_atomic_store(my_state, ((unsigned long long)(epoch | kStateInclusive) << 32) | (excl + count),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bin_base[d] = b.gbase[pass * kRadix + d] + excl - bin_start[d];
    }
    LA_SCLK(4);                                       // this wavefront's walk
    __syncthreads();
    LA_SCLK(5);                                       // ... the slowest walk
    // the global position of tile-local position j is bin_base[digit of the element at j] + j
    uint32_t gpos[kItems];
    if (pass < 4) {
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int j = it * THREADS + threadIdx.x;
            if (j < n_here) {
                const uint32_t v = s_val[j];
                gpos[it] = bin_base[digit_of(pass, 0, v)] + (uint32_t)j;
                vout[gpos[it]] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kItems; ++it)
            if (w0 + it * kWave + lane < b.n) s_stage[loc[it]] = key[it];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int j = it * THREADS + threadIdx.x;
            if (j < n_here) kout[gpos[it]] = s_stage[j];
        }
    } else {
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int j = it * THREADS + threadIdx.x;
            if (j < n_here) {
                const uint64_t k = s_stage[j];
                gpos[it] = bin_base[digit_of(pass, k, 0)] + (uint32_t)j;
                kout[gpos[it]] = k;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kItems; ++it)
            if (w0 + it * kWave + lane < b.n) s_val[loc[it]] = val[it];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int j = it * THREADS + threadIdx.x;
            if (j < n_here) vout[gpos[it]] = s_val[j];
        }
    }
    LA_SCLK(6);                                       // both scatters issued (stores still in flight)
}

template <bool ATOMIC_RANK, int THREADS>
__global__ __launch_bounds__(THREADS, 4) void onesweep_pass_kernel(SortBufs b0, int pass, uint32_t* status, const LargeItem* items,
                                                                   char* scratch) {
    LargeArgs unused{};
    SortBufs b = b0;
    if (items) bind_item(unused, b, items[blockIdx.y], scratch);
    // several topics per launch: the grid is as wide as the launch's largest topic, and only workgroups that have a tile to
    // sort may draw a ticket (tiles are taken by arrival, so exactly n_tiles workgroups of a topic must arrive)
    if ((int)blockIdx.x >= b.n_tiles) return;
    const int slot = pass;                            // control-block slot (skip / cur / ticket / tag of the granules)
    if (b.ctl->skip[slot]) return;
    __shared__ uint32_t s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&b.ticket[slot], 1u);
    onesweep_tile<ATOMIC_RANK, THREADS>(b, slot, status, &s_ticket);
}

// ---- the redo slots of a keys-first sort as ONE launch (round 6) ---------------------------------------------------------------
// A keys-first sort whose tie repair met a run of equal lags longer than a workgroup holds (ctl->redo: the sample said there
// was none, or a test forced keys first) is sorted again in full, ids first.  Rounds 4-5 kept a second set of pass launches
// for that -- up to twelve kernels that return at once in every ordinary sort, 4.4 us apiece: 50 us of a 1.15 ms sort of 33.5 M
// partitions, a fifth of the sort of a 5 M-partition topic.  Now: ONE kernel whose workgroups return at once unless ctl->redo
// is set, and otherwise run all the redo passes themselves -- tiles by ticket as in the pass kernel (a workgroup that holds
// ticket t walks back over tiles that running workgroups hold: the look-back cannot wait for a workgroup that has not
// started), an agent-scope barrier over the launch's workgroups between passes.  The grid is small enough to be resident at
// once (a workgroup keeps its CU until the last pass is done).
template <bool ATOMIC_RANK, int THREADS>
__global__ __launch_bounds__(THREADS, 4) void onesweep_redo_kernel(SortBufs b0, uint32_t pass_mask, uint32_t* status, const LargeItem* items,
                                                                   char* scratch) {
    LargeArgs unused{};
    SortBufs b = b0;
    if (items) bind_item(unused, b, items[blockIdx.y], scratch);
    if (!b.ctl->keys_first || !b.ctl->redo) return;   // (every workgroup of the topic reads the same words: replan_kernel ran before)
    __shared__ uint32_t s_ticket;
    uint32_t passes_done = 0;
    for (int d = 0; d < kDigits; ++d) {
        const int slot = kDigits + d;
        if (!((pass_mask >> d) & 1u) || b.ctl->skip[slot]) continue;          // (uniform over the topic's workgroups)
        for (;;) {
            if (threadIdx.x == 0) s_ticket = atomicAdd(&b.ticket[slot], 1u);
            __syncthreads();
            const bool more = (int)s_ticket < b.n_tiles;
            if (!more) break;                          // (uniform: one shared word; nobody writes it before the barrier below)
            onesweep_tile<ATOMIC_RANK, THREADS>(b, slot, status, &s_ticket);
            __syncthreads();                           // the tile's last reads of the staging buffer and of s_ticket
        }
        // every workgroup's stores of this pass, visible to every other before the next pass reads them
        ++passes_done;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&b.ctl->redo_arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t want = passes_done * gridDim.x;
            while (__hip_atomic_load(&b.ctl->redo_arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    }
}

// ---- outputs that do not depend on the greedy ----------------------------------------------------
// It also CHECKS the sort: neighbours must be in (key, id) order.  Every LSD pass has to be stable, and the default ranking of a pass
// (one returning LDS atomic per element) is stable only because colliding lanes are served in lane order -- a property of
// the hardware checked once at la_create on an idle device, not a promise of the ISA.  A violation under load would scramble
// the order silently; here it raises kStatusOrder (LA_EHIP at the next sync) for 8 B more read per element.
// ---- what the rounds kernel reads and writes per round (round 6) ---------------------------------------------------------------
// greedy_rounds_kernel is ONE workgroup on one compute unit, and every round of a topic of C consumers used to pull C sorted 64-bit
// keys in and push C 32-bit consumer indices out through that unit's address path -- 96 KB per round at 8 192 consumers, in 16-byte
// pieces that each touch every cache line of a wavefront's stretch: measured with the stores left out and with half the loads
// (profiles/r06_am_rounds_memory.txt), 1.65 us of a 3.4 us round in which nothing else happens, ~0.2 ms of cfg5's 0.83.  The
// NARROW form: where the topic has more than 4 096 consumers, its bins pack and no lag needs more than 32 bits, emit_ids_kernel leaves the lags as 32-bit words in
// the sort's other key buffer and the rounds kernel leaves its results as 16-bit consumer indices in the sort's other value
// buffer -- both with every round's stretch starting on a multiple of 8 elements (stride `cpad`), so a thread's 8 lags are two
// aligned 16-byte loads, its 8 results ONE 16-byte store, and a wavefront's accesses are contiguous; map_ranks_kernel, which
// turns indices into member ranks on the whole chip anyway, reads the 16-bit form.  emit_ids_kernel, greedy_rounds_kernel and
// map_ranks_kernel each decide from the same sorted keys (the first carries the largest lag, the last the smallest).
__host__ __device__ inline void rounds_class(int64_t n_cons, int* ec, int* threads);
struct RoundsIo {
    int idx_bits;          // the bins pack into (total << idx_bits) | index; 0: they do not (96-bit bins)
    bool narrow;           // 32-bit lags in, 16-bit indices out, rounds `cpad` elements apart
    int64_t cpad;          // C rounded up to a multiple of 8
    bool zero_tail;        // the smallest lag is zero: the topic's last rounds may hand out zeros only (greedy_rounds_packed)
};
// only_narrow: the caller wants `narrow` alone (emit_ids_kernel, map_ranks_kernel) -- decided without a look at the keys where it can be
__device__ __forceinline__ RoundsIo rounds_io(const LargeArgs& a, const uint64_t* key, int64_t P, int64_t C, const bool only_narrow = false) {
    RoundsIo io{0, false, (C + 7) & ~(int64_t)7, false};
    if (P <= 0 || C <= 0) return io;
    int ec = 1, threads = 64;
    rounds_class(C, &ec, &threads);
    if (only_narrow && !(LA_ROUNDS_NARROW_IO != 0 && a.rounds_follow != 0 && ec >= LA_ROUNDS_NARROW_EC && P >= 8)) return io;   // (no key is read)
    const int n = ec * threads;
    const int64_t rounds = (P + C - 1) / C;
    const int64_t lmax = (int64_t)(key[0] ^ kLagKeyFlip), lmin = (int64_t)(key[P - 1] ^ kLagKeyFlip);
    const int idx_bits = 31 - __builtin_clz(n);                                  // n is a power of two
    const int lag_bits = lmax > 0 ? 64 - __builtin_clzll((unsigned long long)lmax) : 0;
    const int round_bits = 64 - __builtin_clzll((unsigned long long)rounds);
    if (lmin >= 0 && lag_bits + round_bits + idx_bits <= 62) io.idx_bits = idx_bits;
    io.zero_tail = lmin == 0;
    // (the last element written is P - 1 + 7 (rounds - 1) < 2 P: both buffers hold twice as many narrow elements as partitions)
    io.narrow = LA_ROUNDS_NARROW_IO != 0 && a.rounds_follow != 0 && io.idx_bits != 0 && ec >= LA_ROUNDS_NARROW_EC && lag_bits <= 32 && P >= 8;
    return io;
}

__global__ __launch_bounds__(256) void emit_ids_kernel(LargeArgs a0, SortBufs b0, const LargeItem* items, char* scratch) {
    LA_PICK_ITEM(a, b, a0, b0, items, scratch, blockIdx.y)
    if ((int64_t)blockIdx.x * blockDim.x >= b.n) return;
    const bool fill_rank_minus1 = a.n_cons == 0;                       // nobody to assign to: Main.java:211-214
    const uint32_t fin = b.ctl->cur[kFinal];
    const uint32_t* val = val_buf(b, fin);
    const uint64_t* key = key_buf(b, fin);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const RoundsIo io = rounds_io(a, key, b.n, a.n_cons, true);
    uint32_t* lag32 = reinterpret_cast<uint32_t*>(key_buf(b, fin ^ 1u));          // (narrow form: the rounds' lags)
    const uint32_t C32 = (uint32_t)a.n_cons;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < b.n; i += stride) {
        const uint32_t v = val[i];
        const uint64_t k = key[i];
        if (i > 0) {
            const uint64_t kp = key[i - 1];
            bad |= kp > k || (kp == k && val[i - 1] > v);
        }
        if (io.narrow) {
            const uint32_t q = (uint32_t)i / C32;                                // (b.n < 2^31)
            lag32[(int64_t)q * io.cpad + ((uint32_t)i - q * C32)] = (uint32_t)(k ^ kLagKeyFlip);
        }
        a.out_pid[a.p0 + i] = (int32_t)(v ^ kPidBias);
        if (fill_rank_minus1) a.out_rank[a.p0 + i] = -1;
    }
    if (__any(bad) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, kStatusOrder);
}

// ---- kernel 3, packed form: bins are one 64-bit word -----------------------------------------------------
// (total << idx_bits) | index, ascending = (total, member) ascending, the comparator of Main.java:253-259.
// Element i = tid*EC + r.  Strides inside a wavefront run through the instruction-level networks of
// la_sort64.h (registers, DPP); strides across wavefronts go through LDS: 10 exchanges per round at 8 192 bins.
template <int EC>
__device__ __forceinline__ void dpp_fence(P64 (&rec)[EC]) {
    // the next instruction-level block reads these registers through DPP: 2 wait states after their last
    // (compiler-generated) write
#pragma unroll
    for (int r = 0; r < EC; ++r) asm volatile("s_nop 1" : "+v"(rec[r].lo), "+v"(rec[r].hi));
}

// LDS layout is register-major (slot r of thread t at r * blockDim + t): consecutive lanes touch consecutive
// 8-byte words, where element order (t * EC + r) would put a 64-byte stride between lanes (16-way conflicts).
// One barrier per step: consecutive steps ALTERNATE between two exchange buffers (ExchangeBufs::next), so a thread
// that runs ahead writes the other buffer while slower threads still read this one; by the time a buffer comes
// round again every thread has passed the barrier of the step in between, i.e. has finished reading it.
// (Base + offset, not an array of two pointers: indexed dynamically the array lands in scratch memory, the pointers
// come back from there as generic ones, and every LDS access of the exchange turns into a flat_load / flat_store.)
struct ExchangeBufs {
    uint64_t* base;
    int stride;                                     // words between the two buffers
    int at = 0;
    __device__ __forceinline__ uint64_t* next() { at ^= 1; return base + (at ? stride : 0); }
};

template <int EC>
__device__ __forceinline__ void cross_wave_step(P64 (&rec)[EC], ExchangeBufs& xb, int tid, int mask, int min_bit) {
    const int nt = blockDim.x;
    uint64_t* s_bin = xb.next();
#pragma unroll
    for (int r = 0; r < EC; ++r) s_bin[r * nt + tid] = p64_value(rec[r]);
    lds_barrier();
#pragma unroll
    for (int r = 0; r < EC; ++r) {
        const int i = tid * EC + r;
        const int pi = i ^ mask;                                     // partner element: thread pi / EC, register pi % EC
        const uint64_t o = s_bin[(pi % EC) * nt + pi / EC], x = p64_value(rec[r]);
        const bool keep_min = (i & min_bit) == 0;
        const uint64_t lo = o < x ? o : x, hi = o < x ? x : o;
        rec[r] = p64_from(keep_min ? lo : hi);
    }
}

#ifdef LA_ROUND_CLOCKS   // development build: thread 0 accumulates the cycles of every phase of a round (tools/cfg5_probe.py)
__device__ unsigned long long g_round_clocks[16];
__device__ unsigned long long g_round_stamp[520];        // [q] clock at the start of round q; [260 + q] path << 32 | bins that moved
#define LA_STAMP(q, path, m)                                                                           \
    do {                                                                                               \
        if (threadIdx.x == 0 && (q) < 259) {                                                           \
            g_round_stamp[q] = clock64();                                                              \
            g_round_stamp[260 + (q)] = ((unsigned long long)(path) << 32) | (unsigned)(m);             \
        }                                                                                              \
    } while (0)
#define LA_CLK(i)                                                          \
    do {                                                                   \
        if (threadIdx.x == 0) {                                            \
            const unsigned long long now_ = clock64();                     \
            g_round_clocks[i] += now_ - clk_;                              \
            clk_ = now_;                                                   \
        }                                                                  \
    } while (0)
#define LA_CLK_START unsigned long long clk_ = clock64()
#define LA_CLK_PARAM , unsigned long long& clk_          // a helper that times its phases on the caller's clock
#define LA_CLK_ARG , clk_
#else
#define LA_CLK_PARAM
#define LA_CLK_ARG
#define LA_CLK(i) do {} while (0)
#define LA_CLK_START do {} while (0)
#define LA_STAMP(q, path, m) do {} while (0)
#endif

constexpr int kSampleThreads = 1024;
constexpr uint32_t kMaxBucket = 96;
constexpr uint32_t kMaxBucket2 = 192;                        // second level: samples per super-sample bucket (~16)
// The rank walks read a bucket four staged bins at a time and may run past its end -- past the end of the staged array for
// the last bucket: that many sentinels (all ones: larger than any bin) close the array, so the loops need no clamp.
constexpr int kWalkPad = (int)kMaxBucket + 4, kWalkPad2 = (int)kMaxBucket2 + 4;

constexpr int kSamplesPerThread = 1;                         // 1 024 splitters: buckets of ~EC bins (2 per thread: the
                                                             // walk halves, but the sample sort's chain of steps grows more)

struct SampleLds {
    ulonglong2* stage;    // [n]  staged bins in bucket order: (bin, bucket size << 16 | slot); the final order
                          //      (uint64 [n]) is written over it once every staged bin is back in a register
    uint64_t* spl;        // [2 * NS] splitters in the first half; before that the two exchange buffers of the sample
                          //      sort's LDS steps (nothing else may be in flight there: the staging area is still being read
                          //      by slower threads when the fastest start the next round)
    uint32_t* cnt;        // [NS + 1] bucket counts, then (first position | size << 16)
    uint32_t* misc;       // [32] per-wavefront sums and maxima; [32 .. 32 + 66) second-level bucket counts, flag
    uint32_t* moved;      // [kMovedWords] the moved-bins sort's own words (moved_sort_bins): nothing else touches them
};
constexpr int kMovedWords = 352;

__host__ __device__ constexpr size_t sample_lds_bytes(int ec) {
    return ((size_t)ec * kSampleThreads + kWalkPad) * 16 + (size_t)kSamplesPerThread * kSampleThreads * (16 + 4) + 64 + (32 + 128) * 4 +
           (size_t)kMovedWords * 4;
}

__device__ __forceinline__ SampleLds sample_lds_carve(void* smem, int n) {
    SampleLds L;
    L.stage = reinterpret_cast<ulonglong2*>(smem);
    L.spl = reinterpret_cast<uint64_t*>(L.stage + n + kWalkPad);   // stage[n ..): sentinels, larger than any bin
    L.cnt = reinterpret_cast<uint32_t*>(L.spl + 2 * kSamplesPerThread * kSampleThreads);
    L.misc = L.cnt + kSamplesPerThread * kSampleThreads + 16;
    L.moved = L.misc + 32 + 128;
    return L;
}

// max over the wavefront, in every lane, without the LDS pipe (DPP + v_permlane*_swap butterflies)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, shfl_xor<1>(v));
    v = max(v, shfl_xor<2>(v));
    v = max(v, shfl_xor<4>(v));
    v = max(v, shfl_xor<8>(v));
    v = max(v, shfl_xor<16>(v));
    v = max(v, shfl_xor<32>(v));
    return v;
}

template <int EC>
__device__ __forceinline__ bool sample_sort_bins(P64 (&rec)[EC], const SampleLds& L, int tid, uint32_t bucket_limit) {
    constexpr int NT = kSampleThreads, N = EC * NT;
    constexpr int SPT = (EC >= 2 * kSamplesPerThread) ? kSamplesPerThread : 1;   // samples per thread
    constexpr int NS = SPT * NT;                                                   // splitters; NS + 1 buckets
    const int lane = tid & 63, wave = tid >> 6;
    LA_CLK_START;
    // 1. splitters: regular samples of last round's order, one bin per thread, SORTED -- by the same scheme one level
    //    down: 64 of the 1 024 samples are sorted by one wavefront (one per lane, DPP only), every sample finds its
    //    bucket among them (6 LDS reads), takes a slot, is staged, and is ranked by the walk over its bucket (~16
    //    samples).  6 barriers and ~50 LDS operations per thread, where the block-wide network over the samples
    //    needs 45 DPP steps, 10 LDS exchanges and 11 barriers -- a chain of latencies, 14k cycles of a 53k-cycle round.
    //    A second-level bucket above kMaxBucket2 sends the samples through the network instead.
    static_assert(SPT == 1, "one sample per thread");
    bool samples_sorted = false;
    {
        constexpr int S2 = 64;                                    // super-samples = lanes of the sorting wavefront
        uint64_t* smpbuf = L.spl + NS;                            // not the staging area: slower threads may still be
        uint32_t* cnt2 = L.misc + 32;                             // reading last round's final order out of it
        const uint64_t mine = p64_value(rec[EC / 2]);
        smpbuf[tid] = mine;
        L.cnt[tid] = 0;
        if (tid == 0) L.cnt[NS] = 0;
        if (tid < S2 + 2) cnt2[tid] = 0;
        lds_barrier();                                          // (1) from here on the staging area is free
        uint64_t* sup = reinterpret_cast<uint64_t*>(L.stage);     // [64] sorted super-samples
        ulonglong2* stage2 = L.stage + 64;                        // [NS + 2] staged samples
        if (wave == 0) {
            P64 v = p64_from(smpbuf[lane * (NS / S2) + NS / (2 * S2)]);
            asm volatile("s_nop 1" : "+v"(v.lo), "+v"(v.hi));
            bitonic_sort_lanes_p64<64>(v);
            sup[lane] = p64_value(v);
        } else if (tid - 64 < kWalkPad2) {
            stage2[NS + tid - 64] = make_ulonglong2(~0ull, 0);    // sentinels of the walk below
        }
        lds_barrier();                                          // (2)
        uint32_t b2 = 0;
#pragma unroll
        for (int step = S2 / 2; step >= 1; step >>= 1) b2 += (sup[b2 + step - 1] < mine) ? (uint32_t)step : 0u;
        b2 += (sup[S2 - 1] < mine) ? 1u : 0u;
        const uint32_t slot2 = atomicAdd(&cnt2[b2], 1u);
        lds_barrier();                                          // (3)
        if (wave == 0) {                                          // first positions of the 65 buckets; the largest one
            const uint32_t c = cnt2[lane];
            const uint32_t incl = wave_incl_scan_u32(c);
            const uint32_t last = cnt2[S2];
            const uint32_t mx = wave_max_u32(max(c, last));
            __builtin_amdgcn_wave_barrier();
            cnt2[lane] = (incl - c) | (c << 16);
            if (lane == 63) { cnt2[S2] = incl | (last << 16); cnt2[S2 + 1] = mx; }
        }
        lds_barrier();                                          // (4)
        // (the test hook that tightens the first level tightens this one too, so that all three forms of a round mix)
        if (cnt2[S2 + 1] <= (bucket_limit < kMaxBucket ? 24u : kMaxBucket2)) {      // workgroup-uniform
            const uint32_t sc = cnt2[b2];
            stage2[(sc & 0xFFFFu) + slot2] = make_ulonglong2(mine, (uint64_t)((sc & 0xFFFF0000u) | slot2));
            lds_barrier();                                      // (5)
            const ulonglong2 e = stage2[tid];
            const uint32_t inf = (uint32_t)e.y;
            const uint32_t s0 = (uint32_t)tid - (inf & 0xFFFFu);
            const uint32_t cmax = wave_max_u32(inf >> 16);
            const uint64_t* keys2 = reinterpret_cast<const uint64_t*>(stage2 + s0);      // the bucket's first staged sample
            uint32_t below = 0;
#pragma unroll 2
            for (uint32_t k = 0; k < cmax; k += 4) {
                const uint64_t o0 = keys2[2 * k], o1 = keys2[2 * k + 2], o2 = keys2[2 * k + 4], o3 = keys2[2 * k + 6];
                below += (o0 < e.x ? 1u : 0u) + (o1 < e.x ? 1u : 0u) + (o2 < e.x ? 1u : 0u) + (o3 < e.x ? 1u : 0u);
            }
            L.spl[s0 + below] = e.x;
            samples_sorted = true;
        }
    }
    if (!samples_sorted) {
        // the block-wide network over the samples (one per thread)
        lds_barrier();
        P64 smp[SPT];
#pragma unroll
        for (int u = 0; u < SPT; ++u) smp[u] = rec[(2 * u + 1) * EC / (2 * SPT)];
        dpp_fence<SPT>(smp);
        bitonic_sort_tile_p64<64, SPT>(smp);
        ExchangeBufs xb{L.spl, NS};
        for (int K = 128 * SPT; K <= NS; K <<= 1) {
            cross_wave_step<SPT>(smp, xb, tid, K - 1, K >> 1);
            for (int j = K >> 2; j >= 64 * SPT; j >>= 1) cross_wave_step<SPT>(smp, xb, tid, j, j);
            dpp_fence<SPT>(smp);
            clean_p64<64, SPT, 32 * SPT, false>(smp);
        }
        lds_barrier();                             // the last step's reads, before the splitters go over its buffer
#pragma unroll
        for (int u = 0; u < SPT; ++u) L.spl[tid * SPT + u] = p64_value(smp[u]);
    }
    lds_barrier();
    LA_CLK(0);
    // 2. bucket of every bin = number of splitters below it (0 .. NS); a slot inside the bucket
    uint64_t x[EC];
    uint32_t b[EC], slot[EC];
#pragma unroll
    for (int r = 0; r < EC; ++r) { x[r] = p64_value(rec[r]); b[r] = 0; }
#pragma unroll
    for (int step = NS / 2; step >= 1; step >>= 1) {
#pragma unroll
        for (int r = 0; r < EC; ++r) b[r] += (L.spl[b[r] + step - 1] < x[r]) ? (uint32_t)step : 0u;
    }
    {
        const uint64_t top = L.spl[NS - 1];
#pragma unroll
        for (int r = 0; r < EC; ++r) b[r] += (top < x[r]) ? 1u : 0u;     // only where b is already NS - 1
    }
    LA_CLK(1);
#pragma unroll
    for (int r = 0; r < EC; ++r) slot[r] = atomicAdd(&L.cnt[b[r]], 1u);
    lds_barrier();
    {   // exclusive scan of the counts (thread t owns buckets t*SPT ..; bucket NS is what is left), largest bucket
        uint32_t c[SPT], sum = 0, m = 0;
#pragma unroll
        for (int u = 0; u < SPT; ++u) { c[u] = L.cnt[tid * SPT + u]; sum += c[u]; m = max(m, c[u]); }
        const uint32_t incl = wave_incl_scan_u32(sum);
        if (tid == 0) m = max(m, L.cnt[NS]);
        m = wave_max_u32(m);
        if (lane == 63) L.misc[wave] = incl;
        if (lane == 0) L.misc[16 + wave] = m;
        lds_barrier();
        const uint4* mv = reinterpret_cast<const uint4*>(L.misc);          // 16 sums, 16 maxima: eight 16-byte reads
        uint32_t base = 0, mx = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 sv = mv[q], xv = mv[4 + q];
            base += (4 * q + 0 < wave ? sv.x : 0u) + (4 * q + 1 < wave ? sv.y : 0u) + (4 * q + 2 < wave ? sv.z : 0u) +
                    (4 * q + 3 < wave ? sv.w : 0u);
            mx = max(max(mx, max(xv.x, xv.y)), max(xv.z, xv.w));
        }
        if (mx > bucket_limit) return false;          // workgroup-uniform: the caller sorts with the full network
        uint32_t first = base + incl - sum;
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            L.cnt[tid * SPT + u] = first | (c[u] << 16);
            first += c[u];
        }
        if (tid == NT - 1) L.cnt[NS] = first | (((uint32_t)N - first) << 16);
    }
    lds_barrier();
    LA_CLK(2);
    // 3. stage bucket by bucket, with (bucket size << 16 | slot) beside every staged bin
#pragma unroll
    for (int r = 0; r < EC; ++r) {
        const uint32_t sc = L.cnt[b[r]];
        L.stage[(sc & 0xFFFFu) + slot[r]] = make_ulonglong2(x[r], (uint64_t)((sc & 0xFFFF0000u) | slot[r]));
    }
    lds_barrier();
    LA_CLK(3);
    // Walk the staged bins wave-striped: 64 consecutive positions per step, a handful of consecutive buckets.
    // Final position = bucket start + members below the bin.  Reading past the bucket's end is harmless -- the bins
    // of later buckets are all larger, and two sentinels close the array -- so the loop has no per-lane bound.
    uint32_t pos[EC];
    uint64_t y[EC];
    const uint64_t* keys = reinterpret_cast<const uint64_t*>(L.stage);       // staged bin j at keys[2 * j]
#pragma unroll
    for (int it = 0; it < EC; ++it) {
        const uint32_t j = (uint32_t)(wave * 64 * EC + it * 64 + lane);
        const ulonglong2 e = L.stage[j];
        y[it] = e.x;
        const uint32_t inf = (uint32_t)e.y;
        const uint32_t s0 = j - (inf & 0xFFFFu);
        const uint32_t cmax = wave_max_u32(inf >> 16);
        const uint64_t* bk = keys + 2 * s0;                          // the bucket's first staged bin
        uint32_t below = 0;
#pragma unroll 2
        for (uint32_t k = 0; k < cmax; k += 4) {                     // past the array's end: kWalkPad sentinels
            const uint64_t o0 = bk[2 * k], o1 = bk[2 * k + 2], o2 = bk[2 * k + 4], o3 = bk[2 * k + 6];
            below += (o0 < y[it] ? 1u : 0u) + (o1 < y[it] ? 1u : 0u) + (o2 < y[it] ? 1u : 0u) + (o3 < y[it] ? 1u : 0u);
        }
        pos[it] = s0 + below;
    }
    LA_CLK(4);
    lds_barrier();                                 // every staged bin is in a register: the final order goes over them
    // 4. final order, back to blocked registers
    uint64_t* sorted = reinterpret_cast<uint64_t*>(L.stage);
#pragma unroll
    for (int it = 0; it < EC; ++it) sorted[pos[it]] = y[it];
    lds_barrier();
#pragma unroll
    for (int r = 0; r < EC; ++r) rec[r] = p64_from(sorted[tid * EC + r]);
    LA_CLK(5);
    return true;
}

// ---- a round whose bins are a few ascending runs: merge them ------------------------------------------------------------
// Out of a round the bins are (ascending totals) + (descending lags).  Where the lags of a round take few distinct values
// -- the flat tail of a power-law topic, ties, zero lags -- the new values are a handful of ascending RUNS (cfg5: 1 700
// runs in round 2, 95 in round 16, 30 in round 32, 4 - 16 from round 48 on; one run = already in order).  Sorting them is
// then a tree of pairwise merges of adjacent runs, ceil(log2 R) levels, each level one pass of the bins through LDS:
// every thread owns 8 consecutive OUTPUT positions, finds where its first output starts in the two runs of its pair
// (merge path: one binary search), and merges sequentially from there -- ~45 dependent LDS reads per thread and level
// where a sample-sorted round costs ~350 LDS operations and 16 barriers.
// Returns false (and leaves rec alone) when there are more than kMaxRuns runs: the caller sorts as before.  *runs_out = the
// number of runs found, so that the caller can decide when to look again.
#ifndef LA_MAX_RUNS
#define LA_MAX_RUNS 32
#endif
constexpr int kMaxRuns = LA_MAX_RUNS;

template <int EC>
__device__ __forceinline__ bool merge_runs_bins(P64 (&rec)[EC], const SampleLds& L, int tid, int* runs_out) {
    constexpr int NT = kSampleThreads, N = EC * NT;
    const int lane = tid & 63, wave = tid >> 6;
    uint64_t* buf_a = reinterpret_cast<uint64_t*>(L.stage);          // [N] the runs
    uint64_t* buf_b = buf_a + N;                                     // [N] the merged runs of a level
    uint32_t* bnd = L.cnt;                                           // [R + 1] first position of every run, then N
    uint32_t* wsum = L.misc;                                         // [16] descents per wavefront
    // Position p of the sorted order lives at word (p % EC) * NT + p / EC of a buffer (register-major): a thread's EC
    // consecutive positions are NT words apart, so the 64 lanes of every blocked read or write hit 64 consecutive words
    // (position-major, they would sit 64 bytes apart: 16-way bank conflicts on every level's output).
    auto at = [](uint32_t p) -> uint32_t { return (p % EC) * NT + p / EC; };
    lds_barrier();                                                   // last round's readers of the staging area are done
    uint64_t v[EC];
#pragma unroll
    for (int r = 0; r < EC; ++r) { v[r] = p64_value(rec[r]); buf_a[r * NT + tid] = v[r]; }
    lds_barrier();
    // descents: position p starts a new run when bins[p - 1] > bins[p]  (p = tid * EC + r; the bins are distinct)
    const uint64_t before = tid > 0 ? buf_a[(EC - 1) * NT + tid - 1] : 0;
    uint32_t starts = 0, mine = 0;                                   // bit r: my r-th position starts a run
#pragma unroll
    for (int r = 0; r < EC; ++r) {
        const bool st = (r == 0 ? before : v[r - 1]) > v[r];
        starts |= st ? (1u << r) : 0u;
        mine += st ? 1u : 0u;
    }
    const uint32_t incl = wave_incl_scan_u32(mine);
    if (lane == 63) wsum[wave] = incl;
    lds_barrier();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const uint32_t x = wsum[w];
        base += w < wave ? x : 0u;
        total += x;
    }
    const int R = (int)total + 1;
    *runs_out = R;
    if (R > kMaxRuns) return false;                                  // workgroup-uniform
    if (R == 1) return true;                                         // in order already: rec is untouched and right
    {
        uint32_t at = 1 + base + incl - mine;                        // run number of my first start
#pragma unroll
        for (int r = 0; r < EC; ++r)
            if (starts & (1u << r)) bnd[at++] = (uint32_t)(tid * EC + r);
        if (tid == 0) { bnd[0] = 0; bnd[R] = (uint32_t)N; }
    }
    lds_barrier();
    // the tree: at level `stride` the runs are bnd[0], bnd[stride], bnd[2 stride] ..; pair k merges runs 2k and 2k + 1 of them
    uint64_t* src = buf_a;
    uint64_t* dst = buf_b;
    for (int stride = 1; stride < R; stride <<= 1) {
        auto at_run = [&](int i) -> uint32_t { return bnd[i < R ? i : R]; };
        const int n_pairs = (R + 2 * stride - 1) / (2 * stride);
        const uint32_t o = (uint32_t)(tid * EC);                     // my first output position
        // the pair my first output falls into: the last k with at_run(2 k stride) <= o
        int k = 0;
        for (int step = kMaxRuns / 2; step >= 1; step >>= 1)         // n_pairs <= kMaxRuns / 2
            if (k + step < n_pairs && at_run(2 * (k + step) * stride) <= o) k += step;
        uint32_t a0 = at_run(2 * k * stride), a1 = at_run((2 * k + 1) * stride), a2 = at_run((2 * k + 2) * stride);
        // merge path: of the d = o - a0 outputs before mine, i come from the first run and d - i from the second
        const uint32_t d = o - a0, len_a = a1 - a0, len_b = a2 - a1;
        uint32_t lo = d > len_b ? d - len_b : 0u, hi = d < len_a ? d : len_a;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (src[at(a0 + mid)] < src[at(a1 + (d - mid - 1))]) lo = mid + 1; else hi = mid;
        }
        uint32_t ia = a0 + lo, ib = a1 + (d - lo);                   // next element of either run
        uint64_t xa = ia < a1 ? src[at(ia)] : ~0ull, xb = ib < a2 ? src[at(ib)] : ~0ull;
        uint64_t out[EC];
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            if (ia >= a1 && ib >= a2) {                              // my outputs run on into the next pair
                ++k;
                a0 = a2; a1 = at_run((2 * k + 1) * stride); a2 = at_run((2 * k + 2) * stride);
                ia = a0; ib = a1;
                xa = ia < a1 ? src[at(ia)] : ~0ull;
                xb = ib < a2 ? src[at(ib)] : ~0ull;
            }
            const bool take_a = ib >= a2 || (ia < a1 && xa < xb);
            out[r] = take_a ? xa : xb;
            if (take_a) { ++ia; xa = ia < a1 ? src[at(ia)] : ~0ull; }
            else { ++ib; xb = ib < a2 ? src[at(ib)] : ~0ull; }
        }
#pragma unroll
        for (int r = 0; r < EC; ++r) dst[r * NT + tid] = out[r];
        lds_barrier();
        uint64_t* t = src; src = dst; dst = t;
    }
#pragma unroll
    for (int r = 0; r < EC; ++r) rec[r] = p64_from(src[r * NT + tid]);
    return true;
}

// ---- a round that moves few bins: only those are sorted ------------------------------------------------------------------
// Out of a round the bins are (ascending totals) + (descending lags).  On a power-law topic most of the consumers stand far
// apart after the first round -- the few that took the huge lags, a long thin tail -- and a round's lags differ by less than
// their distances: they keep their places.  Only the dense bulk moves: cfg5 (8 192 consumers) moves 3 168 bins in round 1,
// 1 263 in round 2, 850 - 1 060 in rounds 3 - 75 and ~500 after that, while the round's bins are up to 600 ascending runs
// (`profiles/archive/r04_cfg5_rounds.txt`).  A bin STAYS where it is when it is larger than every bin before it and smaller than
// every bin behind it (a prefix maximum and a suffix minimum); such a bin separates what comes before it from what comes
// behind it, so the bins that do not stay are sorted among themselves and go back, in order, to the places they came from.
//   1. per thread (EC consecutive bins): ascending inside, above the maximum of everything before, below the minimum of
//      everything behind?  A wavefront whose 64 * EC bins ascend needs no scan for that (its maximum is its last bin); the
//      others scan their lanes (DPP).  Across wavefronts: 16 maxima / minima through LDS.  Only the wavefronts that hold a
//      thread with a bin that moves look at single bins and count them.
//   2. the bins that move are written side by side, in the order of their places (m of them, at most a quarter of the
//      round's bins) and sorted with one or two per thread by the scheme of the sample sort one level down: 64 of them, evenly
//      spaced, are ranked against each other (four per wavefront: one compare + ballot each), every bin finds its bucket
//      among those (7 LDS reads), takes a slot, is staged by bucket and ranked by the walk over its bucket (~m / 64 bins).
//      m <= 64: one wavefront sorts them in registers.
//   3. the sorted bins go back to the places the bins that move came from.
// Six barriers and ~60 LDS operations in the threads that take part; the wavefronts whose bins all stay (13 of 16 on cfg5)
// spend ~100 VALU instructions on finding that out -- on a CU with 16 wavefronts every instruction that all of them execute
// costs 16 cycles, which is what a round is made of.  Returns false -- rec untouched -- when more bins move than fit (or a
// bucket is larger than `bucket_limit`): the caller sorts the round as before.  *moved_out = the number of bins that move.
constexpr int kMovedPad = 164;                               // sentinels behind the staged bins (the walk reads four at a time)
constexpr uint32_t kMovedBucket = 160;

// inclusive prefix maximum over the lanes of the wavefront (row shifts, then the row broadcasts; a lane without a source
// reads zero, the identity)
__device__ __forceinline__ uint64_t wave_incl_max_u64(uint64_t v) {
#define LA_MAX_STEP(CTRL, ROWS)                                                                                          \
    {                                                                                                                    \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROWS, 0xF, false);         \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROWS, 0xF, false); \
        const uint64_t o_ = ((uint64_t)hi_ << 32) | lo_;                                                                 \
        v = o_ > v ? o_ : v;                                                                                             \
    }
    LA_MAX_STEP(0x111, 0xF)        // row_shr:1
    LA_MAX_STEP(0x112, 0xF)        // row_shr:2
    LA_MAX_STEP(0x114, 0xF)        // row_shr:4
    LA_MAX_STEP(0x118, 0xF)        // row_shr:8
    LA_MAX_STEP(0x142, 0xA)        // row_bcast:15 -> rows 1, 3
    LA_MAX_STEP(0x143, 0xC)        // row_bcast:31 -> rows 2, 3
#undef LA_MAX_STEP
    return v;
}

__device__ __forceinline__ uint64_t wave_shr1_u64(uint64_t v) {            // lane - 1's value; lane 0 reads zero
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_mirror_u64(uint64_t v) {          // lane 63 - lane's value
    return ((uint64_t)shfl_mirror<64>((uint32_t)(v >> 32)) << 32) | shfl_mirror<64>((uint32_t)v);
}

template <int J>
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v) {
    return ((uint64_t)shfl_xor<J>((uint32_t)(v >> 32)) << 32) | shfl_xor<J>((uint32_t)v);
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {   // (lane: wavefront-uniform)
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
}

// The small sort of moved_sort_bins: the m bins that move stand side by side in comp[0 .. m); on return they stand there in
// order.  NS samples (64 or 128) cut them into NS + 1 buckets; a bin's place is its bucket's first place plus the number of
// bins of its bucket below it.  CPT = bins per thread (m <= CPT * 1 024).  false: a bucket grew beyond `bucket_limit` (the
// caller sorts the round another way; comp is spent either way).
template <int CPT, int NS>
__device__ __forceinline__ bool moved_small_sort(uint64_t* comp, ulonglong2* stage2, uint64_t* sup, uint64_t* sup1, uint32_t* cnt2,
                                                 const int m, const int tid, const uint32_t bucket_limit LA_CLK_PARAM) {
    constexpr int NT = kSampleThreads;
    constexpr int NSUP = NS / 8;                                 // every eighth sample
    static_assert(NS == 64 || NS == 128, "one or two samples per lane");
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        // NS evenly spaced samples (distinct places: m > NS), every wavefront ranks NS / 16 of them: the number of samples
        // below one is its place among them
        if constexpr (NS == 64) {
            const uint64_t smp = comp[((uint32_t)lane * (uint32_t)m + (uint32_t)m / 2u) >> 6];
#pragma unroll
            for (int k = 0; k < 64 / (NT / 64); ++k) {
                const uint64_t s = readlane_u64(smp, wave * (64 / (NT / 64)) + k);
                const int rank = __builtin_popcountll(__builtin_amdgcn_ballot_w64(smp < s));
                if (lane == 0) {
                    sup[rank] = s;
                    if ((rank & 7) == 7) sup1[rank >> 3] = s;
                }
            }
        } else {
            const uint64_t smp0 = comp[((uint32_t)lane * (uint32_t)m + (uint32_t)m / 2u) >> 7];
            const uint64_t smp1 = comp[((uint32_t)(64 + lane) * (uint32_t)m + (uint32_t)m / 2u) >> 7];
            const uint64_t mine = wave < 8 ? smp0 : smp1;        // wavefronts 0 .. 7 rank samples 0 .. 63, the others 64 .. 127
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint64_t s = readlane_u64(mine, (wave & 7) * 8 + k);
                const int rank = __builtin_popcountll(__builtin_amdgcn_ballot_w64(smp0 < s)) +
                                 __builtin_popcountll(__builtin_amdgcn_ballot_w64(smp1 < s));
                if (lane == 0) {
                    sup[rank] = s;
                    if ((rank & 7) == 7) sup1[rank >> 3] = s;
                }
            }
        }
        if (tid < kMovedPad) stage2[m + tid] = make_ulonglong2(~0ull, 0);     // sentinels behind the staged bins
    }
    lds_barrier();                                               // (3)
    LA_CLK(10);
    uint64_t x[CPT];
    uint32_t b2[CPT], slot[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int c = tid + u * NT;
        x[u] = ~0ull; b2[u] = 0; slot[u] = 0;
        if (u * NT + wave * 64 < m) {                            // (wavefront-uniform: a wavefront without a bin skips the phase)
            const bool valid = c < m;
            x[u] = comp[valid ? c : 0];
            // bucket = the number of samples below the bin, in two steps (two dependent trips to LDS where a binary search
            // makes seven): which eighth (sixteenth) of the samples, then where inside it
            uint32_t c1 = 0, c2 = 0;
#pragma unroll
            for (int i = 0; i < NSUP; ++i) c1 += (sup1[i] < x[u]) ? 1u : 0u;
            const uint32_t r1 = c1 < (uint32_t)(NSUP - 1) ? c1 : (uint32_t)(NSUP - 1);
            const uint64_t* row = sup + 8u * r1;
#pragma unroll
            for (int i = 0; i < 8; ++i) c2 += (row[i] < x[u]) ? 1u : 0u;           // (c1 < NSUP: row[7] is not below)
            const uint32_t b = 8u * r1 + c2;
            b2[u] = b;
            if (valid) slot[u] = atomicAdd(&cnt2[b], 1u);
        }
    }
    lds_barrier();                                               // (4)
    LA_CLK(11);
    // first places of the NS + 1 buckets, the largest bucket: every wavefront for itself (a scan over its lanes), no barrier
    uint32_t first[CPT], size[CPT];
    if constexpr (NS == 64) {
        const uint32_t c = cnt2[lane];
        const uint32_t incl = wave_incl_scan_u32(c);
        const uint32_t last = cnt2[64];
        if (wave_max_u32(max(c, last)) > bucket_limit) return false;             // (workgroup-uniform: same counts everywhere)
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const uint32_t bl = b2[u] & 63u;
            const uint32_t f = (uint32_t)__shfl((int)(incl - c), (int)bl), z = (uint32_t)__shfl((int)c, (int)bl);
            first[u] = b2[u] < 64u ? f : all;
            size[u] = b2[u] < 64u ? z : last;
        }
    } else {
        const uint32_t c0 = cnt2[lane], c1 = cnt2[64 + lane];
        const uint32_t last = cnt2[128];
        if (wave_max_u32(max(max(c0, c1), last)) > bucket_limit) return false;   // (workgroup-uniform)
        const uint32_t incl0 = wave_incl_scan_u32(c0);
        const uint32_t all0 = (uint32_t)__builtin_amdgcn_readlane((int)incl0, 63);
        const uint32_t incl1 = wave_incl_scan_u32(c1) + all0;
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl1, 63);
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const uint32_t bl = b2[u] & 63u;
            const uint32_t f0 = (uint32_t)__shfl((int)(incl0 - c0), (int)bl), z0 = (uint32_t)__shfl((int)c0, (int)bl);
            const uint32_t f1 = (uint32_t)__shfl((int)(incl1 - c1), (int)bl), z1 = (uint32_t)__shfl((int)c1, (int)bl);
            first[u] = b2[u] < 64u ? f0 : (b2[u] < 128u ? f1 : all);
            size[u] = b2[u] < 64u ? z0 : (b2[u] < 128u ? z1 : last);
        }
    }
    LA_CLK(12);
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int c = tid + u * NT;
        if (c < m) stage2[first[u] + slot[u]] = make_ulonglong2(x[u], (uint64_t)((size[u] << 16) | slot[u]));
    }
    lds_barrier();                                               // (5)
    LA_CLK(13);
    const uint64_t* keys = reinterpret_cast<const uint64_t*>(stage2);
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int c = tid + u * NT;
        if (u * NT + wave * 64 < m) {                            // (wavefront-uniform: the walk's bound is a wavefront maximum)
            const bool valid = c < m;
            const ulonglong2 e = valid ? stage2[c] : make_ulonglong2(~0ull, 0);
            const uint32_t inf = (uint32_t)e.y;
            const uint32_t s0 = valid ? (uint32_t)c - (inf & 0xFFFFu) : 0u;
            const uint32_t cmax = wave_max_u32(inf >> 16);
            const uint64_t* bk = keys + 2 * s0;
            uint32_t below = 0;
#pragma unroll 2
            for (uint32_t k = 0; k < cmax; k += 4) {
                const uint64_t o0 = bk[2 * k], o1 = bk[2 * k + 2], o2 = bk[2 * k + 4], o3 = bk[2 * k + 6];
                below += (o0 < e.x ? 1u : 0u) + (o1 < e.x ? 1u : 0u) + (o2 < e.x ? 1u : 0u) + (o3 < e.x ? 1u : 0u);
            }
            if (valid) comp[s0 + below] = e.x;                   // (comp was last read before barrier 4)
        }
    }
    lds_barrier();                                               // (6)
    LA_CLK(14);
    return true;
}

template <int EC>
__device__ __forceinline__ bool moved_sort_bins(P64 (&rec)[EC], const SampleLds& L, int tid, uint32_t bucket_limit, int* moved_out,
                                                const int parity) {
    constexpr int NT = kSampleThreads;
    constexpr int CPT = EC >= 8 ? 2 : 1;                         // bins of the small sort per thread
    constexpr int kCapN = 256 * EC;                              // its capacity (a quarter of the round's bins) in the narrow form
    constexpr bool kWide = LA_WIDE_MOVED != 0 && EC >= 4;        // the wide form: twice the capacity, twice the bins per thread, 128 buckets
    constexpr int kCap = kWide ? 2 * kCapN : kCapN;              // what the side-by-side array holds
    static_assert(kCapN <= CPT * NT && NT == 1024, "capacity");
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // [16] a wavefront's largest bin, [16] its smallest, [16] whether its bins ascend: two sets, calls alternate (`parity`) -- a round
    // in which nothing moves leaves after barrier (1), and the fastest wavefront writes the next call's words while the slowest
    // still reads this call's
    uint64_t* wmax = reinterpret_cast<uint64_t*>(L.moved + ((parity & 1) ? 256 : 0));
    uint64_t* wmin = wmax + 16;
    uint32_t* wasc = L.moved + 320 + ((parity & 1) ? 16 : 0);
    uint32_t* wcnt = L.moved + 64;                               // [16] its bins that move
    uint32_t* cnt2 = L.moved + 80;                               // [128 + 1] bucket counts (64 + 1 in the narrow form)
    uint32_t* taken = L.moved + 210;                             // [1] places of the side-by-side array handed out so far
    uint64_t* comp = reinterpret_cast<uint64_t*>(L.stage);       // [kCap] the bins that move, in the order of their places; then sorted
    ulonglong2* stage2 = reinterpret_cast<ulonglong2*>(comp + kCap);     // [kCap + kMovedPad] staged by bucket
    uint64_t* sup = reinterpret_cast<uint64_t*>(stage2 + kCap + kMovedPad);   // [64 | 128] the samples in order
    uint64_t* sup1 = sup + 128;                                  // [8 | 16] every eighth of them (sup[7], sup[15] ..)
    LA_CLK_START;
    // 1. who stays
    uint64_t v[EC];
#pragma unroll
    for (int r = 0; r < EC; ++r) v[r] = p64_value(rec[r]);
    bool asc = true;
#pragma unroll
    for (int r = 1; r < EC; ++r) asc &= v[r - 1] < v[r];
    const uint64_t prev_last = wave_shr1_u64(v[EC - 1]);         // (every lane executes the shift: la_sort64.h on masked sources)
    const bool asc_w = asc & ((lane == 0) | (prev_last < v[0]));
    const bool wave_asc = __builtin_amdgcn_ballot_w64(!asc_w) == 0;              // wavefront-uniform
    uint64_t pm_in = 0, sm_in = ~0ull;                           // over the earlier / later lanes of this wavefront
    uint64_t w_hi, w_lo;
    if (wave_asc) {
        w_hi = readlane_u64(v[EC - 1], 63);
        w_lo = readlane_u64(v[0], 0);
    } else {
        uint64_t tmax = v[0], tmin = v[0];
#pragma unroll
        for (int r = 1; r < EC; ++r) { tmax = v[r] > tmax ? v[r] : tmax; tmin = v[r] < tmin ? v[r] : tmin; }
        const uint64_t incl = wave_incl_max_u64(tmax);
        pm_in = wave_shr1_u64(incl);
        w_hi = readlane_u64(incl, 63);
        // the suffix minimum is the prefix maximum of the complements, lanes mirrored
        const uint64_t incl_m = wave_incl_max_u64(wave_mirror_u64(~tmin));
        sm_in = ~wave_mirror_u64(wave_shr1_u64(incl_m));
        w_lo = ~readlane_u64(incl_m, 63);
    }
    if (lane == 0) { wmax[wave] = w_hi; wmin[wave] = w_lo; wasc[wave] = wave_asc ? 1u : 0u; }
    if (tid < 129) cnt2[tid] = 0;
    if (tid == 129) *taken = 0;
    lds_barrier();                                               // (1)
    LA_CLK(7);
    // the sixteen wavefronts' words, one read each (lane l and its copies in the other rows hold wavefront l & 15's)
    const uint64_t hi_l = wmax[lane & 15], lo_l = wmin[lane & 15];
    {
        // nothing moves at all?  Every wavefront ascends and begins above the one before it: the round's bins are in order, and
        // the round is over -- no second barrier, no counts (a topic of equal lags, a consumer group that has caught up: every round)
        const uint32_t as_l = wasc[lane & 15];
        const uint32_t hb_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)hi_l, 0x111, 0xF, 0xF, false);          // row_shr:1
        const uint32_t hb_hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(hi_l >> 32), 0x111, 0xF, 0xF, false);
        const uint64_t hi_before = ((uint64_t)hb_hi << 32) | hb_lo;                // (a row's first lane: zero)
        const bool fine = as_l != 0 && hi_before < lo_l;
        if (__builtin_amdgcn_ballot_w64(!fine) == 0) {
            *moved_out = 0;
            return true;
        }
    }
    uint64_t pm_w, sm_w;                                         // over the earlier / later wavefronts
    {
        uint64_t a = lane < wave ? hi_l : 0;
        uint64_t b = (lane > wave && lane < NT / 64) ? lo_l : ~0ull;
#define LA_RED_STEP(J)                                                        \
        {                                                                     \
            const uint64_t oa_ = shfl_xor_u64<J>(a), ob_ = shfl_xor_u64<J>(b); \
            a = oa_ > a ? oa_ : a;                                            \
            b = ob_ < b ? ob_ : b;                                            \
        }
        LA_RED_STEP(1) LA_RED_STEP(2) LA_RED_STEP(4) LA_RED_STEP(8)
#undef LA_RED_STEP
        pm_w = readlane_u64(a, 0);
        sm_w = readlane_u64(b, 0);
    }
    const uint64_t before = pm_in > pm_w ? pm_in : pm_w, behind = sm_in < sm_w ? sm_in : sm_w;
    const bool moves = !(asc & (before < v[0]) & (v[EC - 1] < behind));
    const bool wave_moves = __builtin_amdgcn_ballot_w64(moves) != 0;             // wavefront-uniform
    uint32_t mask = 0, off = 0;                                  // my bins that move; how many move before them
    if (wave_moves) {
        // bin r stays when everything before it is smaller and everything behind it is larger
        uint64_t sm[EC];
        {
            uint64_t x = behind;
#pragma unroll
            for (int r = EC - 1; r >= 0; --r) { sm[r] = x; x = v[r] < x ? v[r] : x; }
        }
        uint64_t run = before;
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            mask |= ((run < v[r]) & (v[r] < sm[r])) ? 0u : (1u << r);
            run = v[r] > run ? v[r] : run;
        }
        const uint32_t mine = (uint32_t)__builtin_popcount(mask);
        const uint32_t incl = wave_incl_scan_u32(mine);
        off = incl - mine;
        // 2. the bins that move, side by side: the wavefront takes as many places as it needs (in whatever order the
        //    wavefronts come: the order of the PLACES is kept in `off`, for the way back) and fills them
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t k = 0;
        if (lane == 0) { k = atomicAdd(taken, total); wcnt[wave] = total; }
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k) + off;
#pragma unroll
        for (int r = 0; r < EC; ++r)
            if (mask & (1u << r)) {
                if (k < (uint32_t)kCap) comp[k] = v[r];
                ++k;
            }
    } else if (lane == 0) {
        wcnt[wave] = 0;
    }
    lds_barrier();                                               // (2)
    LA_CLK(8);
    int m = 0;
    {
        const uint4* cv = reinterpret_cast<const uint4*>(wcnt);
        uint32_t base = 0;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const uint4 c = cv[q4];
            const uint32_t c0 = __builtin_amdgcn_readfirstlane(c.x), c1 = __builtin_amdgcn_readfirstlane(c.y),
                           c2 = __builtin_amdgcn_readfirstlane(c.z), c3 = __builtin_amdgcn_readfirstlane(c.w);
            base += (4 * q4 + 0 < wave ? c0 : 0u) + (4 * q4 + 1 < wave ? c1 : 0u) + (4 * q4 + 2 < wave ? c2 : 0u) +
                    (4 * q4 + 3 < wave ? c3 : 0u);
            m += (int)(c0 + c1 + c2 + c3);
        }
        off += base;
    }
    *moved_out = m;
    if (m == 0) return true;                                     // in order already
    if (m > kCap) return false;                                  // (workgroup-uniform)
    if (m <= 64) {
        if (wave == 0) {                                         // one wavefront sorts them
            P64 s = p64_from(lane < m ? comp[lane] : ~0ull);
            asm volatile("s_nop 1" : "+v"(s.lo), "+v"(s.hi));
            bitonic_sort_lanes_p64<64>(s);
            if (lane < m) comp[lane] = p64_value(s);
        }
        lds_barrier();
    } else if (__builtin_expect(m <= kCapN, 1)) {
        if (!moved_small_sort<CPT, 64>(comp, stage2, sup, sup1, cnt2, m, tid, bucket_limit LA_CLK_ARG)) return false;
    } else {
        if constexpr (kWide) {
            // more bins move than the narrow form holds, not more than twice as many (the dense bulk of a power-law topic of
            // 131 072 .. 262 144 or of ~2 M partitions over 8 192 consumers: 2 050 - 3 100 per round): the same sort with four bins
            // per thread and 128 buckets -- a round of 27 - 33 k cycles where the sample sort over all bins takes 51 k and the run
            // merge 55 - 59 k.  Decided per round from what the round's bins say (round 6's first wide form was chosen by the
            // host from the number of rounds, kept 64 buckets, and lost wherever ~4 000 bins moved).
            if (m > LA_WIDE_MOVED_LIMIT * EC) return false;
            if (!moved_small_sort<2 * CPT, 128>(comp, stage2, sup, sup1, cnt2, m, tid, bucket_limit LA_CLK_ARG)) return false;
        } else {
            return false;
        }
    }
    // 3. back to the places the bins that move came from
    if (wave_moves) {
        uint32_t k = off;
#pragma unroll
        for (int r = 0; r < EC; ++r)
            if (mask & (1u << r)) rec[r] = p64_from(comp[k++]);
    }
    LA_CLK(15);
    return true;
}

template <int EC>
__device__ void greedy_rounds_packed(const LargeArgs& a, const uint64_t* key, int64_t P, int C, int64_t rounds,
                                     int idx_bits, void* smem, const uint32_t* lag32, uint16_t* idx16, const int64_t cpad,
                                     const bool zero_tail) {
    const int tid = threadIdx.x;
    const int n = EC * blockDim.x;
    constexpr int kSpan = 64 * EC;                       // elements of one wavefront
    const uint32_t idx_mask = (1u << idx_bits) - 1;
    uint64_t* s_bin = reinterpret_cast<uint64_t*>(smem);
    [[maybe_unused]] const SampleLds L = sample_lds_carve(smem, n);
    [[maybe_unused]] const bool use_sample = (EC >= 2) && blockDim.x == kSampleThreads && a.no_sample_sort != 1;
    if (use_sample && tid < kWalkPad) L.stage[n + tid] = make_ulonglong2(~0ull, 0);      // first read after many barriers
    P64 rec[EC];
    // raw sorted keys of the round's partitions, one round ahead.  The loads are UNCONDITIONAL (index clamped) and
    // issued back to back: a branch around a load makes hipcc wait for it before issuing the next one -- eight
    // serialized memory round trips per round here, 6k of a round's 52k cycles; the select happens at the use, a
    // whole round of sorting later.
    // lag[]: the lags themselves (since round 6: the keys' flip is undone when they arrive, not when they are added).  narrow: the
    // 32-bit lags emit_ids_kernel left, a round's stretch `cpad` elements after the one before (rounds_io).
    const bool narrow = EC >= LA_ROUNDS_NARROW_EC && EC >= 4 && lag32 != nullptr;     // (workgroup-uniform; never for fewer bins per thread)
    const int64_t cap8 = (2 * P - 8) & ~(int64_t)3;                              // the last aligned place a 8-element read may start
    uint64_t lag[EC];
#pragma unroll
    for (int r = 0; r < EC; ++r) {
        const int i = tid * EC + r;
        // pads (slots beyond the C consumers) sort behind every bin and, like the bins, are all different
        rec[r] = p64_from(i < C ? (uint64_t)i : ((~0ull << idx_bits) | (uint64_t)i));
        if (narrow) lag[r] = lag32[i < cap8 ? i : cap8];                          // (round 0 starts at element 0)
        else lag[r] = key[i < P ? i : P - 1] ^ kLagKeyFlip;
    }
    [[maybe_unused]] int64_t next_look = 1;              // the next round that looks whether its bins are a few ascending runs
    [[maybe_unused]] int64_t next_moved = 2;             // the next round that looks whether few of its bins move (round 1 sorts
                                                         // what round 0 made of equal bins: every bin moves)
    [[maybe_unused]] int moved_wait = 0;
    [[maybe_unused]] int moved_calls = 0;                // (the moved-bins sort alternates between two sets of its first words)
    // The lags descend, so a round whose FIRST lag is zero hands out zeros only, and so does every round behind it: nothing is
    // added any more, the order the bins stand in when that round begins is final and the rounds behind it only write it down again
    // -- no order check, no barrier.  (A consumer group that has caught up on most partitions: the long tail of a big topic's
    // rounds.)  Which round that is, is found ONCE, before the loop -- 64 rounds probed per step by the lanes of every wavefront for
    // itself: two or three dependent loads -- because anything added to the loop for it cost every topic (a load of the next
    // round's first lag per round, scalar, vector or by one lane: cfg5 +7 %: one more instruction through the one compute unit's
    // address path per wavefront and round, or a memory round trip inside the next LDS barrier's lgkmcnt(0)).
    int64_t first_zero_round = rounds;                   // (workgroup-uniform)
    if (zero_tail) {                                     // (the topic's smallest lag is zero at all)
        const int lane = tid & 63;
        int64_t lo = 0, hi = rounds;                     // rounds below lo begin with a lag > 0; round hi (if any) with zero
        while (lo < hi) {
            const int64_t step = (hi - lo + 63) / 64;
            const int64_t r = lo + lane * step;
            uint64_t first = 0;
            if (r < hi) {
                if (narrow) { const int64_t f = r * cpad; first = lag32[f < cap8 ? f : cap8]; }
                else first = key[r * (int64_t)C] ^ kLagKeyFlip;
            }
            const uint64_t zeros = __builtin_amdgcn_ballot_w64(first == 0);      // (lanes beyond hi count as zero)
            if (zeros == 0) { lo = lo + 63 * step + 1; continue; }                // (cannot happen while 64 * step >= hi - lo; kept for safety)
            const int k = __builtin_ctzll(zeros);
            hi = lo + k * step < hi ? lo + k * step : hi;
            lo = k > 0 ? lo + (int64_t)(k - 1) * step + 1 : lo;
            if (k == 0) hi = lo;
        }
        first_zero_round = lo;
    }
    const int tid_fixed = tid;
    for (int64_t q = 0; q < rounds; ++q) {
        // The thread index, opaque once per round: everything a round derives from it (lane predicates, LDS addresses, masks)
        // is then computed in the round -- a few VALU instructions -- instead of once before the loop, where hipcc parks the
        // values in scratch memory (the kernel sits at its 128-register cap) and reloads them in the middle of the chain.
        int tid = tid_fixed;
        asm volatile("" : "+v"(tid));
        [[maybe_unused]] int path_ = 0, moved_ = 0;
        if (q > 0 && q <= first_zero_round) {
            bool sorted = false;
#if LA_ROUND1_REVERSE
            if constexpr (EC >= 2) {
                // Round 0 handed the DESCENDING lags to the bins in index order (all totals were 0): where the lags differ the
                // bins stand in exactly the reverse of the order round 1 needs, and only runs of equal lags (index order = the
                // right order already) are out of place once the C bins are turned around.  So round 1 turns them around through
                // LDS and hands the result to the moved-bins sort (cfg5: 82 of 8 192 bins move then) instead of sorting 8 192
                // bins from scratch -- when most threads' bins descend; a topic of equal lags is in order as it stands.
                if (q == 1 && use_sample && a.no_moved_sort == 0 && a.no_sample_sort == 0) {
                    uint32_t* desc_threads = L.moved + 212;
                    if (tid == 0) *desc_threads = 0;
                    lds_barrier();
                    uint64_t v[EC];
                    bool desc = true;
#pragma unroll
                    for (int r = 0; r < EC; ++r) v[r] = p64_value(rec[r]);
#pragma unroll
                    for (int r = 1; r < EC; ++r) desc &= v[r - 1] > v[r];
                    desc &= tid * EC + EC <= C;
                    const uint64_t votes = __builtin_amdgcn_ballot_w64(desc);
                    if ((tid & 63) == 0 && votes) atomicAdd(desc_threads, (uint32_t)__builtin_popcountll(votes));
#pragma unroll
                    for (int r = 0; r < EC; ++r) s_bin[tid * EC + r] = v[r];
                    lds_barrier();
                    if (4u * *desc_threads >= 3u * (uint32_t)(C / EC)) {          // (workgroup-uniform)
#pragma unroll
                        for (int r = 0; r < EC; ++r) {
                            const int i = tid * EC + r;
                            if (i < C) rec[r] = p64_from(s_bin[C - 1 - i]);
                        }
                        next_moved = 1;
                    }
                    // (the moved-bins sort's first barrier stands between these reads and its writes to the same memory)
                }
            }
#endif
            if constexpr (EC >= 2) {
                // few bins move?  A topic that has the property has it round after round (the bulk of a power-law topic);
                // one that does not (lags spread like the totals) has every bin move in every round: after a look that
                // found more than twice what fits, the next one waits 1, 2, 4 .. 32 rounds.
                if (use_sample && a.no_moved_sort == 0 && q >= next_moved) {
                    int moved = 0;
                    sorted = moved_sort_bins<EC>(rec, L, tid, a.no_sample_sort == 2 ? 20u : kMovedBucket, &moved, moved_calls++);
                    moved_ = moved; if (sorted) path_ = 1;
                    if (sorted || moved <= 512 * EC) moved_wait = 0;
                    else moved_wait = moved_wait ? (moved_wait < 32 ? 2 * moved_wait : 32) : 1;
                    next_moved = q + 1 + moved_wait;
                }
                // few ascending runs?  The number of runs falls from round to round on the topics that have the property
                // and stays in the thousands on those that do not: look again after as many rounds as the last look
                // missed by (a factor of two per round is about what cfg5 does), every round once it has held.
                if (!sorted && use_sample && a.no_run_merge == 0 && q >= next_look) {
                    int runs = 0;
                    sorted = merge_runs_bins<EC>(rec, L, tid, &runs);
                    if (sorted) path_ = 2;
                    int wait = 0;
                    for (int x = runs; x > 2 * kMaxRuns && wait < 16; x >>= 1) ++wait;
                    next_look = q + 1 + wait;
                    if (!sorted) lds_barrier();              // (its LDS use ends before the sample sort's begins)
                }
                if (!sorted && use_sample) { sorted = sample_sort_bins<EC>(rec, L, tid, a.no_sample_sort == 2 ? 6u : kMaxBucket); if (sorted) path_ = 3; }
            }
            if (!sorted) {
                path_ = 4;
                // sort the n bins: inside every wavefront first, then merges across wavefronts
                dpp_fence<EC>(rec);
                bitonic_sort_tile_p64<64, EC>(rec);
                ExchangeBufs xb{s_bin, n};
                for (int K = 2 * kSpan; K <= n; K <<= 1) {
                    cross_wave_step<EC>(rec, xb, tid, K - 1, K >> 1);                        // mirror: i <-> i ^ (K-1)
                    for (int j = K >> 2; j >= kSpan; j >>= 1) cross_wave_step<EC>(rec, xb, tid, j, j);
                    dpp_fence<EC>(rec);
                    clean_p64<64, EC, kSpan / 2, false>(rec);                                 // i <-> i ^ j, j < span
                }
                if (n > kSpan) lds_barrier();      // the last step's reads, before anything else goes over its buffer
            }
        }
        // position i of the sorted bins takes partition q*C + i; the next round's lags are fetched now
        LA_STAMP(q, path_, moved_);
        LA_CLK_START;
        // A thread's EC positions are consecutive (blocked layout), so are its keys and its results: 16 bytes per
        // instruction -- two keys per load, four results per store -- instead of one element each.  This kernel runs on
        // ONE CU, whose address path handled 16 narrow instructions per thread and round, 64 cache lines apiece.
        uint64_t next_lag[EC];
        if (narrow) {
            if constexpr (EC >= 4) {
                typedef uint32_t U32x4 __attribute__((ext_vector_type(4), aligned(16)));
                int64_t sb = (q + 1) * cpad + tid * EC;                          // a multiple of 4: 16-byte aligned
                sb = sb < cap8 ? sb : cap8;                                      // (past the topic: somewhere inside the buffer, never used)
#pragma unroll
                for (int r = 0; r < EC; r += 4) {
                    const U32x4 v = *reinterpret_cast<const U32x4*>(lag32 + sb + r);
                    next_lag[r] = v.x; next_lag[r + 1] = v.y; next_lag[r + 2] = v.z; next_lag[r + 3] = v.w;
                }
            }
        } else if constexpr (EC >= 2) {
            struct __attribute__((aligned(8))) U64x2 { uint64_t x, y; };
            // No branch around the loads, not even a uniform one (round 3 had `if (P >= 2)` here and hipcc put an
            // s_waitcnt vmcnt(0) between every two of them: four serialized round trips to L2 per round, ~2k cycles).  A
            // one-partition topic reads key[0 .. 1]: inside the buffer, whose size is rounded up to 256 bytes (sort_layout).
            const int64_t last_pair = P >= 2 ? P - 2 : 0;
#pragma unroll
            for (int r = 0; r < EC; r += 2) {
                const int64_t s = (q + 1) * C + tid * EC + r;
                const int64_t base = s < last_pair ? s : last_pair;              // the pair stays inside the array
                const U64x2 v = *reinterpret_cast<const U64x2*>(key + base);
                next_lag[r] = (base == s ? v.x : v.y) ^ kLagKeyFlip;        // s == P - 1: its key is the pair's second word
                next_lag[r + 1] = v.y ^ kLagKeyFlip;                        // (positions past P - 1 are never used)
            }
        } else {
            const int64_t s = (q + 1) * C + tid;
            next_lag[0] = key[s < P ? s : P - 1] ^ kLagKeyFlip;
        }
        int32_t who[EC];
        bool live[EC];
        if (EC >= 4 && C == n && (q + 1) * C <= P) {
            // a full round of a topic whose consumers fill every slot (cfg5: 127 of 128 rounds): every bin takes a partition
            // -- no predicates, no 64-bit position arithmetic, two 16-byte stores
#pragma unroll
            for (int r = 0; r < EC; ++r) {
                const uint64_t nb = p64_value(rec[r]) + (lag[r] << idx_bits);       // Main.java:265
                rec[r] = p64_from(nb);
                who[r] = (int32_t)((uint32_t)nb & idx_mask);
                lag[r] = next_lag[r];
            }
            if constexpr (EC >= 4) {
                if (narrow) {
                    // the thread's EC consumer indices as 16-bit words: one aligned store (two 16-byte stores of 32-bit words before)
                    typedef uint32_t U32xH __attribute__((ext_vector_type(EC / 2), aligned(2 * EC)));
                    U32xH v;
#pragma unroll
                    for (int r = 0; r < EC; r += 2) v[r / 2] = (uint32_t)who[r] | ((uint32_t)who[r + 1] << 16);
                    *reinterpret_cast<U32xH*>(idx16 + q * cpad + tid * EC) = v;
                } else {
                    typedef int I32x4 __attribute__((ext_vector_type(4), aligned(4)));
                    int32_t* dst = a.out_rank + a.p0 + q * C + tid * EC;
#pragma unroll
                    for (int r = 0; r < EC; r += 4) {
                        const I32x4 v4 = {who[r], who[r + 1], who[r + 2], who[r + 3]};
                        *reinterpret_cast<I32x4*>(dst + r) = v4;
                    }
                }
            }
            LA_CLK(6);
            continue;
        }
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const int i = tid * EC + r;
            const int64_t s = q * C + i;
            live[r] = i < C && s < P;
            const uint64_t nb = p64_value(rec[r]) + (lag[r] << idx_bits);       // Main.java:265
            if (live[r]) rec[r] = p64_from(nb);
            who[r] = (int32_t)((uint32_t)nb & idx_mask);    // consumer index; map_ranks_kernel turns it into the rank
            lag[r] = next_lag[r];
        }
        if (EC >= 4 && narrow) {
            if constexpr (EC >= 4) {
                uint16_t* dst = idx16 + q * cpad + tid * EC;
                if (live[EC - 1]) {                         // the last is live: so are all before it
                    typedef uint32_t U32xH __attribute__((ext_vector_type(EC / 2), aligned(2 * EC)));
                    U32xH v;
#pragma unroll
                    for (int r = 0; r < EC; r += 2) v[r / 2] = (uint32_t)who[r] | ((uint32_t)who[r + 1] << 16);
                    *reinterpret_cast<U32xH*>(dst) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < EC - 1; ++r)
                        if (live[r]) dst[r] = (uint16_t)who[r];
                }
            }
        } else if constexpr (EC >= 4) {
            typedef int I32x4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
            for (int r = 0; r < EC; r += 4) {
                int32_t* dst = a.out_rank + a.p0 + q * C + tid * EC + r;
                if (live[r + 3]) {                          // the fourth is live: so are the three before it
                    const I32x4 v = {who[r], who[r + 1], who[r + 2], who[r + 3]};
                    *reinterpret_cast<I32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (live[r + k]) dst[k] = who[r + k];
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < EC; ++r)
                if (live[r]) a.out_rank[a.p0 + q * C + tid * EC + r] = who[r];
        }
        LA_CLK(6);
    }
    LA_STAMP(rounds < 258 ? rounds : 258, 9, 0);
    if (a.out_total) {
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const uint64_t v = p64_value(rec[r]);
            const uint32_t who = (uint32_t)v & idx_mask;
            if (who < (uint32_t)C) a.out_total[a.c0 + who] = (int64_t)(v >> idx_bits);
        }
    }
}

// ---- kernel 3: round-structured greedy, one workgroup -----------------------------------------------
// n = EC * blockDim.x consumer slots (power of two >= C); slot i = tid*EC + r.
template <int EC>
__global__ __launch_bounds__(1024) void greedy_rounds_kernel(LargeArgs a0, SortBufs b0, const LargeItem* items, char* scratch,
                                                            const int32_t* order) {
    LA_PICK_ITEM(a, b, a0, b0, items, scratch, order[blockIdx.x])               // one workgroup per topic of this (EC, threads) class
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int tid = threadIdx.x;
    const int n = EC * blockDim.x;
    uint32_t* s_hi = smem;
    uint32_t* s_lo = smem + n;
    uint32_t* s_tb = smem + 2 * n;
    const uint64_t* key = key_buf(b, b.ctl->cur[kFinal]);
    const int64_t P = a.n_part;
    const int C = (int)a.n_cons;

    const int64_t rounds = (P + C - 1) / C;
    {
        // Packed bins (total << idx_bits) | index when nothing can overflow: the keys are sorted, so the
        // first one carries the largest lag and the last one the smallest (rounds_io).
        const RoundsIo io = rounds_io(a, key, P, C);
        if (io.idx_bits) {
            const uint32_t other = b.ctl->cur[kFinal] ^ 1u;
            greedy_rounds_packed<EC>(a, key, P, C, rounds, io.idx_bits, smem,
                                     io.narrow ? reinterpret_cast<const uint32_t*>(key_buf(b, other)) : nullptr,
                                     reinterpret_cast<uint16_t*>(val_buf(b, other)), io.cpad, io.zero_tail);
            return;
        }
    }
    Rec rec[EC];
#pragma unroll
    for (int r = 0; r < EC; ++r) {
        const int i = tid * EC + r;
        if (i < C) { rec[r].hi = (uint32_t)(kTotalBias >> 32); rec[r].lo = 0; rec[r].tb = (uint32_t)i; }
        else rec[r].hi = rec[r].lo = rec[r].tb = 0xFFFFFFFFu;
    }
    for (int64_t q = 0; q < rounds; ++q) {
        if (q > 0) {
            // bitonic sort of the n bins, ascending by (biased total, index)
            for (int k = 2; k <= n; k <<= 1) {
                for (int j = k >> 1; j >= 64 * EC; j >>= 1) {            // across waves: LDS
#pragma unroll
                    for (int r = 0; r < EC; ++r) {
                        const int i = tid * EC + r;
                        s_hi[i] = rec[r].hi; s_lo[i] = rec[r].lo; s_tb[i] = rec[r].tb;
                    }
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < EC; ++r) {
                        const int i = tid * EC + r;
                        Rec o;
                        o.hi = s_hi[i ^ j]; o.lo = s_lo[i ^ j]; o.tb = s_tb[i ^ j];
                        const bool keep_min = (((i & j) == 0) == ((i & k) == 0));
                        const bool take = (rec_less(o, rec[r]) == keep_min);
                        rec[r].hi = take ? o.hi : rec[r].hi;
                        rec[r].lo = take ? o.lo : rec[r].lo;
                        rec[r].tb = take ? o.tb : rec[r].tb;
                    }
                    __syncthreads();
                }
                for (int jl = 32; jl >= 1; jl >>= 1) {                    // across lanes: DPP
                    const int j = jl * EC;
                    if (j < k) {
#pragma unroll
                        for (int r = 0; r < EC; ++r) {
                            const int i = tid * EC + r;
                            cmpx_lanes_dyn(rec[r], jl, (((i & j) == 0) == ((i & k) == 0)));
                        }
                    }
                }
#pragma unroll
                for (int J = EC / 2; J >= 1; J >>= 1) {                   // inside the lane: registers
                    if (J < k) {
#pragma unroll
                        for (int r = 0; r < EC; ++r)
                            if ((r & J) == 0) cmpx_regs(rec[r], rec[r | J], ((tid * EC + r) & k) == 0);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const int k = tid * EC + r;
            const int64_t s = q * C + k;
            if (k < C && s < P) {
                const uint64_t lag = key[s] ^ kLagKeyFlip;
                const uint64_t t = (((uint64_t)rec[r].hi << 32) | rec[r].lo) + lag;      // Main.java:265
                rec[r].hi = (uint32_t)(t >> 32);
                rec[r].lo = (uint32_t)t;
                a.out_rank[a.p0 + s] = (int32_t)rec[r].tb;                       // consumer index, as above
            }
        }
    }
    if (a.out_total) {
#pragma unroll
        for (int r = 0; r < EC; ++r)
            if (rec[r].tb < (uint32_t)C)
                a.out_total[a.c0 + rec[r].tb] = (int64_t)((((uint64_t)rec[r].hi << 32) | rec[r].lo) ^ kTotalBias);
    }
}

// ---- keys-first sorts: every run of equal keys into id order ------------------------------------------------------------------
// After the key passes of a keys-first sort the partitions are in (lag desc) order and partitions with EQUAL lags stand in
// input order; the comparator wants them in id order (Main.java:228-235).  Windows of kRepairWindow positions, one workgroup
// per window (grid-stride): a window without two equal neighbours -- nearly all of them when lags are wide -- cost its 8 B
// per partition of reading in tie_scan_kernel and is skipped here.  Otherwise the workgroup takes the runs that START in its window, whole, up to kRepairCap
// positions: (index of the run) << 32 | id into registers, the block-wide bitonic network of the greedy's bins (la_sort64.h
// inside a wavefront, LDS exchanges across), ids back in place.  A window writes only the runs that start in it and reads no
// ids but theirs, so windows do not interfere.  A run that reaches beyond the capacity is left alone and raises ctl->redo:
// the sample said there was no such run (or the test hook forced keys first), and the redo slots then sort in full.

// First a plain streaming read of the sorted keys at full occupancy (the repair kernel below keeps 128 KB of LDS per workgroup
// -- one workgroup per CU -- and would read them at a sixth of the bandwidth).  The thread that meets the START of a run of
// equal keys settles it on the spot when the run has at most four members -- with wide lags nearly every tie is a pair: four
// ids into registers, a five-step network, back -- and otherwise marks the window the run starts in for tie_repair_kernel.
__device__ __forceinline__ void tie_run_start(const uint64_t* key, uint32_t* val, int64_t p, int64_t n, uint32_t* tie_flag) {
    const uint64_t k = key[p];
    int len = 2;                                                            // (key[p + 1] == k is why we are here)
    while (len < 5 && p + len < n && key[p + len] == k) ++len;
    if (len > 4) { tie_flag[p / kRepairWindow] = 1; return; }
    uint32_t v0 = val[p], v1 = val[p + 1], v2 = len > 2 ? val[p + 2] : 0xFFFFFFFFu, v3 = len > 3 ? val[p + 3] : 0xFFFFFFFFu;
    auto cx = [](uint32_t& a, uint32_t& b) { const uint32_t lo = a < b ? a : b, hi = a < b ? b : a; a = lo; b = hi; };
    cx(v0, v1); cx(v2, v3); cx(v0, v2); cx(v1, v3); cx(v1, v2);             // (pads are the largest value: they stay behind)
    val[p] = v0; val[p + 1] = v1;
    if (len > 2) val[p + 2] = v2;
    if (len > 3) val[p + 3] = v3;
}

__global__ __launch_bounds__(256) void tie_scan_kernel(SortBufs b0, const LargeItem* items, char* scratch) {
    LargeArgs unused{};
    SortBufs b = b0;
    if (items) bind_item(unused, b, items[blockIdx.y], scratch);
    if (!b.ctl->keys_first) return;
    const uint32_t fin = b.ctl->cur[kDigits];
    const uint64_t* key = key_buf(b, fin);
    uint32_t* val = val_buf(b, fin);
    const int64_t n = b.n, pairs = (n + 1) / 2;
    struct __attribute__((aligned(16))) U64x2 { uint64_t x, y; };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < pairs; j += stride) {
        const int64_t i = 2 * j;                                            // positions i, i + 1 (one 16-byte load) and their neighbours
        uint64_t km = 0, k0, k1 = 0, k2 = 0;
        if (i + 1 < n) { const U64x2 v = *reinterpret_cast<const U64x2*>(key + i); k0 = v.x; k1 = v.y; }
        else k0 = key[i];
        if (i > 0) km = key[i - 1];
        if (i + 2 < n) k2 = key[i + 2];
        if (i + 1 < n && k0 == k1 && (i == 0 || km != k0)) tie_run_start(key, val, i, n, b.tie_flag);
        if (i + 2 < n && k1 == k2 && k0 != k1) tie_run_start(key, val, i + 1, n, b.tie_flag);
    }
}

__global__ __launch_bounds__(kRepairThreads) void tie_repair_kernel(SortBufs b0, const LargeItem* items, char* scratch) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_x[];        // two exchange buffers of kRepairCap words
    __shared__ int s_first, s_last, s_end;
    __shared__ uint32_t s_wsum[kRepairThreads / kWave];
    LargeArgs unused{};
    SortBufs b = b0;
    if (items) bind_item(unused, b, items[blockIdx.y], scratch);
    if (!b.ctl->keys_first) return;
    constexpr int EC = kRepairEC, NT = kRepairThreads, W = kRepairWindow, CAP = kRepairCap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t fin = b.ctl->cur[kDigits];                               // (the data after the sort's own slots)
    const uint64_t* key = key_buf(b, fin);
    uint32_t* val = val_buf(b, fin);
    const int64_t n = b.n;
    for (int64_t w0 = (int64_t)blockIdx.x * W; w0 < n; w0 += (int64_t)gridDim.x * W) {
        const int64_t w1 = w0 + W < n ? w0 + W : n;
        if (b.tie_flag[w0 / W] == 0) continue;                              // (uniform) no run of more than four starts here
        // the first and the last run start inside the window; where the last run ends (searched up to the capacity)
        __syncthreads();                                                    // (the window before this one is done with s_*)
        if (tid == 0) { s_first = 0x7FFFFFFF; s_last = -1; s_end = 0x7FFFFFFF; }
        __syncthreads();
        {
            int lo = 0x7FFFFFFF, hi = -1;
            for (int64_t p = w0 + tid; p < w1; p += NT)
                if (p == 0 || key[p] != key[p - 1]) { const int o = (int)(p - w0); lo = o < lo ? o : lo; hi = o > hi ? o : hi; }
            if (lo != 0x7FFFFFFF) { atomicMin(&s_first, lo); atomicMax(&s_last, hi); }
        }
        __syncthreads();
        if (s_last < 0) continue;                                           // the whole window lies inside an earlier run (uniform)
        const int64_t first = w0 + s_first;
        {
            int e = 0x7FFFFFFF;
            for (int64_t p = w1 + tid; p <= first + CAP && p <= n; p += NT)
                if (p == n || key[p] != key[p - 1]) { const int o = (int)(p - first); e = o < e ? o : e; }
            if (e != 0x7FFFFFFF) atomicMin(&s_end, e);
        }
        __syncthreads();
        int len;
        if (s_end != 0x7FFFFFFF) {
            len = s_end;                                                    // every run that starts in the window, whole
        } else {
            len = (int)(w0 + s_last - first);                               // the last run does not fit: without it,
            if (tid == 0) b.ctl->redo = 1;                                  // and the sort is redone in full
        }
        __syncthreads();                                                    // (s_* are rewritten by the next window)
        if (len <= 1) continue;
        // composites: (index of the run in the region) << 32 | biased id; slots beyond the region sort last
        P64 rec[EC];
        uint32_t flags = 0, own = 0;
        uint64_t prev = first + (int64_t)tid * EC > 0 && tid * EC < len ? key[first + (int64_t)tid * EC - 1] : 0;
        uint32_t v[EC];
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const int i = tid * EC + r;
            const bool valid = i < len;
            const uint64_t k = valid ? key[first + i] : 0;
            v[r] = valid ? val[first + i] : 0;
            const bool start = valid && (i == 0 || k != prev);
            flags |= start ? 1u << r : 0u;
            own += start ? 1u : 0u;
            prev = k;
        }
        uint32_t incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - own;                                          // runs that start before this thread's positions
        for (int x = 0; x < wave; ++x) run += s_wsum[x];
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            run += (flags >> r) & 1u;
            rec[r] = p64_from(tid * EC + r < len ? ((uint64_t)run << 32) | v[r] : ~0ull);
        }
        // the network of greedy_rounds_packed's bins: inside every wavefront, then merges across wavefronts
        constexpr int kSpan = 64 * EC;
        dpp_fence<EC>(rec);
        bitonic_sort_tile_p64<64, EC>(rec);
        ExchangeBufs xb{s_x, CAP};
        for (int K = 2 * kSpan; K <= CAP; K <<= 1) {
            cross_wave_step<EC>(rec, xb, tid, K - 1, K >> 1);
            for (int j = K >> 2; j >= kSpan; j >>= 1) cross_wave_step<EC>(rec, xb, tid, j, j);
            dpp_fence<EC>(rec);
            clean_p64<64, EC, kSpan / 2, false>(rec);
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < EC; ++r) {
            const int i = tid * EC + r;
            if (i < len) val[first + i] = (uint32_t)p64_value(rec[r]);
        }
        __syncthreads();
    }
}

// ---- kernel 3, literal form: bins in LDS, wavefront argmin per partition ------------------------------
__global__ __launch_bounds__(1024) void greedy_argmin_kernel(LargeArgs a, SortBufs b) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int C = (int)a.n_cons;
    uint32_t* s_cnt = smem;                 // assigned count per bin
    uint32_t* s_hi = smem + C;              // biased assigned lag, high / low dword
    uint32_t* s_lo = smem + 2 * C;
    uint32_t* w_best = smem + 3 * C;        // [16 waves][4]
    const uint64_t* key = b.key[b.ctl->cur[kFinal]];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int i = tid; i < C; i += blockDim.x) { s_cnt[i] = 0; s_hi[i] = (uint32_t)(kTotalBias >> 32); s_lo[i] = 0; }
    __syncthreads();
    for (int64_t s = 0; s < a.n_part; ++s) {
        uint32_t bc = 0xFFFFFFFFu;
        Rec best; best.hi = best.lo = best.tb = 0xFFFFFFFFu;
        for (int i = tid; i < C; i += blockDim.x) {          // comparator of Main.java:243-261
            Rec c; c.hi = s_hi[i]; c.lo = s_lo[i]; c.tb = (uint32_t)i;
            const uint32_t cc = s_cnt[i];
            const bool take = (cc < bc) | ((cc == bc) & rec_less(c, best));
            bc = take ? cc : bc; best.hi = take ? c.hi : best.hi; best.lo = take ? c.lo : best.lo;
            best.tb = take ? c.tb : best.tb;
        }
        for (int j = 1; j < 64; j <<= 1) {                   // wavefront argmin
            const Rec o = shfl_xor_dyn(best, j);
            const uint32_t oc = (uint32_t)__shfl_xor((int)bc, j);
            const bool take = (oc < bc) | ((oc == bc) & rec_less(o, best));
            bc = take ? oc : bc; best.hi = take ? o.hi : best.hi; best.lo = take ? o.lo : best.lo;
            best.tb = take ? o.tb : best.tb;
        }
        if (lane == 0) { w_best[wave * 4] = bc; w_best[wave * 4 + 1] = best.hi; w_best[wave * 4 + 2] = best.lo; w_best[wave * 4 + 3] = best.tb; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < nw; ++w) {
                Rec o; o.hi = w_best[w * 4 + 1]; o.lo = w_best[w * 4 + 2]; o.tb = w_best[w * 4 + 3];
                const uint32_t oc = w_best[w * 4];
                const bool take = (oc < bc) | ((oc == bc) & rec_less(o, best));
                if (take) { bc = oc; best = o; }
            }
            const uint32_t m = best.tb;
            const uint64_t t = (((uint64_t)s_hi[m] << 32) | s_lo[m]) + (key[s] ^ kLagKeyFlip);
            s_hi[m] = (uint32_t)(t >> 32);
            s_lo[m] = (uint32_t)t;
            s_cnt[m] += 1;
            a.out_rank[a.p0 + s] = a.cons_rank[a.c0 + m];
        }
        __syncthreads();
    }
    if (a.out_total)
        for (int i = tid; i < C; i += blockDim.x)
            a.out_total[a.c0 + i] = (int64_t)((((uint64_t)s_hi[i] << 32) | s_lo[i]) ^ kTotalBias);
}

// The rounds kernels store the chosen consumer's INDEX (position in the topic's rank-sorted list): looking the rank up
// there would put a dependent global load into every round of the one-workgroup chain (~6 us of 27 per round at
// 8 192 consumers).  This pass, over all CUs, turns the indices into member ranks.
__global__ __launch_bounds__(256) void map_ranks_kernel(LargeArgs a0, SortBufs b0, const LargeItem* items, char* scratch) {
    LA_PICK_ITEM(a, b, a0, b0, items, scratch, blockIdx.y)
    if (a.n_cons == 0) return;                                         // emit_ids_kernel wrote -1: no index to map
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint32_t fin = b.ctl->cur[kFinal];
    const RoundsIo io = rounds_io(a, key_buf(b, fin), a.n_part, a.n_cons, true);
    if (io.narrow) {
        // the rounds kernel left 16-bit consumer indices, every round's stretch on a multiple of 8 (rounds_io)
        const uint16_t* idx16 = reinterpret_cast<const uint16_t*>(val_buf(b, fin ^ 1u));
        const uint32_t C32 = (uint32_t)a.n_cons;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_part; i += stride) {
            const uint32_t q = (uint32_t)i / C32;
            a.out_rank[a.p0 + i] = a.cons_rank[a.c0 + idx16[(int64_t)q * io.cpad + ((uint32_t)i - q * C32)]];
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_part; i += stride)
        a.out_rank[a.p0 + i] = a.cons_rank[a.c0 + a.out_rank[a.p0 + i]];
}

// `items` null: ONE topic (a, b).  Otherwise `count` topics of this (EC, threads) class, one workgroup each, side by side:
// the chain of rounds is serial inside a topic and independent across topics (Main.java:177-184).
template <int EC>
hipError_t launch_rounds(const LargeArgs& a, const SortBufs& b, int threads, hipStream_t stream, const LargeItem* items = nullptr,
                         char* scratch = nullptr, const int32_t* order = nullptr, int count = 1) {
    // packed bins: two exchange buffers of 8 B per bin; 96-bit bins: 12 B per bin; sample sort (EC >= 2): its own layout
    size_t lds = (size_t)4 * EC * threads * sizeof(uint32_t);
    if (EC >= 2 && sample_lds_bytes(EC) > lds) lds = sample_lds_bytes(EC);
    static PerDeviceOnce lds_opt_in;
    const hipError_t e = lds_opt_in.run([] {
        return hipFuncSetAttribute((const void*)greedy_rounds_kernel<EC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024);
    });
    if (e != hipSuccess) return e;
    LA_LAUNCH((greedy_rounds_kernel<EC>), dim3(count), dim3(threads), lds, stream, a, b, items, scratch, order);
    if (!items) {
        int grid = (int)((a.n_part + 255) / 256);
        LA_LAUNCH(map_ranks_kernel, dim3(grid > 2048 ? 2048 : grid), dim3(256), 0, stream, a, b, (const LargeItem*)nullptr, (char*)nullptr);
    }
    return hipGetLastError();
}

// (EC, threads) of the rounds kernel for a topic with n_cons consumers (1 <= n_cons <= kLargeMaxConsumers)
__host__ __device__ inline void rounds_class(int64_t n_cons, int* ec, int* threads) {
    int cp2 = 64;
    while (cp2 < n_cons) cp2 <<= 1;
    if (cp2 <= 1024) { *ec = 1; *threads = cp2; }
    else { *ec = cp2 / 1024 >= 8 ? 8 : cp2 / 1024; *threads = 1024; }
}

inline hipError_t launch_rounds_class(int ec, int threads, const LargeArgs& a, const SortBufs& b, hipStream_t stream,
                                      const LargeItem* items = nullptr, char* scratch = nullptr, const int32_t* order = nullptr,
                                      int count = 1) {
    switch (ec) {
        case 1: return launch_rounds<1>(a, b, threads, stream, items, scratch, order, count);
        case 2: return launch_rounds<2>(a, b, threads, stream, items, scratch, order, count);
        case 4: return launch_rounds<4>(a, b, threads, stream, items, scratch, order, count);
        default: return launch_rounds<8>(a, b, threads, stream, items, scratch, order, count);
    }
}


// ---- assignment -> per-member lists (the wrap step of Main.java:152-156, 171-174, 264) -------
// The reference appends to each member's list topic by topic, and inside a topic in assignment order: that is
// the order of the output arrays.  Grouping by member is therefore a STABLE sort of the entry indices by member
// rank: key = rank + 1 (0 = "topic had no consumers": those entries come first and are dropped), payload =
// entry index, already ascending, so only the key's ceil(log2(M+1)/8) digit passes run.
__global__ __launch_bounds__(256) void member_keys_kernel(const int32_t* member_rank, SortBufs b) {
    __shared__ uint32_t h[kDigits * kRadix];
    for (int i = threadIdx.x; i < kDigits * kRadix; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < b.n; base += stride) {
        const int64_t i = base + threadIdx.x;
        if (i < b.n) {
            const uint64_t key = (uint64_t)(uint32_t)(member_rank[i] + 1);
            b.key[0][i] = key;
            b.val[0][i] = (uint32_t)i;
#pragma unroll
            for (int p = 4; p < 8; ++p) atomicAdd(&h[p * kRadix + ((uint32_t)(key >> (8 * (p - 4))) & 0xFFu)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kDigits * kRadix; i += blockDim.x)
        if (h[i]) atomicAdd(&b.hist[i], h[i]);
    // digits the loop did not count: ids are ascending (skipped by the plan), key bits 32..63 are zero
    if (blockIdx.x == 0 && threadIdx.x < kDigits && (threadIdx.x < 4 || threadIdx.x >= 8))
        b.hist[threadIdx.x * kRadix] = (uint32_t)b.n;
}

// member_off[r] = first grouped position of member r (r = 0..M); grouped_* = the entries in grouped order
__global__ __launch_bounds__(256) void member_emit_kernel(SortBufs b, int32_t n_members, int64_t n_topics,
                                                          const int64_t* part_off, const int32_t* out_partition,
                                                          int64_t* member_off, int32_t* grouped_topic,
                                                          int32_t* grouped_partition, int32_t* grouped_entry, uint32_t* status) {
    const uint32_t fin = b.ctl->cur[kFinal];
    const uint64_t* key = b.key[fin];
    const uint32_t* val = b.val[fin];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= b.n; i += stride) {
        // boundaries: key k (= rank + 1) starts at the first i with key[i] >= k
        const int64_t k_prev = i > 0 ? (int64_t)key[i - 1] : 0;
        const int64_t k_here = i < b.n ? (int64_t)key[i] : (int64_t)n_members + 1;
        // the sort's result, checked on the way (see emit_ids_kernel): groups ascending, entry indices ascending inside a group
        if (i > 0 && i < b.n && status && (k_prev > k_here || (k_prev == k_here && val[i - 1] >= val[i])))
            atomicOr(status, kStatusOrder);
        for (int64_t k = k_prev + 1; k <= k_here; ++k)
            if (k >= 1 && k <= (int64_t)n_members + 1) member_off[k - 1] = i;
        if (i < b.n) {
            const uint32_t e = val[i];
            if (grouped_entry) grouped_entry[i] = (int32_t)e;
            if (grouped_partition) grouped_partition[i] = out_partition[e];
            if (grouped_topic) {
                int64_t lo = 0, hi = n_topics;                   // largest t with part_off[t] <= e
                while (hi - lo > 1) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (part_off[mid] <= (int64_t)e) lo = mid; else hi = mid;
                }
                grouped_topic[i] = (int32_t)lo;
            }
        }
    }
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

// Per device, once (la_create_multi; synchronous): are the pre-values of colliding lanes of a returning LDS atomic in lane
// order (rank_in_wave)?  The launchers rank with the match form where they are not (or the check could not run).
static std::atomic<int> g_atomic_rank_ok[32];       // 0 = not checked, 1 = no, 2 = yes

hipError_t large_init_device() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 32 || g_atomic_rank_ok[dev].load(std::memory_order_acquire) != 0) return hipSuccess;
    int verdict = 1;
    const char* env = getenv("LA_SORT_RANK");                  // "match": never the atomic form (A/B, tests)
    if (!(env && env[0] == 'm')) {
        uint32_t* d_bad = nullptr;
        uint32_t h_bad = 1;
        if ((e = hipMalloc((void**)&d_bad, sizeof(uint32_t))) != hipSuccess) return e;
        if ((e = hipMemset(d_bad, 0, sizeof(uint32_t))) == hipSuccess) {
            LA_LAUNCH(lds_atomic_order_test_kernel, dim3(8), dim3(256), 0, nullptr, d_bad);
            e = hipMemcpy(&h_bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost);
        }
        (void)hipFree(d_bad);
        if (e != hipSuccess) return e;
        verdict = h_bad == 0 ? 2 : 1;
    }
    g_atomic_rank_ok[dev].store(verdict, std::memory_order_release);
    return hipSuccess;
}

int large_atomic_rank_supported() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 0;
    return g_atomic_rank_ok[dev].load(std::memory_order_acquire) == 2 ? 1 : 0;
}

// The sort's working set for n elements, in two parts: what a sort expects zeroed (ctl, hist and -- single-kernel passes --
// the arrival tickets and the look-back granules) and the rest.  Several topics sorted side by side put all their zero parts
// first, so that ONE memset clears them.
// `multi_kernel`: the four-kernel passes (count, scans, scatter) instead of the single-kernel ones; also taken when a
// digit count would not fit a granule's 32-bit count with room for the tag arithmetic (n >= 2^30).
struct SortLayout {
    int64_t n = 0;
    bool multi_kernel = false;
    int sweep_threads = 256, n_tiles = 0, n_groups = 0;
    size_t o_ctl = 0, o_hist = 0, o_ticket = 0, o_state = 0, o_samp = 0, o_flag = 0, zero_bytes = 0;  // offsets into the zero part
    int keys_first = 0;                                                                             // 0 never, 1 by the sample, 2 forced
    uint32_t samp_bits = 0;
    size_t o_gbase = 0, o_k0 = 0, o_k1 = 0, o_v0 = 0, o_v1 = 0, o_to = 0, o_gs = 0, data_bytes = 0;   // ... into the data part
};

// Keys-first sorts (plan_kernel): worth their extra launches from a few million partitions on; LA_SORT_KEYS_FIRST=0 never,
// =2 always and whatever the sample says (test hook: small topics, long runs through the redo slots).
static int keys_first_mode(int64_t n, bool multi_kernel) {
    int mode = 1;
    if (const char* env = getenv("LA_SORT_KEYS_FIRST")) mode = atoi(env);
    if (multi_kernel || mode <= 0) return 0;
    if (mode >= 2) return 2;
    return n >= ((int64_t)1 << 22) ? 1 : 0;
}

static SortLayout sort_layout(int64_t n, bool multi_kernel, bool may_sort_keys_first = false) {
    SortLayout L;
    L.n = n;
    if (n >= ((int64_t)1 << 30)) multi_kernel = true;
    if (const char* env = getenv("LA_SORT_MULTIKERNEL")) multi_kernel = multi_kernel || atoi(env) != 0;
    L.multi_kernel = multi_kernel;
    // Tiles of 16 elements per thread: the larger the workgroup, the fewer tiles publish and walk (the look-back costs a
    // tile ~10 us whatever its size) and the longer the runs a tile writes per digit -- 33.5 M partitions: 2.25 / 2.02 / 1.93 ms
    // with 256 / 512 / 1 024 threads on the same box, four-kernel passes 2.56; but a 1 M-partition topic (cfg5) has only 64
    // tiles of 16 384: 0.136 / 0.125 / 0.144 ms (profiles/archive/r03_sort_tile_sizes.txt)
    int sweep_threads = n >= (3 << 20) ? 1024 : (n >= (1 << 17) ? 512 : 256);
    if (const char* env = getenv("LA_SWEEP_THREADS")) { const int w = atoi(env); sweep_threads = w == 256 || w == 1024 ? w : 512; }
    L.sweep_threads = sweep_threads;
    const int64_t tile = multi_kernel ? kTile : (int64_t)sweep_threads * kItems;
    L.n_tiles = (int)((n + tile - 1) / tile);
    L.n_groups = (L.n_tiles + kScanRows - 1) / kScanRows;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.o_ctl = carve(sizeof(SortCtl));
    L.o_hist = carve(sizeof(uint32_t) * kDigits * kRadix);
    L.o_ticket = carve(sizeof(uint32_t) * kSlots);
    L.o_state = carve(multi_kernel ? 0 : sizeof(unsigned long long) * kRadix * (size_t)L.n_tiles);
    L.keys_first = may_sort_keys_first ? keys_first_mode(n, multi_kernel) : 0;
    if (L.keys_first) {
        L.samp_bits = 10;                                           // n / 32 counters for n / 256 samples (at least 1 024)
        while (((int64_t)1 << L.samp_bits) < n / 32 && L.samp_bits < 28) ++L.samp_bits;
        L.o_samp = carve(sizeof(uint32_t) << L.samp_bits);
        L.o_flag = carve(sizeof(uint32_t) * (size_t)((n + kRepairWindow - 1) / kRepairWindow));
    }
    L.zero_bytes = off;
    off = 0;
    L.o_gbase = carve(sizeof(uint32_t) * kDigits * kRadix);
    L.o_k0 = carve(sizeof(uint64_t) * n); L.o_k1 = carve(sizeof(uint64_t) * n);
    L.o_v0 = carve(sizeof(uint32_t) * n); L.o_v1 = carve(sizeof(uint32_t) * n);
    L.o_to = carve(multi_kernel ? sizeof(uint32_t) * kRadix * (size_t)L.n_tiles : 0);
    L.o_gs = carve(multi_kernel ? sizeof(uint32_t) * kRadix * (size_t)L.n_groups : 0);
    L.data_bytes = off;
    return L;
}

static SortBufs sort_bind(const SortLayout& L, char* zero, char* data) {
    SortBufs b{};
    b.ctl = (SortCtl*)(zero + L.o_ctl);
    b.hist = (uint32_t*)(zero + L.o_hist);
    b.ticket = (uint32_t*)(zero + L.o_ticket);
    b.tile_state = L.multi_kernel ? nullptr : (unsigned long long*)(zero + L.o_state);
    b.gbase = (uint32_t*)(data + L.o_gbase);
    b.key[0] = (uint64_t*)(data + L.o_k0);
    b.key[1] = (uint64_t*)(data + L.o_k1);
    b.val[0] = (uint32_t*)(data + L.o_v0);
    b.val[1] = (uint32_t*)(data + L.o_v1);
    b.tile_off = (uint32_t*)(data + L.o_to);
    b.group_sum = (uint32_t*)(data + L.o_gs);
    b.n = L.n;
    b.n_tiles = L.n_tiles;
    b.n_groups = L.n_groups;
    b.atomic_rank = large_atomic_rank_supported();
    b.sweep_threads = L.sweep_threads;
    b.samp = L.keys_first ? (uint32_t*)(zero + L.o_samp) : nullptr;
    b.tie_flag = (uint32_t*)(zero + L.o_flag);
    b.samp_bits = L.samp_bits;
    b.keys_first_force = L.keys_first == 2 ? 1 : 0;
    return b;
}

// grow-only; earlier work on `stream` may still use the old block
static hipError_t scratch_reserve(LargeScratch& scratch, size_t bytes, hipStream_t stream) {
    if (bytes <= scratch.cap) return hipSuccess;
    hipError_t e;
    if (scratch.buf) {
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        if ((e = hipFree(scratch.buf)) != hipSuccess) return e;
        scratch.buf = nullptr;
        scratch.cap = 0;
    }
    const size_t want = bytes + bytes / 4;
    if ((e = hipMalloc(&scratch.buf, want)) != hipSuccess) return e;
    scratch.cap = want;
    return hipSuccess;
}

static hipError_t sort_prepare(LargeScratch& scratch, int64_t n, hipStream_t stream, SortBufs* out, bool multi_kernel = false,
                               bool may_sort_keys_first = false) {
    const SortLayout L = sort_layout(n, multi_kernel, may_sort_keys_first);
    hipError_t e;
    if ((e = scratch_reserve(scratch, L.zero_bytes + L.data_bytes, stream)) != hipSuccess) return e;
    char* base = (char*)scratch.buf;
    *out = sort_bind(L, base, base + L.zero_bytes);
    return hipMemsetAsync(base, 0, L.zero_bytes, stream);
}

// The passes a caller's bounds (LA_FLAG_BOUNDS: 0 <= lag <= max_lag, 0 <= id <= max_id) leave possible: a key digit above the
// lag's bits holds the same value in every record (key = lag ^ 0x7FFF..., so the bits above are all ones), an id digit above the
// id's bits likewise (val = id ^ 0x80000000) -- the device-side plan would find them constant and skip them, but each skipped
// pass is still a launch (3-4.5 us apiece: cfg5 launches 6 such, a 33.5 M-partition keys-first sort 3 + 3 redo slots).
// build_keys_kernel raises kStatusBounds on a partition outside the bounds.
static uint32_t bounds_pass_mask(const LargeArgs& a) {
    uint32_t mask = (1u << kDigits) - 1;
    if (a.max_lag_hint < 0 || a.max_id_hint < 0) return mask;
    const int lag_bits = a.max_lag_hint > 0 ? 64 - __builtin_clzll((unsigned long long)a.max_lag_hint) : 0;
    const int id_bits = a.max_id_hint > 0 ? 64 - __builtin_clzll((unsigned long long)a.max_id_hint) : 0;
    mask = 0;
    for (int d = 0; d < 4; ++d) if (8 * d < id_bits) mask |= 1u << d;
    for (int d = 0; d < 8; ++d) if (8 * d < lag_bits) mask |= 1u << (4 + d);
    return mask;
}

// plan + the 12 (mostly skipped) passes; keys/vals/hist/unsorted flag must already be in buffer 0
// `pass_mask`: passes the HOST knows can matter (bit p = digit p); the kernels of the others are not even launched.  The
// device-side plan still decides among the launched ones (a launched pass whose digit turns out constant returns at
// once), and it marks every pass outside the mask as skipped by itself -- the caller guarantees those digits are
// constant (key bits that cannot be set) or the ids ascending.
// A pass is ONE launch (onesweep_pass_kernel) -- or four, when the sort was prepared for the multi-kernel form.
static void sort_run_passes(const SortBufs& b, hipStream_t stream, uint32_t* status, hipEvent_t planned = nullptr,
                            uint32_t pass_mask = (1u << kDigits) - 1, const LargeItem* items = nullptr, int count = 1,
                            int max_tiles = 0, char* scratch = nullptr, int slot0 = 0) {
    // `items`: `count` topics of ONE tile class (b.sweep_threads), sorted side by side -- grid.y picks the topic, grid.x is as
    // wide as the class's largest topic (max_tiles); the plan of all of them is the caller's (one launch over every class).
    // slot0 = kDigits: the redo slots of a keys-first sort (same digits, see tie_repair_kernel).
    if (!items && slot0 == 0)
        LA_LAUNCH(plan_kernel, dim3(kDigits), dim3(kRadix), 0, stream, b, (const LargeItem*)nullptr, (char*)nullptr);
    if (planned) (void)hipEventRecord(planned, stream);
    if (slot0 == kDigits && b.tile_state) {
        // the redo slots of a keys-first sort: one launch (onesweep_redo_kernel), a grid that is resident at once -- at most 64
        // workgroups over all the launch's topics (a workgroup of 1 024 threads keeps 130 KB of LDS: one per CU; the lanes of a
        // host-buffer call may run up to four such launches side by side on their streams: 4 x 64 = the chip's 256 CUs, so
        // every launch's workgroups become resident whatever the others do -- a workgroup waiting for a CU behind spinning
        // ones of ANOTHER launch would otherwise be a deadlock in the making)
        int gx = items ? max_tiles : b.n_tiles;
        const int per_topic = 64 / (items ? (count < 64 ? count : 64) : 1);
        if (gx > per_topic) gx = per_topic < 1 ? 1 : per_topic;
        const dim3 grid(gx, items ? count : 1);
        if (b.sweep_threads == 1024) {
            if (b.atomic_rank) LA_LAUNCH((onesweep_redo_kernel<true, 1024>), grid, dim3(1024), 0, stream, b, pass_mask, status, items, scratch);
            else LA_LAUNCH((onesweep_redo_kernel<false, 1024>), grid, dim3(1024), 0, stream, b, pass_mask, status, items, scratch);
        } else if (b.sweep_threads == 512) {
            if (b.atomic_rank) LA_LAUNCH((onesweep_redo_kernel<true, 512>), grid, dim3(512), 0, stream, b, pass_mask, status, items, scratch);
            else LA_LAUNCH((onesweep_redo_kernel<false, 512>), grid, dim3(512), 0, stream, b, pass_mask, status, items, scratch);
        } else {
            if (b.atomic_rank) LA_LAUNCH((onesweep_redo_kernel<true, 256>), grid, dim3(256), 0, stream, b, pass_mask, status, items, scratch);
            else LA_LAUNCH((onesweep_redo_kernel<false, 256>), grid, dim3(256), 0, stream, b, pass_mask, status, items, scratch);
        }
        return;
    }
    for (int d = 0; d < kDigits; ++d) {
        if (!((pass_mask >> d) & 1u)) continue;
        const int p = slot0 + d;
        if (b.tile_state) {
            const dim3 grid(items ? max_tiles : b.n_tiles, items ? count : 1);
            if (b.sweep_threads == 1024) {
                if (b.atomic_rank) LA_LAUNCH((onesweep_pass_kernel<true, 1024>), grid, dim3(1024), 0, stream, b, p, status, items, scratch);
                else LA_LAUNCH((onesweep_pass_kernel<false, 1024>), grid, dim3(1024), 0, stream, b, p, status, items, scratch);
            } else if (b.sweep_threads == 512) {
                if (b.atomic_rank) LA_LAUNCH((onesweep_pass_kernel<true, 512>), grid, dim3(512), 0, stream, b, p, status, items, scratch);
                else LA_LAUNCH((onesweep_pass_kernel<false, 512>), grid, dim3(512), 0, stream, b, p, status, items, scratch);
            } else {
                if (b.atomic_rank) LA_LAUNCH((onesweep_pass_kernel<true, 256>), grid, dim3(256), 0, stream, b, p, status, items, scratch);
                else LA_LAUNCH((onesweep_pass_kernel<false, 256>), grid, dim3(256), 0, stream, b, p, status, items, scratch);
            }
            continue;
        }
        LA_LAUNCH(tile_count_kernel, dim3(b.n_tiles < 2048 ? b.n_tiles : 2048), dim3(kSortThreads), 0, stream, b, p);
        LA_LAUNCH(scan_group_sums_kernel, dim3(b.n_groups), dim3(kRadix), 0, stream, b, p);
        LA_LAUNCH(scan_offsets_kernel, dim3(b.n_groups), dim3(kRadix), 0, stream, b, p);
        if (b.atomic_rank) LA_LAUNCH(tile_scatter_kernel<true>, dim3(b.n_tiles), dim3(kSortThreads), 0, stream, b, p);
        else LA_LAUNCH(tile_scatter_kernel<false>, dim3(b.n_tiles), dim3(kSortThreads), 0, stream, b, p);
    }
}

// The tail of sorts that may have gone keys first (b.samp of the single form; any item of a batched launch): ties into id
// order, then -- no-ops unless a run did not fit -- the redo slots.  `max_n`: partitions of the largest topic.
static hipError_t sort_repair_launch(const SortBufs& b, hipStream_t stream, const LargeItem* items, int count, char* scratch,
                                     int64_t max_n) {
    static PerDeviceOnce lds_opt_in;
    const hipError_t e = lds_opt_in.run([] {
        // (exactly what the launch asks for: the kernel also has static LDS, and dynamic + static must fit the CU's 160 KB)
        return hipFuncSetAttribute((const void*)tie_repair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(2 * kRepairCap * sizeof(uint64_t)));
    });
    if (e != hipSuccess) return e;
    int64_t gs = (max_n / 2 + 255) / 256;
    if (gs > 4096) gs = 4096;
    LA_LAUNCH(tie_scan_kernel, dim3((unsigned)(gs < 1 ? 1 : gs), items ? count : 1), dim3(256), 0, stream, b, items, scratch);
    int64_t gx = (max_n + kRepairWindow - 1) / kRepairWindow;
    if (gx > 1024) gx = 1024;                                   // (128 KB of LDS each: one per CU at a time; windows grid-stride)
    LA_LAUNCH(tie_repair_kernel, dim3((unsigned)gx, items ? count : 1), dim3(kRepairThreads),
                       (size_t)2 * kRepairCap * sizeof(uint64_t), stream, b, items, scratch);
    LA_LAUNCH(replan_kernel, dim3(items ? count : 1), dim3(64), 0, stream, b, items, scratch);
    return hipGetLastError();
}

hipError_t large_topic_launch(LargeScratch& scratch, const LargeArgs& a, bool argmin, hipStream_t stream) {
    const int64_t n = a.n_part;
    if (n <= 0) {
        // no partition metadata: every subscribed consumer still reports a total of 0 (Main.java:216-225, :283-291)
        if (a.n_cons > 0 && a.out_total)
            return hipMemsetAsync(a.out_total + a.c0, 0, (size_t)a.n_cons * sizeof(int64_t), stream);
        return hipSuccess;
    }
    if (n > 0x7FFFFFFF || a.n_cons > kLargeMaxConsumers) return hipErrorInvalidValue;
    SortBufs b{};
    hipError_t e;
    if ((e = sort_prepare(scratch, n, stream, &b, a.sort_multi_kernel != 0, true)) != hipSuccess) return e;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    LargeProfile& pf = scratch.prof;
    const bool profile = pf.armed && !pf.recorded;
    if (profile) {
        for (hipEvent_t& ev : pf.ev)
            if (!ev && (e = hipEventCreate(&ev)) != hipSuccess) return e;
        pf.n = n;
        pf.recorded = true;
        (void)hipEventRecord(pf.ev[0], stream);
    }
    struct Done {                                   // the last event, on every way out
        LargeProfile& pf; bool on; hipStream_t st;
        ~Done() { if (on) (void)hipEventRecord(pf.ev[3], st); }
    } done{pf, profile, stream};
    LA_LAUNCH(build_keys_kernel, dim3(keys_grid(n)), dim3(256), 0, stream, a, b, (const LargeItem*)nullptr, (char*)nullptr);
    if (b.samp) LA_LAUNCH(sample_scan_kernel, dim3(256), dim3(256), 0, stream, b, (const LargeItem*)nullptr, (char*)nullptr);
    const uint32_t pass_mask = bounds_pass_mask(a);
    sort_run_passes(b, stream, a.status, profile ? pf.ev[1] : nullptr, pass_mask);
    if (b.samp) {
        if ((e = sort_repair_launch(b, stream, nullptr, 1, nullptr, n)) != hipSuccess) return e;
        sort_run_passes(b, stream, a.status, nullptr, pass_mask, nullptr, 1, 0, nullptr, kDigits);
    }
    if (profile) (void)hipEventRecord(pf.ev[2], stream);
    LargeArgs al = a;
    al.rounds_follow = argmin ? 0 : 1;
    LA_LAUNCH(emit_ids_kernel, dim3(grid), dim3(256), 0, stream, al, b, (const LargeItem*)nullptr, (char*)nullptr);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (a.n_cons == 0) return hipSuccess;

    if (argmin) {
        const int C = (int)a.n_cons;
        int threads = 64;
        while (threads < C && threads < 1024) threads <<= 1;
        const size_t lds = sizeof(uint32_t) * ((size_t)3 * C + 16 * 4);
        static PerDeviceOnce lds_opt_in;
        if ((e = lds_opt_in.run([] {
                 return hipFuncSetAttribute((const void*)greedy_argmin_kernel,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
             })) != hipSuccess)
            return e;
        LA_LAUNCH(greedy_argmin_kernel, dim3(1), dim3(threads), lds, stream, a, b);
        return hipGetLastError();
    }
    int ec = 1, threads = 64;
    rounds_class(a.n_cons, &ec, &threads);
    return launch_rounds_class(ec, threads, al, b, stream);
}

hipError_t large_topics_launch(LargeScratch& scratch, const LargeArgs* args, int count, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipError_t e;
    std::vector<SortLayout> lay((size_t)count);
    bool serial = count == 1;
    for (int i = 0; i < count; ++i) {
        if (args[i].n_part <= 0 || args[i].n_part > 0x7FFFFFFF || args[i].n_cons > kLargeMaxConsumers) return hipErrorInvalidValue;
        lay[(size_t)i] = sort_layout(args[i].n_part, args[i].sort_multi_kernel != 0, true);
        serial = serial || lay[(size_t)i].multi_kernel;
    }
    if (serial) {
        for (int i = 0; i < count; ++i)
            if ((e = large_topic_launch(scratch, args[i], false, stream)) != hipSuccess) return e;
        return hipSuccess;
    }
    // items in tile-class order (every class a contiguous range for the pass launches; stable: the call's first large topic
    // of the first class sits at offset 0, where large_profile_read looks for a control block)
    std::vector<int> idx((size_t)count);
    for (int i = 0; i < count; ++i) idx[(size_t)i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return lay[(size_t)x].sweep_threads < lay[(size_t)y].sweep_threads; });
    size_t zero = 0, data = 0;
    std::vector<size_t> zoff((size_t)count), doff((size_t)count);
    for (int j = 0; j < count; ++j) {
        const SortLayout& L = lay[(size_t)idx[(size_t)j]];
        zoff[(size_t)j] = zero; zero += L.zero_bytes;
        doff[(size_t)j] = data; data += L.data_bytes;
    }
    if ((e = scratch_reserve(scratch, zero + data, stream)) != hipSuccess) return e;
    char* base = (char*)scratch.buf;

    // the argument blocks + the greedy's order list (topics grouped by the (EC, threads) class of their rounds kernel)
    const size_t items_bytes = align_up(sizeof(LargeItem) * (size_t)count, 256), bytes = items_bytes + sizeof(int32_t) * (size_t)count;
    LargeScratch::Stage& sg = scratch.stage[scratch.stage_next++ & 1u];
    if (sg.done) { if ((e = hipEventSynchronize(sg.done)) != hipSuccess) return e; }
    else if ((e = hipEventCreateWithFlags(&sg.done, hipEventDisableTiming)) != hipSuccess) return e;
    if (sg.cap < bytes) {
        if (sg.h) { (void)hipHostFree(sg.h); sg.h = nullptr; sg.cap = 0; }
        if ((e = hipHostMalloc(&sg.h, bytes + bytes / 2, hipHostMallocDefault)) != hipSuccess) return e;
        sg.cap = bytes + bytes / 2;
    }
    if (scratch.d_items_cap < bytes) {
        if (scratch.d_items) {
            if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
            (void)hipFree(scratch.d_items);
            scratch.d_items = nullptr; scratch.d_items_cap = 0;
        }
        if ((e = hipMalloc(&scratch.d_items, bytes + bytes / 2)) != hipSuccess) return e;
        scratch.d_items_cap = bytes + bytes / 2;
    }
    LargeItem* h_items = (LargeItem*)sg.h;
    int32_t* h_order = (int32_t*)((char*)sg.h + items_bytes);
    int64_t max_n = 0;
    bool any_keys_first = false, force_keys_first = false;
    for (int j = 0; j < count; ++j) {
        const int i = idx[(size_t)j];
        const SortLayout& L = lay[(size_t)i];
        LargeItem& it = h_items[j];
        it.p0 = args[i].p0; it.n_part = args[i].n_part; it.c0 = args[i].c0; it.n_cons = args[i].n_cons;
        it.n = L.n; it.n_tiles = L.n_tiles; it.n_groups = L.n_groups;
        const uint64_t z = zoff[(size_t)j], d = zero + doff[(size_t)j];
        it.o_ctl = z + L.o_ctl; it.o_hist = z + L.o_hist; it.o_ticket = z + L.o_ticket; it.o_state = z + L.o_state;
        it.o_gbase = d + L.o_gbase; it.o_k0 = d + L.o_k0; it.o_k1 = d + L.o_k1; it.o_v0 = d + L.o_v0; it.o_v1 = d + L.o_v1;
        it.o_samp = L.keys_first ? z + L.o_samp : 0;               // (z + o_samp > 0: the control block comes first)
        it.o_flag = z + L.o_flag;
        it.samp_bits = L.samp_bits; it.pad = 0;
        any_keys_first = any_keys_first || L.keys_first != 0;
        force_keys_first = force_keys_first || L.keys_first == 2;
        if (args[i].n_part > max_n) max_n = args[i].n_part;
    }
    struct Cls { int ec, threads, first, n; };
    std::vector<Cls> cls;
    {
        std::vector<int> with;                                     // items that have consumers, by (EC, threads)
        for (int j = 0; j < count; ++j) if (h_items[j].n_cons > 0) with.push_back(j);
        auto key = [&](int j) { int ec, th; rounds_class(h_items[j].n_cons, &ec, &th); return ec * 2048 + th; };
        std::stable_sort(with.begin(), with.end(), [&](int x, int y) { return key(x) < key(y); });
        for (size_t k = 0; k < with.size(); ++k) {
            h_order[k] = with[k];
            int ec, th;
            rounds_class(h_items[with[k]].n_cons, &ec, &th);
            if (cls.empty() || cls.back().ec != ec || cls.back().threads != th) cls.push_back({ec, th, (int)k, 0});
            ++cls.back().n;
        }
    }
    if ((e = hipMemcpyAsync(scratch.d_items, sg.h, bytes, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    (void)hipEventRecord(sg.done, stream);
    const LargeItem* d_items = (const LargeItem*)scratch.d_items;
    const int32_t* d_order = (const int32_t*)((const char*)scratch.d_items + items_bytes);
    if ((e = hipMemsetAsync(base, 0, zero, stream)) != hipSuccess) return e;

    LargeProfile& pf = scratch.prof;
    const bool profile = pf.armed && !pf.recorded;
    if (profile) {
        for (hipEvent_t& ev : pf.ev)
            if (!ev && (e = hipEventCreate(&ev)) != hipSuccess) return e;
        pf.n = h_items[0].n_part;
        pf.recorded = true;
        (void)hipEventRecord(pf.ev[0], stream);
    }
    const int gx = keys_grid(max_n);
    // a0: the batch's arrays and flags, shared by every item (an item overrides the four segment fields); b0: what is common
    // to a launch's items besides their buffers
    LargeArgs a0 = args[0];
    SortBufs b0{};
    b0.atomic_rank = large_atomic_rank_supported();
    b0.keys_first_force = force_keys_first ? 1 : 0;
    uint32_t* status = args[0].status;
    LA_LAUNCH(build_keys_kernel, dim3(gx, count), dim3(256), 0, stream, a0, b0, d_items, base);
    if (any_keys_first) LA_LAUNCH(sample_scan_kernel, dim3(64, count), dim3(256), 0, stream, b0, d_items, base);
    LA_LAUNCH(plan_kernel, dim3(kDigits, count), dim3(kRadix), 0, stream, b0, d_items, base);
    if (profile) (void)hipEventRecord(pf.ev[1], stream);
    for (int first = 0; first < count;) {                          // one set of pass launches per tile class
        const int sweep = lay[(size_t)idx[(size_t)first]].sweep_threads;
        int n = 0, max_tiles = 0;
        while (first + n < count && lay[(size_t)idx[(size_t)(first + n)]].sweep_threads == sweep) {
            if (h_items[first + n].n_tiles > max_tiles) max_tiles = h_items[first + n].n_tiles;
            ++n;
        }
        SortBufs bc = b0;
        bc.sweep_threads = sweep;
        bc.tile_state = (unsigned long long*)base;                 // (non-null: the single-kernel passes; items carry the real one)
        sort_run_passes(bc, stream, status, nullptr, bounds_pass_mask(a0), d_items + first, n, max_tiles, base);
        first += n;
    }
    if (any_keys_first) {
        if ((e = sort_repair_launch(b0, stream, d_items, count, base, max_n)) != hipSuccess) return e;
        for (int first = 0; first < count;) {                      // the redo slots (no-ops unless a run did not fit), per class
            const int sweep = lay[(size_t)idx[(size_t)first]].sweep_threads;
            int n = 0, max_tiles = 0;
            while (first + n < count && lay[(size_t)idx[(size_t)(first + n)]].sweep_threads == sweep) {
                if (h_items[first + n].n_tiles > max_tiles) max_tiles = h_items[first + n].n_tiles;
                ++n;
            }
            SortBufs bc = b0;
            bc.sweep_threads = sweep;
            bc.tile_state = (unsigned long long*)base;
            sort_run_passes(bc, stream, status, nullptr, bounds_pass_mask(a0), d_items + first, n, max_tiles, base, kDigits);
            first += n;
        }
    }
    if (profile) (void)hipEventRecord(pf.ev[2], stream);
    a0.rounds_follow = 1;
    LA_LAUNCH(emit_ids_kernel, dim3(gx, count), dim3(256), 0, stream, a0, b0, d_items, base);
    for (const Cls& c : cls)
        if ((e = launch_rounds_class(c.ec, c.threads, a0, b0, stream, d_items, base, d_order + c.first, c.n)) != hipSuccess) return e;
    if (!cls.empty()) LA_LAUNCH(map_ranks_kernel, dim3(gx, count), dim3(256), 0, stream, a0, b0, d_items, base);
    if (profile) (void)hipEventRecord(pf.ev[3], stream);
    return hipGetLastError();
}

// ---- more consumers than one workgroup's registers hold (> kLargeMaxConsumers): bins in HBM -------------------------------
// Collections.min over the bins takes any C (Main.java:240-263).  The round structure still holds: in round r the k-th
// sorted partition of the round goes to the k-th bin in (total, member) order as of the round start -- so a round is ONE
// stable device sort of the C bins by total (payload = consumer index, ascending on input: ties keep member order) with the
// radix passes above, and one pass that hands the round's lags out in that order.  ceil(P / C) rounds of ~11 launches:
// slow next to the one-workgroup kernels (which stop at 8 192 bins), exact, and without a limit.
//   huge_round_kernel   position k of the current order: total += lag of the round's k-th partition, member rank out; the
//                       new total goes to slot `consumer index` of the next sort's input, with its digit histograms.
__global__ __launch_bounds__(256) void huge_round_kernel(LargeArgs a, SortBufs parts, SortBufs cur, SortBufs next, int64_t round,
                                                         int first, int last) {
    __shared__ uint32_t h[8 * kRadix];
    for (int i = threadIdx.x; i < 8 * kRadix; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int64_t C = a.n_cons, P = a.n_part;
    const uint64_t* lagkey = parts.key[parts.ctl->cur[kFinal]];
    const uint32_t fin = first ? 0u : cur.ctl->cur[kFinal];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < C; k += stride) {
        uint64_t total = kTotalBias;                                  // biased: unsigned order = Java's signed order
        uint32_t idx = (uint32_t)k;
        if (!first) { total = cur.key[fin][k]; idx = cur.val[fin][k]; }
        const int64_t s = round * C + k;
        if (s < P) {
            total += lagkey[s] ^ kLagKeyFlip;                         // Main.java:265, wrapping like a long
            a.out_rank[a.p0 + s] = a.cons_rank[a.c0 + idx];
        }
        if (last) {
            if (a.out_total) a.out_total[a.c0 + idx] = (int64_t)(total ^ kTotalBias);
        } else {
            next.key[0][idx] = total;
            next.val[0][idx] = idx;
#pragma unroll
            for (int d = 0; d < 8; ++d) atomicAdd(&h[d * kRadix + ((uint32_t)(total >> (8 * d)) & 0xFFu)], 1u);
        }
    }
    if (last) return;
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * kRadix; i += blockDim.x)
        if (h[i]) atomicAdd(&next.hist[4 * kRadix + i], h[i]);
    // the id digits: the payload is ascending (slot = consumer index), the plan skips their passes
    if (blockIdx.x == 0 && threadIdx.x < 4) next.hist[threadIdx.x * kRadix] = (uint32_t)C;
}

hipError_t huge_topic_launch(LargeScratch& scratch, const LargeArgs& a, hipStream_t stream) {
    const int64_t P = a.n_part, C = a.n_cons;
    if (P <= 0) {
        if (C > 0 && a.out_total) return hipMemsetAsync(a.out_total + a.c0, 0, (size_t)C * sizeof(int64_t), stream);
        return hipSuccess;
    }
    if (P > 0x7FFFFFFF || C > 0x7FFFFFFF || C <= 0) return hipErrorInvalidValue;
    const SortLayout lp = sort_layout(P, a.sort_multi_kernel != 0, true), lc = sort_layout(C, false);
    hipError_t e;
    const size_t zero = lp.zero_bytes + 2 * lc.zero_bytes, data = lp.data_bytes + 2 * lc.data_bytes;
    if ((e = scratch_reserve(scratch, zero + data, stream)) != hipSuccess) return e;
    char* base = (char*)scratch.buf;
    const SortBufs bp = sort_bind(lp, base, base + zero);
    SortBufs bins[2] = {sort_bind(lc, base + lp.zero_bytes, base + zero + lp.data_bytes),
                        sort_bind(lc, base + lp.zero_bytes + lc.zero_bytes, base + zero + lp.data_bytes + lc.data_bytes)};
    if ((e = hipMemsetAsync(base, 0, lp.zero_bytes, stream)) != hipSuccess) return e;
    const int grid = keys_grid(P);
    LA_LAUNCH(build_keys_kernel, dim3(grid), dim3(256), 0, stream, a, bp, (const LargeItem*)nullptr, (char*)nullptr);
    if (bp.samp) LA_LAUNCH(sample_scan_kernel, dim3(256), dim3(256), 0, stream, bp, (const LargeItem*)nullptr, (char*)nullptr);
    sort_run_passes(bp, stream, a.status);
    if (bp.samp) {
        if ((e = sort_repair_launch(bp, stream, nullptr, 1, nullptr, P)) != hipSuccess) return e;
        sort_run_passes(bp, stream, a.status, nullptr, (1u << kDigits) - 1, nullptr, 1, 0, nullptr, kDigits);
    }
    LA_LAUNCH(emit_ids_kernel, dim3(grid), dim3(256), 0, stream, a, bp, (const LargeItem*)nullptr, (char*)nullptr);
    int cgrid = (int)((C + 255) / 256);
    if (cgrid > 1024) cgrid = 1024;
    const int64_t rounds = (P + C - 1) / C;
    for (int64_t r = 0; r < rounds; ++r) {
        const SortBufs& cur = bins[r & 1];
        const SortBufs& next = bins[(r + 1) & 1];
        const int last = r + 1 == rounds;
        if (!last && (e = hipMemsetAsync((char*)next.ctl, 0, lc.zero_bytes, stream)) != hipSuccess) return e;
        LA_LAUNCH(huge_round_kernel, dim3(cgrid), dim3(256), 0, stream, a, bp, cur, next, r, r == 0 ? 1 : 0, last);
        if (!last) sort_run_passes(next, stream, a.status, nullptr, 0xFF0u);       // the 8 digits of the totals; ids stay in order
    }
    return hipGetLastError();
}

// ---- the same grouping for what a real rebalance is: up to a few thousand entries, ONE workgroup (la_group_small.h) ---------
__global__ __launch_bounds__(1024) void group_small_kernel(int n, int32_t n_members, int64_t n_topics, const int64_t* part_off,
                                                           const int32_t* out_partition, const int32_t* member_rank,
                                                           int64_t* member_off, int32_t* grouped_topic,
                                                           int32_t* grouped_partition, int32_t* grouped_entry,
                                                           const uint32_t* fin_status, uint32_t* fin_flag) {
    __shared__ uint32_t start[kSmallGroupM];          // counts, then the groups' cursors
    __shared__ uint32_t wsum[1024 / kWave];
    __shared__ uint32_t turn;
#ifndef LA_GROUP_UNSTAGED
    __shared__ int32_t s_in[7 * kSmallGroupN];        // ranks, ids, topics of the entries; the lists; the chunks' group leaders
    group_small_body_staged<1024, kSmallGroupM>(n, n_members, n_topics, part_off, out_partition, member_rank, member_off,
                                                grouped_topic, grouped_partition, grouped_entry, start, wsum, &turn, s_in,
                                                s_in + kSmallGroupN, s_in + 2 * kSmallGroupN, s_in + 3 * kSmallGroupN, s_in + 4 * kSmallGroupN,
                                                reinterpret_cast<uint32_t*>(s_in + 5 * kSmallGroupN),
                                                             reinterpret_cast<uint32_t*>(s_in + 6 * kSmallGroupN));
#elif defined(LA_GROUP_SUB)                                   // (lab builds: tools/group_probe.py compares the forms on one box)
    group_small_body<1024, kSmallGroupM, LA_GROUP_SUB>(n, n_members, n_topics, part_off, out_partition, member_rank, member_off,
                                                       grouped_topic, grouped_partition, grouped_entry, start, wsum, &turn);
#else
    group_small_body<1024, kSmallGroupM>(n, n_members, n_topics, part_off, out_partition, member_rank, member_off, grouped_topic,
                                         grouped_partition, grouped_entry, start, wsum, &turn);
#endif
    if (fin_flag) {
        // the last launch of a zero-copy call (la_api.hip, assign_small_zc): this ONE workgroup's stores into the host's memory
        // are out, then `done | status` goes where the calling thread is spinning -- no separate finishing launch
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(fin_flag, 0x80000000u | *fin_status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t group_by_member_launch(LargeScratch& scratch, int64_t n, int32_t n_members, int64_t n_topics,
                                  const int64_t* part_off, const int32_t* out_partition, const int32_t* member_rank,
                                  int64_t* member_off, int32_t* grouped_topic, int32_t* grouped_partition,
                                  int32_t* grouped_entry, uint32_t* status, hipStream_t stream, uint32_t* fin_flag, bool* fin_done) {
    if (fin_done) *fin_done = false;
    if (n < 0 || n > 0x7FFFFFFF || n_members < 0) return hipErrorInvalidValue;
    hipError_t e;
    if (n == 0) return hipMemsetAsync(member_off, 0, sizeof(int64_t) * ((size_t)n_members + 1), stream);
    if (n <= kSmallGroupN && (int64_t)n_members + 2 <= kSmallGroupM && ((int64_t)n_members + 1) < ((int64_t)1 << (kSmallGroupBits + 1)) &&
        !getenv("LA_NO_SMALL_GROUP")) {
        LA_LAUNCH(group_small_kernel, dim3(1), dim3(1024), 0, stream, (int)n, n_members, n_topics, part_off, out_partition,
                           member_rank, member_off, grouped_topic, grouped_partition, grouped_entry, (const uint32_t*)status,
                           fin_flag);
        if (fin_done) *fin_done = fin_flag != nullptr;
        return hipGetLastError();
    }
    if (n <= (int64_t)kMidBlock * kMidMaxBlocks && (int64_t)n_members + 2 <= kMidGroupM && !getenv("LA_NO_MID_GROUP")) {
        // a mid-size rebalance with few members: the stable counting sort over several workgroups, two launches
        // (la_group_small.h) instead of the radix form's five
        const int blocks = (int)((n + kMidBlock - 1) / kMidBlock);
        if ((e = scratch_reserve(scratch, ((size_t)blocks * kMidGroupM + 1) * sizeof(uint32_t), stream)) != hipSuccess) return e;
        uint32_t* cnt = (uint32_t*)scratch.buf;
        LA_LAUNCH(group_mid_count_kernel, dim3(blocks), dim3(kMidThreads), 0, stream, (int)n, n_members, member_rank, cnt);
        LA_LAUNCH(group_mid_place_kernel, dim3(blocks), dim3(kMidThreads), 0, stream, (int)n, n_members, n_topics, part_off, out_partition,
                  member_rank, cnt, member_off, grouped_topic, grouped_partition, grouped_entry, (const uint32_t*)status, fin_flag);
        if (fin_done) *fin_done = fin_flag != nullptr;
        return hipGetLastError();
    }
    SortBufs b{};
    if ((e = sort_prepare(scratch, n, stream, &b)) != hipSuccess) return e;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    LA_LAUNCH(member_keys_kernel, dim3(grid), dim3(256), 0, stream, member_rank, b);
    // key = rank + 1 <= n_members: only its low ceil(bits / 8) digits can differ; the payload (entry index) is ascending.
    // Launching just those passes matters for the small batches a real group leader sends: every skipped pass used to
    // cost four empty launches (~50 launches, ~190 us, for a 100-partition rebalance).
    uint32_t mask = 0;
    for (int d = 0; d < 8 && ((uint64_t)n_members >> (8 * d)) != 0; ++d) mask |= 1u << (4 + d);
    sort_run_passes(b, stream, status, nullptr, mask);
    LA_LAUNCH(member_emit_kernel, dim3(grid), dim3(256), 0, stream, b, n_members, n_topics, part_off,
                       out_partition, member_off, grouped_topic, grouped_partition, grouped_entry, status);
    return hipGetLastError();
}

#ifdef LA_GROUP_CLOCKS
extern "C" __attribute__((visibility("default"))) int la_debug_group_clocks(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_group_clocks), sizeof(g_group_clocks));
    if (e == hipSuccess && reset) {
        unsigned long long zero[12] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_group_clocks), zero, sizeof zero);
    }
    return e == hipSuccess ? 0 : -3;
}
#endif

#ifdef LA_SWEEP_CLOCKS
extern "C" __attribute__((visibility("default"))) int la_debug_sweep_clocks(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sweep_clocks), sizeof(g_sweep_clocks));
    if (e == hipSuccess && reset) {
        unsigned long long zero[12] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_sweep_clocks), zero, sizeof zero);
    }
    return e == hipSuccess ? 0 : -3;
}
#endif

#ifdef LA_LOOKBACK_STATS
extern "C" __attribute__((visibility("default"))) int la_debug_lookback_stats(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lookback_stats), sizeof(g_lookback_stats));
    if (e == hipSuccess && reset) {
        unsigned long long zero[8] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_lookback_stats), zero, sizeof zero);
    }
    return e == hipSuccess ? 0 : -3;
}
#endif

#ifdef LA_ROUND_CLOCKS
extern "C" __attribute__((visibility("default"))) int la_debug_round_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_round_stamp), sizeof(g_round_stamp)) == hipSuccess ? 0 : -3;
}
extern "C" __attribute__((visibility("default"))) int la_debug_round_clocks(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_round_clocks), sizeof(g_round_clocks));
    if (e == hipSuccess && reset) {
        unsigned long long zero[16] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_round_clocks), zero, sizeof zero);
    }
    return e == hipSuccess ? 0 : -3;
}
#endif

void large_scratch_release(LargeScratch& s) {
    if (s.buf) (void)hipFree(s.buf);
    s.buf = nullptr;
    s.cap = 0;
    for (LargeScratch::Stage& sg : s.stage) {
        if (sg.done) { (void)hipEventSynchronize(sg.done); (void)hipEventDestroy(sg.done); sg.done = nullptr; }
        if (sg.h) (void)hipHostFree(sg.h);
        sg.h = nullptr;
        sg.cap = 0;
    }
    if (s.d_items) (void)hipFree(s.d_items);
    s.d_items = nullptr;
    s.d_items_cap = 0;
    for (hipEvent_t& ev : s.prof.ev) {
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;
    }
}

hipError_t large_profile_read(LargeScratch& s, float* ms, int* passes, int64_t* n) {
    if (!s.prof.recorded || !s.buf) return hipErrorNotReady;
    hipError_t e;
    if ((e = hipEventSynchronize(s.prof.ev[3])) != hipSuccess) return e;
    for (int i = 0; i < 3; ++i)
        if ((e = hipEventElapsedTime(&ms[i], s.prof.ev[i], s.prof.ev[i + 1])) != hipSuccess) return e;
    SortCtl ctl;                                    // the control block sits at the start of the scratch
    if ((e = hipMemcpy(&ctl, s.buf, sizeof ctl, hipMemcpyDeviceToHost)) != hipSuccess) return e;
    passes[0] = passes[1] = 0;
    for (int p = 0; p < kSlots; ++p)
        if (!ctl.skip[p]) ++passes[p % kDigits < 4 ? 0 : 1];
    passes[2] = (int)ctl.keys_first;
    passes[3] = (int)ctl.redo;
    *n = s.prof.n;
    return hipSuccess;
}

}  // namespace la
