// la_wave_tile.hip -- host-side dispatch of the wave-tile path: picks the tile shape (L lanes per topic, E
// records per lane) and calls the launcher of that group width (la_wave_tile_l{8,16,32,64}.hip; device code
// in la_wave_tile_impl.h).
#include "la_kernels.h"

#include <cstdlib>

namespace la {

hipError_t wave_tile_launch_l8(int e, const TileArgs& a, int mode, hipStream_t stream, bool* tail_done);
hipError_t wave_tile_launch_l16(int e, const TileArgs& a, int mode, hipStream_t stream, bool* tail_done);
hipError_t wave_tile_launch_l32(int e, const TileArgs& a, int mode, hipStream_t stream, bool* tail_done);
hipError_t wave_tile_launch_l64(int e, const TileArgs& a, int mode, hipStream_t stream, bool* tail_done);

static int pow2ceil(int64_t x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
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

// A launch that leaves most of the chip idle is a matter of one wavefront's latency, not of throughput: the chain a wavefront
// runs is E x steps(L x E) slots of the partitions' sort network (steps(n) = the bitonic network's log2 n (log2 n + 1) / 2)
// plus the greedy rounds, whose network is as wide as the consumers need whatever L is (greedy_rounds_tile<L, LB>).  So with
// wavefronts to spare a topic gets twice the lanes and half the records per lane, again and again, as long as the launch stays
// within two wavefronts per SIMD (2 048 on this chip).  BASELINE config 3 (1 000 x 256 x 32) runs as 64 lanes x 4 records
// instead of 32 x 8: 0.0139 -> 0.0111 ms; 4 000 x 100 x 5: 0.0164 -> 0.0097 ms; 2 000 x 128 x 4: 0.0136 -> 0.0096 ms; a
// 2 000-partition rebalance at the C ABI 40 -> 36 us.  Limit 1 024 / 2 048 / 4 096 / 8 192 wavefronts: 2 048 takes most of the
// gain, 8 192 loses on 8 000 x 256 x 32 and 12 000 x 64 x 8; a cost model that charged the rounds steps(L) stopped too early
// (2 000 x 128 x 4: 0.0111 ms).  profiles/r05_w_tile_widen.txt.  Same results for any valid shape (the tests run them all; the
// narrow shapes at small sizes in a process with LA_NO_TILE_WIDEN=1, which keeps round 4's pick).  L x E does not change, so
// wave_tile_always_packs (the packed format's capacity term) holds for the widened shape as well.
constexpr int64_t kLatencyWaves = 2048;

static void wave_tile_widen(int64_t n_topics, int64_t max_p, int* L, int* E) {
    static const bool off = getenv("LA_NO_TILE_WIDEN") != nullptr;
    if (off) return;
    static const int64_t max_waves = [] { const char* e = getenv("LA_TILE_WIDEN_WAVES"); return e ? (int64_t)atoll(e) : kLatencyWaves; }();  // (lab)
    while (*L < 64 && *E >= 2) {
        const int l2 = *L * 2;
        const int64_t per_wave = 64 / l2;
        if ((n_topics + per_wave - 1) / per_wave > max_waves) break;
        const int e2 = pow2ceil((max_p + l2 - 1) / l2);
        *L = l2;
        *E = e2 < 1 ? 1 : e2;
    }
}

// The packed format's condition (la_wave_tile_impl.h, packed_tile): sh = bits of the ids' OR (at least 1), lbw = bits of the
// largest lag; a wavefront packs when sh < 32 and lbw <= min(63 - sh, 57 - log2(lanes x records)).  Bounds on ids and lags
// bound sh and lbw from above for every wavefront.
bool wave_tile_always_packs(int64_t max_p, int64_t max_c, int64_t max_lag, int64_t max_id) {
    if (max_lag < 0 || max_id < 0 || max_id > 0x7FFFFFFFll) return false;
    int L, E;
    wave_tile_pick(max_p, max_c, &L, &E);
    auto bits = [](uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; };
    int log2cap = 0;
    while ((1 << log2cap) < L * E) ++log2cap;
    const int sh = bits((uint64_t)max_id | 1u), lbw = bits((uint64_t)max_lag);
    int lim = 63 - sh;
    if (lim > 57 - log2cap) lim = 57 - log2cap;
    return sh < 32 && lbw <= lim;
}

hipError_t wave_tile_launch(TileArgs a, int64_t max_p, int64_t max_c, int mode, hipStream_t stream, bool* tail_done) {
    int L, E;
    if (tail_done) *tail_done = false;
    if (a.n_total <= 0) return hipSuccess;
    if (max_p > a.n_total) max_p = a.n_total;      // no topic holds more than the batch (E >= 2 needs 2 elements)
    wave_tile_pick(max_p, max_c, &L, &E);
    wave_tile_widen(a.n_topics, max_p, &L, &E);
    switch (L) {
        case 8: return wave_tile_launch_l8(E, a, mode, stream, tail_done);
        case 16: return wave_tile_launch_l16(E, a, mode, stream, tail_done);
        case 32: return wave_tile_launch_l32(E, a, mode, stream, tail_done);
        default: return wave_tile_launch_l64(E, a, mode, stream, tail_done);
    }
}

}  // namespace la
