// la_wave_tile.hip -- host-side dispatch of the wave-tile path: picks the tile shape (L lanes per topic, E
// records per lane) and calls the launcher of that group width (la_wave_tile_l{8,16,32,64}.hip; device code
// in la_wave_tile_impl.h).
#include "la_kernels.h"

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
    switch (L) {
        case 8: return wave_tile_launch_l8(E, a, mode, stream, tail_done);
        case 16: return wave_tile_launch_l16(E, a, mode, stream, tail_done);
        case 32: return wave_tile_launch_l32(E, a, mode, stream, tail_done);
        default: return wave_tile_launch_l64(E, a, mode, stream, tail_done);
    }
}

}  // namespace la
