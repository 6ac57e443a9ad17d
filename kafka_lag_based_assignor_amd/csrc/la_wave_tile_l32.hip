// la_wave_tile_l32.hip -- the wave-tile kernels for groups of 32 lanes (all E, all modes).
#include "la_wave_tile_impl.h"

namespace la {
hipError_t wave_tile_launch_l32(int e, const TileArgs& a, int mode, hipStream_t stream, bool* tail_done) {
    return launch_l<32>(e, a, mode, stream, tail_done);
}
}  // namespace la
