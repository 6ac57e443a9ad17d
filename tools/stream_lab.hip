// stream_lab.hip -- what plain streaming kernels sustain on this MI355X with the tile kernel's traffic shape:
// per partition 8 B + 8 B + 4 B read (committed, end, partition id) and 4 B + 4 B written (partition order, member rank),
// 25.6 M partitions.  The ceiling the tile kernel's memory side (791 MB per launch) is measured against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_lab tools/stream_lab.hip && /tmp/stream_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct __attribute__((aligned(16))) L2 { int64_t x, y; };
struct __attribute__((aligned(8))) I2 { int32_t x, y; };
typedef int I4 __attribute__((ext_vector_type(4)));

// one thread: 2 partitions per step (16 B of each int64 array, 8 B of ids), like a tile lane's pair
template <bool NT>
__global__ __launch_bounds__(256) void mix_kernel(const L2* a, const L2* b, const I2* c, I2* o1, I2* o2, int64_t n2) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const L2 x = a[i], y = b[i];
        const I2 z = c[i];
        I2 r1, r2;
        r1.x = (int32_t)(x.x - y.x) ^ z.x; r1.y = (int32_t)(x.y - y.y) ^ z.y;
        r2.x = (int32_t)(x.x + y.y); r2.y = (int32_t)(x.y + y.x) + z.y;
        if (NT) { __builtin_nontemporal_store(r1.x, &o1[i].x); __builtin_nontemporal_store(r1.y, &o1[i].y);
                  __builtin_nontemporal_store(r2.x, &o2[i].x); __builtin_nontemporal_store(r2.y, &o2[i].y); }
        else { o1[i] = r1; o2[i] = r2; }
    }
}

// the tile kernel's shape: a wavefront owns 512 consecutive partitions, every lane issues ALL its loads first
// (4 x 16 B of each int64 array, 4 x 8 B of ids), then stores 2 x 16 B per output array; one tile per wavefront
__global__ __launch_bounds__(256) void tile_shape_kernel(const L2* a, const L2* b, const I2* c, I4* o1, I4* o2, int64_t n_tiles) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int64_t base2 = tile * 256;                  // pairs
    L2 x[4], y[4];
    I2 z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = a[base2 + k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) { y[k] = b[base2 + k * 64 + lane]; z[k] = c[base2 + k * 64 + lane]; }
    I4 r1[2], r2[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        r1[k] = (I4){(int)(x[2 * k].x - y[2 * k].x), (int)(x[2 * k].y - y[2 * k].y), (int)(x[2 * k + 1].x - y[2 * k + 1].x), (int)(x[2 * k + 1].y - y[2 * k + 1].y)};
        r2[k] = (I4){z[2 * k].x, z[2 * k].y, z[2 * k + 1].x, z[2 * k + 1].y};
    }
    const int64_t base4 = tile * 128;                  // quads
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        __builtin_nontemporal_store(r1[k], &o1[base4 + k * 64 + lane]);
        __builtin_nontemporal_store(r2[k], &o2[base4 + k * 64 + lane]);
    }
}

__global__ __launch_bounds__(256) void read_kernel(const L2* a, const L2* b, const I2* c, int64_t n2, int64_t* sink) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const L2 x = a[i], y = b[i];
        const I2 z = c[i];
        acc += x.x ^ x.y ^ y.x ^ y.y ^ z.x ^ z.y;
    }
    if (acc == 0x123456789) *sink = acc;
}


// ---- the radix pass's traffic shape without its work: every workgroup streams one tile of TILE elements in and writes it
// out as TILE / RUN runs of RUN elements, run d of tile t at  d * (n / bins) + slot(t, d) * RUN  -- what a pass does when
// every digit is equally likely (slot() spreads a digit's runs over the tiles so that the bins do not alias onto one set of
// memory channels, as the real digit counts do).  Two arrays (8 B keys, 4 B ids) or ONE array of REC-byte records.
template <int RUN, int OFF = 0>
__global__ __launch_bounds__(1024) void scatter_two_arrays(const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout,
                                                           int64_t n, int n_tiles) {
    constexpr int TILE = 16384, BINS = TILE / RUN;
    const int t = blockIdx.x;
    const int64_t per_bin = n / BINS;
    uint64_t k[16]; uint32_t v[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) { const int64_t i = (int64_t)t * TILE + it * 1024 + threadIdx.x; k[it] = kin[i]; v[it] = vin[i]; }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int j = it * 1024 + threadIdx.x, d = j / RUN, r = j % RUN;
        const int64_t dst = (int64_t)d * per_bin + (int64_t)((t + d * 37) % n_tiles) * RUN + r + (OFF ? (d * 13 + OFF) % RUN : 0);
        kout[dst] = k[it];
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int j = it * 1024 + threadIdx.x, d = j / RUN, r = j % RUN;
        const int64_t dst = (int64_t)d * per_bin + (int64_t)((t + d * 37) % n_tiles) * RUN + r + (OFF ? (d * 13 + OFF) % RUN : 0);
        vout[dst] = v[it];
    }
}

struct __attribute__((aligned(4))) Rec12 { uint32_t a, b, c; };
struct __attribute__((aligned(16))) Rec16 { uint32_t a, b, c, d; };
template <typename REC, int RUN, int ITEMS>
__global__ __launch_bounds__(1024) void scatter_one_array(const REC* in, REC* out, int64_t n, int n_tiles) {
    constexpr int TILE = 1024 * ITEMS, BINS = TILE / RUN;
    const int t = blockIdx.x;
    const int64_t per_bin = n / BINS;
    REC x[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) x[it] = in[(int64_t)t * TILE + it * 1024 + threadIdx.x];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int j = it * 1024 + threadIdx.x, d = j / RUN, r = j % RUN;
        out[(int64_t)d * per_bin + (int64_t)((t + d * 37) % n_tiles) * RUN + r] = x[it];
    }
}

int main() {
    const int64_t n = 25600000, n2 = n / 2;
    void *a, *b, *c, *o1, *o2; int64_t* sink;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&o1, n * 4)); CK(hipMalloc(&o2, n * 4));
    CK(hipMalloc((void**)&sink, 8));
    CK(hipMemset(a, 1, n * 8)); CK(hipMemset(b, 2, n * 8)); CK(hipMemset(c, 3, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes_mix = (double)n * 28, bytes_rd = (double)n * 20;
    auto time = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < 300; ++i) launch();              // settle (power controller), as the bench does
        hipEventRecord(e0, nullptr);
        const int reps = 1000;
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.2f us per launch  %6.0f GB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
    };
    for (int grid : {2048, 8192, 50000}) {
        char nm[96];
        snprintf(nm, sizeof nm, "mix 28 B/partition, grid %d, plain stores", grid);
        time(nm, bytes_mix, [&] { hipLaunchKernelGGL(mix_kernel<false>, dim3(grid), dim3(256), 0, nullptr, (const L2*)a, (const L2*)b, (const I2*)c, (I2*)o1, (I2*)o2, n2); });
        snprintf(nm, sizeof nm, "mix 28 B/partition, grid %d, nt stores", grid);
        time(nm, bytes_mix, [&] { hipLaunchKernelGGL(mix_kernel<true>, dim3(grid), dim3(256), 0, nullptr, (const L2*)a, (const L2*)b, (const I2*)c, (I2*)o1, (I2*)o2, n2); });
    }
    time("tile shape (one 512-partition tile per wave)", bytes_mix, [&] { hipLaunchKernelGGL(tile_shape_kernel, dim3(12500), dim3(256), 0, nullptr, (const L2*)a, (const L2*)b, (const I2*)c, (I4*)o1, (I4*)o2, (int64_t)50000); });
    time("read only 20 B/partition, grid 8192", bytes_rd, [&] { hipLaunchKernelGGL(read_kernel, dim3(8192), dim3(256), 0, nullptr, (const L2*)a, (const L2*)b, (const I2*)c, n2, sink); });

    {   // radix-pass shapes on 33.5 M elements (402 MB in + 402 MB out per pass: past the Infinity Cache)
        const int64_t m = (int64_t)1 << 25;
        const int tiles = (int)(m / 16384);
        void *ki, *vi, *ko, *vo;
        CK(hipMalloc(&ki, m * 16)); CK(hipMalloc(&vi, m * 4)); CK(hipMalloc(&ko, m * 16)); CK(hipMalloc(&vo, m * 4));
        CK(hipMemset(ki, 1, m * 16)); CK(hipMemset(vi, 2, m * 4));
        const double bytes24 = (double)m * 24, bytes32 = (double)m * 32;
        time("scatter, keys 8 B + ids 4 B, runs of 32", bytes24, [&] { hipLaunchKernelGGL(scatter_two_arrays<32>, dim3(tiles), dim3(1024), 0, nullptr, (const uint64_t*)ki, (const uint32_t*)vi, (uint64_t*)ko, (uint32_t*)vo, m, tiles); });
        time("scatter, keys 8 B + ids 4 B, runs of 64 (ours)", bytes24, [&] { hipLaunchKernelGGL(scatter_two_arrays<64>, dim3(tiles), dim3(1024), 0, nullptr, (const uint64_t*)ki, (const uint32_t*)vi, (uint64_t*)ko, (uint32_t*)vo, m, tiles); });
        time("scatter, runs of 64 starting off the cache lines", bytes24, [&] { hipLaunchKernelGGL((scatter_two_arrays<64, 7>), dim3(tiles), dim3(1024), 0, nullptr, (const uint64_t*)ki, (const uint32_t*)vi, (uint64_t*)ko + 64, (uint32_t*)vo + 64, m - 128, tiles - 1); });
        time("scatter, keys 8 B + ids 4 B, runs of 128", bytes24, [&] { hipLaunchKernelGGL(scatter_two_arrays<128>, dim3(tiles), dim3(1024), 0, nullptr, (const uint64_t*)ki, (const uint32_t*)vi, (uint64_t*)ko, (uint32_t*)vo, m, tiles); });
        time("scatter, keys 8 B + ids 4 B, runs of 1024", bytes24, [&] { hipLaunchKernelGGL(scatter_two_arrays<1024>, dim3(tiles), dim3(1024), 0, nullptr, (const uint64_t*)ki, (const uint32_t*)vi, (uint64_t*)ko, (uint32_t*)vo, m, tiles); });
        const int tiles12 = (int)(m / 12288);
        time("scatter, ONE array of 12 B records, runs of 48", bytes24, [&] { hipLaunchKernelGGL((scatter_one_array<Rec12, 48, 12>), dim3(tiles12), dim3(1024), 0, nullptr, (const Rec12*)ki, (Rec12*)ko, (int64_t)tiles12 * 12288, tiles12); });
        time("scatter, ONE array of 16 B records, runs of 64", bytes32, [&] { hipLaunchKernelGGL((scatter_one_array<Rec16, 64, 16>), dim3(tiles), dim3(1024), 0, nullptr, (const Rec16*)ki, (Rec16*)ko, m, tiles); });
    }
    return 0;
}
