// tools/tile_lab.hip -- developer harness (NOT product, NOT a test): times wave_tile_assign_kernel<32,8>
// variants on the bench's target shape with HIP events, one binary per -DLA_ABLATE value.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLA_LAB -DLA_ABLATE=0 tools/tile_lab.hip -o tools/_lab/lab0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../kafka_lag_based_assignor_amd/csrc/la_wave_tile_l8.hip"
#include "../kafka_lag_based_assignor_amd/csrc/la_wave_tile_l16.hip"
#include "../kafka_lag_based_assignor_amd/csrc/la_wave_tile_l32.hip"
#include "../kafka_lag_based_assignor_amd/csrc/la_wave_tile_l64.hip"
#include "../kafka_lag_based_assignor_amd/csrc/la_wave_tile.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void init_kernel(int64_t n, int P, int64_t* begin, int64_t* end, int64_t* com, int32_t* pid) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const int64_t lag = (int64_t)(z % 1000000000ull);
        const int64_t c = (int64_t)((z >> 40) & 0xFFFFF);
        begin[i] = 0;
        com[i] = ((z >> 33) % 100 == 0) ? -1 : c;
        end[i] = c + lag;
        const int t = (int)(i / P), k = (int)(i % P);
        pid[i] = (k * 77 + 13 * t) % P;
    }
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 100000, P = argc > 2 ? atoi(argv[2]) : 256, C = argc > 3 ? atoi(argv[3]) : 32;
    const int reps = argc > 4 ? atoi(argv[4]) : 30;
    const int64_t n = (int64_t)T * P, k = (int64_t)T * C;
    int64_t *begin, *end, *com, *part_off, *cons_off, *out_total;
    int32_t *pid, *cons_rank, *out_pid, *out_rank;
    uint32_t* status;
    CK(hipMalloc(&begin, n * 8)); CK(hipMalloc(&end, n * 8)); CK(hipMalloc(&com, n * 8)); CK(hipMalloc(&pid, n * 4));
    CK(hipMalloc(&out_pid, n * 4)); CK(hipMalloc(&out_rank, n * 4)); CK(hipMalloc(&out_total, k * 8));
    CK(hipMalloc(&part_off, (T + 1) * 8)); CK(hipMalloc(&cons_off, (T + 1) * 8)); CK(hipMalloc(&cons_rank, k * 4));
    CK(hipMalloc(&status, 256)); CK(hipMemset(status, 0, 256));
    std::vector<int64_t> po(T + 1), co(T + 1);
    std::vector<int32_t> cr(k);
    for (int t = 0; t <= T; ++t) { po[t] = (int64_t)t * P; co[t] = (int64_t)t * C; }
    for (int64_t i = 0; i < k; ++i) cr[i] = (int32_t)(i % C);
    CK(hipMemcpy(part_off, po.data(), (T + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(cons_off, co.data(), (T + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(cons_rank, cr.data(), k * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(init_kernel, dim3(2048), dim3(256), 0, 0, n, P, begin, end, com, pid);
    CK(hipDeviceSynchronize());

    la::TileArgs a{};
    a.n_topics = T; a.part_off = part_off; a.pid = pid; a.begin = begin; a.end = end; a.committed = com; a.lag = nullptr;
    a.cons_off = cons_off; a.cons_rank = cons_rank; a.out_pid = out_pid; a.out_rank = out_rank; a.out_total = out_total;
    a.status = status; a.reset_latest = 0; a.n_total = n; a.k_total = k;
    int32_t* defer; CK(hipMalloc(&defer, (size_t)T * 4 + 64)); CK(hipMemset(defer, 0, 64));
    a.defer_count = defer; a.defer_count_next = defer + 1; a.defer_list = defer + 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        for (int w = 0; w < (reps > 100 ? 300 : 3); ++w) CK(la::wave_tile_launch(a, P, C, mode, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) CK(la::wave_tile_launch(a, P, C, mode, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("ablate=%d mode=%s T=%d P=%d C=%d: %.1f us/launch  %.2f TB/s @36B  %.3g assign/s\n", LA_ABLATE,
               mode ? "wide" : "auto", T, P, C, us, 36.0 * n / us / 1e6, n / us * 1e6);
    }
    uint32_t st = 0;
    CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
    if (st) printf("status=%u\n", st);
    return 0;
}
