// tools/sort_lab.hip -- developer check (NOT product): the multi-wavefront 64-bit bin sort of la_large.hip in isolation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../kafka_lag_based_assignor_amd/csrc/la_large.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
namespace la { namespace {
template <int EC>
__global__ __launch_bounds__(1024) void sort_test_kernel(const uint64_t* in, uint64_t* out, int stop_after) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint64_t* s_bin = reinterpret_cast<uint64_t*>(smem);
    const int tid = threadIdx.x;
    const int n = EC * blockDim.x;
    constexpr int kSpan = 64 * EC;
    P64 rec[EC];
#pragma unroll
    for (int r = 0; r < EC; ++r) rec[r] = p64_from(in[tid * EC + r]);
    dpp_fence<EC>(rec);
    bitonic_sort_tile_p64<64, EC>(rec);
    int step = 0;
    for (int K = 2 * kSpan; K <= n && step < stop_after; K <<= 1) {
        cross_wave_step<EC>(rec, s_bin, tid, K - 1, K >> 1);
        for (int j = K >> 2; j >= kSpan; j >>= 1) cross_wave_step<EC>(rec, s_bin, tid, j, j);
        dpp_fence<EC>(rec);
        clean_p64<64, EC, kSpan / 2, false>(rec);
        ++step;
    }
#pragma unroll
    for (int r = 0; r < EC; ++r) out[tid * EC + r] = p64_value(rec[r]);
}
}}
template <int EC>
void run(int threads, int kind) {
    const int n = EC * threads;
    std::vector<uint64_t> h(n), ref, got(n);
    uint64_t z = 12345 + n * 7 + kind;
    for (int i = 0; i < n; ++i) {
        z = z * 6364136223846793005ull + 1442695040888963407ull;
        h[i] = kind == 0 ? (z >> 1) : kind == 1 ? ((z >> 40) << 13 | i) : (i % 7 == 0 ? ~0ull : (z >> 20));
    }
    ref = h; std::sort(ref.begin(), ref.end());
    uint64_t *din, *dout;
    CK(hipMalloc(&din, n * 8)); CK(hipMalloc(&dout, n * 8));
    CK(hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)la::sort_test_kernel<EC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((la::sort_test_kernel<EC>), dim3(1), dim3(threads), (size_t)n * 12, 0, din, dout, 100);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), dout, n * 8, hipMemcpyDeviceToHost));
    int bad = 0, first = -1;
    for (int i = 0; i < n; ++i) if (got[i] != ref[i]) { if (first < 0) first = i; ++bad; }
    // is every 64*EC block sorted at least?
    printf("EC=%d threads=%d n=%d kind=%d: %s (bad %d first %d)\n", EC, threads, n, kind, bad ? "WRONG" : "ok", bad, first);
    if (bad && n == 128 && kind == 1) {
        std::vector<uint64_t> g2 = got; std::sort(g2.begin(), g2.end());
        printf("  multiset preserved: %d\n", (int)(g2 == ref));
        bool lo_ok = true; uint64_t mx = 0, mn = ~0ull;
        for (int i = 0; i < 64; ++i) mx = std::max(mx, got[i]);
        for (int i = 64; i < 128; ++i) mn = std::min(mn, got[i]);
        printf("  max(lower half) <= min(upper half): %d\n", (int)(mx <= mn));
        for (int i = 0; i < 128; i += 1) printf("%s%llx", i % 8 ? " " : "\n   ", (unsigned long long)(got[i] >> 13));
        printf("\n");
    }
    CK(hipFree(din)); CK(hipFree(dout));
}
int main() {
    for (int kind = 0; kind < 3; ++kind) {
        run<1>(64, kind); run<1>(128, kind); run<1>(256, kind); run<1>(1024, kind);
        run<2>(1024, kind); run<4>(1024, kind); run<8>(1024, kind); run<8>(128, kind); run<4>(64, kind);
    }
    return 0;
}
