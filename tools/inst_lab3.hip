// tools/inst_lab3.hip -- developer microbenchmark: v_min_f64 / v_max_f64 as exact 64-bit unsigned min/max.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
__global__ __launch_bounds__(256) void k(uint64_t* out, int iters) {
    uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull >> 8, b = a ^ 0x1234567ull, c = a + 77, d = b + 13;
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_min_f64 %0, %0, %1\n\tv_max_f64 %1, %1, %2\n\tv_min_f64 %2, %2, %3\n\tv_max_f64 %3, %3, %0\n\t")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if ((a ^ b ^ c ^ d) == 0x12345678u) out[1] = a;
}
// exactness: min/max of bit patterns incl. denormal range and values up to 2^62
__global__ void exact(const uint64_t* x, const uint64_t* y, uint64_t* mn, uint64_t* mx, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint64_t a = x[i], b = y[i], lo, hi;
        asm volatile("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b));
        mn[i] = lo; mx[i] = hi;
    }
}
int main() {
    uint64_t* d_out; CK(hipMalloc(&d_out, 64));
    for (int wps : {1, 4, 8}) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(256 * wps), dim3(256), 0, 0, d_out, 10); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(256 * wps), dim3(256), 0, 0, d_out, 2000);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("v_min/max_f64 w%d: %.2f ns per instruction per SIMD\n", wps, ms * 1e6 / (64.0 * 2000 * wps));
    }
    const int n = 1 << 20;
    uint64_t *x, *y, *mn, *mx;
    CK(hipMallocManaged(&x, n * 8)); CK(hipMallocManaged(&y, n * 8)); CK(hipMallocManaged(&mn, n * 8)); CK(hipMallocManaged(&mx, n * 8));
    uint64_t z = 1;
    for (int i = 0; i < n; ++i) {
        z = z * 6364136223846793005ull + 1442695040888963407ull; uint64_t u = z;
        z = z * 6364136223846793005ull + 1442695040888963407ull; uint64_t v = z;
        int sa = (i * 7) % 63, sb = (i * 13) % 63;
        x[i] = (u >> 2) >> sa; y[i] = (v >> 2) >> sb;          // < 2^62, all magnitudes incl. tiny (denormal patterns)
        if (i % 5 == 0) y[i] = x[i];
        if (i % 11 == 0) y[i] = x[i] + 1;
    }
    hipLaunchKernelGGL(exact, dim3(n / 256), dim3(256), 0, 0, x, y, mn, mx, n); CK(hipDeviceSynchronize());
    int bad = 0;
    for (int i = 0; i < n; ++i) { uint64_t lo = x[i] < y[i] ? x[i] : y[i], hi = x[i] < y[i] ? y[i] : x[i]; if (mn[i] != lo || mx[i] != hi) { if (bad < 5) printf("mismatch %llx %llx -> %llx %llx\n", (unsigned long long)x[i], (unsigned long long)y[i], (unsigned long long)mn[i], (unsigned long long)mx[i]); ++bad; } }
    printf("exactness over %d pairs (< 2^62): %d mismatches\n", n, bad);
    return 0;
}
