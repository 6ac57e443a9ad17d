// tools/inst_lab2.hip -- developer microbenchmark (NOT product): per-SIMD throughput of single VALU forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

#define DEFK(ID, I0, I1, I2, I3)                                                                  \
    __global__ __launch_bounds__(256) void k##ID(uint64_t* out, int iters) {                       \
        uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x9E3779B9u, c = a + 77, d = b + 13;      \
        uint32_t e = a * 3, f = b * 5, g = c * 7, h = d * 9;                                      \
        uint64_t m0 = 0x5555555555555555ull, m1 = 0x3333333333333333ull;                          \
        for (int i = 0; i < iters; ++i)                                                           \
            asm volatile(REP16(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t")                           \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+s"(m0), "+s"(m1) \
                         : : "vcc");                                                              \
        if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)m0 ^ (uint32_t)m1) == 0x12345678u) out[1] = a; \
    }

// %0..%7 = a..h (VGPR), %8 = m0, %9 = m1 (SGPR pairs)
DEFK(0, "v_add_u32 %0, %0, %4", "v_add_u32 %1, %1, %5", "v_add_u32 %2, %2, %6", "v_add_u32 %3, %3, %7")
DEFK(1, "v_min_u32 %0, %0, %4", "v_max_u32 %1, %1, %5", "v_min_u32 %2, %2, %6", "v_max_u32 %3, %3, %7")
DEFK(2, "v_xor_b32 %0, %0, %4", "v_and_b32 %1, %1, %5", "v_or_b32 %2, %2, %6", "v_xor_b32 %3, %3, %7")
DEFK(3, "v_cmp_lt_u32_e64 %8, %0, %4", "v_cmp_lt_u32_e64 %9, %1, %5", "v_cmp_lt_u32_e64 %8, %2, %6", "v_cmp_lt_u32_e64 %9, %3, %7")
DEFK(4, "v_cmp_lt_u32 vcc, %0, %4", "v_cmp_lt_u32 vcc, %1, %5", "v_cmp_lt_u32 vcc, %2, %6", "v_cmp_lt_u32 vcc, %3, %7")
DEFK(5, "v_cndmask_b32_e64 %0, %0, %4, %8", "v_cndmask_b32_e64 %1, %1, %5, %9", "v_cndmask_b32_e64 %2, %2, %6, %8", "v_cndmask_b32_e64 %3, %3, %7, %9")
DEFK(6, "v_mov_b32 %0, %4", "v_mov_b32 %1, %5", "v_mov_b32 %2, %6", "v_mov_b32 %3, %7")
DEFK(7, "v_min3_u32 %0, %0, %4, %5", "v_max3_u32 %1, %1, %5, %6", "v_med3_u32 %2, %2, %6, %7", "v_min3_u32 %3, %3, %7, %4")
DEFK(8, "v_bfi_b32 %0, %4, %0, %5", "v_bfi_b32 %1, %5, %1, %6", "v_bfi_b32 %2, %6, %2, %7", "v_bfi_b32 %3, %7, %3, %4")
DEFK(9, "v_add_co_u32 %0, vcc, %0, %4", "v_addc_co_u32 %1, vcc, %1, %5, vcc", "v_add_co_u32 %2, vcc, %2, %6", "v_addc_co_u32 %3, vcc, %3, %7, vcc")
DEFK(10, "v_sub_co_u32_e64 %0, %8, %0, %4", "v_subb_co_u32_e64 %1, %8, %1, %5, %8", "v_sub_co_u32_e64 %2, %9, %2, %6", "v_subb_co_u32_e64 %3, %9, %3, %7, %9")
DEFK(11, "v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %5 row_mirror row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %2, %6 row_ror:8 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
DEFK(12, "v_add_u32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %1, %5, %1 row_mirror row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %2, %6, %2 row_ror:8 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %3, %7, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
DEFK(13, "v_cndmask_b32 %0, %0, %4, vcc", "v_cndmask_b32 %1, %1, %5, vcc", "v_cndmask_b32 %2, %2, %6, vcc", "v_cndmask_b32 %3, %3, %7, vcc")
DEFK(14, "v_cndmask_b32_e64 %0, %0, %4, vcc", "v_cndmask_b32_e64 %1, %1, %5, vcc", "v_cndmask_b32_e64 %2, %2, %6, vcc", "v_cndmask_b32_e64 %3, %3, %7, vcc")
DEFK(15, "v_pk_min_u16 %0, %0, %4", "v_pk_max_u16 %1, %1, %5", "v_pk_min_u16 %2, %2, %6", "v_pk_max_u16 %3, %3, %7")
DEFK(16, "v_lshl_add_u32 %0, %0, 3, %4", "v_lshl_or_b32 %1, %1, 3, %5", "v_add3_u32 %2, %2, %6, %7", "v_and_or_b32 %3, %3, %7, %4")
DEFK(17, "v_perm_b32 %0, %0, %4, %5", "v_alignbit_b32 %1, %1, %5, 8", "v_perm_b32 %2, %2, %6, %7", "v_alignbit_b32 %3, %3, %7, 8")

DEFK(21, "v_min_f64 %[ab], %[ab], %[cd]", "v_max_f64 %[cd], %[cd], %[ab]", "v_min_f64 %[ab], %[ab], %[cd]", "v_max_f64 %[cd], %[cd], %[ab]")
DEFK(19, "s_xor_b64 %8, %8, %9", "v_add_u32 %0, %0, %4", "s_xor_b64 %9, %9, %8", "v_add_u32 %1, %1, %5")
DEFK(20, "v_readlane_b32 s20, %0, 3", "v_add_u32 %0, %0, %4", "v_readlane_b32 s21, %1, 5", "v_add_u32 %1, %1, %5")

typedef void (*kern_t)(uint64_t*, int);
void run(const char* name, kern_t kf, int valu_per_iter, uint64_t* d_out) {
    const int iters = 2000;
    printf("%-40s", name);
    for (int wps : {1, 4, 8}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), 0, 0, d_out, 10);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), 0, 0, d_out, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  w%d: %5.2f ns", wps, ms * 1e6 / ((double)valu_per_iter * iters * wps));
    }
    printf("   (per instruction per SIMD)\n");
}

int main() {
    uint64_t* d_out; CK(hipMalloc(&d_out, 64));
#define R(ID, NAME) run(NAME, k##ID, 64, d_out)
    R(0, "v_add_u32"); R(1, "v_min/max_u32"); R(2, "v_xor/and/or"); R(3, "v_cmp_lt_u32_e64 -> sgpr"); R(4, "v_cmp_lt_u32 -> vcc");
    R(5, "v_cndmask_e64 sgpr"); R(6, "v_mov_b32"); R(7, "v_min3/max3/med3_u32"); R(8, "v_bfi_b32"); R(9, "v_add_co/addc_co vcc");
    R(10, "v_sub_co/subb_co e64 sgpr"); R(11, "v_mov_b32_dpp"); R(12, "v_add_u32_dpp"); R(13, "v_cndmask e32 vcc"); R(14, "v_cndmask e64 vcc");
    R(15, "v_pk_min/max_u16"); R(16, "v_lshl_add/lshl_or/add3/and_or"); R(17, "v_perm/alignbit"); R(19, "s_xor + v_add (2 VALU of 4)");
    R(20, "v_readlane + v_add");
    return 0;
}
