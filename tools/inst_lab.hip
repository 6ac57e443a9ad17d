// tools/inst_lab.hip -- developer microbenchmark (NOT product): issue cost of the VALU forms the sort uses,
// per SIMD, at 1 / 4 / 8 waves per SIMD.  Prints cycles per instruction relative to s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x9E3779B9u, c = a + 77, d = b + 13, t = 0, u = 0, w = 0;
    const uint64_t keep = 0x5555555555555555ull;
    uint64_t m0 = keep, m1 = keep;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {   // independent v_fma_f32-like: v_add_u32 x4 independent
            asm volatile(REP16("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(t));
        } else if constexpr (KIND == 1) {   // dependent chain v_add_u32
            asm volatile(REP64("v_add_u32 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
        } else if constexpr (KIND == 2) {   // v_cndmask e32 vcc, independent
            asm volatile(REP16("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %3, %3, %0, vcc\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if constexpr (KIND == 3) {   // v_mov_b32_dpp quad_perm independent (4 regs rotate; 3 apart)
            asm volatile(REP16("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %6, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(t), "+v"(u), "+v"(w));
        } else if constexpr (KIND == 4) {   // v_cmp_lt_u64 e64 to sgpr pair, independent
            uint64_t x = ((uint64_t)a << 32) | b, y = ((uint64_t)c << 32) | d;
            asm volatile(REP64("v_cmp_lt_u64 vcc, %0, %1\n\t") : : "v"(x), "v"(y) : "vcc");
        } else if constexpr (KIND == 5) {   // the fused block: sub_dpp, subb_dpp, xor, cnd_dpp x2 on 4 independent records, interleaved by the hw only
            asm volatile(REP16("v_sub_co_u32_dpp %2, vcc, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_subb_co_u32_dpp %2, vcc, %1, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "s_xor_b64 vcc, vcc, %5\n\t"
                               "v_cndmask_b32_dpp %0, %0, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_cndmask_b32_dpp %1, %1, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_sub_co_u32_dpp %2, vcc, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_subb_co_u32_dpp %2, vcc, %4, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "s_xor_b64 vcc, vcc, %5\n\t"
                               "v_cndmask_b32_dpp %3, %3, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_cndmask_b32_dpp %4, %4, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a), "+v"(b), "=&v"(t), "+v"(c), "+v"(d) : "s"(keep) : "vcc", "scc");
        } else if constexpr (KIND == 6) {   // v_sub_co_u32 / v_subb_co_u32 chain pairs, non-dpp
            asm volatile(REP16("v_sub_co_u32 %2, vcc, %0, %1\n\tv_subb_co_u32 %2, vcc, %1, %0, vcc\n\t"
                               "v_sub_co_u32 %3, vcc, %1, %0\n\tv_subb_co_u32 %3, vcc, %0, %1, vcc\n\t")
                         : "+v"(a), "+v"(b), "=&v"(t), "=&v"(u) : : "vcc");
        } else if constexpr (KIND == 7) {   // permlane16 swap
            asm volatile(REP16("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                               "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if constexpr (KIND == 8) {   // v_cndmask_b32_dpp alone, independent
            asm volatile(REP16("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_cndmask_b32_dpp %6, %0, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(t), "+v"(u), "+v"(w) : : "vcc");
        } else if constexpr (KIND == 9) {   // v_cndmask_b32 e64 with sgpr mask, independent
            asm volatile(REP16("v_cndmask_b32_e64 %0, %0, %1, %4\n\tv_cndmask_b32_e64 %1, %1, %2, %4\n\tv_cndmask_b32_e64 %2, %2, %3, %4\n\tv_cndmask_b32_e64 %3, %3, %0, %4\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(keep));
        } else if constexpr (KIND == 11) {  // cndmask e32 vcc, vcc set once per asm block by s_mov
            asm volatile("s_mov_b64 vcc, %4\n\t" REP16("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %3, %3, %0, vcc\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(keep) : "vcc");
        } else if constexpr (KIND == 12) {  // cndmask e64 with vcc named explicitly
            asm volatile("s_mov_b64 vcc, %4\n\t" REP16("v_cndmask_b32_e64 %0, %0, %1, vcc\n\tv_cndmask_b32_e64 %1, %1, %2, vcc\n\tv_cndmask_b32_e64 %2, %2, %3, vcc\n\tv_cndmask_b32_e64 %3, %3, %0, vcc\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(keep) : "vcc");
        } else if constexpr (KIND == 13) {  // v_cmp (writes vcc) + 2 cndmask e32 reading it, 2 independent pairs
            asm volatile(REP16("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %0, vcc\n\t"
                               "v_cmp_lt_u32 vcc, %2, %3\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %3, %3, %2, vcc\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if constexpr (KIND == 14) {  // v_cmp e64 -> sgpr pair + 2 cndmask e64 reading it (compiler's form), with s_nop 1
            asm volatile(REP16("v_cmp_lt_u32_e64 %4, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, %4\n\tv_cndmask_b32_e64 %1, %1, %0, %4\n\t"
                               "v_cmp_lt_u32_e64 %5, %2, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %2, %2, %3, %5\n\tv_cndmask_b32_e64 %3, %3, %2, %5\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&s"(m0), "=&s"(m1));
        } else if constexpr (KIND == 15) {  // s_xor writes sgpr mask then cndmask e64 reads (SALU->VALU)
            asm volatile(REP16("s_xor_b64 %4, %4, %5\n\tv_cndmask_b32_e64 %0, %0, %1, %4\n\tv_cndmask_b32_e64 %1, %1, %0, %4\n\t"
                               "s_xor_b64 %4, %4, %5\n\tv_cndmask_b32_e64 %2, %2, %3, %4\n\tv_cndmask_b32_e64 %3, %3, %2, %4\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(m0) : "s"(keep) : "scc");
        } else if constexpr (KIND == 16) {  // v_min_u32 / v_max_u32 with dpp
            asm volatile(REP16("v_min_u32_dpp %4, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_max_u32_dpp %5, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_min_u32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_max_u32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(t), "+v"(u));
        } else if constexpr (KIND == 17) {  // exec-masked v_swap_b32
            asm volatile(REP16("v_swap_b32 %0, %1\n\tv_swap_b32 %2, %3\n\tv_swap_b32 %0, %1\n\tv_swap_b32 %2, %3\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if constexpr (KIND == 18) {  // 32-bit lane cmpx: min_dpp, max_dpp, cndmask e64 (2 independent keys)
            asm volatile(REP16("v_min_u32_dpp %4, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_max_u32_dpp %5, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_cndmask_b32_e64 %0, %5, %4, %6\n\t"
                               "v_min_u32_dpp %4, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_max_u32_dpp %5, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_cndmask_b32_e64 %1, %5, %4, %6\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t), "=&v"(u) : "s"(keep));
        } else if constexpr (KIND == 19) {  // 64-bit lane cmpx without dpp selects: 2 mov_dpp, sub, subb (e64 sgpr), xor, 2 cndmask e64
            asm volatile(REP16("v_mov_b32_dpp %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %5, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                               "v_sub_co_u32_e64 %6, %7, %4, %0\n\t"
                               "v_subb_co_u32_e64 %6, %7, %5, %1, %7\n\t"
                               "s_xor_b64 %7, %7, %8\n\t"
                               "s_nop 0\n\t"
                               "v_cndmask_b32_e64 %0, %4, %0, %7\n\t"
                               "v_cndmask_b32_e64 %1, %5, %1, %7\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t), "=&v"(u), "=&v"(w), "=&s"(m0) : "s"(keep) : "scc");
        } else if constexpr (KIND == 10) {  // row_mirror dpp mov
            asm volatile(REP16("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %4, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %6, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(t), "+v"(u), "+v"(w));
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if ((a ^ b ^ c ^ d ^ t ^ u ^ w ^ (uint32_t)m0 ^ (uint32_t)m1) == 0x12345678u) out[1] = a;
}

template <int KIND>
void run(const char* name, int valu_per_iter, uint64_t* d_out) {
    const int iters = 2000;
    for (int wps : {1, 4, 8}) {
        // 256 CUs x (wps waves per SIMD x 4 SIMDs) = blocks of 256 threads: wps blocks per CU
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 10);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
        CK(hipGetLastError());
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        uint64_t cyc; CK(hipMemcpy(&cyc, d_out, 8, hipMemcpyDeviceToHost));
        const double insts_per_simd = (double)valu_per_iter * iters * wps;
        printf("%-34s waves/SIMD=%d  %.2f ns/inst/SIMD  (memtime ticks per inst per wave %.2f; x%d waves)\n", name, wps,
               ms * 1e6 / insts_per_simd, (double)cyc / ((double)valu_per_iter * iters), wps);
    }
}

int main() {
    uint64_t* d_out; CK(hipMalloc(&d_out, 64));
    run<0>("v_add_u32 independent", 64, d_out);
    run<1>("v_add_u32 dependent", 64, d_out);
    run<2>("v_cndmask e32 vcc", 64, d_out);
    run<9>("v_cndmask e64 sgpr", 64, d_out);
    run<3>("v_mov_dpp quad_perm", 64, d_out);
    run<10>("v_mov_dpp row_mirror/ror/half", 64, d_out);
    run<8>("v_cndmask_dpp", 64, d_out);
    run<4>("v_cmp_lt_u64", 64, d_out);
    run<6>("v_sub_co/v_subb_co", 64, d_out);
    run<7>("v_permlane16_swap", 64, d_out);
    run<5>("64-bit fused cmpx (4 VALU: sub_dpp subb_dpp cnd_dpp x2)", 16 * 8, d_out);
    run<19>("64-bit cmpx e64 form (6 VALU)", 16 * 6, d_out);
    run<18>("32-bit cmpx (3 VALU: min_dpp max_dpp cnd_e64)", 16 * 6, d_out);
    run<11>("v_cndmask e32 vcc (s_mov vcc once)", 64, d_out);
    run<12>("v_cndmask e64 explicit vcc", 64, d_out);
    run<13>("v_cmp vcc + 2 cndmask e32", 16 * 6, d_out);
    run<14>("v_cmp sgpr + nop + 2 cndmask e64", 16 * 6, d_out);
    run<15>("s_xor sgpr + 2 cndmask e64", 16 * 4, d_out);
    run<16>("v_min/max_u32_dpp", 64, d_out);
    run<17>("v_swap_b32", 64, d_out);
    return 0;
}
