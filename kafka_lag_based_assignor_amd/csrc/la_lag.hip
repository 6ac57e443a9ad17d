// la_lag.hip -- kernel 1: elementwise partition lag, computePartitionLag (Main.java:376-404).
//
// HBM-bound: 24 B read (16 B in LATEST mode, begin is never touched) + 8 B written per
// partition, no reuse.  16 B per lane per load (two int64), grid-stride, <= 2048 workgroups.
#include "la_kernels.h"
#include "la_device.h"

namespace la {

struct alignas(16) i64x2 { int64_t x, y; };

template <bool LATEST>
__global__ __launch_bounds__(256) void lag_kernel_vec2(int64_t n2, const i64x2* __restrict__ begin,
                                                       const i64x2* __restrict__ end,
                                                       const i64x2* __restrict__ committed,
                                                       i64x2* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const i64x2 e = end[i], c = committed[i];
        i64x2 b = {0, 0};
        if constexpr (!LATEST) b = begin[i];
        i64x2 o;
        o.x = partition_lag(b.x, e.x, c.x, LATEST);
        o.y = partition_lag(b.y, e.y, c.y, LATEST);
        out[i] = o;
    }
}

__global__ __launch_bounds__(256) void lag_kernel_scalar(int64_t first, int64_t n, const int64_t* begin,
                                                         const int64_t* end, const int64_t* committed,
                                                         int reset_latest, int64_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t b = (!reset_latest && begin) ? begin[i] : 0;
        out[i] = partition_lag(b, end[i], committed[i], reset_latest != 0);
    }
}

static unsigned grid_for(int64_t work_items) {
    int64_t blocks = (work_items + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

hipError_t lag_launch(int64_t n, const int64_t* begin, const int64_t* end, const int64_t* committed,
                      bool reset_latest, int64_t* out_lag, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const bool need_begin = !reset_latest && begin != nullptr;
    const uintptr_t bits = (uintptr_t)end | (uintptr_t)committed | (uintptr_t)out_lag |
                           (need_begin ? (uintptr_t)begin : 0);
    int64_t done = 0;
    if ((bits & 15) == 0 && (reset_latest || begin != nullptr)) {
        const int64_t n2 = n / 2;
        if (n2 > 0) {
            if (reset_latest)
                LA_LAUNCH(lag_kernel_vec2<true>, dim3(grid_for(n2)), dim3(256), 0, stream, n2,
                                   (const i64x2*)nullptr, (const i64x2*)end, (const i64x2*)committed,
                                   (i64x2*)out_lag);
            else
                LA_LAUNCH(lag_kernel_vec2<false>, dim3(grid_for(n2)), dim3(256), 0, stream, n2,
                                   (const i64x2*)begin, (const i64x2*)end, (const i64x2*)committed,
                                   (i64x2*)out_lag);
        }
        done = n2 * 2;
    }
    if (done < n)
        LA_LAUNCH(lag_kernel_scalar, dim3(grid_for(n - done)), dim3(256), 0, stream, done, n, begin,
                           end, committed, reset_latest ? 1 : 0, out_lag);
    return hipGetLastError();
}

// ---- consumer-list validation ---------------------------------------------------------------
__global__ __launch_bounds__(256) void check_consumers_kernel(int64_t n_topics, const int64_t* cons_off,
                                                              const int32_t* cons_rank, uint32_t* status) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_topics; t += stride) {
        const int64_t c0 = cons_off[t], c1 = cons_off[t + 1];
        for (int64_t k = c0 + 1; k < c1; ++k) bad |= cons_rank[k - 1] >= cons_rank[k];
    }
    if (bad) atomicOr(status, kStatusUnsorted);
}

hipError_t check_consumers_launch(int64_t n_topics, const int64_t* cons_off, const int32_t* cons_rank,
                                  uint32_t* status, hipStream_t stream) {
    if (n_topics <= 0) return hipSuccess;
    LA_LAUNCH(check_consumers_kernel, dim3(grid_for(n_topics)), dim3(256), 0, stream, n_topics,
                       cons_off, cons_rank, status);
    return hipGetLastError();
}

// ---- zero-copy small calls: the last launch of the call hands the device status word to the host --------------------------
// Kernels before it on the stream have completed (their stores to the host-mapped result area included); one plain store of
// `done bit | status` into coherent host memory is what the calling thread spins on instead of a stream synchronize.
__global__ void finish_status_kernel(const uint32_t* d_status, uint32_t* h_flag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const uint32_t st = *d_status;
        __threadfence_system();
        __hip_atomic_store(h_flag, 0x80000000u | st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t finish_status_launch(const uint32_t* d_status, uint32_t* h_flag, hipStream_t stream) {
    LA_LAUNCH(finish_status_kernel, dim3(1), dim3(64), 0, stream, d_status, h_flag);
    return hipGetLastError();
}

// ---- la_wake's empty launch (the streams its one-partition rebalance does not run on; la_api.hip) ------------------------------
__global__ void wake_kernel() {}

hipError_t wake_launch(hipStream_t stream) {
    LA_LAUNCH(wake_kernel, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

// ---- sparse begin offsets (la_assign_batch_sparse) ---------------------------------------------------------
// The earliest-offset fallback reads `begin` only where a partition has no committed offset (Main.java:384-396) -- ~1 % of
// a batch -- so the boundary can hand over (position, begin) pairs for just those instead of a dense 8 B/partition array.
// The dense array the kernels read is rebuilt here: zeroed by the caller (an unlisted partition has begin 0, the reference's
// getOrDefault(tp, 0L), Main.java:350-351), then entry j lands at begin[idx[j] - base].  [lo, hi) is the range of positions
// the sub-list was cut for: an entry outside it means the list was not ascending (or out of the batch) -> kStatusSparse.
__global__ __launch_bounds__(256) void sparse_begin_kernel(int64_t m, const int64_t* __restrict__ idx, const int64_t* __restrict__ val,
                                                           int64_t base, int64_t lo, int64_t hi, int64_t* __restrict__ begin,
                                                           uint32_t* status) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const int64_t g = idx[j];
        if (g < lo || g >= hi) bad = true;
        else begin[g - base] = val[j];
    }
    if (bad) atomicOr(status, kStatusSparse);
}

hipError_t sparse_begin_launch(int64_t m, const int64_t* idx, const int64_t* val, int64_t base, int64_t lo, int64_t hi,
                               int64_t* begin, uint32_t* status, hipStream_t stream) {
    if (m <= 0) return hipSuccess;
    LA_LAUNCH(sparse_begin_kernel, dim3(grid_for(m)), dim3(256), 0, stream, m, idx, val, base, lo, hi, begin, status);
    return hipGetLastError();
}

}  // namespace la
