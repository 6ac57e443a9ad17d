// la_wire.hip -- the narrow wire format of the single all-gather (north star: "a single RCCL all-gather over
// xGMI to reassemble the global assignment map").
//
// What crosses xGMI per assigned partition is a (partition id, member) pair in assignment order (Main.java:264).  As two
// int32 arrays that is 8 B; at the target shape a pair needs 8 + 6 bits.  One element of the wire format is
//
//     ((member rank + 1) << id_bits) | partition id            rank + 1 = 0: the topic had no consumer (Main.java:211-213)
//
// in the fewest of 2 / 4 / 8 bytes that hold it (la_wire_format_for; 8 bytes, id_bits = 32 carries any int32 pair).  The
// xGMI links are the slow station of the N > 1 step (7 links x ~77 GB/s per direction against ~5 TB/s of HBM), so the bytes
// of the gather are what the step costs: 25.6 MB -> 6.4 MB per rank on the 8-GPU target split.
//
// pack:   reads 8 B, writes elem_bytes per partition; raises kStatusWire when a pair does not fit the format it was given
// unpack: reads elem_bytes, writes 8 B per partition (every rank, over the whole gathered map)
// Both are plain streaming kernels: 8 consecutive elements per thread, 16-byte accesses where the pointers allow.
#include "la_kernels.h"
#include "la_device.h"

namespace la {

namespace {

constexpr int kVec = 8;      // elements per thread
typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));     // one 16-byte access

template <typename W>
__device__ __forceinline__ W wire_pack_one(int32_t id, int32_t rank, int id_bits, uint32_t id_mask, uint64_t rank_limit, bool& bad) {
    const uint64_t r1 = (uint64_t)((uint32_t)rank + 1u);                // -1 -> 0; ranks are >= -1 by contract
    if constexpr (sizeof(W) == 8) {
        return (W)((r1 << 32) | (uint32_t)id);                          // any int32 id, any rank >= -1
    } else {
        bad |= ((uint32_t)id & ~id_mask) != 0 || r1 >= rank_limit || rank < -1;
        return (W)((r1 << id_bits) | ((uint32_t)id & id_mask));
    }
}

template <typename W, bool VEC>
__global__ __launch_bounds__(256) void wire_pack_kernel(int64_t n, const int32_t* __restrict__ pid, const int32_t* __restrict__ rank,
                                                        int id_bits, W* __restrict__ out, uint32_t* status) {
    const uint32_t id_mask = id_bits >= 32 ? 0xFFFFFFFFu : ((1u << id_bits) - 1u);
    const uint64_t rank_limit = sizeof(W) == 8 ? ~0ull : (1ull << (8 * sizeof(W) - id_bits));
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * kVec;
    bool bad = false;
    for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * kVec; base < n; base += stride) {
        if (VEC && base + kVec <= n) {
            union { U32x4 v[2]; int32_t e[kVec]; } ps, rs;
            ps.v[0] = *reinterpret_cast<const U32x4*>(pid + base); ps.v[1] = *reinterpret_cast<const U32x4*>(pid + base + 4);
            rs.v[0] = *reinterpret_cast<const U32x4*>(rank + base); rs.v[1] = *reinterpret_cast<const U32x4*>(rank + base + 4);
            constexpr int kWords = kVec * (int)sizeof(W) / 16;           // 16-byte stores: 1, 2 or 4
            union { U32x4 v[kWords]; W e[kVec]; } w;
#pragma unroll
            for (int k = 0; k < kVec; ++k) w.e[k] = wire_pack_one<W>(ps.e[k], rs.e[k], id_bits, id_mask, rank_limit, bad);
            U32x4* dst = reinterpret_cast<U32x4*>(out + base);
#pragma unroll
            for (int k = 0; k < kWords; ++k) __builtin_nontemporal_store(w.v[k], dst + k);
        } else {
            for (int k = 0; k < kVec && base + k < n; ++k)
                out[base + k] = wire_pack_one<W>(pid[base + k], rank[base + k], id_bits, id_mask, rank_limit, bad);
        }
    }
    if (__any(bad) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(status, kStatusWire);
}

template <typename W, bool VEC>
__global__ __launch_bounds__(256) void wire_unpack_kernel(int64_t n, const W* __restrict__ in, int id_bits, int32_t* __restrict__ pid,
                                                          int32_t* __restrict__ rank) {
    const uint32_t id_mask = id_bits >= 32 ? 0xFFFFFFFFu : ((1u << id_bits) - 1u);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * kVec;
    auto one = [&](W w, int32_t& p, int32_t& r) {
        p = (int32_t)((uint32_t)w & id_mask);
        r = (int32_t)(uint32_t)((uint64_t)w >> id_bits) - 1;
    };
    for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * kVec; base < n; base += stride) {
        if (VEC && base + kVec <= n) {
            constexpr int kWords = kVec * (int)sizeof(W) / 16;
            union { U32x4 v[kWords]; W e[kVec]; } w;
            const U32x4* src = reinterpret_cast<const U32x4*>(in + base);
#pragma unroll
            for (int k = 0; k < kWords; ++k) w.v[k] = src[k];
            union { U32x4 v[2]; int32_t e[kVec]; } ps, rs;
#pragma unroll
            for (int k = 0; k < kVec; ++k) one(w.e[k], ps.e[k], rs.e[k]);
            U32x4* dp = reinterpret_cast<U32x4*>(pid + base);
            U32x4* dr = reinterpret_cast<U32x4*>(rank + base);
            __builtin_nontemporal_store(ps.v[0], dp);
            __builtin_nontemporal_store(ps.v[1], dp + 1);
            __builtin_nontemporal_store(rs.v[0], dr);
            __builtin_nontemporal_store(rs.v[1], dr + 1);
        } else {
            for (int k = 0; k < kVec && base + k < n; ++k) one(in[base + k], pid[base + k], rank[base + k]);
        }
    }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

inline int wire_grid(int64_t n) {
    const int64_t g = (n + 256 * kVec - 1) / (256 * kVec);
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

// Pure host arithmetic: the narrowest element that holds ids in [0, max_partition_id] and ranks in [-1, n_members).
// max_partition_id < 0 means "ids may be any int32" (negative ids are legal partition numbers to the kernels).
void wire_format_for(int64_t max_partition_id, int64_t n_members, int* elem_bytes, int* id_bits) {
    auto bits = [](uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; };
    if (max_partition_id < 0 || max_partition_id > 0x7FFFFFFFll || n_members < 0 || n_members > 0x7FFFFFFFll) {
        *elem_bytes = 8; *id_bits = 32;
        return;
    }
    const int ib = bits((uint64_t)max_partition_id), rb = bits((uint64_t)n_members);    // rank + 1 takes the values 0 .. n_members
    if (ib + rb <= 16) { *elem_bytes = 2; *id_bits = ib; }
    else if (ib + rb <= 32) { *elem_bytes = 4; *id_bits = ib; }
    else { *elem_bytes = 8; *id_bits = 32; }
}

bool wire_format_valid(int elem_bytes, int id_bits) {
    if (elem_bytes == 8) return id_bits == 32;
    return (elem_bytes == 2 || elem_bytes == 4) && id_bits >= 0 && id_bits < 8 * elem_bytes;
}

hipError_t wire_pack_launch(int64_t n, const int32_t* pid, const int32_t* rank, int elem_bytes, int id_bits, void* out,
                            uint32_t* status, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (!wire_format_valid(elem_bytes, id_bits)) return hipErrorInvalidValue;
    const bool vec = aligned16(pid) && aligned16(rank) && aligned16(out);
    const dim3 grid(wire_grid(n)), block(256);
#define LA_PACK(W)                                                                                                   \
    do {                                                                                                             \
        if (vec) LA_LAUNCH((wire_pack_kernel<W, true>), grid, block, 0, stream, n, pid, rank, id_bits, (W*)out, status);   \
        else LA_LAUNCH((wire_pack_kernel<W, false>), grid, block, 0, stream, n, pid, rank, id_bits, (W*)out, status);      \
    } while (0)
    if (elem_bytes == 2) LA_PACK(uint16_t);
    else if (elem_bytes == 4) LA_PACK(uint32_t);
    else LA_PACK(uint64_t);
#undef LA_PACK
    return hipGetLastError();
}

hipError_t wire_unpack_launch(int64_t n, const void* in, int elem_bytes, int id_bits, int32_t* pid, int32_t* rank,
                              hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (!wire_format_valid(elem_bytes, id_bits)) return hipErrorInvalidValue;
    const bool vec = aligned16(pid) && aligned16(rank) && aligned16(in);
    const dim3 grid(wire_grid(n)), block(256);
#define LA_UNPACK(W)                                                                                                 \
    do {                                                                                                             \
        if (vec) LA_LAUNCH((wire_unpack_kernel<W, true>), grid, block, 0, stream, n, (const W*)in, id_bits, pid, rank);    \
        else LA_LAUNCH((wire_unpack_kernel<W, false>), grid, block, 0, stream, n, (const W*)in, id_bits, pid, rank);       \
    } while (0)
    if (elem_bytes == 2) LA_UNPACK(uint16_t);
    else if (elem_bytes == 4) LA_UNPACK(uint32_t);
    else LA_UNPACK(uint64_t);
#undef LA_UNPACK
    return hipGetLastError();
}

}  // namespace la
