// Sort-based canonical relabel for the many-classes regime of refine! / Partition(M)
// (src/partitions.jl:24-35,44-66).  With ~n^2/2 distinct signatures (problems without symmetry:
// BASELINE configs[1], configs[2]) the hash-table path degenerates into one global atomic per
// entry on a table far larger than any cache (27 GB/s of algorithmic bytes at n = 4096).  Here:
//   1. stable LSD radix sort of (signature, linear index) pairs  -- rocPRIM's wave-level radix
//      passes through hipcub::DeviceRadixSort, 64 key bits;
//   2. run heads: the first entry of a run of equal signatures carries the run's smallest linear
//      index (stability) = the first occurrence in the column-major scan;
//   3. canonical numbering = rank of the first-occurrence indices: heads scatter a flag to their
//      index, one exclusive sum over the flags ranks them (no second sort);
//   4. every entry fetches the rank of its run head (inclusive max-scan of head positions).
// Signature 0 is the structurally-zero class (label 0, not counted).
#include <hipcub/hipcub.hpp>

#include "sdpsr_internal.h"

namespace sdpsr {

__global__ void rs_iota_kernel(int64_t len, uint32_t* __restrict__ idx, uint32_t* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        idx[e] = (uint32_t)e;
        flags[e] = 0u;
    }
}

__global__ void rs_heads_kernel(int64_t len, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx,
                                uint32_t* __restrict__ headpos, uint32_t* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < len; p += stride) {
        const uint64_t k = keys[p];
        const bool head = (p == 0) || keys[p - 1] != k;
        headpos[p] = head ? (uint32_t)p : 0u;
        if (head && k != 0ull) flags[idx[p]] = 1u;
    }
}

__global__ void rs_labels_kernel(int64_t len, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx,
                                 const uint32_t* __restrict__ headpos, const uint32_t* __restrict__ rank,
                                 uint32_t* __restrict__ labels) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < len; p += stride) {
        const uint64_t k = keys[p];
        labels[idx[p]] = (k == 0ull) ? 0u : rank[idx[headpos[p]]] + 1u;
    }
}

__global__ void rs_count_kernel(int64_t len, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ flags,
                                uint32_t* __restrict__ counters) {
    const uint32_t d = rank[len - 1] + flags[len - 1];
    counters[0] = d;
    counters[1] = 0u;
    counters[2] = d;
}

struct RsMax {
    __host__ __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

// workspace: keys_out (len u64) | idx_in, idx_out, headpos, hp, flags, rank (len u32 each) | cub temp
size_t refine_sorted_workspace_bytes(int64_t len) {
    size_t t1 = 0, t2 = 0, t3 = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, t1, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                       (uint32_t*)nullptr, (int)len, 0, 64, nullptr);
    hipcub::DeviceScan::InclusiveScan(nullptr, t2, (const uint32_t*)nullptr, (uint32_t*)nullptr, RsMax(), (int)len, nullptr);
    hipcub::DeviceScan::ExclusiveSum(nullptr, t3, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)len, nullptr);
    const size_t tmp = std::max(t1, std::max(t2, t3));
    return (size_t)len * 8 + (size_t)len * 4 * 6 + tmp + 1024;
}

// labels_out: canonical labels; counters[0] = counters[2] = number of classes, counters[1] = 0.
bool launch_refine_sorted(hipStream_t s, int64_t len, const uint64_t* sig, uint32_t* labels_out, void* ws, size_t ws_bytes,
                          uint32_t* counters) {
    char* p = (char*)ws;
    uint64_t* keys = (uint64_t*)p;
    p += (size_t)len * 8;
    uint32_t* idx_in = (uint32_t*)p;
    p += (size_t)len * 4;
    uint32_t* idx = (uint32_t*)p;
    p += (size_t)len * 4;
    uint32_t* headpos = (uint32_t*)p;
    p += (size_t)len * 4;
    uint32_t* hp = (uint32_t*)p;
    p += (size_t)len * 4;
    uint32_t* flags = (uint32_t*)p;
    p += (size_t)len * 4;
    uint32_t* rank = (uint32_t*)p;
    p += (size_t)len * 4;
    p = (char*)(((uintptr_t)p + 255) & ~uintptr_t(255));
    size_t tmp = ws_bytes - (size_t)(p - (char*)ws);
    int64_t g = (len + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    rs_iota_kernel<<<(unsigned)g, 256, 0, s>>>(len, idx_in, flags);
    size_t t = tmp;
    if (hipcub::DeviceRadixSort::SortPairs(p, t, sig, keys, idx_in, idx, (int)len, 0, 64, s) != hipSuccess) return false;
    rs_heads_kernel<<<(unsigned)g, 256, 0, s>>>(len, keys, idx, headpos, flags);
    t = tmp;
    if (hipcub::DeviceScan::InclusiveScan(p, t, headpos, hp, RsMax(), (int)len, s) != hipSuccess) return false;
    t = tmp;
    if (hipcub::DeviceScan::ExclusiveSum(p, t, flags, rank, (int)len, s) != hipSuccess) return false;
    rs_labels_kernel<<<(unsigned)g, 256, 0, s>>>(len, keys, idx, hp, rank, labels_out);
    rs_count_kernel<<<1, 1, 0, s>>>(len, rank, flags, counters);
    return true;
}

}  // namespace sdpsr
