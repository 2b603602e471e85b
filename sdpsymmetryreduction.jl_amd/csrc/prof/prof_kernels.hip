// Measurement kernels of libsdpsr_prof.so (not in the product library).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../sdpsr_hash.h"

namespace sdpsr {

// ---------------------------------------------------------------------------
// Shader-clock meter (diagnostic, sdpsr_profile_clock): one wave on a side stream samples
// (clock64 = shader cycles, wall_clock64 = 100 MHz) every ~20 us while the kernels under test run
// on the main stream.  The int8 squares are power-limited on this part: the clock they run at,
// not their cycle count, decides the rate (measured: 2.1-2.2 GHz under MFMAs alone, ~1.4 GHz
// under the full kernel).  Terminates on the flag, after ns samples or after max_ticks.
// ---------------------------------------------------------------------------
__global__ void clock_sampler_kernel(long long* __restrict__ buf, int ns, const unsigned* flag, long long max_ticks,
                                     int* __restrict__ count) {
    if (threadIdx.x != 0) return;
    const long long w0 = wall_clock64();
    long long next = w0;
    int i = 0;
    while (i < ns) {
        const long long w = wall_clock64();
        if (w >= next) {
            buf[2 * i] = clock64();
            buf[2 * i + 1] = w;
            ++i;
            next = w + 2000;
        }
        if (__builtin_nontemporal_load(flag) != 0u) break;
        if (w - w0 > max_ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
    *count = i;
}
__global__ void wall_marker_kernel(long long* out) { *out = wall_clock64(); }

void launch_clock_sampler(hipStream_t s, long long* buf, int ns, const unsigned* flag, long long max_ticks, int* count) {
    clock_sampler_kernel<<<1, 64, 0, s>>>(buf, ns, flag, max_ticks, count);
}
void launch_wall_marker(hipStream_t s, long long* out) { wall_marker_kernel<<<1, 1, 0, s>>>(out); }

// synthetic signatures with `nclasses` distinct non-zero values (measurement hook)
__global__ void fill_test_sig_kernel(int64_t len, int64_t nclasses, uint64_t* __restrict__ sig) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        uint64_t cls = sdpsr_fmix64((uint64_t)e * 0x9E3779B97F4A7C15ULL + 17) % (uint64_t)nclasses;
        sig[e] = sdpsr_fmix64(cls + 1) | 1ull;
    }
}
void launch_fill_test_sig(hipStream_t s, int64_t len, int64_t nclasses, uint64_t* sig) {
    fill_test_sig_kernel<<<(int)((len + 255) / 256 < 2048 ? (len + 255) / 256 : 2048), 256, 0, s>>>(len, nclasses, sig);
}


}  // namespace sdpsr
