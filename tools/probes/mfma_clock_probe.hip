// Probe: what the int8 matrix pipe delivers on this GPU with nothing else going on, and at which
// shader clock.  Every wave issues ITER x 16 independent v_mfma_i32_32x32x32_i8 (16 accumulators,
// no memory traffic); the shader clock is clock64() (s_memtime) against wall_clock64() (100 MHz).
//   hipcc -O3 --offload-arch=gfx950 mfma_clock_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) probe(int iters, long long* out, int* sink) {
    v16i acc[16];
    for (int a = 0; a < 16; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0;
    v4i x = {(int)threadIdx.x, 2, 3, 4}, y = {5, 6, (int)blockIdx.x, 8};
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 16; ++a) acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, acc[a], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    int s = 0;
    for (int a = 0; a < 16; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 0x7fffffff) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = w1 - w0;
    }
}

int main() {
    const int iters = 20000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        const int wgs = 256 * wgs_per_cu, waves = wgs * 4;
        long long* d;
        int* sink;
        hipMalloc(&d, waves * 16);
        hipMalloc(&sink, 64);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        probe<<<wgs, 256>>>(100, d, sink);
        hipEventRecord(e0);
        probe<<<wgs, 256>>>(iters, d, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(waves * 2);
        hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
        double cs = 0, ws = 0;
        for (int w = 0; w < waves; ++w) cs += h[2 * w], ws += h[2 * w + 1];
        const double ops = (double)waves * iters * 16 * 2.0 * 32 * 32 * 32;
        printf("waves/SIMD %d: %.3f ms, %.2f POP/s, shader clock %.0f MHz (clock64 / wall_clock64 at 100 MHz), %.1f clk per MFMA per SIMD\n",
               wgs_per_cu, ms, ops / ms / 1e12, cs / ws * 100.0, cs / waves / ((double)iters * 16) / wgs_per_cu);
        hipFree(d);
        hipFree(sink);
    }
    return 0;
}
