// Probe: cycles per v_mfma_f64_16x16x4_f64 on one SIMD as a function of the number of independent
// accumulators a wave rotates through (1, 2, 3, 4, 8, 16) and of the waves per SIMD (1, 2); no memory
// traffic.  The shader clock is clock64() (s_memtime) against wall_clock64() (100 MHz).
//   hipcc -O3 --offload-arch=gfx950 mfma_f64_probe.hip -o /tmp/mfma_f64_probe && /tmp/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) probe(int iters, long long* out, double* sink) {
    v4d acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = v4d{0.0, 0.0, 0.0, 0.0};
    double x = 1.0 + threadIdx.x * 1e-3, y = 1.0 - blockIdx.x * 1e-6;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48 / NACC; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 4; ++r) s += acc[a][r];
    if (s == 0.123) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = w1 - w0;
    }
}

template <int NACC>
void run(int wgs_per_cu) {
    const int iters = 2000;
    const int wgs = 256 * wgs_per_cu, waves = wgs * 4;
    long long* d;
    double* sink;
    hipMalloc(&d, waves * 16);
    hipMalloc(&sink, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<NACC><<<wgs, 256>>>(10, d, sink);
    hipEventRecord(e0);
    probe<NACC><<<wgs, 256>>>(iters, d, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(waves * 2);
    hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
    double cs = 0, ws = 0;
    for (int w = 0; w < waves; ++w) cs += h[2 * w], ws += h[2 * w + 1];
    const double n_mfma = (double)iters * (48 / NACC) * NACC;
    const double flop = (double)waves * n_mfma * 2.0 * 16 * 16 * 4;
    printf("accumulators %2d, waves/SIMD %d: %.3f ms, %.1f TFLOP/s, shader clock %.0f MHz, %.1f clk per MFMA per SIMD\n", NACC,
           wgs_per_cu, ms, flop / ms / 1e9, cs / ws * 100.0, cs / waves / n_mfma / wgs_per_cu);
    hipFree(d);
    hipFree(sink);
}

int main() {
    for (int k = 1; k <= 2; ++k) {
        run<1>(k);
        run<2>(k);
        run<3>(k);
        run<4>(k);
        run<8>(k);
        run<16>(k);
    }
    return 0;
}
