// Microbenchmark: v_mfma_f64_16x16x4_f64 issue rate with 1 / 2 / 4 independent accumulator chains (gfx950), one wave and four waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ void __launch_bounds__(256, 1) k(double* out, unsigned long long* cyc, double a, double b) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f64x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f64x4{0, 0, 0, 0};
    const double av = a + lane, bv = b - lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 1000; ++rep) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u % NCH] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[u % NCH], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].w;
    out[tid] = s;
    if (lane == 0) cyc[wave] = t1 - t0;
}
template <int NCH>
void run(double* d, unsigned long long* c, int threads) {
    hipLaunchKernelGGL((k<NCH>), dim3(1), dim3(threads), 0, 0, d, c, 1.0, 2.0);
    unsigned long long h[4];
    hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
    printf("chains %d waves %d: %.1f ticks per f64 MFMA\n", NCH, threads / 64, h[0] / 8000.0);
}
int main() {
    double* d; unsigned long long* c;
    hipMalloc(&d, 4096); hipMalloc(&c, 64);
    for (int threads : {64, 256}) { run<1>(d, c, threads); run<2>(d, c, threads); run<4>(d, c, threads); }
    return 0;
}
