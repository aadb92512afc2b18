// Microbenchmark: can one wave overlap independent VALU work with its own MFMAs (gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KV, int DEP>
__global__ void k(float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x;
    float a = lane * 0.001f, b = 1.0001f;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = lane + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 1000; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            if (DEP) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
            else if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < KV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j % 12]) : "v"(b));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc0.x + acc1.x;
    for (int i = 0; i < 12; ++i) s += v[i];
    out[lane] = s;
    if (lane == 0) cyc[0] = (t1 - t0);
}
template <int KV, int DEP>
void run(float* d, unsigned long long* c) {
    hipLaunchKernelGGL((k<KV, DEP>), dim3(1), dim3(64), 0, 0, d, c);
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("KV=%2d dep=%d: %.1f memtime-ticks per (MFMA + KV VALU)\n", KV, DEP, h / 8000.0);
}
int main() {
    float* d; unsigned long long* c; hipMalloc(&d, 256); hipMalloc(&c, 8);
    run<0, 0>(d, c); run<0, 0>(d, c); run<2, 0>(d, c); run<4, 0>(d, c); run<6, 0>(d, c); run<8, 0>(d, c); run<12, 0>(d, c); run<16, 0>(d, c);
    run<0, 1>(d, c); run<4, 1>(d, c); run<8, 1>(d, c); run<12, 1>(d, c);
    return 0;
}
