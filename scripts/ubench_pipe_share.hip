// Diagnostic: do fp32 MFMA (v_mfma_f32_16x16x4_f32) and fp32 VALU share an execution pipe on one SIMD of gfx950?
// One 512-thread workgroup per CU = two waves per SIMD (waves w and w + 4 share SIMD placement by the 0->2->1->3 cyclic
// order; we measure all role assignments).  Roles: M = a chain-free stream of MFMAs on 4 accumulators, V = independent v_fma_f32.
// mode 0: all waves M;  1: all waves V;  2: waves 0-3 M, waves 4-7 V;  3: even waves M, odd waves V;  4: only waves 0-3 run M;  5: only waves 0-3 run V
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k(int mode, int iters, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    int role = 0;   // 0 idle, 1 M, 2 V
    if (mode == 0) role = 1;
    else if (mode == 1) role = 2;
    else if (mode == 2) role = wave < 4 ? 1 : 2;
    else if (mode == 3) role = (wave & 1) ? 2 : 1;
    else if (mode == 4) role = wave < 4 ? 1 : 0;
    else if (mode == 5) role = wave < 4 ? 2 : 0;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
    } else if (role == 2) {
        for (int i = 0; i < iters; ++i) {   // 32 independent FMAs per trip
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v0 = __builtin_fmaf(v0, y, x); v1 = __builtin_fmaf(v1, y, x); v2 = __builtin_fmaf(v2, y, x); v3 = __builtin_fmaf(v3, y, x);
                v4 = __builtin_fmaf(v4, y, x); v5 = __builtin_fmaf(v5, y, x); v6 = __builtin_fmaf(v6, y, x); v7 = __builtin_fmaf(v7, y, x);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = a0.x + a1.y + a2.z + a3.w + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
int main() {
    unsigned long long* d; float* s;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&s, 256 * 512 * 4);
    const int iters = 4096;
    for (int mode = 0; mode < 6; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, d, s);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, d + 8 * 17, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d cycles per trip (4 MFMA = 128 pipe cycles | 32 FMA): ", mode);
        for (int w = 0; w < 8; ++w) printf(" w%d %.1f", w, (double)h[w] / iters);
        printf("\n");
    }
    return 0;
}
