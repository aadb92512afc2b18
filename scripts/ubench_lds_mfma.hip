// Microbenchmark: tile products out of LDS (one b128 per lane and operand) on 1 and 4 waves of a workgroup (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <int MODE, int NT>   // MODE 0: load then use; 1: software pipelined (sched_barrier); 2: no loads.  NT tiles per K
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int J) {
    __shared__ __attribute__((aligned(16))) float Tl[120 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 120 * 256; i += blockDim.x) Tl[i] = 0.001f * (i & 1023);
    __syncthreads();
    f32x4 a[4], b[4];
    for (int t = 0; t < 4; ++t) a[t] = b[t] = f32x4{0, 0, 0, 0};
    const float* pj = Tl + (wave * 7) * 256 + 4 * lane;
    const float* pi = Tl + (40 + wave * 11) * 256 + 4 * lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 3) {   // ping-pong: two register sets, the loads of step K+1 issued before the MFMAs of step K, no copies
        for (int rep = 0; rep < 200; ++rep) {
            f32x4 tj0 = lds4(pj), ti0[4], tj1 = tj0, ti1[4];
#pragma unroll
            for (int t = 0; t < NT; ++t) ti0[t] = ti1[t] = lds4(pi + t * 4096);
            int K = 0;
            for (; K + 1 < J; K += 2) {
                tj1 = lds4(pj + (K + 1) * 256);
#pragma unroll
                for (int t = 0; t < NT; ++t) ti1[t] = lds4(pi + t * 4096 + (K + 1) * 256);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    a[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj0.x, ti0[t].x, a[t], 0, 0, 0);
                    b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj0.y, ti0[t].y, b[t], 0, 0, 0);
                    a[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj0.z, ti0[t].z, a[t], 0, 0, 0);
                    b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj0.w, ti0[t].w, b[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int K2 = (K + 2 < J) ? K + 2 : K;
                tj0 = lds4(pj + K2 * 256);
#pragma unroll
                for (int t = 0; t < NT; ++t) ti0[t] = lds4(pi + t * 4096 + K2 * 256);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    a[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj1.x, ti1[t].x, a[t], 0, 0, 0);
                    b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj1.y, ti1[t].y, b[t], 0, 0, 0);
                    a[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj1.z, ti1[t].z, a[t], 0, 0, 0);
                    b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj1.w, ti1[t].w, b[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (K < J) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    a[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj0.x, ti0[t].x, a[t], 0, 0, 0);
                    b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj0.y, ti0[t].y, b[t], 0, 0, 0);
                    a[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj0.z, ti0[t].z, a[t], 0, 0, 0);
                    b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj0.w, ti0[t].w, b[t], 0, 0, 0);
                }
            }
        }
    } else
    for (int rep = 0; rep < 200; ++rep) {
        f32x4 tj = lds4(pj), ti[4];
        for (int t = 0; t < NT; ++t) ti[t] = lds4(pi + t * 4096);
        for (int K = 0; K < J; ++K) {
            f32x4 tjn = tj, tin[4];
            for (int t = 0; t < NT; ++t) tin[t] = ti[t];
            if (MODE == 1) {
                const int Kn = (K + 1 < J) ? K + 1 : K;
                tjn = lds4(pj + Kn * 256);
#pragma unroll
                for (int t = 0; t < NT; ++t) tin[t] = lds4(pi + t * 4096 + Kn * 256);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 0) {
                tj = lds4(pj + K * 256);
#pragma unroll
                for (int t = 0; t < NT; ++t) ti[t] = lds4(pi + t * 4096 + K * 256);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                a[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj.x, ti[t].x, a[t], 0, 0, 0);
                b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj.y, ti[t].y, b[t], 0, 0, 0);
                a[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj.z, ti[t].z, a[t], 0, 0, 0);
                b[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tj.w, ti[t].w, b[t], 0, 0, 0);
            }
            if (MODE == 1) {
                __builtin_amdgcn_sched_barrier(0);
                tj = tjn;
#pragma unroll
                for (int t = 0; t < NT; ++t) ti[t] = tin[t];
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < 4; ++t) s += a[t].x + b[t].y;
    out[tid] = s;
    if (lane == 0) cyc[wave] = (t1 - t0);
}
template <int MODE, int NT>
void run(float* d, unsigned long long* c, int threads, int J) {
    hipLaunchKernelGGL((k<MODE, NT>), dim3(1), dim3(threads), 0, 0, d, c, J);
    unsigned long long h[4]; hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
    printf("mode %d tiles %d waves %d J=%2d: %.1f ticks per MFMA (wave 0), %.1f (wave %d)\n", MODE, NT, threads / 64, J, h[0] / (200.0 * J * NT * 4),
           h[threads / 64 - 1] / (200.0 * J * NT * 4), threads / 64 - 1);
}
int main() {
    float* d; unsigned long long* c; hipMalloc(&d, 4096); hipMalloc(&c, 64);
    for (int threads : {64, 256}) {
        run<2, 1>(d, c, threads, 8); run<0, 1>(d, c, threads, 8); run<1, 1>(d, c, threads, 8); run<3, 1>(d, c, threads, 8);
        run<2, 2>(d, c, threads, 8); run<0, 2>(d, c, threads, 8); run<1, 2>(d, c, threads, 8); run<3, 2>(d, c, threads, 8);
        run<2, 3>(d, c, threads, 8); run<0, 3>(d, c, threads, 8); run<3, 3>(d, c, threads, 8);
        run<2, 4>(d, c, threads, 8); run<0, 4>(d, c, threads, 8); run<1, 4>(d, c, threads, 8); run<3, 4>(d, c, threads, 8);
        run<0, 2>(d, c, threads, 3); run<3, 2>(d, c, threads, 3); run<0, 2>(d, c, threads, 5); run<3, 2>(d, c, threads, 5);
    }
    return 0;
}
