#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f2 acc[16]; f2 a = {1.0001f, 0.9999f}, b = {0.5f + threadIdx.x * 1e-6f, 0.25f};
    for (int i = 0; i < 16; ++i) acc[i] = (f2){(float)i, (float)i + 1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][0]) : "v"(a[0]), "v"(b[0])); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][1]) : "v"(a[0]), "v"(b[1])); }
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, grid = 256 * 8;   // 8 WG of 4 waves per CU = 8 waves per SIMD
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fma = (double)grid * 256 * iters * 16 * 2;
            if (rep) printf("mode %d: %.3f ms, %.1f TFLOP/s (f32 FMA = 2 flop)\n", mode, ms, fma * 2 / ms / 1e9);
        }
    }
    return 0;
}
