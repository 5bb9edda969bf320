#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ int dppi(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, 0xF, false); }
// mode 0: fma f64 (8 independent chains); 1: dpp wave_shl; 2: dpp row_shr:1; 3: permlane32_swap; 4: dpp quad_perm; 5: mul f64
template <int MODE> __global__ void ub(double* o, int iters) {
    double a[8]; int b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001 + i; b[i] = threadIdx.x + i; }
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = a[i] * 1.0000001 + 0.5;
            if (MODE == 5) a[i] = a[i] * 1.0000001;
            if (MODE == 1) b[i] = dppi<0x130>(b[i], b[i]);
            if (MODE == 2) b[i] = dppi<0x111>(b[i], b[i]);
            if (MODE == 4) b[i] = dppi<0xB1>(b[i], b[i]);
            if (MODE == 3) { auto r = __builtin_amdgcn_permlane32_swap(b[i], b[i], false, false); b[i] = r[0] + r[1]; }
        }
    }
    long long t1 = clock64();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    o[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) o[100000] = (double)(t1 - t0);
}
int main() {
    double* d; hipMalloc(&d, 8 * 200000);
    const int iters = 20000;
    const char* names[] = {"fma_f64", "dpp wave_shl:1", "dpp row_shr:1", "permlane32_swap+add", "dpp quad_perm", "mul_f64"};
    for (int grid : {1, 1024, 2048, 4096}) for (int m = 0; m < 6; ++m) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(ub<0>, dim3(grid), dim3(64), 0, 0, d, iters);
            if (m == 1) hipLaunchKernelGGL(ub<1>, dim3(grid), dim3(64), 0, 0, d, iters);
            if (m == 2) hipLaunchKernelGGL(ub<2>, dim3(grid), dim3(64), 0, 0, d, iters);
            if (m == 3) hipLaunchKernelGGL(ub<3>, dim3(grid), dim3(64), 0, 0, d, iters);
            if (m == 4) hipLaunchKernelGGL(ub<4>, dim3(grid), dim3(64), 0, 0, d, iters);
            if (m == 5) hipLaunchKernelGGL(ub<5>, dim3(grid), dim3(64), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double ticks; hipMemcpy(&ticks, d + 100000, 8, hipMemcpyDeviceToHost);
        printf("grid %4d %-22s %.3f ns per instr (event), %.2f clock64 ticks per instr\n", grid, names[m], ms * 1e6 / (iters * 8.0), ticks / (iters * 8.0));
    }
    return 0;
}
