// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950, and whether fp64 VALU FMAs of a second wavefront on the same SIMD run beside it.
//   hipcc --offload-arch=gfx950 -O3 tools/ub_mfma_f64.hip -o /tmp/ub_mfma && /tmp/ub_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void mfma_chain(double* out, long long* cyc, int iters) {
    v4d acc[NACC];
    const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// wavefronts 0..3 (one per SIMD) issue MFMAs, wavefronts 4..7 (their SIMD partners) issue independent fp64 FMAs
__global__ void mixed(double* out, long long* cyc, int iters, int mode) {
    const int w = threadIdx.x >> 6;
    const bool do_mfma = (mode == 0) || (mode == 2 && w < 4);
    const bool do_valu = (mode == 1) || (mode == 2 && w >= 4);
    v4d acc[4];
    double f[8];
    const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-6;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = q;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (do_mfma) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    if (do_valu) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = __builtin_fma(f[q], b, a);         // 32 FMA instructions per iteration
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += f[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * 1024 * 1024);
    hipMalloc(&cyc, sizeof(long long) * 8192);
    std::vector<long long> h(8192);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(NACC, GRID)                                                                                     \
    {                                                                                                         \
        mfma_chain<NACC><<<GRID, 64>>>(out, cyc, 10);                                                        \
        hipEventRecord(e0); mfma_chain<NACC><<<GRID, 64>>>(out, cyc, iters); hipEventRecord(e1);            \
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);                                 \
        hipMemcpy(h.data(), cyc, sizeof(long long) * 8, hipMemcpyDeviceToHost);                             \
        printf("mfma_f64_16x16x4: %2d independent accumulators, grid %4d x 1 wavefront: %.1f memtime ticks per MFMA, %.1f ns per MFMA (event)\n", \
               NACC, GRID, (double)h[0] / (iters * NACC), 1e6 * ms / (iters * NACC));                        \
    }
    RUN(1, 1) RUN(2, 1) RUN(4, 1) RUN(8, 1) RUN(4, 1024) RUN(8, 1024) RUN(4, 2048)
    for (int mode = 0; mode < 3; ++mode) {
        mixed<<<256, 512>>>(out, cyc, 10, mode);
        hipEventRecord(e0); mixed<<<256, 512>>>(out, cyc, iters, mode); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, sizeof(long long) * 8, hipMemcpyDeviceToHost);
        printf("512-thread workgroup per CU, mode %d (%s): kernel %.3f ms; ticks: wave0 %lld wave4 %lld  (per iteration: 4 MFMAs / 32 FMAs)\n", mode,
               mode == 0 ? "all 8 wavefronts MFMA" : mode == 1 ? "all 8 wavefronts fp64 FMA" : "wavefronts 0-3 MFMA, 4-7 fp64 FMA",
               ms, h[0] / iters, h[4] / iters);
    }
    return 0;
}
