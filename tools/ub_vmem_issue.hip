// What does a vector-memory instruction cost a CU, by access width?  One 1024-thread workgroup per CU (16 wavefronts, as the tall window-panel
// kernel); every wavefront issues NL lane-consecutive loads of 2 / 4 / 8 / 16 bytes per lane back to back, waits for them, repeats.  The
// buffer is small (L2 / Infinity-Cache resident after the first pass) so that the rate is the CU's, not HBM's.
// hipcc --offload-arch=gfx950 -O3 tools/ub_vmem_issue.hip -o /tmp/ub_vmem && /tmp/ub_vmem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
template <class T, int NL>
__global__ __launch_bounds__(1024) void k(const T* __restrict__ buf, size_t nelem, int reps, double* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    size_t base = ((size_t)blockIdx.x * 16 + wv) * 64 * NL;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        T v[NL];
#pragma unroll
        for (int q = 0; q < NL; ++q) v[q] = __builtin_nontemporal_load(buf + (base + (size_t)q * 64 + lane) % nelem);
#pragma unroll
        for (int q = 0; q < NL; ++q) acc += (double)reinterpret_cast<const unsigned char*>(&v[q])[0];
        base += (size_t)gridDim.x * 16 * 64 * NL;
    }
    if (acc == 12345.678) out[0] = acc;
}
template <class T, int NL>
void run(const char* name, void* d, size_t bytes, double* out) {
    const int reps = 200;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<T, NL>), dim3(256), dim3(1024), 0, 0, (const T*)d, bytes / sizeof(T), 10, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<T, NL>), dim3(256), dim3(1024), 0, 0, (const T*)d, bytes / sizeof(T), reps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)reps * NL * 16;
    printf("%-10s %2d loads/wave/round  %8.3f ms  %7.1f ns per wave-instruction per CU  %7.1f GB/s per CU\n", name, NL, ms, 1e6 * ms / instr_per_cu,
           instr_per_cu * 64 * sizeof(T) / (ms * 1e-3) / 1e9);
}
// the slice loads of one tall window-panel segment per wavefront: 8 steps of values + 8 steps of 16-bit offsets + 4 row words, as the kernel issues them
// (20 instructions), against the same bytes in wide loads (4 x 16-byte values, 1 x 16-byte packed offsets + row word: 5 instructions)
template <int MODE>
__global__ __launch_bounds__(1024) void kmix(const char* __restrict__ buf, size_t nbytes, int reps, double* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    size_t base = ((size_t)blockIdx.x * 16 + wv) * 8192;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        const char* p = buf + base % (nbytes - 8192);
        if (MODE == 0) {
            double v[8]; unsigned short c[8], w[4];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load(reinterpret_cast<const double*>(p) + q * 64 + lane);
#pragma unroll
            for (int q = 0; q < 8; ++q) c[q] = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(p + 4096) + q * 64 + lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(p + 5120) + q * 64 + lane);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q] * (double)c[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += (double)w[q];
        } else {
            v2d v[4]; typedef unsigned u4 __attribute__((ext_vector_type(4))); u4 c;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p) + q * 64 + lane);
            c = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p + 4096) + lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += v[q].x * (double)(c[q] & 0xFFFFu) + v[q].y * (double)(c[q] >> 16);
        }
        base += (size_t)gridDim.x * 16 * 8192;
    }
    if (acc == 12345.678) out[0] = acc;
}
template <int MODE>
void runmix(const char* name, void* d, size_t bytes, double* out) {
    const int reps = 200;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((kmix<MODE>), dim3(256), dim3(1024), 0, 0, (const char*)d, bytes, 10, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((kmix<MODE>), dim3(256), dim3(1024), 0, 0, (const char*)d, bytes, reps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-40s %8.3f ms  %6.2f us per round (16 wavefronts x one segment's slice loads)\n", name, ms, 1e3 * ms / reps);
}
int main() {
    size_t bytes = 24u << 20;
    void* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
    double* out; hipMalloc(&out, 8);
    run<unsigned short, 16>("ushort", d, bytes, out);
    run<unsigned, 16>("dword", d, bytes, out);
    run<double, 16>("dwordx2", d, bytes, out);
    run<v2d, 16>("dwordx4", d, bytes, out);
    run<unsigned short, 8>("ushort", d, bytes, out);
    run<double, 8>("dwordx2", d, bytes, out);
    run<v2d, 8>("dwordx4", d, bytes, out);
    runmix<0>("8 x 8 B + 8 x 2 B + 4 x 2 B (20 instr)", d, bytes, out);
    runmix<1>("4 x 16 B + 1 x 16 B packed (5 instr)", d, bytes, out);
    return 0;
}
