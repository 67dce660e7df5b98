// Workgroup dispatch microbenchmark: N workgroups that each spin for a fixed number of shader clocks; how long does the launch take for
// different workgroup sizes / LDS sizes?  (Why do 2048 single-wavefront workgroups of ~17 us each take 100 us in npp_gv_cells_kernel?)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dispatch.hip -o build_ab/dispatch && build_ab/dispatch
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin(long long ticks, int *sink) {
    extern __shared__ int lds[];
    const long long t0 = clock64();
    int v = threadIdx.x;
    while (clock64() - t0 < ticks) v = v * 3 + 1;
    if (v == 0x7fffffff) { lds[threadIdx.x] = v; sink[0] = lds[(threadIdx.x + 1) & 63]; }
}
// the same with ~150 live VGPRs (3 wavefronts per SIMD)
__global__ __launch_bounds__(64, 3) void spin_regs(long long ticks, int *sink, const float *in) {
    extern __shared__ int lds[];
    float r[140];
#pragma unroll
    for (int i = 0; i < 140; i++) r[i] = in[i * 64 + threadIdx.x];
    const long long t0 = clock64();
    while (clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 140; i++) r[i] = r[i] * 1.0001f + r[(i + 1) % 140];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 140; i++) s += r[i];
    if (s == 12345.f) { lds[threadIdx.x] = 1; sink[0] = lds[(threadIdx.x + 1) & 63]; }
}

int main() {
    int *sink; float *in;
    hipMalloc(&sink, 64); hipMalloc(&in, 140 * 64 * 4); hipMemset(in, 0, 140 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long ticks = 40000;   // ~17 us at 2.4 GHz
    for (int regs = 0; regs < 2; regs++)
        for (int block : {64, 256})
            for (int lds : {0, 12288, 32768})
                for (int nwg : {256, 2048, 8192}) {
                    const int grid = nwg * 64 / block;   // same number of wavefronts
                    float best = 1e9f;
                    for (int rep = 0; rep < 4; rep++) {
                        hipEventRecord(e0, 0);
                        if (regs) hipLaunchKernelGGL(spin_regs, dim3(grid), dim3(block), lds, 0, ticks, sink, in);
                        else hipLaunchKernelGGL(spin, dim3(grid), dim3(block), lds, 0, ticks, sink);
                        hipEventRecord(e1, 0);
                        hipEventSynchronize(e1);
                        float ms; hipEventElapsedTime(&ms, e0, e1);
                        if (rep && ms < best) best = ms;
                    }
                    printf("%s block %3d lds %5d B/wg  %5d wavefronts: %7.1f us\n", regs ? "150 VGPR" : "few VGPR", block, lds, nwg, best * 1e3f);
                }
    return 0;
}
