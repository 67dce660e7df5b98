// Latency microbenchmarks for gfx950: one wavefront, dependent chains; cycles per operation from s_memtime / clock64.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/lat.hip -o build_ab/lat && build_ab/lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N 2048
template <int MODE>
__global__ void k(double *out, long long *cyc, double a, double b, int active_lanes) {
    if ((int)threadIdx.x >= active_lanes) return;
    double x = a + threadIdx.x * 1e-9, y = b;
    int lane = threadIdx.x;
    long long t0 = wall_clock64();
    long long c0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N / 16; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (MODE == 0) x = __builtin_fma(x, y, a);                       // dependent fp64 fma
            if (MODE == 1) x = x * y;                                        // dependent fp64 mul
            if (MODE == 2) x = x + y;                                        // dependent fp64 add
            if (MODE == 3) x = __builtin_amdgcn_rcp(x) + y;                  // rcp + add
            if (MODE == 4) x = __builtin_amdgcn_rsq(x) + y;                  // rsq + add
            if (MODE == 5) { int lo = __double2loint(x), hi = __double2hiint(x);   // dpp quad_perm on both halves + add
                             lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true);
                             x = __hiloint2double(hi, lo) + y; }
            if (MODE == 6) { int lo = __double2loint(x), hi = __double2hiint(x);   // ds_bpermute of both halves + add
                             lo = __builtin_amdgcn_ds_bpermute((lane ^ 1) << 2, lo); hi = __builtin_amdgcn_ds_bpermute((lane ^ 1) << 2, hi);
                             x = __hiloint2double(hi, lo) + y; }
            if (MODE == 7) x = (x < a) ? x + y : x - y;                      // compare + select chain
            if (MODE == 8) x = __builtin_fmin(x, y) + a;                     // min + add
            if (MODE == 9) { float f = (float)x; f = f * 1.0001f + 1.0f; x = f; }   // cvt round trip + f32 fma
            if (MODE == 10) { if (__builtin_amdgcn_ballot_w64(x > 1e300)) break; x = x + y; }   // VALU -> SALU branch each step
        }
    }
    long long c1 = clock64();
    long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}

template <int MODE> void run(const char *name, int lanes, int ops_per_step) {
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16);
    long long h[2];
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 0.9999999, lanes);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-34s lanes %2d: %7.2f clock64 / step, %7.2f wall-clock ticks / step (%d dependent ops per step)\n", name, lanes, (double)h[0] / N,
           (double)h[1] / N, ops_per_step);
    hipFree(out); hipFree(cyc);
}

// throughput of INDEPENDENT operations issued by one wavefront: eight interleaved chains
template <int MODE>
__global__ void kt(double *out, long long *cyc, double a, double b) {
    double x[8];
    for (int j = 0; j < 8; j++) x[j] = a + threadIdx.x * 1e-9 + j;
    const double y = b;
    long long c0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N / 16; i++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (MODE == 0) x[j] = __builtin_fma(x[j], y, a);
                if (MODE == 1) x[j] = x[j] * y;
                if (MODE == 2) x[j] = x[j] + y;
                if (MODE == 3) x[j] = __builtin_fmin(x[j], y + j);
                if (MODE == 4) x[j] = (x[j] < a) ? y : x[j];   // cmp + 2 cndmask
                if (MODE == 5) { int lo = __double2loint(x[j]), hi = __double2hiint(x[j]);
                                 lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true);
                                 x[j] = __hiloint2double(hi, lo); }
                if (MODE == 6) { float f = (float)x[j]; x[j] = (double)(f * 1.0001f); }
                if (MODE == 7) { int lo = __double2loint(x[j]); lo = lo * 3 + 1; x[j] = __hiloint2double(__double2hiint(x[j]), lo); }   // int ops
                if (MODE == 8) x[j] = __builtin_amdgcn_rcp(x[j]);
            }
    }
    long long c1 = clock64();
    double sum = 0;
    for (int j = 0; j < 8; j++) sum += x[j];
    out[threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
template <int MODE> void runt(const char *name, int instr_per_op) {
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16);
    long long h[2];
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(kt<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 0.9999999);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("throughput %-30s %7.2f cycles per op (%d instructions per op)\n", name, (double)h[0] / N, instr_per_op);
    hipFree(out); hipFree(cyc);
}

int main() {
    runt<0>("fma f64", 1); runt<1>("mul f64", 1); runt<2>("add f64", 1); runt<3>("min f64", 1); runt<4>("cmp f64 + 2 cndmask", 3);
    runt<5>("2 dpp mov", 2); runt<6>("cvt, mul f32, cvt", 3); runt<7>("int mad (v_mad_u32_u24 or mul+add)", 2); runt<8>("rcp f64", 1);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, shader clock %d kHz\n", rate, clk);
    for (int lanes : {64, 16, 1}) {
        run<0>("fma f64", lanes, 1);
        run<1>("mul f64", lanes, 1);
        run<2>("add f64", lanes, 1);
        run<3>("rcp f64 + add", lanes, 2);
        run<4>("rsq f64 + add", lanes, 2);
        run<5>("dpp mov x2 + add", lanes, 2);
        run<6>("ds_bpermute x2 + add", lanes, 2);
        run<7>("cmp + 2 adds + select", lanes, 3);
        run<8>("min + add", lanes, 2);
        run<9>("cvt f64->f32, fma f32, cvt back", lanes, 3);
        run<10>("cmp + ballot branch + add", lanes, 2);
    }
    return 0;
}
