// Issue cost of the FP64 vector instructions the traversal is made of (one wave per SIMD, 8
// independent chains, s_memtime around an unrolled loop).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 256
template <int OP>
__global__ void k(double* out, const float* fin, long long* cyc, double seed) {
    double a[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; f[i] = fin[(threadIdx.x + i) & 63]; }
    double b = seed * 0.5 + 1.0, c = seed * 0.25 + 0.5;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
            if (OP == 5) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
            if (OP == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (OP == 7) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[i]));
            if (OP == 8) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(f[(i + 1) & 7]) : "vcc");
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, double* out, float* fin, long long* cyc) {
    for (int waves = 1; waves <= 4; waves *= 2) {   // 1 block of 64*4*waves threads on one CU: waves per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256 * waves), 0, 0, out, fin, cyc, 1.5);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost);
        printf("%-16s waves/SIMD %d: %.2f clk per instruction per wave (s_memtime ticks)\n", name, waves, (double)c / (N_ITER * 8));
    }
}
int main() {
    double* out; float* fin; long long* cyc;
    hipMalloc(&out, 8192 * sizeof(double)); hipMalloc(&fin, 64 * sizeof(float)); hipMalloc(&cyc, 64 * sizeof(long long));
    std::vector<float> h(64, 1.25f); hipMemcpy(fin, h.data(), 256, hipMemcpyHostToDevice);
    run<0>("v_mul_f64", out, fin, cyc); run<1>("v_add_f64", out, fin, cyc); run<2>("v_fma_f64", out, fin, cyc);
    run<3>("v_min_f64", out, fin, cyc); run<4>("v_cvt_f64_f32", out, fin, cyc); run<5>("v_rcp_f64", out, fin, cyc);
    run<6>("v_mul_f32", out, fin, cyc); run<7>("v_sqrt_f64", out, fin, cyc); run<8>("v_cmp_lt_f64", out, fin, cyc);
    run<9>("v_cndmask_b32", out, fin, cyc);
    return 0;
}
