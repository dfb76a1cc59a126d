// Issue cost of plain / packed / transcendental fp32 VALU instructions on gfx950, one or two waves per SIMD:
//   hipcc --offload-arch=gfx950 -O3 lab/probes/valu_rate_probe.hip -o lab/probes/valu_rate_probe && gpurun -- lab/probes/valu_rate_probe
// One workgroup of 256 (1 wave / SIMD) or 512 (2 waves / SIMD) threads on one CU runs ITER x 32 independent instructions of one kind
// (8 independent chains, so dependency latency is covered); s_memtime around the loop -> cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2000
template <int KIND>
__global__ void probe(float* out, unsigned long long* clk, float seed) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    typedef float f2 __attribute__((ext_vector_type(2)));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (KIND == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            } else if constexpr (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            } else if constexpr (KIND == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            } else if constexpr (KIND == 3) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    f2 v = {a[i], a[i + 1]}, s = {seed, seed};
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(s));
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(s));
                    a[i] = v[0]; a[i + 1] = v[1];
                }
            } else if constexpr (KIND == 4) {        // the packed SiLU body: 2 pk_mul + 4 exp + 2 pk_add + 4 rcp + 2 pk_mul per 4 values (x2 = 8 values)
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    f2 v = {a[i], a[i + 1]}, s = {seed, seed}, t;
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(s));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(t[0]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(t[1]));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t) : "v"(s));
                    asm volatile("v_rcp_f32 %0, %0" : "+v"(t[0]));
                    asm volatile("v_rcp_f32 %0, %0" : "+v"(t[1]));
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(t));
                    a[i] = v[0]; a[i + 1] = v[1];
                }
            } else if constexpr (KIND == 5) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
            } else if constexpr (KIND == 6) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
            } else if constexpr (KIND == 7) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[threadIdx.x >> 6] = t1 - t0;
}
template <int KIND> void run(const char* name, int per_iter) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 4096); hipMalloc(&clk, 128);
    for (int threads : {256, 512, 1024}) {
        probe<KIND><<<1, threads>>>(out, clk, 1.0001f);
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, clk, 128, hipMemcpyDeviceToHost);
        double c = 0; int nw = threads / 64;
        for (int i = 0; i < nw; ++i) c += (double)h[i];
        c /= nw;
        // per SIMD: (threads / 256) waves each issued ITER * per_iter instructions in c cycles
        printf("%-28s %d wave(s)/SIMD: %.2f cycles per wave-instruction per wave, %.2f per SIMD\n", name, threads / 256, c / (ITER * (double)per_iter),
               c / (ITER * (double)per_iter * (threads / 256)));
    }
    hipFree(out); hipFree(clk);
}
int main() {
    run<0>("v_mul_f32", 32);
    run<6>("v_fma_f32", 32);
    run<3>("v_pk_mul_f32", 32);
    run<7>("v_cvt_pk_bf16_f32", 32);
    run<1>("v_exp_f32", 32);
    run<2>("v_rcp_f32", 32);
    run<5>("v_exp_f16", 32);
    run<4>("packed SiLU body (28/8 values)", 4 * 4 * 7);
    return 0;
}
