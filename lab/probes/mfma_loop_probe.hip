// Probe: steady-state rate of the igemm main loop (128x128 tile, 4 wavefronts of 64x64, K slices of 128 bytes), one
// ingredient at a time.  FLAGS bit 0: fragments come from LDS (ds_read_b128) instead of staying in registers;
// bit 1: one s_barrier per slice; bit 2: the next slice is fetched with buffer_load ... lds (2-stage ring);
// bit 3: fragments of the whole slice are read up front (as igemm does) instead of step by step.
#include <hip/hip_runtime.h>
#include <cstdio>
using lds_ptr_t = __attribute__((address_space(3))) void*;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int FLAGS>
__global__ __launch_bounds__(256) void loop_kernel(const unsigned char* a, const unsigned char* w, unsigned a_bytes, unsigned w_bytes, int iters,
                                                   float* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int RB = 128, STAGE = 256 * RB;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, w_bytes, 0x00020000);
    const int rsub = lane >> 3, dkey = ((wave & 1) << 2) | (rsub >> 1), lslot = (lane & 7) ^ dkey;
    unsigned a_off[4], w_off[4];
    for (int i = 0; i < 4; ++i) {
        const unsigned row = (wave + 4 * i) * 8 + rsub;
        a_off[i] = ((blockIdx.x * 128u + row) * 4096u + lslot * 16u) % (a_bytes - 8192u);
        w_off[i] = (row * 4096u + lslot * 16u) % (w_bytes - 8192u);
    }
    const int fkey = (l31 >> 1) & 7;
    int foff[4];
    for (int s = 0; s < 4; ++s) foff[s] = l31 * RB + (((2 * s + hi) ^ fkey) << 4);
    for (int i = tid; i < 2 * STAGE / 16; i += 256) ((u32x4*)lds)[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    f32x16 acc[2][2];
    for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    u32x4 fp[4][2], fw[4][2];
    for (int s = 0; s < 4; ++s) for (int b = 0; b < 2; ++b) { fp[s][b] = u32x4{1u + s, 2u, 3u, 4u + b}; fw[s][b] = u32x4{5u, 6u + s, 7u + b, 8u}; }
    for (int c = 0; c < iters; ++c) {
        if constexpr (FLAGS & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (FLAGS & 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        const unsigned char* a_s = lds + (c & 1) * STAGE;
        const unsigned char* b_s = a_s + 128 * RB;
        if constexpr ((FLAGS & 1) && (FLAGS & 8)) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    fp[s][b] = *(const u32x4*)(a_s + (wm * 64 + b * 32) * RB + foff[s]);
                    fw[s][b] = *(const u32x4*)(b_s + (wn * 64 + b * 32) * RB + foff[s]);
                }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr ((FLAGS & 1) && !(FLAGS & 8)) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    fp[s][b] = *(const u32x4*)(a_s + (wm * 64 + b * 32) * RB + foff[s]);
                    fw[s][b] = *(const u32x4*)(b_s + (wn * 64 + b * 32) * RB + foff[s]);
                }
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[s][x]), __builtin_bit_cast(bf16x8, fp[s][y]), acc[x][y], 0, 0, 0);
            if constexpr (FLAGS & 4) {
                unsigned char* st = lds + ((c + 1) & 1) * STAGE;
                const unsigned koff = ((unsigned)(c + 1) & 31u) * 128u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, (lds_ptr_t)(st + (wave + 4 * s) * 1024), 16, a_off[s] + koff, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(st + 128 * RB + (wave + 4 * s) * 1024), 16, w_off[s] + koff, 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
    for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int r = 0; r < 16; ++r) t += acc[x][y][r];
    if (t == 123.456f) sink[blockIdx.x * 256 + tid] = t;
}

template <int FLAGS>
static void run(const char* name, int blocks, const unsigned char* a, const unsigned char* w, unsigned ab, unsigned wb, float* sink) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)loop_kernel<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    loop_kernel<FLAGS><<<blocks, 256, 65536>>>(a, w, ab, wb, 50, sink);
    hipEventRecord(e0);
    loop_kernel<FLAGS><<<blocks, 256, 65536>>>(a, w, ab, wb, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * iters * 128.0 * 128.0 * 64.0 * 2.0;
    printf("%-58s blocks %4d: %7.1f TFLOP/s  %6.0f cycles/slice/WG @2.4GHz  feed %5.1f B/clk/CU\n", name, blocks, fl / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / iters, (FLAGS & 4) ? (double)blocks * iters * 32768.0 / (ms * 1e-3) / 256 / 2.4e9 : 0.0);
}

int main() {
    float* sink; hipMalloc(&sink, 4096 * 256 * 4);
    unsigned wb = 8u << 20;
    unsigned char *a, *w; hipMalloc(&a, 256u << 20); hipMalloc(&w, wb); hipMemset(a, 0x3c, 256u << 20); hipMemset(w, 0x3c, wb);
    for (unsigned ab : {16u << 20, 256u << 20})
    for (int blocks : {256, 512}) {
        printf("-- pixel operand window %u MiB (%s)\n", ab >> 20, ab <= (32u << 20) ? "L2-resident" : "HBM / MALL");
        run<0>("MFMA only (operands in registers)", blocks, a, w, ab, wb, sink);
        run<1>("+ ds_read_b128 fragments, step by step", blocks, a, w, ab, wb, sink);
        run<9>("+ ds_read_b128 fragments, whole slice up front", blocks, a, w, ab, wb, sink);
        run<11>("+ barrier per slice (slice up front)", blocks, a, w, ab, wb, sink);
        run<3>("+ barrier per slice (step by step)", blocks, a, w, ab, wb, sink);
        run<15>("+ LDS-DMA of next slice, 2-stage ring (slice up front)", blocks, a, w, ab, wb, sink);
        run<7>("+ LDS-DMA of next slice, 2-stage ring (step by step)", blocks, a, w, ab, wb, sink);
        run<4>("MFMA (registers) + LDS-DMA only", blocks, a, w, ab, wb, sink);
    }
    return 0;
}
