// Probe: L2/MALL -> LDS feed rate of `buffer_load_dwordx4 ... lds` by request granularity (what bounds the igemm).
//   mode 0: each wave instruction fetches 16 segments of 64 B (segment stride = seg_stride bytes)  [ROWB = 64]
//   mode 1: each wave instruction fetches  8 segments of 128 B                                      [ROWB = 128]
//   mode 2: each wave instruction fetches 32 segments of 32 B                                       [Cin = 16 stem]
//   mode 3: one contiguous 1 KiB per wave instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using lds_ptr_t = __attribute__((address_space(3))) void*;

template <int MODE>
__global__ __launch_bounds__(256) void feed(const unsigned char* src, unsigned bytes, unsigned seg_stride, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[48 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    constexpr int LPS = MODE == 0 ? 4 : MODE == 1 ? 8 : MODE == 2 ? 2 : 64;    // lanes per segment
    constexpr int SEGB = LPS * 16;
    const unsigned segs_per_instr = 64 / LPS;
    // every block walks its own window so that consecutive blocks touch neighbouring data (like neighbouring tiles)
    unsigned base = (blockIdx.x * 4 + wave) * segs_per_instr * seg_stride;
    const unsigned lane_off = (lane / LPS) * seg_stride + (lane % LPS) * 16;
    const unsigned wrap = bytes - 64 * seg_stride - SEGB;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {                       // 4 instructions per wave per "slice" (16 KiB per block)
            unsigned off = base + lane_off;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + ((it % 3) * 16 + wave * 4 + k) * 1024), 16, off % wrap, 0, 0, 0);
            base += gridDim.x * 4 * segs_per_instr * seg_stride;
            if (base >= wrap) base -= wrap;
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)lds;
}

template <int MODE>
static void run(const char* name, const unsigned char* d, unsigned bytes, unsigned seg_stride, unsigned* sink) {
    const int blocks = 256 * 3, iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    feed<MODE><<<blocks, 256>>>(d, bytes, seg_stride, 20, sink);
    hipEventRecord(a);
    feed<MODE><<<blocks, 256>>>(d, bytes, seg_stride, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double gb = (double)blocks * iters * 16384 / 1e9;
    printf("%-34s window %4u MiB stride %4u: %7.1f GB/s  (%.1f B/clk/CU @2.1GHz)\n", name, bytes >> 20, seg_stride, gb / (ms * 1e-3),
           gb / (ms * 1e-3) * 1e9 / 256 / 2.1e9);
}

int main() {
    unsigned* sink; hipMalloc(&sink, 4096 * 4);
    for (unsigned mib : {8u, 64u, 1024u}) {
        unsigned bytes = mib << 20;
        unsigned char* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
        run<0>("64 B segments (half lines)", d, bytes, 128, sink);
        run<0>("64 B segments, dense", d, bytes, 64, sink);
        run<1>("128 B segments (full lines)", d, bytes, 128, sink);
        run<1>("128 B segments, stride 256", d, bytes, 256, sink);
        run<2>("32 B segments (quarter lines)", d, bytes, 32, sink);
        run<3>("1 KiB contiguous", d, bytes, 1024, sink);
        hipFree(d);
    }
    return 0;
}
