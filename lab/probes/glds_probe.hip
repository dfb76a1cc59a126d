// Probe: does buffer_load_dwordx4 ... lds (LDS-DMA) zero-fill out-of-range lanes, and is the LDS image lane-linear?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

__global__ void probe(const unsigned int* src, unsigned int nbytes, unsigned int* out) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[2 * 64 * 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2 * 64 * 4; i += 64) lds[i] = 0xffffffffu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // lanes fetch in REVERSE order to show the source address is per-lane while the LDS slot is lane-linear;
    // lanes >= 48 point outside the buffer
    unsigned int voff = (lane < 48) ? (unsigned)(47 - lane) * 16u : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 64 * 4), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 2 * 64 * 4; i += 64) out[i] = lds[i];
}

int main() {
    std::vector<unsigned int> h(64 * 4);
    for (int i = 0; i < 64 * 4; ++i) h[i] = 1000 + i;
    unsigned int *d, *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 2 * 64 * 4 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, 48 * 16, o);
    std::vector<unsigned int> r(2 * 64 * 4);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64 * 4; ++i) if (r[i] != 0xffffffffu) ++bad;                 // first half untouched
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
            unsigned int want = lane < 48 ? 1000 + (47 - lane) * 4 + j : 0u;
            if (r[64 * 4 + lane * 4 + j] != want) { if (bad < 8) printf("lane %d j %d got %u want %u\n", lane, j, r[64 * 4 + lane * 4 + j], want); ++bad; }
        }
    printf("glds probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK: lane-linear image, OOB lanes zero-filled", bad);
    return bad != 0;
}
