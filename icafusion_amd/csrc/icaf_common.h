// Shared device/host helpers for libicaf.so (gfx950 / CDNA4 only — no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <cstdio>
#include <cstdarg>
#include <atomic>
#include "../../include/icaf.h"

namespace icaf {

// ---- error plumbing -------------------------------------------------------------------------------------
std::string& last_error();
int fail(int code, const char* fmt, ...);
// probe knobs set through icaf_set_option (api.hip); 0 = the library's own choice
struct LibOptions { int detect_elementwise = 0, attn_qsplit = 0, sppf_vpb = 0; };
extern LibOptions g_opt;

#define ICAF_HIP(expr)                                                                        \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return icaf::fail(ICAF_ERR_HIP, "%s -> %s (%s:%d)", #expr, hipGetErrorString(_e), \
                              __FILE__, __LINE__);                                            \
    } while (0)

#define ICAF_LAUNCH_CHECK() ICAF_HIP(hipGetLastError())

// ---- vector / element types ---------------------------------------------------------------------------------
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);      // gfx950: v_cvt_pk_bf16_f32, round to nearest even
}
// two floats -> packed bf16x2 / f16x2 in one 32-bit word (lo in bits 0..15)
__device__ __forceinline__ unsigned int pack2_bf16(float lo, float hi) {
    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ unsigned int pack2_f16(float lo, float hi) {
    using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
    f16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float f16_to_f32(unsigned short h) {
    _Float16 v = __builtin_bit_cast(_Float16, h);
    return (float)v;
}
__device__ __forceinline__ unsigned short f32_to_f16(float f) {
    _Float16 v = (_Float16)f;
    return __builtin_bit_cast(unsigned short, v);
}

// Element traits: DT = ICAF_F32 / ICAF_BF16 / ICAF_F16.  VEC = elements per 16-byte vector.
template <int DT> struct Elem;
template <> struct Elem<ICAF_F32> {
    using type = float;
    static constexpr int VEC = 4;
    static constexpr int BYTES = 4;
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<ICAF_BF16> {
    using type = unsigned short;
    static constexpr int VEC = 8;
    static constexpr int BYTES = 2;
    static __device__ __forceinline__ float ld(const unsigned short* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(unsigned short* p, float v) { *p = f32_to_bf16(v); }
};
template <> struct Elem<ICAF_F16> {
    using type = unsigned short;
    static constexpr int VEC = 8;
    static constexpr int BYTES = 2;
    static __device__ __forceinline__ float ld(const unsigned short* p) { return f16_to_f32(*p); }
    static __device__ __forceinline__ void st(unsigned short* p, float v) { *p = f32_to_f16(v); }
};

// unpack / pack one 16-byte vector <-> VEC floats
template <int DT> __device__ __forceinline__ void unpack16(const u32x4& v, float* f);
template <> __device__ __forceinline__ void unpack16<ICAF_F32>(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
}
template <> __device__ __forceinline__ void unpack16<ICAF_BF16>(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void unpack16<ICAF_F16>(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = f16_to_f32((unsigned short)(v[i] & 0xffffu));
        f[2 * i + 1] = f16_to_f32((unsigned short)(v[i] >> 16));
    }
}
template <int DT> __device__ __forceinline__ u32x4 pack16(const float* f);
template <> __device__ __forceinline__ u32x4 pack16<ICAF_F32>(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(f[i]);
    return v;
}
template <> __device__ __forceinline__ u32x4 pack16<ICAF_BF16>(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2_bf16(f[2 * i], f[2 * i + 1]);
    return v;
}
template <> __device__ __forceinline__ u32x4 pack16<ICAF_F16>(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2_f16(f[2 * i], f[2 * i + 1]);
    return v;
}

// ---- MFMA step: consumes one 16-byte K-slice per lane of each operand ------------------------------------------
// Operand convention (both operands K-major in memory): for the step's 2*VEC consecutive K elements, lane-half
// hi = lane>>5 supplies elements [hi*VEC, hi*VEC+VEC) of row (lane&31).  D[i][j] += sum_k A[i][k] * B[j][k];
// D layout: j = lane&31, i = (r&3) + 8*(r>>2) + 4*hi for accumulator register r.
template <int DT> __device__ __forceinline__ void mma_step(f32x16& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma_step<ICAF_BF16>(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_step<ICAF_F16>(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_step<ICAF_F32>(f32x16& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
}

// x * sigmoid(x) with the hardware exp2 / rcp (each ~1 ulp): 5 VALU ops instead of ~25 for expf + IEEE division.
// For very negative x, exp2 overflows to +inf, rcp(inf) = 0 and the product is -0: the correct limit.
__device__ __forceinline__ float silu_f(float v) {
    const float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}
// Four values at once with the plain arithmetic on register PAIRS (v_pk_mul_f32 / v_pk_add_f32: two fp32 lanes per issue slot at the full
// rate; the transcendentals stay scalar).  Same operations in the same order as silu_f, value by value: the same bits — 13 issue slots
// for two values instead of 16.  (Left to itself the compiler packs the bias add and the final product but not the two middle steps.)
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void silu4_f(const float (&x)[4], float (&y)[4]) {
#ifdef ICAF_NO_PK_SILU
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = silu_f(x[e]);
#else
    const f32x2 c = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.0f, 1.0f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 v = {x[2 * h], x[2 * h + 1]};
        const f32x2 t = v * c;
        const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        const f32x2 d = one + e;
        const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const f32x2 o = v * r;
        y[2 * h] = o[0];
        y[2 * h + 1] = o[1];
    }
#endif
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
// GELU(erf) for the 16-bit kernels: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below half a unit of bf16 / f16),
// one hardware exp2 + one rcp + 7 FMAs instead of libm's erff (~100 instructions; measured: 13 k cycles per [64 x 128] hidden chunk,
// 60-75 % of the fused block kernel's MLP phase and most of the per-layer fc1 epilogue).  The fp32 parity build keeps erff.
__device__ __forceinline__ float gelu_fast_f(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.0f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * -1.4426950408889634f);
    const float erf_abs = __builtin_fmaf(-(poly * t), e, 1.0f);         // erf(|x|)
    const float erfv = v < 0.0f ? -erf_abs : erf_abs;
    return 0.5f * v * (1.0f + erfv);
}
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

inline hipStream_t S(icaf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- division by a launch-constant divisor ----------------------------------------------------------------------
// The element-per-thread kernels decompose a flat index into (b, y, x, c); with runtime divisors that is a chain of
// 64-bit divisions (~1000 instructions in detect_decode's ISA for 8 bytes of traffic).  FastDiv holds the round-up
// multiplier of Granlund & Montgomery for one divisor: q = (mulhi(n, m) + n) >> s, exact for every n < 2^31 (mulhi <= n,
// so the sum stays below 2^32).  Launchers use it when the flat index space fits 31 bits and fall back to the 64-bit
// kernel otherwise.
struct FastDiv {
    unsigned int d, m, s;
};
static inline FastDiv make_fastdiv(unsigned int d) {
    unsigned int l = 0;
    while ((1ull << l) < d) ++l;                                   // l = ceil(log2 d)
    const unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
    return FastDiv{d, (unsigned int)m, l};
}
__host__ __device__ __forceinline__ unsigned int fd_div(unsigned int n, FastDiv f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (__umulhi(n, f.m) + n) >> f.s;
#else
    return ((unsigned int)(((unsigned long long)n * f.m) >> 32) + n) >> f.s;      // host twin (tests/test_host_logic.py)
#endif
}
__host__ __device__ __forceinline__ void fd_divmod(unsigned int n, FastDiv f, unsigned int& q, unsigned int& r) {
    q = fd_div(n, f);
    r = n - q * f.d;
}

constexpr int ICAF_MAX_DEVICES = 64;      // per-device one-time kernel attribute flags (hipFuncSetAttribute is per device)

// One-time opt-in of ONE kernel instantiation to `bytes` of dynamic LDS, per DEVICE: hipFuncSetAttribute is per device, and a process may
// drive several GPUs (tests, one process per node).  Expands to a function-local static flag array, i.e. one per template / lambda
// instantiation of the enclosing launcher.  With GROW the largest size enabled so far is tracked instead of a flag (launchers whose LDS
// size depends on the launch).
#define ICAF_LDS_OPTIN(kern, bytes)                                                                                                  \
    do {                                                                                                                             \
        static std::atomic<int> _icaf_lds[icaf::ICAF_MAX_DEVICES];                                                                   \
        int _dev = 0;                                                                                                                \
        ICAF_HIP(hipGetDevice(&_dev));                                                                                               \
        if (_dev < 0 || _dev >= icaf::ICAF_MAX_DEVICES) return icaf::fail(ICAF_ERR_UNSUPPORTED, "device ordinal %d", _dev);          \
        const int _want = (int)(bytes);                                                                                              \
        if (_want > 160 * 1024) return icaf::fail(ICAF_ERR_UNSUPPORTED, "%d bytes of LDS exceed the 160 KiB of a CU", _want);        \
        if (_want > 64 * 1024 && _want > _icaf_lds[_dev]) {                                                                          \
            ICAF_HIP(hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, _want));                   \
            _icaf_lds[_dev] = _want;                                                                                                 \
        }                                                                                                                            \
    } while (0)

}  // namespace icaf
