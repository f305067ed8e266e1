// Shared device helpers for the avsiam gfx950 kernels (CDNA4 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint16_t bf16_t;                                   // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;  // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define AVS_WAVE 64

extern "C" void avs_set_error(const char* fmt, ...);

#define AVS_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            avs_set_error(__VA_ARGS__);          \
            return -2;                           \
        }                                        \
    } while (0)

#define AVS_LAUNCH_CHECK(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            avs_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return -1;                                                           \
        }                                                                        \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }

__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;                                  // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
    return __builtin_bit_cast(bf16_t, b);
}

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 output rounding); ~12 VALU + exp + rcp
// instead of the ~50-instruction libm expansion, which would otherwise dominate the fc1 epilogue.
__device__ __forceinline__ float erf_as(float x, float e /* = exp(-x*x) */) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    return copysignf(1.0f - poly * e, x);
}

// exact (erf) GELU of timm Mlp / nn.GELU (/root/reference/src/models/cav_mae_base.py:115,138-143)
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = x * 0.70710678118654752f;
    return 0.5f * x * (1.0f + erf_as(z, __expf(-z * z)));
}

__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float z = x * 0.70710678118654752f;
    const float e = __expf(-z * z);                       // = exp(-x^2/2)
    return 0.5f * (1.0f + erf_as(z, e)) + x * 0.39894228040143268f * e;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
