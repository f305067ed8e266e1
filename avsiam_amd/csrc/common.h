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

// Exact (erf) GELU of timm Mlp / nn.GELU (/root/reference/src/models/cav_mae_base.py:115,138-143), evaluated as
//     gelu(x) = x * Phi(x),   Phi(x) ~= sigmoid(x (c0 + c1 x^2 + c2 x^4))      (minimax fit, |x| clamped to 10 inside p)
// max |Phi error| 5.0e-5, max |gelu error| 2.5e-5 over all x (fit script: tools/fit_gelu.py) - 80x below the bf16 rounding
// of the stored activation - in 9 VALU ops (one v_exp, one v_rcp) instead of ~23 for an erf expansion: the fc1/fc2-dgrad
// epilogues are VALU-bound, so this is what they cost.  The -log2(e) of exp() is folded into the coefficients.
#define GELU_C0 (-1.5950157683752235f * 1.4426950408889634f)
#define GELU_C1 (-0.07401129188724054f * 1.4426950408889634f)
#define GELU_C2 (0.0007030335403362688f * 1.4426950408889634f)

__device__ __forceinline__ float gelu_phi(float x, float& x2) {
    const float xc = __builtin_amdgcn_fmed3f(x, -10.0f, 10.0f);
    x2 = xc * xc;
    const float p = xc * fmaf(x2, fmaf(x2, GELU_C2, GELU_C1), GELU_C0);     // = -log2(e) * x (c0 + c1 x^2 + c2 x^4)
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p));
}

__device__ __forceinline__ float gelu_erf(float x) {
    float x2;
    return x * gelu_phi(x, x2);
}

// d/dx gelu = Phi(x) + x phi(x),  phi(x) = exp(-x^2/2) / sqrt(2 pi)
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float x2;
    const float cdf = gelu_phi(x, x2);
    const float pdf = __builtin_amdgcn_exp2f(x2 * (-0.5f * 1.4426950408889634f));
    return fmaf(x * 0.39894228040143268f, pdf, cdf);
}

// Workgroups b and b+8 share an XCD (round-robin dispatch; speed only, never correctness).  Map the dispatch index to a
// logical index so that each XCD gets a CONTIGUOUS run of logical work items: neighbouring items (GEMM tiles of one
// A panel, query tiles of one attention sequence, output tiles of one wgrad split) then hit the same 4-MiB L2.
// Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous run of tiles so that
    // neighbouring tiles (same A panel) hit the same L2.  Bijective for any nwg.
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
