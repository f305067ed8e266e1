// Diagnostic (tools/valu_rate.py): issue cost of the VALU instructions the attention softmax is made of, in shader-clock cycles per
// wave64 instruction and SIMD: 1, 2 or 4 waves per SIMD run a long unrolled stream of independent instructions of one kind; cycles come from
// s_memtime, so the figure does not depend on the clock the power management picks.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int KIND>
__global__ __launch_bounds__(1024) void valu_rate_kernel(float* out, unsigned long long* cycles, int iters) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = (float)(threadIdx.x + i) * 1e-3f;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
            if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[i]));
            if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(r[i]));
            if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(r[i]));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
            if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
        }
        if (KIND == 4 || KIND == 5) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v = {r[i], r[i + 1]};
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(v));
                if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(v));
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(v));      // 2 per pair: 16 instructions per iteration like the others
                if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(v));
                r[i] = v[0]; r[i + 1] = v[1];
            }
        }
    }
    const unsigned long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    // the span from the first wave's start to the last wave's end (the arbiter favours the oldest wave: one wave's own time understates)
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&cycles[2 * blockIdx.x], c0);
        atomicMax(&cycles[2 * blockIdx.x + 1], c1);
    }
}

extern "C" int valu_rate(int kind, void* out, void* cycles, int blocks, int threads, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    float* o = (float*)out;
    unsigned long long* c = (unsigned long long*)cycles;
#define K(n) case n: hipLaunchKernelGGL(valu_rate_kernel<n>, dim3(blocks), dim3(threads), 0, s, o, c, iters); break;
    switch (kind) { K(0) K(1) K(2) K(3) K(4) K(5) K(6) K(7) default: return -2; }
#undef K
    return (int)hipGetLastError();
}
