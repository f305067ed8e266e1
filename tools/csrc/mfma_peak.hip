// Diagnostic (tools/mfma_peak.py): what the matrix cores sustain under the board's power cap - every wave loops over 16 independent
// v_mfma_f32_16x16x32_bf16 (2 waves per SIMD on every CU), with the operands
//   mode 0  in registers, loaded once (no LDS, no memory: the ceiling the cap leaves to ANY bf16 GEMM)
//   mode 1  re-read from LDS at the 8-phase GEMM's ratio: 6 ds_read_b128 per 16 MFMAs (128 x 64 wave tile, each fragment read once)
//   mode 2  as 1, plus the GEMM's operand stream: 2 LDS-DMA loads (1 KiB each) per 16 MFMAs from an L2-resident buffer
//   mode 3  12 ds_read_b128 per 16 MFMAs (what a 64 x 64 wave tile would read)
// The nominal 2.5 PFLOP/s assumes 2.4 GHz; under the cap the clock drops, and how far depends on what else draws power.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_AS __attribute__((address_space(3)))
#define GLOBAL_AS __attribute__((address_space(1)))

template <int MODE>
__global__ __launch_bounds__(512) void mfma_peak_kernel(const bf16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[65536 + 16384];
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = ops[(size_t)(tid * 8 + i) & 0xFFFFF];
        b[i] = ops[(size_t)(tid * 8 + 4 + i) & 0xFFFFF];
    }
    for (int i = threadIdx.x; i < 4096; i += 512) reinterpret_cast<bf16x8*>(smem)[i] = ops[(size_t)(blockIdx.x * 4096 + i) & 0xFFFFF];
    __syncthreads();
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned lbase = (unsigned)(size_t)(LDS_AS const char*)smem;
    const char* gsrc = reinterpret_cast<const char*>(ops) + (size_t)(blockIdx.x & 63) * 16384 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {
            // conflict-free 16-B reads, a different 1-KiB row per read and per iteration
            const unsigned ad = lbase + ((it * 6 + wave * 8) & 63) * 1024 + lane * 16;
            asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(ad) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(a[1]) : "v"(ad) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a[2]) : "v"(ad) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(a[3]) : "v"(ad) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(b[0]) : "v"(ad) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(b[1]) : "v"(ad) : "memory");
            if (MODE == 3) {
                asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(b[2]) : "v"(ad) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:7168" : "=v"(b[3]) : "v"(ad) : "memory");
                bf16x8 t0, t1, t2, t3;
                asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(t0) : "v"(ad) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:9216" : "=v"(t1) : "v"(ad) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:10240" : "=v"(t2) : "v"(ad) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:11264" : "=v"(t3) : "v"(ad) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                a[0] ^= t0; a[1] ^= t1; a[2] ^= t2; a[3] ^= t3;
            }
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gsrc + ((it * 2 + j) & 15) * 1024),
                                                     (LDS_AS void*)(smem + 65536 + wave * 2048 + j * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MODE == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[tid] = s[0] + s[1] + s[2] + s[3] + (MODE == 2 ? (float)smem[65536 + threadIdx.x] : 0.f);
}

extern "C" int mfma_peak(const void* ops, void* out, int blocks, int iters, int mode, void* stream) {
    const bf16x8* o = (const bf16x8*)ops;
    float* r = (float*)out;
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
        case 0: hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(blocks), dim3(512), 0, s, o, r, iters); break;
        case 1: hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(blocks), dim3(512), 0, s, o, r, iters); break;
        case 2: hipLaunchKernelGGL(mfma_peak_kernel<2>, dim3(blocks), dim3(512), 0, s, o, r, iters); break;
        case 3: hipLaunchKernelGGL(mfma_peak_kernel<3>, dim3(blocks), dim3(512), 0, s, o, r, iters); break;
        default: return -2;
    }
    return (int)hipGetLastError();
}
