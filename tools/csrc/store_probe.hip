// Store-pattern probe: how fast persistent workgroups (512 threads, one per CU) write 256 x 256 fp32 / bf16 output tiles of a row-major
// [M, N] matrix, by the shape of what ONE store instruction covers:
//   mode 0  the GEMM epilogue's ownership: a wave owns a 64-column slice; one instruction = 4 rows x 256 B (fp32) / 8 rows x 128 B (bf16)
//   mode 1  full tile rows: one instruction = 1 row x 1 KiB (fp32) / 2 rows x 512 B (bf16)
// and optionally a residual read with the same ownership (res != NULL).  `active` workgroups of the grid take part (the others exit): the rate
// one CU reaches when few CUs store at the same time.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE, bool F32, int depth>
__global__ __launch_bounds__(512) void store_probe_kernel(void* out, const float* res, int M, int N, int active, float* sink) {
    extern __shared__ char smem[];                       // 128 KiB requested by the host: one workgroup per CU
    if ((int)blockIdx.x >= active) return;
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};                   // sink != NULL: residual reads only (nothing stored but one value per lane at the end)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt_n = N / 256, ntiles = (M / 256) * nt_n;
    for (int t = blockIdx.x; t < ntiles; t += active) {
        const int m0 = (t / nt_n) * 256, n0 = (t % nt_n) * 256;
        if (F32) {
            if (MODE == 0) {
                const int g = wave >> 2, wc = wave & 3;
                const int n = n0 + wc * 64 + (lane & 15) * 4;
#pragma unroll 1
                for (int mi0 = 0; mi0 < 8; mi0 += depth) {           // `depth` 16-row groups of residual loads in flight
                    f32x4 r[depth][4];
                    _Pragma("unroll") for (int d = 0; d < depth; ++d)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int m = m0 + g * 128 + (mi0 + d) * 16 + i * 4 + (lane >> 4);
                            r[d][i] = res ? *reinterpret_cast<const f32x4*>(res + (size_t)m * N + n) : f32x4{1.f, 2.f, 3.f, 4.f};
                        }
                    _Pragma("unroll") for (int d = 0; d < depth; ++d)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int m = m0 + g * 128 + (mi0 + d) * 16 + i * 4 + (lane >> 4);
                            if (sink) keep += r[d][i]; else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (size_t)m * N + n) = r[d][i] + 1.0f;
                        }
                }
            } else {
#pragma unroll 1
                for (int j0 = 0; j0 < 32; j0 += 4 * depth) {         // wave w: rows w * 32 ... + 31, one full 1-KiB tile row per instruction
                    f32x4 r[depth][4];
                    _Pragma("unroll") for (int d = 0; d < depth; ++d)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int m = m0 + wave * 32 + j0 + d * 4 + i;
                            r[d][i] = res ? *reinterpret_cast<const f32x4*>(res + (size_t)m * N + n0 + lane * 4) : f32x4{1.f, 2.f, 3.f, 4.f};
                        }
                    _Pragma("unroll") for (int d = 0; d < depth; ++d)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int m = m0 + wave * 32 + j0 + d * 4 + i;
                            if (sink) keep += r[d][i]; else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (size_t)m * N + n0 + lane * 4) = r[d][i] + 1.0f;
                        }
                }
            }
        } else {
            uint16_t* o = reinterpret_cast<uint16_t*>(out);
            const u32x4 v = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            if (MODE == 0) {
                const int g = wave >> 2, wc = wave & 3;
                const int n = n0 + wc * 64 + (lane & 7) * 8;
#pragma unroll 1
                for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int m = m0 + g * 128 + mi * 16 + i * 8 + (lane >> 3);
                        *reinterpret_cast<u32x4*>(o + (size_t)m * N + n) = v;
                    }
            } else {
#pragma unroll 1
                for (int j = 0; j < 16; ++j) {                       // wave w: rows w * 32 ...; 2 rows x 512 B per instruction
                    const int m = m0 + wave * 32 + j * 2 + (lane >> 5);
                    *reinterpret_cast<u32x4*>(o + (size_t)m * N + n0 + (lane & 31) * 8) = v;
                }
            }
        }
    }
    if (sink && keep[0] + keep[1] + keep[2] + keep[3] == 12345.f) sink[blockIdx.x * 512 + threadIdx.x] = keep[0];
}

extern "C" int store_probe(int mode, int f32, void* out, const float* res, int M, int N, int grid, int active, int depth, float* sink, hipStream_t s) {
    if (M % 256 || N % 256 || (depth != 1 && depth != 2 && depth != 4)) return -2;
#define L(MODE, F, D) do { hipFuncSetAttribute((const void*)store_probe_kernel<MODE, F, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
                        store_probe_kernel<MODE, F, D><<<grid, 512, 131072, s>>>(out, res, M, N, active, sink); } while (0)
#define LD(MODE, F) do { if (depth == 1) L(MODE, F, 1); else if (depth == 2) L(MODE, F, 2); else L(MODE, F, 4); } while (0)
    if (mode == 0 && f32) LD(0, true); else if (mode == 0) LD(0, false); else if (f32) LD(1, true); else LD(1, false);
#undef LD
#undef L
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
