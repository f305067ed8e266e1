// k2 - LayerNorm forward/backward with a per-row modality affine select.
//
// Replaces nn.LayerNorm as used by Block (/root/reference/src/models/cav_mae_base.py:120-122,135-137,
// 151-152,169-170,190-191) and the final norms (:492,495,563,566,631).  Packing all sequences of a pass
// into one [rows, D] matrix means rows of both modalities share a launch, so the affine pair is picked
// per row (row_mod 0/1).  HBM-bound: one wave per row, 16 B per lane per access, the row stays in
// registers between the statistics and the normalisation (algorithmic bytes: 4D read + 2D written fwd).
// fp32 in (residual stream), bf16 out (GEMM operand), fp32 statistics - the reference's autocast keeps
// LayerNorm in fp32 as well.
#include "common.h"
#include <stdlib.h>

// RPW: rows per wave (1; 8 when the e4m3 copy is written with a device record: the wave then folds the max over its rows into the
// record with ONE atomic instead of one per row - 95 k atomics per launch on a handful of addresses cost more than the kernel itself)
template <int NV, bool F32IO, int RPW = 1>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g0,
                                                     const float* __restrict__ b0, const float* __restrict__ g1,
                                                     const float* __restrict__ b1, const uint8_t* __restrict__ row_mod,
                                                     const int* __restrict__ out_map, void* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int rows, float eps, uint8_t* __restrict__ y8, float q8_host, float* q8_dev) {
    // y8 (optional, bf16 output only): an OCP e4m3 copy of y * q8 for the fp8 forward GEMM that consumes this LayerNorm (engine.FP8) -
    // one more byte per element written here instead of a quantising pass that reads two and writes one.  q8_dev (device record,
    // common.h AVS_Q_*; may be NULL): the scale comes from it and the row's max |y| is folded into its running amax (delayed scaling)
    const float q8 = (y8 && q8_dev) ? q8_dev[AVS_Q_SCALE] : q8_host;
    const float amax_seen = q_amax_peek(y8 ? q8_dev : nullptr);
    float ymax = 0.f;
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
#pragma unroll 1
    for (int row = row0; row < min(rows, row0 + RPW); ++row) {
    // the row's modality and output row are requested FIRST and the affine rows before the statistics: loaded where they are
    // used (after the two reductions) they formed a chain of three dependent memory round trips behind the row itself
    const int mod = row_mod ? row_mod[row] : 0;
    const int orow = out_map ? out_map[row] : row;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = xr[i * 64 + lane];
    const float4* gp = reinterpret_cast<const float4*>(mod ? g1 : g0);
    const float4* bp = reinterpret_cast<const float4*>(mod ? b1 : b0);
    float4 gv[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { gv[i] = gp[i * 64 + lane]; bv[i] = bp[i * 64 + lane]; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum_dpp(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float var = wave_sum_dpp(q) * (1.0f / D);
    const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 g = gv[i], b = bv[i];
        const float4 r = make_float4(v[i].x * rs * g.x + b.x, v[i].y * rs * g.y + b.y, v[i].z * rs * g.z + b.z,
                                     v[i].w * rs * g.w + b.w);
        if (F32IO) {
            reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + (size_t)orow * D)[i * 64 + lane] = r;
        } else {
            uint2 o;
            o.x = pack_bf2(r.x, r.y);
            o.y = pack_bf2(r.z, r.w);
            if (y) reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y) + (size_t)orow * D)[i * 64 + lane] = o;      // (NULL: the e4m3 copy is the only output)
            if (y8) {
                ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(r.x * q8, -448.f, 448.f), __builtin_amdgcn_fmed3f(r.y * q8, -448.f, 448.f), 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(r.z * q8, -448.f, 448.f), __builtin_amdgcn_fmed3f(r.w * q8, -448.f, 448.f), w, true);
                reinterpret_cast<int*>(y8 + (size_t)orow * D)[i * 64 + lane] = w;
            }
        }
    }
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rs;
    }
    }
    if (!F32IO && y8 && q8_dev) q_amax_update(q8_dev, ymax, amax_seen);
}

// Backward.  dx = dres + rstd * (gy - mean(gy) - xhat * mean(gy * xhat)),  gy = dy * gamma.
// Parameter gradients: each block reduces its rows into a private slab ws[block][set][{dgamma,dbeta}][D]
// (plain stores, deterministic); ln_bwd_reduce_kernel sums the slabs into the gradient arena.
constexpr int LN_MIN_ROWS_PER_WAVE = 4;      // the workspace is sized for this (most blocks)
constexpr int LN_SETS = 5;                   // dgamma0, dbeta0, dgamma1, dbeta1, column-sum of dx
constexpr int LN_REDUCE_CHUNKS = 64;

// RES: 0 no residual gradient | 1 fp32 dres | 2 bf16 dres (the bf16 gradient stream: the previous LayerNorm backward's dx_bf16).
// dx (fp32) may be NULL when only the bf16 copy is wanted.
template <int NV, bool F32IO, int RES, int LN_ROWS_PER_WAVE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ g0, const float* __restrict__ g1,
                                                     const uint8_t* __restrict__ row_mod, const int* __restrict__ out_map,
                                                     const void* dres, float* dx, bf16_t* __restrict__ dx_bf16,
                                                     float* __restrict__ ws, int rows, uint8_t* __restrict__ dx8, float* q8) {
    // dx8 / q8 (fp8 backward, may be NULL): an OCP e5m2 copy of dx * q8[0] - the gradient operand of the fp8 input-gradient GEMM that
    // consumes dx - with max |dx| folded into the device record q8 (common.h AVS_Q_*)
    constexpr int D = NV * 256;
    const float q8s = dx8 ? q8[AVS_Q_SCALE] : 0.f;
    const float amax_seen = q_amax_peek(dx8 ? q8 : nullptr);
    float dmax = 0.f;
    constexpr int LN_ROWS_PER_BLOCK = 4 * LN_ROWS_PER_WAVE;
    __shared__ float red[4][D];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float4 dg0[NV], db0[NV], dg1[NV], db1[NV], dc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        dg0[i] = make_float4(0, 0, 0, 0); db0[i] = dg0[i]; dg1[i] = dg0[i]; db1[i] = dg0[i]; dc[i] = dg0[i];
    }
    const int row0 = blockIdx.x * LN_ROWS_PER_BLOCK + wave * LN_ROWS_PER_WAVE;
    // lane r (< 16) fetches the per-row scalars of the wave's r-th row once; the row loop reads them with v_readlane.
    // Loaded inside the loop (modality -> gamma pointer, statistics, output row) they were dependent memory round trips
    // in front of every row, each a vmcnt(0) that also waited for the previous row's stores.
    const int lrow = min(row0 + (lane & (LN_ROWS_PER_WAVE - 1)), rows - 1);
    const int l_mod = row_mod ? row_mod[lrow] : 0;
    const float l_mean = mean_in[lrow], l_rs = rstd_in[lrow];
    const int l_drow = out_map ? out_map[lrow] : lrow;
    for (int rr = 0; rr < LN_ROWS_PER_WAVE; ++rr) {
        const int row = row0 + rr;
        if (row >= rows) break;
        const int mod = __builtin_amdgcn_readlane(l_mod, rr);
        const float mean = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l_mean), rr));
        const float rs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l_rs), rr));
        const int drow = __builtin_amdgcn_readlane(l_drow, rr);
        const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
        const float4* gp = reinterpret_cast<const float4*>(mod ? g1 : g0);
        float4 xh[NV], gy[NV], rsd[NV];
        float s1 = 0.f, s2 = 0.f;
        // the residual-gradient row is loaded together with x and dy (not after the two reductions, where its latency
        // would be exposed once per row)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (RES == 1) {
                rsd[i] = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dres) + (size_t)row * D)[i * 64 + lane];
            } else if (RES == 2) {
                const uint2 rv = reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(dres) + (size_t)row * D)[i * 64 + lane];
                rsd[i] = make_float4(__uint_as_float(rv.x << 16), __uint_as_float(rv.x & 0xffff0000u), __uint_as_float(rv.y << 16),
                                     __uint_as_float(rv.y & 0xffff0000u));
            } else {
                rsd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float4 xv = xr[i * 64 + lane];
            const float4 g = gp[i * 64 + lane];
            float4 d;
            if (F32IO) {
                d = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy) + (size_t)drow * D)[i * 64 + lane];
            } else {
                const uint2 dv = reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(dy) + (size_t)drow * D)[i * 64 + lane];
                d.x = __uint_as_float(dv.x << 16); d.y = __uint_as_float(dv.x & 0xffff0000u);
                d.z = __uint_as_float(dv.y << 16); d.w = __uint_as_float(dv.y & 0xffff0000u);
            }
            xh[i].x = (xv.x - mean) * rs; xh[i].y = (xv.y - mean) * rs;
            xh[i].z = (xv.z - mean) * rs; xh[i].w = (xv.w - mean) * rs;
            gy[i].x = d.x * g.x; gy[i].y = d.y * g.y; gy[i].z = d.z * g.z; gy[i].w = d.w * g.w;
            s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
            s2 += (gy[i].x * xh[i].x + gy[i].y * xh[i].y) + (gy[i].z * xh[i].z + gy[i].w * xh[i].w);
            if (mod) {                                       // wave-uniform branch
                dg1[i].x += d.x * xh[i].x; dg1[i].y += d.y * xh[i].y; dg1[i].z += d.z * xh[i].z; dg1[i].w += d.w * xh[i].w;
                db1[i].x += d.x; db1[i].y += d.y; db1[i].z += d.z; db1[i].w += d.w;
            } else {
                dg0[i].x += d.x * xh[i].x; dg0[i].y += d.y * xh[i].y; dg0[i].z += d.z * xh[i].z; dg0[i].w += d.w * xh[i].w;
                db0[i].x += d.x; db0[i].y += d.y; db0[i].z += d.z; db0[i].w += d.w;
            }
        }
        const float m1 = wave_sum_dpp(s1) * (1.0f / D);
        const float m2 = wave_sum_dpp(s2) * (1.0f / D);
        float4* dxr = reinterpret_cast<float4*>(dx + (size_t)row * D);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float4 o;
            o.x = rs * (gy[i].x - m1 - xh[i].x * m2) + rsd[i].x; o.y = rs * (gy[i].y - m1 - xh[i].y * m2) + rsd[i].y;
            o.z = rs * (gy[i].z - m1 - xh[i].z * m2) + rsd[i].z; o.w = rs * (gy[i].w - m1 - xh[i].w * m2) + rsd[i].w;
            if (dx) dxr[i * 64 + lane] = o;
            dc[i].x += o.x; dc[i].y += o.y; dc[i].z += o.z; dc[i].w += o.w;      // column sum of dx (bias grad of the producer Linear)
            if (dx_bf16) {
                uint2 ob;
                ob.x = pack_bf2(o.x, o.y);
                ob.y = pack_bf2(o.z, o.w);
                reinterpret_cast<uint2*>(dx_bf16 + (size_t)row * D)[i * 64 + lane] = ob;
            }
            if (dx8) {
                dmax = fmaxf(fmaxf(dmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                int w = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(o.x * q8s, -57344.f, 57344.f), __builtin_amdgcn_fmed3f(o.y * q8s, -57344.f, 57344.f), 0, false);
                w = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(o.z * q8s, -57344.f, 57344.f), __builtin_amdgcn_fmed3f(o.w * q8s, -57344.f, 57344.f), w, true);
                reinterpret_cast<int*>(dx8 + (size_t)row * D)[i * 64 + lane] = w;
            }
        }
    }
    if (dx8) q_amax_update(q8, dmax, amax_seen);
    // cross-wave reduction of the five accumulator sets, one set at a time through LDS
    float* slab = ws + (size_t)blockIdx.x * LN_SETS * D;
#pragma unroll
    for (int set = 0; set < LN_SETS; ++set) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float4 a = set == 0 ? dg0[i] : set == 1 ? db0[i] : set == 2 ? dg1[i] : set == 3 ? db1[i] : dc[i];
            reinterpret_cast<float4*>(red[wave])[i * 64 + lane] = a;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) slab[set * D + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same backward with the NEXT row's operands in flight while the current row is reduced (round 4).  The kernel above issues a
// row's loads, waits for them, reduces, stores - per wave one row's ~6 KB in flight for about half of the ~3.5 us a row takes, so
// with the bf16 residual-gradient stream (10 B per element) it sat at 4.1 TB/s.  A second row of loads in REGISTERS costs the
// occupancy step that pays for it (measured in round 3: spills at three waves per SIMD).  Here the rows come in through LDS-DMA
// (global_load_lds, 16 B per lane, no VGPR): each wave owns two 8 D-byte slots [x fp32 | dy bf16 | dres bf16]; at the top of row r the
// DMAs of row r + 1 are issued into the other slot, a COUNTED s_waitcnt leaves them in flight while it waits for row r's (a row's
// stores are issued one row late so that this wait never stands behind a fresh store), and the row is read back with ds_read_b128 / b64 through inline asm (hipcc would put vmcnt(0) in front of every
// LDS read it can see beside a pending DMA).  gamma of the current modality stays in registers (re-loaded, behind a full wait,
// when the modality of the wave's rows changes); the two row reductions use DPP adds instead of six ds_bpermute round trips each.
// Covers the case the step is made of: bf16 dy, bf16 dres (the gradient stream), bf16 dx only; everything else takes the kernel above.
#define LDS_AS __attribute__((address_space(3)))
#define GLOBAL_AS __attribute__((address_space(1)))

template <int OFF>
__device__ __forceinline__ void lds_r128(f32x4& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory");
}
// (a 64-bit INTEGER output: with a two-float vector as the asm's output operand hipcc used the low register for both elements)
template <int OFF>
__device__ __forceinline__ void lds_r64(unsigned long long& v, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory");
}

// G8 (fp8 backward, round 5): also the e5m2 copy of dx with its device record, as ln_bwd_kernel writes it (dx8 / q8) - held back one row like the bf16 row.
template <int NV, int RPW, bool G8 = false>
__global__ __launch_bounds__(256) void ln_bwd_dma_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                         const float* __restrict__ g0, const float* __restrict__ g1,
                                                         const uint8_t* __restrict__ row_mod, const int* __restrict__ out_map,
                                                         const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx_bf16,
                                                         float* __restrict__ ws, int rows, uint8_t* __restrict__ dx8 = nullptr, float* q8 = nullptr) {
    constexpr int D = NV * 256;
    constexpr int SLOT = 8 * D;                        // bytes: x | dy | dres
    constexpr int NH = (NV * 32 + 63) / 64;            // DMA instructions of a bf16 row (16 B per lane; the last one may use half the lanes)
    constexpr int NDMA = NV + 2 * NH;                  // LDS-DMA instructions per row
    extern __shared__ __attribute__((aligned(16))) char smem[];            // 4 x 2 x SLOT bytes (the launcher passes it): 48 KiB at D = 768 = three blocks per
                                                                            // CU, 80 KiB at D = 1280 = two; reused by the slab reduction
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const myslots = smem + wave * 2 * SLOT;
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS const char*)myslots;
    // (ext-vector arrays throughout: arrays of HIP's float4 STRUCT are not always scalarised and then live in scratch, whose
    //  loads and stores are vector-memory operations - they would break the counted waits below)
    f32x4 dg0[NV], db0[NV], dg1[NV], db1[NV], dc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        dg0[i] = f32x4{0.f, 0.f, 0.f, 0.f}; db0[i] = dg0[i]; dg1[i] = dg0[i]; db1[i] = dg0[i]; dc[i] = dg0[i];
    }
    const int row0 = (blockIdx.x * 4 + wave) * RPW;
    const int lrow = min(row0 + (lane & (RPW - 1)), rows - 1);
    const int l_mod = row_mod ? row_mod[lrow] : 0;
    const float l_mean = mean_in[lrow], l_rs = rstd_in[lrow];
    const int l_drow = out_map ? out_map[lrow] : lrow;
    float q8s = 0.f, amax_seen = 0.f, dmax = 0.f;
    if (G8) {                                           // (ordinary loads, waited for with the per-row scalars in front of the loop)
        q8s = q8[AVS_Q_SCALE];
        amax_seen = q_amax_peek(q8);
    }

    auto issue = [&](int row, int drow, int slot) {     // the three operand rows of `row` -> slot, by LDS-DMA
        char* base = myslots + slot * SLOT;
        const char* xs = reinterpret_cast<const char*>(x + (size_t)row * D) + lane * 16;
        const char* ds_ = reinterpret_cast<const char*>(dy + (size_t)drow * D) + lane * 16;
        const char* rs_ = reinterpret_cast<const char*>(dres + (size_t)row * D) + lane * 16;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(xs + i * 1024), (LDS_AS void*)(base + i * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            if ((j + 1) * 64 <= NV * 32 || lane < NV * 32 - j * 64) {          // whole instruction, or the half the row still has
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(ds_ + j * 1024), (LDS_AS void*)(base + 4 * D + j * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(rs_ + j * 1024), (LDS_AS void*)(base + 6 * D + j * 1024), 16, 0, 0);
            }
        }
    };

    f32x4 gcur[NV];
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    u32x2 held[NV];                                    // bf16 output of the previous row (stored one iteration late, see the wait)
    uint32_t held8[NV];                                // ... and its e5m2 copy (G8)
#pragma unroll
    for (int i = 0; i < NV; ++i) { gcur[i] = f32x4{0.f, 0.f, 0.f, 0.f}; held[i] = u32x2{0u, 0u}; held8[i] = 0u; }
    int cur_mod = -1, last = -1;
    // the per-row scalars are consumed here, in the compiler's view: it waits for their loads ONCE, before the loop (left to the first
    // v_readlane it would wait at the loop header - vmcnt(0) in every iteration, in front of the next row's DMA issue)
    asm volatile("" ::"v"(l_mod), "v"(l_mean), "v"(l_rs), "v"(l_drow), "v"(q8s), "v"(amax_seen) : "memory");
    if (row0 < rows) issue(row0, __builtin_amdgcn_readlane(l_drow, 0), 0);
    const int nmine = max(0, min(RPW, rows - row0));          // rows of this wave (one loop exit: with a `break` hipcc keeps a second copy of
                                                              // every accumulator for the merge of the two exits)
#pragma unroll 1
    for (int rr = 0; rr < nmine; ++rr) {
        const int row = row0 + rr;
        const int mod = __builtin_amdgcn_readlane(l_mod, rr);
        const float mean = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l_mean), rr));
        const float rs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l_rs), rr));
        if (mod != cur_mod) {                            // wave-uniform, rare: before the next DMAs are issued, so its wait drains only what is due anyway
            const f32x4* gp = reinterpret_cast<const f32x4*>(mod ? g1 : g0);
#pragma unroll
            for (int i = 0; i < NV; ++i) gcur[i] = gp[i * 64 + lane];
            cur_mod = mod;
            // consumed HERE, in the compiler's view: it then waits for these loads inside the branch; left pending at the join, its wait
            // (vmcnt(0): it cannot see the asm waits) would sit in front of the first use of gamma in EVERY iteration - behind the DMAs
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(gcur[i])::"memory");
        }
        // the next row (clamped: the instruction count per row must not depend on the data) goes into the other slot
        {
            const int nxt = min(rr + 1, RPW - 1);
            const int nrow = min(row0 + nxt, rows - 1);
            issue(nrow, __builtin_amdgcn_readlane(l_drow, nxt), (rr + 1) & 1);
        }
        // Wait for row r's DMAs and leave the ones just issued in flight.  The count must hold whatever the STORES in the queue do:
        // loads retire in order among themselves, but a store may be acknowledged before an older load (a count that budgets for
        // pending stores - vmcnt(NDMA + NST) - read stale rows on hardware), so the only safe count is the loads that may stay, and
        // it then also covers every older store.  To make that free, a row's stores are issued one iteration late (below): by the
        // time of this wait they are a whole row's arithmetic old.
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        const unsigned sb = lds0 + (rr & 1) * SLOT;
        f32x4 xv[NV];
        unsigned long long dv[NV], rv[NV];
        {
            const unsigned ax = sb + lane * 16, ad = sb + 4 * D + lane * 8, ar = sb + 6 * D + lane * 8;
            lds_r128<0>(xv[0], ax); lds_r64<0>(dv[0], ad); lds_r64<0>(rv[0], ar);
            if (NV > 1) { lds_r128<1024>(xv[1], ax); lds_r64<512>(dv[1], ad); lds_r64<512>(rv[1], ar); }
            if (NV > 2) { lds_r128<2048>(xv[2], ax); lds_r64<1024>(dv[2], ad); lds_r64<1024>(rv[2], ar); }
            if (NV > 3) { lds_r128<3072>(xv[3], ax); lds_r64<1536>(dv[3], ad); lds_r64<1536>(rv[3], ar); }
            if (NV > 4) { lds_r128<4096>(xv[4], ax); lds_r64<2048>(dv[4], ad); lds_r64<2048>(rv[4], ar); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        if (rr > 0) {                                     // the previous row's output, held back across the wait above
#pragma unroll
            for (int i = 0; i < NV; ++i) reinterpret_cast<u32x2*>(dx_bf16 + (size_t)(row - 1) * D)[i * 64 + lane] = held[i];
            if (G8) {
#pragma unroll
                for (int i = 0; i < NV; ++i) reinterpret_cast<uint32_t*>(dx8 + (size_t)(row - 1) * D)[i * 64 + lane] = held8[i];
            }
        }
        f32x4 xh[NV], gy[NV];
        float s1 = 0.f, s2 = 0.f;
        const float w1 = mod ? 1.f : 0.f, w0 = 1.f - w1;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 g = gcur[i];
            const unsigned d01 = (unsigned)dv[i], d23 = (unsigned)(dv[i] >> 32);
            const f32x4 d = {__uint_as_float(d01 << 16), __uint_as_float(d01 & 0xffff0000u), __uint_as_float(d23 << 16), __uint_as_float(d23 & 0xffff0000u)};
            xh[i] = (xv[i] - mean) * rs;
            gy[i] = d * g;
            s1 += (gy[i][0] + gy[i][1]) + (gy[i][2] + gy[i][3]);
            const f32x4 gx = gy[i] * xh[i];
            s2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
            // branch-free: a wave-uniform `if (mod)` around the two accumulator sets makes hipcc carry copies of all of them
            // across the join (+70 VGPRs once anything else is loop-carried); two scalar weights cost two FMAs per element instead
            const f32x4 dxh = d * xh[i];
            dg0[i] += w0 * dxh; db0[i] += w0 * d;
            dg1[i] += w1 * dxh; db1[i] += w1 * d;
        }
        const float m1 = wave_sum_dpp(s1) * (1.0f / D);
        const float m2 = wave_sum_dpp(s2) * (1.0f / D);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const unsigned r01 = (unsigned)rv[i], r23 = (unsigned)(rv[i] >> 32);
            const f32x4 rsd = {__uint_as_float(r01 << 16), __uint_as_float(r01 & 0xffff0000u), __uint_as_float(r23 << 16), __uint_as_float(r23 & 0xffff0000u)};
            const f32x4 o = rs * (gy[i] - m1 - xh[i] * m2) + rsd;
            dc[i] += o;
            held[i] = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
            if (G8) {
                dmax = fmaxf(fmaxf(dmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                int w = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(o[0] * q8s, -57344.f, 57344.f), __builtin_amdgcn_fmed3f(o[1] * q8s, -57344.f, 57344.f), 0, false);
                w = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(o[2] * q8s, -57344.f, 57344.f), __builtin_amdgcn_fmed3f(o[3] * q8s, -57344.f, 57344.f), w, true);
                held8[i] = (uint32_t)w;
            }
        }
        last = row;
    }
    if (last >= 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) reinterpret_cast<u32x2*>(dx_bf16 + (size_t)last * D)[i * 64 + lane] = held[i];
        if (G8) {
#pragma unroll
            for (int i = 0; i < NV; ++i) reinterpret_cast<uint32_t*>(dx8 + (size_t)last * D)[i * 64 + lane] = held8[i];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last (dummy) prefetch has landed: the slots can be reused
    if (G8) q_amax_update(q8, dmax, amax_seen);
    float (*red)[D] = reinterpret_cast<float (*)[D]>(smem);
    float* slab = ws + (size_t)blockIdx.x * LN_SETS * D;
#pragma unroll
    for (int set = 0; set < LN_SETS; ++set) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 a = set == 0 ? dg0[i] : set == 1 ? db0[i] : set == 2 ? dg1[i] : set == 3 ? db1[i] : dc[i];
            reinterpret_cast<f32x4*>(red[wave])[i * 64 + lane] = a;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) slab[set * D + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// Sum the per-block slabs: grid (D/256, set, chunk); each chunk of blocks is summed in registers and added with one
// atomic per column (LN_REDUCE_CHUNKS adders per address).
__global__ void ln_bwd_reduce_kernel(const float* __restrict__ ws, int nblocks, int D, float* dg0, float* db0,
                                     float* dg1, float* db1, float* dcol) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int set = blockIdx.y;
    if (c >= D) return;
    float* dst = set == 0 ? dg0 : set == 1 ? db0 : set == 2 ? dg1 : set == 3 ? db1 : dcol;
    if (!dst) return;
    const int per = (nblocks + gridDim.z - 1) / gridDim.z;
    const int b0 = blockIdx.z * per, b1 = min(nblocks, b0 + per);
    if (b0 >= b1) return;
    float s = 0.f;
    for (int b = b0; b < b1; ++b) s += ws[((size_t)b * LN_SETS + set) * D + c];
    atomicAdd(dst + c, s);
}

extern "C" int avs_layernorm_ws_floats(int rows, int D) { return ceil_div(rows, 4 * LN_MIN_ROWS_PER_WAVE) * LN_SETS * D; }

// rows per wave of the backward kernels: a block of 4 waves x RPW rows writes one slab of parameter-gradient partial sums; fewer rows per wave = more
// blocks (helps only when 16 rows leave most CUs with a single block) and more slab traffic.
static int ln_bwd_rpw(int rows) {
    int rpw = avs_tuning().ln_rpw;
    if (rpw != 4 && rpw != 8 && rpw != 16) rpw = rows >= 16384 ? 16 : 8;        // measured (tools/bench_ln.py): 8 wins only on the 8192-row audio tower
    return rpw;
}
// slabs ([LN_SETS][D] floats each) a backward over `rows` rows writes with the current knobs: what a caller that reduces the slabs itself
// (avs_layernorm_bwd with no gradient target, then avs_layernorm_bwd_reduce_batched) must reserve and tell the batched reduce
extern "C" int avs_layernorm_bwd_slabs(int rows) { return rows > 0 ? ceil_div(rows, 4 * ln_bwd_rpw(rows)) : 0; }

// n slab sets in ONE launch (a stack's LayerNorm backwards: one reduce at the end of the stack's backward instead of one ~10-us launch per LayerNorm
// inside it): desc[7 i ..] = {ws, slabs, dg0, db0, dg1, db1, dcol} of set i (gradient targets may be 0), all of width D
__global__ void ln_bwd_reduce_batched_kernel(const long long* __restrict__ desc, int D, int chunks) {
    const long long* d = desc + 7 * (blockIdx.z / chunks);
    const int chunk = blockIdx.z % chunks;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int set = blockIdx.y;
    if (c >= D) return;
    float* dst = reinterpret_cast<float*>(d[2 + set]);
    if (!dst) return;
    const float* ws = reinterpret_cast<const float*>(d[0]);
    const int nblocks = (int)d[1];
    const int per = (nblocks + chunks - 1) / chunks;
    const int b0 = chunk * per, b1 = min(nblocks, b0 + per);
    if (b0 >= b1) return;
    float s = 0.f;
    for (int b = b0; b < b1; ++b) s += ws[((size_t)b * LN_SETS + set) * D + c];
    atomicAdd(dst + c, s);
}

extern "C" int avs_layernorm_bwd_reduce_batched(const long long* desc, int n, int D, hipStream_t stream) {
    AVS_CHECK_ARG(desc && n > 0 && n * LN_REDUCE_CHUNKS <= 65535 && D > 0, "layernorm_bwd_reduce_batched: n=%d D=%d", n, D);
    AVS_CHECK_ARG(!avs_tuning().det, "layernorm_bwd_reduce_batched: not in the deterministic mode (avs_layernorm_bwd reduces per call there)");
    ln_bwd_reduce_batched_kernel<<<dim3(ceil_div(D, 256), LN_SETS, n * LN_REDUCE_CHUNKS), 256, 0, stream>>>(desc, D, LN_REDUCE_CHUNKS);
    AVS_LAUNCH_CHECK("layernorm_bwd_reduce_batched");
    return 0;
}

// knobs (api.cpp, common.h AvsTuning): ln_dma - 1: the LDS-DMA backward kernel where it applies (default), 0: never (A/B);
// ln_rpw - rows per wave of the backward kernel: 0 automatic, 4 / 8 / 16 forced (tuning)

extern "C" int avs_layernorm_fwd_q8(const float* x, const float* g0, const float* b0, const float* g1, const float* b1,
                                    const uint8_t* row_mod, const int* out_map, void* y, int y_f32, float* mean, float* rstd,
                                    int rows, int D, float eps, uint8_t* y8, float q8, float* q8_dev, hipStream_t stream) {
    AVS_CHECK_ARG(!(y8 && y_f32), "layernorm_fwd: the fp8 copy goes with the bf16 output");
    // D = 1536 / 2048 / 2560: the concatenated audio|video feature of the fusion classification head at ViT-B / ViT-L / ViT-H width (forward
    // only, no e4m3 copy)
    AVS_CHECK_ARG(rows > 0 && (D == 512 || D == 768 || D == 1024 || D == 1280 || D == 1536 || D == 2048 || D == 2560), "layernorm_fwd: unsupported rows=%d D=%d", rows, D);
    AVS_CHECK_ARG(!(y8 && D > 1536), "layernorm_fwd: no e4m3 copy at D=%d", D);
    AVS_CHECK_ARG(x && g0 && b0 && (y || (y8 && !y_f32)) && mean && rstd, "layernorm_fwd: null pointer (y may be NULL only beside y8)");
    AVS_CHECK_ARG(!row_mod || (g1 && b1), "layernorm_fwd: row_mod given without second affine set");
    dim3 grid(ceil_div(rows, 4)), block(256);
#define LN_FWD(NV, F) ln_fwd_kernel<NV, F><<<grid, block, 0, stream>>>(x, g0, b0, g1, b1, row_mod, out_map, y, mean, rstd, rows, eps, y8, q8, q8_dev)
#define LN_FWD8(NV) ln_fwd_kernel<NV, false, 8><<<dim3(ceil_div(rows, 32)), block, 0, stream>>>(x, g0, b0, g1, b1, row_mod, out_map, y, mean, rstd, rows, eps, y8, q8, q8_dev)
    if (y8 && q8_dev) {                                   // e4m3 copy with a record: 8 rows per wave, one amax atomic per wave
        if (D == 512) LN_FWD8(2); else if (D == 768) LN_FWD8(3); else if (D == 1024) LN_FWD8(4); else if (D == 1280) LN_FWD8(5); else LN_FWD8(6);
    } else if (y_f32) {
        if (D == 512) LN_FWD(2, true); else if (D == 768) LN_FWD(3, true); else if (D == 1024) LN_FWD(4, true); else if (D == 1280) LN_FWD(5, true);
        else if (D == 1536) LN_FWD(6, true); else if (D == 2048) LN_FWD(8, true); else LN_FWD(10, true);
    } else {
        if (D == 512) LN_FWD(2, false); else if (D == 768) LN_FWD(3, false); else if (D == 1024) LN_FWD(4, false); else if (D == 1280) LN_FWD(5, false);
        else if (D == 1536) LN_FWD(6, false); else if (D == 2048) LN_FWD(8, false); else LN_FWD(10, false);
    }
#undef LN_FWD
#undef LN_FWD8
    AVS_LAUNCH_CHECK("layernorm_fwd");
    return 0;
}

extern "C" int avs_layernorm_fwd(const float* x, const float* g0, const float* b0, const float* g1, const float* b1,
                                 const uint8_t* row_mod, const int* out_map, void* y, int y_f32, float* mean, float* rstd,
                                 int rows, int D, float eps, hipStream_t stream) {
    return avs_layernorm_fwd_q8(x, g0, b0, g1, b1, row_mod, out_map, y, y_f32, mean, rstd, rows, D, eps, nullptr, 1.0f, nullptr, stream);
}

// dg*/db* are ACCUMULATED into (+=); dx may alias dres; dx_bf16 (optional) receives a bf16 copy of dx; dcol (optional)
// accumulates the column sum of dx (the bias gradient of the Linear whose output gradient dx is).
// dres_bf16: dres is a bf16 matrix (the previous LayerNorm backward's dx_bf16: the residual-gradient stream kept in bf16 between the
// blocks, 6 B per element less traffic); dx may then be NULL (only the bf16 copy is written).  dx_bf16 must not alias a bf16 dres.
// ws: avs_layernorm_ws_floats(rows, D) floats.
extern "C" int avs_layernorm_bwd(const void* dy, int dy_f32, const float* x, const float* mean, const float* rstd,
                                 const float* g0, const float* g1, const uint8_t* row_mod, const int* out_map,
                                 const void* dres, int dres_bf16, float* dx, bf16_t* dx_bf16, float* dg0, float* db0, float* dg1,
                                 float* db1, float* dcol, float* ws, int rows, int D, uint8_t* dx8, float* q8, hipStream_t stream) {
    AVS_CHECK_ARG((dx8 == nullptr) == (q8 == nullptr), "layernorm_bwd: dx8 and its record go together");
    AVS_CHECK_ARG(rows > 0 && (D == 512 || D == 768 || D == 1024 || D == 1280), "layernorm_bwd: unsupported rows=%d D=%d", rows, D);
    AVS_CHECK_ARG(dy && x && mean && rstd && g0 && (dx || dx_bf16) && ws, "layernorm_bwd: null pointer");
    AVS_CHECK_ARG(!(dres && dres_bf16 && (const void*)dx_bf16 == dres), "layernorm_bwd: dx_bf16 must not alias a bf16 dres");
    const int rpw = ln_bwd_rpw(rows);
    const int nblocks = ceil_div(rows, 4 * rpw);
    // no gradient target at all: the slabs stay in ws for the caller's own reduce (avs_layernorm_bwd_reduce_batched)
    const bool reduce = dg0 || db0 || dg1 || db1 || dcol;
    dim3 grid(nblocks), block(256);
    // the step's common case - bf16 dy, bf16 residual-gradient stream in, bf16 dx out only - takes the LDS-DMA kernel (D = 1280, round 5: 80 KiB of
    // dynamic LDS, two blocks per CU); AVSIAM_LN_DMA=0: the register-load kernel for everything (A/B)
    if (avs_tuning().ln_dma && !dy_f32 && dres && dres_bf16 && !dx && dx_bf16) {
        const int lds = 4 * 2 * 8 * D;                     // four waves x two slots of [x fp32 | dy bf16 | dres bf16] rows
        static bool attr_done = false;
        if (!attr_done) {                                  // D = 1280: 80 KiB, above the 64 KiB a kernel gets without asking
            const void* ks[6] = {(const void*)ln_bwd_dma_kernel<5, 16>, (const void*)ln_bwd_dma_kernel<5, 8>, (const void*)ln_bwd_dma_kernel<5, 4>,
                                 (const void*)ln_bwd_dma_kernel<5, 16, true>, (const void*)ln_bwd_dma_kernel<5, 8, true>, (const void*)ln_bwd_dma_kernel<5, 4, true>};
            for (const void* k : ks)
                if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 81920) != hipSuccess) {
                    avs_set_error("layernorm_bwd: hipFuncSetAttribute failed");
                    return -1;
                }
            attr_done = true;
        }
#define LN_DMA_(NV, G)                                                                                                                          \
    do {                                                                                                                                        \
        if (rpw == 16) ln_bwd_dma_kernel<NV, 16, G><<<grid, block, lds, stream>>>((const bf16_t*)dy, x, mean, rstd, g0, g1, row_mod, out_map, (const bf16_t*)dres, dx_bf16, ws, rows, dx8, q8); \
        else if (rpw == 8) ln_bwd_dma_kernel<NV, 8, G><<<grid, block, lds, stream>>>((const bf16_t*)dy, x, mean, rstd, g0, g1, row_mod, out_map, (const bf16_t*)dres, dx_bf16, ws, rows, dx8, q8); \
        else ln_bwd_dma_kernel<NV, 4, G><<<grid, block, lds, stream>>>((const bf16_t*)dy, x, mean, rstd, g0, g1, row_mod, out_map, (const bf16_t*)dres, dx_bf16, ws, rows, dx8, q8); \
    } while (0)
#define LN_DMA(NV) do { if (dx8) LN_DMA_(NV, true); else LN_DMA_(NV, false); } while (0)
        if (D == 512) LN_DMA(2); else if (D == 768) LN_DMA(3); else if (D == 1024) LN_DMA(4); else LN_DMA(5);
#undef LN_DMA
#undef LN_DMA_
        AVS_LAUNCH_CHECK("layernorm_bwd_dma");
        if (!reduce) return 0;
        const int chunks_ = avs_tuning().det ? 1 : nblocks < LN_REDUCE_CHUNKS ? nblocks : LN_REDUCE_CHUNKS;       // (det: one adder per address)
        ln_bwd_reduce_kernel<<<dim3(ceil_div(D, 256), LN_SETS, chunks_), 256, 0, stream>>>(ws, nblocks, D, dg0, db0, dg1, db1, dcol);
        AVS_LAUNCH_CHECK("layernorm_bwd_reduce");
        return 0;
    }
#define LN_BWD_R(NV, F, R)                                                                                                                     \
    do {                                                                                                                                       \
        if (dres && dres_bf16) ln_bwd_kernel<NV, F, 2, R><<<grid, block, 0, stream>>>(dy, x, mean, rstd, g0, g1, row_mod, out_map, dres, dx, dx_bf16, ws, rows, dx8, q8);   \
        else if (dres) ln_bwd_kernel<NV, F, 1, R><<<grid, block, 0, stream>>>(dy, x, mean, rstd, g0, g1, row_mod, out_map, dres, dx, dx_bf16, ws, rows, dx8, q8);   \
        else ln_bwd_kernel<NV, F, 0, R><<<grid, block, 0, stream>>>(dy, x, mean, rstd, g0, g1, row_mod, out_map, dres, dx, dx_bf16, ws, rows, dx8, q8);       \
    } while (0)
#define LN_BWD(NV, F)                                                                                                                          \
    do {                                                                                                                                       \
        if (rpw == 16) LN_BWD_R(NV, F, 16); else if (rpw == 8) LN_BWD_R(NV, F, 8); else LN_BWD_R(NV, F, 4);                                       \
    } while (0)
    if (dy_f32) {
        if (D == 512) LN_BWD(2, true); else if (D == 768) LN_BWD(3, true); else if (D == 1024) LN_BWD(4, true); else LN_BWD(5, true);
    } else {
        if (D == 512) LN_BWD(2, false); else if (D == 768) LN_BWD(3, false); else if (D == 1024) LN_BWD(4, false); else LN_BWD(5, false);
    }
#undef LN_BWD
#undef LN_BWD_R
    AVS_LAUNCH_CHECK("layernorm_bwd");
    if (!reduce) return 0;
    const int chunks = avs_tuning().det ? 1 : nblocks < LN_REDUCE_CHUNKS ? nblocks : LN_REDUCE_CHUNKS;            // (det: one adder per address)
    ln_bwd_reduce_kernel<<<dim3(ceil_div(D, 256), LN_SETS, chunks), 256, 0, stream>>>(ws, nblocks, D, dg0, db0, dg1, db1, dcol);
    AVS_LAUNCH_CHECK("layernorm_bwd_reduce");
    return 0;
}
