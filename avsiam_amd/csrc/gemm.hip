// k3/k5/k9 - bf16 MFMA GEMMs of the AVSiam hot path (gfx950 only).
//
//   gemm_nt : out[M,N] = epilogue( A[M,K] . B[N,K]^T )       forward Linear (B = weight, as stored by nn.Linear) and
//                                                            dgrad (A = dY, B = W^T copy)
//   gemm_tn : C[N1,N2] += A[M,N1]^T . B[M,N2]                wgrad (A = dY, B = layer input), fp32 atomics
//
// Replaces Attention.qkv/.proj (/root/reference/src/models/cav_mae_base.py:51,55,60,77), timm Mlp fc1/fc2 (:138-143),
// PatchEmbed.proj (:96-99), decoder_embed / decoder_pred_* (:600,634-635) and their autograd backward.
// Four kernels.  All stream their operands HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip); the LDS image
// is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and again on the read; fp32
// accumulation on v_mfma_f32_16x16x32_bf16 (nt) / v_mfma_f32_32x32x16_bf16 (tn, operands by ds_read_b64_tr_b16).
//   gemm_nt8_kernel  256x256x64 tile, 8-phase schedule (two wave groups offset by a barrier, 16-KiB staging granules,
//                    counted vmcnt): the forward/dgrad GEMMs whose tiling fills the chip - whole rounds of tiles
//   gemm_nt_kernel   two-buffer schedule, one barrier per K-step: <.,4,8> 256x256 (here: the rows left over after the
//                    whole rounds, as 128x256 tiles) and <.,2,4> 128x128 (small problems, 2 workgroups per CU)
//   gemm_tn8_kernel  256x256 wgrad tile, 8-phase schedule, fp32 atomics; from 12 output tiles and 16384 rows
//   gemm_tn_kernel   two-buffer wgrad, 128x128 (<2,2>) or 256x256 (<4,4>, A/B reference) tiles
// One epilogue (nt_epilogue) serves all nt kernels: LDS-staged transposition to 16-B/lane row stores, bias, fp32 residual,
// column-range scale, GELU dual output, GELU', fused column sum.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <algorithm>
#include <numeric>

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

constexpr int BM = 128, BN = 128, BK = 64;

struct GemmNtArgs {
    const bf16_t* A; long long lda;
    const bf16_t* B; long long ldb;
    int M, N, K;
    const float* bias;
    const float* res; long long ldr; const int* res_idx;
    const bf16_t* aux; long long ldaux;
    void* out; long long ldo; int out_f32;
    bf16_t* out2; long long ldo2;
    float alpha; int act;                    // 0 none | 1 gelu (out = gelu'(x), out2 = gelu(x), x = the pre-activation) | 2 gelu-backward (aux = gelu'(x))
    int scale_cols; float col_scale;         // columns [0, scale_cols) are multiplied by col_scale as well (scale_cols % 64 == 0)
    float* colsum;                           // bf16 output only: colsum[n] += sum_m out[m][n] (bias gradient of the layer that produced A)
    int m_full;                              // rows [0, m_full) in full tiles, [m_full, M) in half-height tiles (m_full == M: none)
    // two weight sets in one launch (avs_gemm_nt_bf16_dual): rows [m_split, M) use B2 / bias2 / colsum2 instead of
    // B / bias / colsum.  m_split is a multiple of 256, so a tile (full or half-height) never straddles it.
    int m_split; const bf16_t* B2; const float* bias2; float* colsum2;
    // fp8-forward mode, act 1 only: out8 (may be NULL) receives e4m3(clamp(gelu(x) * out8_scale, +-448)) - the operand of the fc2 GEMM
    uint8_t* out8; long long ldo8; float out8_scale;
    // fp8 with delayed scaling: device records (common.h AVS_Q_*; all may be NULL -> the host floats above apply).  qa / qw / qw2: the
    // operands' records - the de-quantisation factor is qa[INV] * qw[INV] (qw2 for rows from m_split); q8: the record of out8 - its
    // scale is read from it and the largest |gelu(x)| written is folded into its running amax
    const float* qa; const float* qw; const float* qw2; float* q8;
    float q8_seen;                           // set by the kernel: the amax q8 held at kernel start (filter of the epilogue's atomic)
    int tb;                                  // gemm_nt8_kernel: tiles [0, tb) are of the FIRST height class (a multiple of 8 and of N / 256); 0: one class
    // fp8 backward (modes 2 / 3): gelu'(x) travels from the fc1 forward epilogue (ACT 1: `out`) to the fc2 input-gradient epilogue (ACT 2: `aux`) as an
    // 8-bit fixed-point code instead of bf16 - (g' + GP8_BIAS) * GP8_STEPS in [0, 255], g' in [-0.129, 1.129]: a step of 0.005, finer than bf16 near 1 and
    // far below the e5m2 rounding (2 mantissa bits) of the gradient it multiplies.  ldo / ldaux then count bytes.
    int aux8;
};
#define GP8_BIAS 0.1296875f
#define GP8_STEPS 202.0f


// Epilogue, staged through LDS (free once the main loop is done).  The accumulators hold 16-row x 4-column patches
// per lane; written as they stand, one store instruction would touch 16 rows x 32 B, and the per-CU store path is
// ISSUE-bound (a measured ~10 B/clk/CU with 8-byte stores made the epilogue cost 20-50 % of these GEMMs).  Each wave
// therefore transposes one 16x64 fp32 sub-tile at a time through a private LDS patch (272-B row stride: conflict-free)
// and re-reads it so that every lane owns 16 BYTES of one output row:
//   fp32 output: row = i*4 + lane/16, 4 columns (lane%16)*4        -> dwordx4 stores, 256-B row segments
//   bf16 output: row = i*8 + lane/8,  8 columns (lane%8)*8         -> dwordx4 stores, 128-B row segments
// Residual reads (fp32) and the GELU' operand read (bf16) use the same ownership, i.e. are 16 B per lane as well.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// Diagnostic builds (tools/ab_lib.sh with AVSIAM_HIPCC_EXTRA=-DNT8_ABLATE=n; never defined in the product build):
//   1  the epilogue computes everything but issues no global store      2  no epilogue at all (accumulators kept alive)
//   3  as 2, and every tile streams the operands of tile 0 (all operand loads hit the L2: main loop without the memory system)
//   4  as 2, with the phase's 16 MFMAs of 16x16x32 replaced by 8 of 32x32x16 on the same registers (timing only, wrong results)
#ifndef NT8_ABLATE
#define NT8_ABLATE 0
#endif
#if NT8_ABLATE == 1
#define NT_STORE(ptr, val) asm volatile("" ::"v"(val))
#else
#define NT_STORE(ptr, val) *(ptr) = (val)
#endif

__device__ __forceinline__ void epi_apply4(float alpha, int ACT, float (&v)[4], const float4& bias4, uint2 p,
                                           bool has_res, f32x4 r) {
    v[0] += bias4.x; v[1] += bias4.y; v[2] += bias4.z; v[3] += bias4.w;
    if (ACT == 2) {                           // p = gelu'(pre-activation), evaluated once by the forward epilogue (ACT 1)
        v[0] *= __uint_as_float(p.x << 16);
        v[1] *= __uint_as_float(p.x & 0xffff0000u);
        v[2] *= __uint_as_float(p.y << 16);
        v[3] *= __uint_as_float(p.y & 0xffff0000u);
    }
    if (has_res) { v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3]; }
    v[0] *= alpha; v[1] *= alpha; v[2] *= alpha; v[3] *= alpha;
}

// LDS traffic of the epilogue goes through inline asm (like the fragment reads of the 8-phase loops): to hipcc an LDS-DMA in
// flight is a pending LDS store it cannot disambiguate, so beside the next tile's first K-tile DMAs it would put
// s_waitcnt vmcnt(0) in front of every compiler-visible ds_read - which also waits for every global store issued so far.
// An asm statement is opaque to that logic; completion is waited for by hand (epi_lds_wait).
template <int OFF>
__device__ __forceinline__ void epi_lds_w128(unsigned addr, const f32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void epi_lds_r128(f32x4& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void epi_lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Operands the epilogue reads from HBM (fp32 residual rows / bf16 GELU' pre-activations) are loaded in groups that are
// double-buffered in registers: a load in the epilogue is synchronous (its latency is paid by the one workgroup of the
// CU), so group g+1 is in flight while group g is transposed and stored, and group 0 is issued by the caller BEFORE the
// barrier / next-tile prefetch that precede the epilogue.
template <int MI>
struct EpiPrefetch {
    static constexpr int RG = 1;                 // residual: 16 rows per group      (4 float4 per 16 rows -> 16 VGPRs / group)
    static constexpr int AG = MI < 4 ? MI : 4;                 // GELU' aux: 4 x 16 rows per group, single-buffered (32 VGPRs; a second
                                                 // buffer spills next to the 128 accumulator registers and costs more than it hides)
    static constexpr int NB = 3;                 // residual buffers: NB - 1 groups in flight while one is transposed and stored (two buffers until round
                                                 // 5; the third is worth 2 - 4 % on the K = 512 / 768 residual GEMMs, a fourth spills; the epilogue is bound
                                                 // by what a CU can read from HBM, ~15 - 25 GB/s, not by round trips: DESIGN.md 5e item 13)
    f32x4 rs[NB][RG][4];                         // [buffer][16-row block][row quad]   (ext vectors: HIP's float4/uint4
    u32x4 ax[AG][2];                             //  structs in arrays end up in scratch)
};

template <int MI>
__device__ __forceinline__ void epi_load_res(const GemmNtArgs& a, f32x4 (&rs)[EpiPrefetch<MI>::RG][4], int grp, int lane, int mw0, int nw0) {
    const int n = nw0 + (lane & 15) * 4, rq = lane >> 4;
    // the row gather (res_idx) is decided ONCE, outside the loads: a per-element "index or row" select makes hipcc branch
    // around every index load and wait vmcnt(0) in front of every residual load - serialising them behind each other
    // and behind the LDS-DMAs of the next tile that are in flight during the epilogue
    if (a.res_idx) {
        int rrow[EpiPrefetch<MI>::RG][4];
#pragma unroll
        for (int mj = 0; mj < EpiPrefetch<MI>::RG; ++mj)
#pragma unroll
            for (int i = 0; i < 4; ++i) rrow[mj][i] = a.res_idx[min(mw0 + (grp * EpiPrefetch<MI>::RG + mj) * 16 + i * 4 + rq, a.M - 1)];
#pragma unroll
        for (int mj = 0; mj < EpiPrefetch<MI>::RG; ++mj)
#pragma unroll
            for (int i = 0; i < 4; ++i) rs[mj][i] = *reinterpret_cast<const f32x4*>(a.res + (long long)rrow[mj][i] * a.ldr + n);
    } else {
#pragma unroll
        for (int mj = 0; mj < EpiPrefetch<MI>::RG; ++mj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = min(mw0 + (grp * EpiPrefetch<MI>::RG + mj) * 16 + i * 4 + rq, a.M - 1);
                rs[mj][i] = *reinterpret_cast<const f32x4*>(a.res + (long long)m * a.ldr + n);
            }
    }
}

template <int MI>
__device__ __forceinline__ void epi_load_aux(const GemmNtArgs& a, u32x4 (&ax)[EpiPrefetch<MI>::AG][2], int grp, int lane, int mw0, int nw0) {
    const int n = nw0 + (lane & 7) * 8, rq = lane >> 3;
#pragma unroll
    for (int mj = 0; mj < EpiPrefetch<MI>::AG; ++mj)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = min(mw0 + (grp * EpiPrefetch<MI>::AG + mj) * 16 + i * 8 + rq, a.M - 1);
            if (a.aux8) {                                  // 8 codes = 8 bytes per lane (block-uniform branch)
                const uint2 c = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(a.aux) + (size_t)m * a.ldaux + n);
                ax[mj][i] = u32x4{c.x, c.y, 0u, 0u};
            } else
            ax[mj][i] = *reinterpret_cast<const u32x4*>(a.aux + (size_t)m * a.ldaux + n);
        }
}

// group 0 of whatever this epilogue will read; called before the barrier that ends the main loop
template <int ACT, int MI>
__device__ __forceinline__ void nt_epilogue_prefetch(const GemmNtArgs& a, EpiPrefetch<MI>& pf, int lane, int mw0, int nw0) {
    if (ACT == 0 && a.out_f32 && a.res) {
#pragma unroll
        for (int g = 0; g < EpiPrefetch<MI>::NB - 1 && g < MI / EpiPrefetch<MI>::RG; ++g) epi_load_res<MI>(a, pf.rs[g], g, lane, mw0, nw0);
    }
}

// called by the kernels BEFORE they put the next tile's LDS-DMAs in flight: the one global load of the bias path is then
// waited for on its own (beside a pending DMA hipcc's wait for it is vmcnt(0), i.e. it would wait for the DMAs as well)
__device__ __forceinline__ void nt_epilogue_stage_bias(const GemmNtArgs& a, char* smem, int wave, int lane, int mw0, int nw0) {
    float* sbias = reinterpret_cast<float*>(smem) + (blockDim.x >> 6) * (16 * 68) + wave * 64;       // behind the waves' transpose patches
    const float* bias = mw0 >= a.m_split ? a.bias2 : a.bias;
    sbias[lane] = bias ? bias[nw0 + lane] : 0.f;
}

// Q8: 0 no fp8 copy | 1 forward fp8 GEMM: ACT 1 also writes out8 = e4m3(gelu(x)) | 2 backward fp8 GEMM: out8 = e5m2(out) (ACT 0 / 2)
// nrow: rows of the wave's MI x 16 that belong to the tile (a multiple of 16; fewer than MI x 16 in the reduced-height tiles of gemm_nt8_kernel)
template <int ACT, int MI, int Q8 = 0>
__device__ __forceinline__ void nt_epilogue(const GemmNtArgs& a, f32x4 (&acc)[4][MI], EpiPrefetch<MI>& pf, char* smem, int wave, int lane,
                                            int mw0, int nw0, const int nrow = MI * 16) {
    const int fr = lane & 15, fq = lane >> 4;
    // this wave's 16 x 68-float transpose patch and its 64 staged bias values, as LDS byte addresses for the asm accessors
    const unsigned smem_lds = (unsigned)(size_t)(LDS_AS const char*)smem;
    const unsigned stg_lds = smem_lds + wave * (16 * 68) * 4;
    const unsigned sbias_lds = smem_lds + ((blockDim.x >> 6) * (16 * 68) + wave * 64) * 4;       // staged by nt_epilogue_stage_bias
    const unsigned waddr = stg_lds + (fr * 68 + fq * 4) * 4;                       // accumulator block ni goes 64 B further
    const float alpha = nw0 < a.scale_cols ? a.alpha * a.col_scale : a.alpha;      // wave-uniform (a wave owns 64 columns)
    float* const colsum = mw0 >= a.m_split ? a.colsum2 : a.colsum;                 // wave-uniform (a tile lies on one side of m_split)
    // The wave's 64 bias values go through LDS (one coalesced load per tile, re-read per row group with ds_read): kept in
    // registers they do not fit next to 128 accumulators, and hipcc then RE-LOADS them from global memory in front of
    // every row group - each reload a vmcnt(0) that also waits for the stores just issued, i.e. a fully serialised
    // store -> load -> store chain over the whole epilogue.
    if (ACT == 0 && a.out_f32) {                                                   // fp32 output exists without activation only
        const int cc = (lane & 15) * 4, rq = lane >> 4;
        const int n = nw0 + cc;
        float* stg = reinterpret_cast<float*>(smem) + wave * (16 * 68);
        const float* sbias = reinterpret_cast<const float*>(smem) + (blockDim.x >> 6) * (16 * 68) + wave * 64;
        constexpr int RG = EpiPrefetch<MI>::RG, NG = MI / RG, NB = EpiPrefetch<MI>::NB;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g * RG * 16 >= nrow) continue;
            if (a.res && g + NB - 1 < NG) epi_load_res<MI>(a, pf.rs[(g + NB - 1) % NB], g + NB - 1, lane, mw0, nw0);
#pragma unroll
            for (int mj = 0; mj < RG; ++mj) {
                const int mi = g * RG + mj;
                // (compiler-visible LDS accesses on this path: it never runs beside a pending LDS-DMA, and hipcc interleaves
                //  these reads with the residual loads better than a hand-placed lgkmcnt(0) per row block does: -4..9 % measured)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f32x4*>(stg + fr * 68 + ni * 16 + fq * 4) = acc[ni][mi];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rr = i * 4 + rq;
                    const float4 t = *reinterpret_cast<const float4*>(stg + rr * 68 + cc);
                    const int m = mw0 + mi * 16 + rr;
                    if (m >= a.M) continue;
                    float v[4] = {t.x, t.y, t.z, t.w};
                    const float4 bias4 = *reinterpret_cast<const float4*>(sbias + cc);
                    epi_apply4(alpha, 0, v, bias4, make_uint2(0, 0), a.res != nullptr, pf.rs[g % NB][mj][i]);
                    NT_STORE(reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + (size_t)m * a.ldo + n), (f32x4{v[0], v[1], v[2], v[3]}));
                }
            }
        }
    } else {
        const int cc = (lane & 7) * 8, rq = lane >> 3;
        const int n = nw0 + cc;
        const unsigned raddr = stg_lds + (rq * 68 + cc) * 4, baddr = sbias_lds + cc * 4;      // rows rq and rq + 8: 2176 B apart
        constexpr int AG = EpiPrefetch<MI>::AG, NG = MI / AG;
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // this lane's 8 columns, summed over the rows it handles
        const float q8s = (Q8 != 0 && a.out8) ? a.out8_scale : 0.f;          // (with a record: its scale, read by the kernel at its start)
        const float q8seen = a.q8_seen;
        float q8max = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g * AG * 16 >= nrow) continue;
            if (ACT == 2) epi_load_aux<MI>(a, pf.ax, g, lane, mw0, nw0);
#pragma unroll
            for (int mj = 0; mj < AG; ++mj) {
                const int mi = g * AG + mj;
                if (mi * 16 >= nrow) continue;
                epi_lds_w128<0>(waddr, acc[0][mi]); epi_lds_w128<64>(waddr, acc[1][mi]);
                epi_lds_w128<128>(waddr, acc[2][mi]); epi_lds_w128<192>(waddr, acc[3][mi]);
                f32x4 tq[2][2], bq[2];
                epi_lds_r128<0>(tq[0][0], raddr); epi_lds_r128<16>(tq[0][1], raddr);
                epi_lds_r128<2176>(tq[1][0], raddr); epi_lds_r128<2176 + 16>(tq[1][1], raddr);
                epi_lds_r128<0>(bq[0], baddr); epi_lds_r128<16>(bq[1], baddr);
                epi_lds_wait();
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int rr = i * 8 + rq;
                    const int m = mw0 + mi * 16 + rr;
                    if (m >= a.M) continue;
                    float v0[4] = {tq[i][0][0], tq[i][0][1], tq[i][0][2], tq[i][0][3]}, v1[4] = {tq[i][1][0], tq[i][1][1], tq[i][1][2], tq[i][1][3]};
                    // (no residual here: it exists for fp32 output only - an optional per-row load in this loop would put a
                    //  vmcnt(0) at its join point in front of every row group's stores, taken or not)
                    u32x4 ax = {0, 0, 0, 0};
                    if (ACT == 2) ax = pf.ax[mj][i];
                    const f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
                    const float4 bias_lo = make_float4(bq[0][0], bq[0][1], bq[0][2], bq[0][3]), bias_hi = make_float4(bq[1][0], bq[1][1], bq[1][2], bq[1][3]);
                    if (ACT == 2 && Q8 != 1 && a.aux8) {          // the 8-bit gelu': one code per value, four per dword (hipcc turns the byte picks into v_cvt_f32_ubyteN)
                        auto dec = [](float b) { return fmaf(b, 1.0f / GP8_STEPS, -GP8_BIAS); };
                        const float g0[4] = {dec((float)((ax[0] >> 0) & 0xffu)), dec((float)((ax[0] >> 8) & 0xffu)),
                                             dec((float)((ax[0] >> 16) & 0xffu)), dec((float)((ax[0] >> 24) & 0xffu))};
                        const float g1[4] = {dec((float)((ax[1] >> 0) & 0xffu)), dec((float)((ax[1] >> 8) & 0xffu)),
                                             dec((float)((ax[1] >> 16) & 0xffu)), dec((float)((ax[1] >> 24) & 0xffu))};
                        epi_apply4(alpha, 0, v0, bias_lo, make_uint2(0, 0), false, r0);
                        epi_apply4(alpha, 0, v1, bias_hi, make_uint2(0, 0), false, r1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v0[j] *= g0[j]; v1[j] *= g1[j]; }
                    } else {
                    epi_apply4(alpha, ACT, v0, bias_lo, make_uint2(ax[0], ax[1]), false, r0);
                    epi_apply4(alpha, ACT, v1, bias_hi, make_uint2(ax[2], ax[3]), false, r1);
                    }
                    if (colsum) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { cs[j] += v0[j]; cs[4 + j] += v1[j]; }
                    }
                    uint4 o;
                    if (ACT == 1) {
                        // forward of fc1: `out` receives gelu'(x) instead of x.  The pre-activation itself is needed by nothing but
                        // the fc2 input-gradient epilogue (ACT 2), and only through gelu'; evaluated here it shares phi's exponential
                        // with gelu (4 VALU ops more per element) and saves that epilogue 12 of its 14.
                        o.x = pack_bf2(gelu_erf_grad(v0[0]), gelu_erf_grad(v0[1])); o.y = pack_bf2(gelu_erf_grad(v0[2]), gelu_erf_grad(v0[3]));
                        o.z = pack_bf2(gelu_erf_grad(v1[0]), gelu_erf_grad(v1[1])); o.w = pack_bf2(gelu_erf_grad(v1[2]), gelu_erf_grad(v1[3]));
                    } else {
                        o.x = pack_bf2(v0[0], v0[1]); o.y = pack_bf2(v0[2], v0[3]);
                        o.z = pack_bf2(v1[0], v1[1]); o.w = pack_bf2(v1[2], v1[3]);
                    }
                    // (fp8 input-gradient form: `out` may be NULL - every reader of this gradient takes the e5m2 copy below; the bf16 kernels never test it)
                    if (ACT == 1 && Q8 != 2 && a.aux8) {          // gelu'(x) as 8-bit codes (fp8 backward; bf16 with EngineOptions.gelu8): 8 bytes per lane instead of 16
                        auto enc = [](float g, unsigned sel, unsigned old) { return __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g, GP8_STEPS, GP8_BIAS * GP8_STEPS), sel, old); };
                        unsigned c0 = enc(gelu_erf_grad(v0[0]), 0, 0); c0 = enc(gelu_erf_grad(v0[1]), 1, c0); c0 = enc(gelu_erf_grad(v0[2]), 2, c0); c0 = enc(gelu_erf_grad(v0[3]), 3, c0);
                        unsigned c1 = enc(gelu_erf_grad(v1[0]), 0, 0); c1 = enc(gelu_erf_grad(v1[1]), 1, c1); c1 = enc(gelu_erf_grad(v1[2]), 2, c1); c1 = enc(gelu_erf_grad(v1[3]), 3, c1);
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(a.out) + (size_t)m * a.ldo + n) = make_uint2(c0, c1);
                    } else
                    if (Q8 != 2 || a.out) NT_STORE(reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + n), (u32x4{o.x, o.y, o.z, o.w}));
                    if (Q8 == 2 && ACT != 1 && a.out8) {
                        // fp8 backward: the e5m2 copy of this output gradient, the operand of the next input-gradient GEMM
                        auto c8 = [&](float x) { q8max = fmaxf(q8max, fabsf(x)); return __builtin_amdgcn_fmed3f(x * q8s, -57344.f, 57344.f); };
                        int w0 = __builtin_amdgcn_cvt_pk_bf8_f32(c8(v0[0]), c8(v0[1]), 0, false);
                        w0 = __builtin_amdgcn_cvt_pk_bf8_f32(c8(v0[2]), c8(v0[3]), w0, true);
                        int w1 = __builtin_amdgcn_cvt_pk_bf8_f32(c8(v1[0]), c8(v1[1]), 0, false);
                        w1 = __builtin_amdgcn_cvt_pk_bf8_f32(c8(v1[2]), c8(v1[3]), w1, true);
                        *reinterpret_cast<uint2*>(a.out8 + (size_t)m * a.ldo8 + n) = make_uint2((unsigned)w0, (unsigned)w1);
                    }
                    if (ACT == 1) {
                        uint4 gq;
                        gq.x = pack_bf2(gelu_erf(v0[0]), gelu_erf(v0[1])); gq.y = pack_bf2(gelu_erf(v0[2]), gelu_erf(v0[3]));
                        gq.z = pack_bf2(gelu_erf(v1[0]), gelu_erf(v1[1])); gq.w = pack_bf2(gelu_erf(v1[2]), gelu_erf(v1[3]));
                        // (fp8 forward: `out2` may be NULL when fc2 and its weight gradient both read the e4m3 copy)
                        if (Q8 != 1 || a.out2) NT_STORE(reinterpret_cast<u32x4*>(a.out2 + (size_t)m * a.ldo2 + n), (u32x4{gq.x, gq.y, gq.z, gq.w}));
                        if (Q8 == 1 && a.out8) {
                            const float q = q8s;
                            auto q8 = [&](float x) { const float gx = gelu_erf(x); q8max = fmaxf(q8max, fabsf(gx)); return __builtin_amdgcn_fmed3f(gx * q, -448.f, 448.f); };
                            int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q8(v0[0]), q8(v0[1]), 0, false);
                            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q8(v0[2]), q8(v0[3]), w0, true);
                            int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q8(v1[0]), q8(v1[1]), 0, false);
                            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q8(v1[2]), q8(v1[3]), w1, true);
                            *reinterpret_cast<uint2*>(a.out8 + (size_t)m * a.ldo8 + n) = make_uint2((unsigned)w0, (unsigned)w1);
                        }
                    }
                }
            }
        }
        if (Q8 != 0 && a.out8 && a.q8) q_amax_update(a.q8, q8max, q8seen);
        if (colsum) {
            // lanes with the same (lane & 7) hold the same 8 columns for different rows: fold the 8 row groups, then one
            // atomic per column and wave (fp32 atomics, like the weight gradients)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = cs[j];
                t += __shfl_xor(t, 8, 64);
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                cs[j] = t;
            }
            if (rq == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(colsum + n + j, cs[j]);
            }
        }
    }
}

// Tile configurations (NWM = 2 waves along M, NWN waves along N; each wave MI x 4 MFMA tiles of 16x16):
//   <NWN=2, MI=4>: 128x128 tile, 256 threads,  64 KiB LDS, 2 workgroups/CU  - small problems (fills the chip with few rows)
//   <NWN=4, MI=8>: 256x256 tile, 512 threads, 128 KiB LDS, 1 workgroup/CU   - halves the L2->LDS bytes per FLOP, which is
//                  what bounds the 128^2 tile (at full MFMA rate it would need more than the L2 can deliver)
template <int ACT, int NWN, int MI>
__global__ __launch_bounds__(128 * NWN) void gemm_nt_kernel(GemmNtArgs a) {
    constexpr int NT = 128 * NWN;                 // threads
    constexpr int TBM = 2 * MI * 16;              // tile rows (A / activations)
    constexpr int TBN = NWN * 64;                 // tile cols (B rows / weights)
    constexpr int A_BYTES = TBM * 128, B_BYTES = TBN * 128, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int CA = TBM * 8 / NT, CB = TBN * 8 / NT;   // 16-byte chunks staged per thread per K-step
    constexpr int MIH = MI / 2, CAH = CA / 2;     // HALF-height tiles (TBM/2 rows, same LDS image: the A region is half used)
    extern __shared__ __attribute__((aligned(16))) char smem[];            // [buf][A|B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int nt_n = a.N / TBN;
    // Tiles [0, nfull) are TBM x TBN tiles over rows [0, m_full); rows [m_full, M) - what is left after the whole rounds of
    // full tiles, see the host - are cut into half-height tiles so that the last, partial round still covers the chip.
    const int nfull = ((a.m_full + TBM - 1) / TBM) * nt_n;
    const int nhalf = a.m_full < a.M ? ((a.M - a.m_full + TBM / 2 - 1) / (TBM / 2)) * nt_n : 0;
    const int ntiles = nfull + nhalf;
    const int fr = lane & 15, fq = lane >> 4;
    // fragment read offsets: row = base + (lane&15), chunk = kk*4 + (lane>>4), swizzled with row&7 == lane&7
    const int off_k0 = fr * 128 + (((0 + fq) ^ (lane & 7)) << 4);
    const int off_k1 = fr * 128 + (((4 + fq) ^ (lane & 7)) << 4);
    const int nk = a.K / BK;

    // Workgroups are persistent: block b walks tiles b, b + grid, ... (grid is a multiple of 8 or covers every tile, so a
    // block keeps its XCD class and xcd_remap keeps neighbouring tiles - same A panel - on one L2).  The first K-slab of
    // the NEXT tile is put in flight before the epilogue of the current one, and the epilogue's stores drain under the
    // next tile's main loop: with one 128-KiB workgroup per CU nothing else would hide those per-tile costs.
    const bf16_t* srcA[CA];
    const bf16_t* srcB[CB];
    int m0 = 0, n0 = 0;
    bool half = false;                             // kind of the tile srcA/srcB point at (block-uniform)
    auto set_tile = [&](int v) {
        half = v >= nfull;
        if (!half) {
            const int wg = xcd_remap(v, nfull);
            m0 = (wg / nt_n) * TBM;
            n0 = (wg % nt_n) * TBN;
        } else {
            const int h = v - nfull;
            m0 = a.m_full + (h / nt_n) * (TBM / 2);
            n0 = (h % nt_n) * TBN;
        }
#pragma unroll
        for (int i = 0; i < CA; ++i) {             // a half tile uses chunks [0, CAH): rows 0 .. TBM/2-1
            const int p = i * NT + tid, row = p >> 3, c = (p & 7) ^ (row & 7);
            srcA[i] = a.A + (size_t)min(m0 + row, a.M - 1) * a.lda + c * 8;
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int p = i * NT + tid, row = p >> 3, c = (p & 7) ^ (row & 7);
            srcB[i] = (m0 >= a.m_split ? a.B2 : a.B) + (size_t)(n0 + row) * a.ldb + c * 8;
        }
    };
    auto stage = [&](int buf, int k0) {
        char* sa = smem + buf * BUF_BYTES;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < CA; ++i)
            if (i < CAH || !half)
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcA[i] + k0), (LDS_AS void*)(sa + (i * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < CB; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcB[i] + k0), (LDS_AS void*)(sb + (i * NT + wave * 64) * 16), 16, 0, 0);
    };

    int v = blockIdx.x;
    if (v >= ntiles) return;
    set_tile(v);
    stage(0, 0);
    // one tile of 2 x MIT x 16 rows (MIT = MI: full, MI/2: half); srcA/srcB/half describe it on entry and the NEXT tile on exit
    auto run_tile = [&](auto mit) {
        constexpr int MIT = decltype(mit)::value;
        f32x4 acc[4][MIT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MIT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < nk; ++t) {
            // K-slab t must have landed: wait for this wave's LDS-DMA explicitly - hipcc's own vmcnt(0) in front of the
            // barrier is not guaranteed (it is missing from the half-tile instantiation of the GELU kernel)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                               // buffer (t+1)&1 and the epilogue patches are free
            if (t + 1 < nk) stage((t + 1) & 1, (t + 1) * BK);
            const char* sa = smem + (t & 1) * BUF_BYTES + wm * (MIT * 16) * 128;
            const char* sb = smem + (t & 1) * BUF_BYTES + A_BYTES + wn * 64 * 128;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int off = kk ? off_k1 : off_k0;
                bf16x8 wf[4], xf[MIT];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sb + i * 16 * 128 + off);
#pragma unroll
                for (int i = 0; i < MIT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * 128 + off);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int mi = 0; mi < MIT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        const int em = m0 + wm * (MIT * 16), en = n0 + wn * 64;
        __syncthreads();                                   // every wave is done reading the K-slab buffers
        // both LDS buffers are free now: prefetch the next tile's first slab into buffer 0, stage the epilogue through buffer 1
        nt_epilogue_stage_bias(a, smem + BUF_BYTES, wave, lane, em, en);
        // (the epilogue runs with no LDS-DMA in flight, see gemm_nt8_kernel: the next tile's first slab is requested after it)
        const bool has_next = v + (int)gridDim.x < ntiles;
        EpiPrefetch<MIT> pf;
        nt_epilogue_prefetch<ACT, MIT>(a, pf, lane, em, en);
        nt_epilogue<ACT, MIT>(a, acc, pf, smem + BUF_BYTES, wave, lane, em, en);
        if (has_next) {
            set_tile(v + gridDim.x);
            stage(0, 0);
        }
    };
    for (; v < ntiles; v += gridDim.x) {
        if (!half) run_tile(std::integral_constant<int, MI>{});
        else run_tile(std::integral_constant<int, MIH>{});
    }
}

// ---------------------------------------------------------------------------------------------------
// 8-phase variant of the 256x256 tile (after the guide's "256^2 8-phase template": this schedule is derived for THIS
// kernel's LDS image and epilogue).  What it changes against gemm_nt_kernel<.,4,8>:
//   * the two wave groups (G0 = waves 0-3: output rows 0-127, G1 = waves 4-7: rows 128-255) run the same phase sequence
//     offset by ONE barrier: while one group's MFMA segment runs, the other group issues its fragment reads and LDS-DMAs,
//     so the matrix pipe of every SIMD always has a wave with operands in registers;
//   * a K-tile (64 deep) is multiplied in four phases of 16 MFMAs per wave - output quadrants (a0,b0) (a0,b1) (a1,b1)
//     (a1,b0), a = 64-row half of the wave's 128 rows, b = 32-column half of its 64 columns; both b halves stay in
//     registers, so a K-tile's LDS reads are a0+b0 | b1 | a1 | none;
//   * staging granules are 16 KiB (A rows of one group, or the b0 / b1 column halves of all waves); each is re-filled
//     for K-tile t+2 as soon as its last reader is a barrier past it: phase 1: A1(t+1), 2: B1(t+1), 3: B0(t+2), 4: A0(t+2).
//     Three granules are in flight across every barrier (counted vmcnt(6), never 0 inside the loop) instead of one
//     64-KiB slab that has to land completely before the next K-step starts.
// Hazards (barrier b_i; G0's load segment of phase p of K-tile t lies in (b_{8t+2p-3}, b_{8t+2p-2}), G1's one later):
//   WAR  a granule is re-filled at least one full barrier interval after the lgkmcnt(0) of its last reader;
//   RAW  every wave waits for its own share of a granule (counted vmcnt) in the load segment BEFORE the one that reads it.
// counted wait on the vector-memory queue as a REAL s_waitcnt (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 |
// vmcnt[5:4] << 14): hipcc's waitcnt pass reads it and retires the loads it covers from its scoreboard - with an asm
// statement it cannot see, it re-waits (vmcnt(0), every iteration) for epilogue loads whose registers the loop reuses
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

// FP8 = 1 (avs_gemm_nt_fp8): the operands are OCP fp8 (e4m3) instead of bf16; FP8 = 2: the A operand (activation side) is e5m2 - a gradient.  Byte for byte the kernel is the same - an fp8
// matrix [M, K8] with leading dimension lda8 is handed over as the bf16 matrix [M, K8 / 2] / lda8 / 2 it aliases, so tiles, DMA
// granules, waits and the epilogue do not change; a 128-byte K-tile row then holds 128 contraction values instead of 64, each
// lane's two 16-byte fragment chunks are the ADJACENT chunks 2g, 2g + 1 of its row (lane group g: 32 consecutive values) instead of
// chunks g and 4 + g, and a phase issues 8 v_mfma_f32_16x16x128_f8f6f4 (32 cycles each) instead of 16 v_mfma_f32_16x16x32_bf16.
// Tile HEIGHT classes.  A tile is 2 x HG rows (one half per wave group), HG = 64 + 16 NB, NB = 16-row blocks in the second 64-row half of
// a group's rows: 4 = the 256-row tile, 3 = a 224-row tile (the fragment reads, MFMAs and epilogue row blocks of the missing block do not
// exist; the LDS image and the DMA granules keep their 128-row shape and carry rows nobody reads).  The persistent kernel pays whole
// ROUNDS of tiles over the CUs, so the host fills the last round exactly by mixing two heights: tiles [0, tb) (in dispatch order: every
// workgroup meets them first, so every CU gets the same share) have NBA blocks and cover rows [0, (tb / nt_n) * HA), the others NBB.
// The two classes are two inlined copies of the tile body (NB is a compile-time constant in each).
template <int ACT, int FP8 = 0, int NBA = 4, int NBB = 4>
__global__ __launch_bounds__(512) void gemm_nt8_kernel(GemmNtArgs a) {
    constexpr int NT = 512, TBN = 256, MI = 8;
    constexpr int HA = 128 + 32 * NBA, HB = 128 + 32 * NBB;     // tile heights of the two classes
    constexpr int A_BYTES = 256 * 128, BUF_BYTES = 2 * A_BYTES, GR = 16384;     // per K-tile buffer: [A 32 KiB | B 32 KiB] (whatever the tile height)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wc = wave & 3;                   // wave group (row half) and 64-column slice
    const int nt_n = a.N / TBN;
    const int tb = NBA == NBB ? 0 : a.tb;                     // tiles of the first class
    const int rows_a = (tb / nt_n) * HA;                      // rows they cover
    const int ntiles = tb + ((a.M - rows_a + HB - 1) / HB) * nt_n;
    const int fr = lane & 15, fq = lane >> 4;
    const int off_k0 = fr * 128 + (((FP8 ? 2 * fq : 0 + fq) ^ (lane & 7)) << 4);
    const int off_k1 = fr * 128 + (((FP8 ? 2 * fq + 1 : 4 + fq) ^ (lane & 7)) << 4);
    const int nk = a.K / BK;
    const unsigned lds_base = (unsigned)(size_t)(LDS_AS const char*)smem;
    // fp8 with device records: the de-quantisation factors (one per weight set), the scale of the e4m3 / e5m2 output copy and the amax
    // that record already holds are read ONCE, here, before any LDS-DMA is in flight, and kept in scalar registers - read in front of the
    // epilogue (vector loads: the compiler cannot prove the records invariant) they were waited for with vmcnt(0) beside the next
    // tile's DMAs, i.e. every epilogue waited for the next tile's first K-tile to land
    float dq_a = a.alpha, dq_b = a.alpha, q8_scale = a.out8_scale, q8_seen = 0.f;
    if (FP8) {
        if (a.qa) {
            const float ia = a.qa[AVS_Q_INV];
            dq_a = ia * a.qw[AVS_Q_INV];
            dq_b = ia * a.qw2[AVS_Q_INV];
        }
        if (a.out8 && a.q8) { q8_scale = a.q8[AVS_Q_SCALE]; q8_seen = a.q8[AVS_Q_AMAX]; }
        dq_a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dq_a)));
        dq_b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dq_b)));
        q8_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q8_scale)));
        q8_seen = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q8_seen)));
    }

    // DMA bookkeeping.  Granule-local chunk p = j*512 + tid (j = 0,1): local row lr = p >> 3, slot p & 7, source chunk
    // (p & 7) ^ (lr & 7).  A granules: physical row = h*128 + lr; B granules: physical row = (lr>>5)*64 + h*32 + (lr&31).
    const bf16_t* sA[2][2];                                   // [granule half][j]
    const bf16_t* sB[2][2];
    int m0 = 0, n0 = 0;
    auto set_tile = [&](int v) {
#if NT8_ABLATE == 3
        const int wg = 0 * v;                                 // every tile streams tile 0's operands: all loads hit the L2
#else
        // (tb is a multiple of 8, so a dispatch index keeps its XCD class inside either range and xcd_remap stays a per-XCD contiguous run)
        const bool first = v < tb;
        const int wg = first ? xcd_remap(v, tb) : xcd_remap(v - tb, ntiles - tb);
#endif
        const int hg = first ? HA / 2 : HB / 2;               // rows per wave group of this tile
        m0 = first ? (wg / nt_n) * HA : rows_a + (wg / nt_n) * HB;
        n0 = (wg % nt_n) * TBN;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = j * NT + tid, lr = p >> 3, c = (p & 7) ^ (lr & 7);
                sA[h][j] = a.A + (size_t)min(m0 + h * hg + lr, a.M - 1) * a.lda + c * 8;
                sB[h][j] = (m0 >= a.m_split ? a.B2 : a.B) + (size_t)(n0 + (lr >> 5) * 64 + h * 32 + (lr & 31)) * a.ldb + c * 8;
            }
    };
    // LDS destination of this wave's j-th instruction of a granule (wave-uniform; the hardware adds lane*16)
    auto dma_a = [&](int t, int h) {
        char* dst = smem + (t & 1) * BUF_BYTES + h * GR;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(sA[h][j] + t * BK), (LDS_AS void*)(dst + (j * NT + wave * 64) * 16), 16, 0, 0);
    };
    auto dma_b = [&](int t, int h) {
        char* dst = smem + (t & 1) * BUF_BYTES + A_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int lr0 = j * 64 + wave * 8;                // first of the 8 local rows this wave-instruction covers
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(sB[h][j] + t * BK),
                                             (LDS_AS void*)(dst + ((lr0 >> 5) * 64 + h * 32 + (lr0 & 31)) * 128), 16, 0, 0);
        }
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    int v = blockIdx.x;
    if (v >= ntiles) return;
    set_tile(v);
    dma_a(0, 0); dma_a(0, 1); dma_b(0, 0); dma_b(0, 1);
    // one tile; srcA / srcB describe it on entry and the NEXT tile on exit.  nbc: its height class (see above)
    auto run_tile = [&](auto nbc) {
        constexpr int NB1 = decltype(nbc)::value, HG = 64 + 16 * NB1;
        f32x4 acc[4][MI];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if NT8_ABLATE == 4
        f32x16 acc16[2][2][2];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc16[i >> 2][(i >> 1) & 1][i & 1][r] = 0.f;
#endif
        // K-tile 0 (8 DMAs per thread, issued before the previous epilogue) must land; after the barrier every wave has
        // left that epilogue, so buffer 1 (its LDS patches) may take B0(1), A0(1)
        wait_vm<0>();
        bar();
        if (nk > 1) { dma_b(1, 0); dma_a(1, 0); }
        if (g == 1) bar();                                    // the one-barrier offset between the groups
        bf16x8 xf[4][2], wb[2][2][2];                         // a half: [mi][kk];  b halves: [bh][ni][kk]
        for (int t = 0; t < nk; ++t) {
            // fragment reads go through inline asm: hipcc treats an LDS-DMA in flight as a pending store to LDS and, unable to
            // tell the granule being re-filled from the one being read (same K-tile buffer), would put s_waitcnt vmcnt(0)
            // in front of the first ds_read of the K-tile - draining the whole ring.  lgkmcnt is waited for by hand.
            const unsigned la = lds_base + (t & 1) * BUF_BYTES + (g * 128) * 128;
            const unsigned lb = lds_base + (t & 1) * BUF_BYTES + A_BYTES + (wc * 64) * 128;
            const unsigned va0 = la + off_k0, va1 = la + off_k1, vb0 = lb + off_k0, vb1 = lb + off_k1;
            const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
#define NT8_RD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF) : "memory")
#define NT8_RD_A(AH)                                                                                          \
    NT8_RD(xf[0][0], va0, (AH) * 8192 + 0);    NT8_RD(xf[0][1], va1, (AH) * 8192 + 0);                          \
    NT8_RD(xf[1][0], va0, (AH) * 8192 + 2048); NT8_RD(xf[1][1], va1, (AH) * 8192 + 2048);                       \
    NT8_RD(xf[2][0], va0, (AH) * 8192 + 4096); NT8_RD(xf[2][1], va1, (AH) * 8192 + 4096);                       \
    NT8_RD(xf[3][0], va0, (AH) * 8192 + 6144); NT8_RD(xf[3][1], va1, (AH) * 8192 + 6144)
#define NT8_RD_B(BH)                                                                                          \
    NT8_RD(wb[BH][0][0], vb0, (BH) * 4096 + 0);    NT8_RD(wb[BH][0][1], vb1, (BH) * 4096 + 0);                  \
    NT8_RD(wb[BH][1][0], vb0, (BH) * 4096 + 2048); NT8_RD(wb[BH][1][1], vb1, (BH) * 4096 + 2048)
            auto mma = [&](int ah, int bh) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this phase's fragment reads (issued before the barrier)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#if NT8_ABLATE == 4
                // timing only (results are meaningless): the same fragments and flops through 8 MFMAs of 32x32x16
#pragma unroll
                for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        acc16[bh][ah][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[bh][kq >> 1][kq & 1], xf[mb * 2 + (kq >> 1)][kq & 1], acc16[bh][ah][mb], 0, 0, 0);
#else
                if (FP8) {
                    typedef __attribute__((ext_vector_type(8))) int i32x8;
                    typedef __attribute__((ext_vector_type(4))) int i32x4;
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const i32x4 w0 = __builtin_bit_cast(i32x4, wb[bh][ni][0]), w1 = __builtin_bit_cast(i32x4, wb[bh][ni][1]);
                            const i32x4 x0 = __builtin_bit_cast(i32x4, xf[mi][0]), x1 = __builtin_bit_cast(i32x4, xf[mi][1]);
                            const i32x8 wv = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                            const i32x8 xv = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                            // cbsz (first operand: the weight) = 0: e4m3; blgp (second operand: the activation side) = 0: e4m3 (forward),
                            // 1: e5m2 (FP8 == 2: the gradient operand of an input-gradient GEMM); scale selectors 0 = the unscaled instruction
                            acc[bh * 2 + ni][ah * 4 + mi] =
                                __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, xv, acc[bh * 2 + ni][ah * 4 + mi], 0, FP8 == 2 ? 1 : 0, 0, 0, 0, 0);
                        }
                } else {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni) {
                                if (ah == 1 && mi >= NB1) continue;               // (folded: ah and mi are constants after inlining)
                                acc[bh * 2 + ni][ah * 4 + mi] =
                                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[bh][ni][kk], xf[mi][kk], acc[bh * 2 + ni][ah * 4 + mi], 0, 0, 0);
                            }
                }
#endif
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            };
            // ---- phase 1: (a0, b0); re-fill A1 of the next K-tile; B1(t) must have landed for phase 2
            NT8_RD_B(0); NT8_RD_A(0);
            if (n1) { dma_a(t + 1, 1); wait_vm<6>(); } else wait_vm<0>();
            bar(); mma(0, 0); bar();
            // ---- phase 2: (a0, b1)
            NT8_RD_B(1);
            if (n1) dma_b(t + 1, 1);
            bar(); mma(0, 1); bar();
            // ---- phase 3: (a1, b1)
            // (a reduced-height tile reads only the blocks it multiplies: an asm read whose result no instruction uses would leave its
            //  registers free for re-use while the LDS data is still on its way)
            if (NB1 == 4) { NT8_RD_A(1); }
            else {
                NT8_RD(xf[0][0], va0, 8192 + 0);    NT8_RD(xf[0][1], va1, 8192 + 0);
                NT8_RD(xf[1][0], va0, 8192 + 2048); NT8_RD(xf[1][1], va1, 8192 + 2048);
                if (NB1 > 2) { NT8_RD(xf[2][0], va0, 8192 + 4096); NT8_RD(xf[2][1], va1, 8192 + 4096); }
            }
            if (n2) dma_b(t + 2, 0);
            bar(); mma(1, 1); bar();
            // ---- phase 4: (a1, b0); A1(t+1) (and everything older) must have landed for the next K-tile's phase 1
            if (n2) { dma_a(t + 2, 0); wait_vm<6>(); }
            else if (n1) wait_vm<2>();
            bar(); mma(1, 0); bar();
        }
#undef NT8_RD
#undef NT8_RD_A
#undef NT8_RD_B
        if (g == 0) bar();                                    // pairs with G1's last barrier
        const int em = m0 + g * HG, en = n0 + wc * 64;
        __syncthreads();                                      // every wave is done with both buffers, nothing in flight
        nt_epilogue_stage_bias(a, smem + BUF_BYTES, wave, lane, em, en);
        const bool has_next = v + (int)gridDim.x < ntiles;
        // Beside a pending LDS-DMA hipcc waits vmcnt(0) for every ordinary load it uses (bias, fp32 residual rows, GELU'
        // operands) and - because a DMA is a pending LDS write it cannot disambiguate - in front of every group of
        // compiler-visible LDS reads, so every row group's stores would wait for the previous group's stores: a fully
        // serialised epilogue.  Hence: epilogues that load from global memory run with NO DMA in flight (the next tile's first
        // K-tile is requested after them); the others use asm LDS accessors and get it requested before (`early` below).
        // the epilogue's lane-derived constants are recomputed per tile from an opaque copy of the lane id: hoisted out of the
        // tile loop they would be spilled (the K loop owns the register file) and every reload is a vmcnt(0) in the epilogue
        int elane = lane;
        asm volatile("" : "+v"(elane));
        EpiPrefetch<MI> pf;
        nt_epilogue_prefetch<ACT, MI>(a, pf, elane, em, en);
        // Epilogues that read nothing from global memory (bf16 output without / with the GELU pair: bias comes from LDS, all LDS
        // traffic is inline asm) run with the next tile's first K-tile already in flight into buffer 0 - the patches live in
        // buffer 1 - so the 64 KiB land under the stores instead of in front of the next tile's first MFMA.  The other
        // epilogues keep the order "epilogue, then request": their residual / GELU' loads would each be waited for with
        // vmcnt(0) beside a pending DMA (see above).
        const bool early = has_next && ACT != 2 && !a.out_f32;
        int vnext = v + (int)gridDim.x;
        if (early) {
            set_tile(vnext);
            dma_a(0, 0); dma_a(0, 1); dma_b(0, 0); dma_b(0, 1);
        }
#if NT8_ABLATE >= 2
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) asm volatile("" ::"v"(acc[i][j]));
#if NT8_ABLATE == 4
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(acc16[i >> 2][(i >> 1) & 1][i & 1]));
#endif
#else
        if (FP8) {
            // the de-quantisation factor belongs to the product alone (the shared epilogue's alpha also scales bias and residual)
            const float dq = m0 >= a.m_split ? dq_b : dq_a;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) acc[i][j] *= dq;
            GemmNtArgs e = a;
            e.alpha = 1.0f;
            e.out8_scale = q8_scale;            // the epilogue takes the record's scale / amax filter from here, not from memory
            e.q8_seen = q8_seen;
            nt_epilogue<ACT, MI, FP8>(e, acc, pf, smem + BUF_BYTES, wave, elane, em, en, HG);
        } else {
            nt_epilogue<ACT, MI>(a, acc, pf, smem + BUF_BYTES, wave, elane, em, en, HG);
        }
#endif
        // a compiler-visible full drain: the K loop reuses registers the epilogue loaded into, and hipcc would otherwise
        // re-wait for those loads (vmcnt(0)) at the head of EVERY K-tile.  The next tile's first wait drains the stores anyway.
        wait_vm<0>();
        // (the eight DMA source pointers are only computed here - for the early case RE-computed from an opaque copy of the tile
        //  id: alive during the epilogue they push it over the register file, and a spill reload is one more vector-memory op)
        if (has_next) {
            asm volatile("" : "+s"(vnext));
            set_tile(vnext);
            if (!early) { dma_a(0, 0); dma_a(0, 1); dma_b(0, 0); dma_b(0, 1); }
        }
    };
    for (; v < ntiles; v += gridDim.x) {
        if (NBA != NBB && v < tb) run_tile(std::integral_constant<int, NBA>{});
        else run_tile(std::integral_constant<int, NBB>{});
    }
}

// ---------------------------------------------------------------------------------------------------
// Small problems (round 5): when the 128 x 128 tiling gives at most one workgroup per CU, gemm_nt_kernel<., 2, 4> is a chain of load
// latencies - one K-slab in flight, a full drain and a barrier per K-step, nothing else resident on the CU to hide them: a 2 048-row x 768
// GEMM of the reference's batch-4 step takes 26 us for 12 K-steps of 0.25 us of MFMA work each (profiles/r05/b4_kernel_stats.csv).
// This kernel keeps NS - 1 = 3 K-slabs in flight: a ring of NS slots of [A 128 x 64 | B 128 x 64] (32 KiB each), LDS-DMA with counted
// vmcnt, ONE raw barrier per K-step, fragment reads in inline asm (a compiler-visible ds_read behind a pending LDS-DMA drains the ring).
// One tile per workgroup (the grid is at most the CU count by construction), same LDS image, fragments, MFMA order and epilogue as
// gemm_nt_kernel<., 2, 4>: bitwise the same results.
// MI = 4: 128 x 128 tiles; MI = 2: 64 x 128 tiles (24 KiB slots) for problems that would otherwise fill less than half the CUs.
template <int ACT, int MI = 4>
__global__ __launch_bounds__(256) void gemm_nt_ring_kernel(GemmNtArgs a) {
    constexpr int NT = 256, NWN = 2, NS = 4;
    constexpr int TBM = 2 * MI * 16, TBN = 128;
    constexpr int A_BYTES = TBM * 128, B_BYTES = TBN * 128, SLOT = A_BYTES + B_BYTES;
    constexpr int CA = TBM * 8 / NT, CB = TBN * 8 / NT;      // 4 + 4 LDS-DMA instructions per thread and K-slab
    constexpr int DPT = CA + CB;
    extern __shared__ __attribute__((aligned(16))) char smem[];            // NS slots
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int nt_n = a.N / TBN;
    const int ntiles = ((a.M + TBM - 1) / TBM) * nt_n;
    const int v = blockIdx.x;
    if (v >= ntiles) return;
    const int fr = lane & 15, fq = lane >> 4;
    const int off_k0 = fr * 128 + (((0 + fq) ^ (lane & 7)) << 4);
    const int off_k1 = fr * 128 + (((4 + fq) ^ (lane & 7)) << 4);
    const int nk = a.K / BK;
    const int wg = xcd_remap(v, ntiles);
    const int m0 = (wg / nt_n) * TBM, n0 = (wg % nt_n) * TBN;
    const bf16_t* srcA[CA];
    const bf16_t* srcB[CB];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        const int p = i * NT + tid, row = p >> 3, c = (p & 7) ^ (row & 7);
        srcA[i] = a.A + (size_t)min(m0 + row, a.M - 1) * a.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const int p = i * NT + tid, row = p >> 3, c = (p & 7) ^ (row & 7);
        srcB[i] = (m0 >= a.m_split ? a.B2 : a.B) + (size_t)(n0 + row) * a.ldb + c * 8;
    }
    auto stage = [&](int slot, int k0) {
        char* sa = smem + slot * SLOT;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < CA; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcA[i] + k0), (LDS_AS void*)(sa + (i * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < CB; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcB[i] + k0), (LDS_AS void*)(sb + (i * NT + wave * 64) * 16), 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) stage(t, t * BK);

    f32x4 acc[4][MI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned lds_base = (unsigned)(size_t)(LDS_AS const char*)smem;
    const unsigned la = lds_base + wm * (MI * 16) * 128, lb = lds_base + A_BYTES + wn * 64 * 128;
    int slot = 0;
    for (int t = 0; t < nk; ++t) {
        // K-slab t has landed (this wave's share; the barrier covers the others'); the slabs requested behind it stay in flight
        const int ahead = min(NS - 2, nk - 1 - t);
        if (ahead == 2) wait_vm<2 * DPT>();
        else if (ahead == 1) wait_vm<DPT>();
        else wait_vm<0>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // ... and every wave is done reading slab t - 1, whose slot slab t + NS - 1 takes
        asm volatile("" ::: "memory");
        if (t + NS - 1 < nk) stage(slot == 0 ? NS - 1 : slot - 1, (t + NS - 1) * BK);
        const unsigned va0 = la + slot * SLOT + off_k0, va1 = la + slot * SLOT + off_k1;
        const unsigned vb0 = lb + slot * SLOT + off_k0, vb1 = lb + slot * SLOT + off_k1;
        slot = slot + 1 == NS ? 0 : slot + 1;
        bf16x8 wf[2][4], xf[2][MI];
#define NTR_RD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF) : "memory")
        NTR_RD(wf[0][0], vb0, 0); NTR_RD(wf[0][1], vb0, 2048); NTR_RD(wf[0][2], vb0, 4096); NTR_RD(wf[0][3], vb0, 6144);
        NTR_RD(xf[0][0], va0, 0); NTR_RD(xf[0][1], va0, 2048);
        if constexpr (MI == 4) { NTR_RD(xf[0][2], va0, 4096); NTR_RD(xf[0][3], va0, 6144); }
        NTR_RD(wf[1][0], vb1, 0); NTR_RD(wf[1][1], vb1, 2048); NTR_RD(wf[1][2], vb1, 4096); NTR_RD(wf[1][3], vb1, 6144);
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");       // the kk = 0 fragments (everything but the last four reads)
        __builtin_amdgcn_sched_barrier(0);
        NTR_RD(xf[1][0], va1, 0); NTR_RD(xf[1][1], va1, 2048);
        if constexpr (MI == 4) { NTR_RD(xf[1][2], va1, 4096); NTR_RD(xf[1][3], va1, 6144); }
#undef NTR_RD
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ni], xf[0][mi], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ni], xf[1][mi], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    const int em = m0 + wm * (MI * 16), en = n0 + wn * 64;
    __syncthreads();                                       // every wave is done with the ring; nothing is in flight (the last wait was vmcnt(0))
    nt_epilogue_stage_bias(a, smem, wave, lane, em, en);
    EpiPrefetch<MI> pf;
    nt_epilogue_prefetch<ACT, MI>(a, pf, lane, em, en);
    nt_epilogue<ACT, MI>(a, acc, pf, smem, wave, lane, em, en);
}

// ---------------------------------------------------------------------------------------------------
// wgrad.  Both operands are row-major with the CONTRACTION index (token row m) as the slow dimension, so the
// MFMA fragments (8 consecutive k per lane) are columns of the staged tiles: they are read with the gfx950
// transposing LDS read ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group, lane i receives column i).
// Tiles are [64 token rows][128 columns] (256-B rows); chunk swizzle c ^= (row&3)<<2 spreads the 4 rows of a
// transposed read over the four 64-B quarters of the bank row (conflict-free).  The contraction is split over
// gridDim.z; partial tiles are accumulated with fp32 atomics whose wave footprint is two 128-B row segments.
// Contract: A and B are allocated (and zero) up to the next multiple of 64 rows beyond M.
struct GemmTnArgs {
    const bf16_t* A; long long lda;
    const bf16_t* B; long long ldb;
    float* C; long long ldc;
    int M, N1, N2;
    int stages_per_split;
};

// Up to three weight gradients over the SAME token rows in one launch of the 8-phase kernel (avs_gemm_tn_bf16_group3: a block's fc2,
// fc1 and proj gradients, whose operands all exist when the attention backward starts).  Every workgroup adds one 256 x 256 fp32 tile
// with atomics, so a launch costs (workgroups x 256 KiB) of atomic traffic whatever its shape - 64 MB when a single 9 - 36-tile
// gradient is split 7 - 28 ways to fill the chip.  Together the three have 81 tiles (ViT-B), need 3 splits instead of 7 + 7 + 14, and the
// atomic bytes of the three fall from ~158 MB to ~62 MB.  Tiles [tile0[i], tile0[i + 1]) of the launch belong to problem i.
struct TnProb {
    const bf16_t* A; long long lda;
    const bf16_t* B; long long ldb;
    float* C; long long ldc;
    int N1, N2, tile0;
};
struct GemmTnGroupArgs {
    TnProb p[3];
    int nprob, tiles, M, stages_per_split;
};

// The transposing reads are issued through inline asm: hipcc (ROCm 7.2) cannot disambiguate the ds_read_tr builtin from
// the LDS-DMA loads still in flight for the NEXT slab and puts an s_waitcnt vmcnt(0) in front of the first read of every
// stage - which serialises load and compute completely (measured: matrix pipe 21 % busy).  An asm statement is opaque to
// that logic; its completion is waited for by hand (tr_wait) with a sched_barrier so no MFMA is hoisted above the wait.
struct TrFrag { bf16x4 lo, hi; };

template <int ROWB>
__device__ __forceinline__ void lds_tr_issue(TrFrag& f, const char* base) {
    // rows r..r+3 then r+4..r+7 of the same 16 columns -> 8 consecutive k of one column per lane
    const unsigned addr = (unsigned)(size_t)(LDS_AS const char*)base;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                 : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr), "i"(4 * ROWB) : "memory");
}

__device__ __forceinline__ void tr_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int KEEP>                                          // wait until at most KEEP of this wave's LDS reads are outstanding
__device__ __forceinline__ void tr_wait_keep() {
    static_assert(KEEP <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(KEEP) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ bf16x8 tr_join(const TrFrag& f) {
    return bf16x8{f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
}

// <NWC=2, MI=2>: 128x128 output tile, 256 threads, 64 KiB LDS;  <NWC=4, MI=4>: 256x256 tile, 512 threads, 128 KiB LDS.
template <int NWC, int MI>
__global__ __launch_bounds__(128 * NWC) void gemm_tn_kernel(GemmTnArgs a) {
    constexpr int NT = 128 * NWC;
    constexpr int T1 = 2 * MI * 32, T2 = NWC * 64;          // output tile: T1 (columns of A) x T2 (columns of B)
    constexpr int RA = T1 * 2, RB = T2 * 2;                 // LDS row bytes of the staged [64 rows][T] tiles
    constexpr int A_BYTES = 64 * RA, B_BYTES = 64 * RB, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int CA = 64 * (T1 / 8) / NT, CB = 64 * (T2 / 8) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / NWC, wc = wave % NWC;
    // 1-D grid, XCD-aware: logical ids are dealt so that each XCD gets a contiguous run = all output tiles of (about) one
    // split of the contraction.  Those workgroups stream the SAME token rows of A and B at the same time, so each row
    // slab is fetched from HBM once per XCD instead of once per tile (3-12x less HBM traffic on these shapes).
    const int tiles2 = a.N2 / T2, tiles = (a.N1 / T1) * tiles2;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lid / tiles, tile = lid % tiles;
    const int n2_0 = (tile % tiles2) * T2, n1_0 = (tile / tiles2) * T1;
    const int nstages = (a.M + BK - 1) / BK;
    const int s_begin = split * a.stages_per_split;
    const int s_end = min(nstages, s_begin + a.stages_per_split);
    if (s_begin >= s_end) return;

    const bf16_t* srcA[CA];
    const bf16_t* srcB[CB];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        const int p = i * NT + tid, row = p / (T1 / 8), c = (p % (T1 / 8)) ^ ((row & 3) << 2);
        srcA[i] = a.A + (size_t)row * a.lda + n1_0 + c * 8;
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const int p = i * NT + tid, row = p / (T2 / 8), c = (p % (T2 / 8)) ^ ((row & 3) << 2);
        srcB[i] = a.B + (size_t)row * a.ldb + n2_0 + c * 8;
    }
    auto stage = [&](int buf, int s) {
        char* sa = smem + buf * BUF_BYTES;
        char* sb = sa + A_BYTES;
        const size_t ra = (size_t)s * BK * a.lda, rb = (size_t)s * BK * a.ldb;
#pragma unroll
        for (int i = 0; i < CA; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcA[i] + ra), (LDS_AS void*)(sa + (i * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < CB; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcB[i] + rb), (LDS_AS void*)(sb + (i * NT + wave * 64) * 16), 16, 0, 0);
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read address of this lane inside a tile, for k16-step 0 and column block 0:
    //   group G = lane>>4: h = G>>1 (k half), cb = G&1 (16-column half); lane i = lane&15 supplies row q = i>>2,
    //   columns 4*(i&3).. ; row = 8h + q ; col = 16*cb + 4*(i&3)
    const int h = lane >> 5, cb = (lane >> 4) & 1, li = lane & 15, q = li >> 2, p4 = li & 3;
    const int trow = 8 * h + q;
    const int tcol = 16 * cb + 4 * p4;                      // + column block base (multiple of 32)
    auto tr_off = [&](int colbase, int ks, int rowb) {
        const int col = colbase + tcol;
        const int chunk = (col >> 3) ^ (q << 2);            // row&3 == q for every row this lane addresses
        return (ks * 16 + trow) * rowb + (chunk << 4) + (col & 7) * 2;
    };

    stage(0, s_begin);
    for (int s = s_begin; s < s_end; ++s) {
        const int t = s - s_begin;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of slab s has landed (explicit: see gemm_nt_kernel)
        __syncthreads();
        if (s + 1 < s_end) stage((t + 1) & 1, s + 1);
        const char* sa = smem + (t & 1) * BUF_BYTES;
        const char* sb = sa + A_BYTES;
        // fragments are double-buffered by hand: the reads of k16-step ks+1 are in flight while the MFMAs of step ks run
        TrFrag af[2][MI], bfr[2][2];
        auto issue = [&](int ks, TrFrag (&fa)[MI], TrFrag (&fb)[2]) {
#pragma unroll
            for (int i = 0; i < MI; ++i) lds_tr_issue<RA>(fa[i], sa + tr_off(wr * (MI * 32) + i * 32, ks, RA));
#pragma unroll
            for (int i = 0; i < 2; ++i) lds_tr_issue<RB>(fb[i], sb + tr_off(wc * 64 + i * 32, ks, RB));
        };
        issue(0, af[0], bfr[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
                issue(ks + 1, af[(ks + 1) & 1], bfr[(ks + 1) & 1]);
                tr_wait_keep<2 * (MI + 2)>();                  // the (MI+2) x 2 reads just issued may stay in flight
            } else {
                tr_wait_keep<0>();
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(af[ks & 1][i]), tr_join(bfr[ks & 1][j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }

#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n2 = n2_0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n1 = n1_0 + wr * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                atomicAdd(a.C + (size_t)n1 * a.ldc + n2, acc[i][j][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// 8-phase variant of the 256x256 wgrad tile (same idea as gemm_nt8_kernel).  A stage (64 token rows of A and B) is
// multiplied in four phases of 16 rows = 8 MFMAs (32x32x16) per wave; the two wave groups (wr = 0 / 1) are offset by one
// barrier, so a group's 12 transposing fragment reads and 2 LDS-DMAs run under the other group's MFMA segment, and a
// single fragment set suffices.  The 8-KiB (A) + 8-KiB (B) granule a phase consumes is re-filled two phases later with
// the rows of the stage two ahead: granules are issued in reading order, six phases before they are read, and the wait
// in front of every barrier is the constant vmcnt(10) (five younger granules stay in flight).
template <int DUMMY>
__global__ __launch_bounds__(512) void gemm_tn8_kernel(GemmTnGroupArgs g) {
    constexpr int T1 = 256, T2 = 256, MI = 4;
    constexpr int RA = T1 * 2, RB = T2 * 2;                 // LDS row bytes of the staged [64 rows][256] tiles
    constexpr int A_BYTES = 64 * RA, BUF_BYTES = 2 * A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lid / g.tiles;
    int tile = lid % g.tiles;
    // which of the launch's (up to three) gradients this tile belongs to: block-uniform, scalar
    const int pi = (g.nprob > 2 && tile >= g.p[2].tile0) ? 2 : (g.nprob > 1 && tile >= g.p[1].tile0) ? 1 : 0;
    struct { const bf16_t* A; long long lda; const bf16_t* B; long long ldb; float* C; long long ldc; int M; } a
        = {g.p[pi].A, g.p[pi].lda, g.p[pi].B, g.p[pi].ldb, g.p[pi].C, g.p[pi].ldc, g.M};
    tile -= g.p[pi].tile0;
    const int tiles2 = g.p[pi].N2 / T2;
    const int n2_0 = (tile % tiles2) * T2, n1_0 = (tile / tiles2) * T1;
    const int nstages = (a.M + BK - 1) / BK;
    const int s_begin = split * g.stages_per_split;
    const int nst = min(nstages, s_begin + g.stages_per_split) - s_begin;      // stages of this workgroup
    if (nst <= 0) return;

    // granule (stage t, quarter q) = token rows s*64 + q*16 .. +15: 16 rows x 32 chunks = one 16-B chunk per thread
    const int grow = tid >> 5, gc = (tid & 31) ^ ((grow & 3) << 2);
    const bf16_t* srcA = a.A + (size_t)(s_begin * BK + grow) * a.lda + n1_0 + gc * 8;
    const bf16_t* srcB = a.B + (size_t)(s_begin * BK + grow) * a.ldb + n2_0 + gc * 8;
    auto dma = [&](int t, int q) {
        char* dst = smem + (t & 1) * BUF_BYTES + (q * 16) * RA + wave * 1024;
        const size_t r = (size_t)(t * BK + q * 16);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcA + r * a.lda), (LDS_AS void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcB + r * a.ldb), (LDS_AS void*)(dst + A_BYTES), 16, 0, 0);
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int h = lane >> 5, cb = (lane >> 4) & 1, li = lane & 15, q4 = li >> 2, p4 = li & 3;
    const int trow = 8 * h + q4;
    const int tcol = 16 * cb + 4 * p4;
    auto tr_off = [&](int colbase) {                          // offset inside a granule (rows 0..15)
        const int col = colbase + tcol;
        const int chunk = (col >> 3) ^ (q4 << 2);
        return trow * RA + (chunk << 4) + (col & 7) * 2;
    };
    int offA[MI], offB[2];
#pragma unroll
    for (int i = 0; i < MI; ++i) offA[i] = tr_off(wr * (MI * 32) + i * 32);
#pragma unroll
    for (int i = 0; i < 2; ++i) offB[i] = A_BYTES + tr_off(wc * 64 + i * 32);

    // prologue: granules are issued in reading order; six run ahead of the reader
    dma(0, 0); dma(0, 1); dma(0, 2); dma(0, 3);
    if (nst > 1) { dma(1, 0); dma(1, 1); wait_vm<10>(); } else wait_vm<0>();
    bar();
    if (wr == 1) bar();                                       // the one-barrier offset between the groups
    TrFrag af[MI], bfr[2];
    for (int t = 0; t < nst; ++t) {
        const bool full = t + 2 < nst;                        // every granule this stage wants to issue exists
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const char* gr = smem + (t & 1) * BUF_BYTES + (q * 16) * RA;
#pragma unroll
            for (int i = 0; i < MI; ++i) lds_tr_issue<RA>(af[i], gr + offA[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) lds_tr_issue<RB>(bfr[i], gr + offB[i]);
            // re-fill the granule read two phases ago with the rows of the stage after next
            const int ts = q < 2 ? t + 1 : t + 2, qs = (q + 2) & 3;
            if (ts < nst) dma(ts, qs);
            // the granule of the NEXT phase (issued six phases ago) must have landed; five younger ones stay in flight
            if (full) wait_vm<10>(); else wait_vm<0>();
            bar();
            tr_wait();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(af[i]), tr_join(bfr[j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            bar();
        }
    }
    if (wr == 0) bar();                                       // pairs with the other group's last barrier

#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n2 = n2_0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n1 = n1_0 + wr * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                atomicAdd(a.C + (size_t)n1 * a.ldc + n2, acc[i][j][r]);
            }
        }
}

// ===================================================================================================
// Tuning knobs live in api.cpp (common.h AvsTuning; set through avs_tuning_set / the wrappers below, never read from the environment here).
#define g_force_tile (avs_tuning().gemm_tile)
#define g_persistent (avs_tuning().gemm_persistent)
#define g_nt8 (avs_tuning().gemm_nt8)
#define g_force_h (avs_tuning().nt_tile_h)

extern "C" int avs_gemm_set_persistent(int on) { return avs_tuning_set("gemm_persistent", on ? 1 : 0); }

static long long g_nt_dispatches = 0;      // kernel dispatches issued by avs_gemm_nt_bf16 so far (a call is one or two): a counter, not a knob
extern "C" long long avs_gemm_nt_dispatches(void) { return g_nt_dispatches; }

extern "C" int avs_gemm_set_nt8(int on) { return avs_tuning_set("gemm_nt8", on ? 1 : 0); }
extern "C" int avs_gemm_set_tile_height(int h) { return avs_tuning_set("nt_tile_h", h); }
extern "C" int avs_gemm_set_tile(int tile) { return avs_tuning_set("gemm_tile", tile); }

// Host-side choice of the kernel family of a forward / input-gradient GEMM (knobs gemm_tile, nt_big_min, gemm_ring, cu_reserve included; no device
// work).  gemm_nt_launch takes exactly these decisions; avs_gemm_nt_plan exports them so that the dispatch thresholds are pinned by the CPU tests.
enum { NT_TWOBUF = 0, NT_RING = 1, NT_RING_HALF = 2, NT_TWOBUF_HALF = 3, NT_PERSISTENT = 4 };
static bool nt_use_big(int M, int N) {
    const int force = g_force_tile;
    const int big_tiles = ceil_div(M, 256) * (N / 256);
    const int big_min = avs_tuning().nt_big_min > 0 ? avs_tuning().nt_big_min : avs_persistent_slots() / 2;
    return force == 256 ? (N % 256) == 0 : force == 128 ? false : ((N % 256) == 0 && big_tiles >= big_min);
}
// the 128 x 128 side: `whole` = every row in full-height tiles (not the remainder rows of a persistent launch)
static int nt_small_family(int M, int N, int K, bool whole) {
    const int nwg = ceil_div(M, BM) * (N / BN);
    if (avs_tuning().gemm_ring == 2 && 2 * nwg <= avs_persistent_slots() && whole) return K >= 256 ? NT_RING_HALF : NT_TWOBUF_HALF;
    if (avs_tuning().gemm_ring && nwg <= avs_persistent_slots() && K >= 256 && whole) return NT_RING;
    return NT_TWOBUF;
}

extern "C" int avs_gemm_nt_plan(int M, int N, int K, int* family, int* workgroups) {
    AVS_CHECK_ARG(M > 0 && N > 0 && K > 0 && (N % BN) == 0 && (K % BK) == 0 && family && workgroups, "gemm_nt_plan: bad arguments M=%d N=%d K=%d", M, N, K);
    if (nt_use_big(M, N)) {
        const int tiles = ceil_div(M, 256) * (N / 256), slots = avs_persistent_slots();
        *family = NT_PERSISTENT;
        *workgroups = tiles < slots ? tiles : slots;          // (with two tile heights the 8-phase kernel may walk a few more, cheaper tiles)
        return 0;
    }
    const int fam = nt_small_family(M, N, K, true);
    *family = fam;
    *workgroups = ceil_div(M, (fam == NT_RING_HALF || fam == NT_TWOBUF_HALF) ? BM / 2 : BM) * (N / BN);
    return 0;
}

static int gemm_nt_launch(const bf16_t* A, long long lda, const bf16_t* B, long long ldb, int M, int N, int K,
                          const float* bias, const float* res, long long ldr, const int* res_idx, const bf16_t* aux,
                          long long ldaux, void* out, long long ldo, int out_f32, bf16_t* out2, long long ldo2,
                          float alpha, int act, int scale_cols, float col_scale, float* colsum,
                          int m_split, const bf16_t* B2, const float* bias2, float* colsum2, hipStream_t stream) {
    AVS_CHECK_ARG(M > 0 && N > 0 && K > 0 && (N % BN) == 0 && (K % BK) == 0, "gemm_nt: need N%%128==0, K%%64==0 (M=%d N=%d K=%d)", M, N, K);
    AVS_CHECK_ARG(A && B && out, "gemm_nt: null operand");
    // gelu'(x) as 8-bit fixed-point codes (GP8_*; round 6: the bf16 GEMMs too - EngineOptions.gelu8): out_f32 == 2 with act 1 - `out` receives one byte per
    // value (ldo in BYTES); act == 3 = act 2 whose `aux` holds those codes (ldaux in bytes).  Same convention as avs_gemm_nt_fp8's out_f32 == 2 / a_e5m2 == 2.
    int aux8 = 0;
    if (out_f32 == 2) {
        AVS_CHECK_ARG(act == 1 && (ldo % 8) == 0 && ldo >= N, "gemm_nt: out_f32 == 2 (8-bit gelu') goes with act 1");
        out_f32 = 0; aux8 = 1;
    }
    if (act == 3) {
        AVS_CHECK_ARG(aux && ldaux >= N, "gemm_nt: act 3 (act 2 with 8-bit gelu' codes) needs aux");
        act = 2; aux8 = 1;
    }
    AVS_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (ldo % (out_f32 ? 4 : 8)) == 0 && (!out2 || (ldo2 % 8) == 0) && (!aux || (ldaux % 8) == 0),
                  "gemm_nt: leading dimensions must keep 16-byte alignment");
    AVS_CHECK_ARG(act >= 0 && act <= 2 && (act != 1 || out2) && (act != 2 || aux), "gemm_nt: bad activation arguments");
    AVS_CHECK_ARG(!(out_f32 && act != 0), "gemm_nt: fp32 output is supported with act 0 only");
    AVS_CHECK_ARG(!(out_f32 && colsum), "gemm_nt: the fused column sum is implemented for bf16 output");
    AVS_CHECK_ARG(!res || out_f32, "gemm_nt: the residual add is implemented for fp32 output");
    AVS_CHECK_ARG(scale_cols >= 0 && scale_cols <= N && (scale_cols % 64) == 0, "gemm_nt: scale_cols must be a multiple of 64 within N");
    GemmNtArgs a{A, lda, B, ldb, M, N, K, bias, res, ldr, res_idx, aux, ldaux, out, ldo, out_f32, out2, ldo2, alpha, act, scale_cols, col_scale, colsum, M,
                 m_split, B2, bias2, colsum2, nullptr, 0, 1.0f, nullptr, nullptr, nullptr, nullptr, 0.f, 0, aux8};
    // 256^2 tiles once they give half the CUs a workgroup; otherwise 128^2 (4x the workgroups, two per CU).  Rounds 1 - 4 asked for 224 tiles; between
    // 128 and 224 the 128^2 tiling needs 512 ... 896 workgroups = a second round on 512 slots, and the persistent kernel on a part of the chip
    // runs at a higher clock: 135 tiles, K = 3072: 78.1 -> 63.8 us, K = 768: 25.8 -> 23.5; 192 tiles: 80.9 -> 68.9, 28.4 -> 25.7; under
    // 128 tiles the 128^2 tiling fits one round and wins (96 tiles: 48.2 against 62.1 us) - tools/bench_nt_midsize.py, profiles/r05/nt_midsize.log
    const int big_tiles = ceil_div(M, 256) * (N / 256);
    const bool big = nt_use_big(M, N);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipSuccess;
        const void* big_k[3] = {(const void*)gemm_nt_kernel<0, 4, 8>, (const void*)gemm_nt_kernel<1, 4, 8>, (const void*)gemm_nt_kernel<2, 4, 8>};
        const void* k8[6] = {(const void*)gemm_nt8_kernel<0>, (const void*)gemm_nt8_kernel<1>, (const void*)gemm_nt8_kernel<2>,
                             (const void*)gemm_nt8_kernel<0, 0, 4, 3>, (const void*)gemm_nt8_kernel<1, 0, 4, 3>, (const void*)gemm_nt8_kernel<2, 0, 4, 3>};
        for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipFuncSetAttribute(k8[i], hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        const void* small_k[3] = {(const void*)gemm_nt_kernel<0, 2, 4>, (const void*)gemm_nt_kernel<1, 2, 4>, (const void*)gemm_nt_kernel<2, 2, 4>};
        const void* ring_k[3] = {(const void*)gemm_nt_ring_kernel<0>, (const void*)gemm_nt_ring_kernel<1>, (const void*)gemm_nt_ring_kernel<2>};
        const void* ring_h[3] = {(const void*)gemm_nt_ring_kernel<0, 2>, (const void*)gemm_nt_ring_kernel<1, 2>, (const void*)gemm_nt_ring_kernel<2, 2>};
        for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipFuncSetAttribute(ring_h[i], hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
        for (int i = 0; i < 3 && e == hipSuccess; ++i) {
            e = hipFuncSetAttribute(big_k[i], hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            if (e == hipSuccess) e = hipFuncSetAttribute(small_k[i], hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            if (e == hipSuccess) e = hipFuncSetAttribute(ring_k[i], hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        }
        if (e != hipSuccess) {
            avs_set_error("gemm_nt: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return -1;
        }
        attr_done = true;
    }
    auto launch_small = [&](const GemmNtArgs& x) {
        const int nwg = ceil_div(x.M, BM) * (x.N / BN);
        const int fam = nt_small_family(x.M, x.N, x.K, x.m_full >= x.M);
        // fewer workgroups than half the CUs: every row as HALF-height (64 x 128) tiles of the same kernel - twice the workgroups on CUs
        // that would idle (a CU fills its LDS at ~60 GB/s: the 32 KiB K-slab of a 128 x 128 tile takes 0.55 us against 0.25 us of MFMA
        // work, so small GEMMs are bound by how many CUs pull, profiles/r05/small_gemm_ab.log); bitwise the full tiles' results
        if (fam == NT_RING_HALF || fam == NT_TWOBUF_HALF) {
            const int nh = ceil_div(x.M, BM / 2) * (x.N / BN);
            if (fam == NT_RING_HALF) {         // ... of the ring kernel (three K-slabs in flight)
                if (act == 0) gemm_nt_ring_kernel<0, 2><<<nh, 256, 98304, stream>>>(x);
                else if (act == 1) gemm_nt_ring_kernel<1, 2><<<nh, 256, 98304, stream>>>(x);
                else gemm_nt_ring_kernel<2, 2><<<nh, 256, 98304, stream>>>(x);
                return;
            }
            GemmNtArgs h = x;
            h.m_full = 0;
            if (act == 0) gemm_nt_kernel<0, 2, 4><<<nh, 256, 65536, stream>>>(h);
            else if (act == 1) gemm_nt_kernel<1, 2, 4><<<nh, 256, 65536, stream>>>(h);
            else gemm_nt_kernel<2, 2, 4><<<nh, 256, 65536, stream>>>(h);
            return;
        }
        // at most one workgroup per CU: the ring kernel keeps three K-slabs in flight instead of one
        if (fam == NT_RING) {
            if (act == 0) gemm_nt_ring_kernel<0><<<nwg, 256, 131072, stream>>>(x);
            else if (act == 1) gemm_nt_ring_kernel<1><<<nwg, 256, 131072, stream>>>(x);
            else gemm_nt_ring_kernel<2><<<nwg, 256, 131072, stream>>>(x);
            return;
        }
        if (act == 0) gemm_nt_kernel<0, 2, 4><<<nwg, 256, 65536, stream>>>(x);
        else if (act == 1) gemm_nt_kernel<1, 2, 4><<<nwg, 256, 65536, stream>>>(x);
        else gemm_nt_kernel<2, 2, 4><<<nwg, 256, 65536, stream>>>(x);
    };
    if (big) {
        // persistent: one 128-KiB-LDS workgroup per CU walks the tiles (grid = min(tiles, CUs))
        const int ncu = avs_persistent_slots();      // CUs a persistent grid may fill (device CUs - the cu_reserve knob)
        // Two-buffer kernel (8-phase kernels switched off): with T tiles on C CUs the last of ceil(T/C) rounds may be nearly
        // empty (1122 tiles -> 4.4 rounds cost 5).  When the last round would be less than half full, the row panels that fill
        // floor(T/C) rounds stay 256^2 tiles and the remaining rows become 128 x 256 tiles of the SAME launch.
        const int nt_n = N / 256, nt_m = ceil_div(M, 256);
        GemmNtArgs b = a;
        if (g_persistent && g_force_tile == 0 && big_tiles > ncu) {
            const int rows_full = ((big_tiles / ncu) * ncu) / nt_n;              // row panels inside whole rounds
            if (rows_full >= 1 && rows_full < nt_m && (big_tiles % ncu) * 2 <= ncu) b.m_full = rows_full * 256;
        }
        if (g_nt8 >= 1 && K >= 128 && g_persistent && g_force_tile == 0) {
            // 8-phase kernel, one dispatch: a partial last round costs it the same as handing the leftover rows to the half-height-tile
            // kernel in a second dispatch (measured: 185.7 vs 186.0 ms/step).  What it can do about a badly filled last round: the R rounds
            // that 256-row tiles need offer R x CUs tile slots; with some of the row tiles 224 rows high (gemm_nt8_kernel's second height
            // class) the same rows fill MORE of those slots with CHEAPER tiles (a 224-row tile costs ~0.90 of a 256-row one,
            // profiles/r03/gemm_tile_height_by_shape.log) - e.g. 1122 tiles = 4.4 rounds of N = 768 become 21 + 1257 tiles in 5 full rounds:
            // 4.6 tile-times per CU instead of 5.  Two weight sets: their row split is a multiple of 256, so the 256-row class must cover
            // the first set entirely (every 224-row tile then lies in the second).
            const bool one_set = m_split >= M || m_split <= 0;
            const int unit = nt_n * 8 / std::gcd(nt_n, 8);      // the first class holds whole row tiles and a multiple of 8 tiles (XCD classes)
            int tb = -1;                                          // -1: every tile 256 rows (the one-class kernel)
            if (one_set && g_force_h == 224) tb = 0;              // every tile 224 rows
            else if (one_set && g_force_h == 240) tb = ((nt_m / 2) * nt_n / unit) * unit;
            else if (g_force_h == 0) {
                const int rounds = ceil_div(nt_m * nt_n, ncu);
                const int nrt = (rounds * ncu) / nt_n;            // row tiles the slots of those rounds hold
                // a row tiles of 256 + b of 224 must cover M: b <= (nrt * 256 - M) / 32
                int b = (int)(((long long)nrt * 256 - M) / 32);
                if (b > nrt) b = nrt;
                if (b > 0) {
                    int ta = ((nrt - b) * nt_n + unit - 1) / unit * unit;       // first-class tiles, rounded UP (fewer small tiles: still covers M)
                    if (!one_set && ta / nt_n * 256 < m_split) ta = ceil_div(ceil_div(m_split, 256) * nt_n, unit) * unit;
                    if (ta / nt_n <= nrt) {
                        const int na = ta / nt_n, nb = ceil_div(M - (na * 256 < M ? na * 256 : M), 224);
                        // tile-times of the busiest CU: every workgroup meets its first-class tiles first, then the others
                        const int nbig = ceil_div(ta, ncu), nall = ceil_div((na + nb) * nt_n, ncu);
                        const double cost = nbig + 0.90 * (nall - nbig);
                        if (nb > 0 && nall <= rounds && cost < 0.98 * rounds) tb = ta;
                    }
                }
            }
            a.tb = tb < 0 ? 0 : tb;
            const int tiles8 = tb < 0 ? nt_m * nt_n : tb + ceil_div(M - (tb / nt_n) * 256 > 0 ? M - (tb / nt_n) * 256 : 0, 224) * nt_n;
            int grid8 = tiles8 < ncu ? tiles8 : ncu;
            // AVSIAM_NT_GRID: cap the persistent grid (tools/bench_stagger.py: two half-chip GEMMs side by side on two streams)
            { const int gcap = avs_tuning().nt_grid; if (gcap > 0 && gcap < grid8) grid8 = gcap; }
#define NT8_LAUNCH(ACT_)                                                                          \
    do {                                                                                          \
        if (tb >= 0) gemm_nt8_kernel<ACT_, 0, 4, 3><<<grid8, 512, 131072, stream>>>(a);           \
        else gemm_nt8_kernel<ACT_, 0, 4, 4><<<grid8, 512, 131072, stream>>>(a);                   \
    } while (0)
            if (act == 0) NT8_LAUNCH(0);
            else if (act == 1) NT8_LAUNCH(1);
            else NT8_LAUNCH(2);
#undef NT8_LAUNCH
            AVS_LAUNCH_CHECK("gemm_nt8");
            ++g_nt_dispatches;
            return 0;
        }
        const int tiles_b = ceil_div(b.m_full, 256) * nt_n + (b.m_full < M ? ceil_div(M - b.m_full, 128) * nt_n : 0);
        const int grid = g_persistent ? (tiles_b < ncu ? tiles_b : ncu) : tiles_b;
        if (act == 0) gemm_nt_kernel<0, 4, 8><<<grid, 512, 131072, stream>>>(b);
        else if (act == 1) gemm_nt_kernel<1, 4, 8><<<grid, 512, 131072, stream>>>(b);
        else gemm_nt_kernel<2, 4, 8><<<grid, 512, 131072, stream>>>(b);
    } else {
        launch_small(a);
    }
    AVS_LAUNCH_CHECK("gemm_nt");
    ++g_nt_dispatches;
    return 0;
}

extern "C" int avs_gemm_nt_bf16(const bf16_t* A, long long lda, const bf16_t* B, long long ldb, int M, int N, int K,
                                const float* bias, const float* res, long long ldr, const int* res_idx, const bf16_t* aux,
                                long long ldaux, void* out, long long ldo, int out_f32, bf16_t* out2, long long ldo2,
                                float alpha, int act, int scale_cols, float col_scale, float* colsum, hipStream_t stream) {
    return gemm_nt_launch(A, lda, B, ldb, M, N, K, bias, res, ldr, res_idx, aux, ldaux, out, ldo, out_f32, out2, ldo2, alpha, act, scale_cols,
                          col_scale, colsum, 0x7fffffff, nullptr, nullptr, nullptr, stream);
}

// fp8 (OCP e4m3) operands, fp32 accumulation, the epilogues of the bf16 GEMM for act 0 (alpha, bias, fp32 residual, column-range scale,
// bf16 or fp32 output) and act 1 (GELU pair: out = gelu'(x), out2 = gelu(x), both bf16).
// x = alpha * (A8 . B8^T) + bias (+ res); alpha carries the product of the two de-quantisation scales - or, with the device records
// qa / qw (delayed scaling), qa[1] * qw[1] read by the kernel.  m_split / B2 / bias2 / qw2: a second weight set for rows from m_split
// (the MAE pass's two towers in one launch, as avs_gemm_nt_bf16_dual); m_split <= 0 or >= M: one set.
extern "C" int avs_gemm_nt_fp8(const uint8_t* A, long long lda, const uint8_t* B, long long ldb, int M, int N, int K, const float* bias,
                               const float* res, long long ldr, void* out, long long ldo, int out_f32, bf16_t* out2, long long ldo2, float alpha,
                               int act, int scale_cols, float col_scale, uint8_t* out8, long long ldo8, float out8_scale,
                               const float* qa, const float* qw, float* q8, int m_split, const uint8_t* B2, const float* bias2, const float* qw2,
                               int a_e5m2, const bf16_t* aux, long long ldaux, float* colsum, float* colsum2, hipStream_t stream) {
    // 8-bit gelu' (fp8 backward): out_f32 == 2 with act 1 - `out` receives the codes (ldo in bytes); a_e5m2 == 2 with act 2 - `aux` holds them (ldaux in bytes)
    const bool gp8 = (out_f32 == 2) || (a_e5m2 == 2);
    AVS_CHECK_ARG(out_f32 != 2 || (act == 1 && !a_e5m2 && out && (ldo % 8) == 0 && ldo >= N), "gemm_nt_fp8: out_f32 == 2 (8-bit gelu') goes with act 1 of the forward form");
    AVS_CHECK_ARG(a_e5m2 != 2 || (act == 2 && aux && ldaux >= N), "gemm_nt_fp8: a_e5m2 == 2 (8-bit gelu' operand) goes with act 2");
    if (out_f32 == 2) out_f32 = 0;
    if (a_e5m2 == 2) a_e5m2 = 1;
    // a_e5m2 != 0: the INPUT-GRADIENT form - A holds e5m2 gradients (B stays e4m3: the transposed weight copy); act 0, or act 2 with aux =
    // the saved gelu'(x) (bf16) and colsum (fc1 bias gradient) as in avs_gemm_nt_bf16; out8 then receives e5m2(out) for the next such GEMM
    AVS_CHECK_ARG(!out8 || ((ldo8 % 8) == 0 && ldo8 >= N && (a_e5m2 ? act != 1 : act == 1)),
                  "gemm_nt_fp8: out8 = e4m3(gelu(x)) goes with act 1, out8 = e5m2(out) with the input-gradient form");
    AVS_CHECK_ARG(M > 0 && N > 0 && K >= 256 && (N % 256) == 0 && (K % 128) == 0, "gemm_nt_fp8: need N%%256==0, K%%128==0, K>=256 (M=%d N=%d K=%d)", M, N, K);
    // 8-bit-only outputs: `out` may be NULL in the input-gradient form when out8 is given (the bf16 gradient has no reader left: the next
    // input-gradient GEMM and the weight gradient take the e5m2 copy, the bias gradient is the fused colsum); `out2` may be NULL with
    // act 1 when out8 is given (gelu(x) is read by fc2 and its weight gradient only, both in e4m3)
    AVS_CHECK_ARG(A && B && (out || (out8 && a_e5m2 && !out_f32)) && (lda % 16) == 0 && (ldb % 16) == 0 && lda >= K && ldb >= K && (!out || (ldo % (out_f32 ? 4 : 8)) == 0),
                  "gemm_nt_fp8: operands must keep 16-byte alignment (out may be NULL only beside an e5m2 out8)");
    AVS_CHECK_ARG(!res || out_f32, "gemm_nt_fp8: the residual add is implemented for fp32 output");
    AVS_CHECK_ARG((act == 0 && !out2) || (act == 1 && (out2 || out8) && !out_f32 && (!out2 || (ldo2 % 8) == 0) && !a_e5m2) ||
                  (act == 2 && a_e5m2 && aux && !out2 && !out_f32 && (ldaux % 8) == 0),
                  "gemm_nt_fp8: act 0; act 1 with a bf16 output pair (forward; out2 may be NULL beside out8); act 2 with aux (input-gradient form)");
    AVS_CHECK_ARG(!colsum || (!out_f32 && a_e5m2), "gemm_nt_fp8: the fused column sum goes with the bf16 output of the input-gradient form");
    AVS_CHECK_ARG(scale_cols >= 0 && scale_cols <= N && (scale_cols % 64) == 0, "gemm_nt_fp8: scale_cols must be a multiple of 64 within N");
    AVS_CHECK_ARG((qa == nullptr) == (qw == nullptr), "gemm_nt_fp8: qa and qw go together");
    const bool dual = m_split > 0 && m_split < M;
    AVS_CHECK_ARG(!dual || ((m_split % 256) == 0 && B2 && (bias == nullptr) == (bias2 == nullptr) && qa && qw2 && (colsum == nullptr) == (colsum2 == nullptr)),
                  "gemm_nt_fp8: two weight sets need m_split %% 256 == 0, B2, bias2 / colsum2 mirroring bias / colsum, and device records (qa, qw, qw2)");
    static bool attr_done = false;
    if (!attr_done) {
        const void* ks[4] = {(const void*)gemm_nt8_kernel<0, 1>, (const void*)gemm_nt8_kernel<1, 1>, (const void*)gemm_nt8_kernel<0, 2>, (const void*)gemm_nt8_kernel<2, 2>};
        for (int i = 0; i < 4; ++i)
            if (hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) {
                avs_set_error("gemm_nt_fp8: hipFuncSetAttribute failed");
                return -1;
            }
        attr_done = true;
    }
    // the fp8 matrices as the bf16 matrices they alias (see gemm_nt8_kernel): half the columns, half the leading dimension
    GemmNtArgs a{reinterpret_cast<const bf16_t*>(A), lda / 2, reinterpret_cast<const bf16_t*>(B), ldb / 2, M, N, K / 2, bias, res, ldr, nullptr, aux, ldaux,
                 out, ldo, out_f32, out2, ldo2, alpha, act, scale_cols, col_scale, colsum, M, dual ? m_split : 0x7fffffff,
                 dual ? reinterpret_cast<const bf16_t*>(B2) : nullptr, dual ? bias2 : nullptr, dual ? colsum2 : nullptr,
                 out8, ldo8, out8_scale, qa, qw, dual ? qw2 : qw, q8, 0.f};
    a.aux8 = gp8 ? 1 : 0;
    const int ncu = avs_persistent_slots();      // CUs a persistent grid may fill (device CUs - the cu_reserve knob)
    const int tiles = ceil_div(M, 256) * (N / 256);
    const int grid = tiles < ncu ? tiles : ncu;
    if (a_e5m2) {
        if (act == 0) gemm_nt8_kernel<0, 2><<<grid, 512, 131072, stream>>>(a);
        else gemm_nt8_kernel<2, 2><<<grid, 512, 131072, stream>>>(a);
    } else {
        if (act == 0) gemm_nt8_kernel<0, 1><<<grid, 512, 131072, stream>>>(a);
        else gemm_nt8_kernel<1, 1><<<grid, 512, 131072, stream>>>(a);
    }
    AVS_LAUNCH_CHECK("gemm_nt_fp8");
    return 0;
}

extern "C" int avs_gemm_nt_bf16_dual(const bf16_t* A, long long lda, const bf16_t* B, long long ldb, int M, int N, int K,
                                     const float* bias, const float* res, long long ldr, const int* res_idx, const bf16_t* aux,
                                     long long ldaux, void* out, long long ldo, int out_f32, bf16_t* out2, long long ldo2,
                                     float alpha, int act, int scale_cols, float col_scale, float* colsum,
                                     int m_split, const bf16_t* B2, const float* bias2, float* colsum2, hipStream_t stream) {
    AVS_CHECK_ARG(m_split > 0 && m_split < M && (m_split % 256) == 0, "gemm_nt_dual: m_split=%d must be a multiple of 256 inside (0, M=%d)", m_split, M);
    AVS_CHECK_ARG(B2 && (bias == nullptr) == (bias2 == nullptr) && (colsum == nullptr) == (colsum2 == nullptr),
                  "gemm_nt_dual: the second weight set must mirror the first (B2, bias2, colsum2)");
    return gemm_nt_launch(A, lda, B, ldb, M, N, K, bias, res, ldr, res_idx, aux, ldaux, out, ldo, out_f32, out2, ldo2, alpha, act, scale_cols,
                          col_scale, colsum, m_split, B2, bias2, colsum2, stream);
}

// Splits of the token dimension of a weight-gradient launch.  Every split divides the contraction but adds one full-tile fp32 atomic epilogue per
// output tile: tiles x splits x T^2 x 4 bytes at the ~1 TB/s the chip adds floats at (MI355X guide; 252 workgroups x 256 KiB = 48 us).  Rounds 1 - 4
// always filled one resident round (splits = slots / tiles), which is right for 95 630 rows and wrong below: at 27 419 rows the 768 x 768 gradient
// took 72.7 us with 28 splits and 50.0 with 7, at 11 328 rows 55.7 against 29.3, at 2 832 rows 44.6 against 14.9 (tools/bench_tn_splits.py,
// profiles/r05/tn_splits.log).  Minimising  t_stage x stages / s + t_atomic x tiles x s  gives  s* = sqrt(t_stage x stages / (t_atomic x tiles));
// fitted to that log: t_stage = 1.3 us per 64-row stage of a 256^2 tile (0.35 us for a 128^2 tile, two workgroups per CU), t_atomic = 0.2 us per
// 256-KiB tile (0.065 us per 64 KiB).  Capped by one resident round, which is what the 256^2 kernels still get from ~20 000 rows up.
static int tn_splits(int tiles, int nstages, int slots, double t_stage, double t_atomic) {
    if (avs_tuning().det) return 1;          // deterministic mode: one workgroup - one writer, one contraction order - per output tile
    int cap = slots / tiles;
    if (cap < 1) cap = 1;
    int s = (int)(sqrt(t_stage * nstages / (t_atomic * tiles)) + 0.5);
    if (s < 1) s = 1;
    return s < cap ? s : cap;
}

// Host-side plan of one weight-gradient launch - tile size, split count of the token rows, stages per split - as avs_gemm_tn_bf16 takes it
// (the knobs gemm_tile / gemm_nt8 / cu_reserve included).  No device work: exported as avs_gemm_tn_plan so that the heuristic is pinned by the
// CPU test suite against the measured optima of tools/bench_tn_splits.py.
struct TnPlan { int tile, tiles, splits, stages_per_split; };
static TnPlan tn_plan(int M, int N1, int N2, int splits) {
    const int nstages = ceil_div(M, BK);
    const bool can_big = (N1 % 256) == 0 && (N2 % 256) == 0;
    // 256^2 tiles need a long contraction to amortise their 256 KiB atomic epilogue per split
    // ... and enough output tiles that the splits (each adds a full-tile atomic epilogue) stay few
    // (8-phase kernel: from 12 tiles - the decoder's 1536x512 / 2048x512 gradients gain 14-18 % on it; 9 tiles and fewer lose)
    const int min_tiles = g_nt8 >= 1 ? 12 : 24;
    const bool big = g_force_tile == 256 ? can_big : g_force_tile == 128 ? false : (can_big && nstages >= 256 && (N1 / 256) * (N2 / 256) >= min_tiles);
    const int T = big ? 256 : 128;
    const int tiles = (N1 / T) * (N2 / T);
    if (splits <= 0) {
        // at most one resident round: 256^2 tiles hold 128 KiB of LDS (1 workgroup per CU), 128^2 tiles 64 KiB (2 per CU).  A grid
        // slightly LARGER than the resident slots would add a second, almost empty round that doubles the critical path,
        // so the cap is rounded DOWN (e.g. 27 tiles -> at most 9 splits = 243 workgroups on 256 CUs).
        const int ncu = avs_persistent_slots();      // CUs a persistent grid may fill (device CUs - the cu_reserve knob)
        const int slots = big ? ncu : 2 * ncu;
        splits = tn_splits(tiles, nstages, slots, big ? 1.3 : 0.35, big ? 0.2 : 0.065);
    }
    if (splits > nstages) splits = nstages;
    const int per = ceil_div(nstages, splits);
    splits = ceil_div(nstages, per);
    return TnPlan{T, tiles, splits, per};
}

extern "C" int avs_gemm_tn_plan(int M, int N1, int N2, int* tile, int* splits) {
    AVS_CHECK_ARG(M > 0 && N1 > 0 && N2 > 0 && (N1 % 128) == 0 && (N2 % 128) == 0 && tile && splits, "gemm_tn_plan: bad arguments M=%d N1=%d N2=%d", M, N1, N2);
    const TnPlan pl = tn_plan(M, N1, N2, 0);
    *tile = pl.tile;
    *splits = pl.splits;
    return 0;
}

extern "C" int avs_gemm_tn_bf16(const bf16_t* A, long long lda, const bf16_t* B, long long ldb, float* C, long long ldc,
                                int M, int N1, int N2, int splits, hipStream_t stream) {
    AVS_CHECK_ARG(M > 0 && (N1 % 128) == 0 && (N2 % 128) == 0, "gemm_tn: need N1%%128==0 and N2%%128==0 (N1=%d N2=%d)", N1, N2);
    AVS_CHECK_ARG(A && B && C && (lda % 8) == 0 && (ldb % 8) == 0, "gemm_tn: bad operands");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e != hipSuccess) {
            avs_set_error("gemm_tn: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return -1;
        }
        attr_done = true;
    }
    TnPlan pl = tn_plan(M, N1, N2, splits);
    const bool big = pl.tile == 256;
    const int tiles = pl.tiles, per = pl.stages_per_split;
    splits = pl.splits;
    GemmTnArgs a{A, lda, B, ldb, C, ldc, M, N1, N2, per};
    if (big && g_nt8 >= 1) {
        GemmTnGroupArgs g{};
        g.p[0] = TnProb{A, lda, B, ldb, C, ldc, N1, N2, 0};
        g.nprob = 1; g.tiles = tiles; g.M = M; g.stages_per_split = per;
        gemm_tn8_kernel<0><<<tiles * splits, 512, 131072, stream>>>(g);
    }
    else if (big) gemm_tn_kernel<4, 4><<<tiles * splits, 512, 131072, stream>>>(a);
    else gemm_tn_kernel<2, 2><<<tiles * splits, 256, 65536, stream>>>(a);
    AVS_LAUNCH_CHECK("gemm_tn");
    return 0;
}

// Up to three weight gradients over the same M token rows in ONE launch (problem i absent when Ai is NULL; problem 0 must exist):
// Ci[N1i, N2i] += Ai[M, N1i]^T . Bi[M, N2i].  8-phase 256 x 256 kernel only: every N1i, N2i a multiple of 256 and M >= 512.
// Shapes that do not qualify (or the 8-phase kernels switched off) are issued as one avs_gemm_tn_bf16 launch per problem.
// ---------------------------------------------------------------------------------------------------
// fp8 weight gradients (round 4; engine.FP8 = 3): C[N1,N2] += (1 / (sa sb)) (sa dY)^T (sb X) with dY in e5m2 and X in e4m3 - the copies
// the producers already write for the fp8 forward / input-gradient GEMMs.  Same 256 x 256 output tile, same split over the token rows and
// fp32 atomics as gemm_tn8_kernel; what changes is the contraction: a 64-row STAGE is one v_mfma_f32_32x32x64_f8f6f4 per 32 x 32 output
// block (64 cycles) instead of four v_mfma_f32_32x32x16_bf16 (4 x 32), and the staged tiles are [64 rows][256 bytes] (16 KB per operand).
// Fragments: lane (i = lane & 31, h = lane >> 5) of an operand holds the 32 contraction values k = 32 h .. 32 h + 31 of column i - four
// ds_read_b64_tr_b8 (probed on hardware: in a 16-lane group lane t addresses 8 bytes of row t / 2 at column 8 (t & 1); lane i receives
// column i of the 8 x 16 block, rows in byte order), i.e. a quarter of the bf16 kernel's LDS read instructions per contraction value -
// the bf16 kernel is bound by exactly those.  Conflict-free image: the 16-byte chunk c of row r sits at position c ^ ((r & 7) << 1) (applied
// to the DMA source address and to the read), so the 8 rows of a transposing read fall on 8 different bank groups.
// Three stage buffers of 32 KB: [A rows 0-31 | A rows 32-63 | B rows 0-31 | B rows 32-63], each 8-KB granule one 16-byte LDS-DMA per thread;
// stage t + 2 is requested when stage t's reads are issued, the counted wait leaves its four DMAs in flight; the two wave groups run one
// barrier apart (see the loop).
struct TnProb8 {
    const uint8_t* A; long long lda;         // e5m2 gradient [M, N1]
    const uint8_t* B; long long ldb;         // e4m3 activation [M, N2]
    float* C; long long ldc;
    const float* qa; const float* qb;        // device records of the two operands: the product is scaled by qa[INV] * qb[INV]
    int N1, N2, tile0;
};
struct GemmTn8Args {
    TnProb8 p[3];
    int nprob, tiles, M, stages_per_split;
};

__global__ __launch_bounds__(512) void gemm_tn8f_kernel(GemmTn8Args g) {
    constexpr int T1 = 256, T2 = 256, MI = 4;
    constexpr int GRAN = 32 * 256, STAGE = 4 * GRAN, NBUF = 3;          // 8-KB granule, 32-KB stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lid / g.tiles;
    int tile = lid % g.tiles;
    const int pi = (g.nprob > 2 && tile >= g.p[2].tile0) ? 2 : (g.nprob > 1 && tile >= g.p[1].tile0) ? 1 : 0;
    const TnProb8 a = g.p[pi];
    tile -= a.tile0;
    const int tiles2 = a.N2 / T2;
    const int n2_0 = (tile % tiles2) * T2, n1_0 = (tile / tiles2) * T1;
    const int nstages = (g.M + 63) / 64;
    const int s_begin = split * g.stages_per_split;
    const int nst = min(nstages, s_begin + g.stages_per_split) - s_begin;
    if (nst <= 0) return;
    const float alpha = a.qa[AVS_Q_INV] * a.qb[AVS_Q_INV];

    // this thread's chunk of a granule: row tid / 16 of its 32, physical chunk position tid % 16 <- logical chunk (tid % 16) ^ ((row & 7) << 1)
    const int grow = tid >> 4, gch = (tid & 15) ^ ((grow & 7) << 1);
    const uint8_t* srcA = a.A + (size_t)(s_begin * 64 + grow) * a.lda + n1_0 + gch * 16;
    const uint8_t* srcB = a.B + (size_t)(s_begin * 64 + grow) * a.ldb + n2_0 + gch * 16;
    auto dma = [&](int t) {                                              // the four granules of stage t
        char* dst = smem + (t % NBUF) * STAGE + wave * 1024;
        const size_t r = (size_t)t * 64;
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcA + r * a.lda), (LDS_AS void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcA + (r + 32) * a.lda), (LDS_AS void*)(dst + GRAN), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcB + r * a.ldb), (LDS_AS void*)(dst + 2 * GRAN), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(srcB + (r + 32) * a.ldb), (LDS_AS void*)(dst + 3 * GRAN), 16, 0, 0);
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets inside a stage: lane = 16 G + t; h = G >> 1 picks the granule (rows 32 h ..), G & 1 the 16-column half of the
    // 32-column block; the read of row group q (rows 8 q .. 8 q + 7 of the granule) is the immediate offset q * 2048
    const int t16 = lane & 15, G = lane >> 4, hh = G >> 1, cb = G & 1;
    auto frag_off = [&](int colbase) {
        const int c = (colbase >> 4) + cb;                                  // logical 16-byte chunk of the row
        return hh * GRAN + (t16 >> 1) * 256 + ((c ^ (t16 & 14)) << 4) + (t16 & 1) * 8;
    };
    unsigned offA[MI], offB[2];
    const unsigned smem_lds = (unsigned)(size_t)(LDS_AS const char*)smem;
#pragma unroll
    for (int i = 0; i < MI; ++i) offA[i] = smem_lds + frag_off(wr * (MI * 32) + i * 32);
#pragma unroll
    for (int j = 0; j < 2; ++j) offB[j] = smem_lds + 2 * GRAN + frag_off(wc * 64 + j * 32);

    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // The two wave groups (wr = 0 / 1: one wave of each per SIMD) run the same two-barrier stage loop ONE barrier apart, so a group's 24
    // fragment reads, its LDS-DMA issue and its waits run under the other group's eight MFMAs (512 matrix-pipe cycles) instead of beside
    // its own reads.  Barrier k of group 0 is X1(k / 2) or X2(k / 2), of group 1 the extra one, then X1, X2, ...:
    //   X1(t): stage t is complete in LDS - every wave waited for its own part of it in front of the previous barrier (prologue for t = 0);
    //   then reads(t), the DMAs of stage t + 2 (into the buffer of stage t - 1: whichever group issues them, both have finished reading it -
    //   the other group is one barrier behind or ahead and drained its reads in front of its X2), drain the reads, wait for the own part of
    //   stage t + 1; X2(t); MFMAs(t).
    dma(0);
    if (nst > 1) { dma(1); wait_vm<4>(); } else wait_vm<0>();
    if (wr == 1) bar();
    for (int t = 0; t < nst; ++t) {
        bar();                                                             // X1(t)
        const unsigned sb = (t % NBUF) * STAGE;
        unsigned long long fa[MI][4], fb[2][4];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const unsigned ad = offA[i] + sb;
            asm volatile("ds_read_b64_tr_b8 %0, %4\n\tds_read_b64_tr_b8 %1, %4 offset:2048\n\tds_read_b64_tr_b8 %2, %4 offset:4096\n\tds_read_b64_tr_b8 %3, %4 offset:6144"
                         : "=&v"(fa[i][0]), "=&v"(fa[i][1]), "=&v"(fa[i][2]), "=&v"(fa[i][3]) : "v"(ad) : "memory");
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned ad = offB[j] + sb;
            asm volatile("ds_read_b64_tr_b8 %0, %4\n\tds_read_b64_tr_b8 %1, %4 offset:2048\n\tds_read_b64_tr_b8 %2, %4 offset:4096\n\tds_read_b64_tr_b8 %3, %4 offset:6144"
                         : "=&v"(fb[j][0]), "=&v"(fb[j][1]), "=&v"(fb[j][2]), "=&v"(fb[j][3]) : "v"(ad) : "memory");
        }
        if (t + 2 < nst) dma(t + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + 2 < nst) wait_vm<4>(); else wait_vm<0>();                  // this wave's part of stage t + 1 has landed (stage t + 2 stays in flight)
        bar();                                                             // X2(t)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const i32x8 av = {(int)fa[i][0], (int)(fa[i][0] >> 32), (int)fa[i][1], (int)(fa[i][1] >> 32), (int)fa[i][2], (int)(fa[i][2] >> 32),
                              (int)fa[i][3], (int)(fa[i][3] >> 32)};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const i32x8 bv = {(int)fb[j][0], (int)(fb[j][0] >> 32), (int)fb[j][1], (int)(fb[j][1] >> 32), (int)fb[j][2], (int)(fb[j][2] >> 32),
                                  (int)fb[j][3], (int)(fb[j][3] >> 32)};
                // cbsz = 1: the first operand (the gradient) is e5m2; blgp = 0: the second (the activation) e4m3; unscaled
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[i][j], 1, 0, 0, 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wr == 0) bar();                                                    // pairs with the other group's last X2

    const int h = lane >> 5;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n2 = n2_0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n1 = n1_0 + wr * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                atomicAdd(a.C + (size_t)n1 * a.ldc + n2, alpha * acc[i][j][r]);
            }
        }
}

// up to three fp8 weight gradients over the same M token rows: Ci[N1_i, N2_i] += Ai^T . Bi / (scale_a scale_b), Ai e5m2 [M, N1_i], Bi e4m3
// [M, N2_i] (both allocated and ZERO up to the next multiple of 64 rows), qai / qbi their device records; every N a multiple of 256
extern "C" int avs_gemm_tn_fp8_group3(const uint8_t* A0, long long lda0, const uint8_t* B0, long long ldb0, float* C0, int N1_0, int N2_0, const float* qa0, const float* qb0,
                                      const uint8_t* A1, long long lda1, const uint8_t* B1, long long ldb1, float* C1, int N1_1, int N2_1, const float* qa1, const float* qb1,
                                      const uint8_t* A2, long long lda2, const uint8_t* B2, long long ldb2, float* C2, int N1_2, int N2_2, const float* qa2, const float* qb2,
                                      int M, hipStream_t stream) {
    const uint8_t* As[3] = {A0, A1, A2};
    const uint8_t* Bs[3] = {B0, B1, B2};
    float* Cs[3] = {C0, C1, C2};
    const float* qas[3] = {qa0, qa1, qa2};
    const float* qbs[3] = {qb0, qb1, qb2};
    const long long las[3] = {lda0, lda1, lda2}, lbs[3] = {ldb0, ldb1, ldb2};
    const int n1s[3] = {N1_0, N1_1, N1_2}, n2s[3] = {N2_0, N2_1, N2_2};
    AVS_CHECK_ARG(A0 && M > 0, "gemm_tn_fp8_group3: the first problem must exist");
    GemmTn8Args g{};
    int n = 0, tiles = 0;
    for (int i = 0; i < 3; ++i) {
        if (!As[i]) continue;
        AVS_CHECK_ARG(Bs[i] && Cs[i] && qas[i] && qbs[i] && n1s[i] > 0 && n2s[i] > 0 && (las[i] % 16) == 0 && (lbs[i] % 16) == 0 && las[i] >= n1s[i] && lbs[i] >= n2s[i],
                      "gemm_tn_fp8_group3: bad operands of problem %d", i);
        AVS_CHECK_ARG((n1s[i] % 256) == 0 && (n2s[i] % 256) == 0, "gemm_tn_fp8_group3: N1 and N2 must be multiples of 256 (problem %d: %d x %d)", i, n1s[i], n2s[i]);
        g.p[n] = TnProb8{As[i], las[i], Bs[i], lbs[i], Cs[i], (long long)n2s[i], qas[i], qbs[i], n1s[i], n2s[i], tiles};
        tiles += (n1s[i] / 256) * (n2s[i] / 256);
        ++n;
    }
    const int nstages = ceil_div(M, 64);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)gemm_tn8f_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 98304) != hipSuccess) {
            avs_set_error("gemm_tn_fp8_group3: hipFuncSetAttribute failed");
            return -1;
        }
        attr_done = true;
    }
    const int ncu = avs_persistent_slots();      // CUs a persistent grid may fill (device CUs - the cu_reserve knob)
    int splits = tn_splits(tiles, nstages, ncu, 1.0, 0.2);      // at most one resident round (see tn_splits)
    if (splits > nstages / 2) splits = nstages / 2 > 0 ? nstages / 2 : 1;
    const int per = ceil_div(nstages, splits);
    splits = ceil_div(nstages, per);
    g.nprob = n; g.tiles = tiles; g.M = M; g.stages_per_split = per;
    gemm_tn8f_kernel<<<tiles * splits, 512, 98304, stream>>>(g);
    AVS_LAUNCH_CHECK("gemm_tn_fp8_group3");
    return 0;
}

extern "C" int avs_gemm_tn_bf16_group3(const bf16_t* A0, long long lda0, const bf16_t* B0, long long ldb0, float* C0, int N1_0, int N2_0,
                                       const bf16_t* A1, long long lda1, const bf16_t* B1, long long ldb1, float* C1, int N1_1, int N2_1,
                                       const bf16_t* A2, long long lda2, const bf16_t* B2, long long ldb2, float* C2, int N1_2, int N2_2,
                                       int M, hipStream_t stream) {
    const bf16_t* As[3] = {A0, A1, A2};
    const bf16_t* Bs[3] = {B0, B1, B2};
    float* Cs[3] = {C0, C1, C2};
    const long long las[3] = {lda0, lda1, lda2}, lbs[3] = {ldb0, ldb1, ldb2};
    const int n1s[3] = {N1_0, N1_1, N1_2}, n2s[3] = {N2_0, N2_1, N2_2};
    AVS_CHECK_ARG(A0 && M > 0, "gemm_tn_group3: the first problem must exist");
    GemmTnGroupArgs g{};
    int n = 0, tiles = 0;
    bool ok = true;
    for (int i = 0; i < 3; ++i) {
        if (!As[i]) continue;
        AVS_CHECK_ARG(Bs[i] && Cs[i] && n1s[i] > 0 && n2s[i] > 0 && (las[i] % 8) == 0 && (lbs[i] % 8) == 0, "gemm_tn_group3: bad operands of problem %d", i);
        if ((n1s[i] % 256) || (n2s[i] % 256)) ok = false;
        g.p[n] = TnProb{As[i], las[i], Bs[i], lbs[i], Cs[i], (long long)n2s[i], n1s[i], n2s[i], tiles};
        tiles += (n1s[i] / 256) * (n2s[i] / 256);
        ++n;
    }
    const int nstages = ceil_div(M, BK);
    if (!ok || g_nt8 < 1 || g_force_tile == 128 || n < 2 || nstages < 8) {
        for (int i = 0; i < 3; ++i)
            if (As[i])
                if (int e = avs_gemm_tn_bf16(As[i], las[i], Bs[i], lbs[i], Cs[i], n2s[i], M, n1s[i], n2s[i], 0, stream)) return e;
        return 0;
    }
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)gemm_tn8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) {
            avs_set_error("gemm_tn_group3: hipFuncSetAttribute failed");
            return -1;
        }
        attr_done = true;
    }
    const int ncu = avs_persistent_slots();      // CUs a persistent grid may fill (device CUs - the cu_reserve knob)
    int splits = tn_splits(tiles, nstages, ncu, 1.3, 0.2);      // at most one resident round (see tn_splits)
    if (splits > nstages / 2) splits = nstages / 2 > 0 ? nstages / 2 : 1;
    const int per = ceil_div(nstages, splits);
    splits = ceil_div(nstages, per);
    g.nprob = n; g.tiles = tiles; g.M = M; g.stages_per_split = per;
    gemm_tn8_kernel<0><<<tiles * splits, 512, 131072, stream>>>(g);
    AVS_LAUNCH_CHECK("gemm_tn_group3");
    return 0;
}
