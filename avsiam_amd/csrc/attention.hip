// k4 - variable-length multi-head self-attention, forward and backward, for the packed token matrix.
//
// Replaces F.scaled_dot_product_attention inside Attention.forward
// (/root/reference/src/models/cav_mae_base.py:60-68: qkv.reshape(B,N,3,H,hd) -> softmax(q k^T / sqrt(hd)) v,
// no mask, dropout 0) and its autograd backward, for ragged sequences: the reference calls the block once per
// multi-ratio group (:554-558); here every sequence of a pass lives in one [rows, 3*D] qkv matrix and a
// (sequence start, length, q0) triple per 128-row tile drives the grid.
//
// Structure (gfx950, wave64, v_mfma_f32_32x32x16_bf16, fp32 softmax):
//   * the score tile is computed TRANSPOSED (S^T = K.Q^T, keys on the accumulator rows, the query on the lane),
//     so the row max / row sum of a query are per-lane scalars (+ one cross-half shuffle) and the P tile is, with
//     no lane movement, the B operand of the next product O^T += V^T.P^T;
//   * K / V (or Q / dO in the dK,dV kernel) tiles are staged in LDS in 8-row x 32-column sub-tiles with a
//     (row>>2)&3 chunk XOR - one image that is conflict-free for both the row reads (ds_read_b128) feeding the
//     score products and the transposing reads (ds_read_b64_tr_b16) feeding the products that contract over the
//     tile's rows;
//   * backward recomputes P from the saved log-sum-exp (no N x N tensor), in two kernels: dQ (query-major, same
//     shape as forward) and dK/dV (key-major), so no atomics and bitwise-reproducible gradients.
//   * the softmax is VALU-bound at these head dims (measured, tools/valu_rate.py: v_exp_f32 8 SIMD cycles per wave64
//     instruction, v_max3 / v_cvt_pk_bf16 / packed fp32 ops 4, plain fp32 mul / add / fma 2 - against 32 MFMA cycles per
//     32x32x16 product, i.e. at hd 32 each 32-key block costs 128 cycles of v_exp alone beside 128 of MFMA), so the per-score arithmetic is pushed into the MFMA's C operand: q arrives
//     PRE-MULTIPLIED by hd^-0.5 * log2(e) (epilogue of the qkv GEMM, one bf16 rounding as before), the score
//     accumulators start from -running_max (forward) / -lse (backward) and the dP accumulators from -delta, so the
//     matrix core delivers exp2's argument and (dP - delta) directly.  Forward takes its reference point from the first
//     key tile and moves it only when a later tile's row sum says so (no per-tile max; exact: the final division and
//     the saved lse use the same reference).
// Head dims 64 (encoder, 12 heads) and 32 (decoder, 16 heads).
#include "common.h"
#include <type_traits>

#define LDS_AS __attribute__((address_space(3)))
#ifndef ATTN_MFMA_ROWSUM
#define ATTN_MFMA_ROWSUM 0          // 1: diagnostic builds only (attn_fwd_kernel)
#endif

template <int HD>
struct Img {                                   // LDS image of a [rows][HD] bf16 tile
    static constexpr int NSUB = HD / 32;       // 32-column sub-tiles per row
    static constexpr int NCH = HD / 8;         // 16-byte chunks per row
    __device__ static __forceinline__ int off(int row, int ch) {
        return (row >> 3) * (512 * NSUB) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
    }
};

// A-operand fragment of v_mfma_f32_32x32x16_bf16 read by ROWS: lane (r = lane&31, h = lane>>5) gets
// tile[rowbase + r][16*kk + 8h .. +7]
template <int HD>
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int rowbase, int kk, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + Img<HD>::off(rowbase + (lane & 31), 2 * kk + (lane >> 5)));
}

// A-operand fragment of the TRANSPOSED tile: MFMA row = tile column db*32 + (lane&31), k = tile rows
// rowbase16 + {8(j>>2) + 4h + (j&3)} - the k order of an accumulator tile used as the other operand.
template <int HD>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int rowbase16, int db, int lane) {
    const int h = lane >> 5, cb = (lane >> 4) & 1, li = lane & 15, q = li >> 2, p4 = li & 3;
    const int ch = db * 4 + 2 * cb + (p4 >> 1);
    const int r0 = rowbase16 + 4 * h + q;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(tile + Img<HD>::off(r0, ch) + 8 * (p4 & 1)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(tile + Img<HD>::off(r0 + 8, ch) + 8 * (p4 & 1)));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// accumulator registers 8s..8s+7 -> bf16 fragment (B operand, k-step s)
__device__ __forceinline__ bf16x8 acc_frag(const float* v) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    u32x4 u = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, u);
}

// Staging a [64][HD] tile is split (issue early / write late): tile_load puts the next tile's global loads in flight
// before the current tile's MFMA work, tile_store writes them into the LDS image after the barrier that retires the
// previous tile's reads - the HBM/L2 latency hides under the compute instead of being paid per tile.
template <int HD, int NTH>
struct TileRegs {
    static constexpr int NCH = HD / 8;
    static constexpr int PER = 64 * NCH / NTH;
    f32x4 v[PER];
};

// HG: the head dim in memory when it is not a multiple of 32 (80: ViT-H); the LDS image is HD = 96 wide and its columns HG.. are zero.
// Addressing: a thread's chunk of a [64][HD] tile sits at the same (row, column) of EVERY tile, so its byte offset from the tile's
// first row is computed once per kernel (TileOff, one 32-bit register per chunk, shared by the K and V - or Q and dO - streams of the
// same leading dimension); per tile only the wave-uniform tile base moves (scalar arithmetic) and the load takes the
// scalar-base + 32-bit-vector-offset form.  Recomputed with 64-bit multiplies per load it cost ~70 VALU issue cycles per tile and
// wave in kernels whose bound IS the VALU issue (decoder attention: ~10 % of it).  Only a sequence's last, partial tile clamps rows.
template <int HD, int NTH>
struct TileOff {
    static constexpr int NCH = HD / 8;
    static constexpr int PER = 64 * NCH / NTH;
    unsigned o[PER];
    __device__ __forceinline__ void init(long long ld, int tid) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = i * NTH + tid;
            o[i] = ((unsigned)(c / NCH) * (unsigned)ld + (unsigned)(c % NCH) * 8u) * 2u;
        }
    }
};

template <int HD, int NTH, int HG = HD>
__device__ __forceinline__ void tile_load(TileRegs<HD, NTH>& t, const TileOff<HD, NTH>& to, const bf16_t* src, long long ld, int row0, int last_row,
                                          int tid) {
    constexpr int NCH = HD / 8;
    const char* base = reinterpret_cast<const char*>(src + (size_t)row0 * ld);       // wave-uniform
    unsigned off[TileRegs<HD, NTH>::PER];
#pragma unroll
    for (int i = 0; i < TileRegs<HD, NTH>::PER; ++i) off[i] = to.o[i];
    if (row0 + 63 > last_row) {                                // block-uniform: the sequence's last, partial tile clamps its rows
        asm volatile("; partial tile: clamp rows" ::: "memory");
#pragma unroll
        for (int i = 0; i < TileRegs<HD, NTH>::PER; ++i) {
            const int c = i * NTH + tid;
            off[i] = ((unsigned)min(c / NCH, last_row - row0) * (unsigned)ld + (unsigned)(c % NCH) * 8u) * 2u;
        }
    }
#pragma unroll
    for (int i = 0; i < TileRegs<HD, NTH>::PER; ++i) {
        const int ch = (i * NTH + tid) % NCH;
        if (HG == HD || ch < HG / 8) t.v[i] = *reinterpret_cast<const f32x4*>(base + off[i]);
        else t.v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <int HD, int NTH>
__device__ __forceinline__ void tile_store(const TileRegs<HD, NTH>& t, char* tile, int tid) {
    constexpr int NCH = HD / 8;
#pragma unroll
    for (int i = 0; i < TileRegs<HD, NTH>::PER; ++i) {
        const int c = i * NTH + tid;
        *reinterpret_cast<f32x4*>(tile + Img<HD>::off(c / NCH, c % NCH)) = t.v[i];
    }
}

// One 32-column block of an accumulator tile -> a row of the output.  A lane holds 16 values of ONE row (its query / key): four
// consecutive columns at column 8 t + 4 hh for t = 0..3; lanes l and l + 32 hold the two halves of the same row.  Stored as they stand
// that is four 8-byte pieces per lane and block (four 4-byte pieces for an fp8 copy), and the per-CU store path is issue-bound (cf. the
// GEMM epilogue).  v_permlane32_swap hands each lane its partner's half of a 16-column pair instead: two 16-byte stores (two 8-byte
// ones for fp8) of 8 consecutive columns.  ATTN_WIDE_STORE=0 (diagnostic builds) keeps the narrow form for A/B.
#ifndef ATTN_WIDE_STORE
#define ATTN_WIDE_STORE 1
#endif
template <int HG, bool BF8>
__device__ __forceinline__ void store_block(bf16_t* rowp, int d, int hh, const f32x16& acc, float mul, uint8_t* rowp8, float s8, float& gmax) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    auto q8 = [&](float a, float b, float c, float e) {          // four values -> one dword of fp8 (e5m2 for gradients, e4m3 for activations)
        gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(a), fabsf(b))), fmaxf(fabsf(c), fabsf(e)));
        constexpr float M = BF8 ? 57344.f : 448.f;
        int w;
        if (BF8) {
            w = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(a * s8, -M, M), __builtin_amdgcn_fmed3f(b * s8, -M, M), 0, false);
            w = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(c * s8, -M, M), __builtin_amdgcn_fmed3f(e * s8, -M, M), w, true);
        } else {
            w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a * s8, -M, M), __builtin_amdgcn_fmed3f(b * s8, -M, M), 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c * s8, -M, M), __builtin_amdgcn_fmed3f(e * s8, -M, M), w, true);
        }
        return (uint32_t)w;
    };
#if ATTN_WIDE_STORE
    uint32_t p8[2][2] = {{0u, 0u}, {0u, 0u}};                      // the 8-bit copy: [u] = {columns 16u + 8hh .. + 3, .. + 4 .. + 7} after the first exchange
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (d * 32 + 16 * u >= HG) continue;                       // (HG is a multiple of 16: a pair is valid or not as a whole)
        const float e0 = acc[8 * u + 0] * mul, e1 = acc[8 * u + 1] * mul, e2 = acc[8 * u + 2] * mul, e3 = acc[8 * u + 3] * mul;      // t = 2u
        const float o0 = acc[8 * u + 4] * mul, o1 = acc[8 * u + 5] * mul, o2 = acc[8 * u + 6] * mul, o3 = acc[8 * u + 7] * mul;      // t = 2u + 1
        // lanes 0-31 keep their t = 2u piece and receive the partner's (columns 16u .. 16u + 7); lanes 32-63 receive the partner's t = 2u + 1
        // piece and keep theirs (columns 16u + 8 .. 16u + 15)
        const auto r0 = __builtin_amdgcn_permlane32_swap(pack_bf2(e0, e1), pack_bf2(o0, o1), false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(pack_bf2(e2, e3), pack_bf2(o2, o3), false, false);
        if (rowp) *reinterpret_cast<u32x4*>(rowp + d * 32 + 16 * u + 8 * hh) = u32x4{r0[0], r1[0], r0[1], r1[1]};      // (NULL: the 8-bit copy only)
        if (rowp8) {
            const auto r8 = __builtin_amdgcn_permlane32_swap(q8(e0, e1, e2, e3), q8(o0, o1, o2, o3), false, false);
            p8[u][0] = r8[0]; p8[u][1] = r8[1];
        }
    }
    if (rowp8) {
        if (d * 32 + 16 < HG) {
            // both 16-column pairs exist: a second exchange (pair 0 of the upper lanes against pair 1 of the lower ones) leaves lanes 0-31 with
            // columns 0 .. 15 and lanes 32-63 with 16 .. 31 of the block - ONE 16-byte store per lane instead of two 8-byte ones (these copies
            // cost the short-sequence kernels 20 - 35 % as 8-byte pieces)
            const auto xa = __builtin_amdgcn_permlane32_swap(p8[0][0], p8[1][0], false, false);
            const auto xb = __builtin_amdgcn_permlane32_swap(p8[0][1], p8[1][1], false, false);
            *reinterpret_cast<u32x4*>(rowp8 + d * 32 + 16 * hh) = u32x4{xa[0], xb[0], xa[1], xb[1]};
        } else if (d * 32 < HG) {
            *reinterpret_cast<uint2*>(rowp8 + d * 32 + 8 * hh) = make_uint2(p8[0][0], p8[0][1]);      // (hd 80: the last block has one pair)
        }
        // one block's conversions at a time: scheduled across blocks, the temporaries of the 8-bit copies cost the hd-64 kernels an occupancy
        // step (dq<64,2> 164 -> 174 registers, fused<64,2> 166 -> 186: two waves per SIMD instead of three)
        __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (d * 32 + 8 * t >= HG) continue;
        const float v0 = acc[4 * t + 0] * mul, v1 = acc[4 * t + 1] * mul, v2 = acc[4 * t + 2] * mul, v3 = acc[4 * t + 3] * mul;
        if (rowp) *reinterpret_cast<uint2*>(rowp + d * 32 + 8 * t + 4 * hh) = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
        if (rowp8) *reinterpret_cast<uint32_t*>(rowp8 + d * 32 + 8 * t + 4 * hh) = q8(v0, v1, v2, v3);
    }
#endif
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// Packed fp32 (v_pk_add_f32 / v_pk_mul_f32) must stay out of these loops: beside MFMAs a packed fp32 instruction costs the SIMD's issue port
// about twice a plain one, and under -O3 hipcc's SLP vectoriser packs adjacent scalar adds / subs / muls (round 5: the 16 v_pk_add_f32 of a
// 64-key tile's row sum).  This file is therefore compiled with -fno-slp-vectorize (avsiam_amd/build.py) and the arithmetic is written
// scalar.  NOT by inline asm: an asm v_add / v_mul that reads a v_exp or MFMA result hides the producer -> consumer wait states from hipcc's
// hazard recogniser (tried first: NaNs and wrong sums on hardware).
__device__ __forceinline__ float add1(float a, float b) { return a + b; }
__device__ __forceinline__ float sub1(float a, float b) { return a - b; }
__device__ __forceinline__ float mul1(float a, float b) { return a * b; }
constexpr float LN2 = 0.6931471805599453f;

struct AttnArgs {
    const bf16_t* qkv; long long ld; int D;     // [rows, 3*D]: q | k | v, head h at columns h*HD
    const int* tile_start;                      // first packed row of the tile's sequence
    const int* tile_len;                        // length of that sequence
    const int* tile_q0;                         // first row (within the sequence) of this (32 * waves)-row tile
    int ntiles;
    bf16_t* out; long long ldo;                 // fwd: attention output [rows, D]
    float* lse; int rows_total;                 // [H][rows_total] natural-log sum-exp of the scaled scores
    const bf16_t* dout;                         // bwd: dO [rows, D] (ldo)
    float* delta;                               // bwd: rowsum(dO * O) [H][rows_total]
    bf16_t* dqkv;                               // bwd: [rows, 3*D] (ld)
    float scale;                                // hd^-0.5 (q columns of qkv hold q * scale * log2(e))
    uint8_t* out8; long long ldo8; float* q8;   // fwd, fp8 mode (may be NULL): e4m3 copy of the output for the proj GEMM, scaled by the
                                                // device record q8 (common.h AVS_Q_*), whose running amax takes the largest |o| written
    uint8_t* dqkv8; long long ld8; float* qd8;  // bwd, fp8 mode (may be NULL): e5m2 copy of dqkv [rows, 3*D] - the gradient operand of the fp8 qkv
                                                // input-gradient GEMM - scaled by the device record qd8, whose running amax takes the largest |dqkv|
    int kv16;                                   // bwd with dqkv8: 0 = the key / value thirds of the bf16 dqkv are NOT written (fp8 mode 3: the input- and
                                                // weight-gradient GEMMs read the e5m2 copy, only the query third's column sum - the bias gradient - reads bf16)
    int lq;                                     // > 0 (equal-length sequences only): only the FIRST lq rows of every sequence are queries - keys / values are all
                                                // of its rows - and `out` / `dout` are COMPACT: sequence s (= first row / length) owns their rows s * lq .. + lq - 1.
                                                // The decoder's last block (cav_mae_base.py:629-635,679-682): rows whose prediction is never scored (mask 0) feed
                                                // nothing but the keys and values of that block; lse / delta / dqkv keep the packed row numbering.  0: every row.
};

// queries of a sequence of length L, and the row of `out` / `dout` that holds the sequence's first query (AttnArgs::lq)
__device__ __forceinline__ int attn_lq(const AttnArgs& a, int L) { return a.lq > 0 ? a.lq : L; }
__device__ __forceinline__ size_t attn_orow0(const AttnArgs& a, int seq0, int L) { return a.lq > 0 ? (size_t)(seq0 / L) * (size_t)a.lq : (size_t)seq0; }

// four consecutive gradient values -> one dword of e5m2 at `dst` (scaled, clamped to the e5m2 range), their |max| folded into gmax
__device__ __forceinline__ void store_bf8x4(uint8_t* dst, float v0, float v1, float v2, float v3, float scale, float& gmax) {
    gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(v0), fabsf(v1))), fmaxf(fabsf(v2), fabsf(v3)));
    int w = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(v0 * scale, -57344.f, 57344.f), __builtin_amdgcn_fmed3f(v1 * scale, -57344.f, 57344.f), 0, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(v2 * scale, -57344.f, 57344.f), __builtin_amdgcn_fmed3f(v3 * scale, -57344.f, 57344.f), w, true);
    *reinterpret_cast<int*>(dst) = w;
}

__device__ __forceinline__ f32x16 splat16(float v) {
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = v;
    asm volatile("" : "+v"(t));                 // keep the 16 registers: they are an MFMA C operand, not a scalar to re-broadcast
    return t;
}

// ---------------------------------------------------------------------------------------------------
template <int HD, int NW, int HG = HD>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(AttnArgs a) {
    constexpr int NTH = 64 * NW;
    constexpr int NKK = HG / 16, NDB = HD / 32;          // contraction steps over the real head dim; 32-wide output blocks of the image
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * HD * 2];
    char* sK = smem;
    char* sV = smem + 64 * HD * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    // fp8 mode: the output record's scale and the amax it already holds are requested up front (their latency hides under the key loop)
    const float q8s = a.out8 ? a.q8[AVS_Q_SCALE] : 0.f;
    const float q8seen = q_amax_peek(a.out8 ? a.q8 : nullptr);
    // 1-D grid, XCD-aware: logical id = head * ntiles + tile, so the query/key tiles of one (sequence, head) - which all
    // stream the same K/V (or Q/dO) rows - run on one XCD and re-read them from its L2 instead of from HBM.
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int head = lid / a.ntiles, tix = lid - head * a.ntiles;
    const int seq0 = a.tile_start[tix], L = a.tile_len[tix];
    const int Lq = attn_lq(a, L);                       // (AttnArgs::lq: the queries are the first Lq rows of the sequence; normally all L)
    if (a.tile_q0[tix] >= Lq) return;                   // block-uniform, before any barrier: a tile of rows that are keys / values only
    const size_t orow0 = attn_orow0(a, seq0, L);
    const int qw = a.tile_q0[tix] + 32 * wave;          // first query of this wave
    const bool active = qw < Lq;
    const int q = min(qw + (lane & 31), Lq - 1);
    const bf16_t* base = a.qkv + (size_t)seq0 * a.ld + head * HG;

    bf16x8 qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(base + (size_t)q * a.ld + (2 * kk + hh) * 8);

    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    // scores are in log2 units (q is pre-scaled).  m_run is the reference point the accumulators start from: the first tile's
    // row max, moved only when a later tile's row sum of p = exp2(s - m_run) reaches LAZY_SUM = 2^40 (far from fp32 overflow:
    // 2472 keys x 2^40 x |v| stays below 2^60) - the final division and the saved lse use the same reference, so the result
    // is the softmax whatever the reference is.
    constexpr float LAZY_SUM = 1099511627776.0f;
    float m_run = 0.f, l_run = 0.f;
    f32x16 negm = splat16(0.f);
#if ATTN_MFMA_ROWSUM
    // DIAGNOSTIC build only (tools/ab_lib.sh with AVSIAM_HIPCC_EXTRA=-DATTN_MFMA_ROWSUM=1): the row sum of P on the matrix pipe - one
    // extra product per 16-key step with an all-ones A operand (every row of the result is sum_k P[k][q]) - instead of 32 VALU adds per tile.  No
    // lazy-rescale detector in this form (it would need the sum before the P.V products): timing on random data only.
    f32x16 lsum = splat16(0.f);
    const bf16x8 ones = {(bf16_t)0x3F80, (bf16_t)0x3F80, (bf16_t)0x3F80, (bf16_t)0x3F80, (bf16_t)0x3F80, (bf16_t)0x3F80, (bf16_t)0x3F80, (bf16_t)0x3F80};
#endif

    TileRegs<HD, NTH> rk, rv;
    TileOff<HD, NTH> toff;
    toff.init(a.ld, tid);
    tile_load<HD, NTH, HG>(rk, toff, base + a.D, a.ld, 0, L - 1, tid);
    tile_load<HD, NTH, HG>(rv, toff, base + 2 * a.D, a.ld, 0, L - 1, tid);
    for (int k0 = 0; k0 < L; k0 += 64) {
        __syncthreads();
        tile_store<HD, NTH>(rk, sK, tid);
        tile_store<HD, NTH>(rv, sV, tid);
        __syncthreads();
        if (k0 + 64 < L) {
            tile_load<HD, NTH, HG>(rk, toff, base + a.D, a.ld, k0 + 64, L - 1, tid);
            tile_load<HD, NTH, HG>(rv, toff, base + 2 * a.D, a.ld, k0 + 64, L - 1, tid);
        }
        if (!active) continue;
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sK, kb * 32, 0, lane), qf[0], negm, 0, 0, 0);
#pragma unroll
            for (int kk = 1; kk < NKK; ++kk)
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sK, kb * 32, kk, lane), qf[kk], s[kb], 0, 0, 0);
        }
        if (k0 + 64 > L) {
            // block-uniform branch; the empty asm keeps hipcc from if-converting the body into per-element selects that
            // every (full) tile would execute
            asm volatile("; tail key tile: mask" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[kb][r] = -INFINITY;
        }
        if (k0 == 0) {
            // the first tile fixes the reference point at its row max
            float t = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) t = fmaxf(t, s[kb][r]);
            t = fmaxf(t, __shfl_xor(t, 32, 64));
            m_run = t;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] -= t;
            negm = splat16(-m_run);
        }
        float psum;
        float p[2][16];
        auto exp_tile = [&]() {
            // two running sums (even / odd scores: the order the packed adds of rounds 1 - 4 summed in, bit for bit) with PLAIN adds - add1
            float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    p[kb][r] = fast_exp2(s[kb][r]);
                    p[kb][r + 1] = fast_exp2(s[kb][r + 1]);
#if !ATTN_MFMA_ROWSUM
                    ps0 = add1(ps0, p[kb][r]);
                    ps1 = add1(ps1, p[kb][r + 1]);
#endif
                }
            psum = ps0 + ps1;
        };
        exp_tile();
        if (k0 != 0 && __any(!(psum < LAZY_SUM))) {
            // wave-uniform and rare: some score of this tile lies more than ~2^LAZY above the reference point (or the sum overflowed).
            // Recompute the tile's scores (K is still in LDS), move the reference point to the new max and rescale what has been
            // accumulated - exactly the online-softmax step, taken only when it is needed.  No per-tile max otherwise: the row sum
            // the tile needs anyway is the detector.
            asm volatile("; softmax: new reference max" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sK, kb * 32, 0, lane), qf[0], negm, 0, 0, 0);
#pragma unroll
                for (int kk = 1; kk < NKK; ++kk)
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sK, kb * 32, kk, lane), qf[kk], s[kb], 0, 0, 0);
            }
            float t = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (k0 + 64 > L && k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[kb][r] = -INFINITY;
                    t = fmaxf(t, s[kb][r]);
                }
            t = fmaxf(t, __shfl_xor(t, 32, 64));
            const float d = fmaxf(t, 0.f);
            const float alpha = fast_exp2(-d);
            l_run *= alpha;
#pragma unroll
            for (int dd = 0; dd < NDB; ++dd)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
            m_run += d;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] -= d;
            negm = splat16(-m_run);
            exp_tile();
        }
        l_run += psum;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8 pf = acc_frag(&p[kb][8 * st]);
#pragma unroll
                for (int d = 0; d < NDB; ++d)
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sV, kb * 32 + 16 * st, d, lane), pf, o[d], 0, 0, 0);
#if ATTN_MFMA_ROWSUM
                lsum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lsum, 0, 0, 0);
#endif
            }
    }
    if (!active) return;
#if ATTN_MFMA_ROWSUM
    const float l_tot = lsum[0];
#else
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
#endif
    const float inv = 1.0f / l_tot;
    const int qq = qw + (lane & 31);
    float omax = 0.f;
    if (qq < Lq) {
        bf16_t* orow = a.out + (orow0 + qq) * a.ldo + head * HG;
        uint8_t* orow8 = a.out8 ? a.out8 + (orow0 + qq) * a.ldo8 + head * HG : nullptr;
#pragma unroll
        for (int d = 0; d < NDB; ++d) store_block<HG, false>(orow, d, hh, o[d], inv, orow8, q8s, omax);      // (columns of the image beyond the real head dim are skipped)
        if (hh == 0) a.lse[(size_t)head * a.rows_total + seq0 + qq] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
    }
    if (a.out8) q_amax_update(a.q8, omax, q8seen);
}

// ===================================================================================================
// LDS-DMA ring variants (round 5).  The kernels above stage a 64-row tile global -> VGPR -> ds_write_b128 with two barriers per tile.
// Here the tiles arrive by LDS-DMA (global_load_lds, 16 B per lane, no staging registers, no ds_write) into a ring of NS slots; ONE
// barrier per tile:
//     iteration t:  wait until this wave's DMAs of tile t have landed (counted vmcnt: the tiles behind it stay in flight)
//                   s_barrier                      every wave's share of tile t has landed AND every wave is done reading tile t - 1
//                   DMA of tile t + NS - 1         into the slot tile t - 1 occupied
//                   compute on tile t
// The LDS image of a tile is the one the register-staged kernels write (Img<HD>: 8-row x 32-column sub-tiles, chunk XOR (row>>2)&3), so
// the fragment addressing - and therefore every product, in the same order - is unchanged: results are BITWISE those of the kernels above
// (tests/test_kernels_gpu.py::test_attention_ring_kernels_match_the_register_staged_ones).  An LDS-DMA writes wave-uniform base + 16 x lane,
// so the image's permutation is applied to the per-lane SOURCE address: LDS chunk i of an image holds (row, chunk) = RingOff::rc(i).
// Every LDS read is inline asm: to hipcc an LDS-DMA in flight is an LDS store it cannot disambiguate, and a compiler-visible ds_read
// behind it gets s_waitcnt vmcnt(0) - the ring would drain at every tile (the same reason as in gemm.hip).  Completion of the reads is
// waited for by hand (lds_wait: s_waitcnt lgkmcnt(0) + sched_barrier, so no MFMA is hoisted above the wait).
#define GLOBAL_AS __attribute__((address_space(1)))

template <int N>
__device__ __forceinline__ void wait_vm() {          // counted wait on the vector-memory queue as a REAL s_waitcnt (see gemm.hip wait_vm)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void ring_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int OFF>
__device__ __forceinline__ void lds_r128(bf16x8& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_tr64(bf16x4& v, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// per-thread source offsets (bytes from the tile's first row) of the chunks this thread's DMA instructions fetch for one [64][HD] image
template <int HD, int NTH>
struct RingOff {
    static constexpr int NSUB = HD / 32;
    static constexpr int NCHUNK = 64 * HD / 8;
    static constexpr int PER = NCHUNK / NTH;
    static_assert(NCHUNK % NTH == 0 && (NSUB == 1 || NSUB == 2), "ring: HD 32 / 64");
    unsigned o[PER];
    // LDS chunk index i of the image -> (row, 16-byte chunk of the row) it holds: the inverse of Img<HD>::off / 16
    __device__ static __forceinline__ void rc(int i, int& row, int& ch) {
        const int blk = i / (32 * NSUB), within = i % (32 * NSUB);
        row = blk * 8 + ((within % 32) >> 2);
        ch = (within / 32) * 4 + ((i & 3) ^ ((row >> 2) & 3));
    }
    __device__ __forceinline__ void init(long long ld, int tid) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            int row, ch;
            rc(j * NTH + tid, row, ch);
            o[j] = ((unsigned)row * (unsigned)ld + (unsigned)ch * 8u) * 2u;
        }
    }
};

// request the [64][HD] image of rows row0 .. row0 + 63 of `src` (clamped to last_row: a sequence's last, partial tile) into `img`
template <int HD, int NTH>
__device__ __forceinline__ void ring_issue(const RingOff<HD, NTH>& ro, const bf16_t* src, long long ld, int row0, int last_row, char* img, int tid) {
    const char* base = reinterpret_cast<const char*>(src + (size_t)row0 * ld);        // wave-uniform
    const int wave = tid >> 6;
    unsigned off[RingOff<HD, NTH>::PER];
#pragma unroll
    for (int j = 0; j < RingOff<HD, NTH>::PER; ++j) off[j] = ro.o[j];
    if (row0 + 63 > last_row) {                                // block-uniform
        asm volatile("; partial tile: clamp rows" ::: "memory");
#pragma unroll
        for (int j = 0; j < RingOff<HD, NTH>::PER; ++j) {
            int row, ch;
            RingOff<HD, NTH>::rc(j * NTH + tid, row, ch);
            off[j] = ((unsigned)min(row, last_row - row0) * (unsigned)ld + (unsigned)ch * 8u) * 2u;
        }
    }
#pragma unroll
    for (int j = 0; j < RingOff<HD, NTH>::PER; ++j)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(base + off[j]), (LDS_AS void*)(img + (j * NTH + wave * 64) * 16), 16, 0, 0);
}

// lane-dependent parts of the fragment addresses inside an image (the row-block / sub-tile parts are immediates of the reads):
//   row fragment (A operand by rows), k-step kk, row block rb (32 rows):  rk[kk & 1] + 2048 NSUB rb + 512 (kk >> 1)
//   transposed fragment of rows 16 m .. 16 m + 15, 32-column block db:    lo: tv[0] + 1024 NSUB m + 512 db;  hi: tv[1] + 1024 NSUB m + 512 NSUB + 512 db
template <int HD>
struct FragAddr {
    unsigned rk[2], tv[2];
    __device__ __forceinline__ void init(int lane) {
        const int r = lane & 31, hh = lane >> 5;
        rk[0] = (unsigned)Img<HD>::off(r, hh);
        rk[1] = (unsigned)Img<HD>::off(r, 2 + hh);
        const int cb = (lane >> 4) & 1, li = lane & 15, q = li >> 2, p4 = li & 3;
        const int ch = 2 * cb + (p4 >> 1);
        tv[0] = (unsigned)(Img<HD>::off(4 * hh + q, ch) + 8 * (p4 & 1));
        tv[1] = (unsigned)(Img<HD>::off(8 + 4 * hh + q, ch) + 8 * (p4 & 1)) - 512u * (HD / 32);
    }
};

template <int HD, int RB, int KK>
__device__ __forceinline__ void ring_row_frag(bf16x8& v, unsigned img, const FragAddr<HD>& fa) {
    lds_r128<2048 * (HD / 32) * RB + 512 * (KK >> 1)>(v, img + fa.rk[KK & 1]);
}
template <int HD, int M16, int DB>
__device__ __forceinline__ void ring_tr_frag(bf16x4& lo, bf16x4& hi, unsigned img, const FragAddr<HD>& fa) {
    lds_tr64<1024 * (HD / 32) * M16 + 512 * DB>(lo, img + fa.tv[0]);
    lds_tr64<1024 * (HD / 32) * M16 + 512 * (HD / 32) + 512 * DB>(hi, img + fa.tv[1]);
}
__device__ __forceinline__ bf16x8 join8(const bf16x4& lo, const bf16x4& hi) { return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

template <int HD, int NW, int NS>
__global__ __launch_bounds__(64 * NW, (NW == 4 && HD == 32) ? 4 : (NW == 4 && HD == 64) ? 3 : 1) void attn_fwd_ring_kernel(AttnArgs a) {
    constexpr int NTH = 64 * NW;
    constexpr int NKK = HD / 16, NDB = HD / 32;
    constexpr int IMG = 64 * HD * 2, SLOT = 2 * IMG;        // a slot: the K image, then the V image of one 64-key tile
    constexpr int DPT = 2 * RingOff<HD, NTH>::PER;          // DMA instructions per thread and tile
    static_assert(NS >= 2 && NS <= 4 && (NS - 2) * DPT <= 63, "ring depth");
    __shared__ __attribute__((aligned(16))) char smem[NS * SLOT];
    const unsigned smem_lds = (unsigned)(size_t)(LDS_AS const char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const float q8s = a.out8 ? a.q8[AVS_Q_SCALE] : 0.f;
    const float q8seen = q_amax_peek(a.out8 ? a.q8 : nullptr);
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int head = lid / a.ntiles, tix = lid - head * a.ntiles;
    const int seq0 = a.tile_start[tix], L = a.tile_len[tix];
    const int qw = a.tile_q0[tix] + 32 * wave;          // first query of this wave
    const bool active = qw < L;
    const int q = min(qw + (lane & 31), L - 1);
    const bf16_t* base = a.qkv + (size_t)seq0 * a.ld + head * HD;

    RingOff<HD, NTH> ro;
    ro.init(a.ld, tid);
    const int ntile = (L + 63) >> 6;
    auto issue = [&](int t, int slot) {
        ring_issue<HD, NTH>(ro, base + a.D, a.ld, 64 * t, L - 1, smem + slot * SLOT, tid);
        ring_issue<HD, NTH>(ro, base + 2 * a.D, a.ld, 64 * t, L - 1, smem + slot * SLOT + IMG, tid);
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < ntile) issue(t, t);

    bf16x8 qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(base + (size_t)q * a.ld + (2 * kk + hh) * 8);
    FragAddr<HD> fa;
    fa.init(lane);

    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    constexpr float LAZY_SUM = 1099511627776.0f;         // (see attn_fwd_kernel)
    float m_run = 0.f, l_run = 0.f;
    f32x16 negm = splat16(0.f);

    int slot = 0;
    for (int t = 0; t < ntile; ++t) {
        // this wave's DMAs of tile t have landed; the tiles requested behind it (at most NS - 2, fewer at the end) stay in flight
        const int ahead = min(NS - 2, ntile - 1 - t);
        if (NS >= 4 && ahead == 2) wait_vm<2 * DPT>();
        else if (NS >= 3 && ahead >= 1) wait_vm<DPT>();
        else wait_vm<0>();
        ring_barrier();
        if (t + NS - 1 < ntile) issue(t + NS - 1, slot == 0 ? NS - 1 : slot - 1);
        const unsigned sK = smem_lds + slot * SLOT, sV = sK + IMG;
        slot = slot + 1 == NS ? 0 : slot + 1;
        if (!active) continue;
        const int k0 = 64 * t;
        bf16x8 kf[2][NKK];
        auto read_k = [&]() {
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                if (kk == 0) { ring_row_frag<HD, 0, 0>(kf[0][0], sK, fa); ring_row_frag<HD, 1, 0>(kf[1][0], sK, fa); }
                if (kk == 1) { ring_row_frag<HD, 0, 1>(kf[0][1], sK, fa); ring_row_frag<HD, 1, 1>(kf[1][1], sK, fa); }
                if constexpr (NKK > 2) {
                    if (kk == 2) { ring_row_frag<HD, 0, 2>(kf[0][2], sK, fa); ring_row_frag<HD, 1, 2>(kf[1][2], sK, fa); }
                    if (kk == 3) { ring_row_frag<HD, 0, 3>(kf[0][3], sK, fa); ring_row_frag<HD, 1, 3>(kf[1][3], sK, fa); }
                }
            }
            lds_wait();
        };
        f32x16 s[2];
        auto scores = [&]() {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][0], qf[0], negm, 0, 0, 0);
#pragma unroll
                for (int kk = 1; kk < NKK; ++kk) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][kk], qf[kk], s[kb], 0, 0, 0);
            }
        };
        read_k();
        scores();
        // the V fragments of the first 32 keys are requested now and land under the softmax arithmetic; those of the other 32 are requested
        // when the first have landed and land under the first half's products (one register set per half)
        bf16x4 vlo[2][2][NDB], vhi[2][2][NDB];
        ring_tr_frag<HD, 0, 0>(vlo[0][0][0], vhi[0][0][0], sV, fa); ring_tr_frag<HD, 1, 0>(vlo[0][1][0], vhi[0][1][0], sV, fa);
        if constexpr (NDB > 1) { ring_tr_frag<HD, 0, 1>(vlo[0][0][1], vhi[0][0][1], sV, fa); ring_tr_frag<HD, 1, 1>(vlo[0][1][1], vhi[0][1][1], sV, fa); }
        if (k0 + 64 > L) {
            asm volatile("; tail key tile: mask" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[kb][r] = -INFINITY;
        }
        if (k0 == 0) {
            float tm = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tm = fmaxf(tm, s[kb][r]);
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            m_run = tm;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] -= tm;
            negm = splat16(-m_run);
        }
        float psum;
        float p[2][16];
        auto exp_tile = [&]() {
            float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    p[kb][r] = fast_exp2(s[kb][r]);
                    p[kb][r + 1] = fast_exp2(s[kb][r + 1]);
                    ps0 = add1(ps0, p[kb][r]);
                    ps1 = add1(ps1, p[kb][r + 1]);
                }
            psum = ps0 + ps1;
        };
        exp_tile();
        if (k0 != 0 && __any(!(psum < LAZY_SUM))) {
            // wave-uniform and rare: move the reference point (see attn_fwd_kernel); the key fragments are read again (K is still in its slot)
            asm volatile("; softmax: new reference max" ::: "memory");
            lds_wait();                                    // (the V reads in flight: the key fragments are read again into the same registers)
            read_k();
            scores();
            float tm = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (k0 + 64 > L && k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[kb][r] = -INFINITY;
                    tm = fmaxf(tm, s[kb][r]);
                }
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            const float d = fmaxf(tm, 0.f);
            const float alpha = fast_exp2(-d);
            l_run *= alpha;
#pragma unroll
            for (int dd = 0; dd < NDB; ++dd)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
            m_run += d;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] -= d;
            negm = splat16(-m_run);
            exp_tile();
        }
        l_run += psum;
        bf16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < 2; ++st) pf[kb][st] = acc_frag(&p[kb][8 * st]);
        lds_wait();                                        // the V fragments of keys 0 - 31
        ring_tr_frag<HD, 2, 0>(vlo[1][0][0], vhi[1][0][0], sV, fa); ring_tr_frag<HD, 3, 0>(vlo[1][1][0], vhi[1][1][0], sV, fa);
        if constexpr (NDB > 1) { ring_tr_frag<HD, 2, 1>(vlo[1][0][1], vhi[1][0][1], sV, fa); ring_tr_frag<HD, 3, 1>(vlo[1][1][1], vhi[1][1][1], sV, fa); }
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int d = 0; d < NDB; ++d)
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(vlo[0][st][d], vhi[0][st][d]), pf[0][st], o[d], 0, 0, 0);
        lds_wait();                                        // keys 32 - 63
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int d = 0; d < NDB; ++d)
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(vlo[1][st][d], vhi[1][st][d]), pf[1][st], o[d], 0, 0, 0);
    }
    if (!active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qq = qw + (lane & 31);
    float omax = 0.f;
    if (qq < L) {
        bf16_t* orow = a.out + (size_t)(seq0 + qq) * a.ldo + head * HD;
        uint8_t* orow8 = a.out8 ? a.out8 + (size_t)(seq0 + qq) * a.ldo8 + head * HD : nullptr;
#pragma unroll
        for (int d = 0; d < NDB; ++d) store_block<HD, false>(orow, d, hh, o[d], inv, orow8, q8s, omax);
        if (hh == 0) a.lse[(size_t)head * a.rows_total + seq0 + qq] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
    }
    if (a.out8) q_amax_update(a.q8, omax, q8seen);
}

// ---------------------------------------------------------------------------------------------------
// dQ (query-major).  Also produces delta = rowsum(dO * O), reused by the dK/dV kernel.
// (hd 32, 4 waves: the second launch-bound argument - 4 waves per SIMD - holds the kernel to 128 registers; it needs 132 otherwise and
//  would run at three waves per SIMD)
// G8: the instantiation that also writes the e5m2 copy of its third of dqkv (fp8 backward).  A template parameter, not a run-time test: the
// extra epilogue costs the bf16 instantiations registers they do not have (hd 32: a spill under the 128-register bound; hd 64, 2 waves:
// 172 instead of 164, an occupancy step).
template <int HD, int NW, int HG = HD, bool G8 = false>
__global__ __launch_bounds__(64 * NW, (HD == 32 && NW == 4) ? 4 : (HD == 64 && NW == 2 && G8) ? 3 : 1) void attn_bwd_dq_kernel(AttnArgs a) {
    constexpr int NTH = 64 * NW;
    constexpr int NKK = HG / 16, NDB = HD / 32;          // contraction steps over the real head dim; 32-wide output blocks of the image
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * HD * 2];
    char* sK = smem;
    char* sV = smem + 64 * HD * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    // 1-D grid, XCD-aware: logical id = head * ntiles + tile, so the query/key tiles of one (sequence, head) - which all
    // stream the same K/V (or Q/dO) rows - run on one XCD and re-read them from its L2 instead of from HBM.
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int head = lid / a.ntiles, tix = lid - head * a.ntiles;
    const int seq0 = a.tile_start[tix], L = a.tile_len[tix];
    const int Lq = attn_lq(a, L);                       // (AttnArgs::lq)
    if (a.tile_q0[tix] >= Lq) return;                   // block-uniform, before any barrier
    const size_t orow0 = attn_orow0(a, seq0, L);
    const int qw = a.tile_q0[tix] + 32 * wave;
    const bool active = qw < Lq;
    const int q = min(qw + (lane & 31), Lq - 1);
    const bf16_t* base = a.qkv + (size_t)seq0 * a.ld + head * HG;

    bf16x8 qf[NKK], dof[NKK];
    float dpart = 0.f;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        qf[kk] = *reinterpret_cast<const bf16x8*>(base + (size_t)q * a.ld + (2 * kk + hh) * 8);
        dof[kk] = *reinterpret_cast<const bf16x8*>(a.dout + (orow0 + q) * a.ldo + head * HG + (2 * kk + hh) * 8);
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(a.out + (orow0 + q) * a.ldo + head * HG + (2 * kk + hh) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) dpart += bf2f((bf16_t)dof[kk][j]) * bf2f((bf16_t)of[j]);
    }
    const float delta = dpart + __shfl_xor(dpart, 32, 64);
    const float lse2 = a.lse[(size_t)head * a.rows_total + seq0 + q] * 1.4426950408889634f;
    if (active && hh == 0 && qw + (lane & 31) < Lq) a.delta[(size_t)head * a.rows_total + seq0 + q] = delta;
    // the query is on the lane, so -lse and -delta are per-lane constants: as the accumulators' initial values they make
    // the MFMAs deliver log2(p) and (dP - delta) with no VALU work.  Only for hd 32 (VALU-bound, registers to spare): at
    // hd 64 the two 16-register tuples cost an occupancy step (174 vs 142 VGPRs) that outweighs two VALU ops per score.
    constexpr bool C_INIT = HD == 32;
    f32x16 nlse, ndel;
    if (C_INIT) { nlse = splat16(-lse2); ndel = splat16(-delta); }

    f32x16 dq[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    TileRegs<HD, NTH> rk, rv;
    TileOff<HD, NTH> toff;
    toff.init(a.ld, tid);
    tile_load<HD, NTH, HG>(rk, toff, base + a.D, a.ld, 0, L - 1, tid);
    tile_load<HD, NTH, HG>(rv, toff, base + 2 * a.D, a.ld, 0, L - 1, tid);
    for (int k0 = 0; k0 < L; k0 += 64) {
        __syncthreads();
        tile_store<HD, NTH>(rk, sK, tid);
        tile_store<HD, NTH>(rv, sV, tid);
        __syncthreads();
        if (k0 + 64 < L) {
            tile_load<HD, NTH, HG>(rk, toff, base + a.D, a.ld, k0 + 64, L - 1, tid);
            tile_load<HD, NTH, HG>(rv, toff, base + 2 * a.D, a.ld, k0 + 64, L - 1, tid);
        }
        if (!active) continue;
        const bool tail_tile = k0 + 64 > L;                 // block-uniform
        auto key_block = [&](int kb) {
            f32x16 s, dp;
            if (C_INIT) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sK, kb * 32, 0, lane), qf[0], nlse, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sV, kb * 32, 0, lane), dof[0], ndel, 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sK, kb * 32, 0, lane), qf[0], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sV, kb * 32, 0, lane), dof[0], dp, 0, 0, 0);
            }
#pragma unroll
            for (int kk = 1; kk < NKK; ++kk) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sK, kb * 32, kk, lane), qf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sV, kb * 32, kk, lane), dof[kk], dp, 0, 0, 0);
            }
            if (tail_tile) {
                // real branch (the empty asm blocks if-conversion): only the last, partial key tile pays for masking
                asm volatile("; tail key tile: mask" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[r] = -INFINITY;     // -> p = exp2(-inf) = 0
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ds[r] = C_INIT ? mul1(fast_exp2(s[r]), dp[r]) : mul1(fast_exp2(sub1(s[r], lse2)), sub1(dp[r], delta));
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8 dsf = acc_frag(&ds[8 * st]);
#pragma unroll
                for (int d = 0; d < NDB; ++d)
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sK, kb * 32 + 16 * st, d, lane), dsf, dq[d], 0, 0, 0);
            }
        };
        key_block(0);
        key_block(1);
    }
    if (!active) return;
    // fp8 backward: the record's scale and the amax it holds are read HERE (scalar loads of a uniform address), not at kernel start: three
    // more live registers across the key loop spill in the 128-register hd-32 instantiation, which the bf16 path shares
    float g8s = 0.f, g8seen = 0.f, g8max = 0.f;
    if (G8) { g8s = a.qd8[AVS_Q_SCALE]; g8seen = a.qd8[AVS_Q_AMAX]; }
    const int qq = qw + (lane & 31);
    if (qq < Lq) {
        bf16_t* drow = a.dqkv + (size_t)(seq0 + qq) * a.ld + head * HG;
        uint8_t* drow8 = G8 ? a.dqkv8 + (size_t)(seq0 + qq) * a.ld8 + head * HG : nullptr;
#pragma unroll
        for (int d = 0; d < NDB; ++d) store_block<HG, true>(drow, d, hh, dq[d], a.scale, drow8, g8s, g8max);
    }
    if (G8) q_amax_update(a.qd8, g8max, g8seen);
}

// ---------------------------------------------------------------------------------------------------
// dQ with the K / V tiles by LDS-DMA ring (see attn_fwd_ring_kernel): same image, same fragments, same products in the same order as
// attn_bwd_dq_kernel.  Per 32-key block: K and V row fragments -> S and dP; the K fragments of the dQ product (transposed reads) are
// requested behind those MFMAs and land under the exponentials.
template <int HD, int NW, int NS, bool G8 = false>
__global__ __launch_bounds__(64 * NW, (HD == 32 && NW == 4) ? 4 : (HD == 64 && NW == 4) ? 3 : 1)
void attn_bwd_dq_ring_kernel(AttnArgs a) {
    constexpr int NTH = 64 * NW;
    constexpr int NKK = HD / 16, NDB = HD / 32;
    constexpr int IMG = 64 * HD * 2, SLOT = 2 * IMG;
    constexpr int DPT = 2 * RingOff<HD, NTH>::PER;
    static_assert(NS >= 2 && NS <= 4 && (NS - 2) * DPT <= 63, "ring depth");
    __shared__ __attribute__((aligned(16))) char smem[NS * SLOT];
    const unsigned smem_lds = (unsigned)(size_t)(LDS_AS const char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int head = lid / a.ntiles, tix = lid - head * a.ntiles;
    const int seq0 = a.tile_start[tix], L = a.tile_len[tix];
    const int qw = a.tile_q0[tix] + 32 * wave;
    const bool active = qw < L;
    const int q = min(qw + (lane & 31), L - 1);
    const bf16_t* base = a.qkv + (size_t)seq0 * a.ld + head * HD;

    RingOff<HD, NTH> ro;
    ro.init(a.ld, tid);
    const int ntile = (L + 63) >> 6;
    auto issue = [&](int t, int slot) {
        ring_issue<HD, NTH>(ro, base + a.D, a.ld, 64 * t, L - 1, smem + slot * SLOT, tid);
        ring_issue<HD, NTH>(ro, base + 2 * a.D, a.ld, 64 * t, L - 1, smem + slot * SLOT + IMG, tid);
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < ntile) issue(t, t);

    bf16x8 qf[NKK], dof[NKK];
    float dpart = 0.f;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        qf[kk] = *reinterpret_cast<const bf16x8*>(base + (size_t)q * a.ld + (2 * kk + hh) * 8);
        dof[kk] = *reinterpret_cast<const bf16x8*>(a.dout + (size_t)(seq0 + q) * a.ldo + head * HD + (2 * kk + hh) * 8);
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(a.out + (size_t)(seq0 + q) * a.ldo + head * HD + (2 * kk + hh) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) dpart += bf2f((bf16_t)dof[kk][j]) * bf2f((bf16_t)of[j]);
    }
    const float delta = dpart + __shfl_xor(dpart, 32, 64);
    const float lse2 = a.lse[(size_t)head * a.rows_total + seq0 + q] * 1.4426950408889634f;
    if (active && hh == 0 && qw + (lane & 31) < L) a.delta[(size_t)head * a.rows_total + seq0 + q] = delta;
    constexpr bool C_INIT = HD == 32;            // (see attn_bwd_dq_kernel)
    f32x16 nlse, ndel;
    if (C_INIT) { nlse = splat16(-lse2); ndel = splat16(-delta); }
    FragAddr<HD> fa;
    fa.init(lane);

    f32x16 dq[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    int slot = 0;
    for (int t = 0; t < ntile; ++t) {
        const int ahead = min(NS - 2, ntile - 1 - t);
        if (NS >= 4 && ahead == 2) wait_vm<2 * DPT>();
        else if (NS >= 3 && ahead >= 1) wait_vm<DPT>();
        else wait_vm<0>();
        ring_barrier();
        if (t + NS - 1 < ntile) issue(t + NS - 1, slot == 0 ? NS - 1 : slot - 1);
        const unsigned sK = smem_lds + slot * SLOT, sV = sK + IMG;
        slot = slot + 1 == NS ? 0 : slot + 1;
        if (!active) continue;
        const int k0 = 64 * t;
        const bool tail_tile = k0 + 64 > L;                 // block-uniform
        auto key_block = [&](auto KB) {
            constexpr int kb = decltype(KB)::value;
            bf16x8 kf[NKK], vf[NKK];
            ring_row_frag<HD, kb, 0>(kf[0], sK, fa); ring_row_frag<HD, kb, 0>(vf[0], sV, fa);
            ring_row_frag<HD, kb, 1>(kf[1], sK, fa); ring_row_frag<HD, kb, 1>(vf[1], sV, fa);
            if constexpr (NKK > 2) {
                ring_row_frag<HD, kb, 2>(kf[2], sK, fa); ring_row_frag<HD, kb, 2>(vf[2], sV, fa);
                ring_row_frag<HD, kb, 3>(kf[3], sK, fa); ring_row_frag<HD, kb, 3>(vf[3], sV, fa);
            }
            lds_wait();
            f32x16 s, dp;
            if (C_INIT) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], nlse, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], dof[0], ndel, 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], dof[0], dp, 0, 0, 0);
            }
#pragma unroll
            for (int kk = 1; kk < NKK; ++kk) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kk], dof[kk], dp, 0, 0, 0);
            }
            // the transposed key fragments of the dQ product land under the exponentials
            bf16x4 tlo[2][NDB], thi[2][NDB];
            ring_tr_frag<HD, 2 * kb, 0>(tlo[0][0], thi[0][0], sK, fa); ring_tr_frag<HD, 2 * kb + 1, 0>(tlo[1][0], thi[1][0], sK, fa);
            if constexpr (NDB > 1) { ring_tr_frag<HD, 2 * kb, 1>(tlo[0][1], thi[0][1], sK, fa); ring_tr_frag<HD, 2 * kb + 1, 1>(tlo[1][1], thi[1][1], sK, fa); }
            if (tail_tile) {
                asm volatile("; tail key tile: mask" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[r] = -INFINITY;     // -> p = exp2(-inf) = 0
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = C_INIT ? mul1(fast_exp2(s[r]), dp[r]) : mul1(fast_exp2(sub1(s[r], lse2)), sub1(dp[r], delta));
            bf16x8 dsf[2];
            dsf[0] = acc_frag(&ds[0]);
            dsf[1] = acc_frag(&ds[8]);
            lds_wait();
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int d = 0; d < NDB; ++d)
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join8(tlo[st][d], thi[st][d]), dsf[st], dq[d], 0, 0, 0);
        };
        key_block(std::integral_constant<int, 0>{});
        key_block(std::integral_constant<int, 1>{});
    }
    if (!active) return;
    float g8s = 0.f, g8seen = 0.f, g8max = 0.f;
    if (G8) { g8s = a.qd8[AVS_Q_SCALE]; g8seen = a.qd8[AVS_Q_AMAX]; }
    const int qq = qw + (lane & 31);
    if (qq < L) {
        bf16_t* drow = a.dqkv + (size_t)(seq0 + qq) * a.ld + head * HD;
        uint8_t* drow8 = G8 ? a.dqkv8 + (size_t)(seq0 + qq) * a.ld8 + head * HD : nullptr;
#pragma unroll
        for (int d = 0; d < NDB; ++d) store_block<HD, true>(drow, d, hh, dq[d], a.scale, drow8, g8s, g8max);
    }
    if (G8) q_amax_update(a.qd8, g8max, g8seen);
}

// ---------------------------------------------------------------------------------------------------
// dK, dV (key-major): each wave owns 32 keys (the lane) and walks the query rows of the sequence.
template <int HD, int NW, int HG = HD, bool G8 = false>
__global__ __launch_bounds__(64 * NW, (HD == 96 && NW == 2) ? 2 : 1) void attn_bwd_dkv_kernel(AttnArgs a) {
    constexpr int NTH = 64 * NW;
    constexpr int NKK = HG / 16, NDB = HD / 32;          // contraction steps over the real head dim; 32-wide output blocks of the image
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * HD * 2 + 2 * 64 * 4];
    char* sQ = smem;
    char* sDO = smem + 64 * HD * 2;
    float* sLse = reinterpret_cast<float*>(smem + 2 * 64 * HD * 2);
    float* sDel = sLse + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    // 1-D grid, XCD-aware: logical id = head * ntiles + tile, so the query/key tiles of one (sequence, head) - which all
    // stream the same K/V (or Q/dO) rows - run on one XCD and re-read them from its L2 instead of from HBM.
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int head = lid / a.ntiles, tix = lid - head * a.ntiles;
    const int seq0 = a.tile_start[tix], L = a.tile_len[tix];
    const int kw = a.tile_q0[tix] + 32 * wave;          // first key of this wave
    const bool active = kw < L;
    const int key = min(kw + (lane & 31), L - 1);
    const bf16_t* base = a.qkv + (size_t)seq0 * a.ld + head * HG;

    bf16x8 kf[NKK], vf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        kf[kk] = *reinterpret_cast<const bf16x8*>(base + a.D + (size_t)key * a.ld + (2 * kk + hh) * 8);
        vf[kk] = *reinterpret_cast<const bf16x8*>(base + 2 * a.D + (size_t)key * a.ld + (2 * kk + hh) * 8);
    }
    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }

    // the query side: the sequence's first Lq rows (AttnArgs::lq; normally all L), dO in the compact row numbering when lq is set
    const int Lq = attn_lq(a, L);
    const bf16_t* dobase = a.dout + attn_orow0(a, seq0, L) * a.ldo + head * HG;
    const float* lsebase = a.lse + (size_t)head * a.rows_total + seq0;
    const float* delbase = a.delta + (size_t)head * a.rows_total + seq0;
    TileRegs<HD, NTH> rq, rdo;
    TileOff<HD, NTH> toq, tod;                          // q rows have the qkv matrix's leading dimension, dO rows the output's
    toq.init(a.ld, tid);
    tod.init(a.ldo, tid);
    float rl = 0.f, rd = 0.f;
    tile_load<HD, NTH, HG>(rq, toq, base, a.ld, 0, Lq - 1, tid);
    tile_load<HD, NTH, HG>(rdo, tod, dobase, a.ldo, 0, Lq - 1, tid);
    if (tid < 64) { rl = lsebase[min(tid, Lq - 1)]; rd = delbase[min(tid, Lq - 1)]; }
    for (int q0 = 0; q0 < Lq; q0 += 64) {
        __syncthreads();
        tile_store<HD, NTH>(rq, sQ, tid);
        tile_store<HD, NTH>(rdo, sDO, tid);
        if (tid < 64) { sLse[tid] = -rl * 1.4426950408889634f; sDel[tid] = -rd; }    // negated: MFMA C operands
        __syncthreads();
        if (q0 + 64 < Lq) {
            tile_load<HD, NTH, HG>(rq, toq, base, a.ld, q0 + 64, Lq - 1, tid);
            tile_load<HD, NTH, HG>(rdo, tod, dobase, a.ldo, q0 + 64, Lq - 1, tid);
            if (tid < 64) { rl = lsebase[min(q0 + 64 + tid, Lq - 1)]; rd = delbase[min(q0 + 64 + tid, Lq - 1)]; }
        }
        if (!active) continue;
        const bool tail_tile = q0 + 64 > Lq;                 // block-uniform
        auto q_block = [&](int qb) {
            // the query is on the accumulator ROWS here: -lse / -delta of the 16 rows this lane holds come straight from LDS
            // into the C operands
            f32x16 s, dp;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rb = qb * 32 + 8 * t + 4 * hh;                 // rows rb..rb+3 of the staged tile
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + rb);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDel + rb);
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[4 * t + j] = l4[j]; dp[4 * t + j] = d4[j]; }
            }
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sQ, qb * 32, kk, lane), kf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sDO, qb * 32, kk, lane), vf[kk], dp, 0, 0, 0);
            }
            if (tail_tile) {
                // real branch (the empty asm blocks if-conversion): only the last, partial query tile pays for masking
                asm volatile("; tail query tile: mask" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (q0 + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= Lq) s[r] = -INFINITY;     // -> p = exp2(-inf) = 0
            }
            float p[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = fast_exp2(s[r]);
                ds[r] = p[r] * dp[r];
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8 pf = acc_frag(&p[8 * st]);
                const bf16x8 dsf = acc_frag(&ds[8 * st]);
#pragma unroll
                for (int d = 0; d < NDB; ++d) {
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sDO, qb * 32 + 16 * st, d, lane), pf, dv[d], 0, 0, 0);
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sQ, qb * 32 + 16 * st, d, lane), dsf, dk[d], 0, 0, 0);
                }
            }
        };
        q_block(0);
        q_block(1);
    }
    if (!active) return;
    // fp8 backward: the record's scale and the amax it holds are read HERE (scalar loads of a uniform address), not at kernel start: three
    // more live registers across the key loop spill in the 128-register hd-32 instantiation, which the bf16 path shares
    float g8s = 0.f, g8seen = 0.f, g8max = 0.f;
    if (G8) { g8s = a.qd8[AVS_Q_SCALE]; g8seen = a.qd8[AVS_Q_AMAX]; }
    const int kq = kw + (lane & 31);
    if (kq < L) {
        bf16_t* krow = (G8 && !a.kv16) ? nullptr : a.dqkv + (size_t)(seq0 + kq) * a.ld + a.D + head * HG;
        bf16_t* vrow = krow ? krow + a.D : nullptr;
        uint8_t* krow8 = G8 ? a.dqkv8 + (size_t)(seq0 + kq) * a.ld8 + a.D + head * HG : nullptr;
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            store_block<HG, true>(krow, d, hh, dk[d], LN2, krow8, g8s, g8max);      // dK = dS^T . (q * scale) and the staged q is q * scale * log2(e)
            store_block<HG, true>(vrow, d, hh, dv[d], 1.0f, G8 ? krow8 + a.D : nullptr, g8s, g8max);
        }
    }
    if (G8) q_amax_update(a.qd8, g8max, g8seen);
}

// ---------------------------------------------------------------------------------------------------
// Fused backward for sequences that fit ONE workgroup (L <= R = 32 * NW rows; the encoder's 39 - 128-token sequences): dQ, dK and dV
// from ONE read of q, k, v, dO, o and ONE evaluation of S and its exponentials - 20 MFMA products per 32 x 32 score tile instead of
// the 28 of the two-kernel form, half the v_exp, a third of the bytes, one launch.
//   phase 1 (as attn_bwd_dkv): wave w owns keys 32w.. (the lane), walks the query blocks; S and dP come out of the MFMAs with the
//     query on the accumulator rows and -lse / -delta as their initial values; dK^T, dV^T accumulate in registers.  The dS tile also
//     goes into an LDS image [key row][query column] (bf16, the rounding the dQ product applies anyway);
//   phase 2 (as attn_bwd_dq): wave w owns queries 32w..: dQ^T += K^T . dS^T with BOTH operands read transposed from LDS (K image
//     written from the key fragments the wave already holds, into the space of the Q image; the dS image written in phase 1).
// No atomics, no second pass; per (sequence, head) the arithmetic and its order equal the two-kernel form's.
// HG: the head dim in memory (80: ViT-H, in a 96-wide image whose columns 80.. are zero - cf. attn_fwd_kernel)
template <int HD, int NW, bool G8 = false, int HG = HD>
__global__ __launch_bounds__(64 * NW, (HD == 64 && NW == 2 && G8) ? 3 : 1) void attn_bwd_fused_kernel(AttnArgs a) {
    constexpr int R = 32 * NW;                            // rows (queries = keys) a workgroup holds
    constexpr int NKK = HG / 16, NDB = HD / 32;          // contraction steps over the real head dim; 32-wide output blocks of the image
    constexpr int IMG = R * HD * 2;                       // bytes of a [R][HD] bf16 image
    __shared__ __attribute__((aligned(16))) char smem[2 * IMG + R * R * 2 + 2 * R * 4];
    char* sQ = smem;                                      // phase 2: the K image
    char* sDO = smem + IMG;
    char* sDS = smem + 2 * IMG;                           // [key][query] bf16, Img<R>
    float* sLse = reinterpret_cast<float*>(smem + 2 * IMG + R * R * 2);
    float* sDel = sLse + R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const int head = blockIdx.x / a.ntiles, tix = blockIdx.x - head * a.ntiles;
    const int seq0 = a.tile_start[tix], L = a.tile_len[tix];
    const int rl = 32 * wave + (lane & 31);               // this lane's row of the images: its key (phase 1) and its query (phase 2)
    const int row = min(rl, L - 1);
    const bool active = 32 * wave < L;
    const bf16_t* qp = a.qkv + (size_t)(seq0 + row) * a.ld + head * HG;
    const bf16_t* dop = a.dout + (size_t)(seq0 + row) * a.ldo + head * HG;
    const bf16_t* op = a.out + (size_t)(seq0 + row) * a.ldo + head * HG;

    bf16x8 kf[NKK], vf[NKK];
    {
        bf16x8 qv[NKK], dov[NKK], ov[NKK];
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int c = (2 * kk + hh) * 8;
            qv[kk] = *reinterpret_cast<const bf16x8*>(qp + c);
            kf[kk] = *reinterpret_cast<const bf16x8*>(qp + a.D + c);
            vf[kk] = *reinterpret_cast<const bf16x8*>(qp + 2 * a.D + c);
            dov[kk] = *reinterpret_cast<const bf16x8*>(dop + c);
            ov[kk] = *reinterpret_cast<const bf16x8*>(op + c);
        }
        if (HG != HD) {                                    // the image's columns beyond the real head dim (the K image written over sQ below keeps them)
#pragma unroll
            for (int ch = HG / 8 + hh; ch < HD / 8; ch += 2) {
                *reinterpret_cast<f32x4*>(sQ + Img<HD>::off(rl, ch)) = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(sDO + Img<HD>::off(rl, ch)) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        float dpart = 0.f;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            *reinterpret_cast<bf16x8*>(sQ + Img<HD>::off(rl, 2 * kk + hh)) = qv[kk];
            *reinterpret_cast<bf16x8*>(sDO + Img<HD>::off(rl, 2 * kk + hh)) = dov[kk];
#pragma unroll
            for (int j = 0; j < 8; ++j) dpart += bf2f((bf16_t)dov[kk][j]) * bf2f((bf16_t)ov[kk][j]);
        }
        const float delta = dpart + __shfl_xor(dpart, 32, 64);
        if (hh == 0) {
            sLse[rl] = -a.lse[(size_t)head * a.rows_total + seq0 + row] * 1.4426950408889634f;       // negated: MFMA C operands
            sDel[rl] = -delta;
        }
    }
    __syncthreads();

    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }
    if (active) {
        const bool key_tail = 32 * wave + 32 > L;          // wave-uniform: some of this wave's keys lie beyond the sequence
        const bool key_dead = rl >= L;
        for (int qb = 0; qb * 32 < L; ++qb) {
            f32x16 s, dp;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rb = qb * 32 + 8 * t + 4 * hh;
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + rb);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDel + rb);
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[4 * t + j] = l4[j]; dp[4 * t + j] = d4[j]; }
            }
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sQ, qb * 32, kk, lane), kf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sDO, qb * 32, kk, lane), vf[kk], dp, 0, 0, 0);
            }
            if (qb * 32 + 32 > L || key_tail) {
                // real branch (the empty asm blocks if-conversion): only blocks touching the end of the sequence pay for masking
                asm volatile("; sequence end: mask" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key_dead || qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[r] = -INFINITY;      // -> p = exp2(-inf) = 0
            }
            float p[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = fast_exp2(s[r]);
                ds[r] = p[r] * dp[r];
            }
            // dS for phase 2: rows 8t + 4hh .. +3 of this query block are 4 consecutive columns of the lane's key row
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint2 w;
                w.x = pack_bf2(ds[4 * t + 0], ds[4 * t + 1]);
                w.y = pack_bf2(ds[4 * t + 2], ds[4 * t + 3]);
                *reinterpret_cast<uint2*>(sDS + Img<R>::off(rl, qb * 4 + t) + 8 * hh) = w;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8 pf = acc_frag(&p[8 * st]);
                const bf16x8 dsf = acc_frag(&ds[8 * st]);
#pragma unroll
                for (int d = 0; d < NDB; ++d) {
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sDO, qb * 32 + 16 * st, d, lane), pf, dv[d], 0, 0, 0);
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sQ, qb * 32 + 16 * st, d, lane), dsf, dk[d], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                                       // every dS tile is written, nobody reads the Q image any more
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) *reinterpret_cast<bf16x8*>(sQ + Img<HD>::off(rl, 2 * kk + hh)) = kf[kk];      // K image over the Q image
    __syncthreads();
    if (!active) return;

    f32x16 dq[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
    for (int ks = 0; ks * 16 < L; ++ks) {
        const bf16x8 dsf = tr_frag<R>(sDS, 16 * ks, wave, lane);        // B operand: k = keys 16ks.., n = this wave's queries
#pragma unroll
        for (int d = 0; d < NDB; ++d)
            dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sQ, 16 * ks, d, lane), dsf, dq[d], 0, 0, 0);
    }
    // fp8 backward: the record's scale and the amax it holds are read HERE (scalar loads of a uniform address), not at kernel start: three
    // more live registers across the key loop spill in the 128-register hd-32 instantiation, which the bf16 path shares
    float g8s = 0.f, g8seen = 0.f, g8max = 0.f;
    if (G8) { g8s = a.qd8[AVS_Q_SCALE]; g8seen = a.qd8[AVS_Q_AMAX]; }
    if (rl < L) {
        bf16_t* qrow = a.dqkv + (size_t)(seq0 + rl) * a.ld + head * HG;
        bf16_t* krow = (G8 && !a.kv16) ? nullptr : qrow + a.D;
        bf16_t* vrow = krow ? krow + a.D : nullptr;
        uint8_t* q8p = G8 ? a.dqkv8 + (size_t)(seq0 + rl) * a.ld8 + head * HG : nullptr;
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            store_block<HG, true>(qrow, d, hh, dq[d], a.scale, q8p, g8s, g8max);
            store_block<HG, true>(krow, d, hh, dk[d], LN2, G8 ? q8p + a.D : nullptr, g8s, g8max);      // dK = dS^T . (q * scale), the staged q is q * scale * log2(e)
            store_block<HG, true>(vrow, d, hh, dv[d], 1.0f, G8 ? q8p + 2 * a.D : nullptr, g8s, g8max);
        }
    }
    if (G8) q_amax_update(a.qd8, g8max, g8seen);
}

// ---------------------------------------------------------------------------------------------------
// The same fusion for sequences of 129 .. 224 tokens at head dim 64 (round 6: the encoder's 156 / 196-token frame sequences and 204-token audio
// sequences - two thirds of the contrastive pass's rows, which the two-kernel form read twice and exponentiated twice).  Seven waves; wave w
// owns keys 32w.. in phase 1 and queries 32w.. in phase 2, as above.  A [224][224] dS image would not fit beside the Q / dO / K images
// (3 x 28 KB), so the queries are taken in two HALVES of 128: phase 1 over the half's query blocks (dK^T / dV^T keep accumulating in
// registers) drops its dS tiles into a [224 keys][128 queries] image, phase 2 runs on the waves whose query block lies in that half
// (waves 0-3, then 4-6: every wave forms dQ for its 32 queries exactly once), then the image is reused.  K has an image of its own
// (the Q image is still read by the second half).  142 KB of LDS: one workgroup per CU.  Same products in the same order as the
// two-kernel form for every (sequence, head).
__global__ __launch_bounds__(448, 1) void attn_bwd_fused224_kernel(AttnArgs a) {
    constexpr int HD = 64, R = 224, QH = 128;
    constexpr int NKK = HD / 16, NDB = HD / 32;
    constexpr int IMG = R * HD * 2;                       // bytes of a [R][HD] bf16 image
    extern __shared__ __attribute__((aligned(16))) char smem_f224[];
    char* sQ = smem_f224;
    char* sDO = smem_f224 + IMG;
    char* sK = smem_f224 + 2 * IMG;
    char* sDS = smem_f224 + 3 * IMG;                      // [key][query of the half] bf16, Img<QH>
    float* sLse = reinterpret_cast<float*>(smem_f224 + 3 * IMG + R * QH * 2);
    float* sDel = sLse + R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const int head = blockIdx.x / a.ntiles, tix = blockIdx.x - head * a.ntiles;
    const int seq0 = a.tile_start[tix], L = a.tile_len[tix];
    const int rl = 32 * wave + (lane & 31);               // this lane's row of the images: its key (phase 1) and its query (phase 2)
    const int row = min(rl, L - 1);
    const bool active = 32 * wave < L;
    const bf16_t* qp = a.qkv + (size_t)(seq0 + row) * a.ld + head * HD;
    const bf16_t* dop = a.dout + (size_t)(seq0 + row) * a.ldo + head * HD;
    const bf16_t* op = a.out + (size_t)(seq0 + row) * a.ldo + head * HD;

    bf16x8 kf[NKK], vf[NKK];
    {
        bf16x8 qv[NKK], dov[NKK], ov[NKK];
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int c = (2 * kk + hh) * 8;
            qv[kk] = *reinterpret_cast<const bf16x8*>(qp + c);
            kf[kk] = *reinterpret_cast<const bf16x8*>(qp + a.D + c);
            vf[kk] = *reinterpret_cast<const bf16x8*>(qp + 2 * a.D + c);
            dov[kk] = *reinterpret_cast<const bf16x8*>(dop + c);
            ov[kk] = *reinterpret_cast<const bf16x8*>(op + c);
        }
        float dpart = 0.f;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            *reinterpret_cast<bf16x8*>(sQ + Img<HD>::off(rl, 2 * kk + hh)) = qv[kk];
            *reinterpret_cast<bf16x8*>(sDO + Img<HD>::off(rl, 2 * kk + hh)) = dov[kk];
            *reinterpret_cast<bf16x8*>(sK + Img<HD>::off(rl, 2 * kk + hh)) = kf[kk];
#pragma unroll
            for (int j = 0; j < 8; ++j) dpart += bf2f((bf16_t)dov[kk][j]) * bf2f((bf16_t)ov[kk][j]);
        }
        const float delta = dpart + __shfl_xor(dpart, 32, 64);
        if (hh == 0) {
            sLse[rl] = -a.lse[(size_t)head * a.rows_total + seq0 + row] * 1.4426950408889634f;       // negated: MFMA C operands
            sDel[rl] = -delta;
        }
    }
    __syncthreads();

    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }
    const bool key_tail = 32 * wave + 32 > L;              // wave-uniform: some of this wave's keys lie beyond the sequence
    const bool key_dead = rl >= L;
    for (int h = 0; h * QH < L; ++h) {                      // the queries, 128 at a time
        if (active) {
            for (int qb = 4 * h; qb < 4 * h + 4 && qb * 32 < L; ++qb) {
                f32x16 s, dp;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rb = qb * 32 + 8 * t + 4 * hh;
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + rb);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDel + rb);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s[4 * t + j] = l4[j]; dp[4 * t + j] = d4[j]; }
                }
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sQ, qb * 32, kk, lane), kf[kk], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sDO, qb * 32, kk, lane), vf[kk], dp, 0, 0, 0);
                }
                if (qb * 32 + 32 > L || key_tail) {
                    // real branch (the empty asm blocks if-conversion): only blocks touching the end of the sequence pay for masking
                    asm volatile("; sequence end: mask" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (key_dead || qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[r] = -INFINITY;      // -> p = exp2(-inf) = 0
                }
                float p[16], ds[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p[r] = fast_exp2(s[r]);
                    ds[r] = p[r] * dp[r];
                }
                // dS for phase 2: rows 8t + 4hh .. +3 of this query block are 4 consecutive columns of the lane's key row (columns of the HALF)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    uint2 w;
                    w.x = pack_bf2(ds[4 * t + 0], ds[4 * t + 1]);
                    w.y = pack_bf2(ds[4 * t + 2], ds[4 * t + 3]);
                    *reinterpret_cast<uint2*>(sDS + Img<QH>::off(rl, (qb - 4 * h) * 4 + t) + 8 * hh) = w;
                }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const bf16x8 pf = acc_frag(&p[8 * st]);
                    const bf16x8 dsf = acc_frag(&ds[8 * st]);
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {
                        dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sDO, qb * 32 + 16 * st, d, lane), pf, dv[d], 0, 0, 0);
                        dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sQ, qb * 32 + 16 * st, d, lane), dsf, dk[d], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                                   // the half's dS tiles are written
        if (active && (wave >> 2) == h) {                  // this wave's 32 queries lie in this half: dQ^T += K^T . dS^T over every key
            f32x16 dq[NDB];
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
            for (int ks = 0; ks * 16 < L; ++ks) {
                const bf16x8 dsf = tr_frag<QH>(sDS, 16 * ks, wave & 3, lane);       // B operand: k = keys 16ks.., n = this wave's queries
#pragma unroll
                for (int d = 0; d < NDB; ++d)
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sK, 16 * ks, d, lane), dsf, dq[d], 0, 0, 0);
            }
            if (rl < L) {
                bf16_t* qrow = a.dqkv + (size_t)(seq0 + rl) * a.ld + head * HD;
                float unused = 0.f;
#pragma unroll
                for (int d = 0; d < NDB; ++d) store_block<HD, true>(qrow, d, hh, dq[d], a.scale, nullptr, 0.f, unused);
            }
        }
        __syncthreads();                                   // ... and read: the next half may overwrite the image
    }
    if (!active || rl >= L) return;
    bf16_t* krow = a.dqkv + (size_t)(seq0 + rl) * a.ld + a.D + head * HD;
    float unused = 0.f;
#pragma unroll
    for (int d = 0; d < NDB; ++d) {
        store_block<HD, true>(krow, d, hh, dk[d], LN2, nullptr, 0.f, unused);      // dK = dS^T . (q * scale), the staged q is q * scale * log2(e)
        store_block<HD, true>(krow + a.D, d, hh, dv[d], 1.0f, nullptr, 0.f, unused);
    }
}

// ===================================================================================================
static int check_common(const char* name, const void* qkv, long long ld, int D, int H, int hd, const int* ts, const int* tl,
                        const int* tq, int ntiles) {
    if (!(qkv && ts && tl && tq && ntiles > 0 && H > 0 && (hd == 32 || hd == 64 || hd == 80) && D == H * hd && ld >= 3LL * D && (ld % 8) == 0)) {
        avs_set_error("%s: bad arguments (D=%d H=%d hd=%d ld=%lld ntiles=%d)", name, D, H, hd, ld, ntiles);
        return -2;
    }
    return 0;
}

// out8 / ldo8 / q8 (all or none): also write e4m3(clamp(out * q8[0], +-448)) - the fp8 operand of the proj GEMM - and fold max |out| into q8[2]
static int attn_fwd_impl(const bf16_t* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                         const int* tile_q0, int ntiles, int tile_rows, bf16_t* out, long long ldo, float* lse, int rows_total,
                         uint8_t* out8, long long ldo8, float* q8, int lq, hipStream_t stream) {
    AVS_CHECK_ARG(lq >= 0, "attn_fwd: lq < 0");
    AVS_CHECK_ARG((out8 == nullptr) == (q8 == nullptr) && (!out8 || (ldo8 >= D && (ldo8 % 16) == 0)), "attn_fwd: out8 and q8 go together, ldo8 %% 16 == 0");
    AVS_CHECK_ARG(tile_rows == 128 || tile_rows == 64, "attn_fwd: tile_rows must be 64 or 128");
    const int hd = H > 0 ? D / H : 0;
    if (int e = check_common("attn_fwd", qkv, ld, D, H, hd, tile_start, tile_len, tile_q0, ntiles)) return e;
    AVS_CHECK_ARG(out && lse, "attn_fwd: null output");
    AVS_CHECK_ARG((ldo % 8) == 0 && ldo >= D, "attn_fwd: ldo=%lld must be a multiple of 8 (the epilogue stores 16 bytes per lane) and >= D", ldo);
    AttnArgs a{qkv, ld, D, tile_start, tile_len, tile_q0, ntiles, out, ldo, lse, rows_total, nullptr, nullptr, nullptr, 1.0f / sqrtf((float)hd),
               out8, ldo8, q8, nullptr, 0, nullptr, 1, lq};
    dim3 grid(ntiles * H);
    // hd 80 (ViT-H: 1280 / 16 heads): a 96-wide LDS image whose last 16 columns are zero, five contraction steps
    const bool ring = avs_tuning().attn_ring != 0 && lq == 0;          // K / V tiles by LDS-DMA ring (hd 32 / 64; bitwise the register-staged kernels' results)
    if (hd == 80) { if (tile_rows == 128) attn_fwd_kernel<96, 4, 80><<<grid, 256, 0, stream>>>(a); else attn_fwd_kernel<96, 2, 80><<<grid, 128, 0, stream>>>(a); }
    else if (tile_rows == 128) {
        if (hd == 64) { if (ring) attn_fwd_ring_kernel<64, 4, 3><<<grid, 256, 0, stream>>>(a); else attn_fwd_kernel<64, 4><<<grid, 256, 0, stream>>>(a); }
        else { if (ring) attn_fwd_ring_kernel<32, 4, 3><<<grid, 256, 0, stream>>>(a); else attn_fwd_kernel<32, 4><<<grid, 256, 0, stream>>>(a); }
    } else {
        if (hd == 64) { if (ring) attn_fwd_ring_kernel<64, 2, 3><<<grid, 128, 0, stream>>>(a); else attn_fwd_kernel<64, 2><<<grid, 128, 0, stream>>>(a); }
        else { if (ring) attn_fwd_ring_kernel<32, 2, 3><<<grid, 128, 0, stream>>>(a); else attn_fwd_kernel<32, 2><<<grid, 128, 0, stream>>>(a); }
    }
    AVS_LAUNCH_CHECK("attn_fwd");
    return 0;
}

extern "C" int avs_attn_fwd_q8(const bf16_t* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                               const int* tile_q0, int ntiles, int tile_rows, bf16_t* out, long long ldo, float* lse, int rows_total,
                               uint8_t* out8, long long ldo8, float* q8, hipStream_t stream) {
    return attn_fwd_impl(qkv, ld, D, H, tile_start, tile_len, tile_q0, ntiles, tile_rows, out, ldo, lse, rows_total, out8, ldo8, q8, 0, stream);
}

// The same with only the first `lq` rows of every sequence as QUERIES (all rows are keys / values) and a COMPACT output: sequence s owns rows
// s * lq .. of `out`; every sequence of the launch must have the same length (tile_len), lse keeps the packed numbering (AttnArgs::lq).
extern "C" int avs_attn_fwd_cq(const bf16_t* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                               const int* tile_q0, int ntiles, int tile_rows, bf16_t* out, long long ldo, float* lse, int rows_total,
                               int lq, hipStream_t stream) {
    AVS_CHECK_ARG(lq > 0, "attn_fwd_cq: lq must be positive");
    return attn_fwd_impl(qkv, ld, D, H, tile_start, tile_len, tile_q0, ntiles, tile_rows, out, ldo, lse, rows_total, nullptr, 0, nullptr, lq, stream);
}

extern "C" int avs_attn_fwd(const bf16_t* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                            const int* tile_q0, int ntiles, int tile_rows, bf16_t* out, long long ldo, float* lse, int rows_total,
                            hipStream_t stream) {
    return avs_attn_fwd_q8(qkv, ld, D, H, tile_start, tile_len, tile_q0, ntiles, tile_rows, out, ldo, lse, rows_total, nullptr, 0, nullptr, stream);
}

// dqkv8 / ld8 / qd8 (all or none): also write e5m2(clamp(dqkv * qd8[0], +-57344)) [rows, 3*D] / ld8 - the gradient operand of the fp8 qkv
// input-gradient GEMM - and fold max |dqkv| into the device record qd8
static int attn_bwd_impl(const bf16_t* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                         const int* tile_q0, int ntiles, int tile_rows, const bf16_t* out, const bf16_t* dout, long long ldo,
                         const float* lse, float* delta, int rows_total, bf16_t* dqkv, uint8_t* dqkv8, long long ld8, float* qd8,
                         int kv_bf16, int lq, hipStream_t stream) {
    AVS_CHECK_ARG(lq >= 0, "attn_bwd: lq < 0");
    // kv_bf16 == 0 (with dqkv8 only): the key and value thirds of the bf16 dqkv are left unwritten - their only readers take the e5m2 copy
    AVS_CHECK_ARG(kv_bf16 || dqkv8, "attn_bwd: kv_bf16 = 0 needs the e5m2 copy");
    AVS_CHECK_ARG((dqkv8 == nullptr) == (qd8 == nullptr) && (!dqkv8 || (ld8 >= 3LL * D && (ld8 % 16) == 0)), "attn_bwd: dqkv8 and its record go together, ld8 %% 16 == 0");
    AVS_CHECK_ARG(tile_rows == 128 || tile_rows == 64, "attn_bwd: tile_rows must be 64 or 128");
    const int hd = H > 0 ? D / H : 0;
    if (int e = check_common("attn_bwd", qkv, ld, D, H, hd, tile_start, tile_len, tile_q0, ntiles)) return e;
    AVS_CHECK_ARG(out && dout && lse && delta && dqkv, "attn_bwd: null pointer");
    AVS_CHECK_ARG((ldo % 8) == 0 && ldo >= D, "attn_bwd: ldo=%lld must be a multiple of 8 (16-byte fragment loads of out / dO rows) and >= D", ldo);
    AttnArgs a{qkv, ld, D, tile_start, tile_len, tile_q0, ntiles, const_cast<bf16_t*>(out), ldo, const_cast<float*>(lse), rows_total,
               dout, delta, dqkv, 1.0f / sqrtf((float)hd), nullptr, 0, nullptr, dqkv8, ld8, qd8, kv_bf16, lq};
    dim3 grid(ntiles * H);
#define ATTN_BWD2(K, G)                                                                                     \
    do {                                                                                                    \
        if (hd == 80) {                                                                                     \
            if (tile_rows == 128) K<96, 4, 80, G><<<grid, 256, 0, stream>>>(a);                             \
            else K<96, 2, 80, G><<<grid, 128, 0, stream>>>(a);                                              \
        } else if (tile_rows == 128) {                                                                     \
            if (hd == 64) K<64, 4, 64, G><<<grid, 256, 0, stream>>>(a);                                     \
            else K<32, 4, 32, G><<<grid, 256, 0, stream>>>(a);                                              \
        } else {                                                                                            \
            if (hd == 64) K<64, 2, 64, G><<<grid, 128, 0, stream>>>(a);                                     \
            else K<32, 2, 32, G><<<grid, 128, 0, stream>>>(a);                                              \
        }                                                                                                   \
    } while (0)
    const bool ring = avs_tuning().attn_ring != 0 && hd != 80 && lq == 0;
#define ATTN_BWD2R(K, G)                                                                                    \
    do {                                                                                                    \
        if (tile_rows == 128) {                                                                             \
            if (hd == 64) K<64, 4, 3, G><<<grid, 256, 0, stream>>>(a);                                      \
            else K<32, 4, 3, G><<<grid, 256, 0, stream>>>(a);                                               \
        } else {                                                                                            \
            if (hd == 64) K<64, 2, 3, G><<<grid, 128, 0, stream>>>(a);                                      \
            else K<32, 2, 3, G><<<grid, 128, 0, stream>>>(a);                                               \
        }                                                                                                   \
    } while (0)
    if (ring) { if (dqkv8) ATTN_BWD2R(attn_bwd_dq_ring_kernel, true); else ATTN_BWD2R(attn_bwd_dq_ring_kernel, false); }
    else if (dqkv8) ATTN_BWD2(attn_bwd_dq_kernel, true); else ATTN_BWD2(attn_bwd_dq_kernel, false);
#undef ATTN_BWD2R
    AVS_LAUNCH_CHECK("attn_bwd_dq");
    if (dqkv8) ATTN_BWD2(attn_bwd_dkv_kernel, true); else ATTN_BWD2(attn_bwd_dkv_kernel, false);
#undef ATTN_BWD2
    AVS_LAUNCH_CHECK("attn_bwd_dkv");
    return 0;
}

extern "C" int avs_attn_bwd_q8(const bf16_t* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                               const int* tile_q0, int ntiles, int tile_rows, const bf16_t* out, const bf16_t* dout, long long ldo,
                               const float* lse, float* delta, int rows_total, bf16_t* dqkv, uint8_t* dqkv8, long long ld8, float* qd8,
                               int kv_bf16, hipStream_t stream) {
    return attn_bwd_impl(qkv, ld, D, H, tile_start, tile_len, tile_q0, ntiles, tile_rows, out, dout, ldo, lse, delta, rows_total, dqkv, dqkv8, ld8, qd8, kv_bf16, 0, stream);
}

// The backward of avs_attn_fwd_cq: `out` / `dout` compact (sequence s owns rows s * lq ..), dq is written for the first lq rows of every
// sequence (the query third of the OTHER rows of dqkv is left untouched: the caller zeroes it), dk / dv for all rows.
extern "C" int avs_attn_bwd_cq(const bf16_t* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                               const int* tile_q0, int ntiles, int tile_rows, const bf16_t* out, const bf16_t* dout, long long ldo,
                               const float* lse, float* delta, int rows_total, bf16_t* dqkv, int lq, hipStream_t stream) {
    AVS_CHECK_ARG(lq > 0, "attn_bwd_cq: lq must be positive");
    return attn_bwd_impl(qkv, ld, D, H, tile_start, tile_len, tile_q0, ntiles, tile_rows, out, dout, ldo, lse, delta, rows_total, dqkv, nullptr, 0, nullptr, 1, lq, stream);
}

extern "C" int avs_attn_bwd(const bf16_t* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                            const int* tile_q0, int ntiles, int tile_rows, const bf16_t* out, const bf16_t* dout, long long ldo,
                            const float* lse, float* delta, int rows_total, bf16_t* dqkv, hipStream_t stream) {
    return avs_attn_bwd_q8(qkv, ld, D, H, tile_start, tile_len, tile_q0, ntiles, tile_rows, out, dout, ldo, lse, delta, rows_total, dqkv, nullptr, 0,
                           nullptr, 1, stream);
}

// One workgroup per (sequence, head) for sequences of at most `rows_per_wg` (64 or 128) tokens: seq_start / seq_len [nseq].
extern "C" int avs_attn_bwd_fused_q8(const bf16_t* qkv, long long ld, int D, int H, const int* seq_start, const int* seq_len, int nseq,
                                     int rows_per_wg, const bf16_t* out, const bf16_t* dout, long long ldo, const float* lse, int rows_total,
                                     bf16_t* dqkv, uint8_t* dqkv8, long long ld8, float* qd8, int kv_bf16, hipStream_t stream) {
    AVS_CHECK_ARG(kv_bf16 || dqkv8, "attn_bwd_fused: kv_bf16 = 0 needs the e5m2 copy");
    AVS_CHECK_ARG((dqkv8 == nullptr) == (qd8 == nullptr) && (!dqkv8 || (ld8 >= 3LL * D && (ld8 % 16) == 0)), "attn_bwd_fused: dqkv8 and its record go together, ld8 %% 16 == 0");
    AVS_CHECK_ARG(rows_per_wg == 64 || rows_per_wg == 128 || rows_per_wg == 224, "attn_bwd_fused: rows_per_wg must be 64, 128 or 224");
    const int hd = H > 0 ? D / H : 0;
    AVS_CHECK_ARG(qkv && seq_start && seq_len && nseq > 0 && H > 0 && (hd == 32 || hd == 64 || hd == 80) && D == H * hd && ld >= 3LL * D && (ld % 8) == 0,
                  "attn_bwd_fused: bad arguments (D=%d H=%d hd=%d ld=%lld nseq=%d)", D, H, hd, ld, nseq);
    AVS_CHECK_ARG(out && dout && lse && dqkv && (ldo % 8) == 0, "attn_bwd_fused: null pointer");
    AVS_CHECK_ARG(rows_per_wg != 224 || (hd == 64 && !dqkv8), "attn_bwd_fused: 224-row workgroups exist for head dim 64 without the e5m2 copy");
    AVS_CHECK_ARG(hd != 80 || rows_per_wg == 64, "attn_bwd_fused: head dim 80 runs with 64-row workgroups only (a 128-row one needs 81 KB of LDS)");
    AttnArgs a{qkv, ld, D, seq_start, seq_len, nullptr, nseq, const_cast<bf16_t*>(out), ldo, const_cast<float*>(lse), rows_total,
               dout, nullptr, dqkv, 1.0f / sqrtf((float)hd), nullptr, 0, nullptr, dqkv8, ld8, qd8, kv_bf16, 0};
    dim3 grid(nseq * H);
    if (rows_per_wg == 224) {
        constexpr int SMEM224 = 3 * 224 * 64 * 2 + 224 * 128 * 2 + 2 * 224 * 4;
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute((const void*)attn_bwd_fused224_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM224) != hipSuccess) {
                avs_set_error("attn_bwd_fused: cannot reserve %d bytes of LDS", SMEM224);
                return -1;
            }
            attr_done = true;
        }
        attn_bwd_fused224_kernel<<<grid, 448, SMEM224, stream>>>(a);
        AVS_LAUNCH_CHECK("attn_bwd_fused224");
        return 0;
    }
#define ATTN_BWDF(G)                                                                                        \
    do {                                                                                                    \
        if (hd == 80) attn_bwd_fused_kernel<96, 2, G, 80><<<grid, 128, 0, stream>>>(a);                     \
        else if (rows_per_wg == 128) {                                                                         \
            if (hd == 64) attn_bwd_fused_kernel<64, 4, G><<<grid, 256, 0, stream>>>(a);                     \
            else attn_bwd_fused_kernel<32, 4, G><<<grid, 256, 0, stream>>>(a);                              \
        } else {                                                                                            \
            if (hd == 64) attn_bwd_fused_kernel<64, 2, G><<<grid, 128, 0, stream>>>(a);                     \
            else attn_bwd_fused_kernel<32, 2, G><<<grid, 128, 0, stream>>>(a);                              \
        }                                                                                                   \
    } while (0)
    if (dqkv8) ATTN_BWDF(true); else ATTN_BWDF(false);
#undef ATTN_BWDF
    AVS_LAUNCH_CHECK("attn_bwd_fused");
    return 0;
}

extern "C" int avs_attn_bwd_fused(const bf16_t* qkv, long long ld, int D, int H, const int* seq_start, const int* seq_len, int nseq,
                                  int rows_per_wg, const bf16_t* out, const bf16_t* dout, long long ldo, const float* lse, int rows_total,
                                  bf16_t* dqkv, hipStream_t stream) {
    return avs_attn_bwd_fused_q8(qkv, ld, D, H, seq_start, seq_len, nseq, rows_per_wg, out, dout, ldo, lse, rows_total, dqkv, nullptr, 0, nullptr, 1, stream);
}
