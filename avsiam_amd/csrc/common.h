// Shared device helpers for the avsiam gfx950 kernels (CDNA4 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint16_t bf16_t;                                   // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;  // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define AVS_WAVE 64

extern "C" void avs_set_error(const char* fmt, ...);

// host-side descriptor of a raw input (the C ABI's `avs_input_xf`, include/avsiam_hip.h)
typedef struct avs_input_xf_t {
    int kind;                        // 0 none | 1 un-normalised fp32 fbank | 2 uint8 frames
    float mean[3], std[3];           // kind 1: [0] only (dataset mean / std); kind 2: per channel
    const int* shift;                // kind 1, device, per sample, may be NULL: time roll (dataloader.py:513)
    const float* amp;                // kind 1, device, per sample, may be NULL: noise amplitude (:512)
    unsigned long long seed;         // Philox key of the noise
} avs_input_xf_t;

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

// The same sum with DPP adds instead of six ds_bpermute round trips through the LDS pipe (~60 cycles of dependent latency each):
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
    return v + __builtin_bit_cast(float, t);
}

// sum over the 64 lanes, returned wave-uniform (every lane gets it): quad swaps, half-row / row mirrors, then the row totals walk down
// the rows (row_bcast15 / 31) and lane 63 holds the wave's sum (checked on hardware against 1 + 2 + ... + 64)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = dpp_add<0xB1, 0xF>(v);                      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);                      // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);                     // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);                     // row_mirror: every lane of a row of 16 holds the row's sum
    v = dpp_add<0x142, 0xA>(v);                     // row_bcast15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);                     // row_bcast31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
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

// ---- process-wide tuning knobs (api.cpp).  Set ONLY through the C ABI (avs_tuning_set and the avs_gemm_set_* wrappers) - the host
// binding reads the AVSIAM_* environment once at load and calls them; no launcher reads the environment or initialises anything
// lazily.  Launchers only READ this struct (plain ints, written before any kernel is queued).
struct AvsTuning {
    int gemm_tile;         // 0 auto | 128 | 256: force the nt / tn GEMM tile (tuning + tests)                       AVSIAM_GEMM_TILE
    int gemm_persistent;   // 1: 256^2 nt tiles run as persistent workgroups (0: one workgroup per tile, A/B tests)
    int gemm_nt8;          // 1: 256^2 GEMMs run the 8-phase kernels                                                 AVSIAM_GEMM_NT8
    int nt_tile_h;         // 8-phase nt kernel tile heights: 0 automatic | 256 | 224 | 240 (half and half)          AVSIAM_NT_TILE_H
    int nt_grid;           // > 0: cap of the persistent nt grid (tools/bench_stagger.py)                            AVSIAM_NT_GRID
    int cu_reserve;        // CUs every PERSISTENT kernel (gemm_nt8 / gemm_nt 256^2, gemm_tn8, gemm_tn8f, fp8 nt) leaves free: their grids
                           // and split factors are sized for (CUs - cu_reserve), so that a collective's kernels (RCCL) find CUs WHILE a
                           // GEMM runs instead of only at kernel boundaries.  0 on one GPU; the host sets 8 when world > 1 and the
                           // gradient all-reduce overlaps the backward (comm.py)                                    AVSIAM_CU_RESERVE
    int ln_dma;            // 1: LayerNorm backward by the LDS-DMA kernel where it applies | 0 never                 AVSIAM_LN_DMA
    int ln_rpw;            // rows per wave of the LayerNorm backward: 0 automatic | 4 | 8 | 16                      AVSIAM_LN_RPW
    int gemm_ring;         // small forward / input-gradient GEMMs (128 x 128 tiling, at most one workgroup per CU): 0 the two-buffer kernel | 1 the
                           // 4-slot LDS-DMA ring kernel (three K-slabs in flight) | 2 (default) also: under half the CUs -> every row as 64 x 128
                           // half-height tiles (twice the workgroups).  Bitwise the same results in every setting          AVSIAM_GEMM_RING
    int attn_ring;         // 1: attention forward / dQ with the K/V tiles by LDS-DMA ring (hd 32 / 64) | 0 (default): register-staged kernels -
                           // bitwise the same results, measured neutral to slower                     AVSIAM_ATTN_RING
    int nt_big_min;        // forward / input-gradient GEMMs with at least this many 256^2 output tiles run the persistent 256^2 kernels, smaller
                           // ones the 128 x 128 kernels; 0 (default): half the persistent slots                      AVSIAM_NT_BIG_MIN
    int det;               // 1: every reduction into a parameter gradient has ONE writer per element and a fixed order - the weight-gradient GEMMs do
                           // not split their token rows (one workgroup per output tile), column sums / the vector-matrix product / the LayerNorm slab
                           // reduce run as one block per column group, the positional-embedding scatter and the un-shuffle's token sums take their
                           // atomics-free forms: two runs of a step give the same bits (debugging; slower).  0 (default)        AVSIAM_DET
};
AvsTuning& avs_tuning();
extern "C" int avs_tuning_set(const char* name, int value);
extern "C" int avs_device_cu_count(void);
// compute units a persistent kernel may fill: the device's CU count minus cu_reserve (never below 8)
int avs_persistent_slots();

// ---- fp8 (e4m3) per-tensor quantisation record, on the DEVICE (engine.FP8; delayed scaling): q[0] = scale (x -> x * scale -> e4m3),
// q[1] = 1 / scale, q[2] = running max |x| of the values quantised since the last avs_fp8_scale_update, q[3] = number of updates that
// found q[2] * scale > 448 (the tensor saturated under the scale it was quantised with).  Producers of an e4m3 operand read q[0]
// and fold what they saw into q[2]; nothing on this path ever synchronises with the host.
#define AVS_Q_SCALE 0
#define AVS_Q_INV 1
#define AVS_Q_AMAX 2
#define AVS_Q_SAT 3
// A record is AVS_Q_STRIDE floats: the four above (one 256-byte line) and AVS_Q_NSHARD shards of the running amax.  Producers fold their |max| into ONE shard,
// picked by workgroup (tens of thousands of waves adding to a single address serialise at the memory side: the LayerNorm forward took
// 5x its time that way when most rows carry a value near the tensor's maximum); avs_fp8_scale_update takes the max over [2] and the shards.
// Each shard sits in a 256-byte line of its own (same-line atomics still serialise at one memory channel).
#define AVS_Q_NSHARD 15
#define AVS_Q_SHARD_STRIDE 64                                   // floats between shards
#define AVS_Q_SHARD0 64                                         // first shard: the line after the header
#define AVS_Q_STRIDE (64 * (1 + AVS_Q_NSHARD))

// fold a wave's max |x| (m >= 0, any lane's value; reduced here) into q[AVS_Q_AMAX].  `seen` = q[AVS_Q_AMAX] as read by q_amax_peek at the
// START of the kernel (a plain load whose latency hides under the kernel's work; a load here, at the end, was a dependent memory round
// trip in every wave's tail - it doubled the LayerNorm forward's time).  It only filters: once the record holds the tensor's typical
// maximum almost no wave issues the atomic (thousands of waves hitting one address would serialise); a stale value costs a few atomics.
__device__ __forceinline__ float q_amax_peek(const float* q) { return q ? q[AVS_Q_AMAX] : 0.f; }

__device__ __forceinline__ void q_amax_update(float* q, float m, float seen) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > seen)      // non-negative floats order like their bits
        atomicMax(reinterpret_cast<int*>(q + AVS_Q_SHARD0 + ((blockIdx.x + (threadIdx.x >> 6)) % AVS_Q_NSHARD) * AVS_Q_SHARD_STRIDE), __float_as_int(m));
}

// ---- raw inputs: the arithmetic of the reference's dataset (/root/reference/src/dataloader.py:505-513
// audio, :461-462 + :152-155 frames) applied WHERE THE INPUT IS READ - the patch gather of the embedding and the target gather of
// the reconstruction loss - instead of in a pass of its own.  Mirrors `avs_input_xf` of include/avsiam_hip.h.
//   kind 0  the tensor is already normalised fp32 (the reference's forward() contract)
//   kind 1  audio: un-normalised fp32 fbank; value(b, t, f) = (in[b, (t - shift_b) mod T, f] - mean0) * inv_std0 + amp_b * U(b, ts, f)
//           (shift / amp: per-sample arrays or NULL; U: Philox4x32-10 keyed by `seed`, counter (ts * F + f, b) - the same stream
//           avs_normalize_audio draws, so the fused and the two-pass paths agree bit for bit)
//   kind 2  frames: uint8; value(n, c, y, x) = (in / 255 - mean_c) * inv_std_c
struct InXf {
    int kind;
    float mean[3], inv_std[3];
    const int* shift;
    const float* amp;
    uint32_t seed_lo, seed_hi;
};

__device__ __forceinline__ uint32_t xf_philox(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1) {
    uint32_t c2 = 0, c3 = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

// audio element (sample b, time frame t, mel bin f) of a [B, T, F] spectrogram as the model sees it
__device__ __forceinline__ float xf_audio(const float* __restrict__ a, const InXf& x, int b, int t, int f, int T, int F) {
    if (x.kind == 0) return a[((size_t)b * T + t) * F + f];
    int sh = x.shift ? x.shift[b] % T : 0;
    if (sh < 0) sh += T;
    int ts = t - sh;
    if (ts < 0) ts += T;
    float v = (a[((size_t)b * T + ts) * F + f] - x.mean[0]) * x.inv_std[0];
    const float amp = x.amp ? x.amp[b] : 0.f;
    if (amp != 0.f) v += amp * ((float)(xf_philox((uint32_t)(ts * F + f), (uint32_t)b, x.seed_lo, x.seed_hi) >> 8) * (1.0f / 16777216.0f));
    return v;
}

// frame element at flat index i of channel c (fp32 normalised, or uint8 raw)
__device__ __forceinline__ float xf_video(const void* __restrict__ v, const InXf& x, size_t i, int c) {
    if (x.kind == 0) return reinterpret_cast<const float*>(v)[i];
    return ((float)reinterpret_cast<const uint8_t*>(v)[i] * (1.0f / 255.0f) - x.mean[c]) * x.inv_std[c];
}
