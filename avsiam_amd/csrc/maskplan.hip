// k7 - random masking on the device: per-sequence random permutation by sorting noise, keep the first `keep` tokens.
//
// Replaces random_masking_unstructured / random_masking_structured of the reference
// (/root/reference/src/models/cav_mae_base.py:365-439): noise = rand(L); [structured: noise = 1.1 on chosen time
// columns / frequency rows, :415-422]; ids_shuffle = argsort(noise); ids_restore = argsort(ids_shuffle); keep the first
// int(L (1 - ratio)).  The reference spends two argsorts, a gather and B*(t+f) scalar kernel launches per call; here
// ONE launch handles every sequence of a pass (B audio + B*T video sequences): one workgroup per sequence draws the
// noise from a counter-based Philox4x32-10 stream (key = seed, counter = (token, sequence)), sorts (noise, index)
// pairs with a bitonic network in LDS (ties broken by index -> deterministic, unlike the reference's unstable argsort)
// and writes what the engine consumes directly:
//   row_src/row_tok[row_off + j]   = (src_id, token id)            for the kept tokens, in keep order (patch gather)
//   src_row[dec_off + token]       = enc_base + j | -1             for the decoder un-shuffle
//   mask[mask_off + token]         = 0 kept | 1 removed            the loss mask the forward returns (:385-388)
//   ids_out[ids_off + j]           = j-th token of the shuffle     (optional: lets tests rebuild the plan)
// GROUPED decoder layout (round 6; PlanSeq.dec_m_off >= 0, avs_mask_plan_grouped): the decoder rows of a sample are ordered
// [tokens whose prediction is scored (mask 1) | kept tokens] instead of by position, so that the rows the LAST decoder block and the
// prediction heads still have to compute (cav_mae_base.py:629-635,679-682: loss * mask) are the first rows of every sequence:
//   src_row[dec row]               = enc_base + j | -1             dec row = dec_k_off + j (kept, j < keep) | dec_m_off + j - keep (masked)
//   pos_row[dec row]               = pos_base + token              which row of the positional table [pos_a ; pos_v] the row's token takes
//   row_of_pos[dec_off + token]    = dec row                       the inverse (un-shuffle backward walks the positions)
//   pred_id[pred_off + j - keep]   = mask_off + token              which (sample, token) a compact prediction row scores
#include "common.h"

struct PlanSeq {        // one sequence to draw (int32 x 16, filled by the host)
    int L;              // tokens before masking
    int keep;           // tokens kept
    int row_off;        // first row of this sequence in the packed token matrix
    int src_id;         // sample (audio) / frame image (video) index written to row_src
    int dec_off;        // offset of this sequence in the decoder layout (src_row / mask), or -1
    int enc_base;       // row of the sequence's first kept token in the joint encoder layout
    int t_patches;      // > 0: structured audio masking on a [f][t] grid with the two bit masks below
    int ids_off;        // offset into ids_out, or -1
    int mask_off;       // offset of this sequence in mask_out (loss-mask layout), used when dec_off >= 0
    int tmask_x;        // bits 64..95 of the structured time mask (grids of more than 64 time patches: 73 at patch stride 14)
    int pad1, pad2;
    int dec_m_off;      // grouped decoder layout: decoder row of this sequence's first MASKED token, -1 = the classic (position-ordered) layout
    int dec_k_off;      //   decoder row of its first KEPT token
    int pred_off;       //   compact prediction row of its first masked token
    int pos_base;       //   row of the decoder's positional table [pos_a ; pos_v] its token 0 takes (audio: 0; every frame: La - the frames share pos_v)
};

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }

// Philox4x32-10 (Salmon et al., SC'11) - counter-based, so every (sequence, token) draw is independent of launch shape
__device__ __forceinline__ uint32_t philox_first(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

__global__ __launch_bounds__(256) void mask_plan_kernel(const PlanSeq* __restrict__ seqs, const unsigned* __restrict__ tmask_lo,
                                                        const unsigned* __restrict__ tmask_hi, const unsigned* __restrict__ fmask,
                                                        unsigned seed_lo, unsigned seed_hi, int* __restrict__ row_src,
                                                        int* __restrict__ row_tok, int* __restrict__ src_row,
                                                        float* __restrict__ mask_out, int* __restrict__ ids_out,
                                                        const unsigned long long* __restrict__ seed_dev, int* __restrict__ pos_row,
                                                        int* __restrict__ row_of_pos, int* __restrict__ pred_id) {
    __shared__ float key[1024];
    __shared__ short idx[1024];
    if (seed_dev) {                             // the key lives in device memory (a step replayed from a captured hipGraph: avs_mask_plan_dev)
        const unsigned long long sd = seed_dev[0];
        seed_lo = (unsigned)sd;
        seed_hi = (unsigned)(sd >> 32);
    }
    const int sq = blockIdx.x;
    const PlanSeq s = seqs[sq];
    int n2 = 64;
    while (n2 < s.L) n2 <<= 1;
    unsigned tl = 0, th = 0, fm = 0;
    if (s.t_patches > 0) { tl = tmask_lo[sq]; th = tmask_hi[sq]; fm = fmask[sq]; }
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        float v = 2.0f;                                           // padding sorts last
        if (i < s.L) {
            const uint32_t r = philox_first((uint32_t)i, (uint32_t)sq, 0u, 0u, seed_lo, seed_hi);
            v = (float)(r >> 8) * (1.0f / 16777216.0f);           // uniform [0,1) with 24 bits, like torch.rand
            if (s.t_patches > 0) {
                const int f = i / s.t_patches, t = i - f * s.t_patches;
                const bool tm = t < 32 ? (tl >> t) & 1u : t < 64 ? (th >> (t - 32)) & 1u : ((unsigned)s.tmask_x >> (t - 64)) & 1u;
                if (tm || ((fm >> f) & 1u)) v = 1.1f;             // "large value will be removed" (:408,413,418,422)
            }
        }
        key[i] = v;
        idx[i] = (short)i;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const float a = key[i], b = key[p];
                    const short ia = idx[i], ib = idx[p];
                    const bool up = (i & k) == 0;
                    const bool gt = a > b || (a == b && ia > ib);
                    if (gt == up) { key[i] = b; key[p] = a; idx[i] = ib; idx[p] = ia; }
                }
            }
            __syncthreads();
        }
    for (int j = threadIdx.x; j < s.L; j += blockDim.x) {
        const int tok = idx[j];
        if (j < s.keep) {
            row_src[s.row_off + j] = s.src_id;
            row_tok[s.row_off + j] = tok;
        }
        if (s.dec_off >= 0) {
            if (s.dec_m_off >= 0) {                               // grouped decoder layout: scored rows first
                const int dr = j < s.keep ? s.dec_k_off + j : s.dec_m_off + (j - s.keep);
                src_row[dr] = j < s.keep ? s.enc_base + j : -1;
                pos_row[dr] = s.pos_base + tok;
                row_of_pos[s.dec_off + tok] = dr;
                if (j >= s.keep) pred_id[s.pred_off + (j - s.keep)] = s.mask_off + tok;
            } else {
                src_row[s.dec_off + tok] = j < s.keep ? s.enc_base + j : -1;
            }
            mask_out[s.mask_off + tok] = j < s.keep ? 0.0f : 1.0f;
        }
        if (s.ids_off >= 0) ids_out[s.ids_off + j] = tok;
    }
}

extern "C" int avs_mask_plan(const int* seqs, int nseq, const unsigned* tmask_lo, const unsigned* tmask_hi, const unsigned* fmask,
                             unsigned long long seed, int* row_src, int* row_tok, int* src_row, float* mask_out, int* ids_out,
                             hipStream_t stream) {
    AVS_CHECK_ARG(seqs && nseq > 0 && row_src && row_tok, "mask_plan: bad arguments");
    mask_plan_kernel<<<nseq, 256, 0, stream>>>(reinterpret_cast<const PlanSeq*>(seqs), tmask_lo, tmask_hi, fmask, (unsigned)seed,
                                               (unsigned)(seed >> 32), row_src, row_tok, src_row, mask_out, ids_out, nullptr, nullptr, nullptr, nullptr);
    AVS_LAUNCH_CHECK("mask_plan");
    return 0;
}

// the same with the Philox key read from DEVICE memory (seed_dev[0]) when the kernel runs: kernel arguments are frozen in a captured
// hipGraph, a key that advances from step to step is advanced by a node of the graph itself
extern "C" int avs_mask_plan_dev(const int* seqs, int nseq, const unsigned* tmask_lo, const unsigned* tmask_hi, const unsigned* fmask,
                                 const unsigned long long* seed_dev, int* row_src, int* row_tok, int* src_row, float* mask_out, int* ids_out,
                                 hipStream_t stream) {
    AVS_CHECK_ARG(seqs && nseq > 0 && row_src && row_tok && seed_dev, "mask_plan_dev: bad arguments");
    mask_plan_kernel<<<nseq, 256, 0, stream>>>(reinterpret_cast<const PlanSeq*>(seqs), tmask_lo, tmask_hi, fmask, 0u, 0u, row_src, row_tok, src_row,
                                               mask_out, ids_out, seed_dev, nullptr, nullptr, nullptr);
    AVS_LAUNCH_CHECK("mask_plan_dev");
    return 0;
}

// Sequences whose descriptor says dec_m_off >= 0 are laid out in the GROUPED decoder order (file header) and need the three extra index
// arrays; seed_dev (may be NULL): the Philox key in device memory, as avs_mask_plan_dev, else `seed`.
extern "C" int avs_mask_plan_grouped(const int* seqs, int nseq, const unsigned* tmask_lo, const unsigned* tmask_hi, const unsigned* fmask,
                                     unsigned long long seed, const unsigned long long* seed_dev, int* row_src, int* row_tok, int* src_row,
                                     float* mask_out, int* ids_out, int* pos_row, int* row_of_pos, int* pred_id, hipStream_t stream) {
    AVS_CHECK_ARG(seqs && nseq > 0 && row_src && row_tok && src_row && mask_out && pos_row && row_of_pos && pred_id, "mask_plan_grouped: bad arguments");
    mask_plan_kernel<<<nseq, 256, 0, stream>>>(reinterpret_cast<const PlanSeq*>(seqs), tmask_lo, tmask_hi, fmask, (unsigned)seed, (unsigned)(seed >> 32),
                                               row_src, row_tok, src_row, mask_out, ids_out, seed_dev, pos_row, row_of_pos, pred_id);
    AVS_LAUNCH_CHECK("mask_plan_grouped");
    return 0;
}
