// HBM-bound data-movement kernels of the AVSiam hot path: patch gather (k1/k7), decoder un-shuffle (k8),
// token-mean pooling, column sums (bias grads), positional-embedding scatter, weight cast/transposes and
// the fused Adam step.  All accesses are 8-16 B per lane and coalesced along the feature dimension.
#include "common.h"

// ---------------------------------------------------------------------------------------------------
// k1 + k7: im2col of the KEPT patches only.  The reference embeds every patch and discards 75 % of them
// (/root/reference/src/models/cav_mae_base.py:448-455 then :476-477); conv k=s=16 is a per-patch linear
// map, so gathering first is identical math.  Output rows are GEMM A-operands (bf16).
// audio: a [B, time, mel] fp32; token = f*tP + t (:444-445 transpose -> image [mel, time]);
//        out[r, p*16+q] = a[b, t*16+q, f*16+p]
__global__ void im2col_audio_kernel(const float* __restrict__ a, const int* __restrict__ row_b,
                                    const int* __restrict__ row_tok, bf16_t* __restrict__ out, int rows, int tlen,
                                    int mel, int tP, InXf xf) {
    const int r = blockIdx.x;
    const int q = threadIdx.x >> 4, p = threadIdx.x & 15;      // consecutive threads read consecutive mel bins
    const int b = row_b[r], tok = row_tok[r];
    const int f = tok / tP, t = tok - f * tP;
    const float v = xf_audio(a, xf, b, t * 16 + q, f * 16 + p, tlen, mel);
    out[(size_t)r * 256 + p * 16 + q] = f2bf(v);
}

// video: v [NF, C, H, W] fp32; token = gy*G + gx; out[r, c*256 + p*16 + q] = v[img, c, gy*16+p, gx*16+q]
__global__ void im2col_video_kernel(const void* __restrict__ vin, const int* __restrict__ row_img,
                                    const int* __restrict__ row_tok, bf16_t* __restrict__ out, int rows, int C, int H,
                                    int W, int G, InXf xf) {
    const int r = blockIdx.x;
    const int img = row_img[r], tok = row_tok[r];
    const int gy = tok / G, gx = tok - gy * G;
    const int K = C * 256;
    for (int e = threadIdx.x * 4; e < K; e += blockDim.x * 4) {
        const int c = e >> 8, p = (e >> 4) & 15, q = e & 15;
        const size_t i = (((size_t)img * C + c) * H + gy * 16 + p) * W + gx * 16 + q;
        float4 x;
        if (xf.kind == 0) {
            x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(vin) + i);
        } else {                                               // four uint8 pixels in one 4-byte load
            const uint32_t w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(vin) + i);
            const float m = xf.mean[c], is = xf.inv_std[c];
            x = make_float4(((float)(w & 0xff) * (1.0f / 255.0f) - m) * is, ((float)((w >> 8) & 0xff) * (1.0f / 255.0f) - m) * is,
                            ((float)((w >> 16) & 0xff) * (1.0f / 255.0f) - m) * is, ((float)(w >> 24) * (1.0f / 255.0f) - m) * is);
        }
        uint2 o;
        o.x = pack_bf2(x.x, x.y);
        o.y = pack_bf2(x.z, x.w);
        *reinterpret_cast<uint2*>(out + (size_t)r * K + e) = o;
    }
}

// The same gathers for a patch STRIDE S < 16 on 16 x 16 patch storage (config.stride: the 14 x 14 grid of ViT-H/14): token (t, f) / (gy, gx)
// starts at pixel t*S / gy*S, only the S x S corner of the 256 positions per channel is read, the rest of the row is zero (so the padded
// positions of the patch-embedding kernel never contribute and never receive a gradient).  Element-wise loads: S need not be a multiple of 4.
__global__ void im2col_audio_s_kernel(const float* __restrict__ a, const int* __restrict__ row_b, const int* __restrict__ row_tok,
                                      bf16_t* __restrict__ out, int rows, int tlen, int mel, int tP, int S, InXf xf) {
    const int r = blockIdx.x;
    const int q = threadIdx.x >> 4, p = threadIdx.x & 15;
    const int b = row_b[r], tok = row_tok[r];
    const int f = tok / tP, t = tok - f * tP;
    const float v = (q < S && p < S) ? xf_audio(a, xf, b, t * S + q, f * S + p, tlen, mel) : 0.f;
    out[(size_t)r * 256 + p * 16 + q] = f2bf(v);
}

__global__ void im2col_video_s_kernel(const void* __restrict__ vin, const int* __restrict__ row_img, const int* __restrict__ row_tok,
                                      bf16_t* __restrict__ out, int rows, int C, int H, int W, int G, int S, InXf xf) {
    const int r = blockIdx.x;
    const int img = row_img[r], tok = row_tok[r];
    const int gy = tok / G, gx = tok - gy * G;
    const int K = C * 256;
    for (int e = threadIdx.x; e < K; e += blockDim.x) {
        const int c = e >> 8, p = (e >> 4) & 15, q = e & 15;
        float v = 0.f;
        if (p < S && q < S) v = xf_video(vin, xf, (((size_t)img * C + c) * H + gy * S + p) * W + gx * S + q, c);
        out[(size_t)r * K + e] = f2bf(v);
    }
}

// fp32 -> bf16 with a scale (d(2*(conv+pos)) = 2*dOut for the embed prologue `x + norm_pre(x)`, :449-450)
__global__ void cast_scale_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n4, float alpha) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        uint2 o;
        o.x = pack_bf2(v.x * alpha, v.y * alpha);
        o.y = pack_bf2(v.z * alpha, v.w * alpha);
        reinterpret_cast<uint2*>(y)[i] = o;
    }
}

// dst[idx[r], :] += scale * src[r, :]   (positional-embedding gradient: many samples hit the same row)
__global__ void scatter_add_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx, float* dst,
                                        int rows, int D, float scale) {
    const int r = blockIdx.x;
    float* d = dst + (size_t)idx[r] * D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) atomicAdd(d + c, scale * bf2f(src[(size_t)r * D + c]));
}

// the same without atomics (AvsTuning::det): a block owns 64 columns and walks ALL rows in order - one writer per element, fixed order
__global__ __launch_bounds__(64) void scatter_add_rows_det_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx, float* dst,
                                                                  int rows, int D, float scale) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= D) return;
    for (int r = 0; r < rows; ++r) dst[(size_t)idx[r] * D + c] += scale * bf2f(src[(size_t)r * D + c]);
}

// out[c] += sum_r x[r, c]  (bias gradients).  A block owns 64 columns x COLSUM_ROWS rows: 8 threads cover the 64
// columns with one 16-byte load each (a full 128-B line per row), 32 row-lanes stride the rows; partial sums are
// combined through LDS and leave as one atomic per column per block.
constexpr int COLSUM_ROWS = 512;
// rows_per_block: COLSUM_ROWS, or all the rows (AvsTuning::det: one block - one adder - per 64 columns)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, long long ld, float* out, int rows, int rows_per_block) {
    __shared__ float red[32][65];
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = blockIdx.x * 64 + cg * 8;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = r0 + rl; r < r1; r += 32) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + (size_t)r * ld + c0);
        acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xffff0000u);
        acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xffff0000u);
        acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xffff0000u);
        acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xffff0000u);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += red[k][threadIdx.x];
        atomicAdd(out + blockIdx.x * 64 + threadIdx.x, s);
    }
}

// ---------------------------------------------------------------------------------------------------
// k8: decoder un-shuffle (forward_decoder :604-626).  Output row r of [B*(La+T*Lv), Dd]:
//   src_row[r] >= 0 -> x[src_row[r]]  else mask_token;  + pos[pos_row[r]] + modality[row_mod[r]]
// pos is the concatenation [decoder_pos_embed_a ; decoder_pos_embed_v].
__global__ void unshuffle_fwd_kernel(const float* __restrict__ x, const int* __restrict__ src_row,
                                     const int* __restrict__ pos_row, const uint8_t* __restrict__ row_mod,
                                     const float* __restrict__ mask_token, const float* __restrict__ pos_a,
                                     const float* __restrict__ pos_v, int La, const float* __restrict__ mod_a,
                                     const float* __restrict__ mod_v, float* __restrict__ out, int rows, int D) {
    const int r = blockIdx.x;
    const int s = src_row[r];
    const int pr = pos_row[r];
    const float4* src = reinterpret_cast<const float4*>(s >= 0 ? x + (size_t)s * D : mask_token);
    const float4* pp = reinterpret_cast<const float4*>(pr < La ? pos_a + (size_t)pr * D : pos_v + (size_t)(pr - La) * D);
    const float4* mm = reinterpret_cast<const float4*>(row_mod[r] ? mod_v : mod_a);
    float4* o = reinterpret_cast<float4*>(out + (size_t)r * D);
    for (int c = threadIdx.x; c < D / 4; c += blockDim.x) {
        const float4 a = src[c], p = pp[c], m = mm[c];
        o[c] = make_float4(a.x + p.x + m.x, a.y + p.y + m.y, a.z + p.z + m.z, a.w + p.w + m.w);
    }
}

// Backward of the un-shuffle.  One block per position l of [La + Lv]; it walks every (sample, frame) row at
// that position: sums the positional gradient in registers (no atomics), routes kept rows back to the encoder
// layout (dx[src] = dout row) and accumulates mask-token / modality sums (one atomic per column per block).
// row_of_pos (may be NULL): the decoder row that holds position (b, l) when the decoder rows are not in position order (the grouped layout of
// maskplan.hip: scored rows first); src_row is indexed by decoder row.
// One block per position, UB_G row groups of 128 threads each (round 6: a single group walked the 640 rows of a frame position one after the other -
// the launch took 0.5 ms at 0.6 TB/s): group g takes the (sample, frame) pairs g, g + UB_G, ..., the groups' sums fold through LDS in a fixed order, so
// the positional gradient still has ONE writer; the token sums (mask token, modality embeddings) leave as one atomic per column and block as before.
// det (AvsTuning::det): the token sums come from unshuffle_tokens_det_kernel instead.
constexpr int UB_G = 4;
__global__ __launch_bounds__(128 * UB_G) void unshuffle_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ src_row, int B, int T,
                                     int La, int Lv, float* __restrict__ dx, float* dpos_a, float* dpos_v,
                                     float* dmask, float* dmod_a, float* dmod_v, int D, const int* __restrict__ row_of_pos, int det) {
    __shared__ float4 red[2][UB_G][128];
    const int l = blockIdx.x;
    const bool audio = l < La;
    const int Ltot = La + T * Lv;
    const int grp = threadIdx.x >> 7, tc = threadIdx.x & 127;
    const int reps = audio ? 1 : T;
    for (int c0 = 0; c0 < D / 4; c0 += 128) {
        const int c = c0 + tc;
        float4 sp = make_float4(0, 0, 0, 0), sm = sp;
        if (c < D / 4)
            for (int i = grp; i < B * reps; i += UB_G) {
                const int b = i / reps, t = i - b * reps;
                const int rp = b * Ltot + (audio ? l : La + t * Lv + (l - La));
                const int r = row_of_pos ? row_of_pos[rp] : rp;
                const float4 g = reinterpret_cast<const float4*>(dout + (size_t)r * D)[c];
                sp.x += g.x; sp.y += g.y; sp.z += g.z; sp.w += g.w;
                const int s = src_row[r];
                if (s >= 0) reinterpret_cast<float4*>(dx + (size_t)s * D)[c] = g;
                else { sm.x += g.x; sm.y += g.y; sm.z += g.z; sm.w += g.w; }
            }
        __syncthreads();
        red[0][grp][tc] = sp; red[1][grp][tc] = sm;
        __syncthreads();
        if (grp == 0 && c < D / 4) {
            sp = red[0][0][tc]; sm = red[1][0][tc];
#pragma unroll
            for (int g2 = 1; g2 < UB_G; ++g2) {
                const float4 a = red[0][g2][tc], m2 = red[1][g2][tc];
                sp.x += a.x; sp.y += a.y; sp.z += a.z; sp.w += a.w;
                sm.x += m2.x; sm.y += m2.y; sm.z += m2.z; sm.w += m2.w;
            }
            float* dp = audio ? dpos_a + (size_t)l * D : dpos_v + (size_t)(l - La) * D;
            float4 old = reinterpret_cast<float4*>(dp)[c];
            reinterpret_cast<float4*>(dp)[c] = make_float4(old.x + sp.x, old.y + sp.y, old.z + sp.z, old.w + sp.w);
            if (!det) {                                           // (AvsTuning::det: the token sums come from unshuffle_tokens_det_kernel, one writer each)
                float* dm = audio ? dmod_a : dmod_v;
                atomicAdd(dm + c * 4 + 0, sp.x); atomicAdd(dm + c * 4 + 1, sp.y);
                atomicAdd(dm + c * 4 + 2, sp.z); atomicAdd(dm + c * 4 + 3, sp.w);
                atomicAdd(dmask + c * 4 + 0, sm.x); atomicAdd(dmask + c * 4 + 1, sm.y);
                atomicAdd(dmask + c * 4 + 2, sm.z); atomicAdd(dmask + c * 4 + 3, sm.w);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// token mean of each packed sequence (``.mean(dim=1)`` at :563,566) on the fp32 final-norm output:
// reps[s] = mean_{r in seg s} y[r]
// grid (segments, D/128): a workgroup owns 32 float4 columns of one segment; its 8 row groups stride the rows and are
// folded through LDS (one workgroup per segment walking up to 1960 rows serially took 0.85 ms per call).
// row_map (may be NULL): the mean of segment s goes to reps[row_map[s]] - the contrastive pass's slot order -> sample order
// (cav_mae_base.py:584-590: the reference's inverse permutation + gather) without a pass of its own
__global__ __launch_bounds__(256) void segment_mean_fwd_kernel(const float* __restrict__ y, const int* __restrict__ seg_start,
                                                               float* __restrict__ reps, int D, const int* __restrict__ row_map) {
    __shared__ float4 part[8][32];
    const int s = blockIdx.x;
    const int r0 = seg_start[s], r1 = seg_start[s + 1];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.y * 32 + cl;                       // float4 column
    float4 a = make_float4(0, 0, 0, 0);
    if (c < D / 4)
        for (int r = r0 + rg; r < r1; r += 8) {
            const float4 v = reinterpret_cast<const float4*>(y + (size_t)r * D)[c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    part[rg][cl] = a;
    __syncthreads();
    if (rg == 0 && c < D / 4) {
#pragma unroll
        for (int g = 1; g < 8; ++g) {
            const float4 v = part[g][cl];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        const float inv = 1.0f / (float)(r1 - r0);
        reinterpret_cast<float4*>(reps + (size_t)(row_map ? row_map[s] : s) * D)[c] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
    }
}

// dy[r] = scale * dreps[row_map ? row_map[seg(r)] : seg(r)] / len(seg)
__global__ void segment_mean_bwd_kernel(const float* __restrict__ dreps, const int* __restrict__ seg_start, float* __restrict__ dy,
                                        int D, float scale, const int* __restrict__ row_map) {
    const int s = blockIdx.x;
    const int r0 = seg_start[s], r1 = seg_start[s + 1];
    const float inv = scale / (float)(r1 - r0);
    for (int c = threadIdx.x; c < D / 4; c += blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(dreps + (size_t)(row_map ? row_map[s] : s) * D)[c];
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        for (int r = r0; r < r1; ++r) reinterpret_cast<float4*>(dy + (size_t)r * D)[c] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// bf16 transpose [R, C] -> [C, R] through LDS (weight copies for the dgrad GEMMs).
__global__ void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R, int C) {
    __shared__ bf16_t tile[64][66];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    for (int i = threadIdx.y; i < 64; i += blockDim.y) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < R && c < C) ? in[(size_t)r * C + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 64; i += blockDim.y) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (c < C && r < R) out[(size_t)c * R + r] = tile[threadIdx.x][i];
    }
}

// Batched form: one launch transposes every weight of a pass.  desc[m] = {src, dst, R, C, first_tile, tiles_per_row}
// (int64 x 6); tile_map[b] = matrix of workgroup b.
__global__ void transpose_batched_kernel(const long long* __restrict__ desc, const int* __restrict__ tile_map) {
    __shared__ bf16_t tile[64][66];
    const long long* d = desc + (size_t)tile_map[blockIdx.x] * 6;
    const bf16_t* in = reinterpret_cast<const bf16_t*>(d[0]);
    bf16_t* out = reinterpret_cast<bf16_t*>(d[1]);
    const int R = (int)d[2], C = (int)d[3];
    const int local = blockIdx.x - (int)d[4], tpr = (int)d[5];
    const int c0 = (local % tpr) * 64, r0 = (local / tpr) * 64;
    for (int i = threadIdx.y; i < 64; i += blockDim.y) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < R && c < C) ? in[(size_t)r * C + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 64; i += blockDim.y) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (c < C && r < R) out[(size_t)c * R + r] = tile[threadIdx.x][i];
    }
}

__global__ void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        uint2 o;
        o.x = pack_bf2(v.x, v.y);
        o.y = pack_bf2(v.z, v.w);
        reinterpret_cast<uint2*>(y)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] = f2bf(x[n4 * 4 + threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------
// Fused Adam over a contiguous slice of the flat arena: torch.optim.Adam(lr, betas=(0.95,0.999), eps=1e-8,
// weight_decay=5e-7) as constructed at /root/reference/src/traintest_cavmae_base.py:64-66 (L2 decay folded into
// the gradient, bias correction).  Also refreshes the bf16 shadow copy the GEMMs read.  28 B/param of traffic.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            bf16_t* __restrict__ pb, size_t n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2_sqrt, float gscale) {
    const size_t n4 = n / 4;
    const float step = lr / bc1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = gp[k] * gscale + wd * pp[k];
            mp[k] = b1 * mp[k] + (1.f - b1) * gg;
            vp[k] = b2 * vp[k] + (1.f - b2) * gg * gg;
            pp[k] -= step * mp[k] / (sqrtf(vp[k]) / bc2_sqrt + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (pb) {
            uint2 o;
            o.x = pack_bf2(pv.x, pv.y);
            o.y = pack_bf2(pv.z, pv.w);
            reinterpret_cast<uint2*>(pb)[i] = o;
        }
    }
}

// The same update with the step count in DEVICE memory (a training step replayed from a captured hipGraph: kernel arguments are frozen at
// capture, so what changes from step to step - this count, the mask-plan seeds - lives in memory the graph's own nodes advance).
// step_dev[0] >= 1 is the count of THIS update; the bias corrections are evaluated in double, as torch.optim.Adam and avs_adam do.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                bf16_t* __restrict__ pb, size_t n, float lr, float b1, float b2, float eps, float wd,
                                const int* __restrict__ step_dev, float gscale) {
    const double t = (double)step_dev[0];
    const float bc1 = (float)(1.0 - pow((double)b1, t));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    const size_t n4 = n / 4;
    const float step = lr / bc1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = gp[k] * gscale + wd * pp[k];
            mp[k] = b1 * mp[k] + (1.f - b1) * gg;
            vp[k] = b2 * vp[k] + (1.f - b2) * gg * gg;
            pp[k] -= step * mp[k] / (sqrtf(vp[k]) / bc2_sqrt + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (pb) {
            uint2 o;
            o.x = pack_bf2(pv.x, pv.y);
            o.y = pack_bf2(pv.z, pv.w);
            reinterpret_cast<uint2*>(pb)[i] = o;
        }
    }
}

// ===================================================================================================
static inline int grid_1d(size_t n, int block) {
    size_t g = (n + block - 1) / block;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

// host-side check + copy of the caller's descriptor (include/avsiam_hip.h: avs_input_xf) into the kernel argument
int avs_make_xf(const avs_input_xf_t* x, int want_kind, InXf* out, const char* who) {
    InXf r{};
    if (x && x->kind != 0) {
        if (x->kind != want_kind) { avs_set_error("%s: input transform kind %d does not fit this input (want %d)", who, x->kind, want_kind); return -2; }
        r.kind = x->kind;
        for (int c = 0; c < 3; ++c) {
            if (x->std[c] == 0.f && (c == 0 || want_kind == 2)) { avs_set_error("%s: zero std in the input transform", who); return -2; }
            r.mean[c] = x->mean[c];
            r.inv_std[c] = x->std[c] != 0.f ? 1.0f / x->std[c] : 0.f;
        }
        r.shift = x->shift; r.amp = x->amp;
        r.seed_lo = (uint32_t)x->seed; r.seed_hi = (uint32_t)(x->seed >> 32);
    }
    *out = r;
    return 0;
}

extern "C" int avs_im2col_audio_xf(const float* a, const int* row_b, const int* row_tok, bf16_t* out, int rows, int tlen,
                                   int mel, int t_patches, const avs_input_xf_t* xf, hipStream_t stream) {
    AVS_CHECK_ARG(rows > 0 && a && row_b && row_tok && out, "im2col_audio: bad args");
    InXf x;
    if (int rc = avs_make_xf(xf, 1, &x, "im2col_audio")) return rc;
    im2col_audio_kernel<<<rows, 256, 0, stream>>>(a, row_b, row_tok, out, rows, tlen, mel, t_patches, x);
    AVS_LAUNCH_CHECK("im2col_audio");
    return 0;
}

extern "C" int avs_im2col_audio(const float* a, const int* row_b, const int* row_tok, bf16_t* out, int rows, int tlen,
                                int mel, int t_patches, hipStream_t stream) {
    return avs_im2col_audio_xf(a, row_b, row_tok, out, rows, tlen, mel, t_patches, nullptr, stream);
}

extern "C" int avs_im2col_video_xf(const void* v, const int* row_img, const int* row_tok, bf16_t* out, int rows, int C, int H,
                                   int W, const avs_input_xf_t* xf, hipStream_t stream) {
    AVS_CHECK_ARG(rows > 0 && v && row_img && row_tok && out && (W % 16) == 0 && (H % 16) == 0, "im2col_video: bad args");
    InXf x;
    if (int rc = avs_make_xf(xf, 2, &x, "im2col_video")) return rc;
    AVS_CHECK_ARG(x.kind == 0 || C == 3, "im2col_video: uint8 frames need 3 channels");
    im2col_video_kernel<<<rows, 192, 0, stream>>>(v, row_img, row_tok, out, rows, C, H, W, W / 16, x);
    AVS_LAUNCH_CHECK("im2col_video");
    return 0;
}

extern "C" int avs_im2col_video(const float* v, const int* row_img, const int* row_tok, bf16_t* out, int rows, int C, int H,
                                int W, hipStream_t stream) {
    return avs_im2col_video_xf(v, row_img, row_tok, out, rows, C, H, W, nullptr, stream);
}

extern "C" int avs_im2col_audio_s(const float* a, const int* row_b, const int* row_tok, bf16_t* out, int rows, int tlen, int mel,
                                 int t_patches, int stride, const avs_input_xf_t* xf, hipStream_t stream) {
    if (stride == 16) return avs_im2col_audio_xf(a, row_b, row_tok, out, rows, tlen, mel, t_patches, xf, stream);
    AVS_CHECK_ARG(rows > 0 && a && row_b && row_tok && out && stride > 0 && stride < 16, "im2col_audio_s: bad args (stride %d)", stride);
    InXf x;
    if (int rc = avs_make_xf(xf, 1, &x, "im2col_audio_s")) return rc;
    im2col_audio_s_kernel<<<rows, 256, 0, stream>>>(a, row_b, row_tok, out, rows, tlen, mel, t_patches, stride, x);
    AVS_LAUNCH_CHECK("im2col_audio_s");
    return 0;
}

extern "C" int avs_im2col_video_s(const void* v, const int* row_img, const int* row_tok, bf16_t* out, int rows, int C, int H, int W,
                                 int stride, const avs_input_xf_t* xf, hipStream_t stream) {
    if (stride == 16) return avs_im2col_video_xf(v, row_img, row_tok, out, rows, C, H, W, xf, stream);
    AVS_CHECK_ARG(rows > 0 && v && row_img && row_tok && out && stride > 0 && stride < 16 && (W % stride) == 0 && (H % stride) == 0,
                  "im2col_video_s: bad args (stride %d)", stride);
    InXf x;
    if (int rc = avs_make_xf(xf, 2, &x, "im2col_video_s")) return rc;
    AVS_CHECK_ARG(x.kind == 0 || C == 3, "im2col_video_s: uint8 frames need 3 channels");
    im2col_video_s_kernel<<<rows, 256, 0, stream>>>(v, row_img, row_tok, out, rows, C, H, W, W / stride, stride, x);
    AVS_LAUNCH_CHECK("im2col_video_s");
    return 0;
}

extern "C" int avs_cast_scale_bf16(const float* x, bf16_t* y, long long n, float alpha, hipStream_t stream) {
    AVS_CHECK_ARG(n > 0 && (n % 4) == 0 && x && y, "cast_scale: n must be a positive multiple of 4");
    cast_scale_kernel<<<grid_1d(n / 4, 256), 256, 0, stream>>>(x, y, (size_t)n / 4, alpha);
    AVS_LAUNCH_CHECK("cast_scale");
    return 0;
}

extern "C" int avs_scatter_add_rows(const bf16_t* src, const int* idx, float* dst, int rows, int D, float scale,
                                    hipStream_t stream) {
    AVS_CHECK_ARG(rows > 0 && src && idx && dst, "scatter_add_rows: bad args");
    if (avs_tuning().det) scatter_add_rows_det_kernel<<<ceil_div(D, 64), 64, 0, stream>>>(src, idx, dst, rows, D, scale);
    else
    scatter_add_rows_kernel<<<rows, 256, 0, stream>>>(src, idx, dst, rows, D, scale);
    AVS_LAUNCH_CHECK("scatter_add_rows");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) operand preparation for avs_gemm_nt_fp8: |x| maximum of a tensor (the host derives the per-tensor scale 448 / amax
// from it - delayed scaling reads it a step later, calibration reads it at once) and y = e4m3(clamp(x * scale, +-448)).
__global__ void absmax_kernel(const void* __restrict__ x, int is_f32, size_t n8, size_t n, float* out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t e = i * 8 + j;
            if (e < n) m = fmaxf(m, fabsf(is_f32 ? reinterpret_cast<const float*>(x)[e] : bf2f(reinterpret_cast<const bf16_t*>(x)[e])));
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));      // non-negative floats order like their bit patterns
}

__global__ void quantize_fp8_kernel(const void* __restrict__ x, int is_f32, uint8_t* __restrict__ y, size_t n4, float scale_host, float* q, int e5m2) {
    // q (device record, common.h AVS_Q_*; may be NULL): the scale comes from q[0] and the largest |x| seen is folded into q[2]
    const float scale = q ? q[AVS_Q_SCALE] : scale_host;
    const float amax_seen = q_amax_peek(q);
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float v[4];
        if (is_f32) {
            const float4 f = reinterpret_cast<const float4*>(x)[i];
            v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
        } else {
            const uint2 u = reinterpret_cast<const uint2*>(x)[i];
            v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
            v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        }
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        const float lim = e5m2 ? 57344.0f : 448.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j] * scale, -lim, lim);
        int w;
        if (e5m2) {
            w = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
            w = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], w, true);
        } else {
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
        }
        reinterpret_cast<int*>(y)[i] = w;
    }
    if (q) q_amax_update(q, m, amax_seen);
}

// Delayed scaling (engine.FP8): per record s in [0, n): the amax gathered since the last update goes into the history ring
// hist[pos][s], the scale becomes fmax / (margin * max over the ring), the running amax restarts (at 0.9 x itself), and the saturation counter
// advances when the values just quantised exceeded the range of the scale they were quantised with.  One thread per record.
__global__ void fp8_scale_update_kernel(float* q, float* hist, int n, int nhist, int pos, float margin, int first, int count, float fmax) {
    const int s = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= first + count) return;
    float* r = q + AVS_Q_STRIDE * (size_t)s;
    float a = r[AVS_Q_AMAX];
    for (int k = 0; k < AVS_Q_NSHARD; ++k) {                   // the producers' shards (common.h)
        a = fmaxf(a, r[AVS_Q_SHARD0 + k * AVS_Q_SHARD_STRIDE]);
        r[AVS_Q_SHARD0 + k * AVS_Q_SHARD_STRIDE] = 0.f;
    }
    const float sc = r[AVS_Q_SCALE];
    if (sc > 0.f && a * sc > fmax) r[AVS_Q_SAT] += 1.0f;
    hist[(size_t)pos * n + s] = a;
    float m = 0.f;
    for (int h = 0; h < nhist; ++h) m = fmaxf(m, hist[(size_t)h * n + s]);
    if (m > 0.f) {
        const float ns = fmax / (margin * m);
        r[AVS_Q_SCALE] = ns;
        r[AVS_Q_INV] = 1.0f / ns;
    }
    // the running amax restarts at 0.9 x what was just recorded, not at 0: producers filter their atomics against the value they read at
    // kernel START (q_amax_peek) - from 0 every wave of the first kernel of a step would issue one (95 k atomics on one address: ~1 ms
    // per LayerNorm launch) - and with the floor only rows near the tensor's maximum do.  The recorded amax can fall by 10 % per step.
    r[AVS_Q_AMAX] = 0.9f * a;
}

extern "C" int avs_absmax(const void* x, int is_f32, long long n, float* out, hipStream_t stream) {
    AVS_CHECK_ARG(x && out && n > 0, "absmax: bad args");
    const size_t n8 = ((size_t)n + 7) / 8;
    const int blocks = (int)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
    absmax_kernel<<<blocks, 256, 0, stream>>>(x, is_f32, n8, (size_t)n, out);
    AVS_LAUNCH_CHECK("absmax");
    return 0;
}

extern "C" int avs_quantize_fp8(const void* x, int is_f32, uint8_t* y, long long n, float scale, float* q, int e5m2, hipStream_t stream) {
    AVS_CHECK_ARG(x && y && n > 0 && (n % 4) == 0, "quantize_fp8: n must be a multiple of 4");
    const size_t n4 = (size_t)n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    quantize_fp8_kernel<<<blocks, 256, 0, stream>>>(x, is_f32, y, n4, scale, q, e5m2);
    AVS_LAUNCH_CHECK("quantize_fp8");
    return 0;
}

// One launch quantises MANY bf16 tensors (all weights of a stack, once per forward: the per-GEMM quantising passes were ~800 launches
// of a few microseconds each per step).  desc[d] = {src (bf16), dst (u8), numel / 4, record index}; chunk c of 8192 elements belongs to
// descriptor cmap[2c] and starts at 4-element group cmap[2c + 1]; the scale comes from q[record], the |max| seen is folded into it.
__global__ __launch_bounds__(256) void quantize_fp8_batched_kernel(const long long* __restrict__ desc, const int* __restrict__ cmap, float* q, int e5m2) {
    const int d = cmap[2 * blockIdx.x];
    const size_t g0 = (size_t)cmap[2 * blockIdx.x + 1];
    const uint2* src = reinterpret_cast<const uint2*>(desc[4 * d]);
    int* dst = reinterpret_cast<int*>(desc[4 * d + 1]);
    const size_t n4 = (size_t)desc[4 * d + 2];
    float* rec = q + AVS_Q_STRIDE * desc[4 * d + 3];
    const float scale = rec[AVS_Q_SCALE], seen = rec[AVS_Q_AMAX];
    const float lim = e5m2 ? 57344.0f : 448.0f;
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const size_t i = g0 + (size_t)it * 256 + threadIdx.x;
        if (i >= n4) break;
        const uint2 u = src[i];
        float v[4] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j] * scale, -lim, lim);
        int w;
        if (e5m2) {
            w = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
            w = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], w, true);
        } else {
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
        }
        dst[i] = w;
    }
    q_amax_update(rec, m, seen);
}

extern "C" int avs_quantize_fp8_batched(const long long* desc, const int* cmap, int nchunks, float* q, int e5m2, hipStream_t stream) {
    AVS_CHECK_ARG(desc && cmap && q && nchunks > 0, "quantize_fp8_batched: bad args");
    quantize_fp8_batched_kernel<<<nchunks, 256, 0, stream>>>(desc, cmap, q, e5m2);
    AVS_LAUNCH_CHECK("quantize_fp8_batched");
    return 0;
}

// Many small regions zeroed by ONE launch: desc[d] = {address (16-byte aligned), bytes / 16}; chunk c of 64 KB belongs to descriptor
// cmap[2c] and starts at 16-byte group cmap[2c + 1].  The pad rows (beyond the token rows, up to the 64- / 128- / 256-row granule the
// GEMMs read) of a stack's buffers when the passes of a step SHARE one activation pool (engine.BufferPool): another pass has written
// there since, and the weight-gradient GEMMs contract over those rows.
__global__ __launch_bounds__(256) void zero_batched_kernel(const long long* __restrict__ desc, const int* __restrict__ cmap) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const int d = cmap[2 * blockIdx.x];
    const size_t g0 = (size_t)cmap[2 * blockIdx.x + 1];
    u32x4* dst = reinterpret_cast<u32x4*>(desc[2 * d]);
    const size_t n16 = (size_t)desc[2 * d + 1];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const size_t i = g0 + (size_t)it * 256 + threadIdx.x;
        if (i >= n16) break;
        dst[i] = u32x4{0u, 0u, 0u, 0u};
    }
}

extern "C" int avs_zero_batched(const long long* desc, const int* cmap, int nchunks, hipStream_t stream) {
    AVS_CHECK_ARG(desc && cmap && nchunks > 0, "zero_batched: bad args");
    zero_batched_kernel<<<nchunks, 256, 0, stream>>>(desc, cmap);
    AVS_LAUNCH_CHECK("zero_batched");
    return 0;
}

extern "C" int avs_fp8_scale_update(float* q, float* hist, int n, int nhist, int pos, float margin, int first, int count, float fmax, hipStream_t stream) {
    AVS_CHECK_ARG(fmax == 448.0f || fmax == 57344.0f, "fp8_scale_update: fmax is 448 (e4m3) or 57344 (e5m2)");
    AVS_CHECK_ARG(q && hist && n > 0 && nhist > 0 && pos >= 0 && pos < nhist && margin >= 1.0f && first >= 0 && count > 0 && first + count <= n,
                  "fp8_scale_update: bad args (n=%d nhist=%d pos=%d first=%d count=%d)", n, nhist, pos, first, count);
    fp8_scale_update_kernel<<<ceil_div(count, 256), 256, 0, stream>>>(q, hist, n, nhist, pos, margin, first, count, fmax);
    AVS_LAUNCH_CHECK("fp8_scale_update");
    return 0;
}

extern "C" int avs_colsum_bf16(const bf16_t* x, long long ld, float* out, int rows, int C, hipStream_t stream) {
    AVS_CHECK_ARG(rows > 0 && (C % 64) == 0 && ld >= C && (ld % 8) == 0 && x && out, "colsum: C must be a multiple of 64, ld of 8");
    const int rpb = avs_tuning().det ? rows : COLSUM_ROWS;
    dim3 grid(C / 64, ceil_div(rows, rpb));
    colsum_bf16_kernel<<<grid, 256, 0, stream>>>(x, ld, out, rows, rpb);
    AVS_LAUNCH_CHECK("colsum");
    return 0;
}

// y[n] += alpha * sum_k x[k] * W[k][n]   (x fp32 [K], W bf16 [K][N] with leading dimension ld, y fp32 [N]; N % 256 == 0, K % 32 == 0).
// The value third of the qkv bias gradient: rows of the softmax sum to one, so d(loss)/d(b_v) = column sum of dO, and dO = dY.W_proj
// gives colsum(dO) = colsum(dY).W_proj = (proj bias gradient).W_proj - a [D] x [D, D] product of quantities the backward already
// holds, instead of a pass over the [rows, 3D] qkv gradient (the key third is identically zero: a key bias shifts every score of a
// query alike).  Attention.qkv / proj: /root/reference/src/models/cav_mae_base.py:51,60-77.
__global__ __launch_bounds__(256) void vecmat_bf16_kernel(const float* __restrict__ x, const bf16_t* __restrict__ W, long long ld,
                                                          float* y, float alpha) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 256 + lane * 4;
    const int k0 = blockIdx.y * 32 + wave * 8;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xv = x[k0 + j];
        const uint2 w = *reinterpret_cast<const uint2*>(W + (size_t)(k0 + j) * ld + n);
        acc[0] = fmaf(xv, __uint_as_float(w.x << 16), acc[0]); acc[1] = fmaf(xv, __uint_as_float(w.x & 0xffff0000u), acc[1]);
        acc[2] = fmaf(xv, __uint_as_float(w.y << 16), acc[2]); acc[3] = fmaf(xv, __uint_as_float(w.y & 0xffff0000u), acc[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(y + n + j, alpha * acc[j]);
}

// n such products in ONE launch (a stack's blocks: the value thirds of all their qkv bias gradients at the end of the backward instead of one
// ~10-us launch per block inside it): desc[3 i .. 3 i + 2] = {x, W, y} of product i, all of the same K, N and leading dimension
__global__ __launch_bounds__(256) void vecmat_bf16_batched_kernel(const long long* __restrict__ desc, long long ld, float alpha) {
    const long long* d = desc + 3 * blockIdx.z;
    const float* x = reinterpret_cast<const float*>(d[0]);
    const bf16_t* W = reinterpret_cast<const bf16_t*>(d[1]);
    float* y = reinterpret_cast<float*>(d[2]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 256 + lane * 4;
    const int k0 = blockIdx.y * 32 + wave * 8;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xv = x[k0 + j];
        const uint2 w = *reinterpret_cast<const uint2*>(W + (size_t)(k0 + j) * ld + n);
        acc[0] = fmaf(xv, __uint_as_float(w.x << 16), acc[0]); acc[1] = fmaf(xv, __uint_as_float(w.x & 0xffff0000u), acc[1]);
        acc[2] = fmaf(xv, __uint_as_float(w.y << 16), acc[2]); acc[3] = fmaf(xv, __uint_as_float(w.y & 0xffff0000u), acc[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(y + n + j, alpha * acc[j]);
}

extern "C" int avs_vecmat_bf16_batched(const long long* desc, int n, long long ld, int K, int N, float alpha, hipStream_t stream) {
    AVS_CHECK_ARG(desc && n > 0 && n <= 65535 && K > 0 && N > 0 && (N % 256) == 0 && (K % 32) == 0 && ld >= N && (ld % 4) == 0, "vecmat_batched: N %% 256, K %% 32, n <= 65535");
    AVS_CHECK_ARG(!avs_tuning().det, "vecmat_batched: not in the deterministic mode (one avs_vecmat_bf16 per product there)");
    vecmat_bf16_batched_kernel<<<dim3(N / 256, K / 32, n), 256, 0, stream>>>(desc, ld, alpha);
    AVS_LAUNCH_CHECK("vecmat_batched");
    return 0;
}

// the same with one writer per element (AvsTuning::det): a block owns 256 columns and walks all of K, its four waves fold through LDS in a fixed order
__global__ __launch_bounds__(256) void vecmat_bf16_det_kernel(const float* __restrict__ x, const bf16_t* __restrict__ W, long long ld,
                                                              float* y, int K, float alpha) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 256 + lane * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = wave * 8; k0 < K; k0 += 32)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = x[k0 + j];
            const uint2 w = *reinterpret_cast<const uint2*>(W + (size_t)(k0 + j) * ld + n);
            acc[0] = fmaf(xv, __uint_as_float(w.x << 16), acc[0]); acc[1] = fmaf(xv, __uint_as_float(w.x & 0xffff0000u), acc[1]);
            acc[2] = fmaf(xv, __uint_as_float(w.y << 16), acc[2]); acc[3] = fmaf(xv, __uint_as_float(w.y & 0xffff0000u), acc[3]);
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][lane * 4 + j] = acc[j];
    __syncthreads();
    const int c = threadIdx.x;
    y[blockIdx.x * 256 + c] += alpha * ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
}

extern "C" int avs_vecmat_bf16(const float* x, const bf16_t* W, long long ld, float* y, int K, int N, float alpha, hipStream_t stream) {
    AVS_CHECK_ARG(x && W && y && K > 0 && N > 0 && (N % 256) == 0 && (K % 32) == 0 && ld >= N && (ld % 4) == 0, "vecmat: N %% 256, K %% 32");
    if (avs_tuning().det) vecmat_bf16_det_kernel<<<N / 256, 256, 0, stream>>>(x, W, ld, y, K, alpha);
    else
    vecmat_bf16_kernel<<<dim3(N / 256, K / 32), 256, 0, stream>>>(x, W, ld, y, alpha);
    AVS_LAUNCH_CHECK("vecmat");
    return 0;
}

extern "C" int avs_unshuffle_fwd(const float* x, const int* src_row, const int* pos_row, const uint8_t* row_mod,
                                 const float* mask_token, const float* pos_a, const float* pos_v, int La,
                                 const float* mod_a, const float* mod_v, float* out, int rows, int D, hipStream_t stream) {
    AVS_CHECK_ARG(rows > 0 && (D % 4) == 0 && x && src_row && pos_row && row_mod && out, "unshuffle_fwd: bad args");
    unshuffle_fwd_kernel<<<rows, 128, 0, stream>>>(x, src_row, pos_row, row_mod, mask_token, pos_a, pos_v, La, mod_a, mod_v,
                                                   out, rows, D);
    AVS_LAUNCH_CHECK("unshuffle_fwd");
    return 0;
}

// The three sums over decoder rows that unshuffle_bwd_kernel forms with atomics - mask-token gradient (rows with no encoder source), the two
// modality-embedding gradients - with ONE writer per element and a fixed order (AvsTuning::det): a block owns 64 columns, its four row lanes
// stride the positions and fold through LDS.
__global__ __launch_bounds__(256) void unshuffle_tokens_det_kernel(const float* __restrict__ dout, const int* __restrict__ src_row, int B, int T, int La, int Lv,
                                                                   float* dmask, float* dmod_a, float* dmod_v, int D, const int* __restrict__ row_of_pos) {
    __shared__ float red[3][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int Ltot = La + T * Lv;
    float sm = 0.f, sa = 0.f, sv = 0.f;
    if (c < D)
        for (int rp = rl; rp < B * Ltot; rp += 4) {
            const int r = row_of_pos ? row_of_pos[rp] : rp;
            const float g = dout[(size_t)r * D + c];
            if (rp % Ltot < La) sa += g; else sv += g;
            if (src_row[r] < 0) sm += g;
        }
    red[0][rl][threadIdx.x & 63] = sm; red[1][rl][threadIdx.x & 63] = sa; red[2][rl][threadIdx.x & 63] = sv;
    __syncthreads();
    if (threadIdx.x < 64 && c < D) {
        const int t = threadIdx.x;
        dmask[c] += (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]);
        dmod_a[c] += (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]);
        dmod_v[c] += (red[2][0][t] + red[2][1][t]) + (red[2][2][t] + red[2][3][t]);
    }
}

extern "C" int avs_unshuffle_bwd_map(const float* dout, const int* src_row, int B, int T, int La, int Lv, float* dx,
                                     float* dpos_a, float* dpos_v, float* dmask, float* dmod_a, float* dmod_v, int D,
                                     const int* row_of_pos, hipStream_t stream) {
    AVS_CHECK_ARG(B > 0 && T > 0 && (D % 4) == 0 && dout && src_row && dx, "unshuffle_bwd: bad args");
    const int det = avs_tuning().det;
    unshuffle_bwd_kernel<<<La + Lv, 128 * UB_G, 0, stream>>>(dout, src_row, B, T, La, Lv, dx, dpos_a, dpos_v, dmask, dmod_a, dmod_v, D, row_of_pos, det);
    if (det) unshuffle_tokens_det_kernel<<<ceil_div(D, 64), 256, 0, stream>>>(dout, src_row, B, T, La, Lv, dmask, dmod_a, dmod_v, D, row_of_pos);
    AVS_LAUNCH_CHECK("unshuffle_bwd");
    return 0;
}

extern "C" int avs_unshuffle_bwd(const float* dout, const int* src_row, int B, int T, int La, int Lv, float* dx,
                                 float* dpos_a, float* dpos_v, float* dmask, float* dmod_a, float* dmod_v, int D,
                                 hipStream_t stream) {
    return avs_unshuffle_bwd_map(dout, src_row, B, T, La, Lv, dx, dpos_a, dpos_v, dmask, dmod_a, dmod_v, D, nullptr, stream);
}

// out[r, 0:cols] = src_row[r] >= 0 ? in[src_row[r], 0:cols] : 0  (bf16 rows of `cols` values, a multiple of 8; leading dimensions in elements).
// in == NULL: only the rows with src_row[r] < 0 are written (zeroed), the others are left as they are.  Used by the decoder's last block in
// the pruned form (engine.Stack): the compact gradient of the scored rows back into the packed row numbering (zeros for the other rows), and the
// zero query gradient of the rows that were keys / values only.
__global__ void expand_rows_kernel(const bf16_t* __restrict__ in, long long ld_in, const int* __restrict__ src_row, bf16_t* __restrict__ out,
                                   long long ld_out, int rows, int cols) {
    const int per = cols / 8;                                  // 16-byte chunks per row
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)rows * per; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / per), c = (int)(i - (long long)r * per);
        const int s = src_row[r];
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (s >= 0) {
            if (!in) continue;
            v = *reinterpret_cast<const uint4*>(in + (size_t)s * ld_in + c * 8);
        }
        *reinterpret_cast<uint4*>(out + (size_t)r * ld_out + c * 8) = v;
    }
}

extern "C" int avs_expand_rows_bf16(const bf16_t* in, long long ld_in, const int* src_row, bf16_t* out, long long ld_out, int rows, int cols,
                                    hipStream_t stream) {
    AVS_CHECK_ARG(src_row && out && rows > 0 && cols > 0 && (cols % 8) == 0 && (ld_out % 8) == 0 && ld_out >= cols && (!in || ((ld_in % 8) == 0 && ld_in >= cols)),
                  "expand_rows: bad args (rows=%d cols=%d)", rows, cols);
    expand_rows_kernel<<<grid_1d((long long)rows * (cols / 8), 256), 256, 0, stream>>>(in, ld_in, src_row, out, ld_out, rows, cols);
    AVS_LAUNCH_CHECK("expand_rows");
    return 0;
}

extern "C" int avs_segment_mean_fwd(const float* y, const int* seg_start, float* reps, int nseg, int D, const int* row_map, hipStream_t stream) {
    AVS_CHECK_ARG(nseg > 0 && (D % 4) == 0, "segment_mean_fwd: bad args");
    segment_mean_fwd_kernel<<<dim3(nseg, ceil_div(D, 128)), 256, 0, stream>>>(y, seg_start, reps, D, row_map);
    AVS_LAUNCH_CHECK("segment_mean_fwd");
    return 0;
}

extern "C" int avs_segment_mean_bwd(const float* dreps, const int* seg_start, float* dy, int nseg, int D, float scale, const int* row_map,
                                    hipStream_t stream) {
    AVS_CHECK_ARG(nseg > 0 && (D % 4) == 0, "segment_mean_bwd: bad args");
    segment_mean_bwd_kernel<<<nseg, 256, 0, stream>>>(dreps, seg_start, dy, D, scale, row_map);
    AVS_LAUNCH_CHECK("segment_mean_bwd");
    return 0;
}

extern "C" int avs_transpose_bf16(const bf16_t* in, bf16_t* out, int R, int C, hipStream_t stream) {
    AVS_CHECK_ARG(R > 0 && C > 0 && in && out, "transpose: bad args");
    transpose_bf16_kernel<<<dim3(ceil_div(C, 64), ceil_div(R, 64)), dim3(64, 4), 0, stream>>>(in, out, R, C);
    AVS_LAUNCH_CHECK("transpose");
    return 0;
}

extern "C" int avs_transpose_batched(const long long* desc, const int* tile_map, int ntiles, hipStream_t stream) {
    AVS_CHECK_ARG(desc && tile_map && ntiles > 0, "transpose_batched: bad args");
    transpose_batched_kernel<<<ntiles, dim3(64, 4), 0, stream>>>(desc, tile_map);
    AVS_LAUNCH_CHECK("transpose_batched");
    return 0;
}

extern "C" int avs_cast_bf16(const float* x, bf16_t* y, long long n, hipStream_t stream) {
    AVS_CHECK_ARG(n > 0 && x && y, "cast_bf16: bad args");
    cast_bf16_kernel<<<grid_1d(n / 4 + 1, 256), 256, 0, stream>>>(x, y, (size_t)n);
    AVS_LAUNCH_CHECK("cast_bf16");
    return 0;
}

extern "C" int avs_adam(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, long long n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int step, float grad_scale, hipStream_t stream) {
    AVS_CHECK_ARG(n > 0 && (n % 4) == 0 && step >= 1 && p && g && m && v, "adam: n must be a positive multiple of 4, step>=1");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);          // torch computes these in double
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    adam_kernel<<<grid_1d(n / 4, 256), 256, 0, stream>>>(p, g, m, v, p_bf16, (size_t)n, lr, beta1, beta2, eps, weight_decay,
                                                         (float)bc1, (float)sqrt(bc2), grad_scale);
    AVS_LAUNCH_CHECK("adam");
    return 0;
}

extern "C" int avs_adam_dev(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, long long n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, const int* step_dev, float grad_scale, hipStream_t stream) {
    AVS_CHECK_ARG(n > 0 && (n % 4) == 0 && step_dev && p && g && m && v, "adam_dev: n must be a positive multiple of 4, step_dev a device pointer");
    adam_dev_kernel<<<grid_1d(n / 4, 256), 256, 0, stream>>>(p, g, m, v, p_bf16, (size_t)n, lr, beta1, beta2, eps, weight_decay, step_dev, grad_scale);
    AVS_LAUNCH_CHECK("adam_dev");
    return 0;
}
