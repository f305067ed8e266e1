// k10 (patchify-on-the-fly masked MSE) and k11 (bidirectional InfoNCE over all audio/visual pairs).
#include "common.h"

// ---------------------------------------------------------------------------------------------------
// k10.  forward_mae_loss (/root/reference/src/models/cav_mae_base.py:663-683) with patchify (:343-351) folded
// into the index arithmetic: target[b, l, (p*16+q)*C + c] = img[b, c, gy*16+p, gx*16+q]; for audio the image is
// the transposed spectrogram (:666-668) so target[b, f*tP+t, p*16+q] = a[b, t*16+q, f*16+p].
// row_loss[r] = mask[r] * mean_e (pred - target)^2 ; loss = sum_r row_loss / sum(mask).
// S: patch stride = the S x S corner of the 16 x 16 positions that is scored (16: all of them; 14: config.stride, ViT-H/14);
// `valid` says whether element e belongs to it.
__device__ __forceinline__ float mae_target(const void* __restrict__ inp, int audio, int r, int e, int L, int C, int H,
                                            int W, int G, int S, const InXf& xf, bool& valid) {
    const int n = r / L, l = r - n * L;
    if (audio) {                       // H = time frames, W = mel bins, G = time patches
        const int f = l / G, t = l - f * G;
        const int p = e >> 4, q = e & 15;
        valid = p < S && q < S;
        return valid ? xf_audio(reinterpret_cast<const float*>(inp), xf, n, t * S + q, f * S + p, H, W) : 0.f;
    }
    const int gy = l / G, gx = l - gy * G;
    const int c = e % C, pq = e / C;
    const int p = pq >> 4, q = pq & 15;
    valid = p < S && q < S;
    return valid ? xf_video(inp, xf, (((size_t)n * C + c) * H + gy * S + p) * W + gx * S + q, c) : 0.f;
}

// row_id (may be NULL): prediction row pr scores token row_id[pr] - id_base of the [N * L] (sample, token) numbering the mask and the targets are
// indexed by - COMPACT predictions: only the rows whose mask is 1 are computed at all (maskplan.hip: pred_id).  NULL: pr is that index itself.
// One WAVE per prediction row, four rows per block (round 6: a 256-thread block per row spent its time in the block-wide reduction - 118 k blocks,
// 0.32 ms per direction at 1.9 TB/s): a lane takes elements lane, lane + 64, ..., the row sum is a wave reduction.
__global__ __launch_bounds__(256) void mae_loss_fwd_kernel(const float* __restrict__ pred, const void* __restrict__ inp,
                                                          const float* __restrict__ mask, float* __restrict__ row_loss, int rows, int audio, int L, int C,
                                                          int H, int W, int G, int P, int S, InXf xf, const int* __restrict__ row_id, int id_base) {
    const int pr = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pr >= rows) return;
    const int r = row_id ? row_id[pr] - id_base : pr;
    const float m = mask[r];
    if (m == 0.f) {
        if (lane == 0) row_loss[pr] = 0.f;
        return;
    }
    float s = 0.f;
    for (int e = lane; e < P; e += 64) {
        bool valid;
        const float tg = mae_target(inp, audio, r, e, L, C, H, W, G, S, xf, valid);
        const float d = valid ? pred[(size_t)pr * P + e] - tg : 0.f;
        s += d * d;
    }
    s = wave_sum(s);
    if (lane == 0) row_loss[pr] = s / (float)(P / 256 * S * S) * m;      // mean over the scored elements
}

// deterministic single-block sum: out[0] = scale * sum(x); optionally total[0] = (total_init ? 0 : total[0]) + out[0]
// (loss_mae = loss_mae_a + loss_mae_v, cav_mae_base.py:707)
__global__ void sum_scale_kernel(const float* __restrict__ x, int n, float scale, float* out, float* total, int total_init) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        out[0] = t * scale;
        if (total) total[0] = (total_init ? 0.f : total[0]) + t * scale;
    }
}

// dpred[r, e] = g * 2 (pred - target) mask[r] / (P * nmask)   (bf16: operand of the prediction-head GEMMs)
__global__ __launch_bounds__(256) void mae_loss_bwd_kernel(const float* __restrict__ pred, const void* __restrict__ inp,
                                                          const float* __restrict__ mask, const float* __restrict__ gout, bf16_t* __restrict__ dpred, int rows,
                                                          int audio, int L, int C, int H, int W, int G, int P, int S, float inv_nmask, InXf xf,
                                                          const int* __restrict__ row_id, int id_base) {
    const int pr = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;          // one wave per prediction row (see the forward)
    if (pr >= rows) return;
    const int r = row_id ? row_id[pr] - id_base : pr;
    const float m = mask[r];
    const float k = gout[0] * 2.0f * m * inv_nmask / (float)(P / 256 * S * S);
    for (int e = lane; e < P; e += 64) {
        float d = 0.f;
        if (m != 0.f) {
            bool valid;
            const float tg = mae_target(inp, audio, r, e, L, C, H, W, G, S, xf, valid);
            if (valid) d = k * (pred[(size_t)pr * P + e] - tg);
        }
        dpred[(size_t)pr * P + e] = f2bf(d);
    }
}

// ---------------------------------------------------------------------------------------------------
// k11.  forward_contrastive(bidirect_contrast=True) (:641-661).
// F.normalize: x / max(||x||, 1e-12)
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ xn, float* __restrict__ norm, int D) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < D; c += blockDim.x) { const float v = x[(size_t)r * D + c]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float n = fmaxf(sqrtf((red[0] + red[1]) + (red[2] + red[3])), 1e-12f);
    for (int c = threadIdx.x; c < D; c += blockDim.x) xn[(size_t)r * D + c] = x[(size_t)r * D + c] / n;
    if (threadIdx.x == 0) norm[r] = n;
}

// dx = scale * (dxn - xn (xn . dxn)) / n
__global__ void l2norm_bwd_kernel(const float* __restrict__ dxn, const float* __restrict__ xn, const float* __restrict__ norm,
                                  float* __restrict__ dx, int D, float scale) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < D; c += blockDim.x) s += dxn[(size_t)r * D + c] * xn[(size_t)r * D + c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float dot = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = scale / norm[r];
    for (int c = threadIdx.x; c < D; c += blockDim.x)
        dx[(size_t)r * D + c] = (dxn[(size_t)r * D + c] - xn[(size_t)r * D + c] * dot) * inv;
}

// Small exact-fp32 GEMM on the f32-input matrix cores: C[m,n] = alpha * sum_k A(m,k) B(k,n) with arbitrary element
// strides (so A.B^T, A^T.B and A.B all map here).  One wave per 32x32 tile, v_mfma_f32_32x32x2_f32
// (lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]); the all-pairs similarity A.V^T of :647 and its two gradient
// products are genuine dense contractions, but tiny (<= 512 x 512 x 768), so no LDS staging.
__global__ __launch_bounds__(64) void gemm_f32_small_kernel(const float* __restrict__ A, long long sam, long long sak,
                                                            const float* __restrict__ B, long long sbk, long long sbn,
                                                            float* __restrict__ Cm, long long scm, int M, int N, int K,
                                                            float alpha) {
    const int lane = threadIdx.x;
    const int i = blockIdx.y * 32 + (lane & 31);
    const int j = blockIdx.x * 32 + (lane & 31);
    const int kh = lane >> 5;
    f32x16 acc = {0};
    const bool iv = i < M, jv = j < N;
    for (int k = 0; k < K; k += 2) {
        const int kk = k + kh;
        const float a = (iv && kk < K) ? A[i * sam + kk * sak] : 0.f;
        const float b = (jv && kk < K) ? B[kk * sbk + j * sbn] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (!jv) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = blockIdx.y * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < M) Cm[row * scm + j] = alpha * acc[r];
    }
}

// Per index i: column-i statistics of `total` (log_softmax(total, dim=0), :655) and row-i statistics
// (log_softmax(total.t(), dim=0), :656), the diagonal, and the two argmax hits (:657-658; first maximum wins).
// stats[i] = {lse_col, lse_row, diag, hits}
__global__ void infonce_stats_kernel(const float* __restrict__ total, int N, float* __restrict__ stats) {
    __shared__ float redf[4][2];
    __shared__ int redi[4][2];
    const int i = blockIdx.x;
    float mc = -INFINITY, mr = -INFINITY;
    int ac = 0x7fffffff, ar = 0x7fffffff;
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        const float vc = total[(size_t)k * N + i], vr = total[(size_t)i * N + k];
        if (vc > mc) { mc = vc; ac = k; }
        if (vr > mr) { mr = vr; ar = k; }
    }
    // wave argmax with first-index tie break
    for (int o = 32; o > 0; o >>= 1) {
        const float omc = __shfl_xor(mc, o, 64), omr = __shfl_xor(mr, o, 64);
        const int oac = __shfl_xor(ac, o, 64), oar = __shfl_xor(ar, o, 64);
        if (omc > mc || (omc == mc && oac < ac)) { mc = omc; ac = oac; }
        if (omr > mr || (omr == mr && oar < ar)) { mr = omr; ar = oar; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { redf[w][0] = mc; redf[w][1] = mr; redi[w][0] = ac; redi[w][1] = ar; }
    __syncthreads();
    mc = redf[0][0]; ac = redi[0][0]; mr = redf[0][1]; ar = redi[0][1];
    for (int k = 1; k < 4; ++k) {
        if (redf[k][0] > mc || (redf[k][0] == mc && redi[k][0] < ac)) { mc = redf[k][0]; ac = redi[k][0]; }
        if (redf[k][1] > mr || (redf[k][1] == mr && redi[k][1] < ar)) { mr = redf[k][1]; ar = redi[k][1]; }
    }
    __syncthreads();
    float sc = 0.f, sr = 0.f;
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        sc += expf(total[(size_t)k * N + i] - mc);
        sr += expf(total[(size_t)i * N + k] - mr);
    }
    sc = wave_sum(sc); sr = wave_sum(sr);
    if ((threadIdx.x & 63) == 0) { redf[w][0] = sc; redf[w][1] = sr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sc = (redf[0][0] + redf[1][0]) + (redf[2][0] + redf[3][0]);
        sr = (redf[0][1] + redf[1][1]) + (redf[2][1] + redf[3][1]);
        stats[i * 4 + 0] = mc + logf(sc);
        stats[i * 4 + 1] = mr + logf(sr);
        stats[i * 4 + 2] = total[(size_t)i * N + i];
        stats[i * 4 + 3] = (float)((ac == i) + (ar == i));
    }
}

// out[0] = nce = (mean(lse_col - diag) + mean(lse_row - diag)) / 2 ;  out[1] = c_acc = hits / (2N) ;
// out[2] = weight * nce (loss_c = contrast_loss_weight * loss_c, cav_mae_base.py:735)
__global__ void infonce_reduce_kernel(const float* __restrict__ stats, int N, float weight, float* out) {
    __shared__ float red[4][2];
    float s = 0.f, h = 0.f;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        s += (stats[i * 4 + 0] - stats[i * 4 + 2]) + (stats[i * 4 + 1] - stats[i * 4 + 2]);
        h += stats[i * 4 + 3];
    }
    s = wave_sum(s); h = wave_sum(h);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = h; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        h = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        out[0] = s / (2.0f * N);
        out[1] = h / (2.0f * N);
        out[2] = weight * (s / (2.0f * N));
    }
}

// dtotal[r][c] = coef * (exp(t - lse_col[c]) + exp(t - lse_row[r]) - 2 [r == c]),  coef = g * weight / (2N)
__global__ void infonce_dlogits_kernel(const float* __restrict__ total, const float* __restrict__ stats, int N,
                                       const float* __restrict__ gout, float weight, float* __restrict__ dtotal) {
    const int r = blockIdx.x;
    const float coef = gout[0] * weight / (2.0f * N);
    const float lr = stats[r * 4 + 1];
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        const float t = total[(size_t)r * N + c];
        dtotal[(size_t)r * N + c] = coef * (expf(t - stats[c * 4 + 0]) + expf(t - lr) - (r == c ? 2.0f : 0.0f));
    }
}

// ===================================================================================================
int avs_make_xf(const avs_input_xf_t* x, int want_kind, InXf* out, const char* who);     // elementwise.hip

extern "C" int avs_mae_loss_fwd_id(const float* pred, const void* inp, const float* mask, float* row_loss, float* loss,
                                   float* total, int total_init, int rows, int audio, int L, int C, int H, int W, float nmask,
                                   int stride, const avs_input_xf_t* xf, const int* row_id, int id_base, hipStream_t stream) {
    AVS_CHECK_ARG(rows > 0 && pred && inp && mask && row_loss && loss && nmask > 0 && stride > 0 && stride <= 16, "mae_loss_fwd: bad args");
    InXf x;
    if (int rc = avs_make_xf(xf, audio ? 1 : 2, &x, "mae_loss_fwd")) return rc;
    const int G = audio ? H / stride : W / stride;
    const int P = 256 * (audio ? 1 : C);
    mae_loss_fwd_kernel<<<ceil_div(rows, 4), 256, 0, stream>>>(pred, inp, mask, row_loss, rows, audio, L, C, H, W, G, P, stride, x, row_id, id_base);
    AVS_LAUNCH_CHECK("mae_loss_fwd");
    sum_scale_kernel<<<1, 1024, 0, stream>>>(row_loss, rows, 1.0f / nmask, loss, total, total_init);
    AVS_LAUNCH_CHECK("mae_loss_sum");
    return 0;
}

extern "C" int avs_mae_loss_fwd_s(const float* pred, const void* inp, const float* mask, float* row_loss, float* loss,
                                  float* total, int total_init, int rows, int audio, int L, int C, int H, int W, float nmask,
                                  int stride, const avs_input_xf_t* xf, hipStream_t stream) {
    return avs_mae_loss_fwd_id(pred, inp, mask, row_loss, loss, total, total_init, rows, audio, L, C, H, W, nmask, stride, xf, nullptr, 0, stream);
}

extern "C" int avs_mae_loss_fwd_xf(const float* pred, const void* inp, const float* mask, float* row_loss, float* loss,
                                   float* total, int total_init, int rows, int audio, int L, int C, int H, int W, float nmask,
                                   const avs_input_xf_t* xf, hipStream_t stream) {
    return avs_mae_loss_fwd_s(pred, inp, mask, row_loss, loss, total, total_init, rows, audio, L, C, H, W, nmask, 16, xf, stream);
}

extern "C" int avs_mae_loss_fwd(const float* pred, const float* inp, const float* mask, float* row_loss, float* loss,
                                float* total, int total_init, int rows, int audio, int L, int C, int H, int W, float nmask,
                                hipStream_t stream) {
    return avs_mae_loss_fwd_xf(pred, inp, mask, row_loss, loss, total, total_init, rows, audio, L, C, H, W, nmask, nullptr, stream);
}

extern "C" int avs_mae_loss_bwd_id(const float* pred, const void* inp, const float* mask, const float* gout, bf16_t* dpred,
                                   int rows, int audio, int L, int C, int H, int W, float nmask, int stride, const avs_input_xf_t* xf,
                                   const int* row_id, int id_base, hipStream_t stream) {
    AVS_CHECK_ARG(rows > 0 && pred && inp && mask && gout && dpred && nmask > 0 && stride > 0 && stride <= 16, "mae_loss_bwd: bad args");
    InXf x;
    if (int rc = avs_make_xf(xf, audio ? 1 : 2, &x, "mae_loss_bwd")) return rc;
    const int G = audio ? H / stride : W / stride;
    const int P = 256 * (audio ? 1 : C);
    mae_loss_bwd_kernel<<<ceil_div(rows, 4), 256, 0, stream>>>(pred, inp, mask, gout, dpred, rows, audio, L, C, H, W, G, P, stride, 1.0f / nmask, x, row_id, id_base);
    AVS_LAUNCH_CHECK("mae_loss_bwd");
    return 0;
}

extern "C" int avs_mae_loss_bwd_s(const float* pred, const void* inp, const float* mask, const float* gout, bf16_t* dpred,
                                  int rows, int audio, int L, int C, int H, int W, float nmask, int stride, const avs_input_xf_t* xf,
                                  hipStream_t stream) {
    return avs_mae_loss_bwd_id(pred, inp, mask, gout, dpred, rows, audio, L, C, H, W, nmask, stride, xf, nullptr, 0, stream);
}

extern "C" int avs_mae_loss_bwd_xf(const float* pred, const void* inp, const float* mask, const float* gout, bf16_t* dpred,
                                   int rows, int audio, int L, int C, int H, int W, float nmask, const avs_input_xf_t* xf,
                                   hipStream_t stream) {
    return avs_mae_loss_bwd_s(pred, inp, mask, gout, dpred, rows, audio, L, C, H, W, nmask, 16, xf, stream);
}

extern "C" int avs_mae_loss_bwd(const float* pred, const float* inp, const float* mask, const float* gout, bf16_t* dpred,
                                int rows, int audio, int L, int C, int H, int W, float nmask, hipStream_t stream) {
    return avs_mae_loss_bwd_xf(pred, inp, mask, gout, dpred, rows, audio, L, C, H, W, nmask, nullptr, stream);
}

extern "C" int avs_l2norm_fwd(const float* x, float* xn, float* norm, int rows, int D, hipStream_t stream) {
    AVS_CHECK_ARG(rows > 0 && D > 0 && x && xn && norm, "l2norm_fwd: bad args");
    l2norm_fwd_kernel<<<rows, 256, 0, stream>>>(x, xn, norm, D);
    AVS_LAUNCH_CHECK("l2norm_fwd");
    return 0;
}

extern "C" int avs_l2norm_bwd(const float* dxn, const float* xn, const float* norm, float* dx, int rows, int D, float scale,
                              hipStream_t stream) {
    AVS_CHECK_ARG(rows > 0 && D > 0 && dxn && xn && norm && dx, "l2norm_bwd: bad args");
    l2norm_bwd_kernel<<<rows, 256, 0, stream>>>(dxn, xn, norm, dx, D, scale);
    AVS_LAUNCH_CHECK("l2norm_bwd");
    return 0;
}

extern "C" int avs_gemm_f32_small(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn,
                                  float* C, long long scm, int M, int N, int K, float alpha, hipStream_t stream) {
    AVS_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C, "gemm_f32_small: bad args");
    gemm_f32_small_kernel<<<dim3(ceil_div(N, 32), ceil_div(M, 32)), 64, 0, stream>>>(A, sam, sak, B, sbk, sbn, C, scm, M, N, K, alpha);
    AVS_LAUNCH_CHECK("gemm_f32_small");
    return 0;
}

// total [N,N] -> stats [N,4], out {nce, c_acc, weight * nce}
extern "C" int avs_infonce_fwd(const float* total, float* stats, float* out, int N, float weight, hipStream_t stream) {
    AVS_CHECK_ARG(N > 0 && total && stats && out, "infonce_fwd: bad args");
    infonce_stats_kernel<<<N, 256, 0, stream>>>(total, N, stats);
    AVS_LAUNCH_CHECK("infonce_stats");
    infonce_reduce_kernel<<<1, 256, 0, stream>>>(stats, N, weight, out);
    AVS_LAUNCH_CHECK("infonce_reduce");
    return 0;
}

extern "C" int avs_infonce_dlogits(const float* total, const float* stats, const float* gout, float weight, float* dtotal, int N,
                                   hipStream_t stream) {
    AVS_CHECK_ARG(N > 0 && total && stats && gout && dtotal, "infonce_dlogits: bad args");
    infonce_dlogits_kernel<<<N, 256, 0, stream>>>(total, stats, N, gout, weight, dtotal);
    AVS_LAUNCH_CHECK("infonce_dlogits");
    return 0;
}
