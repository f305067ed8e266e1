// k10 - input normalisation on the device: what the reference does per sample on the host in
// its dataloader, so that raw AudioSet-shaped tensors (fp32 fbank, uint8 frames) can be handed to the model directly.
//
//   audio  (/root/reference/src/dataloader.py:505-513): fbank = (fbank - norm_mean) / norm_std; with `noise`:
//          fbank += rand(T, F) * amp  (amp = np.random.rand() / 10, one scalar per sample), then
//          fbank = roll(fbank, shift, dim 0)  (shift in [-T, T) per sample).
//   frames (:461-462 + my_normalize :152-155): image / 255, then (x - mean_c) / std_c per channel.
//
// HBM-bound, one pass each.  The noise is a counter-based stream (Philox4x32-10 keyed by the caller's seed, counter =
// (element, sample)): reproducible for a given seed whatever the launch shape; the reference's torch.rand stream cannot
// be reproduced bit-for-bit on a device, its DISTRIBUTION (uniform [0,1) scaled by amp) is what is kept.
#include "common.h"

// out[b, t, :] = (in[b, (t - shift_b) mod T, :] - mean) / std + amp_b * U[b, (t - shift_b) mod T, :]
// grid (B, ceil(T*F/4 / 256)); F % 4 == 0
__global__ __launch_bounds__(256) void normalize_audio_kernel(const float* __restrict__ in, float* __restrict__ out, int T, int F,
                                                              float mean, float inv_std, const int* __restrict__ shift,
                                                              const float* __restrict__ amp, uint32_t seed_lo, uint32_t seed_hi) {
    const int b = blockIdx.x;
    const int i4 = blockIdx.y * 256 + threadIdx.x;              // float4 index inside the sample
    if (i4 >= T * F / 4) return;
    const int t = (i4 * 4) / F, f = (i4 * 4) - t * F;
    int s = shift ? shift[b] % T : 0;
    if (s < 0) s += T;
    int ts = t - s;
    if (ts < 0) ts += T;
    const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)b * T + ts) * F + f);
    float4 o = make_float4((v.x - mean) * inv_std, (v.y - mean) * inv_std, (v.z - mean) * inv_std, (v.w - mean) * inv_std);
    const float a = amp ? amp[b] : 0.f;
    if (a != 0.f) {                                             // sample-uniform
        const uint32_t e = (uint32_t)(ts * F + f);
        o.x += a * ((float)(xf_philox(e + 0, (uint32_t)b, seed_lo, seed_hi) >> 8) * (1.0f / 16777216.0f));
        o.y += a * ((float)(xf_philox(e + 1, (uint32_t)b, seed_lo, seed_hi) >> 8) * (1.0f / 16777216.0f));
        o.z += a * ((float)(xf_philox(e + 2, (uint32_t)b, seed_lo, seed_hi) >> 8) * (1.0f / 16777216.0f));
        o.w += a * ((float)(xf_philox(e + 3, (uint32_t)b, seed_lo, seed_hi) >> 8) * (1.0f / 16777216.0f));
    }
    *reinterpret_cast<float4*>(out + ((size_t)b * T + t) * F + f) = o;
}

// out[n, c, :] = (in[n, c, :] / 255 - mean_c) / std_c ; plane = H*W, plane % 4 == 0; grid (N*3, ceil(plane/4 / 256))
__global__ __launch_bounds__(256) void normalize_frames_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int plane,
                                                               float m0, float m1, float m2, float s0, float s1, float s2) {
    const int nc = blockIdx.x, c = nc % 3;
    const int i4 = blockIdx.y * 256 + threadIdx.x;
    if (i4 >= plane / 4) return;
    const float mean = c == 0 ? m0 : c == 1 ? m1 : m2;
    const float inv = 1.0f / (c == 0 ? s0 : c == 1 ? s1 : s2);
    const uint32_t p = *reinterpret_cast<const uint32_t*>(in + (size_t)nc * plane + (size_t)i4 * 4);
    float4 o;
    o.x = ((float)(p & 0xff) * (1.0f / 255.0f) - mean) * inv;
    o.y = ((float)((p >> 8) & 0xff) * (1.0f / 255.0f) - mean) * inv;
    o.z = ((float)((p >> 16) & 0xff) * (1.0f / 255.0f) - mean) * inv;
    o.w = ((float)(p >> 24) * (1.0f / 255.0f) - mean) * inv;
    *reinterpret_cast<float4*>(out + (size_t)nc * plane + (size_t)i4 * 4) = o;
}

extern "C" int avs_normalize_audio(const float* in, float* out, int B, int T, int F, float mean, float std, const int* shift,
                                   const float* amp, unsigned long long seed, hipStream_t stream) {
    AVS_CHECK_ARG(in && out && in != out && B > 0 && T > 0 && F > 0 && (F % 4) == 0 && std != 0.f,
                  "normalize_audio: bad arguments (out of place; F %% 4 == 0; std != 0)");
    dim3 grid(B, ceil_div(T * F / 4, 256));
    normalize_audio_kernel<<<grid, 256, 0, stream>>>(in, out, T, F, mean, 1.0f / std, shift, amp, (uint32_t)seed, (uint32_t)(seed >> 32));
    AVS_LAUNCH_CHECK("normalize_audio");
    return 0;
}

extern "C" int avs_normalize_frames_u8(const uint8_t* in, float* out, int n_images, int plane, const float* mean3,
                                       const float* std3, hipStream_t stream) {
    AVS_CHECK_ARG(in && out && n_images > 0 && plane > 0 && (plane % 4) == 0 && mean3 && std3, "normalize_frames: bad arguments (H*W %% 4 == 0)");
    AVS_CHECK_ARG(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "normalize_frames: zero std");
    dim3 grid(n_images * 3, ceil_div(plane / 4, 256));
    normalize_frames_kernel<<<grid, 256, 0, stream>>>(in, out, plane, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    AVS_LAUNCH_CHECK("normalize_frames");
    return 0;
}
