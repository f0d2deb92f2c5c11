// Upsampler tail: PixelShuffle + bicubic skip + clamp (forward), clamp mask + un-shuffle (backward).
#include "common.h"

namespace nvq {

// Keys cubic convolution, A = -0.75 (PyTorch upsample_bicubic2d, align_corners=False)
__device__ __forceinline__ float cc1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__device__ __forceinline__ void cubic_taps(int dst, float scale, int size, int idx[4], float w[4]) {
    const float real = scale * ((float)dst + 0.5f) - 0.5f;
    const float fl = floorf(real);
    const float t = real - fl;
    const int i0 = (int)fl;
    w[0] = cc2(t + 1.f);
    w[1] = cc1(t);
    w[2] = cc1(1.f - t);
    w[3] = cc2((1.f - t) + 1.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) idx[k] = min(max(i0 - 1 + k, 0), size - 1);
}

// one thread per HR pixel (b, oy, ox); loops over image channels
__global__ __launch_bounds__(256) void shuffle_bicubic_clamp_kernel(const float* __restrict__ u, int u_ld,
                                                                     const float* __restrict__ frames, int T,
                                                                     int t_center, int Cimg, int H, int W, int s,
                                                                     float scale, float* __restrict__ out,
                                                                     uint8_t* __restrict__ pass, long total) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const int OW = W * s, OH = H * s;
    const long orow = idiv(gid, OW, total);
    const int ox = (int)(gid - orow * OW);
    const int b = (int)idiv(orow, OH, total);
    const int oy = (int)(orow - (long)b * OH);
    const int h = oy / s, i = oy - h * s, w = ox / s, j = ox - w * s;
    int xi[4], yi[4];
    float wx[4], wy[4];
    cubic_taps(ox, scale, W, xi, wx);
    cubic_taps(oy, scale, H, yi, wy);
    const float* up = u + ((size_t)(b * H + h) * W + w) * u_ld + i * s + j;
    for (int c = 0; c < Cimg; ++c) {
        const float* img = frames + ((size_t)(b * T + t_center) * Cimg + c) * H * W;
        float bic = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float* row = img + (size_t)yi[a] * W;
            const float r = wx[0] * row[xi[0]] + wx[1] * row[xi[1]] + wx[2] * row[xi[2]] + wx[3] * row[xi[3]];
            bic += wy[a] * r;
        }
        const float pre = bic + up[c * s * s];
        const size_t o = ((size_t)(b * Cimg + c) * OH + oy) * OW + ox;
        pass[o] = (pre >= 0.f && pre <= 1.f) ? 1 : 0;
        out[o] = fminf(fmaxf(pre, 0.f), 1.f);
    }
}

// plain nn.PixelShuffle(s) (efficient_layers.py:101-106) for the stand-alone PixelShuffleUpsampler module:
// img[b, c, h*s+i, w*s+j] <-> u[b, h, w, c*s*s + i*s + j]; one thread per HR pixel, coalesced on the image side.
// backward != 0: u is written from img (channels beyond C*s*s zero-filled by the thread of phase (0,0)).
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(float* __restrict__ u, int u_ld, int C, int H, int W, int s,
                                                            float* __restrict__ img, long total, int backward) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const int OW = W * s, OH = H * s;
    const long orow = idiv(gid, OW, total);
    const int ox = (int)(gid - orow * OW);
    const int b = (int)idiv(orow, OH, total);
    const int oy = (int)(orow - (long)b * OH);
    const int h = oy / s, i = oy - h * s, w = ox / s, j = ox - w * s;
    float* up = u + ((size_t)(b * H + h) * W + w) * u_ld;
    for (int c = 0; c < C; ++c) {
        const size_t o = ((size_t)(b * C + c) * OH + oy) * OW + ox;
        if (backward) up[c * s * s + i * s + j] = img[o];
        else img[o] = up[c * s * s + i * s + j];
    }
    if (backward && i == 0 && j == 0)
        for (int k = C * s * s; k < u_ld; ++k) up[k] = 0.f;
}

// one thread per LR pixel.  S > 0 (S = 2, 3, 4 with RGB frames, du_ld = pad4(3 S^2), aligned pointers): the pixel's S x S block
// of every channel is read row by row as 8- / 16-byte loads (S = 2 / 4; scalar for S = 3) and its 3 S^2 values leave as
// 16-byte stores - one 4-byte store per value at a du_ld * 4-byte lane stride ran at 0.9 TB/s.  S = 0: any shape.
template <int S>
__global__ __launch_bounds__(256) void shuffle_clamp_bwd_kernel(const float* __restrict__ dout,
                                                                 const uint8_t* __restrict__ pass, int Cimg, int H,
                                                                 int W, int s, float* __restrict__ du, int du_ld,
                                                                 long total) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const long lrow = idiv(gid, W, total);
    const int w = (int)(gid - lrow * W);
    const int b = (int)idiv(lrow, H, total);
    const int h = (int)(lrow - (long)b * H);
    float* dp = du + (size_t)gid * du_ld;
    if constexpr (S > 0) {
        constexpr int K = 3 * S * S, KP = (K + 3) / 4 * 4;
        const int OW = W * S, OH = H * S;
        float v[KP];
#pragma unroll
        for (int k = K; k < KP; ++k) v[k] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const size_t o = ((size_t)(b * 3 + c) * OH + h * S + i) * OW + w * S;
                float d[S];
                uint8_t m[S];
                if constexpr (S == 2) {
                    const float2 d2 = *reinterpret_cast<const float2*>(dout + o);
                    const uchar2 p2 = *reinterpret_cast<const uchar2*>(pass + o);
                    d[0] = d2.x; d[1] = d2.y; m[0] = p2.x; m[1] = p2.y;
                } else if constexpr (S == 4) {
                    const float4 d4 = *reinterpret_cast<const float4*>(dout + o);
                    const uchar4 p4 = *reinterpret_cast<const uchar4*>(pass + o);
                    d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w; m[0] = p4.x; m[1] = p4.y; m[2] = p4.z; m[3] = p4.w;
                } else {
#pragma unroll
                    for (int j = 0; j < S; ++j) { d[j] = dout[o + j]; m[j] = pass[o + j]; }
                }
#pragma unroll
                for (int j = 0; j < S; ++j) v[c * S * S + i * S + j] = m[j] ? d[j] : 0.f;
            }
#pragma unroll
        for (int q = 0; q < KP / 4; ++q) st4(dp + 4 * q, make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]));
    } else {
        const int OW = W * s, OH = H * s;
        const int K = Cimg * s * s;
        for (int k = 0; k < K; ++k) {
            const int c = k / (s * s), r = k - c * s * s, i = r / s, j = r - i * s;
            const size_t o = ((size_t)(b * Cimg + c) * OH + h * s + i) * OW + w * s + j;
            dp[k] = pass[o] ? dout[o] : 0.f;
        }
        for (int k = K; k < du_ld; ++k) dp[k] = 0.f;
    }
}

// out = strength * sr + (1 - strength) * bicubic(frames[:, t_center])   (EnhancementEngine strength < 1 blend)
__global__ __launch_bounds__(256) void bicubic_blend_kernel(const float* __restrict__ sr,
                                                            const float* __restrict__ frames, int T, int t_center,
                                                            int Cimg, int H, int W, int s, float scale, float strength,
                                                            float* __restrict__ out, long total) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const int OW = W * s, OH = H * s;
    const long orow = idiv(gid, OW, total);
    const int ox = (int)(gid - orow * OW);
    const int b = (int)idiv(orow, OH, total);
    const int oy = (int)(orow - (long)b * OH);
    int xi[4], yi[4];
    float wx[4], wy[4];
    cubic_taps(ox, scale, W, xi, wx);
    cubic_taps(oy, scale, H, yi, wy);
    for (int c = 0; c < Cimg; ++c) {
        const float* img = frames + ((size_t)(b * T + t_center) * Cimg + c) * H * W;
        float bic = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float* row = img + (size_t)yi[a] * W;
            bic += wy[a] * (wx[0] * row[xi[0]] + wx[1] * row[xi[1]] + wx[2] * row[xi[2]] + wx[3] * row[xi[3]]);
        }
        const size_t o = ((size_t)(b * Cimg + c) * OH + oy) * OW + ox;
        out[o] = strength * sr[o] + (1.f - strength) * bic;
    }
}

}  // namespace nvq

using namespace nvq;

extern "C" {

int nvq_shuffle_bicubic_clamp(const float* u, int u_ld, const float* frames, int B, int T, int t_center, int Cimg,
                              int H, int W, int s, float* out, uint8_t* pass, void* stream) {
    NVQ_REQUIRE(s >= 1 && s <= 8 && u_ld >= Cimg * s * s && t_center >= 0 && t_center < T, "shuffle_bicubic_clamp: args");
    const long total = (long)B * H * s * W * s;
    const float scale = (float)(1.0 / (double)s);
    hipLaunchKernelGGL(shuffle_bicubic_clamp_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, u,
                       u_ld, frames, T, t_center, Cimg, H, W, s, scale, out, pass, total);
    return check_launch("shuffle_bicubic_clamp");
}

int nvq_bicubic_blend(const float* sr, const float* frames, int B, int T, int t_center, int Cimg, int H, int W, int s,
                      float strength, float* out, void* stream) {
    NVQ_REQUIRE(s >= 1 && s <= 8 && t_center >= 0 && t_center < T, "bicubic_blend: args");
    const long total = (long)B * H * s * W * s;
    hipLaunchKernelGGL(bicubic_blend_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, sr, frames, T,
                       t_center, Cimg, H, W, s, (float)(1.0 / (double)s), strength, out, total);
    return check_launch("bicubic_blend");
}

int nvq_shuffle_clamp_backward(const float* dout, const uint8_t* pass, int B, int Cimg, int H, int W, int s,
                               float* du, int du_ld, void* stream) {
    NVQ_REQUIRE(s >= 1 && s <= 8 && du_ld >= Cimg * s * s, "shuffle_clamp_backward: args");
    const long total = (long)B * H * W;
    const bool fast = Cimg == 3 && s >= 2 && s <= 4 && du_ld == (3 * s * s + 3) / 4 * 4 && aligned16(du) &&
                      (reinterpret_cast<uintptr_t>(dout) & 15) == 0 && (reinterpret_cast<uintptr_t>(pass) & 3) == 0;
#define NVQ_SCB(S_) hipLaunchKernelGGL(shuffle_clamp_bwd_kernel<S_>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, \
                                       dout, pass, Cimg, H, W, s, du, du_ld, total)
    if (fast && s == 2) NVQ_SCB(2); else if (fast && s == 3) NVQ_SCB(3); else if (fast && s == 4) NVQ_SCB(4); else NVQ_SCB(0);
#undef NVQ_SCB
    return check_launch("shuffle_clamp_backward");
}

int nvq_pixel_shuffle(float* u, int u_ld, int B, int C, int H, int W, int s, float* img, int backward, void* stream) {
    NVQ_REQUIRE(s >= 1 && s <= 8 && u_ld >= C * s * s, "pixel_shuffle: args");
    const long total = (long)B * H * s * W * s;
    hipLaunchKernelGGL(pixel_shuffle_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, u, u_ld, C, H, W, s, img,
                       total, backward);
    return check_launch("pixel_shuffle");
}

}  // extern "C"
