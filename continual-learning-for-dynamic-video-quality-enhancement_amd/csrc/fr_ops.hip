// Generic fp32 NHWC kernels for the layers of FrameRecoveryNet that the SR hot path does not have
// (reference nerve_cl/models/frame_recovery.py:23-446, layers efficient_layers.py:109-151,231-294):
// layout conversion, BatchNorm over an arbitrary channel count (BatchNorm2d / BatchNorm3d in train and eval mode, optional
// residual add and ReLU), max-pooling with PyTorch's first-maximum tie rule, stride-2 subsampling (1x1 stride-2 convs),
// bilinear resize (align_corners = False), depth <-> space (ConvTranspose2d k4 s2 p1 as a phase-packed 3x3 conv), the
// fusion module's softmax-weighted channel means, tanh, the mask blend, and the 7x7 stride-2 stem convolution.
// Every tensor is [N, H, W, ld] fp32 with the logical channel count C <= ld, ld % 4 == 0; channels [C, ld) are kept zero.
// All of them are HBM-bound element-wise / small-reduction kernels: one coalesced pass over their operands.
#include "common.h"

namespace nvq {

// ------------------------------------------------------------------------------------------------ layout
// dst[n, p, dst_coff + c] = src[n * src_nstride + c * HW + p] for c < C, 0 for C <= c < Czero.  64 pixels x 64 channels per
// workgroup through LDS so that both sides are coalesced.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, long src_nstride, int C, int Czero,
                                                           long HW, float* __restrict__ dst, int dst_ld, int dst_coff) {
    __shared__ float tile[64][65];
    const int n = blockIdx.z, c0 = blockIdx.y * 64;
    const long p0 = (long)blockIdx.x * 64;
    const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
    for (int cc = row; cc < 64; cc += 4) {
        const int c = c0 + cc;
        const long p = p0 + lane;
        tile[cc][lane] = (c < C && p < HW) ? src[(size_t)n * src_nstride + (size_t)c * HW + p] : 0.f;
    }
    __syncthreads();
    for (int pp = row; pp < 64; pp += 4) {
        const int c = c0 + lane;
        const long p = p0 + pp;
        if (c < Czero && p < HW) dst[((size_t)n * HW + p) * dst_ld + dst_coff + c] = tile[lane][pp];
    }
}

// dst[n * dst_nstride + c * HW + p] (+)= src[n, p, src_coff + c]
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, int src_ld, int src_coff, int C,
                                                           long HW, float* __restrict__ dst, long dst_nstride) {
    __shared__ float tile[64][65];
    const int n = blockIdx.z, c0 = blockIdx.y * 64;
    const long p0 = (long)blockIdx.x * 64;
    const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
    for (int pp = row; pp < 64; pp += 4) {
        const int c = c0 + lane;
        const long p = p0 + pp;
        tile[pp][lane] = (c < C && p < HW) ? src[((size_t)n * HW + p) * src_ld + src_coff + c] : 0.f;
    }
    __syncthreads();
    for (int cc = row; cc < 64; cc += 4) {
        const int c = c0 + cc;
        const long p = p0 + lane;
        if (c < C && p < HW) dst[(size_t)n * dst_nstride + (size_t)c * HW + p] = tile[lane][cc];
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm, any C
constexpr int BN2_MAXBLK = 1024;

// A thread owns one 4-channel group g of a pixel row; the 256 / G4 threads that share g are summed through LDS.
// part[blk][0][c] = sum a, part[blk][1][c] = sum b over the workgroup's pixel range:
// MODE 0 (forward statistics): a = x, b = x^2.
// MODE 1 (backward): g = dy masked by the ReLU of the forward output (relu != 0: y = bn(x) (+ res) > 0), a = g,
//                    b = g * xhat; g is also written to gout when gout != nullptr.
// Padding channels (>= C) of every tensor are zero, so whole float4 groups are processed without channel masks; the
// per-channel parameter arrays are read through cpar(), which guards the last partial group.
__device__ __forceinline__ float4 cpar(const float* __restrict__ p, int c, int C) {
    return make_float4(p[c], c + 1 < C ? p[c + 1] : 0.f, c + 2 < C ? p[c + 2] : 0.f, c + 3 < C ? p[c + 3] : 0.f);
}

// BF: the tensors' storage type at compile time (as a run-time flag every ldx4 is a branch that waits for its own load: the
// two or three loads of a trip ran one after the other)
template <int MODE, bool BF>
__global__ __launch_bounds__(256) void bn2_partial_kernel(const float* __restrict__ x, int x_ld, int C, long npix,
                                                          const float* __restrict__ dy, int dy_ld,
                                                          const float* __restrict__ res, int res_ld,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int relu, float* __restrict__ gout, int gout_ld,
                                                          float* __restrict__ part) {
    constexpr int bf = BF;
    __shared__ float red[256 * 8];
    const int G4 = (C + 3) >> 2;
    const int rows = 256 / G4;                       // pixel rows handled per iteration (G4 <= 256)
    const int g = threadIdx.x % G4, row = threadIdx.x / G4;
    const int c = 4 * g;
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f), is = m, ga = m, be = m;
        if (MODE == 1) { m = cpar(mean, c, C); is = cpar(invstd, c, C); ga = cpar(gamma, c, C); be = cpar(beta, c, C); }
        for (long p = p0 + row; p < p1; p += rows) {
            const float4 v = ldx4(x, p * x_ld + c, bf);
            if (MODE == 0) {
                s0[0] += v.x; s0[1] += v.y; s0[2] += v.z; s0[3] += v.w;
                s1[0] += v.x * v.x; s1[1] += v.y * v.y; s1[2] += v.z * v.z; s1[3] += v.w * v.w;
            } else {
                const float xh[4] = {(v.x - m.x) * is.x, (v.y - m.y) * is.y, (v.z - m.z) * is.z, (v.w - m.w) * is.w};
                const float4 d = ldx4(dy, p * dy_ld + c, bf);
                float gg[4] = {d.x, d.y, d.z, d.w};
                if (relu) {
                    float y[4] = {xh[0] * ga.x + be.x, xh[1] * ga.y + be.y, xh[2] * ga.z + be.z, xh[3] * ga.w + be.w};
                    if (res) {
                        const float4 r = ldx4(res, p * res_ld + c, bf);
                        y[0] += r.x; y[1] += r.y; y[2] += r.z; y[3] += r.w;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (!(y[k] > 0.f)) gg[k] = 0.f;
                }
                if (gout) stx4(gout, p * gout_ld + c, bf, make_float4(gg[0], gg[1], gg[2], gg[3]));
#pragma unroll
                for (int k = 0; k < 4; ++k) { s0[k] += gg[k]; s1[k] += gg[k] * xh[k]; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[threadIdx.x * 8 + k] = s0[k]; red[threadIdx.x * 8 + 4 + k] = s1[k]; }
    __syncthreads();
    if (row == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = 0.f;
            for (int r = 0; r < rows; ++r) t += red[(r * G4 + g) * 8 + k];
            const int cc = c + (k & 3);
            if (cc < C) part[((size_t)blockIdx.x * 2 + (k >> 2)) * C + cc] = t;
        }
    }
}

// Sum of the workgroups' partials of 16 channels per workgroup: thread (c = tid & 15, j = tid >> 4) adds blocks j, j+16, ...
// in double, the 16 partial sums per channel meet in LDS.  Returns (sum a, sum b) of channel c0 + (tid & 15) to tid < 16.
__device__ __forceinline__ void bn2_reduce16(const float* __restrict__ part, int nblk, int C, double* sh, double& s, double& ss) {
    const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    s = 0.0; ss = 0.0;
    if (c < C)
        for (int b = j; b < nblk; b += 16) {
            s += (double)part[((size_t)b * 2 + 0) * C + c];
            ss += (double)part[((size_t)b * 2 + 1) * C + c];
        }
    sh[threadIdx.x * 2] = s;
    sh[threadIdx.x * 2 + 1] = ss;
    __syncthreads();
    if (threadIdx.x < 16) {
        s = 0.0; ss = 0.0;
        for (int k = 0; k < 16; ++k) { s += sh[(k * 16 + cl) * 2]; ss += sh[(k * 16 + cl) * 2 + 1]; }
    }
}

// forward statistics: mean / invstd of the batch, running statistics updated as nn.BatchNorm does (momentum, unbiased var)
__global__ __launch_bounds__(256) void bn2_stats_final_kernel(const float* __restrict__ part, int nblk, int C, long npix,
                                                              float eps, float momentum, float* __restrict__ mean,
                                                              float* __restrict__ invstd, float* __restrict__ rmean,
                                                              float* __restrict__ rvar) {
    __shared__ double sh[512];
    double s, ss;
    bn2_reduce16(part, nblk, C, sh, s, ss);
    const int c = blockIdx.x * 16 + threadIdx.x;
    if (threadIdx.x >= 16 || c >= C) return;
    const double m = s / (double)npix;
    double var = ss / (double)npix - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        const double unb = npix > 1 ? var * ((double)npix / (double)(npix - 1)) : var;
        rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * m);
        rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unb);
    }
}

__global__ __launch_bounds__(256) void bn2_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                             int C, float eps, float* __restrict__ mean,
                                                             float* __restrict__ invstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    mean[c] = rmean[c];
    invstd[c] = 1.f / sqrtf(rvar[c] + eps);
}

// y = bn(x) (+ res) (ReLU); channels [C, out_ld) written as 0.  A thread owns one 4-channel group (its parameters are
// loaded once) and walks the workgroup's pixel range, 256 / G4 pixels per step.
template <bool BF>
__global__ __launch_bounds__(256) void bn2_apply_kernel(const float* __restrict__ x, int x_ld, int C, long npix,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ res, int res_ld, int relu,
                                                        float* __restrict__ out, int out_ld) {
    constexpr int bf = BF;
    const int G4 = out_ld >> 2;                      // groups written per pixel (padding groups get zeros)
    const int rows = 256 / G4;
    const int g = threadIdx.x % G4, row = threadIdx.x / G4;
    if (row >= rows) return;
    const int c = 4 * g;
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
    if (c >= C) {
        for (long p = p0 + row; p < p1; p += rows) stx4(out, p * out_ld + c, bf, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    const float4 m = cpar(mean, c, C), is = cpar(invstd, c, C), ga = cpar(gamma, c, C), be = cpar(beta, c, C);
    for (long p = p0 + row; p < p1; p += rows) {
        const float4 v = ldx4(x, p * x_ld + c, bf);
        float4 o = make_float4((v.x - m.x) * is.x * ga.x + be.x, (v.y - m.y) * is.y * ga.y + be.y, (v.z - m.z) * is.z * ga.z + be.z,
                               (v.w - m.w) * is.w * ga.w + be.w);
        if (res) {
            const float4 r = ldx4(res, p * res_ld + c, bf);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        if (c + 3 >= C) {                            // last, partial group: padding channels stay 0
            if (c + 1 >= C) o.y = 0.f;
            if (c + 2 >= C) o.z = 0.f;
            o.w = 0.f;
        }
        stx4(out, p * out_ld + c, bf, o);
    }
}

__global__ __launch_bounds__(256) void bn2_bwd_final_kernel(const float* __restrict__ part, int nblk, int C,
                                                            float* __restrict__ sums, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
    __shared__ double sh[512];
    double s, ss;
    bn2_reduce16(part, nblk, C, sh, s, ss);
    const int c = blockIdx.x * 16 + threadIdx.x;
    if (threadIdx.x >= 16 || c >= C) return;
    sums[c] = (float)s;
    sums[C + c] = (float)ss;
    dbeta[c] = (float)s;
    dgamma[c] = (float)ss;
}

// dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat))   (training)   |   gamma * invstd * g   (eval)
// g is read from `g` when given (written by the partial pass), else recomputed from dy and the ReLU of bn(x).
// Same thread mapping as bn2_apply_kernel.
template <bool BF>
__global__ __launch_bounds__(256) void bn2_bwd_apply_kernel(const float* __restrict__ x, int x_ld, int C, long npix,
                                                            const float* __restrict__ dy, int dy_ld,
                                                            const float* __restrict__ g, int g_ld,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ sums, int relu, int training,
                                                            float* __restrict__ dx, int dx_ld) {
    constexpr int bf = BF;
    const int G4 = dx_ld >> 2;
    const int rows = 256 / G4;
    const int gi = threadIdx.x % G4, row = threadIdx.x / G4;
    if (row >= rows) return;
    const int c = 4 * gi;
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
    if (c >= C) {
        for (long p = p0 + row; p < p1; p += rows) stx4(dx, p * dx_ld + c, bf, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    const float inv_n = 1.f / (float)npix;
    const float4 m4 = cpar(mean, c, C), is4 = cpar(invstd, c, C), ga4 = cpar(gamma, c, C), be4 = cpar(beta, c, C);
    const float4 sa = cpar(sums, c, C), sb = cpar(sums + C, c, C);
    const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w}, ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w};
    const float be[4] = {be4.x, be4.y, be4.z, be4.w};
    const float k1[4] = {sa.x * inv_n, sa.y * inv_n, sa.z * inv_n, sa.w * inv_n};
    const float k2[4] = {sb.x * inv_n, sb.y * inv_n, sb.z * inv_n, sb.w * inv_n};
    for (long p = p0 + row; p < p1; p += rows) {
        const float4 v = ldx4(x, p * x_ld + c, bf);
        const float4 t = g ? ldx4(g, p * g_ld + c, bf) : ldx4(dy, p * dy_ld + c, bf);
        const float xv[4] = {v.x, v.y, v.z, v.w};
        const float gv[4] = {t.x, t.y, t.z, t.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - mm[k]) * is[k];
            float gg = gv[k];
            if (!g && relu && !(xh * ga[k] + be[k] > 0.f)) gg = 0.f;
            o[k] = training ? ga[k] * is[k] * (gg - k1[k] - xh * k2[k]) : ga[k] * is[k] * gg;   // padding channels: ga = is = 0
        }
        stx4(dx, p * dx_ld + c, bf, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// ------------------------------------------------------------------------------------------------ max pooling
// nn.MaxPool2d(k, s, p) / F.max_pool3d(x, (1, k, k)): the first maximum in (ky, kx) scan order wins, positions outside
// the image are skipped.  idx[n, oy, ox, c] = ky * k + kx of the winner (one byte).
// (BF, here and below: the tensors are stored as bf16.  A template parameter: as a run-time flag every ldx4 is a branch whose bf16
// arm converts - waits for - its own load, so a thread's loads run one after the other.)
template <bool BF>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int ld, int H, int W, int OH, int OW,
                                                          int k, int s, int pad, long total, float* __restrict__ out,
                                                          uint8_t* __restrict__ idx) {
    constexpr int bf = BF;
    const int g4 = ld >> 2;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    long q = idiv(gid, g4, total);
    const int c4 = (int)(gid - q * g4);
    long q2 = idiv(q, OW, total);
    const int ox = (int)(q - q2 * OW);
    const int n = (int)idiv(q2, OH, total);
    const int oy = (int)(q2 - (long)n * OH);
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first = true;
    for (int ky = 0; ky < k; ++ky) {
        const int iy = oy * s - pad + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = ox * s - pad + kx;
            if (ix < 0 || ix >= W) continue;
            const float4 v = ldx4(x, ((size_t)(n * H + iy) * W + ix) * ld + 4 * c4, bf);
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (first || e[j] > best[j]) { best[j] = e[j]; bi[j] = ky * k + kx; }
            first = false;
        }
    }
    const size_t o = ((size_t)(n * OH + oy) * OW + ox) * ld + 4 * c4;
    stx4(out, o, bf, make_float4(best[0], best[1], best[2], best[3]));
    *reinterpret_cast<uchar4*>(idx + o) = make_uchar4((uint8_t)bi[0], (uint8_t)bi[1], (uint8_t)bi[2], (uint8_t)bi[3]);
}

// gather form: every input pixel collects dy of the windows it won (fixed order, no atomics)
template <bool BF>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          int ld, int H, int W, int OH, int OW, int k, int s, int pad,
                                                          long total, float* __restrict__ dx) {
    constexpr int bf = BF;
    const int g4 = ld >> 2;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    long q = idiv(gid, g4, total);
    const int c4 = (int)(gid - q * g4);
    long q2 = idiv(q, W, total);
    const int ix = (int)(q - q2 * W);
    const int n = (int)idiv(q2, H, total);
    const int iy = (int)(q2 - (long)n * H);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int oy0 = (iy + pad - k + 1 + s - 1) / s, oy1 = (iy + pad) / s;      // ceil / floor
    if (iy + pad - k + 1 < 0) oy0 = 0;
    int ox0 = (ix + pad - k + 1 + s - 1) / s, ox1 = (ix + pad) / s;
    if (ix + pad - k + 1 < 0) ox0 = 0;
    if (oy1 > OH - 1) oy1 = OH - 1;
    if (ox1 > OW - 1) ox1 = OW - 1;
    for (int oy = oy0; oy <= oy1; ++oy)
        for (int ox = ox0; ox <= ox1; ++ox) {
            const int me = (iy - (oy * s - pad)) * k + (ix - (ox * s - pad));
            const size_t o = ((size_t)(n * OH + oy) * OW + ox) * ld + 4 * c4;
            const uchar4 w = *reinterpret_cast<const uchar4*>(idx + o);
            const float4 g = ldx4(dy, o, bf);
            if (w.x == me) acc[0] += g.x;
            if (w.y == me) acc[1] += g.y;
            if (w.z == me) acc[2] += g.z;
            if (w.w == me) acc[3] += g.w;
        }
    stx4(dx, ((size_t)(n * H + iy) * W + ix) * ld + 4 * c4, bf, make_float4(acc[0], acc[1], acc[2], acc[3]));
}

// ------------------------------------------------------------------------------------------------ stride-2 subsampling
// forward: out[n, y, x] = in[n, 2y, 2x]  (the input side of a 1x1 stride-2 convolution); backward: zero-insertion.
template <bool BF>
__global__ __launch_bounds__(256) void subsample2_kernel(const float* __restrict__ in, int ld, int H, int W, int OH, int OW,
                                                         long total, float* __restrict__ out, int backward) {
    constexpr int bf = BF;
    const int g4 = ld >> 2;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    long q = idiv(gid, g4, total);
    const int c4 = (int)(gid - q * g4);
    if (!backward) {
        const long q2 = idiv(q, OW, total);
        const int ox = (int)(q - q2 * OW);
        const int n = (int)idiv(q2, OH, total);
        const int oy = (int)(q2 - (long)n * OH);
        stx4(out, ((size_t)(n * OH + oy) * OW + ox) * ld + 4 * c4, bf,
             ldx4(in, ((size_t)(n * H + 2 * oy) * W + 2 * ox) * ld + 4 * c4, bf));
    } else {            // `in` = gradient at [OH, OW], `out` = gradient at [H, W]
        const long q2 = idiv(q, W, total);
        const int ix = (int)(q - q2 * W);
        const int n = (int)idiv(q2, H, total);
        const int iy = (int)(q2 - (long)n * H);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(iy & 1) && !(ix & 1)) v = ldx4(in, ((size_t)(n * OH + (iy >> 1)) * OW + (ix >> 1)) * ld + 4 * c4, bf);
        stx4(out, ((size_t)(n * H + iy) * W + ix) * ld + 4 * c4, bf, v);
    }
}

// ------------------------------------------------------------------------------------------------ bilinear resize
// F.interpolate(mode='bilinear', align_corners=False): src = (dst + 0.5) * (in / out) - 0.5, clamped at 0; second tap
// clamped at the border.
__device__ __forceinline__ void bil_taps(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    float src = ((float)o + 0.5f) * scale - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ in, int ld, int H, int W, int OH, int OW,
                                                           float sy, float sx, long total, float* __restrict__ out) {
    const int g4 = ld >> 2;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    long q = idiv(gid, g4, total);
    const int c4 = (int)(gid - q * g4);
    long q2 = idiv(q, OW, total);
    const int ox = (int)(q - q2 * OW);
    const int n = (int)idiv(q2, OH, total);
    const int oy = (int)(q2 - (long)n * OH);
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    bil_taps(oy, sy, H, y0, y1, ly0, ly1);
    bil_taps(ox, sx, W, x0, x1, lx0, lx1);
    const float* b = in + (size_t)n * H * W * ld + 4 * c4;
    const float4 a = ld4(b + ((size_t)y0 * W + x0) * ld), bb = ld4(b + ((size_t)y0 * W + x1) * ld);
    const float4 c = ld4(b + ((size_t)y1 * W + x0) * ld), d = ld4(b + ((size_t)y1 * W + x1) * ld);
    float4 r;
    r.x = ly0 * (lx0 * a.x + lx1 * bb.x) + ly1 * (lx0 * c.x + lx1 * d.x);
    r.y = ly0 * (lx0 * a.y + lx1 * bb.y) + ly1 * (lx0 * c.y + lx1 * d.y);
    r.z = ly0 * (lx0 * a.z + lx1 * bb.z) + ly1 * (lx0 * c.z + lx1 * d.z);
    r.w = ly0 * (lx0 * a.w + lx1 * bb.w) + ly1 * (lx0 * c.w + lx1 * d.w);
    st4(out + ((size_t)(n * OH + oy) * OW + ox) * ld + 4 * c4, r);
}

// gather form of the adjoint: input pixel (iy, ix) scans the output range whose taps can touch it
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, int ld, int H, int W, int OH, int OW,
                                                           float sy, float sx, long total, float* __restrict__ dx) {
    const int g4 = ld >> 2;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    long q = idiv(gid, g4, total);
    const int c4 = (int)(gid - q * g4);
    long q2 = idiv(q, W, total);
    const int ix = (int)(q - q2 * W);
    const int n = (int)idiv(q2, H, total);
    const int iy = (int)(q2 - (long)n * H);
    int oy0 = (int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1, oy1 = (int)ceilf(((float)iy + 1.5f) / sy - 0.5f) + 1;
    int ox0 = (int)floorf(((float)ix - 0.5f) / sx - 0.5f) - 1, ox1 = (int)ceilf(((float)ix + 1.5f) / sx - 0.5f) + 1;
    if (oy0 < 0) oy0 = 0;
    if (ox0 < 0) ox0 = 0;
    if (oy1 > OH - 1) oy1 = OH - 1;
    if (ox1 > OW - 1) ox1 = OW - 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = oy0; oy <= oy1; ++oy) {
        int y0, y1;
        float ly0, ly1;
        bil_taps(oy, sy, H, y0, y1, ly0, ly1);
        const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
        if (wy == 0.f) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            int x0, x1;
            float lx0, lx1;
            bil_taps(ox, sx, W, x0, x1, lx0, lx1);
            const float w = wy * ((x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f));
            if (w == 0.f) continue;
            const float4 g = ld4(dy + ((size_t)(n * OH + oy) * OW + ox) * ld + 4 * c4);
            acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
        }
    }
    st4(dx + ((size_t)(n * H + iy) * W + ix) * ld + 4 * c4, acc);
}

// ------------------------------------------------------------------------------------------------ depth <-> space (block 2)
// to_space: out[n, 2y+py, 2x+px, c] = in[n, y, x, (py*2+px)*Co + c]; to_depth: the inverse.  Co % 4 == 0.
template <bool BF>
__global__ __launch_bounds__(256) void depth_space2_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                           int Co, long total, int to_depth) {
    constexpr int bf = BF;
    const int g4 = Co >> 2;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;                       // total = N * 2H * 2W * g4
    long q = idiv(gid, g4, total);
    const int c4 = (int)(gid - q * g4);
    const long q2 = idiv(q, 2 * W, total);
    const int X = (int)(q - q2 * (2 * W));
    const int n = (int)idiv(q2, 2 * H, total);
    const int Y = (int)(q2 - (long)n * (2 * H));
    const size_t deep = ((size_t)(n * H + (Y >> 1)) * W + (X >> 1)) * (4 * Co) + (size_t)(((Y & 1) * 2 + (X & 1)) * Co) + 4 * c4;
    const size_t wide = ((size_t)(n * 2 * H + Y) * (2 * W) + X) * Co + 4 * c4;
    if (to_depth) stx4(out, deep, bf, ldx4(in, wide, bf));
    else stx4(out, wide, bf, ldx4(in, deep, bf));
}

// ConvTranspose2d(k=4, s=2, p=1) weight [Ci, Co, 4, 4]  <->  3x3 conv weight [4*Co, Ci, 3, 3] (output-phase major):
// output row 2y + py takes input rows y - 1 + ty with kernel row ky(py, ty): (0,0)->3, (0,1)->1, (1,1)->2, (1,2)->0.
__device__ __forceinline__ int convt_k(int ph, int t) { return ph == 0 ? (t == 0 ? 3 : t == 1 ? 1 : -1) : (t == 1 ? 2 : t == 2 ? 0 : -1); }

__global__ __launch_bounds__(256) void convt_pack_kernel(const float* __restrict__ w, int Ci, int Co, float* __restrict__ w3,
                                                         long total) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;                       // total = 4*Co*Ci*9
    const int tx = (int)(gid % 3), ty = (int)((gid / 3) % 3);
    long q = gid / 9;
    const int ci = (int)(q % Ci);
    q /= Ci;
    const int co = (int)(q % Co), ph = (int)(q / Co);
    const int ky = convt_k(ph >> 1, ty), kx = convt_k(ph & 1, tx);
    w3[gid] = (ky >= 0 && kx >= 0) ? w[(((size_t)ci * Co + co) * 4 + ky) * 4 + kx] : 0.f;
}

__global__ __launch_bounds__(256) void convt_unpack_kernel(const float* __restrict__ dw3, int Ci, int Co, float* __restrict__ dw,
                                                           long total) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;                       // total = Ci*Co*16
    const int kx = (int)(gid & 3), ky = (int)((gid >> 2) & 3);
    long q = gid >> 4;
    const int co = (int)(q % Co), ci = (int)(q / Co);
    const int py = (ky & 1) ? 0 : 1, px = (kx & 1) ? 0 : 1;
    const int ty = ky == 0 ? 2 : ky == 3 ? 0 : 1, tx = kx == 0 ? 2 : kx == 3 ? 0 : 1;
    dw[gid] = dw3[((((size_t)(py * 2 + px) * Co + co) * Ci + ci) * 3 + ty) * 3 + tx];
}

// nn.Conv3d(Ci, Co, (3,1,1)) weight [Co, Ci, 3] <-> three 1x1 conv weights [3][Co][Ci] (efficient_layers.py:271-278)
__global__ __launch_bounds__(256) void tconv_relayout_kernel(const float* __restrict__ in, float* __restrict__ out, long CoCi,
                                                             int to_taps) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= 3 * CoCi) return;
    const long i = gid / 3;
    const int k = (int)(gid % 3);                   // gid indexes the [Co*Ci][3] side
    if (to_taps) out[(size_t)k * CoCi + i] = in[gid];
    else out[gid] = in[(size_t)k * CoCi + i];
}

// The same (3,1,1) convolution over a TIME-IN-CHANNELS activation [B,H,W,T*Cp] (frame t = channels [t*Cp, t*Cp+C)): out frame
// t is ONE 1x1 convolution over the contiguous channel range of frames t-1..t+1 with the tap weights laid side by side.
// transpose == 0: out[r][j*Cp + c] = w[r][c][k0 + j]            (rows r = Co, c < C = Ci: forward / weight-gradient form)
// transpose == 1: out[r][j*Cp + c] = w[c][r][k0 + nk - 1 - j]   (rows r = Ci, c < C = Co: input-gradient form over dy frames)
// columns c in [C, Cp) are zero.  w is [Co][Ci][3].
__global__ __launch_bounds__(256) void tconv_cat_kernel(const float* __restrict__ w, int Co, int Ci, int rows, int C, int Cp,
                                                        int k0, int nk, int transpose, float* __restrict__ out, long total) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;                       // total = rows * nk * Cp
    const int c = (int)(gid % Cp);
    const long q = gid / Cp;
    const int j = (int)(q % nk), r = (int)(q / nk);
    float v = 0.f;
    if (c < C) v = transpose ? w[((size_t)c * Ci + r) * 3 + k0 + nk - 1 - j] : w[((size_t)r * Ci + c) * 3 + k0 + j];
    out[gid] = v;
}

// dw[co][ci][k] from the three side-by-side gradient blocks: first (frame 0: taps 1, 2), mid (taps 0, 1, 2; may be null),
// last (frame T-1: taps 0, 1); each [rows >= Co][ntaps * Cp]
__global__ __launch_bounds__(256) void tconv_grad_combine_kernel(const float* __restrict__ gf, const float* __restrict__ gm,
                                                                 const float* __restrict__ gl, int Co, int Ci, int Cp,
                                                                 float* __restrict__ dw, long total) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;                       // total = Co * Ci * 3
    const int k = (int)(gid % 3);
    const long q = gid / 3;
    const int ci = (int)(q % Ci), co = (int)(q / Ci);
    float v = 0.f;
    if (k >= 1) v += gf[(size_t)co * 2 * Cp + (k - 1) * Cp + ci];
    if (gm) v += gm[(size_t)co * 3 * Cp + k * Cp + ci];
    if (k <= 1) v += gl[(size_t)co * 2 * Cp + k * Cp + ci];
    dw[gid] = v;
}

// ------------------------------------------------------------------------------------------------ small helpers
// part[n][blk][c] = sum over the blk-th pixel range of image n (input of nvq_cbam_channel).  grid (nblk, N)
__global__ __launch_bounds__(256) void gap_partial_kernel(const float* __restrict__ x, int ld, int C, long HW,
                                                          float* __restrict__ part) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const long per = (HW + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        float s = 0.f;
        if (c < C)
            for (long p = p0 + row; p < p1; p += 4) s += x[((size_t)n * HW + p) * ld + c];
        red[row][lane] = s;
        __syncthreads();
        if (row == 0 && c < C)
            part[((size_t)n * gridDim.x + blockIdx.x) * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        __syncthreads();
    }
}

// x[n, p, c] += v[n, c]
__global__ __launch_bounds__(256) void add_image_channel_kernel(float* __restrict__ x, int ld, int C, long HW, long total,
                                                                const float* __restrict__ v) {
    const int g4 = C >> 2;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const long p = idiv(gid, g4, total);
    const int c4 = (int)(gid - p * g4);
    const int n = (int)idiv(p, HW, total);
    float4 a = ld4(x + p * ld + 4 * c4);
    const float4 b = ld4(v + (size_t)n * C + 4 * c4);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    st4(x + p * ld + 4 * c4, a);
}

// dst[p, dst_coff + c] (+)= alpha * src[p, src_coff + c] for c < C (C % 4 == 0), either side fp32 or bf16: the storage-type
// boundary between the bf16 stages of FrameRecoveryNet and its fp32 attention / fusion stages, and mean / broadcast over time
template <bool DB, bool SB>
__global__ __launch_bounds__(256) void cast_slice_kernel(float* __restrict__ dst, int dst_ld, int dst_coff,
                                                         const float* __restrict__ src, int src_ld, int src_coff,
                                                         int C, long total, float alpha, int accumulate) {
    constexpr int dst_bf = DB, src_bf = SB;
    const int g4 = C >> 2;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const long p = idiv(gid, g4, total);
    const int c = 4 * (int)(gid - p * g4);
    float4 v = ldx4(src, p * src_ld + src_coff + c, src_bf);
    v = make_float4(alpha * v.x, alpha * v.y, alpha * v.z, alpha * v.w);
    if (accumulate) {
        const float4 o = ldx4(dst, p * dst_ld + dst_coff + c, dst_bf);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    stx4(dst, p * dst_ld + dst_coff + c, dst_bf, v);
}

// y = tanh(x) | dx = dy * (1 - y^2), over whole rows (ld floats per pixel; padding channels stay 0)
__global__ __launch_bounds__(256) void tanh_kernel(const float* __restrict__ a, const float* __restrict__ y, long n4,
                                                   float* __restrict__ out, int backward) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= n4) return;
    const float4 v = ld4(a + 4 * gid);
    float4 r;
    if (!backward) {
        r = make_float4(tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w));
    } else {
        const float4 t = ld4(y + 4 * gid);
        r = make_float4(v.x * (1.f - t.x * t.x), v.y * (1.f - t.y * t.y), v.z * (1.f - t.z * t.z), v.w * (1.f - t.w * t.w));
    }
    st4(out + 4 * gid, r);
}

// FusionModule (frame_recovery.py:239-254): y = aligned + a0 * mean_c(sp) + a1 * mean_c(tp), (a0, a1) = softmax(logits[0:2]).
// One group of C/4 lanes per pixel (C a power of two, 16 <= C <= 256); saves attn[p][2] and means[p][2].
__global__ __launch_bounds__(256) void fusion_mix_fwd_kernel(const float* __restrict__ aligned, const float* __restrict__ logits,
                                                             int lg_ld, const float* __restrict__ sp, int sp_ld,
                                                             const float* __restrict__ tp, int tp_ld, int C, long npix,
                                                             float* __restrict__ out, float* __restrict__ attn,
                                                             float* __restrict__ means) {
    const int C4 = C >> 2;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= npix * C4) return;
    const int c4 = (int)(gid % C4);
    const long p = gid / C4;
    const float4 s = ld4(sp + p * sp_ld + 4 * c4), t = ld4(tp + p * tp_ld + 4 * c4);
    const float ms = group_sum((s.x + s.y) + (s.z + s.w), C4) / (float)C;
    const float mt = group_sum((t.x + t.y) + (t.z + t.w), C4) / (float)C;
    const float l0 = logits[p * lg_ld], l1 = logits[p * lg_ld + 1];
    const float mx = fmaxf(l0, l1);
    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
    const float a0 = e0 / (e0 + e1), a1 = e1 / (e0 + e1);
    const float add = a0 * ms + a1 * mt;
    const float4 a = ld4(aligned + p * C + 4 * c4);
    st4(out + p * C + 4 * c4, make_float4(a.x + add, a.y + add, a.z + add, a.w + add));
    if (c4 == 0) {
        attn[2 * p] = a0; attn[2 * p + 1] = a1;
        means[2 * p] = ms; means[2 * p + 1] = mt;
    }
}

__global__ __launch_bounds__(256) void fusion_mix_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ attn,
                                                             const float* __restrict__ means, int C, long npix,
                                                             float* __restrict__ dlogits, int lg_ld, float* __restrict__ dsp,
                                                             int sp_ld, float* __restrict__ dtp, int tp_ld) {
    const int C4 = C >> 2;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= npix * C4) return;
    const int c4 = (int)(gid % C4);
    const long p = gid / C4;
    const float4 g = ld4(dy + p * C + 4 * c4);
    const float S = group_sum((g.x + g.y) + (g.z + g.w), C4);
    const float a0 = attn[2 * p], a1 = attn[2 * p + 1];
    const float ds = a0 * S / (float)C, dt = a1 * S / (float)C;
    st4(dsp + p * sp_ld + 4 * c4, make_float4(ds, ds, ds, ds));
    st4(dtp + p * tp_ld + 4 * c4, make_float4(dt, dt, dt, dt));
    if (c4 == 0) {
        const float d0 = S * means[2 * p], d1 = S * means[2 * p + 1];
        const float dot = a0 * d0 + a1 * d1;
        dlogits[p * lg_ld] = a0 * (d0 - dot);
        dlogits[p * lg_ld + 1] = a1 * (d1 - dot);
        for (int k = 2; k < lg_ld; ++k) dlogits[p * lg_ld + k] = 0.f;
    }
}

// FrameRecoveryNet blend (frame_recovery.py:439-440): out = frame * (1 - m) + rec * m.  frame / out NCHW, rec NHWC, m [N,1,H,W]
__global__ __launch_bounds__(256) void mask_blend_kernel(const float* __restrict__ frame, const float* __restrict__ rec,
                                                         int rec_ld, const float* __restrict__ mask, int C, long HW, long total,
                                                         float* __restrict__ out) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;                       // total = N * C * HW
    const long p = gid % HW;
    const long q = gid / HW;
    const int c = (int)(q % C);
    const long n = q / C;
    const float m = mask[n * HW + p];
    out[gid] = frame[gid] * (1.f - m) + rec[(n * HW + p) * rec_ld + c] * m;
}

// drec[n, p, c] = dout[n, c, p] * m[n, p]; channels [C, ld) = 0
__global__ __launch_bounds__(256) void mask_blend_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ mask,
                                                             int C, long HW, long total, float* __restrict__ drec, int ld) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;                       // total = N * HW
    const long n = gid / HW, p = gid % HW;
    const float m = mask[gid];
    for (int c = 0; c < ld; ++c) drec[gid * ld + c] = c < C ? dout[(n * C + c) * HW + p] * m : 0.f;
}

// ------------------------------------------------------------------------------------------------ 7x7 stride-2 stem conv
// nn.Conv2d(4, Co, 7, 2, 3, bias=False) (frame_recovery.py:42-47) on a 4-channel NHWC image.  Workgroup = 8x8 output
// pixels x all Co (<= 64 per pass): the 21x21x4 input patch and the [k][co] weight slab live in LDS, a thread owns one
// pixel and 16 output channels.  K = 196, AI ~ 100 FLOP/B: tiny next to the rest of the net, so plain FMAs.
constexpr int ST_T = 8, ST_P = 2 * ST_T + 5;        // 21

__global__ __launch_bounds__(256) void stem7_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, int H, int W,
                                                        int OH, int OW, int Co, int tilesX, float* __restrict__ out, int out_ld,
                                                        int out_bf) {
    extern __shared__ float smem[];
    float* patch = smem;                             // [21*21][4]
    float* wl = smem + ST_P * ST_P * 4;              // [196][64]
    const int n = blockIdx.y;
    const int oy0 = (blockIdx.x / tilesX) * ST_T, ox0 = (blockIdx.x % tilesX) * ST_T;
    for (int i = threadIdx.x; i < ST_P * ST_P; i += 256) {
        const int iy = 2 * oy0 - 3 + i / ST_P, ix = 2 * ox0 - 3 + i % ST_P;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld4(x + ((size_t)(n * H + iy) * W + ix) * 4);
        st4(patch + 4 * i, v);
    }
    const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;          // pixel of the tile, group of 16 channels
    const int py_ = px / ST_T, px_ = px % ST_T;
    for (int co0 = 0; co0 < Co; co0 += 64) {
        __syncthreads();
        // wl[k][co] with k = (ky*7 + kx)*4 + ci  <-  w[co][ci][ky][kx]
        for (int i = threadIdx.x; i < 196 * 64; i += 256) {
            const int co = i & 63, k = i >> 6;
            const int ci = k & 3, t = k >> 2;
            wl[i] = (co0 + co < Co) ? w[((size_t)(co0 + co) * 4 + ci) * 49 + t] : 0.f;
        }
        __syncthreads();
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int ky = 0; ky < 7; ++ky)
            for (int kx = 0; kx < 7; ++kx) {
                const float4 v = ld4(patch + 4 * ((2 * py_ + ky) * ST_P + 2 * px_ + kx));
                const float e[4] = {v.x, v.y, v.z, v.w};
                const float* wr = wl + ((ky * 7 + kx) * 4) * 64 + cg * 16;
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const float4 ww = ld4(wr + ci * 64 + 4 * j4);
                        acc[4 * j4 + 0] += e[ci] * ww.x;
                        acc[4 * j4 + 1] += e[ci] * ww.y;
                        acc[4 * j4 + 2] += e[ci] * ww.z;
                        acc[4 * j4 + 3] += e[ci] * ww.w;
                    }
            }
        const int oy = oy0 + py_, ox = ox0 + px_;
        if (oy < OH && ox < OW) {
            const size_t o = ((size_t)(n * OH + oy) * OW + ox) * out_ld + co0 + cg * 16;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
                if (co0 + cg * 16 + 4 * j4 < Co)
                    stx4(out, o + 4 * j4, out_bf, make_float4(acc[4 * j4], acc[4 * j4 + 1], acc[4 * j4 + 2], acc[4 * j4 + 3]));
        }
    }
}

// weight gradient: dw[co][ci][ky][kx] = sum_{n, oy, ox} dy[n, oy, ox, co] * x[n, 2oy+ky-3, 2ox+kx-3, ci].  Persistent
// workgroups walk the tiles; a thread owns channel co = lane (64 per pass) and the 49 k-values k = wave + 4 j; partial
// slabs [workgroup][co][196] are summed by launch_reduce_partials in a fixed order.
__global__ __launch_bounds__(256) void stem7_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int dy_ld,
                                                          int N, int H, int W, int OH, int OW, int Co, int co0, int tilesX,
                                                          int tilesY, float* __restrict__ part, int dy_bf) {
    __shared__ float patch[ST_P * ST_P * 4];
    __shared__ float dtile[ST_T * ST_T][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[49];
#pragma unroll
    for (int j = 0; j < 49; ++j) acc[j] = 0.f;
    const long ntiles = (long)N * tilesX * tilesY;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = (int)(tile / (tilesX * tilesY));
        const int tt = (int)(tile % (tilesX * tilesY));
        const int oy0 = (tt / tilesX) * ST_T, ox0 = (tt % tilesX) * ST_T;
        __syncthreads();
        for (int i = threadIdx.x; i < ST_P * ST_P; i += 256) {
            const int iy = 2 * oy0 - 3 + i / ST_P, ix = 2 * ox0 - 3 + i % ST_P;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld4(x + ((size_t)(n * H + iy) * W + ix) * 4);
            st4(patch + 4 * i, v);
        }
        for (int i = threadIdx.x; i < ST_T * ST_T * 64; i += 256) {
            const int co = i & 63, p = i >> 6;
            const int oy = oy0 + p / ST_T, ox = ox0 + p % ST_T;
            float dv = 0.f;
            if (oy < OH && ox < OW && co0 + co < Co) {
                const size_t e = ((size_t)(n * OH + oy) * OW + ox) * dy_ld + co0 + co;
                dv = dy_bf ? (float)reinterpret_cast<const __bf16*>(dy)[e] : dy[e];
            }
            dtile[p][co] = dv;
        }
        __syncthreads();
        for (int p = 0; p < ST_T * ST_T; ++p) {
            const float g = dtile[p][lane];
            const int by = 2 * (p / ST_T), bx = 2 * (p % ST_T);
#pragma unroll
            for (int j = 0; j < 49; ++j) {
                const int k = wave + 4 * j;                      // k = (ky*7 + kx)*4 + ci
                const int ci = k & 3, t = k >> 2;
                acc[j] += g * patch[4 * ((by + t / 7) * ST_P + bx + t % 7) + ci];
            }
        }
    }
    // part[blk][co][ci][ky][kx]  (PyTorch weight order inside a slab of cn * 196, cn = channels of this pass)
    const int cn = Co - co0 < 64 ? Co - co0 : 64;
    if (lane < cn) {
#pragma unroll
        for (int j = 0; j < 49; ++j) {
            const int k = wave + 4 * j;
            const int ci = k & 3, t = k >> 2;
            part[((size_t)blockIdx.x * cn + lane) * 196 + ci * 49 + t] = acc[j];
        }
    }
}

static inline int blocks_for(long total) { return (int)((total + 255) / 256); }

}  // namespace nvq

using namespace nvq;

extern "C" {

int nvq_nchw_to_nhwc(const float* src, long src_nstride, int N, int C, int H, int W, float* dst, int dst_ld, int dst_coff,
                     int czero, void* stream) {
    NVQ_REQUIRE(N > 0 && C > 0 && czero >= C && dst_coff + czero <= dst_ld, "nchw_to_nhwc: C %d czero %d coff %d ld %d", C, czero,
                dst_coff, dst_ld);
    const long HW = (long)H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ceil_div(HW, 64), ceil_div(czero, 64), N), dim3(256), 0, (hipStream_t)stream, src,
                       src_nstride, C, czero, HW, dst, dst_ld, dst_coff);
    return check_launch("nchw_to_nhwc");
}

int nvq_nhwc_to_nchw(const float* src, int src_ld, int src_coff, int N, int C, int H, int W, float* dst, long dst_nstride,
                     void* stream) {
    NVQ_REQUIRE(N > 0 && C > 0 && src_coff + C <= src_ld, "nhwc_to_nchw: C %d coff %d ld %d", C, src_coff, src_ld);
    const long HW = (long)H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ceil_div(HW, 64), ceil_div(C, 64), N), dim3(256), 0, (hipStream_t)stream, src,
                       src_ld, src_coff, C, HW, dst, dst_nstride);
    return check_launch("nhwc_to_nchw");
}

static int bn2_nblk(long npix) {
    int nb = ceil_div(npix, 256);
    if (nb > BN2_MAXBLK) nb = BN2_MAXBLK;
    return nb < 1 ? 1 : nb;
}

// element-wise passes: ~8 pixel steps per thread, at most 16 workgroups per CU
static int bn2_ew_blocks(long npix, int ld) {
    const int rows = 256 / (ld / 4);
    long nb = (npix + (long)rows * 8 - 1) / ((long)rows * 8);
    if (nb > 4096) nb = 4096;
    return nb < 1 ? 1 : (int)nb;
}

size_t nvq_bn2_workspace_bytes(int C) { return (size_t)BN2_MAXBLK * 2 * (size_t)C * sizeof(float); }

int nvq_bn2_stats(const float* x, int x_ld, int C, long npix, float eps, float momentum, float* mean, float* invstd,
                  float* running_mean, float* running_var, float* workspace, size_t workspace_bytes, int bf16, void* stream) {
    NVQ_REQUIRE(C > 0 && C <= 1024 && ((C + 3) & ~3) <= x_ld && x_ld % 4 == 0 && aligned16(x) && npix > 0, "bn2_stats: C %d ld %d", C, x_ld);
    NVQ_REQUIRE(workspace_bytes >= nvq_bn2_workspace_bytes(C), "bn2_stats: workspace");
    hipStream_t s = (hipStream_t)stream;
    const int nb = bn2_nblk(npix);
#define NVQ_B2P(B_) hipLaunchKernelGGL((bn2_partial_kernel<0, B_>), dim3(nb), dim3(256), 0, s, x, x_ld, C, npix, nullptr, 0, nullptr, 0, \
                                       nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, workspace)
    if (bf16) NVQ_B2P(true); else NVQ_B2P(false);
#undef NVQ_B2P
    int rc = check_launch("bn2_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(bn2_stats_final_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, s, workspace, nb, C, npix, eps, momentum, mean,
                       invstd, running_mean, running_var);
    return check_launch("bn2_stats_final");
}

int nvq_bn2_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* invstd,
                       void* stream) {
    hipLaunchKernelGGL(bn2_eval_stats_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, running_mean, running_var,
                       C, eps, mean, invstd);
    return check_launch("bn2_eval_stats");
}

int nvq_bn2_apply(const float* x, int x_ld, int C, long npix, const float* mean, const float* invstd, const float* gamma,
                  const float* beta, const float* res, int res_ld, int relu, float* out, int out_ld, int bf16, void* stream) {
    NVQ_REQUIRE(C > 0 && ((C + 3) & ~3) <= x_ld && x_ld % 4 == 0 && C <= out_ld && out_ld % 4 == 0 && aligned16(out) && aligned16(x) &&
                    (!res || (((C + 3) & ~3) <= res_ld && res_ld % 4 == 0 && aligned16(res))),
                "bn2_apply: C %d ld %d/%d", C, x_ld, out_ld);
    NVQ_REQUIRE(out_ld <= 1024, "bn2_apply: ld %d", out_ld);
#define NVQ_B2A(B_) hipLaunchKernelGGL(bn2_apply_kernel<B_>, dim3(bn2_ew_blocks(npix, out_ld)), dim3(256), 0, (hipStream_t)stream, x, \
                                       x_ld, C, npix, mean, invstd, gamma, beta, res, res_ld, relu, out, out_ld)
    if (bf16) NVQ_B2A(true); else NVQ_B2A(false);
#undef NVQ_B2A
    return check_launch("bn2_apply");
}

int nvq_bn2_backward(const float* dy, int dy_ld, const float* x, int x_ld, int C, long npix, const float* mean,
                     const float* invstd, const float* gamma, const float* beta, const float* res, int res_ld, int relu,
                     int training, float* dx, int dx_ld, float* dres, int dres_ld, float* dgamma, float* dbeta,
                     float* workspace, size_t workspace_bytes, int bf16, void* stream) {
    const int C4 = (C + 3) & ~3;
    NVQ_REQUIRE(C > 0 && C <= 1024 && C4 <= x_ld && C4 <= dy_ld && C4 <= dx_ld && x_ld % 4 == 0 && dy_ld % 4 == 0 && dx_ld % 4 == 0 &&
                    aligned16(dx) && aligned16(x) && aligned16(dy), "bn2_backward: C %d", C);
    NVQ_REQUIRE(!res || (dres && C4 <= dres_ld && dres_ld % 4 == 0 && C4 <= res_ld && res_ld % 4 == 0 && aligned16(res) && aligned16(dres)),
                "bn2_backward: a residual input needs its gradient buffer");
    NVQ_REQUIRE(workspace_bytes >= nvq_bn2_workspace_bytes(C) + 2 * (size_t)C * sizeof(float), "bn2_backward: workspace");
    hipStream_t s = (hipStream_t)stream;
    const int nb = bn2_nblk(npix);
    float* sums = workspace + (size_t)BN2_MAXBLK * 2 * C;
#define NVQ_B2P(B_) hipLaunchKernelGGL((bn2_partial_kernel<1, B_>), dim3(nb), dim3(256), 0, s, x, x_ld, C, npix, dy, dy_ld, res, res_ld, \
                                       mean, invstd, gamma, beta, relu, dres, dres_ld, workspace)
    if (bf16) NVQ_B2P(true); else NVQ_B2P(false);
#undef NVQ_B2P
    int rc = check_launch("bn2_bwd_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(bn2_bwd_final_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, s, workspace, nb, C, sums, dgamma, dbeta);
    rc = check_launch("bn2_bwd_final");
    if (rc) return rc;
    NVQ_REQUIRE(dx_ld <= 1024, "bn2_backward: ld %d", dx_ld);
#define NVQ_B2B(B_) hipLaunchKernelGGL(bn2_bwd_apply_kernel<B_>, dim3(bn2_ew_blocks(npix, dx_ld)), dim3(256), 0, s, x, x_ld, C, npix, dy, \
                                       dy_ld, dres, dres_ld, mean, invstd, gamma, beta, sums, relu, training, dx, dx_ld)
    if (bf16) NVQ_B2B(true); else NVQ_B2B(false);
#undef NVQ_B2B
    return check_launch("bn2_bwd_apply");
}

int nvq_maxpool_forward(const float* x, int ld, int N, int H, int W, int k, int s, int pad, float* out, uint8_t* idx,
                        int bf16, void* stream) {
    NVQ_REQUIRE(ld % 4 == 0 && k >= 1 && k <= 7 && s >= 1 && pad * 2 <= k && aligned16(x) && aligned16(out), "maxpool_forward: args");
    const int OH = (H + 2 * pad - k) / s + 1, OW = (W + 2 * pad - k) / s + 1;
    NVQ_REQUIRE(OH > 0 && OW > 0, "maxpool_forward: empty output");
    const long total = (long)N * OH * OW * (ld / 4);
    if (bf16)
        hipLaunchKernelGGL(maxpool_fwd_kernel<true>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, ld, H, W, OH, OW, k,
                           s, pad, total, out, idx);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel<false>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, ld, H, W, OH, OW, k,
                           s, pad, total, out, idx);
    return check_launch("maxpool_forward");
}

int nvq_maxpool_backward(const float* dy, const uint8_t* idx, int ld, int N, int H, int W, int k, int s, int pad, float* dx,
                         int bf16, void* stream) {
    NVQ_REQUIRE(ld % 4 == 0 && k >= 1 && k <= 7 && s >= 1 && pad * 2 <= k, "maxpool_backward: args");
    const int OH = (H + 2 * pad - k) / s + 1, OW = (W + 2 * pad - k) / s + 1;
    const long total = (long)N * H * W * (ld / 4);
    if (bf16)
        hipLaunchKernelGGL(maxpool_bwd_kernel<true>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dy, idx, ld, H, W, OH,
                           OW, k, s, pad, total, dx);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<false>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dy, idx, ld, H, W, OH,
                           OW, k, s, pad, total, dx);
    return check_launch("maxpool_backward");
}

int nvq_subsample2(const float* in, int ld, int N, int H, int W, float* out, int backward, int bf16, void* stream) {
    NVQ_REQUIRE(ld % 4 == 0 && N > 0 && H > 0 && W > 0, "subsample2: args");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long total = backward ? (long)N * H * W * (ld / 4) : (long)N * OH * OW * (ld / 4);
    if (bf16)
        hipLaunchKernelGGL(subsample2_kernel<true>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, in, ld, H, W, OH, OW,
                           total, out, backward);
    else
        hipLaunchKernelGGL(subsample2_kernel<false>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, in, ld, H, W, OH, OW,
                           total, out, backward);
    return check_launch("subsample2");
}

int nvq_bilinear_resize(const float* in, int ld, int N, int H, int W, int OH, int OW, float* out, int backward, void* stream) {
    NVQ_REQUIRE(ld % 4 == 0 && N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "bilinear_resize: args");
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    if (!backward) {
        const long total = (long)N * OH * OW * (ld / 4);
        hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, in, ld, H, W, OH, OW, sy,
                           sx, total, out);
    } else {            // in = dy [N, OH, OW], out = dx [N, H, W]
        const long total = (long)N * H * W * (ld / 4);
        hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, in, ld, H, W, OH, OW, sy,
                           sx, total, out);
    }
    return check_launch("bilinear_resize");
}

int nvq_depth_space2(const float* in, float* out, int N, int H, int W, int Co, int to_depth, int bf16, void* stream) {
    NVQ_REQUIRE(Co % 4 == 0 && N > 0 && H > 0 && W > 0, "depth_space2: Co %d", Co);
    const long total = (long)N * 2 * H * 2 * W * (Co / 4);
    if (bf16)
        hipLaunchKernelGGL(depth_space2_kernel<true>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, H, W, Co,
                           total, to_depth);
    else
        hipLaunchKernelGGL(depth_space2_kernel<false>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, H, W, Co,
                           total, to_depth);
    return check_launch("depth_space2");
}

int nvq_convt_pack(const float* w, int Ci, int Co, float* w3, void* stream) {
    const long total = 4L * Co * Ci * 9;
    hipLaunchKernelGGL(convt_pack_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, w, Ci, Co, w3, total);
    return check_launch("convt_pack");
}

int nvq_convt_unpack_grad(const float* dw3, int Ci, int Co, float* dw, void* stream) {
    const long total = 16L * Co * Ci;
    hipLaunchKernelGGL(convt_unpack_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dw3, Ci, Co, dw, total);
    return check_launch("convt_unpack_grad");
}

int nvq_tconv_relayout(const float* in, float* out, int Co, int Ci, int to_taps, void* stream) {
    const long n = (long)Co * Ci;
    hipLaunchKernelGGL(tconv_relayout_kernel, dim3(blocks_for(3 * n)), dim3(256), 0, (hipStream_t)stream, in, out, n, to_taps);
    return check_launch("tconv_relayout");
}

int nvq_tconv_cat(const float* w, int Co, int Ci, int Cp, int k0, int nk, int transpose, float* out, void* stream) {
    NVQ_REQUIRE(k0 >= 0 && nk >= 1 && k0 + nk <= 3 && Cp >= (transpose ? Co : Ci), "tconv_cat: taps %d..%d, Cp %d", k0, k0 + nk, Cp);
    const int rows = transpose ? Ci : Co, C = transpose ? Co : Ci;
    const long total = (long)rows * nk * Cp;
    hipLaunchKernelGGL(tconv_cat_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, w, Co, Ci, rows, C, Cp, k0, nk,
                       transpose, out, total);
    return check_launch("tconv_cat");
}

int nvq_tconv_grad_combine(const float* g_first, const float* g_mid, const float* g_last, int Co, int Ci, int Cp, float* dw,
                           void* stream) {
    NVQ_REQUIRE(g_first && g_last && Cp >= Ci, "tconv_grad_combine: args");
    const long total = (long)Co * Ci * 3;
    hipLaunchKernelGGL(tconv_grad_combine_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, g_first, g_mid, g_last,
                       Co, Ci, Cp, dw, total);
    return check_launch("tconv_grad_combine");
}

int nvq_gap_blocks(int H, int W) {
    int nb = ceil_div((long)H * W, 256);
    return nb > 64 ? 64 : (nb < 1 ? 1 : nb);
}

int nvq_gap_partial(const float* x, int ld, int C, int N, int H, int W, float* part, void* stream) {
    NVQ_REQUIRE(C > 0 && C <= ld, "gap_partial: C %d ld %d", C, ld);
    hipLaunchKernelGGL(gap_partial_kernel, dim3(nvq_gap_blocks(H, W), N), dim3(256), 0, (hipStream_t)stream, x, ld, C, (long)H * W,
                       part);
    return check_launch("gap_partial");
}

int nvq_add_image_channel(float* x, int ld, int C, int N, int H, int W, const float* v, void* stream) {
    NVQ_REQUIRE(C % 4 == 0 && ld % 4 == 0 && C <= ld && aligned16(x) && aligned16(v), "add_image_channel: C %d ld %d", C, ld);
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(add_image_channel_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, ld, C, (long)H * W,
                       total, v);
    return check_launch("add_image_channel");
}

int nvq_cast_slice(float* dst, int dst_ld, int dst_coff, int dst_bf16, const float* src, int src_ld, int src_coff, int src_bf16,
                   int C, long npix, float alpha, int accumulate, void* stream) {
    NVQ_REQUIRE(C > 0 && C % 4 == 0 && dst_ld % 4 == 0 && dst_coff % 4 == 0 && src_ld % 4 == 0 && src_coff % 4 == 0 &&
                    dst_coff + C <= dst_ld && src_coff + C <= src_ld, "cast_slice: C %d ld %d/%d", C, dst_ld, src_ld);
    const long total = npix * (C / 4);
#define NVQ_CAST(DB, SB)                                                                                                  \
    hipLaunchKernelGGL((cast_slice_kernel<DB, SB>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dst, dst_ld, \
                       dst_coff, src, src_ld, src_coff, C, total, alpha, accumulate)
    if (dst_bf16) { if (src_bf16) NVQ_CAST(true, true); else NVQ_CAST(true, false); }
    else { if (src_bf16) NVQ_CAST(false, true); else NVQ_CAST(false, false); }
#undef NVQ_CAST
    return check_launch("cast_slice");
}

int nvq_tanh(const float* a, const float* y, long n, float* out, int backward, void* stream) {
    NVQ_REQUIRE(n % 4 == 0 && aligned16(a) && aligned16(out) && (!backward || y), "tanh: n %ld", n);
    hipLaunchKernelGGL(tanh_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, a, y, n / 4, out, backward);
    return check_launch("tanh");
}

static bool fusion_c_ok(int C) { return C >= 16 && C <= 256 && (C & (C - 1)) == 0; }

int nvq_fusion_mix_forward(const float* aligned, const float* logits, int logits_ld, const float* sp, int sp_ld, const float* tp,
                           int tp_ld, int C, long npix, float* out, float* attn, float* means, void* stream) {
    NVQ_REQUIRE(fusion_c_ok(C) && sp_ld % 4 == 0 && tp_ld % 4 == 0 && logits_ld >= 2, "fusion_mix_forward: C %d", C);
    hipLaunchKernelGGL(fusion_mix_fwd_kernel, dim3(blocks_for(npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, aligned, logits,
                       logits_ld, sp, sp_ld, tp, tp_ld, C, npix, out, attn, means);
    return check_launch("fusion_mix_forward");
}

int nvq_fusion_mix_backward(const float* dy, const float* attn, const float* means, int C, long npix, float* dlogits,
                            int logits_ld, float* dsp, int sp_ld, float* dtp, int tp_ld, void* stream) {
    NVQ_REQUIRE(fusion_c_ok(C) && sp_ld % 4 == 0 && tp_ld % 4 == 0 && logits_ld >= 2, "fusion_mix_backward: C %d", C);
    hipLaunchKernelGGL(fusion_mix_bwd_kernel, dim3(blocks_for(npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, attn, means, C,
                       npix, dlogits, logits_ld, dsp, sp_ld, dtp, tp_ld);
    return check_launch("fusion_mix_backward");
}

int nvq_mask_blend(const float* frame, const float* rec, int rec_ld, const float* mask, int N, int C, int H, int W, float* out,
                   void* stream) {
    NVQ_REQUIRE(C <= rec_ld, "mask_blend: C %d ld %d", C, rec_ld);
    const long HW = (long)H * W, total = (long)N * C * HW;
    hipLaunchKernelGGL(mask_blend_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, frame, rec, rec_ld, mask, C, HW,
                       total, out);
    return check_launch("mask_blend");
}

int nvq_mask_blend_backward(const float* dout, const float* mask, int N, int C, int H, int W, float* drec, int rec_ld,
                            void* stream) {
    NVQ_REQUIRE(C <= rec_ld, "mask_blend_backward: C %d ld %d", C, rec_ld);
    const long HW = (long)H * W, total = (long)N * HW;
    hipLaunchKernelGGL(mask_blend_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dout, mask, C, HW, total,
                       drec, rec_ld);
    return check_launch("mask_blend_backward");
}

int nvq_stem7_forward(const float* x, const float* w, int N, int H, int W, int Co, float* out, int out_ld, int out_bf16,
                      void* stream) {
    NVQ_REQUIRE(Co > 0 && Co % 4 == 0 && Co <= out_ld && aligned16(x) && aligned16(out) && out_ld % 4 == 0, "stem7_forward: Co %d", Co);
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    const int tx = ceil_div(OW, ST_T), ty = ceil_div(OH, ST_T);
    const size_t lds = (size_t)(ST_P * ST_P * 4 + 196 * 64) * sizeof(float);
    hipLaunchKernelGGL(stem7_fwd_kernel, dim3(tx * ty, N), dim3(256), lds, (hipStream_t)stream, x, w, H, W, OH, OW, Co, tx, out, out_ld,
                       out_bf16);
    return check_launch("stem7_forward");
}

int nvq_stem7_wgrad(const float* x, const float* dy, int dy_ld, int N, int H, int W, int Co, float* dw, float* workspace,
                    size_t workspace_bytes, int dy_bf16, void* stream) {
    NVQ_REQUIRE(Co > 0 && Co <= dy_ld && aligned16(x), "stem7_wgrad: Co %d", Co);
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    const int tx = ceil_div(OW, ST_T), ty = ceil_div(OH, ST_T);
    long nt = (long)N * tx * ty;
    const int nblk = nt < 256 ? (int)nt : 256;
    NVQ_REQUIRE(workspace_bytes >= (size_t)nblk * 64 * 196 * sizeof(float), "stem7_wgrad: workspace");
    hipStream_t s = (hipStream_t)stream;
    for (int co0 = 0; co0 < Co; co0 += 64) {
        hipLaunchKernelGGL(stem7_wgrad_kernel, dim3(nblk), dim3(256), 0, s, x, dy, dy_ld, N, H, W, OH, OW, Co, co0, tx, ty, workspace,
                           dy_bf16);
        int rc = check_launch("stem7_wgrad");
        if (rc) return rc;
        const int cn = Co - co0 < 64 ? Co - co0 : 64;
        rc = launch_reduce_partials(workspace, nblk, cn * 196, 1.f, dw + (size_t)co0 * 196, 0, s);
        if (rc) return rc;
    }
    return NVQ_OK;
}

}  // extern "C"
