// Memory-bound stages of the feature extractor (head conv, depthwise conv, BatchNorm) and
// slice helpers.  fp32 NHWC; one thread = 4 consecutive channels of one pixel, so a pixel's
// C floats are read/written by C/4 adjacent lanes as contiguous 16-byte pieces.
#include "common.h"

namespace nvq {

struct SlotMap { int t[NVQ_MAX_T]; };

// ---------------------------------------------------------------- head conv (NCHW image -> NHWC features)
// Work item = one row segment of HEAD_SEG pixels x 4 output channels.  A thread walks its segment with a 3-column
// window of the CIN x 3 input rows in registers (CIN*3 image loads per pixel instead of CIN*9, each a broadcast to the
// F/4 lanes of the pixel) and keeps its 4 x CIN*9 weights in registers for the whole (persistent) launch.
constexpr int HEAD_SEG = 32;

template <int CIN>
struct HeadWindow {
    const float* row[CIN][3];   // row pointers (clamped); rv = row inside the image
    bool rv[3];
    float v[CIN][3][3];
    int W;
    __device__ __forceinline__ void open(const float* img, int H, int W_, int y) {
        W = W_;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            rv[dy] = yy >= 0 && yy < H;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) row[ci][dy] = img + ((size_t)ci * H + (rv[dy] ? yy : 0)) * W;
        }
    }
    __device__ __forceinline__ void load_col(int slot, int xx) {      // column xx of the window into slot
        const bool ok = xx >= 0 && xx < W;
        const int xc = ok ? xx : 0;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float t = row[ci][dy][xc];
                v[ci][dy][slot] = (ok && rv[dy]) ? t : 0.f;
            }
    }
    __device__ __forceinline__ void shift() {
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) { v[ci][dy][0] = v[ci][dy][1]; v[ci][dy][1] = v[ci][dy][2]; }
    }
};

template <int CIN>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ frames, int B, int T,
                                                       int H, int W, SlotMap sm,
                                                       const float* __restrict__ weight,
                                                       const float* __restrict__ bias, int F,
                                                       float* __restrict__ out, int out_ld, int out_bf16,
                                                       __bf16* __restrict__ img8, long nseg, int segsX) {
    constexpr int K = CIN * 9;
    const int F4 = F >> 2;
    const int npl = 256 / F4;
    const int c4 = threadIdx.x % F4;
    const int pl = threadIdx.x / F4;
    float4 w[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
        w[k] = make_float4(weight[(4 * c4 + 0) * K + k], weight[(4 * c4 + 1) * K + k], weight[(4 * c4 + 2) * K + k],
                           weight[(4 * c4 + 3) * K + k]);
    const float4 bv = ld4(bias + 4 * c4);
    HeadWindow<CIN> win;
    for (long item = (long)blockIdx.x * npl + pl; item < nseg; item += (long)gridDim.x * npl) {
        const int xs = item % segsX;
        const int y = (item / segsX) % H;
        const int n = item / ((long)segsX * H);
        const int slot = n / B, b = n - slot * B;
        win.open(frames + ((size_t)(b * T + sm.t[slot]) * CIN) * H * W, H, W, y);
        const int x0 = xs * HEAD_SEG, x1 = min(x0 + HEAD_SEG, W);
        win.load_col(0, x0 - 1);
        win.load_col(1, x0);
        const size_t orow = ((size_t)(n * H + y) * W) * out_ld + 4 * c4;
        for (int x = x0; x < x1; ++x) {
            win.load_col(2, x + 1);
            float4 acc = bv;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float v = win.v[ci][dy][dx];
                        const float4 wk = w[ci * 9 + dy * 3 + dx];
                        acc.x += v * wk.x; acc.y += v * wk.y; acc.z += v * wk.z; acc.w += v * wk.w;
                    }
            acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
            stx4(out, orow + (size_t)x * out_ld, out_bf16, acc);
            if (img8 && c4 == 0) {                            // the frame itself as bf16 NHWC-8 (slot order), see nvq.h
                bf16x8 px = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) px[ci] = (__bf16)win.v[ci][1][1];
                *reinterpret_cast<bf16x8*>(img8 + ((size_t)(n * H + y) * W + x) * 8) = px;
            }
            win.shift();
        }
    }
}

// ---------------------------------------------------------------- head conv on the matrix cores (NVQ_MATH_BF16, 3 channels)
// The 27-term contraction of the 3 -> F head conv as v_mfma_f32_16x16x32_bf16 (weights rounded to bf16, the frame taken as a
// hi + lo pair of bf16 values): K index k = dy*12 + dx*4 + ci over a halo
// tile staged in LDS as bf16 [18][66][4] (3 channels + a zero: 8 B per pixel), so that a lane's 8 consecutive k are two
// 8-byte pieces = two pixels of the tile; k = 32..35 (the last pixel of the window) takes a second MFMA whose other
// operands are zero.  Output rows are PERMUTED: row i of block cb is channel (i/4)*(F/4) + 4*cb + i%4, so lane (pixel c,
// group g) ends up with the F/4 consecutive channels [g*F/4, (g+1)*F/4) of its pixel and stores them as 16-byte pieces.
// The fp32 kernel above issues 27 FMAs per 4 outputs and ran at 0.95 TB/s of output; this one is bound by its stores.
constexpr int HM_TH = 16, HM_TW = 64, HM_HW = HM_TW + 2, HM_HH = HM_TH + 2;

template <int NB>
__global__ __launch_bounds__(256) void head_mfma_kernel(const float* __restrict__ frames, int B, int T, int H, int W,
                                                        SlotMap sm, const float* __restrict__ weight,
                                                        const float* __restrict__ bias, float* __restrict__ out, int out_ld,
                                                        int out_bf16, __bf16* __restrict__ img8, int tilesX, int tilesY) {
    constexpr int F = 16 * NB, FQ = F / 4;
    // the frame as hi + lo bf16 pairs (x = hi + lo to ~16 bits): the image is the one operand of the network that is not
    // already a rounded quantity, and the second pair of MFMAs costs nothing next to the stores
    __shared__ __attribute__((aligned(16))) bf16x4 xs[HM_HH * HM_HW + 4];
    __shared__ __attribute__((aligned(16))) bf16x4 xl[HM_HH * HM_HW + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    int bt = blockIdx.x;
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int slot = n / B, b = n - slot * B;
    const float* img = frames + ((size_t)(b * T + sm.t[slot]) * 3) * H * W;

    // weights: A fragments of both K steps, kept for the whole tile
    auto wk = [&](int co, int k) -> float {
        const int dy = k / 12, j = k - dy * 12, dx = j >> 2, ci = j & 3;
        return (k < 36 && ci < 3) ? weight[co * 27 + ci * 9 + dy * 3 + dx] : 0.f;
    };
    bf16x8 a0[NB], a1[NB];
    float4 bv[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        const int co = (c >> 2) * FQ + 4 * cb + (c & 3);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a0[cb][e] = (__bf16)wk(co, 8 * g + e);
            a1[cb][e] = (__bf16)wk(co, 32 + 8 * g + e);
        }
        bv[cb] = ld4(bias + g * FQ + 4 * cb);
    }
    // halo tile -> LDS
    for (int item = tid; item < HM_HH * HM_HW; item += 256) {
        const int hy = item / HM_HW, hx = item - hy * HM_HW;
        const int gy = ty * HM_TH + hy - 1, gx = tx * HM_TW + hx - 1;
        const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = ok ? (size_t)gy * W + gx : 0;
        const float v0 = img[o], v1 = img[(size_t)H * W + o], v2 = img[(size_t)2 * H * W + o];
        const bf16x4 hi = ok ? (bf16x4){(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)0.f}
                             : (bf16x4){(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        xs[item] = hi;
        xl[item] = ok ? (bf16x4){(__bf16)(v0 - (float)hi[0]), (__bf16)(v1 - (float)hi[1]), (__bf16)(v2 - (float)hi[2]), (__bf16)0.f}
                      : (bf16x4){(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    }
    __syncthreads();
    // pixel offsets (in pixels, relative to the window's top-left) of this lane's two 8-byte pieces of K step 0
    const int o0 = g == 0 ? 0 : g == 1 ? 2 : g == 2 ? HM_HW + 1 : 2 * HM_HW;
    const int o1 = g == 0 ? 1 : g == 1 ? HM_HW : g == 2 ? HM_HW + 2 : 2 * HM_HW + 1;
    const bf16x4 z4 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll 2
    for (int k = 0; k < 16; ++k) {
        const int bi = wave * 16 + k;
        const int ry = bi >> 2, xb = (bi & 3) * 16;
        const int base = ry * HM_HW + xb + c;
        const bf16x4 p0 = xs[base + o0], p1 = xs[base + o1];
        const bf16x4 p2 = g == 0 ? xs[base + 2 * HM_HW + 2] : z4;
        const bf16x8 b0 = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
        const bf16x8 b1 = {p2[0], p2[1], p2[2], p2[3], z4[0], z4[1], z4[2], z4[3]};
        const bf16x4 q0 = xl[base + o0], q1 = xl[base + o1];
        const bf16x4 q2 = g == 0 ? xl[base + 2 * HM_HW + 2] : z4;
        const bf16x8 c0 = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
        const bf16x8 c1 = {q2[0], q2[1], q2[2], q2[3], z4[0], z4[1], z4[2], z4[3]};
        const int gy = ty * HM_TH + ry, gx = tx * HM_TW + xb + c;
        const bool ok = gy < H && gx < W;
        const size_t pix = (size_t)(n * H + gy) * W + gx;
        f32x4 acc[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            acc[cb] = (f32x4){bv[cb].x, bv[cb].y, bv[cb].z, bv[cb].w};
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[cb], c0, acc[cb], 0, 0, 0);   // small terms first
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[cb], c1, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[cb], b0, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[cb], b1, acc[cb], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[cb][e] = fmaxf(acc[cb][e], 0.f);
        }
        if (ok) {
            if (out_bf16) {
                __bf16* o16 = reinterpret_cast<__bf16*>(out) + pix * out_ld + g * FQ;
                if constexpr (NB == 1) {
                    *reinterpret_cast<bf16x4*>(o16) = (bf16x4){(__bf16)acc[0][0], (__bf16)acc[0][1], (__bf16)acc[0][2], (__bf16)acc[0][3]};
                } else {
#pragma unroll
                    for (int cb = 0; cb < NB; cb += 2)
                        *reinterpret_cast<bf16x8*>(o16 + 4 * cb) =
                            (bf16x8){(__bf16)acc[cb][0], (__bf16)acc[cb][1], (__bf16)acc[cb][2], (__bf16)acc[cb][3],
                                     (__bf16)acc[cb + 1][0], (__bf16)acc[cb + 1][1], (__bf16)acc[cb + 1][2], (__bf16)acc[cb + 1][3]};
                }
            } else {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb)
                    st4(out + pix * out_ld + g * FQ + 4 * cb, make_float4(acc[cb][0], acc[cb][1], acc[cb][2], acc[cb][3]));
            }
            if (img8 && g == 0) {                          // the frame itself as bf16 NHWC-8 (slot order), see nvq.h
                const bf16x4 px = xs[base + HM_HW + 1];
                *reinterpret_cast<bf16x8*>(img8 + pix * 8) = (bf16x8){px[0], px[1], px[2], z4[0], z4[0], z4[0], z4[0], z4[0]};
            }
        }
    }
}

// Cross pixel-lane reduction of a float4 inside a 256-thread block whose threads are laid out
// as (pixel lane, c4) with C4 lanes per pixel.  Threads with pixel lane 0 get the sum.
__device__ __forceinline__ float4 plane_reduce4(float4 v, float4* buf, int C4, int npl) {
    __syncthreads();
    buf[threadIdx.x] = v;
    __syncthreads();
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((int)threadIdx.x < C4) {
        for (int k = 0; k < npl; ++k) {
            const float4 t = buf[k * C4 + threadIdx.x];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
    }
    return s;
}

// dW[co][ci][tap], db[co] partials: part[blk][F*CIN*9 + F]; same segment walk as head_fwd_kernel
template <int CIN, int ACT_BF16>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ frames, int B, int T,
                                                         int H, int W, SlotMap sm,
                                                         const float* __restrict__ dout, int dout_ld,
                                                         const float* __restrict__ dout2, int dout2_ld,
                                                         const float* __restrict__ act, int act_ld,
                                                         int F, long nseg, int segsX, float* __restrict__ part) {
    __shared__ float4 buf[256];
    constexpr int K = CIN * 9;
    const int F4 = F >> 2;
    const int npl = 256 / F4;  // F4 is a power of two <= 64
    const int c4 = threadIdx.x % F4;
    const int pl = threadIdx.x / F4;
    float4 acc[K + 1];
#pragma unroll
    for (int k = 0; k <= K; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    HeadWindow<CIN> win;
    for (long item = (long)blockIdx.x * npl + pl; item < nseg; item += (long)gridDim.x * npl) {
        const int xs = item % segsX;
        const int y = (item / segsX) % H;
        const int n = item / ((long)segsX * H);
        const int slot = n / B, b = n - slot * B;
        win.open(frames + ((size_t)(b * T + sm.t[slot]) * CIN) * H * W, H, W, y);
        const int x0 = xs * HEAD_SEG, x1 = min(x0 + HEAD_SEG, W);
        win.load_col(0, x0 - 1);
        win.load_col(1, x0);
        const size_t prow0 = (size_t)(n * H + y) * W;
        auto ldg = [&](size_t pp) -> float4 {                 // gradient of the head features: dout (+ dout2, the skip path)
            float4 v = ld4(dout + pp * dout_ld + 4 * c4);
            if (dout2) {
                const float4 u = ld4(dout2 + pp * dout2_ld + 4 * c4);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            return v;
        };
        float4 g = ldg(prow0 + x0);
        float4 a = ldx4(act, (prow0 + x0) * act_ld + 4 * c4, ACT_BF16);
        for (int x = x0; x < x1; ++x) {
            win.load_col(2, x + 1);
            const int xn = x + 1 < x1 ? x + 1 : x;              // next pixel's gradient / activation, one ahead
            const float4 gn = ldg(prow0 + xn);
            const float4 an = ldx4(act, (prow0 + xn) * act_ld + 4 * c4, ACT_BF16);
            if (!(a.x > 0.f)) g.x = 0.f;
            if (!(a.y > 0.f)) g.y = 0.f;
            if (!(a.z > 0.f)) g.z = 0.f;
            if (!(a.w > 0.f)) g.w = 0.f;
            acc[K].x += g.x; acc[K].y += g.y; acc[K].z += g.z; acc[K].w += g.w;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float v = win.v[ci][dy][dx];
                        float4& s = acc[ci * 9 + dy * 3 + dx];
                        s.x += v * g.x; s.y += v * g.y; s.z += v * g.z; s.w += v * g.w;
                    }
            g = gn; a = an;
            win.shift();
        }
    }
    float* prow = part + (size_t)blockIdx.x * (F * K + F);
#pragma unroll
    for (int k = 0; k <= K; ++k) {
        const float4 s = plane_reduce4(acc[k], buf, F4, npl);
        if ((int)threadIdx.x < F4) {
            const int co = 4 * c4;
            if (k < K) {
                prow[(co + 0) * K + k] = s.x; prow[(co + 1) * K + k] = s.y;
                prow[(co + 2) * K + k] = s.z; prow[(co + 3) * K + k] = s.w;
            } else {
                st4(prow + F * K + co, s);
            }
        }
    }
}

// ---------------------------------------------------------------- depthwise 3x3
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ in, int in_ld,
                                                     const float* __restrict__ weight, int C,
                                                     float* __restrict__ out, int out_ld, int H, int W,
                                                     int flip, int in_bf16, int out_bf16, long total) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [9][C]
    for (int i = threadIdx.x; i < 9 * C; i += 256) {
        const int c = i % C, tap = i / C;
        wl[i] = weight[c * 9 + (flip ? 8 - tap : tap)];
    }
    __syncthreads();
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const int C4 = C >> 2;
    const long pixlin = idiv(gid, C4, total);
    const int c4 = (int)(gid - pixlin * C4);
    const long rowi = idiv(pixlin, W, total);
    const int x = (int)(pixlin - rowi * W);
    const int y = (int)(rowi - idiv(rowi, H, total) * H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = x + dx - 1;
            if (xx < 0 || xx >= W) continue;
            const float4 v = ldx4(in, (size_t)(pixlin + (long)(dy - 1) * W + (dx - 1)) * in_ld + 4 * c4, in_bf16);
            const float4 w = ld4(wl + (dy * 3 + dx) * C + 4 * c4);
            acc.x += v.x * w.x; acc.y += v.y * w.y; acc.z += v.z * w.z; acc.w += v.w * w.w;
        }
    }
    stx4(out, (size_t)pixlin * out_ld + 4 * c4, out_bf16, acc);
}

// part[blk][C*9]
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x, int x_ld,
                                                           const float* __restrict__ dy, int dy_ld, int C,
                                                           int H, int W, long npix, int x_bf16, int dy_bf16,
                                                           float* __restrict__ part) {
    __shared__ float4 buf[256];
    const int C4 = C >> 2;
    const int npl = 256 / C4;
    const int c4 = threadIdx.x % C4;
    const int pl = threadIdx.x / C4;
    float4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long p = (long)blockIdx.x * npl + pl; p < npix; p += (long)gridDim.x * npl) {
        const long rowi = idiv(p, W, npix);
        const int xx0 = (int)(p - rowi * W);
        const int yy0 = (int)(rowi - idiv(rowi, H, npix) * H);
        const float4 g = ldx4(dy, (size_t)p * dy_ld + 4 * c4, dy_bf16);
#pragma unroll
        for (int ddy = 0; ddy < 3; ++ddy) {
            const int yy = yy0 + ddy - 1;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int ddx = 0; ddx < 3; ++ddx) {
                const int xx = xx0 + ddx - 1;
                if (xx < 0 || xx >= W) continue;
                const float4 v = ldx4(x, (size_t)(p + (long)(ddy - 1) * W + (ddx - 1)) * x_ld + 4 * c4, x_bf16);
                float4& s = acc[ddy * 3 + ddx];
                s.x += v.x * g.x; s.y += v.y * g.y; s.z += v.z * g.z; s.w += v.w * g.w;
            }
        }
    }
    float* prow = part + (size_t)blockIdx.x * C * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float4 s = plane_reduce4(acc[k], buf, C4, npl);
        if ((int)threadIdx.x < C4) {
            const int c = 4 * c4;
            prow[(c + 0) * 9 + k] = s.x; prow[(c + 1) * 9 + k] = s.y;
            prow[(c + 2) * 9 + k] = s.z; prow[(c + 3) * 9 + k] = s.w;
        }
    }
}

// ---------------------------------------------------------------- depthwise 3x3, LDS-tiled (C % 32 == 0)
// Workgroup = 8x32-pixel tile x 32 channels.  The halo tile sits in LDS as fp32 [340 px][32 ch]; thread
// (c4 = tid & 7, x = tid >> 3) walks its column down the 8 rows with a 3-row window, so every input value is
// read from LDS ~1.25 times instead of 9 global loads per output (the untiled kernel is bound by load
// instructions, not bytes: halving the bytes with bf16 storage did not change its time).
constexpr int DT_H = 8, DT_W = 32, DT_C = 32;
constexpr int DT_HW = DT_W + 2, DT_HH = DT_H + 2, DT_NPIX = DT_HW * DT_HH;

__device__ __forceinline__ void dw_stage_halo(const float* __restrict__ in, int in_ld, int in_bf16, int n, int H, int W,
                                              int ty0, int tx0, int ch0, float* xs) {
    if (in_bf16) {
        for (int item = threadIdx.x; item < DT_NPIX * 4; item += 256) {      // 8 bf16 channels per piece
            const int hp = item >> 2, q = item & 3;
            const int hy = hp / DT_HW, hx = hp - hy * DT_HW;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const size_t idx = ((size_t)(n * H + gy) * W + gx) * in_ld + ch0 + 8 * q;
                a = ldx4(in, idx, 1);
                b = ldx4(in, idx + 4, 1);
            }
            st4(xs + hp * DT_C + 8 * q, a);
            st4(xs + hp * DT_C + 8 * q + 4, b);
        }
    } else {
        for (int item = threadIdx.x; item < DT_NPIX * 8; item += 256) {
            const int hp = item >> 3, q = item & 7;
            const int hy = hp / DT_HW, hx = hp - hy * DT_HW;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                a = ld4(in + ((size_t)(n * H + gy) * W + gx) * in_ld + ch0 + 4 * q);
            st4(xs + hp * DT_C + 4 * q, a);
        }
    }
}

// grid (tiles, C/32)
__global__ __launch_bounds__(256) void dwconv_tiled_kernel(const float* __restrict__ in, int in_ld,
                                                           const float* __restrict__ weight, int C,
                                                           float* __restrict__ out, int out_ld, int H, int W,
                                                           int tilesX, int tilesY, int flip, int in_bf16, int out_bf16) {
    __shared__ __attribute__((aligned(16))) float xs[DT_NPIX * DT_C];
    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int ch0 = blockIdx.y * DT_C;
    const int c4 = threadIdx.x & 7, x = threadIdx.x >> 3;
    float4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tt = flip ? 8 - t : t;
        const int c = ch0 + 4 * c4;
        w[t] = make_float4(weight[(c + 0) * 9 + tt], weight[(c + 1) * 9 + tt], weight[(c + 2) * 9 + tt], weight[(c + 3) * 9 + tt]);
    }
    dw_stage_halo(in, in_ld, in_bf16, n, H, W, ty * DT_H, tx * DT_W, ch0, xs);
    __syncthreads();
    const int gx = tx * DT_W + x;
    float4 r[3][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) r[a][b] = ld4(xs + (a * DT_HW + x + b) * DT_C + 4 * c4);
#pragma unroll
    for (int y = 0; y < DT_H; ++y) {
#pragma unroll
        for (int b = 0; b < 3; ++b) r[2][b] = ld4(xs + ((y + 2) * DT_HW + x + b) * DT_C + 4 * c4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float4 v = r[a][b], ww = w[a * 3 + b];
                acc.x += v.x * ww.x; acc.y += v.y * ww.y; acc.z += v.z * ww.z; acc.w += v.w * ww.w;
            }
        const int gy = ty * DT_H + y;
        if (gy < H && gx < W) stx4(out, ((size_t)(n * H + gy) * W + gx) * out_ld + ch0 + 4 * c4, out_bf16, acc);
#pragma unroll
        for (int b = 0; b < 3; ++b) { r[0][b] = r[1][b]; r[1][b] = r[2][b]; }
    }
}

// grid (<= 512 persistent blocks over tiles, C/32): part[blockIdx.x][C*9]
__global__ __launch_bounds__(256) void dwconv_wgrad_tiled_kernel(const float* __restrict__ x, int x_ld,
                                                                 const float* __restrict__ dy, int dy_ld, int C, int H,
                                                                 int W, int tilesX, int tilesY, int ntiles, int x_bf16,
                                                                 int dy_bf16, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float xs[DT_NPIX * DT_C];
    const int ch0 = blockIdx.y * DT_C;
    const int c4 = threadIdx.x & 7, xx = threadIdx.x >> 3;
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int bt = xcd_tile(tile, ntiles);
        const int tx = bt % tilesX; bt /= tilesX;
        const int ty = bt % tilesY;
        const int n = bt / tilesY;
        __syncthreads();
        dw_stage_halo(x, x_ld, x_bf16, n, H, W, ty * DT_H, tx * DT_W, ch0, xs);
        __syncthreads();
        const int gx = tx * DT_W + xx;
        float4 r[3][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) r[a][b] = ld4(xs + (a * DT_HW + xx + b) * DT_C + 4 * c4);
#pragma unroll
        for (int y = 0; y < DT_H; ++y) {
#pragma unroll
            for (int b = 0; b < 3; ++b) r[2][b] = ld4(xs + ((y + 2) * DT_HW + xx + b) * DT_C + 4 * c4);
            const int gy = ty * DT_H + y;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < H && gx < W) g = ldx4(dy, ((size_t)(n * H + gy) * W + gx) * dy_ld + ch0 + 4 * c4, dy_bf16);
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float4 v = r[a][b];
                    float4& s = acc[a * 3 + b];
                    s.x += v.x * g.x; s.y += v.y * g.y; s.z += v.z * g.z; s.w += v.w * g.w;
                }
#pragma unroll
            for (int b = 0; b < 3; ++b) { r[0][b] = r[1][b]; r[1][b] = r[2][b]; }
        }
    }
    // reduce over the 32 x-lanes that share c4 (threads c4 + 8k), through LDS
    float4* buf = reinterpret_cast<float4*>(xs);
    float* prow = part + (size_t)blockIdx.x * C * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
        buf[threadIdx.x] = acc[t];
        __syncthreads();
        if (threadIdx.x < 8) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < 32; ++k) {
                const float4 v = buf[threadIdx.x + 8 * k];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const int c = ch0 + 4 * threadIdx.x;
            prow[(c + 0) * 9 + t] = s.x; prow[(c + 1) * 9 + t] = s.y;
            prow[(c + 2) * 9 + t] = s.z; prow[(c + 3) * 9 + t] = s.w;
        }
    }
}

// ---------------------------------------------------------------- depthwise 3x3, bf16 input, 64-channel tiles
// Same tile walk as above for bf16-stored inputs with C % 64 == 0: the workgroup takes 64 channels, i.e. the whole
// 128-B line of a pixel (the 32-channel tiles split every line between two workgroups), the halo tile sits in LDS as
// bf16 [340 px][64 ch]; 512 threads, thread (c4 = tid & 15, x = tid >> 4) owns 4 channels of its column.
constexpr int DB_C = 64, DB_T = 512;
typedef unsigned dw_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned dw_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float4 dw_unpack4(dw_u32x2 v) {
    return make_float4(__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                       __uint_as_float(v[1] & 0xffff0000u));
}

// Optional on-the-fly input transform of the 64-channel depthwise kernels: x := relu(bn(x)) with the statistics of the
// image's group - the BatchNorm + ReLU between a pointwise conv and the next depthwise conv, applied while the halo is
// staged instead of in a pass of its own (same expression and the same single bf16 rounding as bn_apply_relu_kernel, so the
// staged values are the ones that kernel would have stored).  Zero padding stays zero.
// Optional epilogue of dwconv_bf16_kernel: out = (conv + add) masked by (mask > 0); add is fp32, mask bf16 or fp32.
struct DwEpi { const float* add; int add_ld; const float* mask; int mask_ld; int mask_bf16; int add_bf16; };
struct DwBn { const float* mean; const float* invstd; const float* gamma; const float* beta; int group_images; int C; };

__device__ __forceinline__ void dw_stage_halo_bf16(const __bf16* __restrict__ in, int in_ld, int n, int H, int W, int ty0,
                                                   int tx0, int ch0, __bf16* xs, const DwBn bn) {
    constexpr int PER = (DT_NPIX * 8 + DB_T - 1) / DB_T;
    dw_u32x4 v[PER];
    bool okv[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {                      // all loads first (clamped addresses), then the stores
        const int item = threadIdx.x + k * DB_T;
        const int hp = item >> 3, q = item & 7;
        const int hy = hp / DT_HW, hx = hp - hy * DT_HW;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool ok = item < DT_NPIX * 8 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        okv[k] = ok;
        const size_t idx = ok ? ((size_t)(n * H + gy) * W + gx) * in_ld + ch0 + 8 * q : 0;
        v[k] = *reinterpret_cast<const dw_u32x4*>(in + idx);
        if (!ok) v[k] = (dw_u32x4){0u, 0u, 0u, 0u};
    }
    if (bn.mean) {                                        // uniform
        const int c0 = ch0 + 8 * (threadIdx.x & 7);      // this thread's channel group (DB_T % 8 == 0)
        const int g = n / bn.group_images;
        float m[8], is[8], ga[8], be[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            m[e] = bn.mean[g * bn.C + c0 + e]; is[e] = bn.invstd[g * bn.C + c0 + e];
            ga[e] = bn.gamma[c0 + e]; be[e] = bn.beta[c0 + e];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (!okv[k]) continue;
            dw_u32x4 o;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                const float a = __uint_as_float(v[k][w2] << 16), b = __uint_as_float(v[k][w2] & 0xffff0000u);
                const float ya = fmaxf((a - m[2 * w2]) * is[2 * w2] * ga[2 * w2] + be[2 * w2], 0.f);
                const float yb = fmaxf((b - m[2 * w2 + 1]) * is[2 * w2 + 1] * ga[2 * w2 + 1] + be[2 * w2 + 1], 0.f);
                typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                const b2 pk = {(__bf16)ya, (__bf16)yb};
                o[w2] = __builtin_bit_cast(unsigned, pk);
            }
            v[k] = o;
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int item = threadIdx.x + k * DB_T;
        if (item < DT_NPIX * 8) *reinterpret_cast<dw_u32x4*>(xs + (item >> 3) * DB_C + 8 * (item & 7)) = v[k];
    }
}

__device__ __forceinline__ float4 dw_lds4(const __bf16* xs, int hp, int c4) {
    return dw_unpack4(*reinterpret_cast<const dw_u32x2*>(xs + hp * DB_C + 4 * c4));
}

// grid (<= DB_MAXWG persistent workgroups over the tiles, C/64): the next tile's halo is fetched into registers while the
// current tile is computed from LDS, so a workgroup always has loads in flight (the one-tile-per-workgroup form spent the
// whole memory latency of every tile in its staging phase, with only the CU's second workgroup to cover it).
constexpr int DB_MAXWG = 512;
// HAS_BN / HAS_EPI: the optional input transform / output epilogue as template parameters - as run-time branches their
// registers (32 for the BatchNorm parameters) counted against the 128-register budget of every launch.
// EPI: 0 no epilogue, 1 the general one (storage types as run-time flags), 2 add and mask both present and both bf16 - their
// two 8-byte loads are then issued together (behind a run-time flag each is a branch that waits for its own load).
// (HAS_BN with EPI = 2 needs 131 registers: three waves per SIMD instead of 12 bytes of scratch behind the prefetch)
template <bool HAS_BN, int EPI>
__global__ __launch_bounds__(DB_T, (HAS_BN && EPI == 2 ? 3 : 4)) void dwconv_bf16_kernel(const __bf16* __restrict__ in, int in_ld,
                                                           const float* __restrict__ weight, int C,
                                                           float* __restrict__ out, int out_ld, int H, int W, int tilesX,
                                                           int tilesY, int ntiles, int flip, int out_bf16, DwBn bn, DwEpi ep) {
    __shared__ __attribute__((aligned(16))) __bf16 xs[DT_NPIX * DB_C];
    constexpr int PER = (DT_NPIX * 8 + DB_T - 1) / DB_T;
    const int ch0 = blockIdx.y * DB_C;
    const int c4 = threadIdx.x & 15, x = threadIdx.x >> 4;
    float4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tt = flip ? 8 - t : t;
        const int c = ch0 + 4 * c4;
        w[t] = make_float4(weight[(c + 0) * 9 + tt], weight[(c + 1) * 9 + tt], weight[(c + 2) * 9 + tt], weight[(c + 3) * 9 + tt]);
    }
    dw_u32x4 v[PER];
    unsigned okm = 0;
    auto fetch = [&](int tile) {                              // raw loads (clamped addresses); masked at commit
        int bt = xcd_tile(tile, ntiles);
        const int tx = bt % tilesX; bt /= tilesX;
        const int ty = bt % tilesY;
        const int n = bt / tilesY;
        okm = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int item = threadIdx.x + k * DB_T;
            const int hp = item >> 3, q = item & 7;
            const int hy = hp / DT_HW, hx = hp - hy * DT_HW;
            const int gy = ty * DT_H + hy - 1, gx = tx * DT_W + hx - 1;
            const bool ok = item < DT_NPIX * 8 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            okm |= (ok ? 1u : 0u) << k;
            v[k] = *reinterpret_cast<const dw_u32x4*>(in + (ok ? ((size_t)(n * H + gy) * W + gx) * in_ld + ch0 + 8 * q : 0));
        }
    };
    auto commit = [&](int n) {                                // (optional relu(bn(.))) -> LDS; zero padding stays zero
        if constexpr (HAS_BN) {                                        // uniform
            const int c0 = ch0 + 8 * (threadIdx.x & 7);       // this thread's channel group (DB_T % 8 == 0)
            const int g = n / bn.group_images;
            float m[8], is[8], ga[8], be[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                m[e] = bn.mean[g * bn.C + c0 + e]; is[e] = bn.invstd[g * bn.C + c0 + e];
                ga[e] = bn.gamma[c0 + e]; be[e] = bn.beta[c0 + e];
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                dw_u32x4 o;
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) {
                    const float a = __uint_as_float(v[k][w2] << 16), b = __uint_as_float(v[k][w2] & 0xffff0000u);
                    const float ya = fmaxf((a - m[2 * w2]) * is[2 * w2] * ga[2 * w2] + be[2 * w2], 0.f);
                    const float yb = fmaxf((b - m[2 * w2 + 1]) * is[2 * w2 + 1] * ga[2 * w2 + 1] + be[2 * w2 + 1], 0.f);
                    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                    const b2 pk = {(__bf16)ya, (__bf16)yb};
                    o[w2] = __builtin_bit_cast(unsigned, pk);
                }
                v[k] = o;
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int item = threadIdx.x + k * DB_T;
            dw_u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (okm >> k) & 1 ? v[k][e] : 0u;
            if (item < DT_NPIX * 8) *reinterpret_cast<dw_u32x4*>(xs + (item >> 3) * DB_C + 8 * (item & 7)) = o;
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        int bt = xcd_tile(tile, ntiles);
        const int tx = bt % tilesX; bt /= tilesX;
        const int ty = bt % tilesY;
        const int n = bt / tilesY;
        commit(n);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        const int gx = tx * DT_W + x;
        // Input-row stationary: halo row r feeds the three output rows r, r - 1, r - 2 (tap rows 0, 1, 2), so only one row of
        // the window (3 taps) and three accumulators are live instead of a 3x3 window - the kernel sits at its 128-register
        // budget (weights 36, prefetched halo pieces 24) and the window's 36 registers pushed the prefetched pieces into
        // scratch, i.e. made every prefetch wait for its own loads.  Same summation order per output (tap row major).
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;      // output rows r, r - 1, r - 2
#pragma unroll 1
        for (int r = 0; r < DT_H + 2; ++r) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float4 vv = dw_lds4(xs, r * DT_HW + x + b, c4);
                if (r < DT_H) { const float4 ww = w[b]; a0.x = __builtin_fmaf(vv.x, ww.x, a0.x); a0.y = __builtin_fmaf(vv.y, ww.y, a0.y); a0.z = __builtin_fmaf(vv.z, ww.z, a0.z); a0.w = __builtin_fmaf(vv.w, ww.w, a0.w); }
                if (r >= 1 && r <= DT_H) { const float4 ww = w[3 + b]; a1.x = __builtin_fmaf(vv.x, ww.x, a1.x); a1.y = __builtin_fmaf(vv.y, ww.y, a1.y); a1.z = __builtin_fmaf(vv.z, ww.z, a1.z); a1.w = __builtin_fmaf(vv.w, ww.w, a1.w); }
                if (r >= 2) { const float4 ww = w[6 + b]; a2.x = __builtin_fmaf(vv.x, ww.x, a2.x); a2.y = __builtin_fmaf(vv.y, ww.y, a2.y); a2.z = __builtin_fmaf(vv.z, ww.z, a2.z); a2.w = __builtin_fmaf(vv.w, ww.w, a2.w); }
            }
            if (r >= 2) {
                const int y = r - 2;
                float4 acc = a2;
                const int gy = ty * DT_H + y;
                if (gy < H && gx < W) {
                    const size_t pix = (size_t)(n * H + gy) * W + gx;
                    if constexpr (EPI == 2) {
                        const dw_u32x2 ra = *reinterpret_cast<const dw_u32x2*>(reinterpret_cast<const __bf16*>(ep.add) + pix * ep.add_ld + ch0 + 4 * c4);
                        const dw_u32x2 rm = *reinterpret_cast<const dw_u32x2*>(reinterpret_cast<const __bf16*>(ep.mask) + pix * ep.mask_ld + ch0 + 4 * c4);
                        const float4 a = dw_unpack4(ra), m = dw_unpack4(rm);
                        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
                        if (!(m.x > 0.f)) acc.x = 0.f;
                        if (!(m.y > 0.f)) acc.y = 0.f;
                        if (!(m.z > 0.f)) acc.z = 0.f;
                        if (!(m.w > 0.f)) acc.w = 0.f;
                    }
                    if (EPI == 1 && ep.add) {
                        const float4 a = ldx4(ep.add, pix * ep.add_ld + ch0 + 4 * c4, ep.add_bf16);
                        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
                    }
                    if (EPI == 1 && ep.mask) {
                        const float4 m = ldx4(ep.mask, pix * ep.mask_ld + ch0 + 4 * c4, ep.mask_bf16);
                        if (!(m.x > 0.f)) acc.x = 0.f;
                        if (!(m.y > 0.f)) acc.y = 0.f;
                        if (!(m.z > 0.f)) acc.z = 0.f;
                        if (!(m.w > 0.f)) acc.w = 0.f;
                    }
                    stx4(out, pix * out_ld + ch0 + 4 * c4, out_bf16, acc);
                }
            }
            a2 = a1; a1 = a0; a0 = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();                                      // everyone is done with xs before the next commit
    }
}

// grid (<= 512 persistent blocks over tiles, C/64): part[blockIdx.x][C*9]; x and dy stored as bf16
__global__ __launch_bounds__(DB_T) void dwconv_wgrad_bf16_kernel(const __bf16* __restrict__ x, int x_ld,
                                                                 const __bf16* __restrict__ dy, int dy_ld, int C, int H,
                                                                 int W, int tilesX, int tilesY, int ntiles,
                                                                 float* __restrict__ part, DwBn bn) {
    __shared__ __attribute__((aligned(16))) __bf16 xs[DT_NPIX * DB_C];
    const int ch0 = blockIdx.y * DB_C;
    const int c4 = threadIdx.x & 15, xx = threadIdx.x >> 4;
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int bt = xcd_tile(tile, ntiles);
        const int tx = bt % tilesX; bt /= tilesX;
        const int ty = bt % tilesY;
        const int n = bt / tilesY;
        const int gx = tx * DT_W + xx;
        dw_u32x2 g[DT_H];                                 // this thread's dy pieces of the tile, loaded up front
#pragma unroll
        for (int y = 0; y < DT_H; ++y) {
            const int gy = ty * DT_H + y;
            const bool ok = gy < H && gx < W;
            g[y] = *reinterpret_cast<const dw_u32x2*>(dy + (ok ? ((size_t)(n * H + gy) * W + gx) * dy_ld + ch0 + 4 * c4 : 0));
            if (!ok) g[y] = (dw_u32x2){0u, 0u};
        }
        __syncthreads();
        dw_stage_halo_bf16(x, x_ld, n, H, W, ty * DT_H, tx * DT_W, ch0, xs, bn);
        __syncthreads();
        float4 r[3][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) r[a][b] = dw_lds4(xs, a * DT_HW + xx + b, c4);
#pragma unroll
        for (int y = 0; y < DT_H; ++y) {
#pragma unroll
            for (int b = 0; b < 3; ++b) r[2][b] = dw_lds4(xs, (y + 2) * DT_HW + xx + b, c4);
            const float4 gf = dw_unpack4(g[y]);
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float4 v = r[a][b];
                    float4& s = acc[a * 3 + b];
                    s.x += v.x * gf.x; s.y += v.y * gf.y; s.z += v.z * gf.z; s.w += v.w * gf.w;
                }
#pragma unroll
            for (int b = 0; b < 3; ++b) { r[0][b] = r[1][b]; r[1][b] = r[2][b]; }
        }
    }
    // sum over the 32 x-lanes that share c4: the 4 inside a wave by shuffles, then the 8 waves through LDS
    __syncthreads();
    float* red = reinterpret_cast<float*>(xs);           // [8 waves][16 c4][36]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float v[4] = {acc[t].x, acc[t].y, acc[t].z, acc[t].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] += __shfl_xor(v[e], 16, 64);
            v[e] += __shfl_xor(v[e], 32, 64);
            if (lane < 16) red[(wave * 16 + lane) * 36 + t * 4 + e] = v[e];
        }
    }
    __syncthreads();
    float* prow = part + (size_t)blockIdx.x * C * 9;
    for (int i = threadIdx.x; i < 16 * 36; i += DB_T) {
        const int ci = i / 36, k = i - ci * 36;           // k = t*4 + e
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) v += red[(wv * 16 + ci) * 36 + k];
        prow[(ch0 + 4 * ci + (k & 3)) * 9 + (k >> 2)] = v;
    }
}

// ---------------------------------------------------------------- BatchNorm
// grid (blocks per group, G): part[g][blk][2C] = {sum, sum of squares}
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int x_ld, int C,
                                                       long group_pix, int x_bf16, float* __restrict__ part) {
    __shared__ float4 buf[256];
    const int C4 = C >> 2;
    const int npl = 256 / C4;
    const int c4 = threadIdx.x % C4;
    const int pl = threadIdx.x / C4;
    const long base = (long)blockIdx.y * group_pix;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    for (long p = (long)blockIdx.x * npl + pl; p < group_pix; p += (long)gridDim.x * npl) {
        const float4 v = ldx4(x, (size_t)(base + p) * x_ld + 4 * c4, x_bf16);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
    }
    float* prow = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
    const float4 rs = plane_reduce4(s, buf, C4, npl);
    if ((int)threadIdx.x < C4) st4(prow + 4 * c4, rs);
    const float4 rq = plane_reduce4(q, buf, C4, npl);
    if ((int)threadIdx.x < C4) st4(prow + C + 4 * c4, rq);
}

struct GroupOrder { int g[NVQ_MAX_T]; };

__device__ __forceinline__ double block_sum_double(double v, double* scratch) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// one 256-thread block per channel; groups are folded into the running statistics in `order`
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nblk, int C, int G,
                                                          long group_pix, float eps, float momentum, GroupOrder order,
                                                          float* __restrict__ mean, float* __restrict__ invstd,
                                                          float* __restrict__ rmean, float* __restrict__ rvar) {
    __shared__ double scratch[4];
    const int c = blockIdx.x;
    float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
    for (int gi = 0; gi < G; ++gi) {
        const int g = order.g[gi];
        const float* p = part + (size_t)g * nblk * 2 * C;
        double s = 0.0, q = 0.0;
        for (int b = threadIdx.x; b < nblk; b += 256) {
            s += (double)p[(size_t)b * 2 * C + c];
            q += (double)p[(size_t)b * 2 * C + C + c];
        }
        s = block_sum_double(s, scratch);
        q = block_sum_double(q, scratch);
        const double m = s / (double)group_pix;
        double var = q / (double)group_pix - m * m;
        if (var < 0.0) var = 0.0;
        const double unb = group_pix > 1 ? var * (double)group_pix / (double)(group_pix - 1) : var;
        rm = (1.f - momentum) * rm + momentum * (float)m;
        rv = (1.f - momentum) * rv + momentum * (float)unb;
        if (threadIdx.x == 0) {
            mean[g * C + c] = (float)m;
            invstd[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    if (threadIdx.x == 0) {
        if (rmean) rmean[c] = rm;
        if (rvar) rvar[c] = rv;
    }
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                     int C, int G, float eps, float* __restrict__ mean,
                                     float* __restrict__ invstd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * G) return;
    const int c = i % C;
    mean[i] = rmean[c];
    invstd[i] = 1.f / sqrtf(rvar[c] + eps);
}

// The two element-wise BatchNorm passes: grid (blocks, groups), EW_ITEMS 4-channel pieces per thread, all 32-bit index
// arithmetic (the first versions spent their time in three 64-bit divisions per 8-byte load: 3.6 TB/s; the reduce kernels,
// which loop without dividing, ran at 5).  256 % (C/4) == 0, so a thread's pieces all have the same channels and the
// per-channel constants are loaded once.
constexpr int EW_ITEMS = 4;

// ALLB: every tensor is stored as bf16, known at compile time.  With run-time storage flags each ldx4 is a branch whose bf16 arm
// converts - i.e. waits for - its own load, so a thread's 2 * EW_ITEMS loads run one after the other.
template <bool ALLB>
__global__ __launch_bounds__(256) void bn_apply_relu_kernel(
    const float* __restrict__ x, int x_ld, int C, int c4_shift, unsigned group_items, long group_pix,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ res, int res_ld, float* __restrict__ outA,
    int outA_ld, int outA_coff, long split_pix, float* __restrict__ outB, int outB_ld, int outB_coff,
    int x_bf16_, int out_bf16_, int res_bf16_) {
    const int x_bf16 = ALLB ? 1 : x_bf16_, out_bf16 = ALLB ? 1 : out_bf16_, res_bf16 = ALLB ? 1 : res_bf16_;
    const int g = blockIdx.y;
    const int c4 = threadIdx.x & ((C >> 2) - 1);
    const float4 m = ld4(mean + g * C + 4 * c4), is = ld4(invstd + g * C + 4 * c4);
    const float4 ga = ld4(gamma + 4 * c4), be = ld4(beta + 4 * c4);
    const long base = (long)g * group_pix;
    float4 v[EW_ITEMS], r[EW_ITEMS];
    long pix[EW_ITEMS];
    bool ok[EW_ITEMS];
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        const unsigned i = (blockIdx.x * EW_ITEMS + k) * 256u + threadIdx.x;
        ok[k] = i < group_items;
        pix[k] = base + (ok[k] ? (i >> c4_shift) : 0u);
        v[k] = ldx4(x, (size_t)pix[k] * x_ld + 4 * c4, x_bf16);
        r[k] = res ? ldx4(res, (size_t)pix[k] * res_ld + 4 * c4, res_bf16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        float4 y;
        y.x = fmaxf((v[k].x - m.x) * is.x * ga.x + be.x, 0.f);
        y.y = fmaxf((v[k].y - m.y) * is.y * ga.y + be.y, 0.f);
        y.z = fmaxf((v[k].z - m.z) * is.z * ga.z + be.z, 0.f);
        y.w = fmaxf((v[k].w - m.w) * is.w * ga.w + be.w, 0.f);
        if (res) { y.x += r[k].x; y.y += r[k].y; y.z += r[k].z; y.w += r[k].w; }
        if (!ok[k]) continue;
        if (pix[k] < split_pix)
            stx4(outA, (size_t)pix[k] * outA_ld + outA_coff + 4 * c4, out_bf16, y);
        else
            stx4(outB, (size_t)(pix[k] - split_pix) * outB_ld + outB_coff + 4 * c4, out_bf16, y);
    }
}

// backward pass 1: part[g][blk][2C] = {sum dyr, sum dyr*xhat}, dyr = dy * [bn(x) > 0]
template <bool DYB, bool XB>                                  // storage types at compile time, see bn_apply_relu_kernel
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float* __restrict__ dy, int dy_ld, const float* __restrict__ x, int x_ld, int C,
    long group_pix, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ part) {
    constexpr int dy_bf16 = DYB, x_bf16 = XB;
    __shared__ float4 buf[256];
    const int C4 = C >> 2;
    const int npl = 256 / C4;
    const int c4 = threadIdx.x % C4;
    const int pl = threadIdx.x / C4;
    const int g = blockIdx.y;
    const long base = (long)g * group_pix;
    const float4 m = ld4(mean + g * C + 4 * c4), is = ld4(invstd + g * C + 4 * c4);
    const float4 ga = ld4(gamma + 4 * c4), be = ld4(beta + 4 * c4);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (long p = (long)blockIdx.x * npl + pl; p < group_pix; p += (long)gridDim.x * npl) {
        const float4 v = ldx4(x, (size_t)(base + p) * x_ld + 4 * c4, x_bf16);
        float4 d = ldx4(dy, (size_t)(base + p) * dy_ld + 4 * c4, dy_bf16);
        const float hx = (v.x - m.x) * is.x, hy = (v.y - m.y) * is.y, hz = (v.z - m.z) * is.z,
                    hw = (v.w - m.w) * is.w;
        if (!(hx * ga.x + be.x > 0.f)) d.x = 0.f;
        if (!(hy * ga.y + be.y > 0.f)) d.y = 0.f;
        if (!(hz * ga.z + be.z > 0.f)) d.z = 0.f;
        if (!(hw * ga.w + be.w > 0.f)) d.w = 0.f;
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        s2.x += d.x * hx; s2.y += d.y * hy; s2.z += d.z * hz; s2.w += d.w * hw;
    }
    float* prow = part + ((size_t)g * gridDim.x + blockIdx.x) * 2 * C;
    const float4 r1 = plane_reduce4(s1, buf, C4, npl);
    if ((int)threadIdx.x < C4) st4(prow + 4 * c4, r1);
    const float4 r2 = plane_reduce4(s2, buf, C4, npl);
    if ((int)threadIdx.x < C4) st4(prow + C + 4 * c4, r2);
}

// sums[g][2C] (device), dgamma/dbeta
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, int G,
                                                              float* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
    __shared__ double scratch[4];
    const int c = blockIdx.x;
    double tg = 0.0, tb = 0.0;
    for (int g = 0; g < G; ++g) {
        const float* p = part + (size_t)g * nblk * 2 * C;
        double s1 = 0.0, s2 = 0.0;
        for (int b = threadIdx.x; b < nblk; b += 256) {
            s1 += (double)p[(size_t)b * 2 * C + c];
            s2 += (double)p[(size_t)b * 2 * C + C + c];
        }
        s1 = block_sum_double(s1, scratch);
        s2 = block_sum_double(s2, scratch);
        if (threadIdx.x == 0) {
            sums[(size_t)g * 2 * C + c] = (float)s1;
            sums[(size_t)g * 2 * C + C + c] = (float)s2;
        }
        tb += s1;
        tg += s2;
    }
    if (threadIdx.x == 0) {
        dgamma[c] = accumulate ? dgamma[c] + (float)tg : (float)tg;
        dbeta[c] = accumulate ? dbeta[c] + (float)tb : (float)tb;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, int dy_ld, const float* __restrict__ x, int x_ld, int C, int c4_shift,
    unsigned group_items, long group_pix, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ sums,
    int training, float* __restrict__ dx, int dx_ld, int dy_bf16, int x_bf16, int dx_bf16) {
    const int g = blockIdx.y;
    const int c4 = threadIdx.x & ((C >> 2) - 1);
    const float4 m = ld4(mean + g * C + 4 * c4), is = ld4(invstd + g * C + 4 * c4);
    const float4 ga = ld4(gamma + 4 * c4), be = ld4(beta + 4 * c4);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (training) {
        s1 = ld4(sums + (size_t)g * 2 * C + 4 * c4);
        s2 = ld4(sums + (size_t)g * 2 * C + C + 4 * c4);
    }
    const float inv_n = 1.f / (float)group_pix;
    const float mv[4] = {m.x, m.y, m.z, m.w}, iv[4] = {is.x, is.y, is.z, is.w};
    const float gv[4] = {ga.x, ga.y, ga.z, ga.w}, bv[4] = {be.x, be.y, be.z, be.w};
    const float a1[4] = {s1.x, s1.y, s1.z, s1.w}, a2[4] = {s2.x, s2.y, s2.z, s2.w};
    const long base = (long)g * group_pix;
    float4 v[EW_ITEMS], d0[EW_ITEMS];
    long pix[EW_ITEMS];
    bool ok[EW_ITEMS];
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        const unsigned i = (blockIdx.x * EW_ITEMS + k) * 256u + threadIdx.x;
        ok[k] = i < group_items;
        pix[k] = base + (ok[k] ? (i >> c4_shift) : 0u);
        v[k] = ldx4(x, (size_t)pix[k] * x_ld + 4 * c4, x_bf16);
        d0[k] = ldx4(dy, (size_t)pix[k] * dy_ld + 4 * c4, dy_bf16);
    }
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        const float xv[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, dv[4] = {d0[k].x, d0[k].y, d0[k].z, d0[k].w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h = (xv[e] - mv[e]) * iv[e];
            const float d = (h * gv[e] + bv[e] > 0.f) ? dv[e] : 0.f;
            o[e] = training ? gv[e] * iv[e] * (d - a1[e] * inv_n - h * a2[e] * inv_n) : gv[e] * iv[e] * d;
        }
        if (ok[k]) stx4(dx, (size_t)pix[k] * dx_ld + 4 * c4, dx_bf16, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// ---------------------------------------------------------------- slice axpy
template <bool SB>   // the source is stored as bf16 (a template parameter: see warp_fwd_kernel)
__global__ __launch_bounds__(256) void axpy_slice_kernel(float* __restrict__ dst, int dst_ld, int dst_coff,
                                                         const float* __restrict__ src, int src_ld,
                                                         int src_coff, const float* __restrict__ mask,
                                                         int mask_ld, int mask_coff, int C, float alpha,
                                                         int accumulate, long total) {
    constexpr int src_bf16 = SB;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const int C4 = C >> 2;
    const long pix = idiv(gid, C4, total);
    const int c4 = (int)(gid - pix * C4);
    float4 v = ldx4(src, (size_t)pix * src_ld + src_coff + 4 * c4, src_bf16);
    v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
    if (mask) {
        const float4 m = ld4(mask + pix * mask_ld + mask_coff + 4 * c4);
        if (!(m.x > 0.f)) v.x = 0.f;
        if (!(m.y > 0.f)) v.y = 0.f;
        if (!(m.z > 0.f)) v.z = 0.f;
        if (!(m.w > 0.f)) v.w = 0.f;
    }
    float* dp = dst + pix * dst_ld + dst_coff + 4 * c4;
    if (accumulate) {
        const float4 o = ld4(dp);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    st4(dp, v);
}

static bool pow2_c4(int C) {
    const int c4 = C >> 2;
    return C % 4 == 0 && c4 >= 1 && c4 <= 64 && (c4 & (c4 - 1)) == 0;
}
static int blocks_for(long npix, int C);

// BatchNorm backward, first half: the per-group sums  sum(g), sum(g xhat)  (g = dy masked by relu(bn(x)) > 0) in
// sums[G][2C] inside the workspace, and dgamma / dbeta.  Used by nvq_bn_relu_backward and nvq_pw_bn_backward (pw_bwd.hip).
static void launch_bn_bwd_reduce(dim3 grid, hipStream_t s, const float* dy, int dy_ld, const float* x, int x_ld, int C,
                                 long group_pix, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                 int dy_bf16, int x_bf16, float* part) {
#define NVQ_BNR(D_, X_) hipLaunchKernelGGL((bn_bwd_reduce_kernel<D_, X_>), grid, dim3(256), 0, s, dy, dy_ld, x, x_ld, C, group_pix, \
                                           mean, invstd, gamma, beta, part)
    if (dy_bf16) { if (x_bf16) NVQ_BNR(true, true); else NVQ_BNR(true, false); }
    else { if (x_bf16) NVQ_BNR(false, true); else NVQ_BNR(false, false); }
#undef NVQ_BNR
}

int bn_backward_sums(const float* dy, int dy_ld, const float* x, int x_ld, int C, int G, long group_pix, const float* mean,
                     const float* invstd, const float* gamma, const float* beta, float* dgamma, float* dbeta, float* workspace,
                     size_t workspace_bytes, int dy_bf16, int x_bf16, float** sums_out, hipStream_t s) {
    const int nblk = blocks_for(group_pix, C);
    const size_t part_floats = (size_t)G * nblk * 2 * C;
    if ((part_floats + (size_t)G * 2 * C) * sizeof(float) > workspace_bytes) { set_error("bn backward: workspace"); return NVQ_EWORKSPACE; }
    float* sums = workspace + part_floats;
    launch_bn_bwd_reduce(dim3(nblk, G), s, dy, dy_ld, x, x_ld, C, group_pix, mean, invstd, gamma, beta, dy_bf16, x_bf16, workspace);
    int rc = check_launch("bn_bwd_reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, workspace, nblk, C, G, sums, dgamma, dbeta, 0);
    *sums_out = sums;
    return check_launch("bn_bwd_finalize");
}

// part[g][blk][2C] = {sum g, sum g xhat} -> sums[g][2C], dgamma, dbeta (overwritten).  Shared with dw_bwd.hip, whose kernel can
// produce these partials itself.
int bn_bwd_finalize_launch(const float* part, int nblk, int C, int G, float* sums, float* dgamma, float* dbeta, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, part, nblk, C, G, sums, dgamma, dbeta, 0);
    return check_launch("bn_bwd_finalize");
}

// part[g][blk][2C] = {sum, sum of squares} -> mean / invstd per group; the groups are folded into the running statistics in
// `order_host` (identity when null).  Shared by nvq_bn_stats and the fused forward kernel (dwpw_fwd.hip).
int bn_finalize_launch(const float* part, int nblk, int C, int G, long group_pix, float eps, float momentum,
                       const int* order_host, float* mean, float* invstd, float* rmean, float* rvar, hipStream_t s) {
    GroupOrder order;
    for (int i = 0; i < NVQ_MAX_T; ++i) order.g[i] = i < G ? (order_host ? order_host[i] : i) : 0;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, s, part, nblk, C, G, group_pix, eps, momentum, order, mean,
                       invstd, rmean, rvar);
    return check_launch("bn_finalize");
}

static int blocks_for(long npix, int C) {
    const int npl = 256 / (C >> 2);
    int nb = ceil_div(npix, (long)npl * 16);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return nb;
}

}  // namespace nvq

using namespace nvq;

extern "C" {

int nvq_head_forward(const float* frames, int B, int T, int Cin, int H, int W,
                     const int* t_of_slot_host, int nslots, const float* weight, const float* bias,
                     int F, float* out, int out_ld, int out_bf16, float* img8, int math, void* stream) {
    NVQ_REQUIRE(Cin == 3 || Cin == 1, "head_forward: in_channels %d not supported (1 or 3)", Cin);
    NVQ_REQUIRE(pow2_c4(F) && out_ld % 4 == 0 && aligned16(out), "head_forward: F %d (power of two in [4,256]) / ld %d", F, out_ld);
    NVQ_REQUIRE(nslots >= 1 && nslots <= NVQ_MAX_T && T <= NVQ_MAX_T, "head_forward: T %d slots %d", T, nslots);
    SlotMap sm;
    for (int i = 0; i < NVQ_MAX_T; ++i) sm.t[i] = i < nslots ? t_of_slot_host[i] : 0;
    NVQ_REQUIRE(math == NVQ_MATH_F32 || math == NVQ_MATH_BF16, "head_forward: math mode %d", math);
    NVQ_REQUIRE(!out_bf16 || out_ld % 8 == 0, "head_forward: a bf16 output needs ld %% 8 == 0 (%d)", out_ld);
    hipStream_t s = (hipStream_t)stream;
    if (math == NVQ_MATH_BF16 && Cin == 3 && (F == 16 || F == 32 || F == 64)) {
        const int tilesX = (W + HM_TW - 1) / HM_TW, tilesY = (H + HM_TH - 1) / HM_TH;
        const long nwg = (long)tilesX * tilesY * nslots * B;
        NVQ_REQUIRE(nwg < ((long)1 << 31), "head_forward: too many tiles");
        __bf16* i8 = reinterpret_cast<__bf16*>(img8);
#define NVQ_HEAD_MFMA(NB) hipLaunchKernelGGL((head_mfma_kernel<NB>), dim3((unsigned)nwg), dim3(256), 0, s, frames, B, T, H, W, \
                                             sm, weight, bias, out, out_ld, out_bf16, i8, tilesX, tilesY)
        if (F == 16) NVQ_HEAD_MFMA(1);
        else if (F == 32) NVQ_HEAD_MFMA(2);
        else NVQ_HEAD_MFMA(4);
#undef NVQ_HEAD_MFMA
        return check_launch("head_forward(mfma)");
    }
    const int segsX = (W + HEAD_SEG - 1) / HEAD_SEG;
    const long nseg = (long)nslots * B * H * segsX;
    const int npl = 256 / (F / 4);
    int nblk = ceil_div(nseg, npl);
    if (nblk > 2048) nblk = 2048;
    if (Cin == 3)
        hipLaunchKernelGGL((head_fwd_kernel<3>), dim3(nblk), dim3(256), 0, s, frames, B, T, H, W, sm, weight, bias, F, out, out_ld, out_bf16, reinterpret_cast<__bf16*>(img8), nseg, segsX);
    else
        hipLaunchKernelGGL((head_fwd_kernel<1>), dim3(nblk), dim3(256), 0, s, frames, B, T, H, W, sm, weight, bias, F, out, out_ld, out_bf16, reinterpret_cast<__bf16*>(img8), nseg, segsX);
    return check_launch("head_forward");
}

int nvq_head_wgrad(const float* frames, int B, int T, int Cin, int H, int W, const int* t_of_slot_host,
                   int nslots, const float* dout, int dout_ld, const float* dout2, int dout2_ld, const float* act,
                   int act_ld, int F, float* dweight, float* dbias, float* workspace, size_t workspace_bytes,
                   int accumulate, int act_bf16, void* stream) {
    NVQ_REQUIRE(Cin == 3 || Cin == 1, "head_wgrad: in_channels %d not supported (1 or 3)", Cin);
    NVQ_REQUIRE(pow2_c4(F), "head_wgrad: F %d must be a power of two in [4,256]", F);
    NVQ_REQUIRE(dout_ld % 4 == 0 && act_ld % 4 == 0 && (!dout2 || dout2_ld % 4 == 0), "head_wgrad: ld");
    SlotMap sm;
    for (int i = 0; i < NVQ_MAX_T; ++i) sm.t[i] = i < nslots ? t_of_slot_host[i] : 0;
    const int K = Cin * 9;
    const int segsX = (W + HEAD_SEG - 1) / HEAD_SEG;
    const long nseg = (long)nslots * B * H * segsX;
    int nblk = ceil_div(nseg, 256 / (F / 4));
    if (nblk > 1024) nblk = 1024;       // 4 workgroups per CU: the 112 accumulators per thread need the latency cover
    const size_t row = (size_t)F * K + F;
    if (row * nblk * sizeof(float) > workspace_bytes) { set_error("head_wgrad: workspace"); return NVQ_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
#define NVQ_LAUNCH_HW(CIN, AB) \
    hipLaunchKernelGGL((head_wgrad_kernel<CIN, AB>), dim3(nblk), dim3(256), 0, s, frames, B, T, H, W, sm, dout, dout_ld, dout2, dout2_ld, act, act_ld, F, nseg, segsX, workspace)
    if (Cin == 3) { if (act_bf16) NVQ_LAUNCH_HW(3, 1); else NVQ_LAUNCH_HW(3, 0); }
    else { if (act_bf16) NVQ_LAUNCH_HW(1, 1); else NVQ_LAUNCH_HW(1, 0); }
#undef NVQ_LAUNCH_HW
    int rc = check_launch("head_wgrad");
    if (rc) return rc;
    // rows are [F*K weights | F biases]; reduce the two pieces separately (row stride = row)
    // by viewing the partials as nblk rows of `row` floats.
    // launch_reduce_partials expects dense rows, so reduce the whole row into a scratch tail.
    float* scratch = workspace + row * nblk;
    if ((row * (nblk + 1)) * sizeof(float) > workspace_bytes) { set_error("head_wgrad: workspace"); return NVQ_EWORKSPACE; }
    rc = launch_reduce_partials(workspace, nblk, (int)row, 1.f, scratch, 0, s);
    if (rc) return rc;
    rc = nvq_axpy_slice(dweight, F * K, 0, scratch, F * K, 0, nullptr, 0, 0, F * K, 1, 1.f, accumulate, 0, stream);
    if (rc) return rc;
    return nvq_axpy_slice(dbias, F, 0, scratch + F * K, F, 0, nullptr, 0, 0, F, 1, 1.f, accumulate, 0, stream);
}

static DwBn make_dwbn(const nvq_bn_input* b, int C) {
    DwBn r = {nullptr, nullptr, nullptr, nullptr, 1, C};
    if (b) { r.mean = b->mean; r.invstd = b->invstd; r.gamma = b->gamma; r.beta = b->beta; r.group_images = b->group_images; }
    return r;
}

int nvq_dwconv_forward(const float* in, int in_ld, const float* weight, int C, float* out, int out_ld,
                       int N, int H, int W, int flip, int in_bf16, int out_bf16, const nvq_bn_input* bn,
                       const nvq_dw_epilogue* epi, void* stream) {
    NVQ_REQUIRE(!epi || (C % DB_C == 0 && in_bf16 && (!epi->add || epi->add_ld % 4 == 0) &&
                         (!epi->mask || epi->mask_ld % 4 == 0)),
                "dwconv_forward: the add / mask epilogue needs a bf16 input with C %% 64 == 0");
    DwEpi ep = {nullptr, 0, nullptr, 0, 0, 0};
    if (epi) { ep.add = epi->add; ep.add_ld = epi->add_ld; ep.mask = epi->mask; ep.mask_ld = epi->mask_ld; ep.mask_bf16 = epi->mask_bf16; ep.add_bf16 = epi->add_bf16; }
    NVQ_REQUIRE(!bn || (C % DB_C == 0 && in_bf16 && bn->group_images > 0 && N % bn->group_images == 0),
                "dwconv_forward: the fused BatchNorm input needs a bf16 tensor with C %% 64 == 0");
    NVQ_REQUIRE(C % 4 == 0 && C <= 1024 && in_ld % 4 == 0 && out_ld % 4 == 0 && aligned16(in) && aligned16(out),
                "dwconv_forward: C %d ld %d/%d", C, in_ld, out_ld);
    if (C % DB_C == 0 && in_bf16) {
        NVQ_REQUIRE(in_ld % 8 == 0 && out_ld % 8 == 0, "dwconv_forward(bf16): ld %d/%d must be multiples of 8", in_ld, out_ld);
        const int tilesX = (W + DT_W - 1) / DT_W, tilesY = (H + DT_H - 1) / DT_H;
        const int ntiles = tilesX * tilesY * N;
        int nwg = ntiles < DB_MAXWG ? ntiles : DB_MAXWG;
        if (nwg >= 8) nwg &= ~7;                              // multiple of the XCD count, see xcd_tile()
#define NVQ_DWB(B_, E_)                                                                                                 \
    hipLaunchKernelGGL((dwconv_bf16_kernel<B_, E_>), dim3(nwg, C / DB_C), dim3(DB_T), 0, (hipStream_t)stream,               \
                       reinterpret_cast<const __bf16*>(in), in_ld, weight, C, out, out_ld, H, W, tilesX, tilesY, ntiles, flip, \
                       out_bf16, make_dwbn(bn, C), ep)
        const bool epb = epi && epi->add && epi->mask && epi->add_bf16 && epi->mask_bf16 && epi->add_ld % 4 == 0 &&
                         epi->mask_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(epi->add) & 7) == 0 &&
                         (reinterpret_cast<uintptr_t>(epi->mask) & 7) == 0;
        if (bn && epb) NVQ_DWB(true, 2);
        else if (bn && epi) NVQ_DWB(true, 1);
        else if (bn) NVQ_DWB(true, 0);
        else if (epb) NVQ_DWB(false, 2);
        else if (epi) NVQ_DWB(false, 1);
        else NVQ_DWB(false, 0);
#undef NVQ_DWB
        return check_launch("dwconv_bf16");
    }
    if (C % DT_C == 0) {
        const int tilesX = (W + DT_W - 1) / DT_W, tilesY = (H + DT_H - 1) / DT_H;
        hipLaunchKernelGGL(dwconv_tiled_kernel, dim3((unsigned)((long)tilesX * tilesY * N), C / DT_C), dim3(256), 0,
                           (hipStream_t)stream, in, in_ld, weight, C, out, out_ld, H, W, tilesX, tilesY, flip, in_bf16,
                           out_bf16);
        return check_launch("dwconv_tiled");
    }
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(dwconv_kernel, dim3(ceil_div(total, 256)), dim3(256), (size_t)9 * C * sizeof(float),
                       (hipStream_t)stream, in, in_ld, weight, C, out, out_ld, H, W, flip, in_bf16, out_bf16, total);
    return check_launch("dwconv_forward");
}

int nvq_dwconv_wgrad(const float* x, int x_ld, const float* dy, int dy_ld, int C, int N, int H, int W,
                     float* dweight, float* workspace, size_t workspace_bytes, int accumulate,
                     int x_bf16, int dy_bf16, const nvq_bn_input* bn, void* stream) {
    NVQ_REQUIRE(!bn || (C % DB_C == 0 && x_bf16 && dy_bf16 && x_ld % 8 == 0 && dy_ld % 8 == 0 && bn->group_images > 0 &&
                        N % bn->group_images == 0),
                "dwconv_wgrad: the fused BatchNorm input needs bf16 tensors with C %% 64 == 0");
    NVQ_REQUIRE(pow2_c4(C), "dwconv_wgrad: C %d must be a power of two in [4,256]", C);
    NVQ_REQUIRE(x_ld % 4 == 0 && dy_ld % 4 == 0, "dwconv_wgrad: ld");
    const long npix = (long)N * H * W;
    if (C % DB_C == 0 && x_bf16 && dy_bf16 && x_ld % 8 == 0 && dy_ld % 8 == 0) {
        const int tilesX = (W + DT_W - 1) / DT_W, tilesY = (H + DT_H - 1) / DT_H;
        const int ntiles = tilesX * tilesY * N;
        const int nb = ntiles < 512 ? ntiles : 512;
        if ((size_t)nb * C * 9 * sizeof(float) > workspace_bytes) { set_error("dwconv_wgrad: workspace"); return NVQ_EWORKSPACE; }
        hipLaunchKernelGGL(dwconv_wgrad_bf16_kernel, dim3(nb, C / DB_C), dim3(DB_T), 0, (hipStream_t)stream,
                           reinterpret_cast<const __bf16*>(x), x_ld, reinterpret_cast<const __bf16*>(dy), dy_ld, C, H, W,
                           tilesX, tilesY, ntiles, workspace, make_dwbn(bn, C));
        int rc0 = check_launch("dwconv_wgrad_bf16");
        if (rc0) return rc0;
        return launch_reduce_partials(workspace, nb, C * 9, 1.f, dweight, accumulate, (hipStream_t)stream);
    }
    if (C % DT_C == 0) {
        const int tilesX = (W + DT_W - 1) / DT_W, tilesY = (H + DT_H - 1) / DT_H;
        const int ntiles = tilesX * tilesY * N;
        const int nb = ntiles < 512 ? ntiles : 512;
        if ((size_t)nb * C * 9 * sizeof(float) > workspace_bytes) { set_error("dwconv_wgrad: workspace"); return NVQ_EWORKSPACE; }
        hipLaunchKernelGGL(dwconv_wgrad_tiled_kernel, dim3(nb, C / DT_C), dim3(256), 0, (hipStream_t)stream, x, x_ld, dy,
                           dy_ld, C, H, W, tilesX, tilesY, ntiles, x_bf16, dy_bf16, workspace);
        int rc0 = check_launch("dwconv_wgrad_tiled");
        if (rc0) return rc0;
        return launch_reduce_partials(workspace, nb, C * 9, 1.f, dweight, accumulate, (hipStream_t)stream);
    }
    const int nblk = blocks_for(npix, C);
    if ((size_t)nblk * C * 9 * sizeof(float) > workspace_bytes) { set_error("dwconv_wgrad: workspace"); return NVQ_EWORKSPACE; }
    hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, x_ld, dy, dy_ld, C, H, W, npix, x_bf16, dy_bf16, workspace);
    int rc = check_launch("dwconv_wgrad");
    if (rc) return rc;
    return launch_reduce_partials(workspace, nblk, C * 9, 1.f, dweight, accumulate, (hipStream_t)stream);
}

int nvq_bn_stats(const float* x, int x_ld, int C, int N, int group_images, int H, int W, float eps,
                 float momentum, const int* order_host, float* mean, float* invstd, float* running_mean,
                 float* running_var, float* workspace, size_t workspace_bytes, int x_bf16, void* stream) {
    NVQ_REQUIRE(pow2_c4(C), "bn_stats: C %d must be a power of two in [4,256]", C);
    NVQ_REQUIRE(group_images > 0 && N % group_images == 0 && N / group_images <= NVQ_MAX_T, "bn_stats: groups");
    NVQ_REQUIRE(x_ld % 4 == 0 && aligned16(x), "bn_stats: ld");
    const int G = N / group_images;
    const long group_pix = (long)group_images * H * W;
    const int nblk = blocks_for(group_pix, C);
    if ((size_t)G * nblk * 2 * C * sizeof(float) > workspace_bytes) { set_error("bn_stats: workspace"); return NVQ_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk, G), dim3(256), 0, s, x, x_ld, C, group_pix, x_bf16, workspace);
    int rc = check_launch("bn_stats");
    if (rc) return rc;
    return bn_finalize_launch(workspace, nblk, C, G, group_pix, eps, momentum, order_host, mean, invstd, running_mean,
                              running_var, s);
}

int nvq_bn_eval_stats(const float* running_mean, const float* running_var, int C, int G, float eps,
                      float* mean, float* invstd, void* stream) {
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(ceil_div((long)C * G, 256)), dim3(256), 0, (hipStream_t)stream,
                       running_mean, running_var, C, G, eps, mean, invstd);
    return check_launch("bn_eval_stats");
}

int nvq_bn_apply_relu(const float* x, int x_ld, int C, int N, int group_images, int H, int W,
                      const float* mean, const float* invstd, const float* gamma, const float* beta,
                      const float* res, int res_ld, float* outA, int outA_ld, int outA_coff,
                      int split_images, float* outB, int outB_ld, int outB_coff, int x_bf16, int out_bf16,
                      int res_bf16, void* stream) {
    NVQ_REQUIRE(C % 4 == 0 && x_ld % 4 == 0 && outA_ld % 4 == 0 && outA_coff % 4 == 0 && outB_ld % 4 == 0 &&
                    outB_coff % 4 == 0 && (!res || res_ld % 4 == 0),
                "bn_apply_relu: alignment");
    NVQ_REQUIRE(split_images >= N || outB, "bn_apply_relu: outB missing");
    NVQ_REQUIRE(pow2_c4(C), "bn_apply_relu: C %d must be a power of two in [4,256]", C);
    NVQ_REQUIRE(group_images > 0 && N % group_images == 0, "bn_apply_relu: groups");
    const int G = N / group_images;
    const long group_pix = (long)group_images * H * W;
    NVQ_REQUIRE(group_pix * (C / 4) < ((long)1 << 32) - 256 * EW_ITEMS, "bn_apply_relu: group of %ld pixels too large", group_pix);
    int shift = 0;
    while ((1 << shift) < C / 4) ++shift;
    const unsigned items = (unsigned)(group_pix * (C / 4));
#define NVQ_BNA(A_)                                                                                                            \
    hipLaunchKernelGGL(bn_apply_relu_kernel<A_>, dim3(ceil_div((long)items, 256 * EW_ITEMS), G), dim3(256), 0, (hipStream_t)stream, \
                       x, x_ld, C, shift, items, group_pix, mean, invstd, gamma, beta, res, res_ld, outA, outA_ld, outA_coff,        \
                       (long)(split_images < N ? split_images : N) * H * W, outB, outB_ld, outB_coff, x_bf16, out_bf16, res_bf16)
    if (x_bf16 && out_bf16 && (res_bf16 || !res)) NVQ_BNA(true); else NVQ_BNA(false);
#undef NVQ_BNA
    return check_launch("bn_apply_relu");
}

int nvq_bn_relu_backward(const float* dy, int dy_ld, const float* x, int x_ld, int C, int N,
                         int group_images, int H, int W, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, int training, float* dx, int dx_ld,
                         float* dgamma, float* dbeta, float* workspace, size_t workspace_bytes,
                         int accumulate, int dy_bf16, int x_bf16, int dx_bf16, void* stream) {
    NVQ_REQUIRE(pow2_c4(C), "bn_relu_backward: C %d must be a power of two in [4,256]", C);
    NVQ_REQUIRE(dy_ld % 4 == 0 && x_ld % 4 == 0 && dx_ld % 4 == 0, "bn_relu_backward: ld");
    NVQ_REQUIRE(group_images > 0 && N % group_images == 0 && N / group_images <= NVQ_MAX_T, "bn_relu_backward: groups");
    const int G = N / group_images;
    const long group_pix = (long)group_images * H * W;
    const int nblk = blocks_for(group_pix, C);
    const size_t part_floats = (size_t)G * nblk * 2 * C;
    if ((part_floats + (size_t)G * 2 * C) * sizeof(float) > workspace_bytes) { set_error("bn_relu_backward: workspace"); return NVQ_EWORKSPACE; }
    float* sums = workspace + part_floats;
    hipStream_t s = (hipStream_t)stream;
    launch_bn_bwd_reduce(dim3(nblk, G), s, dy, dy_ld, x, x_ld, C, group_pix, mean, invstd, gamma, beta, dy_bf16, x_bf16, workspace);
    int rc = check_launch("bn_bwd_reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, workspace, nblk, C, G, sums,
                       dgamma, dbeta, accumulate);
    rc = check_launch("bn_bwd_finalize");
    if (rc) return rc;
    NVQ_REQUIRE(group_pix * (C / 4) < ((long)1 << 32) - 256 * EW_ITEMS, "bn_relu_backward: group of %ld pixels too large", group_pix);
    int shift = 0;
    while ((1 << shift) < C / 4) ++shift;
    const unsigned items = (unsigned)(group_pix * (C / 4));
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ceil_div((long)items, 256 * EW_ITEMS), G), dim3(256), 0, s, dy, dy_ld, x,
                       x_ld, C, shift, items, group_pix, mean, invstd, gamma, beta, sums, training, dx, dx_ld, dy_bf16,
                       x_bf16, dx_bf16);
    return check_launch("bn_bwd_apply");
}

int nvq_axpy_slice(float* dst, int dst_ld, int dst_coff, const float* src, int src_ld, int src_coff,
                   const float* mask, int mask_ld, int mask_coff, int C, long npix, float alpha,
                   int accumulate, int src_bf16, void* stream) {
    NVQ_REQUIRE(C % 4 == 0 && dst_ld % 4 == 0 && dst_coff % 4 == 0 && src_ld % 4 == 0 && src_coff % 4 == 0 &&
                    (!mask || (mask_ld % 4 == 0 && mask_coff % 4 == 0)) && aligned16(dst) && aligned16(src),
                "axpy_slice: alignment (C %d ld %d/%d)", C, dst_ld, src_ld);
    const long total = npix * (C / 4);
    if (total == 0) return NVQ_OK;
    if (src_bf16)
        hipLaunchKernelGGL(axpy_slice_kernel<true>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, dst, dst_ld,
                           dst_coff, src, src_ld, src_coff, mask, mask_ld, mask_coff, C, alpha, accumulate, total);
    else
        hipLaunchKernelGGL(axpy_slice_kernel<false>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, dst, dst_ld,
                           dst_coff, src, src_ld, src_coff, mask, mask_ld, mask_coff, C, alpha, accumulate, total);
    return check_launch("axpy_slice");
}

}  // extern "C"
