// Motion estimation stages: 9x9 local correlation (forward + both gradients) and the
// bilinear flow warp (forward + scatter/flow gradients).  fp32 NHWC.
#include "common.h"

namespace nvq {

constexpr int CT_H = 8, CT_W = 32;    // pixel tile, one thread per pixel
constexpr int CD = 4;                 // max displacement
constexpr int CN = 2 * CD + 1;        // 9
constexpr int ND = CN * CN;           // 81
constexpr int CHW = CT_W + 2 * CD;    // 40
constexpr int CHH = CT_H + 2 * CD;    // 16
constexpr int CCH = 16;               // channels per staged chunk
constexpr int CLD = 20;               // floats per staged pixel (pad: 16 consecutive pixels -> 16 distinct 16-B slots)

// Stage the (CHH x CHW) halo tile of `src` image `n` for channels [ch0, ch0+16) into LDS.
__device__ __forceinline__ void stage_halo(const float* __restrict__ src, int ld, int C, int n, int H, int W,
                                           int ty0, int tx0, int ch0, float* xs) {
    for (int item = threadIdx.x; item < CHH * CHW * 4; item += 256) {
        const int hp = item >> 2, q = item & 3;
        const int hy = hp / CHW, hx = hp - hy * CHW;
        const int gy = ty0 + hy - CD, gx = tx0 + hx - CD;
        const int ch = ch0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < H && gx >= 0 && gx < W && ch < C)
            v = ld4(src + ((size_t)(n * H + gy) * W + gx) * ld + ch);
        st4(xs + hp * CLD + 4 * q, v);
    }
}

// out[n,p,d] = (1/C) sum_c x1[n,p,c] * x2[n % x2_images, p + off(d), c]
__global__ __launch_bounds__(256) void corr_fwd_kernel(const float* __restrict__ x1, int x1_ld,
                                                       const float* __restrict__ x2, int x2_ld,
                                                       int x2_images, int C, int H, int W, int tilesX,
                                                       int tilesY, float* __restrict__ out, int out_ld) {
    __shared__ __attribute__((aligned(16))) float xs[CHH * CHW * CLD];
    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int py = threadIdx.x / CT_W, px = threadIdx.x % CT_W;
    const int gy = ty * CT_H + py, gx = tx * CT_W + px;
    const bool inside = gy < H && gx < W;
    const size_t pix = (size_t)(n * H + (inside ? gy : 0)) * W + (inside ? gx : 0);
    float acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = 0.f;
    for (int ch0 = 0; ch0 < C; ch0 += CCH) {
        __syncthreads();
        stage_halo(x2, x2_ld, C, n % x2_images, H, W, ty * CT_H, tx * CT_W, ch0, xs);
        __syncthreads();
        float4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (inside && ch0 + 4 * q < C) a[q] = ld4(x1 + pix * x1_ld + ch0 + 4 * q);
        }
#pragma unroll
        for (int i = 0; i < CN; ++i)
#pragma unroll
            for (int j = 0; j < CN; ++j) {
                const float* p = xs + ((py + i) * CHW + px + j) * CLD;
                float s = acc[i * CN + j];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = ld4(p + 4 * q);
                    s += a[q].x * b.x; s += a[q].y * b.y; s += a[q].z * b.z; s += a[q].w * b.w;
                }
                acc[i * CN + j] = s;
            }
    }
    if (!inside) return;
    const float inv = 1.f / (float)C;
    float* op = out + pix * out_ld;
#pragma unroll
    for (int k = 0; k < ND / 4; ++k)
        st4(op + 4 * k, make_float4(acc[4 * k] * inv, acc[4 * k + 1] * inv, acc[4 * k + 2] * inv, acc[4 * k + 3] * inv));
    st4(op + 80, make_float4(acc[80] * inv, 0.f, 0.f, 0.f));
    for (int k = 84; k < out_ld; k += 4) st4(op + k, make_float4(0.f, 0.f, 0.f, 0.f));
}

// WHICH == 1: dx[n,p,c] (+)= (1/C) sum_d dcorr[n,p,d]        * other[n % oi, p + off(d), c]
// WHICH == 2: dx[n,q,c] (+)= (1/C) sum_d dcorr[n,q-off(d),d] * other[n,      q - off(d), c]
// Structure (each point was a measured stall):
//  * the displacement rows are a runtime loop with 9 weights live at a time: a fully unrolled 81-tap body made
//    the compiler hoist every LDS read and spill ~900 VGPRs;
//  * the next row's 9 weights are loaded while the current row is computed (a row that starts by loading its
//    own weights waits a full memory latency 36 times per tile);
//  * the next channel chunk's halo pieces are fetched into registers during the compute of the current chunk.
constexpr int CH_ITEMS = CHH * CHW * 4;                 // float4 pieces of one 16-channel halo chunk
constexpr int CH_PER = (CH_ITEMS + 255) / 256;

template <int WHICH>
__global__ __launch_bounds__(256) void corr_bwd_kernel(const float* __restrict__ dcorr, int dcorr_ld,
                                                       const float* __restrict__ other, int other_ld,
                                                       int other_images, int C, int H, int W, int tilesX,
                                                       int tilesY, float* __restrict__ dx, int dx_ld,
                                                       int dx_coff, int accumulate) {
    __shared__ __attribute__((aligned(16))) float xs[CHH * CHW * CLD];
    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int tid = threadIdx.x;
    const int py = tid / CT_W, px = tid % CT_W;
    const int gy = ty * CT_H + py, gx = tx * CT_W + px;
    const bool inside = gy < H && gx < W;
    const size_t pix = (size_t)(n * H + (inside ? gy : 0)) * W + (inside ? gx : 0);
    const float inv = 1.f / (float)C;
    const int on = n % other_images;

    // halo staging: per-thread piece offsets (chunk independent); invalid pieces read element 0, masked at commit
    unsigned hoff[CH_PER];
    unsigned hmask = 0;
#pragma unroll
    for (int k = 0; k < CH_PER; ++k) {
        const int item = tid + k * 256;
        const int hp = item >> 2;
        const int hy = hp / CHW, hx = hp - hy * CHW;
        const int yy = ty * CT_H + hy - CD, xx = tx * CT_W + hx - CD;
        const bool ok = item < CH_ITEMS && yy >= 0 && yy < H && xx >= 0 && xx < W;
        hmask |= (ok ? 1u : 0u) << k;
        hoff[k] = ok ? (unsigned)(((size_t)(on * H + yy) * W + xx) * other_ld + 4 * (item & 3)) : 0u;
    }
    const int cq = 4 * (tid & 3);
    float4 hr[CH_PER];
    bool hch = false;
    auto fetch = [&](int ch0) {
        hch = ch0 + cq < C;
        const int o = hch ? ch0 : -cq;
#pragma unroll
        for (int k = 0; k < CH_PER; ++k) hr[k] = ld4(other + hoff[k] + ((hmask >> k) & 1 ? o : 0));
    };
    auto commit = [&]() {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < CH_PER; ++k) {
            const int item = tid + k * 256;
            if (item < CH_ITEMS) st4(xs + (item >> 2) * CLD + 4 * (item & 3), ((hmask >> k) & 1) && hch ? hr[k] : z);
        }
    };
    // weights of displacement row i for this thread's pixel (addresses clamped, masked when used)
    auto load_row = [&](int i, float (&w)[CN], unsigned& wm) {
        wm = 0;
        if (WHICH == 1) {
#pragma unroll
            for (int j = 0; j < CN; ++j) w[j] = dcorr[pix * dcorr_ld + i * CN + j];
            wm = inside ? 0x1ffu : 0u;
        } else {
            const int qy = gy - (i - CD);
#pragma unroll
            for (int j = 0; j < CN; ++j) {
                const int qx = gx - (j - CD);
                const bool ok = inside && qy >= 0 && qy < H && qx >= 0 && qx < W;
                wm |= (ok ? 1u : 0u) << j;
                w[j] = dcorr[(ok ? ((size_t)(n * H + qy) * W + qx) * dcorr_ld : 0) + i * CN + j];
            }
        }
    };

    fetch(0);
    for (int ch0 = 0; ch0 < C; ch0 += CCH) {
        __syncthreads();
        commit();
        __syncthreads();
        if (ch0 + CCH < C) fetch(ch0 + CCH);
        float4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        float wcur[CN], wnext[CN];
        unsigned mcur = 0, mnext = 0;
        load_row(0, wcur, mcur);
#pragma unroll 1
        for (int i = 0; i < CN; ++i) {
            if (i + 1 < CN) load_row(i + 1, wnext, mnext);
            const int hy = WHICH == 1 ? py + i : py + 2 * CD - i;
#pragma unroll
            for (int j = 0; j < CN; ++j) {
                const int hx = WHICH == 1 ? px + j : px + 2 * CD - j;
                const float* p = xs + (hy * CHW + hx) * CLD;
                const float w = (mcur >> j) & 1 ? wcur[j] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = ld4(p + 4 * q);
                    acc[q].x += w * b.x; acc[q].y += w * b.y; acc[q].z += w * b.z; acc[q].w += w * b.w;
                }
            }
#pragma unroll
            for (int j = 0; j < CN; ++j) wcur[j] = wnext[j];
            mcur = mnext;
        }
        if (inside) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ch0 + 4 * q >= C) continue;
                float* op = dx + pix * dx_ld + dx_coff + ch0 + 4 * q;
                float4 v = make_float4(acc[q].x * inv, acc[q].y * inv, acc[q].z * inv, acc[q].w * inv);
                if (accumulate) {
                    const float4 o = ld4(op);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                st4(op, v);
            }
        }
    }
}

// ---------------------------------------------------------------- bilinear warp
struct WarpGeom {
    float ix, iy;
    int x0, y0;           // north-west corner
    float wnw, wne, wsw, wse;
    bool vnw, vne, vsw, vse;
};

// Mirrors the reference's arithmetic: grid = 2*(x+flow)/(W-1) - 1 (super_resolution.py:129-133),
// then grid_sample's align_corners=True un-normalisation ((g+1)/2)*(W-1).
__device__ __forceinline__ WarpGeom warp_geom(float fx, float fy, int x, int y, int H, int W) {
    WarpGeom g;
    const float gxn = 2.0f * ((float)x + fx) / (float)(W - 1) - 1.0f;
    const float gyn = 2.0f * ((float)y + fy) / (float)(H - 1) - 1.0f;
    g.ix = ((gxn + 1.f) / 2.f) * (float)(W - 1);
    g.iy = ((gyn + 1.f) / 2.f) * (float)(H - 1);
    const float fx0 = floorf(g.ix), fy0 = floorf(g.iy);
    // guard the int conversion against wild flows
    g.x0 = (int)fminf(fmaxf(fx0, -2.f), (float)W + 1.f);
    g.y0 = (int)fminf(fmaxf(fy0, -2.f), (float)H + 1.f);
    const float x_se = fx0 + 1.f, y_se = fy0 + 1.f;
    g.wnw = (x_se - g.ix) * (y_se - g.iy);
    g.wne = (g.ix - fx0) * (y_se - g.iy);
    g.wsw = (x_se - g.ix) * (g.iy - fy0);
    g.wse = (g.ix - fx0) * (g.iy - fy0);
    const bool finite_ok = (fx0 > -2.f) && (fx0 < (float)W + 1.f) && (fy0 > -2.f) && (fy0 < (float)H + 1.f);
    const bool xl = g.x0 >= 0 && g.x0 < W, xr = g.x0 + 1 >= 0 && g.x0 + 1 < W;
    const bool yt = g.y0 >= 0 && g.y0 < H, yb = g.y0 + 1 >= 0 && g.y0 + 1 < H;
    g.vnw = finite_ok && xl && yt;
    g.vne = finite_ok && xr && yt;
    g.vsw = finite_ok && xl && yb;
    g.vse = finite_ok && xr && yb;
    return g;
}

// A workgroup takes 256 consecutive pixels.  Phase A: one THREAD per pixel does the geometry (two fp32 divisions, floors,
// validity) once and leaves {4 corner pixel indices, 4 weights} in LDS; phase B: 16 lanes per pixel (8 bytes of channels each)
// read them back as two broadcast b128 reads and do the loads.  With the geometry in every one of a pixel's 16 lanes the
// kernel was VALU-bound (~200 instructions per lane, 0.53 ms for 8 x 540 x 960 x 64 = 2.0 TB/s; more pixels per thread made it
// slower, not faster).  Corner loads are unconditional (a corner outside the image reads the image's first pixel with weight 0:
// fma(v, 0, acc) == acc, so the sum and its order are those of the reference).
// The storage types are template parameters: as run-time flags every ldx4 was a branch with its own vmcnt(0) behind the load -
// the four corner loads of a pixel ran one after the other (2.0 -> 3.6 TB/s with the geometry hoisted as well).
constexpr int WT_H = 8, WT_W = 32, WP = WT_H * WT_W;
template <bool FB, bool OB>
__global__ __launch_bounds__(256) void warp_fwd_kernel(const float* __restrict__ feat, int feat_ld,
                                                       const float* __restrict__ flow, int flow_ld, int C,
                                                       int H, int W, float* __restrict__ out, int out_ld,
                                                       int out_coff, int tilesX, int tilesY) {
    constexpr int feat_bf16 = FB, out_bf16 = OB;
    __shared__ int4 so[WP];
    __shared__ float4 sw[WP];
    // 8 x 32-pixel tiles in XCD order: the two feature rows a pixel row reads are shared with the rows above and below inside
    // the tile (L1) and with the neighbouring tiles of the same XCD (L2) - as 256 consecutive pixels of one row every
    // feature row came in twice
    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int img = n * H * W;                                // first pixel of this image (N*H*W < 2^31: host check)
    {
        const int x = tx * WT_W + (threadIdx.x & (WT_W - 1)), y = ty * WT_H + (threadIdx.x >> 5);
        int4 o = make_int4(0, 0, 0, 0);
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x < W && y < H) {
            const long pix = img + (long)y * W + x;
            const WarpGeom g = warp_geom(flow[pix * flow_ld], flow[pix * flow_ld + 1], x, y, H, W);
            o = make_int4(img + (g.vnw ? g.y0 * W + g.x0 : 0), img + (g.vne ? g.y0 * W + g.x0 + 1 : 0),
                          img + (g.vsw ? (g.y0 + 1) * W + g.x0 : 0), img + (g.vse ? (g.y0 + 1) * W + g.x0 + 1 : 0));
            w = make_float4(g.vnw ? g.wnw : 0.f, g.vne ? g.wne : 0.f, g.vsw ? g.wsw : 0.f, g.vse ? g.wse : 0.f);
        }
        so[threadIdx.x] = o;
        sw[threadIdx.x] = w;
    }
    __syncthreads();
    const int c4 = threadIdx.x & 15;
#pragma unroll 4
    for (int it = 0; it < WP / 16; ++it) {
        const int pl = it * 16 + (threadIdx.x >> 4);
        const int x = tx * WT_W + (pl & (WT_W - 1)), y = ty * WT_H + (pl >> 5);
        if (x >= W || y >= H) continue;
        const long pix = img + (long)y * W + x;
        const int4 o = so[pl];
        const float4 w = sw[pl];
        for (int ch = 4 * c4; ch < C; ch += 64) {
            const float4 v0 = ldx4(feat, (size_t)o.x * feat_ld + ch, feat_bf16);
            const float4 v1 = ldx4(feat, (size_t)o.y * feat_ld + ch, feat_bf16);
            const float4 v2 = ldx4(feat, (size_t)o.z * feat_ld + ch, feat_bf16);
            const float4 v3 = ldx4(feat, (size_t)o.w * feat_ld + ch, feat_bf16);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            acc.x += v0.x * w.x; acc.y += v0.y * w.x; acc.z += v0.z * w.x; acc.w += v0.w * w.x;
            acc.x += v1.x * w.y; acc.y += v1.y * w.y; acc.z += v1.z * w.y; acc.w += v1.w * w.y;
            acc.x += v2.x * w.z; acc.y += v2.y * w.z; acc.z += v2.z * w.z; acc.w += v2.w * w.z;
            acc.x += v3.x * w.w; acc.y += v3.y * w.w; acc.z += v3.z * w.w; acc.w += v3.w * w.w;
            stx4(out, (size_t)pix * out_ld + out_coff + ch, out_bf16, acc);
        }
    }
}

// One thread per (pixel, channel): the lanes of a pixel issue their float atomics on C consecutive floats
// (256 contiguous bytes for C = 64), the access shape global float atomics run at full rate for.
// L = min(C, 64) lanes per pixel (a power of two, so the lanes of one pixel sit in one wave); for C > 64
// every lane walks channels ch, ch + 64, ...
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ dout, int dout_ld, int dout_coff,
                                                       const float* __restrict__ feat, int feat_ld,
                                                       const float* __restrict__ flow, int flow_ld, int C,
                                                       int H, int W, float* __restrict__ dfeat, int dfeat_ld,
                                                       float* __restrict__ dflow, int dflow_ld, long total) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const int L = C < 64 ? C : 64;
    const long pix = idiv(gid, L, total);
    const int ch0 = (int)(gid - pix * L);
    const long rowi = idiv(pix, W, total);
    const int x = (int)(pix - rowi * W);
    const int y = (int)(rowi - idiv(rowi, H, total) * H);
    const long img = pix - ((long)y * W + x);
    const WarpGeom g = warp_geom(flow[pix * flow_ld], flow[pix * flow_ld + 1], x, y, H, W);
    const float fx0 = floorf(g.ix), fy0 = floorf(g.iy);
    const float x_se = fx0 + 1.f, y_se = fy0 + 1.f;
    float gix = 0.f, giy = 0.f;
    for (int ch = ch0; ch < C; ch += L) {
    const float go = dout[pix * dout_ld + dout_coff + ch];
    const float* fb = feat + ch;
    float* db = dfeat + ch;
    if (g.vnw) { const long o = (img + (long)g.y0 * W + g.x0);
        atomicAdd(db + o * dfeat_ld, go * g.wnw);
        const float d = fb[o * feat_ld] * go; gix -= d * (y_se - g.iy); giy -= d * (x_se - g.ix); }
    if (g.vne) { const long o = (img + (long)g.y0 * W + g.x0 + 1);
        atomicAdd(db + o * dfeat_ld, go * g.wne);
        const float d = fb[o * feat_ld] * go; gix += d * (y_se - g.iy); giy -= d * (g.ix - fx0); }
    if (g.vsw) { const long o = (img + (long)(g.y0 + 1) * W + g.x0);
        atomicAdd(db + o * dfeat_ld, go * g.wsw);
        const float d = fb[o * feat_ld] * go; gix -= d * (g.iy - fy0); giy += d * (x_se - g.ix); }
    if (g.vse) { const long o = (img + (long)(g.y0 + 1) * W + g.x0 + 1);
        atomicAdd(db + o * dfeat_ld, go * g.wse);
        const float d = fb[o * feat_ld] * go; gix += d * (g.iy - fy0); giy += d * (g.ix - fx0); }
    }
    gix = group_sum(gix, L);
    giy = group_sum(giy, L);
    if (ch0 == 0) {
        // grid_sample multiplies by (size-1)/2; the normalisation's autograd divides by (size-1) and doubles
        const float gx = 2.0f * ((gix * ((float)(W - 1) / 2.f)) / (float)(W - 1));
        const float gyv = 2.0f * ((giy * ((float)(H - 1) / 2.f)) / (float)(H - 1));
        float* dp = dflow + pix * dflow_ld;
        dp[0] = gx;
        dp[1] = gyv;
        for (int k = 2; k < dflow_ld; ++k) dp[k] = 0.f;
    }
}

// ---------------------------------------------------------------- warp backward without atomics (gather form)
// The scatter kernel above issues 4 float atomics per (pixel, channel); on gfx950 they execute at the memory side (2 GB of
// atomic traffic per call at 540p x 4 images, 1.6 ms) and make the feature gradient order-dependent.  Two passes instead:
//   src pass (per source pixel p): geometry once, the flow gradient (needs feat at the 4 corners), and a 20-byte record
//     {code = corner offset (y0 - y, x0 - x), w[4] = the 4 bilinear weights, 0 where the corner is outside};
//   gather pass (per destination pixel q): q receives w_k(p) * dout[p] from every p in the 9x9 window around it whose corner
//     k is q - found by scanning the window's records (staged in LDS) for  q - p - offset(p) in {0,1}^2.  Fixed summation
//     order => deterministic.
// A source whose corner offset falls outside [-4, 3] (|flow| >= 4 px) cannot be found by the window: the src pass scatters
// it with atomics as before and writes a zero record, so the result is complete for any flow.
constexpr int WG_R = 4;                         // window radius of the gather pass
constexpr int WG_TH = 8, WG_TW = 32;            // destination tile
constexpr int WG_HH = WG_TH + 2 * WG_R, WG_HW = WG_TW + 2 * WG_R;   // 16 x 40 records

// 16 lanes per pixel x float4 = 64 channels per pass (C % 4 == 0; lanes with c4*4 >= C idle; C > 64: channel loop)
template <bool FB, bool DB>
__global__ __launch_bounds__(256) void warp_bwd_src_kernel(const float* __restrict__ dout, int dout_ld, int dout_coff,
                                                           const float* __restrict__ feat, int feat_ld,
                                                           const float* __restrict__ flow, int flow_ld, int C, int H, int W,
                                                           float* __restrict__ dfeat, int dfeat_ld,
                                                           float* __restrict__ dflow, int dflow_ld,
                                                           float4* __restrict__ rec_w, int* __restrict__ rec_code, int tilesX,
                                                           int tilesY, int* __restrict__ far_flag) {
    constexpr int feat_bf16 = FB, dout_bf16 = DB;
    // Phase A: one thread per pixel - geometry, record, and what phase B needs (corner indices, corner weights for a scatter, the
    // four (d ix, d iy) factors of the flow gradient) into LDS.  Phase B: 16 lanes per pixel do the loads and the channel sums.
    // (With the geometry in every one of the 16 lanes the kernel was VALU-bound, see warp_fwd_kernel.)
    __shared__ int4 so[WP];
    __shared__ float4 swt[WP], ssx[WP], ssy[WP];
    __shared__ int sflag[WP];                                  // bit k: corner k valid; bit 4: scatter this source
    int bt = xcd_tile(blockIdx.x, gridDim.x);                 // 8 x 32-pixel tiles in XCD order, see warp_fwd_kernel
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int img = n * H * W;                                // (N*H*W < 2^31: host check)
    {
        const int x = tx * WT_W + (threadIdx.x & (WT_W - 1)), y = ty * WT_H + (threadIdx.x >> 5);
        int4 o = make_int4(0, 0, 0, 0);
        float4 wt = make_float4(0.f, 0.f, 0.f, 0.f), sx = wt, sy = wt;
        int flag = 0;
        if (x < W && y < H) {
            const long pix = img + (long)y * W + x;
            const WarpGeom g = warp_geom(flow[pix * flow_ld], flow[pix * flow_ld + 1], x, y, H, W);
            const float fx0 = floorf(g.ix), fy0 = floorf(g.iy);
            const float x_se = fx0 + 1.f, y_se = fy0 + 1.f;
            const int ox = g.x0 - x, oy = g.y0 - y;
            const bool any = g.vnw || g.vne || g.vsw || g.vse;
            const bool near = ox >= -WG_R && ox <= WG_R - 1 && oy >= -WG_R && oy <= WG_R - 1;
            // far_flag != NULL (overwrite mode: dfeat is not initialised yet): a far source is only flagged here and scattered by
            // warp_bwd_far_kernel behind the gather pass
            const bool scatter = any && !near && far_flag == nullptr;
            if (any && !near && far_flag != nullptr) *far_flag = 1;
            o = make_int4(img + (g.vnw ? g.y0 * W + g.x0 : 0), img + (g.vne ? g.y0 * W + g.x0 + 1 : 0),
                          img + (g.vsw ? (g.y0 + 1) * W + g.x0 : 0), img + (g.vse ? (g.y0 + 1) * W + g.x0 + 1 : 0));
            wt = make_float4(g.wnw, g.wne, g.wsw, g.wse);
            sx = make_float4(-(y_se - g.iy), (y_se - g.iy), -(g.iy - fy0), (g.iy - fy0));
            sy = make_float4(-(x_se - g.ix), -(g.ix - fx0), (x_se - g.ix), (g.ix - fx0));
            flag = (g.vnw ? 1 : 0) | (g.vne ? 2 : 0) | (g.vsw ? 4 : 0) | (g.vse ? 8 : 0) | (scatter ? 16 : 0);
            const bool rec = any && near;
            rec_w[pix] = rec ? make_float4(g.vnw ? g.wnw : 0.f, g.vne ? g.wne : 0.f, g.vsw ? g.wsw : 0.f, g.vse ? g.wse : 0.f)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
            rec_code[pix] = rec ? ((oy + WG_R) << 4) | (ox + WG_R) : -1;
        }
        so[threadIdx.x] = o; swt[threadIdx.x] = wt; ssx[threadIdx.x] = sx; ssy[threadIdx.x] = sy; sflag[threadIdx.x] = flag;
    }
    __syncthreads();
    const int c4 = threadIdx.x & 15;
#pragma unroll 2
    for (int it = 0; it < WP / 16; ++it) {
        const int pl = it * 16 + (threadIdx.x >> 4);
        const int x = tx * WT_W + (pl & (WT_W - 1)), y = ty * WT_H + (pl >> 5);
        if (x >= W || y >= H) continue;                       // (whole 16-lane groups skip together)
        const long pix = img + (long)y * W + x;
        const int4 o = so[pl];
        const float4 wt = swt[pl], sx = ssx[pl], sy = ssy[pl];
        const int flag = sflag[pl];
        float gix = 0.f, giy = 0.f;
        for (int ch = 4 * c4; ch < C; ch += 64) {
            const float4 go = ldx4(dout, (size_t)pix * dout_ld + dout_coff + ch, dout_bf16);
            // unconditional corner loads (an outside corner reads the image's first pixel and contributes nothing)
            const float4 f0 = ldx4(feat, (size_t)o.x * feat_ld + ch, feat_bf16), f1 = ldx4(feat, (size_t)o.y * feat_ld + ch, feat_bf16);
            const float4 f2 = ldx4(feat, (size_t)o.z * feat_ld + ch, feat_bf16), f3 = ldx4(feat, (size_t)o.w * feat_ld + ch, feat_bf16);
            auto corner = [&](int bit, int oo, const float4& f, float wgt, float csx, float csy) {
                if (!(flag & bit)) return;
                const float dsum = f.x * go.x + f.y * go.y + f.z * go.z + f.w * go.w;
                gix += csx * dsum;
                giy += csy * dsum;
                if (flag & 16) {
                    float* db = dfeat + (size_t)oo * dfeat_ld + ch;
                    atomicAdd(db, go.x * wgt); atomicAdd(db + 1, go.y * wgt); atomicAdd(db + 2, go.z * wgt); atomicAdd(db + 3, go.w * wgt);
                }
            };
            corner(1, o.x, f0, wt.x, sx.x, sy.x);
            corner(2, o.y, f1, wt.y, sx.y, sy.y);
            corner(4, o.z, f2, wt.z, sx.z, sy.z);
            corner(8, o.w, f3, wt.w, sx.w, sy.w);
        }
        gix = group_sum(gix, 16);
        giy = group_sum(giy, 16);
        if (c4 == 0) {
            // grid_sample multiplies by (size-1)/2; the normalisation's autograd divides by (size-1) and doubles
            const float gx = 2.0f * ((gix * ((float)(W - 1) / 2.f)) / (float)(W - 1));
            const float gyv = 2.0f * ((giy * ((float)(H - 1) / 2.f)) / (float)(H - 1));
            float* dp = dflow + pix * dflow_ld;
            dp[0] = gx;
            dp[1] = gyv;
            for (int k = 2; k < dflow_ld; ++k) dp[k] = 0.f;
        }
    }
}

// dst[0..3] += v for an fp32 or a bf16-stored gradient.  bf16: a compare-and-swap loop on each 32-bit word (two bf16 values);
// only the rare paths use it (far sources, hit-list overflow).
template <bool FB16>
__device__ __forceinline__ void warp_atomic_add4(float* base, size_t idx, float4 v) {
    if constexpr (!FB16) {
        float* p = base + idx;
        atomicAdd(p, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
    } else {
        unsigned* w = reinterpret_cast<unsigned*>(reinterpret_cast<__bf16*>(base) + idx);     // idx % 4 == 0: 8-byte aligned
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float a = h ? v.z : v.x, b = h ? v.w : v.y;
            unsigned old = w[h], assumed;
            do {
                assumed = old;
                typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                const b2 pk = {(__bf16)(__uint_as_float(assumed << 16) + a), (__bf16)(__uint_as_float(assumed & 0xffff0000u) + b)};
                old = atomicCAS(w + h, assumed, __builtin_bit_cast(unsigned, pk));
            } while (old != assumed);
        }
    }
}

// Overwrite mode, behind the gather pass: the sources the window cannot reach (|flow| >= 4 px), scattered with atomics as the
// src pass does in the accumulate mode.  Leaves at once when the src pass flagged none (the usual case).  One thread per source
// pixel finds the far ones; each far source of a wave is then scattered by the whole wave: lane = (corner, 16-byte channel
// piece), i.e. a corner's C channels are coalesced 256-byte accesses and the four corners go side by side (one thread per
// pixel looping over C channels x 4 corners made a motion field with many far sources a serial tail: with 20 % of the sources
// 5 - 8 px away, 8 x 540 x 960 x 64 channels took 11.3 ms (fp32 dfeat) / 5.4 ms (bf16) that way and 3.8 / 2.1 ms this way,
// against 1.3 ms when no source is far: tools/warp_probe.py).
template <bool FB16>                                          // FB16: dfeat is stored as bf16
__global__ __launch_bounds__(256) void warp_bwd_far_kernel(const float* __restrict__ dout, int dout_ld, int dout_coff,
                                                           const float* __restrict__ flow, int flow_ld, int C, int H, int W,
                                                           float* __restrict__ dfeat, int dfeat_ld, long npix, int dout_bf16,
                                                           const int* __restrict__ far_flag) {
    if (*far_flag == 0) return;
    const long pix = blockIdx.x * 256L + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool far = false;
    int orel = 0, vmask = 0;                                  // north-west corner relative to the source pixel; valid corners
    float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pix < npix) {
        const long rowi = idiv(pix, W, npix);
        const int x = (int)(pix - rowi * W);
        const int y = (int)(rowi - idiv(rowi, H, npix) * H);
        const WarpGeom g = warp_geom(flow[pix * flow_ld], flow[pix * flow_ld + 1], x, y, H, W);
        const int ox = g.x0 - x, oy = g.y0 - y;
        const bool any = g.vnw || g.vne || g.vsw || g.vse;
        const bool near = ox >= -WG_R && ox <= WG_R - 1 && oy >= -WG_R && oy <= WG_R - 1;
        far = any && !near;
        orel = oy * W + ox;
        vmask = (g.vnw ? 1 : 0) | (g.vne ? 2 : 0) | (g.vsw ? 4 : 0) | (g.vse ? 8 : 0);
        wq = make_float4(g.wnw, g.wne, g.wsw, g.wse);
    }
    unsigned long long m = __ballot(far);                     // (the same in every lane: the loop below is wave-uniform)
    const int corner = lane >> 4, c4 = lane & 15;
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const long spix = pix - lane + src;                   // the wave's pixels are consecutive
        const int so = __shfl(orel, src, 64), sv = __shfl(vmask, src, 64);
        const float w0 = __shfl(wq.x, src, 64), w1 = __shfl(wq.y, src, 64), w2 = __shfl(wq.z, src, 64), w3 = __shfl(wq.w, src, 64);
        const float wgt = corner == 0 ? w0 : corner == 1 ? w1 : corner == 2 ? w2 : w3;
        if ((sv >> corner) & 1) {
            const long o = spix + so + (corner & 1) + (corner >> 1) * W;
            for (int ch = 4 * c4; ch < C; ch += 64) {
                const float4 go = ldx4(dout, (size_t)spix * dout_ld + dout_coff + ch, dout_bf16);
                warp_atomic_add4<FB16>(dfeat, (size_t)o * dfeat_ld + ch, make_float4(go.x * wgt, go.y * wgt, go.z * wgt, go.w * wgt));
            }
        }
    }
}

// grid = destination tiles of 8 x 32 pixels.  Phase 1: thread = one destination pixel, scans its 9x9 window of records and
// lists the contributing sources (window index, weight) in LDS.  Phase 2: 16 lanes per pixel (one float4 of channels
// each, so a pixel's 64 channels are one coalesced 256-byte access) walk the lists and add into dfeat.
constexpr int WG_MAXHIT = 12;                   // list length per pixel; further hits (a strongly contracting flow) are
                                                // applied by phase 1 itself, one pixel per lane
template <bool DB, bool FB16>                                 // DB: dout is bf16; FB16: dfeat is bf16
__global__ __launch_bounds__(256) void warp_bwd_gather_kernel(const float* __restrict__ dout, int dout_ld, int dout_coff,
                                                              const float4* __restrict__ rec_w,
                                                              const int* __restrict__ rec_code, int C, int H, int W,
                                                              int tilesX, int tilesY, float* __restrict__ dfeat,
                                                              int dfeat_ld, int overwrite) {
    constexpr int dout_bf16 = DB;
    __shared__ float4 lw[WG_HH * WG_HW];
    __shared__ int lc[WG_HH * WG_HW];
    __shared__ int hit_n[WG_TH * WG_TW];
    __shared__ int hit_p[WG_TH * WG_TW * WG_MAXHIT];        // source pixel, relative: (sy + R) * 16 + (sx + R)
    __shared__ float hit_w[WG_TH * WG_TW * WG_MAXHIT];
    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    for (int i = threadIdx.x; i < WG_HH * WG_HW; i += 256) {
        const int hy = i / WG_HW, hx = i - hy * WG_HW;
        const int gy = ty * WG_TH + hy - WG_R, gx = tx * WG_TW + hx - WG_R;
        const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const long pp = ok ? (long)(n * H + gy) * W + gx : 0;
        lc[i] = ok ? rec_code[pp] : -1;
        lw[i] = ok ? rec_w[pp] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // window scan of destination pixel (ly, lx): calls hit(k, sy, sx, weight) for its k-th contributing source, in a fixed order
    auto scan = [&](int ly, int lx, auto&& hit) {
        int cnt = 0;
        for (int sy = -WG_R; sy <= WG_R; ++sy)
            for (int sx = -WG_R; sx <= WG_R; ++sx) {
                const int hi = (ly + WG_R + sy) * WG_HW + lx + WG_R + sx;
                const int code = lc[hi];
                if (code < 0) continue;
                const int a = -sy - ((code >> 4) - WG_R), b = -sx - ((code & 15) - WG_R);   // q - p - offset(p)
                if ((unsigned)a > 1u || (unsigned)b > 1u) continue;
                const float4 w4 = lw[hi];
                const float wgt = a == 0 ? (b == 0 ? w4.x : w4.y) : (b == 0 ? w4.z : w4.w);
                if (wgt == 0.f) continue;
                hit(cnt, sy, sx, wgt);
                ++cnt;
            }
        return cnt;
    };
    const int ly1 = threadIdx.x / WG_TW, lx1 = threadIdx.x % WG_TW;
    const int qy1 = ty * WG_TH + ly1, qx1 = tx * WG_TW + lx1;
    int total1 = 0;
    if (qy1 < H && qx1 < W)                                   // ---- phase 1: list the first WG_MAXHIT sources
        total1 = scan(ly1, lx1, [&](int k, int sy, int sx, float wgt) {
            if (k < WG_MAXHIT) {
                hit_p[threadIdx.x * WG_MAXHIT + k] = ((sy + WG_R) << 4) | (sx + WG_R);
                hit_w[threadIdx.x * WG_MAXHIT + k] = wgt;
            }
        });
    hit_n[threadIdx.x] = total1 < WG_MAXHIT ? total1 : WG_MAXHIT;
    __syncthreads();
    // ---- phase 2
    const int c4 = threadIdx.x & 15;
    for (int qi = threadIdx.x >> 4; qi < WG_TH * WG_TW; qi += 16) {
        const int cnt = hit_n[qi];
        const int ly = qi / WG_TW, lx = qi % WG_TW;
        const int qy = ty * WG_TH + ly, qx = tx * WG_TW + lx;
        if ((cnt == 0 && !overwrite) || qy >= H || qx >= W) continue;   // (uniform over the 16 lanes of the pixel)
        const long qpix = (long)(n * H + qy) * W + qx;
        for (int ch = 4 * c4; ch < C; ch += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            // four hits per trip, their loads in flight together (one load -> wait -> fma per hit serialised a memory latency per
            // hit); a slot past the list repeats hit 0 with weight 0: fma(0, v, acc) == acc, so the sum and its order are unchanged
            for (int k0 = 0; k0 < cnt; k0 += 4) {
                float4 v[4];
                float wv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool in = k0 + u < cnt;
                    const int hp = hit_p[qi * WG_MAXHIT + (in ? k0 + u : 0)];
                    wv[u] = in ? hit_w[qi * WG_MAXHIT + k0 + u] : 0.f;
                    const int sy = (hp >> 4) - WG_R, sx = (hp & 15) - WG_R;
                    v[u] = ldx4(dout, ((size_t)(n * H + qy + sy) * W + qx + sx) * dout_ld + dout_coff + ch, dout_bf16);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc.x += wv[u] * v[u].x; acc.y += wv[u] * v[u].y; acc.z += wv[u] * v[u].z; acc.w += wv[u] * v[u].w;
                }
            }
            const size_t di = (size_t)qpix * dfeat_ld + ch;
            if (overwrite) {                                  // the first writer of dfeat: every pixel of the tile is written
                stx4(dfeat, di, FB16, acc);
            } else {
                float4 o = ldx4(dfeat, di, FB16);
                o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
                stx4(dfeat, di, FB16, o);
            }
        }
    }
    // ---- phase 3: a pixel with more sources than the list holds (a strongly contracting flow) adds the rest itself, behind
    // phase 2's write of that pixel (still no atomics: q is ours)
    if (__syncthreads_or(total1 > WG_MAXHIT)) {
        if (total1 > WG_MAXHIT) {
            const size_t dbase = ((size_t)(n * H + qy1) * W + qx1) * dfeat_ld;
            scan(ly1, lx1, [&](int k, int sy, int sx, float wgt) {
                if (k < WG_MAXHIT) return;
                const size_t so = ((size_t)(n * H + qy1 + sy) * W + qx1 + sx) * dout_ld + dout_coff;
                for (int ch = 0; ch < C; ch += 4) {           // C % 4 == 0
                    const float4 v = ldx4(dout, so + ch, dout_bf16);
                    // (atomics only for their L2 scope - phase 2 wrote this pixel from another wave; one thread per pixel, in
                    // scan order: still deterministic)
                    warp_atomic_add4<FB16>(dfeat, dbase + ch, make_float4(wgt * v.x, wgt * v.y, wgt * v.z, wgt * v.w));
                }
            });
        }
    }
}

// NVQ_MATH_BF16 variants (corr_mfma.hip)
bool corr_mfma_supported(int C);
int corr_forward_mfma(const float* x1, int x1_ld, const float* x2, int x2_ld, int x2_images, int C, int N, int H, int W,
                      float* out, int out_ld, int out_bf16, int in_bf16, hipStream_t s);
int corr_backward_mfma(int which, const float* dcorr, int dcorr_ld, int dcorr_bf16, const float* other, int other_ld,
                       int other_images, int C, int N, int H, int W, float* dx, int dx_ld, int dx_coff, int accumulate,
                       int other_bf16, int groups, float* dx16, int dx16_ld, const nvq_corr_addends* addends, hipStream_t s);

}  // namespace nvq

using namespace nvq;

extern "C" {

int nvq_correlation_forward(const float* x1, int x1_ld, const float* x2, int x2_ld, int x2_images, int C,
                            int N, int H, int W, float* out, int out_ld, int math, int out_bf16, int in_bf16,
                            void* stream) {
    NVQ_REQUIRE(C % 4 == 0 && x1_ld % 4 == 0 && x2_ld % 4 == 0 && out_ld % 4 == 0 && out_ld >= 84 &&
                    aligned16(x1) && aligned16(x2) && aligned16(out),
                "correlation_forward: alignment (C %d out_ld %d)", C, out_ld);
    NVQ_REQUIRE(x2_images > 0, "correlation_forward: x2_images");
    if (math == NVQ_MATH_BF16 && corr_mfma_supported(C))
        return corr_forward_mfma(x1, x1_ld, x2, x2_ld, x2_images, C, N, H, W, out, out_ld, out_bf16, in_bf16,
                                 (hipStream_t)stream);
    NVQ_REQUIRE(!out_bf16 && !in_bf16, "correlation_forward: bf16 tensors need NVQ_MATH_BF16 and C in {32, 64}");
    const int tilesX = (W + CT_W - 1) / CT_W, tilesY = (H + CT_H - 1) / CT_H;
    hipLaunchKernelGGL(corr_fwd_kernel, dim3((unsigned)((long)tilesX * tilesY * N)), dim3(256), 0, (hipStream_t)stream,
                       x1, x1_ld, x2, x2_ld, x2_images, C, H, W, tilesX, tilesY, out, out_ld);
    return check_launch("correlation_forward");
}

int nvq_correlation_backward(int which, const float* dcorr, int dcorr_ld, const float* other, int other_ld,
                             int other_images, int C, int N, int H, int W, float* dx, int dx_ld, int dx_coff,
                             int accumulate, int math, int dcorr_bf16, int other_bf16, int groups, float* dx_bf16_out,
                             int dx_bf16_ld, const nvq_corr_addends* addends, void* stream) {
    NVQ_REQUIRE(!addends || dx_bf16_out, "correlation_backward: bf16 addends go with the bf16 output");
    NVQ_REQUIRE(!dx_bf16_out || (math == NVQ_MATH_BF16 && corr_mfma_supported(C) && dx_bf16_ld % 4 == 0 && dx_bf16_ld >= C &&
                                 (reinterpret_cast<uintptr_t>(dx_bf16_out) & 7) == 0),
                "correlation_backward: the bf16 output needs NVQ_MATH_BF16, C in {32, 64} and an 8-byte aligned row");
    NVQ_REQUIRE(which == 1 || which == 2, "correlation_backward: which %d", which);
    NVQ_REQUIRE(groups >= 1 && (groups == 1 || which == 2), "correlation_backward: groups %d (only which == 2 merges frames)", groups);
    NVQ_REQUIRE(groups == 1 || other_images >= N * groups, "correlation_backward: %d groups need %d images of `other`", groups, N * groups);
    NVQ_REQUIRE(C % 4 == 0 && dcorr_ld % 4 == 0 && dcorr_ld >= 84 && other_ld % 4 == 0 && dx_ld % 4 == 0 &&
                    dx_coff % 4 == 0 && aligned16(dcorr) && aligned16(other) && aligned16(dx),
                "correlation_backward: alignment");
    NVQ_REQUIRE(other_images > 0, "correlation_backward: other_images");
    if (math == NVQ_MATH_BF16 && corr_mfma_supported(C))
        return corr_backward_mfma(which, dcorr, dcorr_ld, dcorr_bf16, other, other_ld, other_images, C, N, H, W, dx, dx_ld,
                                  dx_coff, accumulate, other_bf16, groups, dx_bf16_out, dx_bf16_ld, addends, (hipStream_t)stream);
    NVQ_REQUIRE(!dcorr_bf16 && !other_bf16, "correlation_backward: bf16 tensors need NVQ_MATH_BF16 and C in {32, 64}");
    const int tilesX = (W + CT_W - 1) / CT_W, tilesY = (H + CT_H - 1) / CT_H;
    const dim3 grid((unsigned)((long)tilesX * tilesY * N));
    if (which == 1)
        hipLaunchKernelGGL((corr_bwd_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, dcorr, dcorr_ld, other, other_ld,
                           other_images, C, H, W, tilesX, tilesY, dx, dx_ld, dx_coff, accumulate);
    else
        for (int gi = 0; gi < groups; ++gi) {              // exact-fp32 path: one accumulating launch per reference frame
            const size_t img = (size_t)gi * N * H * W;
            hipLaunchKernelGGL((corr_bwd_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, dcorr + img * dcorr_ld, dcorr_ld,
                               other + img * other_ld, other_ld, N, C, H, W, tilesX, tilesY, dx, dx_ld, dx_coff,
                               gi > 0 ? 1 : accumulate);
        }
    return check_launch("correlation_backward");
}

int nvq_warp_forward(const float* feat, int feat_ld, const float* flow, int flow_ld, int C, int N, int H, int W,
                     float* out, int out_ld, int out_coff, int feat_bf16, int out_bf16, void* stream) {
    NVQ_REQUIRE(C % 4 == 0 && feat_ld % 4 == 0 && out_ld % 4 == 0 && out_coff % 4 == 0 && flow_ld >= 2 &&
                    aligned16(feat) && aligned16(out),
                "warp_forward: alignment");
    const long npix = (long)N * H * W;
    NVQ_REQUIRE(npix < ((long)1 << 31), "warp_forward: too many pixels");
    const int tilesX = (W + WT_W - 1) / WT_W, tilesY = (H + WT_H - 1) / WT_H;
#define NVQ_WF(F_, O_) hipLaunchKernelGGL((warp_fwd_kernel<F_, O_>), dim3((unsigned)((long)tilesX * tilesY * N)), dim3(256), 0, \
                                          (hipStream_t)stream, feat, feat_ld, flow, flow_ld, C, H, W, out, out_ld, out_coff, tilesX, tilesY)
    if (feat_bf16) { if (out_bf16) NVQ_WF(true, true); else NVQ_WF(true, false); }
    else { if (out_bf16) NVQ_WF(false, true); else NVQ_WF(false, false); }
#undef NVQ_WF
    return check_launch("warp_forward");
}

int nvq_warp_backward(const float* dout, int dout_ld, int dout_coff, const float* feat, int feat_ld,
                      const float* flow, int flow_ld, int C, int N, int H, int W, float* dfeat, int dfeat_ld,
                      float* dflow, int dflow_ld, float* records, size_t records_bytes, int feat_bf16, int dout_bf16,
                      int overwrite, int dfeat_bf16, void* stream) {
    NVQ_REQUIRE(!dfeat_bf16 || (records && overwrite), "warp_backward: a bf16 dfeat needs the gather form in the overwrite mode");
    NVQ_REQUIRE(C >= 4 && C <= 1024 && (C & (C - 1)) == 0, "warp_backward: C %d must be a power of two >= 4", C);
    NVQ_REQUIRE(!(feat_bf16 || dout_bf16) || records, "warp_backward: bf16-stored tensors need the gather form (records != NULL)");
    NVQ_REQUIRE(flow_ld >= 2 && dflow_ld >= 2, "warp_backward: flow ld");
    if (records) {                                            // gather form (deterministic, no atomics for |flow| < 4 px)
        const long npix = (long)N * H * W;
        NVQ_REQUIRE(records_bytes >= (size_t)npix * 20 + (overwrite ? 16 : 0) && aligned16(records),
                    "warp_backward: records buffer needs 20 B per pixel (+ 16 B in the overwrite mode)");
        NVQ_REQUIRE(dout_ld % 4 == 0 && dout_coff % 4 == 0 && feat_ld % 4 == 0 && dfeat_ld % 4 == 0 && aligned16(dout) &&
                        aligned16(feat) && aligned16(dfeat),
                    "warp_backward: alignment");
        float4* rec_w = reinterpret_cast<float4*>(records);
        int* rec_code = reinterpret_cast<int*>(rec_w + npix);
        hipStream_t s = (hipStream_t)stream;
        int* far_flag = overwrite ? rec_code + npix : nullptr;
        if (far_flag && hipMemsetAsync(far_flag, 0, sizeof(int), s) != hipSuccess) return check_launch("warp_backward(flag)");
        NVQ_REQUIRE(npix < ((long)1 << 31), "warp_backward: too many pixels");
        const int tilesX = (W + WG_TW - 1) / WG_TW, tilesY = (H + WG_TH - 1) / WG_TH;
        static_assert(WG_TW == WT_W && WG_TH == WT_H, "the src and gather passes share the tile shape");
#define NVQ_WS(F_, D_) hipLaunchKernelGGL((warp_bwd_src_kernel<F_, D_>), dim3((unsigned)((long)tilesX * tilesY * N)), dim3(256), 0, s, \
                                          dout, dout_ld, dout_coff, feat, feat_ld, flow, flow_ld, C, H, W, dfeat, dfeat_ld, dflow,       \
                                          dflow_ld, rec_w, rec_code, tilesX, tilesY, far_flag)
        if (feat_bf16) { if (dout_bf16) NVQ_WS(true, true); else NVQ_WS(true, false); }
        else { if (dout_bf16) NVQ_WS(false, true); else NVQ_WS(false, false); }
#undef NVQ_WS
        int rc = check_launch("warp_backward(src)");
        if (rc) return rc;
#define NVQ_WG(D_, F_) hipLaunchKernelGGL((warp_bwd_gather_kernel<D_, F_>), dim3((unsigned)((long)tilesX * tilesY * N)), dim3(256), 0, s, \
                                          dout, dout_ld, dout_coff, rec_w, rec_code, C, H, W, tilesX, tilesY, dfeat, dfeat_ld, overwrite)
        if (dout_bf16) { if (dfeat_bf16) NVQ_WG(true, true); else NVQ_WG(true, false); }
        else { if (dfeat_bf16) NVQ_WG(false, true); else NVQ_WG(false, false); }
#undef NVQ_WG
        rc = check_launch("warp_backward(gather)");
        if (rc || !overwrite) return rc;
if (dfeat_bf16) {
                    hipLaunchKernelGGL(warp_bwd_far_kernel<true>, dim3(ceil_div(npix, 256)), dim3(256), 0, s, dout, dout_ld, dout_coff, flow,
                           flow_ld, C, H, W, dfeat, dfeat_ld, npix, dout_bf16, far_flag);
        } else {
                    hipLaunchKernelGGL(warp_bwd_far_kernel<false>, dim3(ceil_div(npix, 256)), dim3(256), 0, s, dout, dout_ld, dout_coff, flow,
                           flow_ld, C, H, W, dfeat, dfeat_ld, npix, dout_bf16, far_flag);
        }
        return check_launch("warp_backward(far)");
    }
    NVQ_REQUIRE(!overwrite, "warp_backward: the overwrite mode needs the gather form (records != NULL)");
    const long total = (long)N * H * W * (C < 64 ? C : 64);
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, dout_ld,
                       dout_coff, feat, feat_ld, flow, flow_ld, C, H, W, dfeat, dfeat_ld, dflow, dflow_ld, total);
    return check_launch("warp_backward");
}

}  // extern "C"
