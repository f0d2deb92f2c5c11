// Temporal aggregation: softmax-weighted frame sum and CBAM refinement, forward and backward.
// fp32 NHWC; threads are (pixel, 4-channel group); C/4 is a power of two <= 64 so the lanes of
// one pixel share a wave and per-pixel channel reductions are xor-shuffles.
#include "common.h"

namespace nvq {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__device__ __forceinline__ float4 plane_reduce4b(float4 v, float4* buf, int C4, int npl) {
    __syncthreads();
    buf[threadIdx.x] = v;
    __syncthreads();
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((int)threadIdx.x < C4)
        for (int k = 0; k < npl; ++k) {
            const float4 t = buf[k * C4 + threadIdx.x];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
    return s;
}

static inline int tsum_blocks_host(int H, int W) {
    int nb = ceil_div((long)H * W, 1024);
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    return nb;
}

// ---------------------------------------------------------------- softmax-weighted sum
// grid (nblk, N)
template <int TMAX, int TU, bool AB, bool WB>                 // WB: `weighted` is stored as bf16 (the GAP sums use the unrounded values)
__global__ __launch_bounds__(256) void tsum_fwd_kernel(const float* __restrict__ aligned, int aligned_ld,
                                                       const float* __restrict__ logits, int logits_ld, int T,
                                                       int C, long HW, float* __restrict__ attn, int attn_ld,
                                                       float* __restrict__ weighted, int weighted_ld,
                                                       float* __restrict__ gap_partial) {
    constexpr int aligned_bf16 = AB;              // (a template parameter: as a run-time flag every ldx4 is a branch with its own wait)
    __shared__ float4 buf[256];
    const int C4 = C >> 2;
    const int npl = 256 / C4;
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
    const long base = (long)blockIdx.y * HW;
    float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);
    // TU pixels per trip, all their loads issued before the first use (one pixel per trip left 36 B per lane in flight and the
    // kernel at 2.5 TB/s); a pixel past the end is clamped for its loads and skipped at the stores.  T <= TMAX (4: TU = 4;
    // 8: TU = 2 - the register budget of four waves per SIMD)
    const long stride = (long)gridDim.x * npl;
    for (long p0 = (long)blockIdx.x * npl + pl; p0 < HW; p0 += TU * stride) {
        float lg[TU][TMAX];
        float4 v[TU][TMAX];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const long pu = p0 + u * stride;
            const long pix = base + (pu < HW ? pu : p0);
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                lg[u][t] = t < T ? logits[pix * logits_ld + t] : -3.4e38f;
                if (t < T) v[u][t] = ldx4(aligned, (size_t)pix * aligned_ld + t * C + 4 * c4, aligned_bf16);
            }
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const long pu = p0 + u * stride;
            if (pu >= HW) break;
            const long pix = base + pu;
            float mx = -3.4e38f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t) mx = fmaxf(mx, lg[u][t]);
            float den = 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                lg[u][t] = t < T ? __expf(lg[u][t] - mx) : 0.f;
                den += lg[u][t];
            }
            const float inv = 1.f / den;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                if (t < T) {
                    const float a = lg[u][t] * inv;
                    o.x += v[u][t].x * a; o.y += v[u][t].y * a; o.z += v[u][t].z * a; o.w += v[u][t].w * a;
                    if (c4 == 0) attn[pix * attn_ld + t] = a;
                }
            }
            if (c4 == 0)
                for (int t = T; t < attn_ld; ++t) attn[pix * attn_ld + t] = 0.f;
            stx4(weighted, pix * weighted_ld + 4 * c4, WB, o);
            gsum.x += o.x; gsum.y += o.y; gsum.z += o.z; gsum.w += o.w;
        }
    }
    const float4 r = plane_reduce4b(gsum, buf, C4, npl);
    if ((int)threadIdx.x < C4)
        st4(gap_partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * C + 4 * c4, r);
}

template <bool ALLB, bool WB>                                 // ALLB: aligned and daligned are bf16, known at compile time; WB: dweighted too
__global__ __launch_bounds__(256) void tsum_bwd_kernel(const float* __restrict__ dweighted, int dweighted_ld,
                                                       const float* __restrict__ dgap_pix,
                                                       const float* __restrict__ aligned, int aligned_ld,
                                                       const float* __restrict__ attn, int attn_ld, int T, int C,
                                                       long HW, float* __restrict__ daligned, int daligned_ld,
                                                       float* __restrict__ dlogits, int dlogits_ld, long total,
                                                       int aligned_bf16_, int daligned_bf16_) {
    const int aligned_bf16 = ALLB ? 1 : aligned_bf16_, daligned_bf16 = ALLB ? 1 : daligned_bf16_;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const int C4 = C >> 2;
    const long pix = idiv(gid, C4, total);
    const int c4 = (int)(gid - pix * C4);
    const int n = (int)idiv(pix, HW, total);
    float4 dw = ldx4(dweighted, pix * dweighted_ld + 4 * c4, WB);
    if (dgap_pix) {
        const float4 gp = ld4(dgap_pix + (size_t)n * C + 4 * c4);
        dw.x += gp.x; dw.y += gp.y; dw.z += gp.z; dw.w += gp.w;
    }
    float a[NVQ_MAX_T], dot[NVQ_MAX_T];
    float sdot = 0.f;
#pragma unroll
    for (int t = 0; t < NVQ_MAX_T; ++t) {
        a[t] = 0.f; dot[t] = 0.f;
        if (t < T) {
            a[t] = attn[pix * attn_ld + t];
            const float4 v = ldx4(aligned, (size_t)pix * aligned_ld + t * C + 4 * c4, aligned_bf16);
            stx4(daligned, (size_t)pix * daligned_ld + t * C + 4 * c4, daligned_bf16,
                 make_float4(dw.x * a[t], dw.y * a[t], dw.z * a[t], dw.w * a[t]));
            dot[t] = group_sum(dw.x * v.x + dw.y * v.y + dw.z * v.z + dw.w * v.w, C4);
            sdot += a[t] * dot[t];
        }
    }
    if (c4 == 0) {
#pragma unroll
        for (int t = 0; t < NVQ_MAX_T; ++t)
            if (t < T) dlogits[pix * dlogits_ld + t] = a[t] * (dot[t] - sdot);
        for (int t = T; t < dlogits_ld; ++t) dlogits[pix * dlogits_ld + t] = 0.f;
    }
}

// ---------------------------------------------------------------- CBAM forward
// one block per image
__global__ __launch_bounds__(256) void cbam_channel_kernel(const float* __restrict__ gap_partial, int nblk, int C,
                                                           int R, long HW, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, float* __restrict__ gap,
                                                           float* __restrict__ hid, float* __restrict__ ca) {
    __shared__ float sg[256];
    __shared__ float sh[64];
    const int n = blockIdx.x, c = threadIdx.x;
    if (c < C) {
        double s = 0.0;
        const float* p = gap_partial + (size_t)n * nblk * C + c;
        for (int b = 0; b < nblk; ++b) s += (double)p[(size_t)b * C];
        const float g = (float)(s / (double)HW);
        sg[c] = g;
        gap[(size_t)n * C + c] = g;
    }
    __syncthreads();
    if (c < R) {
        float s = 0.f;
        for (int k = 0; k < C; ++k) s += w1[c * C + k] * sg[k];
        s = fmaxf(s, 0.f);
        sh[c] = s;
        hid[(size_t)n * R + c] = s;
    }
    __syncthreads();
    if (c < C) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += w2[c * R + r] * sh[r];
        ca[(size_t)n * C + c] = sigmoidf_(s);
    }
}

template <bool XB>                                            // XB: x is stored as bf16
__global__ __launch_bounds__(256) void cbam_pool_kernel(const float* __restrict__ x, int x_ld,
                                                        const float* __restrict__ ca, int C, long HW,
                                                        float* __restrict__ sm, int* __restrict__ amax, long total) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const int C4 = C >> 2;
    const long pix = idiv(gid, C4, total);
    const int c4 = (int)(gid - pix * C4);
    const int n = (int)idiv(pix, HW, total);
    const float4 v = ldx4(x, pix * x_ld + 4 * c4, XB);
    const float4 a = ld4(ca + (size_t)n * C + 4 * c4);
    const float e[4] = {v.x * a.x, v.y * a.y, v.z * a.z, v.w * a.w};
    const float s = group_sum(e[0] + e[1] + e[2] + e[3], C4);
    float best = e[0];
    int bi = 4 * c4;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (e[k] > best) { best = e[k]; bi = 4 * c4 + k; }
    for (int o = C4 >> 1; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (c4 == 0) {
        sm[pix * 2] = s / (float)C;
        sm[pix * 2 + 1] = best;
        amax[pix] = bi;
    }
}

constexpr int SP_T = 16;          // spatial conv tile
constexpr int SP_H = SP_T + 6;    // with 7x7 halo

// sa = sigmoid(conv7x7(sm)); grid (tilesX*tilesY, N)
__global__ __launch_bounds__(256) void cbam_sa_kernel(const float* __restrict__ sm, const float* __restrict__ w7,
                                                      int H, int W, int tilesX, float* __restrict__ sa) {
    __shared__ float tile[2][SP_H][SP_H + 1];
    __shared__ float wl[98];
    const int n = blockIdx.y;
    const int ty0 = (blockIdx.x / tilesX) * SP_T, tx0 = (blockIdx.x % tilesX) * SP_T;
    if (threadIdx.x < 98) wl[threadIdx.x] = w7[threadIdx.x];
    for (int i = threadIdx.x; i < SP_H * SP_H; i += 256) {
        const int hy = i / SP_H, hx = i % SP_H;
        const int gy = ty0 + hy - 3, gx = tx0 + hx - 3;
        float a = 0.f, b = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t pix = (size_t)(n * H + gy) * W + gx;
            a = sm[pix * 2];
            b = sm[pix * 2 + 1];
        }
        tile[0][hy][hx] = a;
        tile[1][hy][hx] = b;
    }
    __syncthreads();
    const int py = threadIdx.x / SP_T, px = threadIdx.x % SP_T;
    const int gy = ty0 + py, gx = tx0 + px;
    if (gy >= H || gx >= W) return;
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) s += wl[(ch * 7 + ky) * 7 + kx] * tile[ch][py + ky][px + kx];
    sa[(size_t)(n * H + gy) * W + gx] = sigmoidf_(s);
}

template <bool XB>
__global__ __launch_bounds__(256) void cbam_apply_kernel(const float* __restrict__ x, int x_ld,
                                                         const float* __restrict__ ca, const float* __restrict__ sa,
                                                         int C, long HW, float* __restrict__ out, int out_ld,
                                                         int out_coff, int out_bf16, long total) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const int C4 = C >> 2;
    const long pix = idiv(gid, C4, total);
    const int c4 = (int)(gid - pix * C4);
    const int n = (int)idiv(pix, HW, total);
    const float4 v = ldx4(x, pix * x_ld + 4 * c4, XB);
    const float4 a = ld4(ca + (size_t)n * C + 4 * c4);
    const float s = sa[pix];
    const float4 o = make_float4(v.x * a.x * s, v.y * a.y * s, v.z * a.z * s, v.w * a.w * s);
    if (out_bf16) {
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(out) + pix * out_ld + out_coff + 4 * c4) =
            (bf16x4_t){(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
    } else {
        st4(out + pix * out_ld + out_coff + 4 * c4, o);
    }
}

// ---------------------------------------------------------------- CBAM backward
template <bool XB>                                            // XB: dout and x are stored as bf16
__global__ __launch_bounds__(256) void cbam_bwd_pre_kernel(const float* __restrict__ dout, int dout_ld, int dout_coff,
                                                           const float* __restrict__ x, int x_ld,
                                                           const float* __restrict__ ca, const float* __restrict__ sa,
                                                           int C, long HW, float* __restrict__ dpre, long total) {
    const long gid = blockIdx.x * 256L + threadIdx.x;
    if (gid >= total) return;
    const int C4 = C >> 2;
    const long pix = idiv(gid, C4, total);
    const int c4 = (int)(gid - pix * C4);
    const int n = (int)idiv(pix, HW, total);
    const float4 g = ldx4(dout, pix * dout_ld + dout_coff + 4 * c4, XB);
    const float4 v = ldx4(x, pix * x_ld + 4 * c4, XB);
    const float4 a = ld4(ca + (size_t)n * C + 4 * c4);
    const float d = group_sum(g.x * v.x * a.x + g.y * v.y * a.y + g.z * v.z * a.z + g.w * v.w * a.w, C4);
    if (c4 == 0) {
        const float s = sa[pix];
        dpre[pix] = d * s * (1.f - s);
    }
}

// dsm = conv7x7^T(dpre); per-tile partial of dw7. grid (tiles, N); part[(n*tiles + tile)][98]
__global__ __launch_bounds__(256) void cbam_bwd_conv_kernel(const float* __restrict__ dpre,
                                                            const float* __restrict__ sm,
                                                            const float* __restrict__ w7, int H, int W, int tilesX,
                                                            float* __restrict__ dsm, float* __restrict__ part) {
    __shared__ float dtile[SP_H][SP_H + 1];      // dpre with halo
    __shared__ float stile[2][SP_H][SP_H + 1];   // sm with halo
    __shared__ float wl[98];
    const int n = blockIdx.y;
    const int ty0 = (blockIdx.x / tilesX) * SP_T, tx0 = (blockIdx.x % tilesX) * SP_T;
    if (threadIdx.x < 98) wl[threadIdx.x] = w7[threadIdx.x];
    for (int i = threadIdx.x; i < SP_H * SP_H; i += 256) {
        const int hy = i / SP_H, hx = i % SP_H;
        const int gy = ty0 + hy - 3, gx = tx0 + hx - 3;
        float d = 0.f, a = 0.f, b = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t pix = (size_t)(n * H + gy) * W + gx;
            d = dpre[pix];
            a = sm[pix * 2];
            b = sm[pix * 2 + 1];
        }
        dtile[hy][hx] = d;
        stile[0][hy][hx] = a;
        stile[1][hy][hx] = b;
    }
    __syncthreads();
    const int py = threadIdx.x / SP_T, px = threadIdx.x % SP_T;
    const int gy = ty0 + py, gx = tx0 + px;
    if (gy < H && gx < W) {
        // dsm[q][ch] = sum_k w[ch][k] * dpre[q - (k - 3)]
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float d = dtile[py + 6 - ky][px + 6 - kx];
                s0 += wl[ky * 7 + kx] * d;
                s1 += wl[49 + ky * 7 + kx] * d;
            }
        const size_t pix = (size_t)(n * H + gy) * W + gx;
        dsm[pix * 2] = s0;
        dsm[pix * 2 + 1] = s1;
    }
    // dw7[ch][ky][kx] partial = sum_{p in tile} sm[p + (ky-3,kx-3)][ch] * dpre[p]
    if (threadIdx.x < 98) {
        const int ch = threadIdx.x / 49, ky = (threadIdx.x % 49) / 7, kx = threadIdx.x % 7;
        float s = 0.f;
        for (int yy = 0; yy < SP_T; ++yy)
            for (int xx = 0; xx < SP_T; ++xx) s += stile[ch][yy + ky][xx + kx] * dtile[yy + 3][xx + 3];
        part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 98 + threadIdx.x] = s;
    }
}

// grid (nblk, N).  XB: dout, x and dx are stored as bf16
template <bool XB>
__global__ __launch_bounds__(256) void cbam_bwd_scale_kernel(const float* __restrict__ dout, int dout_ld, int dout_coff,
                                                             const float* __restrict__ x, int x_ld,
                                                             const float* __restrict__ ca, const float* __restrict__ sa,
                                                             const float* __restrict__ dsm, const int* __restrict__ amax,
                                                             int C, long HW, float* __restrict__ dx, int dx_ld,
                                                             float* __restrict__ dca_partial) {
    __shared__ float4 buf[256];
    const int C4 = C >> 2;
    const int npl = 256 / C4;
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
    const int n = blockIdx.y;
    const long base = (long)n * HW;
    const float4 a = ld4(ca + (size_t)n * C + 4 * c4);
    const float invC = 1.f / (float)C;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long p = (long)blockIdx.x * npl + pl; p < HW; p += (long)gridDim.x * npl) {
        const long pix = base + p;
        const float4 g = ldx4(dout, pix * dout_ld + dout_coff + 4 * c4, XB);
        const float4 v = ldx4(x, pix * x_ld + 4 * c4, XB);
        const float s = sa[pix];
        const float d0 = dsm[pix * 2] * invC, d1 = dsm[pix * 2 + 1];
        const int am = amax[pix] - 4 * c4;
        float4 dxc;
        dxc.x = g.x * s + d0 + (am == 0 ? d1 : 0.f);
        dxc.y = g.y * s + d0 + (am == 1 ? d1 : 0.f);
        dxc.z = g.z * s + d0 + (am == 2 ? d1 : 0.f);
        dxc.w = g.w * s + d0 + (am == 3 ? d1 : 0.f);
        stx4(dx, pix * dx_ld + 4 * c4, XB, make_float4(dxc.x * a.x, dxc.y * a.y, dxc.z * a.z, dxc.w * a.w));
        acc.x += dxc.x * v.x; acc.y += dxc.y * v.y; acc.z += dxc.z * v.z; acc.w += dxc.w * v.w;
    }
    const float4 r = plane_reduce4b(acc, buf, C4, npl);
    if ((int)threadIdx.x < C4)
        st4(dca_partial + ((size_t)n * gridDim.x + blockIdx.x) * C + 4 * c4, r);
}

// single block; loops over images so that dw1/dw2 are summed in a fixed order
__global__ __launch_bounds__(256) void cbam_bwd_channel_kernel(const float* __restrict__ dca_partial, int nblk, int C,
                                                               int R, int N, long HW, const float* __restrict__ w1,
                                                               const float* __restrict__ w2, const float* __restrict__ gap,
                                                               const float* __restrict__ hid, const float* __restrict__ ca,
                                                               float* __restrict__ dw1, float* __restrict__ dw2,
                                                               float* __restrict__ dgap_pix, int accumulate) {
    __shared__ float dz2[256];
    __shared__ float dz1[16];
    __shared__ float sh[16];
    const int c = threadIdx.x;
    float a1[16], a2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a1[r] = 0.f; a2[r] = 0.f; }
    __shared__ double sp[256];
    const int parts = 256 / C;                       // threads per channel for the partial sums (fixed order => deterministic)
    for (int n = 0; n < N; ++n) {
        __syncthreads();
        {
            const int cc = c % C, part = c / C;
            double s = 0.0;
            if (part < parts) {
                const float* p = dca_partial + (size_t)n * nblk * C + cc;
                for (int b = part; b < nblk; b += parts) s += (double)p[(size_t)b * C];
            }
            sp[c] = s;
        }
        __syncthreads();
        float z = 0.f;
        if (c < C) {
            double s = 0.0;
            for (int k = 0; k < parts; ++k) s += sp[k * C + c];
            const float cav = ca[(size_t)n * C + c];
            z = (float)s * cav * (1.f - cav);
        }
        dz2[c] = z;
        if (c < R) sh[c] = hid[(size_t)n * R + c];
        __syncthreads();
        if (c < R) {
            float s = 0.f;
            for (int k = 0; k < C; ++k) s += w2[k * R + c] * dz2[k];
            dz1[c] = sh[c] > 0.f ? s : 0.f;
        }
        __syncthreads();
        if (c < C) {
            const float gv = gap[(size_t)n * C + c];
            float dg = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r < R) {
                    a2[r] += z * sh[r];          // dw2[c][r]
                    a1[r] += dz1[r] * gv;        // dw1[r][c]
                    dg += w1[r * C + c] * dz1[r];
                }
            dgap_pix[(size_t)n * C + c] = dg / (float)HW;
        }
    }
    if (c < C) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (r < R) {
                dw2[c * R + r] = accumulate ? dw2[c * R + r] + a2[r] : a2[r];
                dw1[r * C + c] = accumulate ? dw1[r * C + c] + a1[r] : a1[r];
            }
    }
}

static bool pow2_c4(int C) {
    const int c4 = C >> 2;
    return C % 4 == 0 && c4 >= 1 && c4 <= 64 && (c4 & (c4 - 1)) == 0;
}

}  // namespace nvq

using namespace nvq;

extern "C" {

int nvq_tsum_blocks(int H, int W) { return tsum_blocks_host(H, W); }

int nvq_tsum_forward(const float* aligned, int aligned_ld, const float* logits, int logits_ld, int T, int C,
                     int N, int H, int W, float* attn, int attn_ld, float* weighted, int weighted_ld,
                     float* gap_partial, int aligned_bf16, int weighted_bf16, void* stream) {
    NVQ_REQUIRE(pow2_c4(C), "tsum_forward: C %d must be a power of two in [4,256]", C);
    NVQ_REQUIRE(T >= 1 && T <= NVQ_MAX_T && logits_ld >= T && attn_ld >= T, "tsum_forward: T %d", T);
    NVQ_REQUIRE(aligned_ld % 4 == 0 && weighted_ld % 4 == 0 && aligned_ld >= T * C, "tsum_forward: ld");
    const dim3 grid(tsum_blocks_host(H, W), N);
#define NVQ_TS(M_, U_, A_, W_) hipLaunchKernelGGL((tsum_fwd_kernel<M_, U_, A_, W_>), grid, dim3(256), 0, (hipStream_t)stream, aligned, \
                                                  aligned_ld, logits, logits_ld, T, C, (long)H * W, attn, attn_ld, weighted, weighted_ld, \
                                                  gap_partial)
#define NVQ_TS2(M_, U_) do { if (aligned_bf16) { if (weighted_bf16) NVQ_TS(M_, U_, true, true); else NVQ_TS(M_, U_, true, false); } \
                             else { if (weighted_bf16) NVQ_TS(M_, U_, false, true); else NVQ_TS(M_, U_, false, false); } } while (0)
    if (T <= 4) NVQ_TS2(4, 4); else NVQ_TS2(NVQ_MAX_T, 2);
#undef NVQ_TS2
#undef NVQ_TS
    return check_launch("tsum_forward");
}

int nvq_tsum_backward(const float* dweighted, int dweighted_ld, const float* dgap_pix, const float* aligned,
                      int aligned_ld, const float* attn, int attn_ld, int T, int C, int N, int H, int W,
                      float* daligned, int daligned_ld, float* dlogits, int dlogits_ld, int aligned_bf16,
                      int daligned_bf16, int dweighted_bf16, void* stream) {
    NVQ_REQUIRE(pow2_c4(C), "tsum_backward: C %d must be a power of two in [4,256]", C);
    NVQ_REQUIRE(T >= 1 && T <= NVQ_MAX_T && dlogits_ld >= T && attn_ld >= T, "tsum_backward: T %d", T);
    NVQ_REQUIRE(aligned_ld % 4 == 0 && dweighted_ld % 4 == 0 && daligned_ld % 4 == 0, "tsum_backward: ld");
    const long total = (long)N * H * W * (C / 4);
#define NVQ_TB(A_, W_) hipLaunchKernelGGL((tsum_bwd_kernel<A_, W_>), dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,  \
                                          dweighted, dweighted_ld, dgap_pix, aligned, aligned_ld, attn, attn_ld, T, C, (long)H * W, \
                                          daligned, daligned_ld, dlogits, dlogits_ld, total, aligned_bf16, daligned_bf16)
    if (aligned_bf16 && daligned_bf16) { if (dweighted_bf16) NVQ_TB(true, true); else NVQ_TB(true, false); }
    else { if (dweighted_bf16) NVQ_TB(false, true); else NVQ_TB(false, false); }
#undef NVQ_TB
    return check_launch("tsum_backward");
}

int nvq_cbam_channel(const float* gap_partial, int nblk, int C, int R, int N, int HW, const float* w1,
                     const float* w2, float* gap, float* hid, float* ca, void* stream) {
    NVQ_REQUIRE(C <= 256 && R >= 1 && R <= 16, "cbam_channel: C %d R %d", C, R);
    hipLaunchKernelGGL(cbam_channel_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, gap_partial, nblk, C, R,
                       (long)HW, w1, w2, gap, hid, ca);
    return check_launch("cbam_channel");
}

int nvq_cbam_pool(const float* x, int x_ld, const float* ca, int C, int N, int H, int W, float* sm, int* amax,
                  int x_bf16, void* stream) {
    NVQ_REQUIRE(pow2_c4(C) && x_ld % 4 == 0, "cbam_pool: C %d must be a power of two in [4,256]", C);
    const long total = (long)N * H * W * (C / 4);
    if (x_bf16)
        hipLaunchKernelGGL(cbam_pool_kernel<true>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_ld, ca, C,
                           (long)H * W, sm, amax, total);
    else
        hipLaunchKernelGGL(cbam_pool_kernel<false>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_ld, ca, C,
                           (long)H * W, sm, amax, total);
    return check_launch("cbam_pool");
}

int nvq_cbam_spatial_apply(const float* x, int x_ld, const float* ca, const float* sm, const float* w7, int C,
                           int N, int H, int W, float* sa, float* out, int out_ld, int out_coff, int out_bf16,
                           int x_bf16, void* stream) {
    NVQ_REQUIRE(C % 4 == 0 && x_ld % 4 == 0 && out_ld % 4 == 0 && out_coff % 4 == 0, "cbam_spatial_apply: alignment");
    const int tilesX = (W + SP_T - 1) / SP_T, tilesY = (H + SP_T - 1) / SP_T;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cbam_sa_kernel, dim3(tilesX * tilesY, N), dim3(256), 0, s, sm, w7, H, W, tilesX, sa);
    int rc = check_launch("cbam_sa");
    if (rc) return rc;
    const long total = (long)N * H * W * (C / 4);
    if (x_bf16)
        hipLaunchKernelGGL(cbam_apply_kernel<true>, dim3(ceil_div(total, 256)), dim3(256), 0, s, x, x_ld, ca, sa, C, (long)H * W,
                           out, out_ld, out_coff, out_bf16, total);
    else
        hipLaunchKernelGGL(cbam_apply_kernel<false>, dim3(ceil_div(total, 256)), dim3(256), 0, s, x, x_ld, ca, sa, C, (long)H * W,
                           out, out_ld, out_coff, out_bf16, total);
    return check_launch("cbam_apply");
}

int nvq_cbam_bwd_spatial_pre(const float* dout, int dout_ld, int dout_coff, const float* x, int x_ld,
                             const float* ca, const float* sa, int C, int N, int H, int W, float* dpre,
                             int x_bf16, void* stream) {
    NVQ_REQUIRE(pow2_c4(C) && x_ld % 4 == 0 && dout_ld % 4 == 0 && dout_coff % 4 == 0,
                "cbam_bwd_spatial_pre: C %d must be a power of two in [4,256]", C);
    const long total = (long)N * H * W * (C / 4);
    if (x_bf16)
        hipLaunchKernelGGL(cbam_bwd_pre_kernel<true>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, dout_ld,
                           dout_coff, x, x_ld, ca, sa, C, (long)H * W, dpre, total);
    else
        hipLaunchKernelGGL(cbam_bwd_pre_kernel<false>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, dout_ld,
                           dout_coff, x, x_ld, ca, sa, C, (long)H * W, dpre, total);
    return check_launch("cbam_bwd_spatial_pre");
}

int nvq_cbam_bwd_spatial_conv(const float* dpre, const float* sm, const float* w7, int N, int H, int W, float* dsm,
                              float* dw7, float* workspace, size_t workspace_bytes, int accumulate, void* stream) {
    const int tilesX = (W + SP_T - 1) / SP_T, tilesY = (H + SP_T - 1) / SP_T;
    const int nblk = tilesX * tilesY * N;
    if ((size_t)nblk * 98 * sizeof(float) > workspace_bytes) { set_error("cbam_bwd_spatial_conv: workspace"); return NVQ_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cbam_bwd_conv_kernel, dim3(tilesX * tilesY, N), dim3(256), 0, s, dpre, sm, w7, H, W, tilesX, dsm,
                       workspace);
    int rc = check_launch("cbam_bwd_spatial_conv");
    if (rc) return rc;
    return launch_reduce_partials(workspace, nblk, 98, 1.f, dw7, accumulate, s);
}

int nvq_cbam_bwd_scale(const float* dout, int dout_ld, int dout_coff, const float* x, int x_ld, const float* ca,
                       const float* sa, const float* dsm, const int* amax, int C, int N, int H, int W, float* dx,
                       int dx_ld, float* dca_partial, int x_bf16, void* stream) {
    NVQ_REQUIRE(pow2_c4(C) && x_ld % 4 == 0 && dout_ld % 4 == 0 && dout_coff % 4 == 0 && dx_ld % 4 == 0,
                "cbam_bwd_scale: C %d must be a power of two in [4,256]", C);
    if (x_bf16)
        hipLaunchKernelGGL(cbam_bwd_scale_kernel<true>, dim3(tsum_blocks_host(H, W), N), dim3(256), 0, (hipStream_t)stream, dout,
                           dout_ld, dout_coff, x, x_ld, ca, sa, dsm, amax, C, (long)H * W, dx, dx_ld, dca_partial);
    else
        hipLaunchKernelGGL(cbam_bwd_scale_kernel<false>, dim3(tsum_blocks_host(H, W), N), dim3(256), 0, (hipStream_t)stream, dout,
                           dout_ld, dout_coff, x, x_ld, ca, sa, dsm, amax, C, (long)H * W, dx, dx_ld, dca_partial);
    return check_launch("cbam_bwd_scale");
}

int nvq_cbam_bwd_channel(const float* dca_partial, int nblk, int C, int R, int N, int HW, const float* w1,
                         const float* w2, const float* gap, const float* hid, const float* ca, float* dw1,
                         float* dw2, float* dgap_pix, int accumulate, void* stream) {
    NVQ_REQUIRE(C <= 256 && R >= 1 && R <= 16, "cbam_bwd_channel: C %d R %d", C, R);
    hipLaunchKernelGGL(cbam_bwd_channel_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dca_partial, nblk, C, R, N,
                       (long)HW, w1, w2, gap, hid, ca, dw1, dw2, dgap_pix, accumulate);
    return check_launch("cbam_bwd_channel");
}

}  // extern "C"
