// Implicit-GEMM convolution on the gfx950 matrix cores, fp32 NHWC activations.
//
//   forward / input-gradient : D[cout][pixel] += W[cout][k] * X[k][pixel]      (M = cout, N = pixels)
//   weight-gradient          : D[cin][cout]   += X[pixel][cin]^T * dY[pixel][cout]  (K = pixels)
//
// NVQ_MATH_F32 uses v_mfma_f32_16x16x4_f32 (exact fp32, bit-equal to an fmaf chain).
// One workgroup = 256 threads = 4 waves = an 8 x 32 pixel tile; the halo tile of the current
// 16-channel K chunk and the matching weight slab are staged in LDS, every wave owns two tile
// rows (4 pixel blocks of 16) and all NB*16 output channels of the workgroup.
#include "conv_common.h"

namespace nvq {

constexpr int KC = 16;     // input channels per K chunk
constexpr int XS_LD = 20;  // floats per staged pixel (16 + 4 pad: the 16 pixels of one B-fragment hit 16 distinct 16-B slots)

// ---------------------------------------------------------------- weight packing
// wpack[cz][kc][tap][g][n][j]  (g = 0..3 lane group, n = 0..NT-1, j = 0..3) holds
// W[cout = cz*NT + n][channel = kc*16 + 4g + j][tap]; zero outside the real extents.
__device__ __forceinline__ float pack_f32_elem(const float* __restrict__ w, int cout_w, int cin_w, int taps, int transpose,
                                               int cout_keep, int NT, int nkc, long idx) {
    long t = idx;
    const int j = t & 3; t >>= 2;
    const int n = t % NT; t /= NT;
    const int g = t & 3; t >>= 2;
    const int tap = t % taps; t /= taps;
    const int kc = t % nkc; t /= nkc;
    const int cz = (int)t;
    const int co = cz * NT + n;
    const int ch = kc * KC + 4 * g + j;
    float v = 0.f;
    if (!transpose) {
        if (co < cout_w && ch < cin_w) v = w[((long)co * cin_w + ch) * taps + tap];
    } else {
        if (co < cout_keep && ch < cout_w) v = w[((long)ch * cin_w + co) * taps + (taps - 1 - tap)];
    }
    return v;
}

__global__ void pack_kernel(const float* __restrict__ w, int cout_w, int cin_w, int taps,
                            int transpose, int cout_keep, int NT, int ncz, int nkc,
                            float* __restrict__ wp) {
    const long total = (long)ncz * nkc * taps * 4 * NT * 4;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x)
        wp[idx] = pack_f32_elem(w, cout_w, cin_w, taps, transpose, cout_keep, NT, nkc, idx);
}

// grid (blocks, jobs): nvq_conv_pack_batch
__global__ void pack_batch_kernel(const PackJobTable t) {
    const PackJobDev j = t.j[blockIdx.y];
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < j.total; idx += (long)gridDim.x * blockDim.x)
        j.wp[idx] = pack_f32_elem(j.w, j.cout_w, j.cin_w, j.taps, j.transpose, j.cout_keep, j.NT, j.nkc, idx);
}

// ---------------------------------------------------------------- dense-block backward weights
// The gradient w.r.t. growth slice y_i of a residual dense block is a sum over every later consumer:
//   d y_i = mask_i * ( 0.2 * lff^T(gout)[y_i]  +  sum_{j>i} conv_j^T(d y_j)[y_i] )
// With the gradient buffer laid out [gout(F) | dy_4 | dy_3 | dy_2 | dy_1 | dy_0] this is ONE 3x3 convolution
// over the channel prefix [0, F + 32(4-i)) (the 1x1 lff term sits on the centre tap) - the mirror image of
// the forward dense layer, with no read-modify-write accumulation.  This kernel assembles those combined
// weights in PyTorch layout: targets t = 0..4 -> Wb_{4-t} [32][F+32t][3][3], then Wb_x [F][F+160][3][3].
struct RdbSrc { const float* lff; const float* w[5]; };

__global__ void rdb_bwd_weights_kernel(RdbSrc src, int F, float* __restrict__ out, long total) {
    const int CAT = F + 160;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long rem = idx;
        int i = -1, cinb = 0, cout = 0;          // i = growth layer whose gradient this target produces (-1: x)
        for (int t = 0; t < 5; ++t) {
            const long sz = 32L * (F + 32 * t) * 9;
            if (rem < sz) { i = 4 - t; cinb = F + 32 * t; cout = 32; break; }
            rem -= sz;
        }
        if (i < 0) { cinb = CAT; cout = F; }
        const int tap = rem % 9;
        const int c = (rem / 9) % cinb;
        const int o = (int)(rem / (9L * cinb));
        (void)cout;
        const int ych = i >= 0 ? F + 32 * i + o : o;      // channel of the forward concat this output refers to
        float v = 0.f;
        if (c < F) {
            if (tap == 4) v = 0.2f * src.lff[(long)c * CAT + ych];
        } else {
            const int j = 4 - (c - F) / 32, m = (c - F) % 32;
            v = src.w[j][((long)m * (F + 32 * j) + ych) * 9 + (8 - tap)];
        }
        out[idx] = v;
    }
}

// ---------------------------------------------------------------- forward / dgrad
// GS > 1 (small launches: fewer workgroups than the device holds): blockIdx.z = one of GS parts of the packed slab's NT * GS
// output channels, so that a launch of few tiles still puts a workgroup on every CU - each one re-stages the activation tile
// and does 1 / GS of the MFMAs.
template <int NB, int KS, int GS = 1>
__global__ __launch_bounds__(256, 2) void conv_f32_kernel(const nvq_conv_desc d, int tilesX,
                                                           int tilesY, int nkc, int vec_ok) {
    constexpr int NT = NB * 16;                               // output channels of this workgroup
    constexpr int NTW = NT * GS;                              // ... of the packed slab
    constexpr int HALO = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int HW_ = TW + 2 * HALO;
    constexpr int HH_ = TH + 2 * HALO;
    constexpr int NPIX = HW_ * HH_;
    constexpr int WS_FLOATS = TAPS * 4 * NT * 4;
    __shared__ __attribute__((aligned(16))) float lds[NPIX * XS_LD + WS_FLOATS];
    float* xs = lds;
    float* ws = lds + NPIX * XS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = lane & 15;
    const int g = lane >> 4;

    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int part = GS == 1 ? 0 : blockIdx.z;
    const int cz = blockIdx.y * GS + part;                    // in units of NT channels (epilogue)
    const int H = d.h, W = d.w;

    f32x4 acc[NB][4];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* wp_base = d.wpack + (size_t)blockIdx.y * nkc * (WS_FLOATS * GS);
    const float* in = d.in + d.in_coff;

    // Staging through registers, one chunk ahead (the scheme of the bf16 kernels): chunk kc + 1 is fetched while chunk kc is
    // multiplied, every load unconditional (out-of-image pieces read element 0 and are zeroed when committed) and all of a
    // thread's loads in flight at once.  The previous form - load a piece, store it to LDS, next piece - was one memory round
    // trip per piece, 10 .. 15 in a row per chunk, hidden only by the CU's second workgroup (and not at all on small frames,
    // where a CU holds one workgroup or none).
    constexpr int XITEMS = NPIX * 4;                          // (pixel, 4-channel group) pieces per chunk
    constexpr int XPER = (XITEMS + 255) / 256;
    constexpr int WS4 = WS_FLOATS / 4;
    constexpr int WPER = (WS4 + 255) / 256;
    unsigned xoff[XPER];
    unsigned xokm = 0;
#pragma unroll
    for (int k = 0; k < XPER; ++k) {
        const int item = tid + k * 256;
        const int hp = item >> 2;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int gy = ty * TH + hy - HALO, gx = tx * TW + hx - HALO;
        const bool ok = item < XITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        xokm |= (ok ? 1u : 0u) << k;
        xoff[k] = ok ? (unsigned)(((size_t)(n * H + gy) * W + gx) * d.in_ld) : 0u;   // (< 2^32 elements: checked on the host)
    }
    const int chq = 4 * (tid & 3);                            // channel group of this thread's pieces (256 % 4 == 0)
    f32x4 xr[XPER], wr[WPER];                                 // (ext_vector types: HIP's float4 struct is copied around behind waits)
    bool cv = false;                                          // channel validity of the chunk held in xr
    auto fetch = [&](int kc) {                                // raw loads only
        const int ch = kc * KC + chq;
        cv = ch < d.cin;
        const unsigned o = cv ? (unsigned)ch : 0u;
#pragma unroll
        for (int k = 0; k < XPER; ++k) xr[k] = *reinterpret_cast<const f32x4*>(in + (xoff[k] + o));
        const float* wsrc = wp_base + (size_t)kc * (WS_FLOATS * GS);
#pragma unroll
        for (int k = 0; k < WPER; ++k) {
            const int i = tid + k * 256 < WS4 ? tid + k * 256 : 0;                      // (WS4 may be < 256)
            wr[k] = *reinterpret_cast<const f32x4*>(wsrc + 4 * (GS == 1 ? i : (i / NT) * NTW + part * NT + i % NT));   // this part's NT columns
        }
    };
    auto commit = [&]() {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid + k * 256;
            if (item < XITEMS)
                *reinterpret_cast<f32x4*>(xs + (item >> 2) * XS_LD + 4 * (item & 3)) = ((xokm >> k) & 1) && cv ? xr[k] : z;
        }
#pragma unroll
        for (int k = 0; k < WPER; ++k)
            if ((k + 1) * 256 <= WS4 || tid + k * 256 < WS4) *reinterpret_cast<f32x4*>(ws + 4 * (tid + k * 256)) = wr[k];
    };

    fetch(0);
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();
        commit();
        __syncthreads();
        if (kc + 1 < nkc) fetch(kc + 1);

#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = tap / KS, dx = tap - dy * KS;
            float4 xb[4];
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int row = 2 * wave + (pb >> 1);
                const int x0 = (pb & 1) * 16;
                const int hp = (row + dy) * HW_ + x0 + c + dx;
                xb[pb] = ld4(xs + hp * XS_LD + 4 * g);
            }
            float4 wa[NB];
#pragma unroll
            for (int cb = 0; cb < NB; ++cb)
                wa[cb] = ld4(ws + ((tap * 4 + g) * NT + cb * 16 + c) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    const float a = j == 0 ? wa[cb].x : j == 1 ? wa[cb].y : j == 2 ? wa[cb].z : wa[cb].w;
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) {
                        const float b = j == 0 ? xb[pb].x : j == 1 ? xb[pb].y : j == 2 ? xb[pb].z : xb[pb].w;
                        acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[cb][pb], 0, 0, 0);
                    }
                }
            }
        }
    }

    conv_epilogue<NB>(d, acc, n, ty, tx, cz, wave, c, g, vec_ok);
}

// ---------------------------------------------------------------- weight gradient
// grid (split, ci chunk of 32, co chunk of 32). Wave (cib, cob) owns the 16x16 block of
// (ci, co) for all taps; K = the 256 pixels of a tile, 4 per MFMA.  LDS images are
// [pixel][32 ch] with channel ^= (pixel & 1) << 4 so the two pixels read by one 32-lane
// half land on disjoint banks.
template <int KS>
__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(const nvq_wgrad_desc d, int tilesX,
                                                            int tilesY, int ntiles, int nci,
                                                            int nco) {
    constexpr int HALO = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int HW_ = TW + 2 * HALO;
    constexpr int HH_ = TH + 2 * HALO;
    constexpr int NPIX = HW_ * HH_;
    __shared__ __attribute__((aligned(16))) float lds[NPIX * WG_C + TH * TW * WG_C];
    float* xs = lds;
    float* dys = lds + NPIX * WG_C;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 15;
    const int g = lane >> 4;
    const int cib = wave >> 1, cob = wave & 1;
    const int cic = blockIdx.y, coc = blockIdx.z;
    const int H = d.h, W = d.w;
    const float* x = d.x + d.x_coff;
    const float* dy = d.dy + d.dy_coff;

    f32x4 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);   // column sums of dy, channels 4*(tid&7)..+3 (bias gradient)

    // Staging through registers, one tile ahead, every load unconditional and all of them in flight at once (see
    // conv_f32_kernel); dy pieces only when whole 4-channel pieces are the rule (cout % 4 == 0), else they are loaded
    // element-wise when the tile is committed.
    constexpr int XITEMS = NPIX * 8;
    constexpr int XPER = (XITEMS + 255) / 256, YPER = TH * TW * 8 / 256;
    f32x4 xr[XPER], yr[YPER];                                 // (ext_vector types, see conv_f32_kernel)
    unsigned xm = 0, ym = 0;
    const bool yvec = d.cout % 4 == 0;                        // uniform
    const int xch = cic * WG_C + 4 * (tid & 7), ych = coc * WG_C + 4 * (tid & 7);   // (256 % 8 == 0)
    auto fetch = [&](int tile) {
        int bt = xcd_tile(tile, ntiles);
        const int tx = bt % tilesX; bt /= tilesX;
        const int ty = bt % tilesY;
        const int n = bt / tilesY;
        xm = 0; ym = 0;
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid + k * 256;
            const int hp = item >> 3;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int gy = ty * TH + hy - HALO, gx = tx * TW + hx - HALO;
            const bool ok = item < XITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W && xch < d.cin;
            xm |= (ok ? 1u : 0u) << k;
            xr[k] = *reinterpret_cast<const f32x4*>(x + (ok ? ((size_t)(n * H + gy) * W + gx) * d.x_ld + xch : 0));
        }
        if (yvec) {
#pragma unroll
            for (int k = 0; k < YPER; ++k) {
                const int p = (tid + k * 256) >> 3;
                const int py = p / TW, px = p - py * TW;
                const int gy = ty * TH + py, gx = tx * TW + px;
                const bool ok = gy < H && gx < W && ych < d.cout;
                ym |= (ok ? 1u : 0u) << k;
                yr[k] = *reinterpret_cast<const f32x4*>(dy + (ok ? ((size_t)(n * H + gy) * W + gx) * d.dy_ld + ych : 0));
            }
        }
    };
    auto commit = [&](int tile) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid + k * 256;
            const int hp = item >> 3, q = item & 7;
            if (item < XITEMS) *reinterpret_cast<f32x4*>(xs + hp * WG_C + ((4 * q) ^ ((hp & 1) << 4))) = (xm >> k) & 1 ? xr[k] : z;
        }
        if (yvec) {
#pragma unroll
            for (int k = 0; k < YPER; ++k) {
                const int item = tid + k * 256;
                const int p = item >> 3, q = item & 7;
                const f32x4 v = (ym >> k) & 1 ? yr[k] : z;
                *reinterpret_cast<f32x4*>(dys + p * WG_C + ((4 * q) ^ ((p & 1) << 4))) = v;
                bsum.x += v[0]; bsum.y += v[1]; bsum.z += v[2]; bsum.w += v[3];
            }
            return;
        }
        int bt = xcd_tile(tile, ntiles);
        const int tx = bt % tilesX; bt /= tilesX;
        const int ty = bt % tilesY;
        const int n = bt / tilesY;
        for (int item = tid; item < TH * TW * 8; item += 256) {
            const int p = item >> 3, q = item & 7;
            const int py = p / TW, px = p - py * TW;
            const int gy = ty * TH + py, gx = tx * TW + px;
            const int ch = coc * WG_C + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < H && gx < W) {
                const float* src = dy + ((size_t)(n * H + gy) * W + gx) * d.dy_ld + ch;
                if (ch + 3 < d.cout) {
                    v = ld4(src);
                } else {
                    if (ch < d.cout) v.x = src[0];
                    if (ch + 1 < d.cout) v.y = src[1];
                    if (ch + 2 < d.cout) v.z = src[2];
                }
            }
            st4(dys + p * WG_C + ((4 * q) ^ ((p & 1) << 4)), v);
            bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        commit(tile);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);

#pragma unroll 2
        for (int ks = 0; ks < TH * TW / 4; ++ks) {
            const int p = 4 * ks + g;
            const int py = p / TW, px = p - py * TW;
            const float b = dys[p * WG_C + ((cob * 16 + r) ^ ((p & 1) << 4))];
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int ddy = tap / KS, ddx = tap - ddy * KS;
                const int hp = (py + ddy) * HW_ + px + ddx;
                const float a = xs[hp * WG_C + ((cib * 16 + r) ^ ((hp & 1) << 4))];
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[tap], 0, 0, 0);
            }
        }
    }

    float* part = d.workspace +
                  ((size_t)(blockIdx.x * nci + cic) * nco + coc) * (TAPS * WG_C * WG_C);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            part[(tap * WG_C + cib * 16 + 4 * g + e) * WG_C + cob * 16 + r] = acc[tap][e];
    wgrad_bias_partial(d, bsum, lds, nco, cic, coc);
}

// A 256-thread block owns WR_E consecutive elements of the partial layout (coalesced 128-byte reads); the nsplit slabs of an
// element are shared out over WR_Q threads (slab k goes to thread k % WR_Q), whose double sums meet in LDS in a fixed order -
// deterministic.  (One thread per element left a 192 -> 32 3x3 gradient with 216 workgroups and 170 dependent-latency-bound loads
// per thread: 15 us per launch, 61 launches per step.)  The PyTorch-layout write is scattered once.
constexpr int WR_Q = 8, WR_E = 256 / WR_Q;
// Blocks past `wblocks` do the bias gradient of the same launch (bp != NULL; one wave per output channel: bias_part[split][coc][32]
// summed over the splits in double) - a launch of its own cost as much as its arithmetic.
// bx: the block's index within its job (the launch's blockIdx.x)
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, int nsplit, int nci, int nco, int taps,
                                                  int cout, int cin_w, float alpha, int accumulate, float* __restrict__ dw,
                                                  int wblocks, const float* __restrict__ bp, float* __restrict__ dbias,
                                                  int bx, double (&sh)[WR_Q][WR_E]) {
    if (bx >= wblocks) {                                      // (block-uniform)
        const int co = (bx - wblocks) * 4 + (threadIdx.x >> 6);
        const int lane = threadIdx.x & 63;
        if (co >= cout) return;
        const int coc = co / WG_C, col = co % WG_C;
        double s = 0.0;
        for (int k = lane; k < nsplit; k += 64) s += (double)bp[((size_t)k * nco + coc) * WG_C + col];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) {
            const float v = alpha * (float)s;
            dbias[co] = accumulate ? dbias[co] + v : v;
        }
        return;
    }
    const long total = (long)nci * nco * taps * WG_C * WG_C;
    const int e = threadIdx.x & (WR_E - 1), q = threadIdx.x / WR_E;
    for (long base = (long)bx * WR_E; base < total; base += (long)wblocks * WR_E) {           // (uniform trip count per block)
        const long idx = base + e;
        // 4 independent chains: the split loop is a chain of dependent HBM/L2 loads otherwise
        double s4[4] = {0, 0, 0, 0};
        if (idx < total) {
            int k = q;
            for (; k + 3 * WR_Q < nsplit; k += 4 * WR_Q) {
#pragma unroll
                for (int u = 0; u < 4; ++u) s4[u] += (double)part[(size_t)(k + u * WR_Q) * total + idx];
            }
            for (; k < nsplit; k += WR_Q) s4[0] += (double)part[(size_t)k * total + idx];
        }
        sh[q][e] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        __syncthreads();
        if (q == 0 && idx < total) {
            double sum = 0.0;
#pragma unroll
            for (int u = 0; u < WR_Q; ++u) sum += sh[u][e];
            long t = idx;
            const int col = t % WG_C; t /= WG_C;
            const int cil = t % WG_C; t /= WG_C;
            const int tap = t % taps; t /= taps;
            const int coc = t % nco;
            const int cic = (int)(t / nco);
            const int ci = cic * WG_C + cil, co = coc * WG_C + col;
            if (ci < cin_w && co < cout) {
                const float v = alpha * (float)sum;
                const long o = ((long)co * cin_w + ci) * taps + tap;
                dw[o] = accumulate ? dw[o] + v : v;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, int nci, int nco,
                                                           int taps, int cout, int cin_w, float alpha, int accumulate,
                                                           float* __restrict__ dw, int wblocks, const float* __restrict__ bp,
                                                           float* __restrict__ dbias) {
    __shared__ double sh[WR_Q][WR_E];
    wgrad_reduce_body(part, nsplit, nci, nco, taps, cout, cin_w, alpha, accumulate, dw, wblocks, bp, dbias, (int)blockIdx.x, sh);
}

// Several layers' reduces in one launch (nvq_wgrad_reduce_batch): grid = (blocks, jobs); job y takes the first
// min(its own block count, gridDim.x) blocks of row y.  On launch-bound steps (the 64x64 continual-learning step: 61 reduces of
// 6 us, each a dependent launch between two weight-gradient kernels) a dense block's six reduces cost one launch.
constexpr int WR_MAX_JOBS = 16;
struct WgradReduceTable { nvq_wgrad_reduce_job j[WR_MAX_JOBS]; };
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradReduceTable t) {
    __shared__ double sh[WR_Q][WR_E];
    const nvq_wgrad_reduce_job& j = t.j[blockIdx.y];
    const long total = (long)j.nci * j.nco * j.taps * WG_C * WG_C;
    const int bblk = j.dbias ? (j.cout + 3) / 4 : 0;
    long nblk = (total + WR_E - 1) / WR_E;
    if (nblk > (long)gridDim.x - bblk) nblk = (long)gridDim.x - bblk;
    if ((int)blockIdx.x >= nblk + bblk) return;               // (block-uniform)
    wgrad_reduce_body(j.part, j.nsplit, j.nci, j.nco, j.taps, j.cout, j.cin_w, j.alpha, j.accumulate, j.dw, (int)nblk,
                      j.bias_part, j.dbias, (int)blockIdx.x, sh);
}

// (for kernels of other translation units that write partial slabs in this layout: pw_bwd.hip)
static int launch_wgrad_reduce_bias(const float* part, int nsplit, int nci, int nco, int taps, int cout, int cin_w, float alpha,
                                    int accumulate, float* dw, const float* bias_part, float* dbias, hipStream_t s) {
    const long total = (long)nci * nco * taps * WG_C * WG_C;
    int nblk = ceil_div(total, WR_E);
    if (nblk > 4096) nblk = 4096;
    const int bblk = dbias ? ceil_div(cout, 4) : 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk + bblk), dim3(256), 0, s, part, nsplit, nci, nco, taps, cout, cin_w, alpha,
                       accumulate, dw, nblk, bias_part, dbias);
    return check_launch("conv_wgrad_reduce");
}
int launch_wgrad_reduce(const float* part, int nsplit, int nci, int nco, int taps, int cout, int cin_w, float alpha,
                        int accumulate, float* dw, hipStream_t s) {
    return launch_wgrad_reduce_bias(part, nsplit, nci, nco, taps, cout, cin_w, alpha, accumulate, dw, nullptr, nullptr, s);
}

// ---------------------------------------------------------------- column sums (bias gradient)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ld, int coff,
                                                     int C, int CP, long npix,
                                                     float* __restrict__ part) {
    __shared__ float red[256];
    const int c = threadIdx.x % CP;
    const int pl = threadIdx.x / CP;
    const int lanes = 256 / CP;
    float s = 0.f;
    if (c < C)
        for (long p = (long)blockIdx.x * lanes + pl; p < npix; p += (long)gridDim.x * lanes)
            s += x[p * ld + coff + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && c < C) {
        float t = 0.f;
        for (int k = 0; k < lanes; ++k) t += red[k * CP + c];
        part[(long)blockIdx.x * C + c] = t;
    }
}

// one wave per output element: lanes stride over the partial rows, xor-shuffle tree in double
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nblk, int K,
                                                              float alpha, float* __restrict__ out, int accumulate) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= K) return;
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)part[(long)b * K + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        const float v = alpha * (float)s;
        out[k] = accumulate ? out[k] + v : v;
    }
}

int launch_reduce_partials(const float* part, int nblk, int K, float alpha, float* out,
                           int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ceil_div(K, 4)), dim3(256), 0, s, part, nblk,
                       K, alpha, out, accumulate);
    return check_launch("reduce_partials");
}

static int colsum_impl(const float* x, int ld, int coff, int C, long npix, float alpha, float* out,
                       float* ws, size_t ws_bytes, int accumulate, hipStream_t s) {
    NVQ_REQUIRE(C >= 1 && C <= 256, "colsum: C=%d out of range", C);
    int CP = 1;
    while (CP < C) CP <<= 1;
    const int lanes = 256 / CP;
    int nblk = ceil_div(npix, (long)lanes * 8);
    if (nblk > 512) nblk = 512;
    if (nblk < 1) nblk = 1;
    if ((size_t)nblk * C * sizeof(float) > ws_bytes) {
        set_error("colsum: workspace too small");
        return NVQ_EWORKSPACE;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3(nblk), dim3(256), 0, s, x, ld, coff, C, CP, npix, ws);
    int rc = check_launch("colsum");
    if (rc) return rc;
    return launch_reduce_partials(ws, nblk, C, alpha, out, accumulate, s);
}

}  // namespace nvq

using namespace nvq;

extern "C" {

size_t nvq_conv_pack_floats(int cout, int cin_store, int ksize, int math) {
    if (math == NVQ_MATH_BF16) return pack_floats_bf16(cout, cin_store, ksize);
    const int NT = choose_nt(cout);
    const size_t ncz = (cout + NT - 1) / NT, nkc = (cin_store + KC - 1) / KC;
    return ncz * nkc * (size_t)(ksize * ksize) * 4 * NT * 4;
}

int nvq_conv_pack(const float* w, int cout_w, int cin_w, int ksize, int transpose, int cin_store,
                  int cout_keep, int math, float* wpack, void* stream) {
    NVQ_REQUIRE(ksize == 1 || ksize == 3, "conv_pack: ksize %d", ksize);
    const int cout = transpose ? cout_keep : cout_w;
    const int cin_real = transpose ? cout_w : cin_w;
    NVQ_REQUIRE(cin_store >= cin_real && cin_store % 4 == 0, "conv_pack: cin_store %d < %d or not %%4",
                cin_store, cin_real);
    NVQ_REQUIRE(!transpose || cout_keep <= cin_w, "conv_pack: cout_keep %d > cin_w %d", cout_keep, cin_w);
    NVQ_REQUIRE(math == NVQ_MATH_F32 || math == NVQ_MATH_BF16, "conv_pack: math mode %d", math);
    if (math == NVQ_MATH_BF16)
        return pack_bf16(w, cout_w, cin_w, ksize, transpose, cin_store, cout_keep, wpack, (hipStream_t)stream);
    const int NT = choose_nt(cout);
    const int ncz = (cout + NT - 1) / NT, nkc = (cin_store + KC - 1) / KC;
    const long total = (long)ncz * nkc * ksize * ksize * 4 * NT * 4;
    int nblk = ceil_div(total, 256);
    if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL(pack_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, w, cout_w, cin_w,
                       ksize * ksize, transpose, cout_keep, NT, ncz, nkc, wpack);
    return check_launch("conv_pack");
}

int nvq_conv_pack_batch(const nvq_pack_job* jobs, int njobs, int math, void* stream) {
    NVQ_REQUIRE(njobs >= 0 && (njobs == 0 || jobs), "conv_pack_batch: %d jobs", njobs);
    NVQ_REQUIRE(math == NVQ_MATH_F32 || math == NVQ_MATH_BF16, "conv_pack_batch: math mode %d", math);
    for (int i = 0; i < njobs; ++i) {                        // the checks of nvq_conv_pack, for every job, before any launch
        const nvq_pack_job& j = jobs[i];
        NVQ_REQUIRE(j.ksize == 1 || j.ksize == 3, "conv_pack_batch[%d]: ksize %d", i, j.ksize);
        const int cin_real = j.transpose ? j.cout_w : j.cin_w;
        NVQ_REQUIRE(j.cin_store >= cin_real && j.cin_store % 4 == 0, "conv_pack_batch[%d]: cin_store %d < %d or not %%4", i,
                    j.cin_store, cin_real);
        NVQ_REQUIRE(!j.transpose || j.cout_keep <= j.cin_w, "conv_pack_batch[%d]: cout_keep %d > cin_w %d", i, j.cout_keep,
                    j.cin_w);
        NVQ_REQUIRE(j.w && j.wpack, "conv_pack_batch[%d]: null pointer", i);
    }
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < njobs; base += PACK_BATCH) {
        const int n = njobs - base < PACK_BATCH ? njobs - base : PACK_BATCH;
        PackJobTable t{};
        for (int i = 0; i < n; ++i) {
            const nvq_pack_job& j = jobs[base + i];
            if (math == NVQ_MATH_BF16) {
                t.j[i] = pack_job_bf16(j);
            } else {
                const int cout = j.transpose ? j.cout_keep : j.cout_w;
                const int NT = choose_nt(cout);
                const int ncz = (cout + NT - 1) / NT, nkc = (j.cin_store + KC - 1) / KC;
                t.j[i] = PackJobDev{j.w, j.wpack, j.cout_w, j.cin_w, j.ksize * j.ksize, j.transpose, j.cout_keep, NT, ncz, nkc,
                                    (long)ncz * nkc * j.ksize * j.ksize * 4 * NT * 4};
            }
        }
        if (math == NVQ_MATH_BF16) {
            const int rc = pack_batch_bf16(t, n, s);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL(pack_batch_kernel, dim3(32, n), dim3(256), 0, s, t);
            const int rc = check_launch("conv_pack_batch");
            if (rc) return rc;
        }
    }
    return NVQ_OK;
}

size_t nvq_rdb_backward_weights_floats(int F) { return (size_t)9 * (32 * (5 * F + 320) + (size_t)F * (F + 160)); }

int nvq_rdb_backward_weights(const float* lff, const float* w0, const float* w1, const float* w2, const float* w3,
                             const float* w4, int F, float* out, void* stream) {
    NVQ_REQUIRE(F > 0 && F % 4 == 0, "rdb_backward_weights: F %d", F);
    RdbSrc src{lff, {w0, w1, w2, w3, w4}};
    const long total = (long)nvq_rdb_backward_weights_floats(F);
    int nblk = ceil_div(total, 256);
    if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL(rdb_bwd_weights_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, src, F, out, total);
    return check_launch("rdb_backward_weights");
}

#ifdef NVQ_DEBUG_TOOLS
int nvq_debug_set_conv_mode(int mode) { set_conv_debug_mode(mode); return NVQ_OK; }
int nvq_debug_conv_occupancy(int* out6) { conv_occupancy_bf16(out6); return NVQ_OK; }
#endif

size_t nvq_sizeof_conv_desc(void) { return sizeof(nvq_conv_desc); }
size_t nvq_sizeof_wgrad_desc(void) { return sizeof(nvq_wgrad_desc); }
size_t nvq_sizeof_wgrad_reduce_job(void) { return sizeof(nvq_wgrad_reduce_job); }

static int epilogue_vec_ok(const nvq_conv_desc& d) {
    int v = d.out_ld % 4 == 0 && d.out_coff % 4 == 0 && d.cout_store % 4 == 0 && aligned16(d.out) && (d.cout % 4 == 0 || !d.bias);
    if (d.bias && !aligned16(d.bias)) v = 0;
    if (d.out2 && !(d.out2_ld % 4 == 0 && d.out2_coff % 4 == 0 && aligned16(d.out2))) v = 0;
    if (d.res && !(d.res_ld % 4 == 0 && d.res_coff % 4 == 0 && d.res_cmax % 4 == 0 && aligned16(d.res))) v = 0;
    if (d.mask && !(d.mask_ld % 4 == 0 && d.mask_coff % 4 == 0 && d.mask_c0 % 4 == 0 && d.mask_c1 % 4 == 0 && aligned16(d.mask))) v = 0;
    return v;
}

// nvq_conv_desc::in_plane: bf16 input, whole 32-channel chunks, one plane = n*h*w*32 elements, 32-bit element offsets
static bool planar_input_ok(const nvq_conv_desc& d) {
    return d.math == NVQ_MATH_BF16 && d.in_bf16 && d.cin % 32 == 0 && d.in_coff == 0 &&
           (d.in_ld == 32 || d.in_ld == 64 || d.in_ld == 128) && d.in_ld <= d.cin &&
           (size_t)d.in_plane == (size_t)d.n * d.h * d.w * 32 && (size_t)(d.cin / 32) * d.in_plane < ((size_t)1 << 32);
}

int nvq_rdb_tail_forward(const nvq_conv_desc* d3p, const nvq_conv_desc* dlp, void* stream) {
    const nvq_conv_desc d3 = *d3p, dl = *dlp;
    NVQ_REQUIRE(d3.math == NVQ_MATH_BF16 && dl.math == NVQ_MATH_BF16 && d3.in_bf16 && dl.in_bf16,
                "rdb_tail_forward: NVQ_MATH_BF16 with a bf16 concat buffer only");
    NVQ_REQUIRE(d3.ksize == 3 && dl.ksize == 1 && d3.cout == 32 && d3.cout_store == 32 && dl.cout == 64 && dl.cout_store == 64,
                "rdb_tail_forward: 3x3 -> 32 channels followed by 1x1 -> 64 channels");
    NVQ_REQUIRE(d3.cin % 32 == 0 && dl.cin == d3.cin + 32 && dl.in == d3.in && dl.in_ld == d3.in_ld && dl.in_coff == d3.in_coff &&
                    dl.in_plane == d3.in_plane && d3.out_bf16,
                "rdb_tail_forward: both convs must read the same buffer");
    if (d3.in_plane) {       // slice-planar buffer: y4 is the compact plane number cin / 32
        NVQ_REQUIRE(planar_input_ok(d3) && d3.out_ld == 32 && d3.out_coff == 0 &&
                        (const char*)d3.out == (const char*)d3.in + (size_t)(d3.cin / 32) * d3.in_plane * 2,
                    "rdb_tail_forward: the 3x3 layer must write plane cin/32 of the slice-planar buffer both convs read");
    } else {
        NVQ_REQUIRE(d3.out == (float*)d3.in && d3.out_ld == d3.in_ld && d3.out_coff == d3.in_coff + d3.cin,
                    "rdb_tail_forward: the 3x3 layer must write channels [cin, cin+32) of the buffer both convs read");
    }
    NVQ_REQUIRE(d3.n == dl.n && d3.h == dl.h && d3.w == dl.w && d3.n > 0 && d3.h > 0 && d3.w > 0, "rdb_tail_forward: shapes");
    NVQ_REQUIRE(d3.in_ld % 8 == 0 && d3.in_coff % 8 == 0 && aligned16(d3.in) && aligned16(d3.wpack) && aligned16(dl.wpack),
                "rdb_tail_forward: alignment");
    NVQ_REQUIRE(!d3.res && !d3.out2 && !d3.mask && !d3.accumulate && d3.bits_mode != 2 && !dl.bits_mode && !dl.mask && !dl.accumulate,
                "rdb_tail_forward: unsupported epilogue");
    NVQ_REQUIRE(!d3.bias || aligned16(d3.bias), "rdb_tail_forward: bias alignment");
    const int vec3 = epilogue_vec_ok(d3), vecl = epilogue_vec_ok(dl);
    NVQ_REQUIRE(vec3 && vecl, "rdb_tail_forward: the epilogues must be 16-byte addressable");
    NVQ_REQUIRE(!d3.bits_mode || d3.bits, "rdb_tail_forward: bits");
    return rdb_tail_bf16(d3, dl, vec3, vecl, (hipStream_t)stream);
}

int nvq_upsampler_tail_forward(const nvq_conv_desc* dp, const float* frames, int T, int t_center, int Cimg, int s, float* out,
                               uint8_t* pass, void* stream) {
    const nvq_conv_desc d = *dp;
    NVQ_REQUIRE(d.math == NVQ_MATH_BF16 && d.in_bf16 && d.ksize == 3, "upsampler_tail_forward: NVQ_MATH_BF16, bf16 input, 3x3 only");
    NVQ_REQUIRE(d.cin > 0 && d.cin % 8 == 0 && d.in_ld % 8 == 0 && d.in_coff % 8 == 0 && aligned16(d.in) && !d.in_plane,
                "upsampler_tail_forward: input must be a 16-byte addressable bf16 slice");
    NVQ_REQUIRE(s >= 2 && s <= 4 && Cimg >= 1 && d.cout == Cimg * s * s && t_center >= 0 && t_center < T && frames && out && pass,
                "upsampler_tail_forward: cout %d for %d image channels at scale %d", d.cout, Cimg, s);
    NVQ_REQUIRE(aligned16(d.wpack) && d.n > 0 && d.h > 0 && d.w > 0 && d.center_cin == 0, "upsampler_tail_forward: args");
    return upsampler_tail_bf16(d, frames, T, t_center, Cimg, s, out, pass, (hipStream_t)stream);
}

int nvq_conv_forward(const nvq_conv_desc* dp, void* stream) {
    const nvq_conv_desc d = *dp;
    NVQ_REQUIRE(d.math == NVQ_MATH_F32 || d.math == NVQ_MATH_BF16, "conv_forward: math mode %d", d.math);
    NVQ_REQUIRE(d.ksize == 1 || d.ksize == 3, "conv_forward: ksize %d", d.ksize);
    NVQ_REQUIRE(d.cin > 0 && d.cin % 4 == 0 && d.in_ld % 4 == 0 && d.in_coff % 4 == 0 &&
                    aligned16(d.in),
                "conv_forward: input must be 16-byte addressable (cin %d ld %d coff %d)", d.cin,
                d.in_ld, d.in_coff);
    NVQ_REQUIRE(d.cout > 0 && d.cout_store >= d.cout, "conv_forward: cout %d store %d", d.cout,
                d.cout_store);
    NVQ_REQUIRE(d.out_coff + d.cout_store <= d.out_ld, "conv_forward: output slice exceeds ld");
    NVQ_REQUIRE(aligned16(d.wpack), "conv_forward: wpack alignment");
    NVQ_REQUIRE(d.n > 0 && d.h > 0 && d.w > 0, "conv_forward: empty shape");
    NVQ_REQUIRE(!d.in_plane || planar_input_ok(d),
                "conv_forward: a slice-planar input needs NVQ_MATH_BF16, bf16 storage, cin %% 32 == 0, in_coff 0, in_ld in "
                "{32, 64, 128} and in_plane = n*h*w*32 (cin %d ld %d plane %u)", d.cin, d.in_ld, d.in_plane);
    NVQ_REQUIRE(d.center_cin >= 0 && d.center_cin % 32 == 0 && d.center_cin <= d.cin && (d.center_cin == 0 || d.ksize == 3),
                "conv_forward: center_cin %d (3x3 only, multiple of 32, <= cin %d)", d.center_cin, d.cin);
    const int NT = choose_nt(d.cout);
    const int ncz = (d.cout_store + NT - 1) / NT;
    NVQ_REQUIRE(ncz == (d.cout + NT - 1) / NT, "conv_forward: cout_store %d crosses a pack chunk", d.cout_store);
    const int nkc = (d.cin + KC - 1) / KC;
    const int tilesX = (d.w + TW - 1) / TW, tilesY = (d.h + TH - 1) / TH;
    int vec_ok = d.out_ld % 4 == 0 && d.out_coff % 4 == 0 && d.cout_store % 4 == 0 && aligned16(d.out) &&
                 (d.cout % 4 == 0 || !d.bias);
    if (d.bias && !aligned16(d.bias)) vec_ok = 0;
    if (d.bits_mode != 0)
        NVQ_REQUIRE((d.bits_mode == 1 || d.bits_mode == 2) && d.bits && d.math == NVQ_MATH_BF16 &&
                        ((NT <= 32 && ncz == 1) || (NT == 64 && d.ksize == 3 && d.in_bf16 && d.tile_rows != 8)) &&
                        (d.bits_words > 0 ? d.bits_words : 1) * 32 >= d.cout_store && vec_ok && !d.mask && !d.res &&
                        !d.accumulate && !d.out2,
                    "conv_forward: bit masks need NVQ_MATH_BF16, a plain vector epilogue and cout <= 32 (or a 3x3 conv of a bf16 "
                    "input), bits_words * 32 >= cout_store");
    if (d.out2 && !(d.out2_ld % 4 == 0 && d.out2_coff % 4 == 0 && aligned16(d.out2))) vec_ok = 0;
    if (d.res && !(d.res_ld % 4 == 0 && d.res_coff % 4 == 0 && d.res_cmax % 4 == 0 && aligned16(d.res))) vec_ok = 0;
    if (d.mask && !(d.mask_ld % 4 == 0 && d.mask_coff % 4 == 0 && d.mask_c0 % 4 == 0 &&
                    d.mask_c1 % 4 == 0 && aligned16(d.mask)))
        vec_ok = 0;
    const dim3 grid((unsigned)((long)tilesX * tilesY * d.n), ncz);
    hipStream_t s = (hipStream_t)stream;
    const int epi_bf16 = d.out_bf16 | d.out2_bf16 | d.res_bf16 | d.mask_bf16;
    NVQ_REQUIRE(!(epi_bf16 | d.in_bf16) || d.math == NVQ_MATH_BF16, "conv_forward: bf16 tensors need NVQ_MATH_BF16");
    NVQ_REQUIRE(!epi_bf16 || vec_ok, "conv_forward: bf16 output / residual / mask tensors need 4-channel aligned slices");
    NVQ_REQUIRE(!d.in_bf16 || (d.cin % 8 == 0 && d.in_ld % 8 == 0 && d.in_coff % 8 == 0),
                "conv_forward: a bf16 input needs cin, ld, coff %% 8 == 0 (cin %d ld %d coff %d)", d.cin, d.in_ld, d.in_coff);
    if (d.math == NVQ_MATH_BF16) return conv_forward_bf16(d, vec_ok, s);
    // the kernel keeps per-thread activation offsets as 32-bit element counts
    NVQ_REQUIRE((size_t)d.n * d.h * d.w * d.in_ld < ((size_t)1 << 32),
                "conv_forward: input tensor of %d x %d x %d x %d elements exceeds the 32-bit offsets of the kernels", d.n, d.h, d.w,
                d.in_ld);
    // small launches: split the slab's output channels over GS workgroups per tile until the device's ~512 resident
    // workgroups are covered (BASELINE configs[0] / [4]: 8 .. 16 clips of 64x64 are 128 .. 256 tiles)
    int gs = 1;
    while (gs * 2 <= NT / 16 && (long)grid.x * grid.y * gs * 2 <= 512) gs *= 2;
    const dim3 gridz(grid.x, grid.y, gs);
#define NVQ_LAUNCH_CONV(NB, KS, GS) \
    hipLaunchKernelGGL((conv_f32_kernel<NB, KS, GS>), gridz, dim3(256), 0, s, d, tilesX, tilesY, nkc, vec_ok)
#define NVQ_LAUNCH_CONV_KS(KS)                                       \
    do {                                                             \
        if (NT == 16) NVQ_LAUNCH_CONV(1, KS, 1);                     \
        else if (NT == 32) {                                         \
            if (gs == 1) NVQ_LAUNCH_CONV(2, KS, 1);                  \
            else NVQ_LAUNCH_CONV(1, KS, 2);                          \
        } else {                                                     \
            if (gs == 1) NVQ_LAUNCH_CONV(4, KS, 1);                  \
            else if (gs == 2) NVQ_LAUNCH_CONV(2, KS, 2);             \
            else NVQ_LAUNCH_CONV(1, KS, 4);                          \
        }                                                            \
    } while (0)
    if (d.ksize == 3) NVQ_LAUNCH_CONV_KS(3);
    else NVQ_LAUNCH_CONV_KS(1);
#undef NVQ_LAUNCH_CONV_KS
#undef NVQ_LAUNCH_CONV
    return check_launch("conv_forward");
}

size_t nvq_wgrad_workspace_bytes(void) {
    return (size_t)WGRAD_MAX_SLABS * 9 * WG_C * WG_C * sizeof(float) + (size_t)512 * 256 * sizeof(float);
}

// the weight-gradient kernel of *dp (partial slabs into its workspace); *job = the reduce that finishes it
static int wgrad_partial(const nvq_wgrad_desc* dp, nvq_wgrad_reduce_job* job, void* stream) {
    const nvq_wgrad_desc d = *dp;
    NVQ_REQUIRE(d.math == NVQ_MATH_F32 || d.math == NVQ_MATH_BF16, "conv_wgrad: math mode %d", d.math);
    NVQ_REQUIRE(d.ksize == 1 || d.ksize == 3, "conv_wgrad: ksize %d", d.ksize);
    NVQ_REQUIRE(d.cin % 4 == 0 && d.x_ld % 4 == 0 && d.x_coff % 4 == 0 && aligned16(d.x),
                "conv_wgrad: x must be 16-byte addressable");
    NVQ_REQUIRE(d.dy_ld % 4 == 0 && d.dy_coff % 4 == 0 && aligned16(d.dy),
                "conv_wgrad: dy must be 16-byte addressable");
    NVQ_REQUIRE(d.cin_w > 0 && d.cin_w <= d.cin && d.cout > 0, "conv_wgrad: channels");
    NVQ_REQUIRE(!(d.x_bf16 | d.dy_bf16) || d.math == NVQ_MATH_BF16, "conv_wgrad: bf16 tensors need NVQ_MATH_BF16");
    NVQ_REQUIRE(!d.x_bf16 || (d.cin % 8 == 0 && d.x_ld % 8 == 0 && d.x_coff % 8 == 0), "conv_wgrad: bf16 x alignment");
    NVQ_REQUIRE(!d.dy_bf16 || (d.cout % 8 == 0 && d.dy_ld % 8 == 0 && d.dy_coff % 8 == 0), "conv_wgrad: bf16 dy alignment");
    NVQ_REQUIRE(d.math != NVQ_MATH_BF16 || d.dy_coff + ((d.cout + 3) & ~3) <= d.dy_ld,
                "conv_wgrad(bf16): the dy slice must be readable up to a multiple of 4 channels");
    NVQ_REQUIRE(d.workspace_bytes >= nvq_wgrad_workspace_bytes(), "conv_wgrad: workspace too small");
    NVQ_REQUIRE(!d.x_plane || (d.math == NVQ_MATH_BF16 && d.x_bf16 && d.x_coff == 0 && d.cin % 32 == 0 &&
                               (d.x_ld == 32 || d.x_ld == 64 || d.x_ld == 128) && d.x_ld <= d.cin &&
                               (size_t)d.x_plane == (size_t)d.n * d.h * d.w * 32),
                "conv_wgrad: a slice-planar x needs NVQ_MATH_BF16, bf16 storage, cin %% 32 == 0, x_coff 0, x_ld in {32, 64, 128} "
                "and x_plane = n*h*w*32");
    const int taps = d.ksize * d.ksize;
    const int nci = (d.cin_w + WG_C - 1) / WG_C, nco = (d.cout + WG_C - 1) / WG_C;
    const int tilesX = (d.w + TW - 1) / TW, tilesY = (d.h + TH - 1) / TH;
    const int ntiles = tilesX * tilesY * d.n;
    int nsplit = WGRAD_MAX_WG / (nci * nco);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > ntiles) nsplit = ntiles;
    if (nsplit >= 8) nsplit &= ~7;      // multiple of the XCD count: see xcd_tile()
    NVQ_REQUIRE((size_t)nsplit * nci * nco * taps * WG_C * WG_C * sizeof(float) <= d.workspace_bytes,
                "conv_wgrad: %d x %d channel chunks exceed the workspace", nci, nco);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nsplit, nci, nco);
    int rc;
    if (d.math == NVQ_MATH_BF16) {
        const int used = conv_wgrad_bf16(d, nsplit, nci, nco, tilesX, tilesY, ntiles, s);
        if (used < 0) return used;       // refused before any launch (message set)
        rc = check_launch("conv_wgrad_bf16");
        if (used > 0) nsplit = used;     // 64-ci workgroups use a different pixel split
    } else {
        if (d.ksize == 3)
            hipLaunchKernelGGL((wgrad_f32_kernel<3>), grid, dim3(256), 0, s, d, tilesX, tilesY, ntiles, nci, nco);
        else
            hipLaunchKernelGGL((wgrad_f32_kernel<1>), grid, dim3(256), 0, s, d, tilesX, tilesY, ntiles, nci, nco);
        rc = check_launch("conv_wgrad");
    }
    if (rc) return rc;
    // bias_part[split][coc][32] written by the ci-chunk-0 workgroups (channel = coc*32 + lane), reduced by the same launch
    *job = nvq_wgrad_reduce_job{d.workspace, d.workspace + (size_t)WGRAD_MAX_SLABS * 9 * WG_C * WG_C, d.dw, d.dbias, nsplit, nci,
                                nco, taps, d.cout, d.cin_w, d.alpha, d.accumulate};
    return NVQ_OK;
}

int nvq_conv_wgrad(const nvq_wgrad_desc* dp, void* stream) {
    nvq_wgrad_reduce_job j;
    const int rc = wgrad_partial(dp, &j, stream);
    if (rc) return rc;
    return launch_wgrad_reduce_bias(j.part, j.nsplit, j.nci, j.nco, j.taps, j.cout, j.cin_w, j.alpha, j.accumulate, j.dw,
                                    j.bias_part, j.dbias, (hipStream_t)stream);
}

int nvq_conv_wgrad_partial(const nvq_wgrad_desc* dp, nvq_wgrad_reduce_job* job, void* stream) {
    NVQ_REQUIRE(job != nullptr, "conv_wgrad_partial: job");
    return wgrad_partial(dp, job, stream);
}

int nvq_wgrad_reduce_batch(const nvq_wgrad_reduce_job* jobs, int n, void* stream) {
    NVQ_REQUIRE(n >= 0 && (n == 0 || jobs != nullptr), "wgrad_reduce_batch: jobs");
    for (int i = 0; i < n; ++i)
        NVQ_REQUIRE(jobs[i].part && jobs[i].dw && jobs[i].nsplit > 0 && jobs[i].nci > 0 && jobs[i].nco > 0 &&
                        (jobs[i].taps == 1 || jobs[i].taps == 9) && (!jobs[i].dbias || jobs[i].bias_part),
                    "wgrad_reduce_batch: job %d is not one nvq_conv_wgrad_partial filled in", i);
    for (int i0 = 0; i0 < n; i0 += WR_MAX_JOBS) {
        const int m = n - i0 < WR_MAX_JOBS ? n - i0 : WR_MAX_JOBS;
        WgradReduceTable t;
        long need = 1;
        for (int i = 0; i < m; ++i) {
            t.j[i] = jobs[i0 + i];
            const long total = (long)t.j[i].nci * t.j[i].nco * t.j[i].taps * WG_C * WG_C;
            long b = ceil_div(total, WR_E);
            if (b > 1024) b = 1024;
            b += t.j[i].dbias ? ceil_div(t.j[i].cout, 4) : 0;
            if (b > need) need = b;
        }
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)need, (unsigned)m), dim3(256), 0, (hipStream_t)stream, t);
        const int rc = check_launch("wgrad_reduce_batch");
        if (rc) return rc;
    }
    return NVQ_OK;
}


int nvq_colsum(const float* x, int x_ld, int x_coff, int C, long npix, float alpha, float* out,
               float* workspace, size_t workspace_bytes, int accumulate, void* stream) {
    return colsum_impl(x, x_ld, x_coff, C, npix, alpha, out, workspace, workspace_bytes, accumulate,
                       (hipStream_t)stream);
}

}  // extern "C"
