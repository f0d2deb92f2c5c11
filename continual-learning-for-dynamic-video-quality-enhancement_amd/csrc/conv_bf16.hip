// NVQ_MATH_BF16 convolution kernels: fp32 activations in HBM, operands rounded to bf16 while they
// are staged into LDS, v_mfma_f32_16x16x32_bf16 with fp32 accumulation, fp32 epilogue.
//
// At 16x the fp32 MFMA rate these kernels are HBM-bound (a 32-channel chunk of an 8x32 tile is 43.5 KB of
// fp32 reads for ~1.2-2.3k MFMA cycles), so the structure is built around keeping loads in flight:
// the next chunk / tile is fetched into registers while the current one is consumed from LDS.
#include "conv_common.h"

namespace nvq {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int KCB = 32;    // input channels per K chunk
constexpr int XSB = 48;    // bf16 per staged pixel: 32 data + 16 pad (96 B: the 16 pixels x 4 k-groups of one
                           // ds_read_b128 / ds_read_b64_tr_b16 instruction land on distinct 16-B slots)

__device__ __forceinline__ bf16x4 cvt4(float4 v) {
    return (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}
__device__ __forceinline__ bf16x8 cvt8(float4 a, float4 b) {
    return (bf16x8){(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
}

// ---------------------------------------------------------------- weight packing (bf16)
// wpack[cz][kc][tap][g][n][j] (g = 0..3, n = 0..NT-1, j = 0..7) = bf16(W[cout = cz*NT + n][ch = kc*32 + 8g + j][tap])
__global__ void pack_bf16_kernel(const float* __restrict__ w, int cout_w, int cin_w, int taps, int transpose,
                                 int cout_keep, int NT, int ncz, int nkc, __bf16* __restrict__ wp) {
    const long total = (long)ncz * nkc * taps * 4 * NT * 8;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long t = idx;
        const int j = t & 7; t >>= 3;
        const int n = t % NT; t /= NT;
        const int g = t & 3; t >>= 2;
        const int tap = t % taps; t /= taps;
        const int kc = t % nkc; t /= nkc;
        const int cz = (int)t;
        const int co = cz * NT + n;
        const int ch = kc * KCB + 8 * g + j;
        float v = 0.f;
        if (!transpose) {
            if (co < cout_w && ch < cin_w) v = w[((long)co * cin_w + ch) * taps + tap];
        } else {
            if (co < cout_keep && ch < cout_w) v = w[((long)ch * cin_w + co) * taps + (taps - 1 - tap)];
        }
        wp[idx] = (__bf16)v;
    }
}

// ---------------------------------------------------------------- forward / input gradient
template <int NB, int KS>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(const nvq_conv_desc d, int tilesX, int tilesY, int nkc,
                                                            int vec_ok) {
    constexpr int NT = NB * 16;
    constexpr int HALO = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int HW_ = TW + 2 * HALO;
    constexpr int HH_ = TH + 2 * HALO;
    constexpr int NPIX = HW_ * HH_;
    constexpr int WS_HALFS = TAPS * 4 * NT * 8;
    constexpr int XITEMS = NPIX * 4;                          // (pixel, 8-channel group) pieces per chunk
    constexpr int XPER = (XITEMS + 255) / 256;
    constexpr int WPER = (WS_HALFS / 8 + 255) / 256;          // 16-byte pieces per thread
    __shared__ __attribute__((aligned(16))) __bf16 lds[NPIX * XSB + WS_HALFS];
    __bf16* xs = lds;
    __bf16* ws = lds + NPIX * XSB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = lane & 15;
    const int g = lane >> 4;

    int bt = blockIdx.x;
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int cz = blockIdx.y;
    const int H = d.h, W = d.w;

    f32x4 acc[NB][4];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const __bf16* wp_base = reinterpret_cast<const __bf16*>(d.wpack) + (size_t)cz * nkc * WS_HALFS;
    const float* in = d.in + d.in_coff;

    // per-thread global offsets of the activation pieces (independent of the chunk)
    long xoff[XPER];
#pragma unroll
    for (int k = 0; k < XPER; ++k) {
        const int item = tid + k * 256;
        const int hp = item >> 2;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int gy = ty * TH + hy - HALO, gx = tx * TW + hx - HALO;
        xoff[k] = (item < XITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W)
                      ? ((long)(n * H + gy) * W + gx) * d.in_ld + 8 * (item & 3)
                      : -1;
    }
    float4 xr[XPER][2];
    uint4 wr[WPER];

    auto fetch = [&](int kc) {
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            xr[k][0] = make_float4(0.f, 0.f, 0.f, 0.f);
            xr[k][1] = xr[k][0];
            const int ch = kc * KCB + 8 * ((tid + k * 256) & 3);
            if (xoff[k] >= 0) {
                if (ch < d.cin) xr[k][0] = ld4(in + xoff[k] + kc * KCB);
                if (ch + 4 < d.cin) xr[k][1] = ld4(in + xoff[k] + kc * KCB + 4);
            }
        }
        const uint4* wsrc = reinterpret_cast<const uint4*>(wp_base + (size_t)kc * WS_HALFS);
#pragma unroll
        for (int k = 0; k < WPER; ++k) {
            const int i = tid + k * 256;
            if (i < WS_HALFS / 8) wr[k] = wsrc[i];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid + k * 256;
            if (item < XITEMS)
                *reinterpret_cast<bf16x8*>(xs + (item >> 2) * XSB + 8 * (item & 3)) = cvt8(xr[k][0], xr[k][1]);
        }
#pragma unroll
        for (int k = 0; k < WPER; ++k) {
            const int i = tid + k * 256;
            if (i < WS_HALFS / 8) reinterpret_cast<uint4*>(ws)[i] = wr[k];
        }
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
            bf16x8 xb[4];
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int row = 2 * wave + (pb >> 1);
                const int x0 = (pb & 1) * 16;
                const int hp = (row + dy) * HW_ + x0 + c + dx;
                xb[pb] = *reinterpret_cast<const bf16x8*>(xs + hp * XSB + 8 * g);
            }
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                const bf16x8 wa = *reinterpret_cast<const bf16x8*>(ws + ((tap * 4 + g) * NT + cb * 16 + c) * 8);
#pragma unroll
                for (int pb = 0; pb < 4; ++pb)
                    acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb[pb], acc[cb][pb], 0, 0, 0);
            }
        }
    }
    conv_epilogue<NB>(d, acc, n, ty, tx, cz, wave, c, g, vec_ok);
}

// ---------------------------------------------------------------- weight gradient
// Same decomposition as the fp32 kernel (grid = pixel split x 32-ci chunk x 32-co chunk, wave = one 16x16
// block of (ci, co) for all taps) with K = the 32 pixels of one tile row per MFMA.  The LDS images stay
// [pixel][32 ch] (96-B pixels); ds_read_b64_tr_b16 delivers, per 16-lane group, 4 pixels x 16 channels
// transposed, i.e. exactly the k-major operand the MFMA wants.  k order inside a row: group g, element e
// -> x = 4g + e (e < 4), 16 + 4g + e - 4 (e >= 4), identical for both operands, so that the two 16-lane
// groups of a 32-lane half read 8 consecutive pixels = 8 distinct 32-B bank ranges.
template <int KS>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const nvq_wgrad_desc d, int tilesX, int tilesY,
                                                             int ntiles, int nci, int nco) {
    constexpr int HALO = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int HW_ = TW + 2 * HALO;
    constexpr int HH_ = TH + 2 * HALO;
    constexpr int NPIX = HW_ * HH_;
    constexpr int XITEMS = NPIX * 8;                 // float4 pieces of the X halo tile (32 ch)
    constexpr int XPER = (XITEMS + 255) / 256;
    constexpr int YPER = TH * TW * 8 / 256;          // float4 pieces of the dY tile per thread
    __shared__ __attribute__((aligned(16))) __bf16 lds[(NPIX + TH * TW) * XSB];
    __bf16* xs = lds;
    __bf16* dys = lds + NPIX * XSB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 15;
    const int g = lane >> 4;
    const int q = r >> 2, p = r & 3;                 // tr-read role of this lane inside its 16-lane group
    const int cib = wave >> 1, cob = wave & 1;
    const int cic = blockIdx.y, coc = blockIdx.z;
    const int H = d.h, W = d.w;
    const float* x = d.x + d.x_coff;
    const float* dy = d.dy + d.dy_coff;

    f32x4 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 xr[XPER], yr[YPER];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);   // fp32 column sums of dy (bias gradient), channels 4*(tid&7)..+3
    auto fetch = [&](int tile) {
        int bt = tile;
        const int tx = bt % tilesX; bt /= tilesX;
        const int ty = bt % tilesY;
        const int n = bt / tilesY;
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid + k * 256;
            const int hp = item >> 3, qq = item & 7;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int gy = ty * TH + hy - HALO, gx = tx * TW + hx - HALO;
            const int ch = cic * WG_C + 4 * qq;
            xr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (item < XITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W && ch < d.cin)
                xr[k] = ld4(x + ((size_t)(n * H + gy) * W + gx) * d.x_ld + ch);
        }
#pragma unroll
        for (int k = 0; k < YPER; ++k) {
            const int item = tid + k * 256;
            const int pp = item >> 3, qq = item & 7;
            const int py = pp / TW, px = pp - py * TW;
            const int gy = ty * TH + py, gx = tx * TW + px;
            const int ch = coc * WG_C + 4 * qq;
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
            yr[k] = v;
            bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid + k * 256;
            if (item < XITEMS) *reinterpret_cast<bf16x4*>(xs + (item >> 3) * XSB + 4 * (item & 7)) = cvt4(xr[k]);
        }
#pragma unroll
        for (int k = 0; k < YPER; ++k) {
            const int item = tid + k * 256;
            *reinterpret_cast<bf16x4*>(dys + (item >> 3) * XSB + 4 * (item & 7)) = cvt4(yr[k]);
        }
    };
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    auto tr_read = [&](const __bf16* base, int pix, int chblock) -> s16x4 {
        // lane (q, p) supplies row `pix + q`'s address, columns 4p..4p+3 of the 16-channel block
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4_ptr)(base + (pix + q) * XSB + chblock * 16 + 4 * p));
    };

    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        commit();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
#pragma unroll 2
        for (int py = 0; py < TH; ++py) {
            const s16x4 b0 = tr_read(dys, py * TW + 4 * g, cob);
            const s16x4 b1 = tr_read(dys, py * TW + 16 + 4 * g, cob);
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int ddy = tap / KS, ddx = tap - ddy * KS;
                const int hp = (py + ddy) * HW_ + ddx;
                const s16x4 a0 = tr_read(xs, hp + 4 * g, cib);
                const s16x4 a1 = tr_read(xs, hp + 16 + 4 * g, cib);
                const bf16x8 afrag = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag, acc[tap], 0, 0, 0);
            }
        }
    }

    float* part = d.workspace + ((size_t)(blockIdx.x * nci + cic) * nco + coc) * (TAPS * WG_C * WG_C);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            part[(tap * WG_C + cib * 16 + 4 * g + e) * WG_C + cob * 16 + r] = acc[tap][e];
    wgrad_bias_partial(d, bsum, reinterpret_cast<float*>(lds), nco, cic, coc);
}

// ---------------------------------------------------------------- host side
size_t pack_floats_bf16(int cout, int cin_store, int ksize) {
    const int NT = choose_nt(cout);
    const size_t ncz = (cout + NT - 1) / NT, nkc = (cin_store + KCB - 1) / KCB;
    return ncz * nkc * (size_t)(ksize * ksize) * 4 * NT * 8 / 2;   // bf16 pairs per float
}

int pack_bf16(const float* w, int cout_w, int cin_w, int ksize, int transpose, int cin_store, int cout_keep,
              float* wpack, hipStream_t s) {
    const int cout = transpose ? cout_keep : cout_w;
    const int NT = choose_nt(cout);
    const int ncz = (cout + NT - 1) / NT, nkc = (cin_store + KCB - 1) / KCB;
    const long total = (long)ncz * nkc * ksize * ksize * 4 * NT * 8;
    int nblk = ceil_div(total, 256);
    if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(nblk), dim3(256), 0, s, w, cout_w, cin_w, ksize * ksize, transpose,
                       cout_keep, NT, ncz, nkc, reinterpret_cast<__bf16*>(wpack));
    return check_launch("conv_pack_bf16");
}

int conv_forward_bf16(const nvq_conv_desc& d, int vec_ok, hipStream_t s) {
    const int NT = choose_nt(d.cout);
    const int ncz = (d.cout_store + NT - 1) / NT;
    const int nkc = (d.cin + KCB - 1) / KCB;
    const int tilesX = (d.w + TW - 1) / TW, tilesY = (d.h + TH - 1) / TH;
    const dim3 grid((unsigned)((long)tilesX * tilesY * d.n), ncz);
#define NVQ_LAUNCH_CONVB(NB, KS) \
    hipLaunchKernelGGL((conv_bf16_kernel<NB, KS>), grid, dim3(256), 0, s, d, tilesX, tilesY, nkc, vec_ok)
    if (d.ksize == 3) {
        if (NT == 16) NVQ_LAUNCH_CONVB(1, 3);
        else if (NT == 32) NVQ_LAUNCH_CONVB(2, 3);
        else NVQ_LAUNCH_CONVB(4, 3);
    } else {
        if (NT == 16) NVQ_LAUNCH_CONVB(1, 1);
        else if (NT == 32) NVQ_LAUNCH_CONVB(2, 1);
        else NVQ_LAUNCH_CONVB(4, 1);
    }
#undef NVQ_LAUNCH_CONVB
    return check_launch("conv_forward_bf16");
}

int conv_wgrad_bf16(const nvq_wgrad_desc& d, int nsplit, int nci, int nco, int tilesX, int tilesY, int ntiles,
                    hipStream_t s) {
    const dim3 grid(nsplit, nci, nco);
    if (d.ksize == 3)
        hipLaunchKernelGGL((wgrad_bf16_kernel<3>), grid, dim3(256), 0, s, d, tilesX, tilesY, ntiles, nci, nco);
    else
        hipLaunchKernelGGL((wgrad_bf16_kernel<1>), grid, dim3(256), 0, s, d, tilesX, tilesY, ntiles, nci, nco);
    return check_launch("conv_wgrad_bf16");
}

}  // namespace nvq
