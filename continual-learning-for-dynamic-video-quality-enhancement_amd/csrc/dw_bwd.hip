// Backward of one depthwise 3x3 conv of the feature extractor (reference efficient_layers.py:49-66, the autograd backward of
// its nn.Conv2d(groups = C)), bf16 mode, 64 channels: input gradient AND weight gradient from one staged tile.
//
//   forward:  d[q] = sum_tap w[tap] x'[q + tap - 1],     x' = relu(bn_prev(x)) or x
//   dx'[q]   = sum_tap w[tap] dd[q - (tap - 1)]                        (input gradient: the flipped conv of dd)
//   dw[tap]  = sum_q dd[q] x'[q + tap - 1] = sum_q x'[q] dd[q - (tap - 1)]
//
// Written over q, both need the SAME nine values dd[q - (tap - 1)] per pixel: the 3x3 neighbourhood of q in the dd halo
// tile.  So a workgroup stages the dd tile with its halo and the x' tile WITHOUT halo, and every neighbourhood value read
// from LDS feeds two multiply-adds (dx' += w v, dw += x' v).  As two launches (dwconv_wgrad_bf16 + dwconv_bf16 flipped) dd
// is read twice and x with its halo: 4 tensor passes + halos; here 3 (x, dd -> dx').  The previous layer's BatchNorm + ReLU
// is evaluated on the 256 pixels of the x tile while it is staged (the separate weight-gradient kernel did it on the 340 of
// the halo tile).  Optional epilogue as in dwconv_bf16_kernel: dx = (dx' + add) where mask > 0 - the last input gradient of
// the extractor leaves as the ReLU-masked gradient of the head conv.
#include "conv_common.h"

namespace nvq {

int bn_bwd_finalize_launch(const float* part, int nblk, int C, int G, float* sums, float* dgamma, float* dbeta, hipStream_t s);  // pointwise.hip

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int FH = 8, FW = 32, FHW = FW + 2, FNPIX = (FH + 2) * FHW;   // 8 x 32 tile, 10 x 34 halo tile
constexpr int FC = 64;         // channels
constexpr int FT = 256;        // threads
constexpr int PER = (FNPIX * 8 + FT - 1) / FT;     // 16-byte dd halo pieces per thread (11; the last one partly idle)
constexpr int XPER = FH * FW * 8 / FT;             // 16-byte x pieces per thread (8)
constexpr int FMAXWG = 512;    // two workgroups per CU
// dd halo rows per unrolled group (divides 10).  Two waves per SIMD: only the LDS reads a wave has in flight itself cover their
// latency, so the row loop wants unrolling - 1 / 2 / 5 rows: 1.33 / 1.15 / 1.03 ms at 24 x 540 x 960 (all ten: spills, 2.75 ms)
constexpr int RU = 5;

struct DwBwdArgs {
    const __bf16* x; int x_ld;
    const __bf16* dd; int dd_ld;
    const float* w;            // [64][9]
    __bf16* dx; int dx_ld;
    const float *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;   // input transform (HAS_BN): statistics [G][64], affine [64]
    const float* add; int add_ld;                            // epilogue (HAS_EPI): fp32 addend, bf16 mask
    const __bf16* mask; int mask_ld;
    float* part;               // [G * gridDim.x][64 * 9]
    float* part2;              // STATS: [G][gridDim.x][2][64]
    int H, W, tilesX, tilesY, group_images, tiles_per_group;
};

__device__ __forceinline__ float4 unpack4(u32x2 v) {
    return make_float4(__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                       __uint_as_float(v[1] & 0xffff0000u));
}

// STATS (with HAS_BN): the kernel also leaves the BatchNorm-backward sums of the input transform's BatchNorm - per channel
// sum g and sum g xhat over the group, g = dx where relu(bn(x)) > 0 - as per-workgroup partials: the kernel holds both operands
// (x is the BatchNorm input, dx its activation's gradient), so the separate reduce pass over them (bn_bwd_reduce) is not needed.
// x is then staged RAW and the transform applied by the thread that uses the value (which also needs xhat).
template <bool HAS_BN, bool HAS_EPI, bool STATS>
__global__ __launch_bounds__(FT, 2) void dw_bwd_kernel(const DwBwdArgs a) {
    static_assert(!STATS || HAS_BN, "the sums belong to the input transform");
    __shared__ __attribute__((aligned(16))) __bf16 gs[FNPIX * FC];        // dd halo tile, 128 B per pixel
    __shared__ __attribute__((aligned(16))) __bf16 xs[FH * FW * FC];      // x' tile
    __shared__ __attribute__((aligned(16))) float cst[4][FC];             // mean, invstd, gamma, beta of the input transform
    const int tid = threadIdx.x;
    const int g = blockIdx.y;

    if (HAS_BN && tid < FC) {
        cst[0][tid] = a.bn_mean[g * FC + tid];
        cst[1][tid] = a.bn_invstd[g * FC + tid];
        cst[2][tid] = a.bn_gamma[tid];
        cst[3][tid] = a.bn_beta[tid];
    }
    // flipped depthwise weights of this thread's 4 channels: wf[t] = w[8 - t]
    const int c4 = tid & 15, xcol = tid >> 4;
    float4 wf[9], wacc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int c = 4 * c4, tt = 8 - t;
        wf[t] = make_float4(a.w[(c + 0) * 9 + tt], a.w[(c + 1) * 9 + tt], a.w[(c + 2) * 9 + tt], a.w[(c + 3) * 9 + tt]);
        wacc[t] = make_float4(0.f, 0.f, 0.f, 0.f);          // weight gradient of tap 8 - t
    }

    float4 bs1 = make_float4(0.f, 0.f, 0.f, 0.f), bs2 = bs1;  // STATS: sum g, sum g xhat of this thread's 4 channels
    const int tiles_per_image = a.tilesX * a.tilesY;
    u32x4 v[PER], xv[XPER];
    unsigned okm = 0, xokm = 0;
    auto locate = [&](int t, int& n, int& ty, int& tx) {
        int bt = xcd_tile(t, a.tiles_per_group);
        const int im = bt / tiles_per_image;
        bt -= im * tiles_per_image;
        ty = bt / a.tilesX;
        tx = bt - ty * a.tilesX;
        n = g * a.group_images + im;
    };
    auto fetch = [&](int t) {                                 // raw loads (clamped addresses); masked at commit
        int n, ty, tx;
        locate(t, n, ty, tx);
        okm = 0; xokm = 0;
        int tid_o = tid;                                      // opaque copy: the per-piece coordinates are recomputed per tile
        asm volatile("" : "+v"(tid_o));                       // instead of living in hoisted registers
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int item = tid_o + k * FT;
            const int hp = item >> 3, q = item & 7;
            const int hy = hp / FHW, hx = hp - hy * FHW;
            const int gy = ty * FH + hy - 1, gx = tx * FW + hx - 1;
            const bool ok = item < FNPIX * 8 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            okm |= (ok ? 1u : 0u) << k;
            v[k] = *reinterpret_cast<const u32x4*>(a.dd + (ok ? ((size_t)(n * a.H + gy) * a.W + gx) * a.dd_ld + 8 * q : 0));
        }
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid_o + k * FT;
            const int px = item >> 3, q = item & 7;
            const int gy = ty * FH + (px >> 5), gx = tx * FW + (px & 31);
            const bool ok = gy < a.H && gx < a.W;
            xokm |= (ok ? 1u : 0u) << k;
            xv[k] = *reinterpret_cast<const u32x4*>(a.x + (ok ? ((size_t)(n * a.H + gy) * a.W + gx) * a.x_ld + 8 * q : 0));
        }
    };
    auto commit = [&]() {                                     // dd -> LDS; (optional relu(bn(.))) x -> LDS; outside the image: 0
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int item = tid + k * FT;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (okm >> k) & 1 ? v[k][e] : 0u;
            if (item < FNPIX * 8) *reinterpret_cast<u32x4*>(gs + (item >> 3) * FC + 8 * (item & 7)) = o;
        }
        if constexpr (HAS_BN && !STATS) {
            // same expression and the same single bf16 rounding as bn_apply_relu_kernel (pointwise.hip)
            const int c0 = 8 * (tid & 7);
            float m[8], is[8], ga[8], be[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 t0 = *reinterpret_cast<const float4*>(&cst[0][c0 + 4 * h]);
                const float4 t1 = *reinterpret_cast<const float4*>(&cst[1][c0 + 4 * h]);
                const float4 t2 = *reinterpret_cast<const float4*>(&cst[2][c0 + 4 * h]);
                const float4 t3 = *reinterpret_cast<const float4*>(&cst[3][c0 + 4 * h]);
                m[4 * h] = t0.x; m[4 * h + 1] = t0.y; m[4 * h + 2] = t0.z; m[4 * h + 3] = t0.w;
                is[4 * h] = t1.x; is[4 * h + 1] = t1.y; is[4 * h + 2] = t1.z; is[4 * h + 3] = t1.w;
                ga[4 * h] = t2.x; ga[4 * h + 1] = t2.y; ga[4 * h + 2] = t2.z; ga[4 * h + 3] = t2.w;
                be[4 * h] = t3.x; be[4 * h + 1] = t3.y; be[4 * h + 2] = t3.z; be[4 * h + 3] = t3.w;
            }
#pragma unroll
            for (int k = 0; k < XPER; ++k) {
                u32x4 o;
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) {
                    const float x0 = __uint_as_float(xv[k][w2] << 16), x1 = __uint_as_float(xv[k][w2] & 0xffff0000u);
                    const float y0 = fmaxf((x0 - m[2 * w2]) * is[2 * w2] * ga[2 * w2] + be[2 * w2], 0.f);
                    const float y1 = fmaxf((x1 - m[2 * w2 + 1]) * is[2 * w2 + 1] * ga[2 * w2 + 1] + be[2 * w2 + 1], 0.f);
                    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                    const b2 pk = {(__bf16)y0, (__bf16)y1};
                    o[w2] = __builtin_bit_cast(unsigned, pk);
                }
                xv[k] = o;
            }
        }
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid + k * FT;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (xokm >> k) & 1 ? xv[k][e] : 0u;
            *reinterpret_cast<u32x4*>(xs + (item >> 3) * FC + 8 * (item & 7)) = o;
        }
    };

    __syncthreads();                                          // cst
    int t = blockIdx.x;
    if (t < a.tiles_per_group) fetch(t);
    for (; t < a.tiles_per_group; t += gridDim.x) {
        int n, ty, tx;
        locate(t, n, ty, tx);
        commit();
        __syncthreads();                                      // tiles ready
        if (t + (int)gridDim.x < a.tiles_per_group) fetch(t + gridDim.x);

        // dd halo rows stationary: row rr feeds the output rows rr, rr - 1, rr - 2 (neighbourhood rows i = 0, 1, 2)
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int x = xcol + 16 * h;
            const int gx = tx * FW + x;
            float4 a0 = z4, a1 = z4, a2 = z4;                  // dx' of rows rr, rr - 1, rr - 2
            float4 X0 = z4, X1 = z4, X2 = z4;                  // x' of rows rr, rr - 1, rr - 2
            constexpr int RUK = STATS ? 2 : RU;                // (the sums' registers: five unrolled rows spill)
#pragma unroll 1
            for (int r0 = 0; r0 < FH + 2; r0 += RUK)
#pragma unroll
            for (int ri = 0; ri < RUK; ++ri) {
                const int rr = r0 + ri;
                if (rr < FH) {
                    X0 = unpack4(*reinterpret_cast<const u32x2*>(xs + (rr * FW + x) * FC + 4 * c4));
                    if constexpr (STATS) {                    // raw x -> relu(bn(x)), the expression and rounding of the staging form
                        const bool ok = ty * FH + rr < a.H && gx < a.W;
                        const float4 m = *reinterpret_cast<const float4*>(&cst[0][4 * c4]), is = *reinterpret_cast<const float4*>(&cst[1][4 * c4]);
                        const float4 ga = *reinterpret_cast<const float4*>(&cst[2][4 * c4]), be = *reinterpret_cast<const float4*>(&cst[3][4 * c4]);
                        X0.x = ok ? (float)(__bf16)fmaxf((X0.x - m.x) * is.x * ga.x + be.x, 0.f) : 0.f;
                        X0.y = ok ? (float)(__bf16)fmaxf((X0.y - m.y) * is.y * ga.y + be.y, 0.f) : 0.f;
                        X0.z = ok ? (float)(__bf16)fmaxf((X0.z - m.z) * is.z * ga.z + be.z, 0.f) : 0.f;
                        X0.w = ok ? (float)(__bf16)fmaxf((X0.w - m.w) * is.w * ga.w + be.w, 0.f) : 0.f;
                    }
                }
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float4 vv = unpack4(*reinterpret_cast<const u32x2*>(gs + (rr * FHW + x + b) * FC + 4 * c4));
                    if (rr < FH) {
                        const float4 ww = wf[b]; float4& s = wacc[b];
                        a0.x = __builtin_fmaf(vv.x, ww.x, a0.x); a0.y = __builtin_fmaf(vv.y, ww.y, a0.y); a0.z = __builtin_fmaf(vv.z, ww.z, a0.z); a0.w = __builtin_fmaf(vv.w, ww.w, a0.w);
                        s.x += vv.x * X0.x; s.y += vv.y * X0.y; s.z += vv.z * X0.z; s.w += vv.w * X0.w;
                    }
                    if (rr >= 1 && rr <= FH) {
                        const float4 ww = wf[3 + b]; float4& s = wacc[3 + b];
                        a1.x = __builtin_fmaf(vv.x, ww.x, a1.x); a1.y = __builtin_fmaf(vv.y, ww.y, a1.y); a1.z = __builtin_fmaf(vv.z, ww.z, a1.z); a1.w = __builtin_fmaf(vv.w, ww.w, a1.w);
                        s.x += vv.x * X1.x; s.y += vv.y * X1.y; s.z += vv.z * X1.z; s.w += vv.w * X1.w;
                    }
                    if (rr >= 2) {
                        const float4 ww = wf[6 + b]; float4& s = wacc[6 + b];
                        a2.x = __builtin_fmaf(vv.x, ww.x, a2.x); a2.y = __builtin_fmaf(vv.y, ww.y, a2.y); a2.z = __builtin_fmaf(vv.z, ww.z, a2.z); a2.w = __builtin_fmaf(vv.w, ww.w, a2.w);
                        s.x += vv.x * X2.x; s.y += vv.y * X2.y; s.z += vv.z * X2.z; s.w += vv.w * X2.w;
                    }
                }
                if (rr >= 2) {
                    const int gy = ty * FH + rr - 2;
                    if (gy < a.H && gx < a.W) {
                        const size_t pix = (size_t)(n * a.H + gy) * a.W + gx;
                        float4 o = a2;
                        if constexpr (HAS_EPI) {
                            if (a.add) {
                                const float4 ad = ld4(a.add + pix * a.add_ld + 4 * c4);
                                o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
                            }
                            if (a.mask) {
                                const float4 mk = unpack4(*reinterpret_cast<const u32x2*>(a.mask + pix * a.mask_ld + 4 * c4));
                                if (!(mk.x > 0.f)) o.x = 0.f;
                                if (!(mk.y > 0.f)) o.y = 0.f;
                                if (!(mk.z > 0.f)) o.z = 0.f;
                                if (!(mk.w > 0.f)) o.w = 0.f;
                            }
                        }
                        const bf16x4 ob = {(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
                        *reinterpret_cast<bf16x4*>(a.dx + pix * a.dx_ld + 4 * c4) = ob;
                        if constexpr (STATS) {
                            // g = the STORED gradient where the activation is positive (X2 = relu(bn(x)) of this pixel); xhat from
                            // the raw x of the tile
                            const float4 raw = unpack4(*reinterpret_cast<const u32x2*>(xs + ((rr - 2) * FW + x) * FC + 4 * c4));
                            const float4 m = *reinterpret_cast<const float4*>(&cst[0][4 * c4]), is = *reinterpret_cast<const float4*>(&cst[1][4 * c4]);
                            const float g0 = X2.x > 0.f ? (float)ob[0] : 0.f, g1 = X2.y > 0.f ? (float)ob[1] : 0.f;
                            const float g2 = X2.z > 0.f ? (float)ob[2] : 0.f, g3 = X2.w > 0.f ? (float)ob[3] : 0.f;
                            bs1.x += g0; bs1.y += g1; bs1.z += g2; bs1.w += g3;
                            bs2.x += g0 * ((raw.x - m.x) * is.x); bs2.y += g1 * ((raw.y - m.y) * is.y);
                            bs2.z += g2 * ((raw.z - m.z) * is.z); bs2.w += g3 * ((raw.w - m.w) * is.w);
                        }
                    }
                }
                a2 = a1; a1 = a0; a0 = z4;
                X2 = X1; X1 = X0; X0 = z4;
            }
        }
        __syncthreads();                                      // everyone is done with the tiles before the next commit
    }

    // ---- weight-gradient partials: sum over the 16 columns that share c4 (4 lanes of a wave by shuffles, the 4 waves through
    // LDS); wacc[t] belongs to tap 8 - t
    float* red = reinterpret_cast<float*>(gs);                // [4 waves][16 c4][36]
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
        float s[4] = {wacc[t9].x, wacc[t9].y, wacc[t9].z, wacc[t9].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[e] += __shfl_xor(s[e], 16, 64);
            s[e] += __shfl_xor(s[e], 32, 64);
            if (lane < 16) red[(wave * 16 + lane) * 36 + (8 - t9) * 4 + e] = s[e];
        }
    }
    __syncthreads();
    float* prow = a.part + ((size_t)g * gridDim.x + blockIdx.x) * FC * 9;
    for (int i = tid; i < 16 * 36; i += FT) {
        const int ci = i / 36, k = i - ci * 36;               // k = tap * 4 + e
        const float s = red[(0 * 16 + ci) * 36 + k] + red[(1 * 16 + ci) * 36 + k] + red[(2 * 16 + ci) * 36 + k] +
                        red[(3 * 16 + ci) * 36 + k];
        prow[(4 * ci + (k & 3)) * 9 + (k >> 2)] = s;
    }
    if constexpr (STATS) {                                    // the same reduction for the 2 x 4 BatchNorm sums per thread
        __syncthreads();
        float sv[8] = {bs1.x, bs1.y, bs1.z, bs1.w, bs2.x, bs2.y, bs2.z, bs2.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sv[e] += __shfl_xor(sv[e], 16, 64);
            sv[e] += __shfl_xor(sv[e], 32, 64);
            if (lane < 16) red[(wave * 16 + lane) * 8 + e] = sv[e];
        }
        __syncthreads();
        if (tid < 2 * FC) {
            const int which = tid >> 6, c = tid & 63, ci = c >> 2, k = which * 4 + (c & 3);
            a.part2[((size_t)g * gridDim.x + blockIdx.x) * 2 * FC + tid] =
                red[(0 * 16 + ci) * 8 + k] + red[(1 * 16 + ci) * 8 + k] + red[(2 * 16 + ci) * 8 + k] + red[(3 * 16 + ci) * 8 + k];
        }
    }
}

}  // namespace
}  // namespace nvq

using namespace nvq;

extern "C" int nvq_dwconv_backward(const float* x, int x_ld, const nvq_bn_input* bn, const float* dy, int dy_ld,
                                   const float* weight, float* dx, int dx_ld, const nvq_dw_epilogue* epi, int N, int H, int W,
                                   float* dweight, float* bn_sums, float* bn_dgamma, float* bn_dbeta, float* workspace,
                                   size_t workspace_bytes, void* stream) {
    NVQ_REQUIRE(!bn_sums || (bn && !epi && bn_dgamma && bn_dbeta),
                "dwconv_backward: the BatchNorm-backward sums need the input transform (and its dgamma / dbeta), no epilogue");
    NVQ_REQUIRE(!bn || (bn->group_images > 0 && N % bn->group_images == 0 && N / bn->group_images <= NVQ_MAX_T),
                "dwconv_backward: groups of the input transform");
    NVQ_REQUIRE(x_ld % 8 == 0 && dy_ld % 8 == 0 && dx_ld % 8 == 0 && x_ld >= FC && dy_ld >= FC && dx_ld >= FC && aligned16(x) &&
                    aligned16(dy) && aligned16(dx),
                "dwconv_backward: 64-channel bf16 tensors, 16-byte addressable");
    NVQ_REQUIRE(!epi || ((!epi->add || (epi->add_ld % 4 == 0 && aligned16(epi->add) && !epi->add_bf16)) &&
                         (!epi->mask || (epi->mask_bf16 && epi->mask_ld % 4 == 0))),
                "dwconv_backward: epilogue takes an fp32 addend and a bf16 mask");
    NVQ_REQUIRE((long)N * H * W < ((long)1 << 31), "dwconv_backward: too many pixels");
    const int group_images = bn ? bn->group_images : N;
    const int G = N / group_images;
    const int tilesX = (W + FW - 1) / FW, tilesY = (H + FH - 1) / FH;
    const int tpg = tilesX * tilesY * group_images;
    int nwg = FMAXWG / G;
    if (nwg > tpg) nwg = tpg;
    if (nwg >= 8) nwg &= ~7;                                  // multiple of the XCD count, see xcd_tile()
    const size_t part_floats = (size_t)G * nwg * FC * 9, part2_floats = bn_sums ? (size_t)G * nwg * 2 * FC : 0;
    if ((part_floats + part2_floats) * sizeof(float) > workspace_bytes) { set_error("dwconv_backward: workspace"); return NVQ_EWORKSPACE; }
    DwBwdArgs a{reinterpret_cast<const __bf16*>(x), x_ld, reinterpret_cast<const __bf16*>(dy), dy_ld, weight,
                reinterpret_cast<__bf16*>(dx), dx_ld, bn ? bn->mean : nullptr, bn ? bn->invstd : nullptr,
                bn ? bn->gamma : nullptr, bn ? bn->beta : nullptr, epi ? epi->add : nullptr, epi ? epi->add_ld : 0,
                epi ? reinterpret_cast<const __bf16*>(epi->mask) : nullptr, epi ? epi->mask_ld : 0, workspace,
                workspace + part_floats, H, W, tilesX, tilesY, group_images, tpg};
    hipStream_t s = (hipStream_t)stream;
#define NVQ_DWBWD(B_, E_, S_) hipLaunchKernelGGL((dw_bwd_kernel<B_, E_, S_>), dim3(nwg, G), dim3(FT), 0, s, a)
    if (bn_sums) NVQ_DWBWD(true, false, true);
    else if (bn && epi) NVQ_DWBWD(true, true, false);
    else if (bn) NVQ_DWBWD(true, false, false);
    else if (epi) NVQ_DWBWD(false, true, false);
    else NVQ_DWBWD(false, false, false);
#undef NVQ_DWBWD
    int rc = check_launch("dwconv_backward");
    if (rc) return rc;
    rc = launch_reduce_partials(workspace, G * nwg, FC * 9, 1.f, dweight, 0, s);
    if (rc || !bn_sums) return rc;
    return bn_bwd_finalize_launch(workspace + part_floats, nwg, FC, G, bn_sums, bn_dgamma, bn_dbeta, s);
}
