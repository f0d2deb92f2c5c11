// Forward of one DepthwiseSeparableConv of the feature extractor up to its BatchNorm statistics (reference
// efficient_layers.py:49-66: depthwise 3x3 -> pointwise 1x1 -> BatchNorm2d -> ReLU), bf16 mode, 64 channels, in ONE pass:
//
//   x' = relu(bn_prev(x))          optional: the previous layer's BatchNorm + ReLU, applied while the halo tile is staged
//   d  = depthwise3x3(x')          VALU, from the LDS halo tile; stored (the backward needs it) and kept in LDS
//   p  = d W^T                     matrix cores: p[px][co] = sum_ci d[px][ci] W[co][ci]
//   part[g][wg] = {sum p, sum p^2} per channel over the workgroup's pixels, of the bf16 values that are stored
//
// As three launches (dwconv_bf16, conv<4,1>, bn_stats) d is written and read again and p is written and read again: 5 tensor
// passes; here 3 (x -> d, p).  The BatchNorm statistics are per frame group, so the grid is (workgroups, groups) and a
// workgroup only walks tiles of its own group; bn_finalize_kernel turns the partials into mean / invstd / running statistics
// exactly as after bn_stats_kernel.
#include "conv_common.h"

namespace nvq {

int bn_finalize_launch(const float* part, int nblk, int C, int G, long group_pix, float eps, float momentum,
                       const int* order_host, float* mean, float* invstd, float* rmean, float* rvar, hipStream_t s);  // pointwise.hip

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int FH = 8, FW = 32, FHW = FW + 2, FNPIX = (FH + 2) * FHW;   // 8 x 32 output tile, 10 x 34 halo tile
constexpr int FC = 64;         // channels
constexpr int FT = 256;        // threads
constexpr int DPS = 72;        // halfs per pixel of the d / p tile in LDS (144 B: the 16 pixels of a b128 fragment read hit 16 bank groups)
constexpr int PER = (FNPIX * 8 + FT - 1) / FT;     // 16-byte halo pieces per thread (11; the last one partly idle)
constexpr int FMAXWG = 512;    // two workgroups per CU

struct DwPwArgs {
    const __bf16* in; int in_ld;
    const float* dww;          // depthwise weight [64][9]
    const float* pww;          // pointwise weight [64 co][64 ci]
    __bf16* d; int d_ld;
    __bf16* p; int p_ld;
    const float *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;   // input transform (HAS_BN): statistics [G][64], affine [64]
    float* part;               // [G][gridDim.x][2][64] or nullptr (no statistics wanted)
    int H, W, tilesX, tilesY, group_images, tiles_per_group;
};

__device__ __forceinline__ float4 unpack4(u32x2 v) {
    return make_float4(__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                       __uint_as_float(v[1] & 0xffff0000u));
}

template <bool HAS_BN>
__global__ __launch_bounds__(FT, 2) void dwpw_fwd_kernel(const DwPwArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 xs[FNPIX * FC];        // halo tile, 128 B per pixel
    __shared__ __attribute__((aligned(16))) __bf16 ds_[FH * FW * DPS];    // d tile, then (per wave) the p tile
    __shared__ __attribute__((aligned(16))) float cst[4][FC];             // mean, invstd, gamma, beta of the input transform
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g4 = lane >> 4;
    const int g = blockIdx.y;

    if (HAS_BN && tid < FC) {
        cst[0][tid] = a.bn_mean[g * FC + tid];
        cst[1][tid] = a.bn_invstd[g * FC + tid];
        cst[2][tid] = a.bn_gamma[tid];
        cst[3][tid] = a.bn_beta[tid];
    }
    // depthwise weights of this thread's 4 channels
    const int c4 = tid & 15, xcol = tid >> 4;
    float4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int c = 4 * c4;
        w[t] = make_float4(a.dww[(c + 0) * 9 + t], a.dww[(c + 1) * 9 + t], a.dww[(c + 2) * 9 + t], a.dww[(c + 3) * 9 + t]);
    }
    // pointwise weight fragments (A operand: row = co, k = ci): element j = W[co = cb*16 + r][ci = kb*32 + 8 g4 + j]
    bf16x8 wf[4][2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[cb][kb][j] = (__bf16)a.pww[(cb * 16 + r) * FC + kb * 32 + 8 * g4 + j];
    // statistics of this lane's 16 output channels cb*16 + 4 g4 + e
    f32x4 ssum[4], ssq[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) { ssum[cb] = (f32x4){0.f, 0.f, 0.f, 0.f}; ssq[cb] = ssum[cb]; }

    const int tiles_per_image = a.tilesX * a.tilesY;
    u32x4 v[PER];
    unsigned okm = 0;
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
        okm = 0;
        int tid_o = tid;                                      // opaque copy: the per-piece halo coordinates are recomputed per
        asm volatile("" : "+v"(tid_o));                       // tile instead of living in 20+ hoisted registers
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int item = tid_o + k * FT;
            const int hp = item >> 3, q = item & 7;
            const int hy = hp / FHW, hx = hp - hy * FHW;
            const int gy = ty * FH + hy - 1, gx = tx * FW + hx - 1;
            const bool ok = item < FNPIX * 8 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            okm |= (ok ? 1u : 0u) << k;
            v[k] = *reinterpret_cast<const u32x4*>(a.in + (ok ? ((size_t)(n * a.H + gy) * a.W + gx) * a.in_ld + 8 * q : 0));
        }
    };
    auto commit = [&]() {                                     // (optional relu(bn(.))) -> LDS; zero padding stays zero
        if constexpr (HAS_BN) {
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
            for (int k = 0; k < PER; ++k) {
                u32x4 o;
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) {
                    const float x0 = __uint_as_float(v[k][w2] << 16), x1 = __uint_as_float(v[k][w2] & 0xffff0000u);
                    const float y0 = fmaxf((x0 - m[2 * w2]) * is[2 * w2] * ga[2 * w2] + be[2 * w2], 0.f);
                    const float y1 = fmaxf((x1 - m[2 * w2 + 1]) * is[2 * w2 + 1] * ga[2 * w2 + 1] + be[2 * w2 + 1], 0.f);
                    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                    const b2 pk = {(__bf16)y0, (__bf16)y1};
                    o[w2] = __builtin_bit_cast(unsigned, pk);
                }
                v[k] = o;
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int item = tid + k * FT;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (okm >> k) & 1 ? v[k][e] : 0u;
            if (item < FNPIX * 8) *reinterpret_cast<u32x4*>(xs + (item >> 3) * FC + 8 * (item & 7)) = o;
        }
    };

    __syncthreads();                                          // cst
    int t = blockIdx.x;
    if (t < a.tiles_per_group) fetch(t);
    for (; t < a.tiles_per_group; t += gridDim.x) {
        int n, ty, tx;
        locate(t, n, ty, tx);
        commit();
        __syncthreads();                                      // xs ready; every wave is past the previous tile's fragment reads
        if (t + (int)gridDim.x < a.tiles_per_group) fetch(t + gridDim.x);

        // ---- depthwise 3x3, input-row stationary (as dwconv_bf16_kernel): two columns per thread
        auto dw_row = [&](int rr, int x, float4& a0, float4& a1, float4& a2) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float4 vv = unpack4(*reinterpret_cast<const u32x2*>(xs + (rr * FHW + x + b) * FC + 4 * c4));
                if (rr < FH) { const float4 ww = w[b]; a0.x = __builtin_fmaf(vv.x, ww.x, a0.x); a0.y = __builtin_fmaf(vv.y, ww.y, a0.y); a0.z = __builtin_fmaf(vv.z, ww.z, a0.z); a0.w = __builtin_fmaf(vv.w, ww.w, a0.w); }
                if (rr >= 1 && rr <= FH) { const float4 ww = w[3 + b]; a1.x = __builtin_fmaf(vv.x, ww.x, a1.x); a1.y = __builtin_fmaf(vv.y, ww.y, a1.y); a1.z = __builtin_fmaf(vv.z, ww.z, a1.z); a1.w = __builtin_fmaf(vv.w, ww.w, a1.w); }
                if (rr >= 2) { const float4 ww = w[6 + b]; a2.x = __builtin_fmaf(vv.x, ww.x, a2.x); a2.y = __builtin_fmaf(vv.y, ww.y, a2.y); a2.z = __builtin_fmaf(vv.z, ww.z, a2.z); a2.w = __builtin_fmaf(vv.w, ww.w, a2.w); }
            }
        };
        auto dw_out = [&](int rr, int x, const float4& a2) {
            const int y = rr - 2;
            const int gy = ty * FH + y, gx = tx * FW + x;
            const bool ok = gy < a.H && gx < a.W;
            bf16x4 o = {(__bf16)a2.x, (__bf16)a2.y, (__bf16)a2.z, (__bf16)a2.w};
            if (ok) *reinterpret_cast<bf16x4*>(a.d + ((size_t)(n * a.H + gy) * a.W + gx) * a.d_ld + 4 * c4) = o;
            else o = (bf16x4){(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};   // outside the image: p = 0, nothing counted
            *reinterpret_cast<bf16x4*>(ds_ + (y * FW + x) * DPS + 4 * c4) = o;
        };
        // the row loop is unrolled (30 independent LDS reads in flight per column: with two waves per SIMD nothing else covers
        // their latency - 1.23 -> 0.94 ms at 24 x 540 x 960), the two columns are not (registers)
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int x = xcol + 16 * h;
            float4 a0 = z4, a1 = z4, a2 = z4;                  // output rows rr, rr - 1, rr - 2
#pragma unroll
            for (int rr = 0; rr < FH + 2; ++rr) {
                dw_row(rr, x, a0, a1, a2);
                if (rr >= 2) dw_out(rr, x, a2);
                a2 = a1; a1 = a0; a0 = z4;
            }
        }
        __syncthreads();                                      // d tile ready; xs free for the next commit

        // ---- pointwise conv of this wave's 64 pixels (tile rows 2 wave, 2 wave + 1)
        __bf16* wt = ds_ + 64 * wave * DPS;                   // this wave's pixels: nobody else reads or writes them below
        // two halves of 32 pixels (not unrolled: 32 accumulators live instead of 64)
#pragma unroll 1
        for (int ph = 0; ph < 2; ++ph) {
            f32x4 acc[4][2];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            __bf16* wp = wt + (32 * ph + r) * DPS;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(wp + 16 * pb * DPS + kb * 32 + 8 * g4);
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb][kb], bfr, acc[cb][pb], 0, 0, 0);
                }
            // lane (r, g4) holds channels cb*16 + 4 g4 .. + 3 of pixel 32 ph + 16 pb + r: round, count, and stage as whole pixels
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const bf16x4 o = {(__bf16)acc[cb][pb][0], (__bf16)acc[cb][pb][1], (__bf16)acc[cb][pb][2], (__bf16)acc[cb][pb][3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float f = (float)o[e];
                        ssum[cb][e] += f;
                        ssq[cb][e] += f * f;
                    }
                    *reinterpret_cast<bf16x4*>(wp + 16 * pb * DPS + cb * 16 + 4 * g4) = o;
                }
        }
        // the wave's 64 pixels x 128 B as 16-byte pieces: 8 lanes per pixel
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = lane + 64 * i;
            const int px = id >> 3, q = id & 7;
            const int gy = ty * FH + 2 * wave + (px >> 5), gx = tx * FW + (px & 31);
            const u32x4 o = *reinterpret_cast<const u32x4*>(wt + px * DPS + 8 * q);
            if (gy < a.H && gx < a.W)
                *reinterpret_cast<u32x4*>(a.p + ((size_t)(n * a.H + gy) * a.W + gx) * a.p_ld + 8 * q) = o;
        }
    }

    if (a.part == nullptr) return;                            // uniform
    // ---- statistics partials: sum over the 16 r lanes, then over the four waves through LDS
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                ssum[cb][e] += __shfl_xor(ssum[cb][e], o, 64);
                ssq[cb][e] += __shfl_xor(ssq[cb][e], o, 64);
            }
    __syncthreads();                                          // every wave is done with ds_
    float* red = reinterpret_cast<float*>(ds_);               // [wave][2][64]
    if (r == 0) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[(wave * 2 + 0) * FC + cb * 16 + 4 * g4 + e] = ssum[cb][e];
                red[(wave * 2 + 1) * FC + cb * 16 + 4 * g4 + e] = ssq[cb][e];
            }
    }
    __syncthreads();
    if (tid < 2 * FC) {
        const float s = red[tid] + red[2 * FC + tid] + red[4 * FC + tid] + red[6 * FC + tid];
        a.part[((size_t)g * gridDim.x + blockIdx.x) * 2 * FC + tid] = s;
    }
}

}  // namespace
}  // namespace nvq

using namespace nvq;

extern "C" int nvq_dwpw_forward(const float* in, int in_ld, const nvq_bn_input* bn, const float* dw_weight,
                                const float* pw_weight, float* d, int d_ld, float* p, int p_ld, int N, int group_images,
                                int H, int W, int stats, float eps, float momentum, const int* order_host, float* mean,
                                float* invstd, float* running_mean, float* running_var, float* workspace,
                                size_t workspace_bytes, void* stream) {
    NVQ_REQUIRE(group_images > 0 && N % group_images == 0 && N / group_images <= NVQ_MAX_T, "dwpw_forward: groups");
    NVQ_REQUIRE(!bn || bn->group_images == group_images, "dwpw_forward: the input transform has other frame groups");
    NVQ_REQUIRE(in_ld % 8 == 0 && d_ld % 8 == 0 && p_ld % 8 == 0 && in_ld >= FC && d_ld >= FC && p_ld >= FC && aligned16(in) &&
                    aligned16(d) && aligned16(p),
                "dwpw_forward: 64-channel bf16 tensors, 16-byte addressable");
    NVQ_REQUIRE((long)N * H * W < ((long)1 << 31), "dwpw_forward: too many pixels");
    const int G = N / group_images;
    const int tilesX = (W + FW - 1) / FW, tilesY = (H + FH - 1) / FH;
    const int tpg = tilesX * tilesY * group_images;
    int nwg = FMAXWG / G;
    if (nwg > tpg) nwg = tpg;
    if (nwg >= 8) nwg &= ~7;                                  // multiple of the XCD count, see xcd_tile()
    float* part = nullptr;
    if (stats) {
        NVQ_REQUIRE(mean && invstd, "dwpw_forward: statistics wanted but no mean / invstd");
        if ((size_t)G * nwg * 2 * FC * sizeof(float) > workspace_bytes) { set_error("dwpw_forward: workspace"); return NVQ_EWORKSPACE; }
        part = workspace;
    }
    DwPwArgs a{reinterpret_cast<const __bf16*>(in), in_ld, dw_weight, pw_weight, reinterpret_cast<__bf16*>(d), d_ld,
               reinterpret_cast<__bf16*>(p), p_ld, bn ? bn->mean : nullptr, bn ? bn->invstd : nullptr, bn ? bn->gamma : nullptr,
               bn ? bn->beta : nullptr, part, H, W, tilesX, tilesY, group_images, tpg};
    hipStream_t s = (hipStream_t)stream;
    if (bn)
        hipLaunchKernelGGL(dwpw_fwd_kernel<true>, dim3(nwg, G), dim3(FT), 0, s, a);
    else
        hipLaunchKernelGGL(dwpw_fwd_kernel<false>, dim3(nwg, G), dim3(FT), 0, s, a);
    int rc = check_launch("dwpw_forward");
    if (rc || !stats) return rc;
    return bn_finalize_launch(part, nwg, FC, G, (long)group_images * H * W, eps, momentum, order_host, mean, invstd, running_mean,
                              running_var, s);
}
