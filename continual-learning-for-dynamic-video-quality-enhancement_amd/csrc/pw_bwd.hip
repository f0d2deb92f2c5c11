// Backward of one DepthwiseSeparableConv's  pointwise 1x1 conv -> BatchNorm2d -> ReLU  (reference efficient_layers.py:49-66,
// its autograd backward), bf16 mode, 64 channels, in ONE pass over the tensors after the BatchNorm sums:
//
//   g  = dy where relu(bn(p)) > 0, else 0                     (ReLU backward)
//   dp = gamma invstd (g - mean(g) - xhat mean(g xhat))       (BatchNorm backward, training; gamma invstd g in eval mode)
//   dd = dp W                                                 (input gradient of the 1x1 conv:  dd[px][ci] = sum_co dp[px][co] W[co][ci])
//   dW = sum_px dp^T d                                        (its weight gradient:           dW[co][ci] = sum_px dp[px][co] d[px][ci])
//
// As three launches (bn_bwd_apply, conv<4,1> with the transposed pack, wgrad<1>) dp is written once and read twice and p / dy
// are read by the first alone: 7 tensor passes.  Here a workgroup stages a 128-pixel tile of p, dy and d, forms dp on the way
// into LDS and runs both contractions on it with the matrix cores: 3 reads + 1 write (p, dy, d -> dd).  1x1 means no halo, so
// tiles are plain pixel ranges (cut at the BatchNorm groups' boundaries: the statistics are per frame group).  Workgroups
// are persistent over tiles and keep the weight-gradient accumulators (one 16-ci block x the four 16-co blocks per wave) in
// registers; partial 32 x 32 slabs go through the weight-gradient reduce kernel (fixed order, double: deterministic).
#include "conv_common.h"

namespace nvq {

int bn_backward_sums(const float* dy, int dy_ld, const float* x, int x_ld, int C, int G, long group_pix, const float* mean,
                     const float* invstd, const float* gamma, const float* beta, float* dgamma, float* dbeta, float* workspace,
                     size_t workspace_bytes, int dy_bf16, int x_bf16, float** sums_out, hipStream_t s);   // pointwise.hip
int launch_wgrad_reduce(const float* part, int nsplit, int nci, int nco, int taps, int cout, int cin_w, float alpha,
                        int accumulate, float* dw, hipStream_t s);                                        // conv_igemm.hip

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 128;        // pixels per tile
constexpr int PS = 80;         // halfs per staged pixel: 64 channels + 16 of padding (160 B: b128 and tr reads conflict-free)
constexpr int PC = 64;         // channels (in and out)
constexpr int MAXG = NVQ_MAX_T;

struct PwBwdArgs {
    const void* dy; int dy_ld;
    const __bf16* p; int p_ld;
    const __bf16* d; int d_ld;
    const float *mean, *invstd, *gamma, *beta, *sums, *w;
    __bf16* dd; int dd_ld;
    float* part;
    long group_pix;
    int G, training, tiles_per_group, ntiles;
    float inv_n;
};

__device__ __forceinline__ float blo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <bool DYB>
__global__ __launch_bounds__(256, 2) void pw_bn_bwd_kernel(const PwBwdArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * TP * PS];
    __shared__ __attribute__((aligned(16))) float cst[MAXG][6][PC];   // mean, invstd, gamma, beta, mean(g), mean(g xhat)
    __bf16* dps = lds;
    __bf16* ds_ = lds + TP * PS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g4 = lane >> 4;
    const int q = r >> 2, pp = r & 3;

    for (int i = tid; i < a.G * PC; i += 256) {
        const int g = i / PC, c = i - g * PC;
        cst[g][0][c] = a.mean[g * PC + c];
        cst[g][1][c] = a.invstd[g * PC + c];
        cst[g][2][c] = a.gamma[c];
        cst[g][3][c] = a.beta[c];
        cst[g][4][c] = a.training ? a.sums[(size_t)g * 2 * PC + c] * a.inv_n : 0.f;
        cst[g][5][c] = a.training ? a.sums[(size_t)g * 2 * PC + PC + c] * a.inv_n : 0.f;
    }
    // W^T fragments of the input gradient (A operand: row = ci, k = co): element j = W[co = kb*32 + 8 g4 + j][ci = cb*16 + r]
    bf16x8 wf[4][2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[cb][kb][j] = (__bf16)a.w[(kb * 32 + 8 * g4 + j) * PC + cb * 16 + r];
    f32x4 accw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) accw[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging role: 16-byte piece `piece` (8 channels) of pixels px0 + 32 k of the tile
    const int piece = tid & 7, px0 = tid >> 3;
    const unsigned short* p16 = reinterpret_cast<const unsigned short*>(a.p);
    const unsigned short* d16 = reinterpret_cast<const unsigned short*>(a.d);
    constexpr int YR = DYB ? 1 : 2;                        // 16-byte registers per dy piece
    u32x4 rp[4], rd[4], ry[4][YR];
    unsigned okm = 0;
    long nbase = 0;                                        // first pixel / valid pixels / group of the tile held in the registers
    int nvalid = 0, ng = 0;
    auto fetch = [&](int tile) {
        ng = tile / a.tiles_per_group;
        const int local = tile - ng * a.tiles_per_group;
        nbase = (long)ng * a.group_pix + (long)local * TP;
        const long left = a.group_pix - (long)local * TP;
        nvalid = left < TP ? (int)left : TP;
        okm = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int px = px0 + 32 * k;
            const bool ok = px < nvalid;
            okm |= (ok ? 1u : 0u) << k;
            const size_t pix = ok ? (size_t)(nbase + px) : 0;
            rp[k] = *reinterpret_cast<const u32x4*>(p16 + pix * a.p_ld + 8 * piece);
            rd[k] = *reinterpret_cast<const u32x4*>(d16 + pix * a.d_ld + 8 * piece);
            if constexpr (DYB) {
                ry[k][0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(a.dy) + pix * a.dy_ld + 8 * piece);
            } else {
                const float* yp = reinterpret_cast<const float*>(a.dy) + pix * a.dy_ld + 8 * piece;
                ry[k][0] = *reinterpret_cast<const u32x4*>(yp);
                ry[k][YR - 1] = *reinterpret_cast<const u32x4*>(yp + 4);
            }
        }
    };
    long cbase = 0;
    int cvalid = 0;
    auto commit = [&]() {                                  // registers -> LDS: dp formed on the way, d copied
        cbase = nbase; cvalid = nvalid;
        float mv[8], iv[8], gv[8], bv[8], s1[8], s2[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 t0 = *reinterpret_cast<const float4*>(&cst[ng][0][8 * piece + 4 * h]);
            const float4 t1 = *reinterpret_cast<const float4*>(&cst[ng][1][8 * piece + 4 * h]);
            const float4 t2 = *reinterpret_cast<const float4*>(&cst[ng][2][8 * piece + 4 * h]);
            const float4 t3 = *reinterpret_cast<const float4*>(&cst[ng][3][8 * piece + 4 * h]);
            const float4 t4 = *reinterpret_cast<const float4*>(&cst[ng][4][8 * piece + 4 * h]);
            const float4 t5 = *reinterpret_cast<const float4*>(&cst[ng][5][8 * piece + 4 * h]);
            mv[4 * h] = t0.x; mv[4 * h + 1] = t0.y; mv[4 * h + 2] = t0.z; mv[4 * h + 3] = t0.w;
            iv[4 * h] = t1.x; iv[4 * h + 1] = t1.y; iv[4 * h + 2] = t1.z; iv[4 * h + 3] = t1.w;
            gv[4 * h] = t2.x; gv[4 * h + 1] = t2.y; gv[4 * h + 2] = t2.z; gv[4 * h + 3] = t2.w;
            bv[4 * h] = t3.x; bv[4 * h + 1] = t3.y; bv[4 * h + 2] = t3.z; bv[4 * h + 3] = t3.w;
            s1[4 * h] = t4.x; s1[4 * h + 1] = t4.y; s1[4 * h + 2] = t4.z; s1[4 * h + 3] = t4.w;
            s2[4 * h] = t5.x; s2[4 * h + 1] = t5.y; s2[4 * h + 2] = t5.z; s2[4 * h + 3] = t5.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int px = px0 + 32 * k;
            const bool ok = (okm >> k) & 1;
            float xv[8], yv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[2 * e] = blo(rp[k][e]); xv[2 * e + 1] = bhi(rp[k][e]); }
            if constexpr (DYB) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { yv[2 * e] = blo(ry[k][0][e]); yv[2 * e + 1] = bhi(ry[k][0][e]); }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { yv[e] = __uint_as_float(ry[k][0][e]); yv[4 + e] = __uint_as_float(ry[k][YR - 1][e]); }
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {                  // the arithmetic of bn_bwd_apply_kernel (pointwise.hip)
                const float h = (xv[e] - mv[e]) * iv[e];
                const float dv = (h * gv[e] + bv[e] > 0.f) ? yv[e] : 0.f;
                const float v = a.training ? gv[e] * iv[e] * (dv - s1[e] - h * s2[e]) : gv[e] * iv[e] * dv;
                o[e] = (__bf16)(ok ? v : 0.f);
            }
            *reinterpret_cast<bf16x8*>(dps + px * PS + 8 * piece) = o;
            const u32x4 z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(ds_ + px * PS + 8 * piece) = ok ? rd[k] : z;
        }
    };
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    auto tr = [&](const __bf16* base, int pix, int chblock) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + (pix + q) * PS + chblock * 16 + 4 * pp));
    };

    __syncthreads();                                        // cst is ready
    int tile = blockIdx.x;
    if (tile < a.ntiles) fetch(tile);
    for (; tile < a.ntiles; tile += gridDim.x) {
        __syncthreads();
        commit();
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);
        // ---- input gradient of this wave's 32 pixels: dd[ci][px] = sum_co W^T[ci][co] dp[co][px]
        f32x4 acc[4][2];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(dps + (32 * wave + 16 * pb + r) * PS + kb * 32 + 8 * g4);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb][kb], bfr, acc[cb][pb], 0, 0, 0);
            }
        // lane (r, g4) holds channels cb*16 + 4 g4 .. + 3 of pixel 32 wave + 16 pb + r
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const int px = 32 * wave + 16 * pb + r;
            if (px < cvalid) {
                __bf16* o = a.dd + (size_t)(cbase + px) * a.dd_ld + 4 * g4;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    *reinterpret_cast<bf16x4*>(o + cb * 16) = (bf16x4){(__bf16)acc[cb][pb][0], (__bf16)acc[cb][pb][1],
                                                                      (__bf16)acc[cb][pb][2], (__bf16)acc[cb][pb][3]};
            }
        }
        // ---- weight gradient: this wave's 16-ci block x the four 16-co blocks, K = the tile's 128 pixels
#pragma unroll
        for (int ks = 0; ks < TP / 32; ++ks) {
            const s16x4 a0 = tr(ds_, ks * 32 + 4 * g4, wave), a1 = tr(ds_, ks * 32 + 16 + 4 * g4, wave);
            const bf16x8 afr = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int cob = 0; cob < 4; ++cob) {
                const s16x4 b0 = tr(dps, ks * 32 + 4 * g4, cob), b1 = tr(dps, ks * 32 + 16 + 4 * g4, cob);
                const bf16x8 bfr = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                accw[cob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, accw[cob], 0, 0, 0);
            }
        }
    }
    // partial slabs: [split][32-ci unit (2)][32-co unit (2)][32 ci][32 co]; wave = 16-ci block `wave`
    const int slab = wave >> 1, cil0 = (wave & 1) * 16;
#pragma unroll
    for (int cob = 0; cob < 4; ++cob) {
        float* part = a.part + ((size_t)(blockIdx.x * 2 + slab) * 2 + (cob >> 1)) * (WG_C * WG_C);
#pragma unroll
        for (int e = 0; e < 4; ++e) part[(cil0 + 4 * g4 + e) * WG_C + (cob & 1) * 16 + r] = accw[cob][e];
    }
}

}  // namespace
}  // namespace nvq

using namespace nvq;

extern "C" int nvq_pw_bn_backward(const float* dy, int dy_ld, int dy_bf16, const float* p, int p_ld, const float* d, int d_ld,
                                  int N, int group_images, int H, int W, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, int training, const float* weight, float* dd,
                                  int dd_ld, float* dgamma, float* dbeta, float* dweight, const float* sums_in,
                                  float* workspace, size_t workspace_bytes, void* stream) {
    NVQ_REQUIRE(group_images > 0 && N % group_images == 0 && N / group_images <= NVQ_MAX_T, "pw_bn_backward: groups");
    NVQ_REQUIRE(p_ld % 8 == 0 && d_ld % 8 == 0 && dd_ld % 8 == 0 && dy_ld % 8 == 0 && p_ld >= PC && d_ld >= PC && dd_ld >= PC &&
                    dy_ld >= PC && aligned16(p) && aligned16(d) && aligned16(dd) && aligned16(dy),
                "pw_bn_backward: 64-channel bf16 tensors, 16-byte addressable");
    const int G = N / group_images;
    const long group_pix = (long)group_images * H * W;
    hipStream_t s = (hipStream_t)stream;
    const int tpg = ceil_div(group_pix, (long)TP);
    const long ntiles = (long)G * tpg;
    NVQ_REQUIRE(ntiles < ((long)1 << 31), "pw_bn_backward: too many tiles");
    int nsplit = ntiles < WGRAD_MAX_WG ? (int)ntiles : WGRAD_MAX_WG;
    const size_t part_floats = (size_t)nsplit * 4 * WG_C * WG_C;
    NVQ_REQUIRE(part_floats * sizeof(float) < workspace_bytes, "pw_bn_backward: workspace");
    float* sums = const_cast<float*>(sums_in);
    int rc = NVQ_OK;
    if (!sums_in) {
        // BatchNorm sums (and dgamma / dbeta) first: two-stage reduction in the workspace behind the weight-gradient slabs
        rc = bn_backward_sums(dy, dy_ld, p, p_ld, PC, G, group_pix, mean, invstd, gamma, beta, dgamma, dbeta, workspace + part_floats,
                              workspace_bytes - part_floats * sizeof(float), dy_bf16, 1, &sums, s);
        if (rc) return rc;
    }
    PwBwdArgs a{dy, dy_ld, reinterpret_cast<const __bf16*>(p), p_ld, reinterpret_cast<const __bf16*>(d), d_ld, mean, invstd, gamma,
                beta, sums, weight, reinterpret_cast<__bf16*>(dd), dd_ld, workspace, group_pix, G, training, tpg, (int)ntiles,
                1.f / (float)group_pix};
    if (dy_bf16)
        hipLaunchKernelGGL(pw_bn_bwd_kernel<true>, dim3(nsplit), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(pw_bn_bwd_kernel<false>, dim3(nsplit), dim3(256), 0, s, a);
    rc = check_launch("pw_bn_backward");
    if (rc) return rc;
    return launch_wgrad_reduce(workspace, nsplit, 2, 2, 1, PC, PC, 1.f, 0, dweight, s);
}
