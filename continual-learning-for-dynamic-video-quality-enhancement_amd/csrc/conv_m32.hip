// NVQ_MATH_BF16 3x3 convolutions on v_mfma_f32_32x32x16_bf16 (the kernels of conv_bf16.hip issue 16x16x32).
//
// One 32x32x16 MFMA does the work of two 16x16x32 ones in the same 32 matrix-core cycles but holds the SIMD's vector issue
// for 8 of them instead of 16 (MI355X_MICROARCH.md, cycle constants), and its operand fragments are 32 rows x 16 k: with
// M = the 32 output channels of a dense layer (ResidualDenseBlock, super_resolution.py:236-253) and N = the 32 pixels of
// a tile row, a wave's accumulators are whole rows.  The wave tile is R rows x 32 pixels x 32 channels:
//   R = 2, 8 waves: the register budget of the 16x16x32 kernel (128), same LDS reads per FLOP, half the MFMA instructions;
//   R = 4, 4 waves: a weight fragment feeds four MFMAs instead of two and a halo row's fragment up to three: 9 fragment
//                   reads per 12 MFMAs instead of 7 per 6, two waves per SIMD with 256 registers each.
// Both stage a 16 x 32-pixel tile (+ halo) and the packed 32-channel weight slab of one 32-channel K chunk per step, exactly
// like conv_bf16_kernel<2,3,true,8> (same packed weights: slab [tap][g][cout][8] is the A operand of k-step g >> 1, lane
// half g & 1), with the next chunk in registers while the current one is consumed from LDS.
//
// LDS images: activations [pixel][32 ch + 8 pad] (80 B: a ds_read_b128's 16-lane groups are 16 pixels of ONE k-group here,
// and 5 x 16 B is odd, so they land on 16 distinct 16-B slots of the 256-B bank row); weights as packed (a fragment read is
// 1 KB contiguous).
#include "conv_common.h"

namespace nvq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int XS32 = 40;   // bf16 per staged pixel

// Epilogue (vector path only; the host sends everything else to conv_bf16_kernel).  acc[r] is the 32x32 result with
// M = output channel, N = pixel: lane (x = lane & 31, h = lane >> 5) holds channels 8 j + 4 h .. + 3 (j = 0..3, registers
// 4 j .. 4 j + 3) of pixel (row0 + r, gx).  Same arithmetic, in the same order, as conv_epilogue (conv_common.h).
// stage != nullptr: the value goes as bf16 into the wave's [R x 32 px][stage_px] LDS tile instead of global memory.
// cbase: first of the fragment's 32 output channels (the 64-channel kernel calls twice); stage_c0: channel 0 of the staging tile.
// bias_lds: the 32 bias values as the workgroup put them into LDS at its entry (zero beyond cout), read instead of d.bias;
// bits_pre: the pixels' ReLU-mask words (bits_mode 2) as loaded at the workgroup's entry - either spares the tile's tail a
// memory round trip (~1.5 us for the cached bias, 2-4 us for the mask words, of ~25 us per tile: tools/tile_timeline.py).
// FULL = false: the residual is loaded per (row, channel group) where it is used (4 registers instead of 16 R), for kernels
// at the 128-register budget whose epilogue would otherwise spill.
template <int R, bool FULL = true>
__device__ __forceinline__ void conv_epilogue_m32(const nvq_conv_desc& d, f32x16 (&acc)[R], int n, int row0, int gx, int x,
                                                  int h, __bf16* stage, const float* bias_lds = nullptr,
                                                  const unsigned* bits_pre = nullptr, int cbase = 0, int stage_c0 = 0,
                                                  int stage_px = STAGE_PX) {
    const int H = d.h, W = d.w;
    bool okp[R];
    size_t pixv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        okp[r] = row0 + r < H && gx < W;                  // (both lanes of a pixel leave together)
        pixv[r] = okp[r] ? (size_t)(n * H + row0 + r) * W + gx : 0;
    }
    int cov[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cov[j] = cbase + 8 * j + 4 * h;
    const int wsel = cbase >> 5;                          // the fragment's word of a pixel's bits_words mask words
    // ---- phase 1: every operand load back to back (see conv_epilogue)
    float4 bias4[4];
    unsigned bits_in[R];
    raw4 res4[FULL ? 4 : 1][FULL ? R : 1];
    const int bw = d.bits_words > 0 ? d.bits_words : 1;
    if (d.bits_mode == 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) bits_in[r] = bits_pre ? bits_pre[r] : d.bits[pixv[r] * bw + wsel];
    }
    if (d.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            bias4[j] = bias_lds ? *reinterpret_cast<const float4*>(bias_lds + cov[j] - cbase) : ld4(d.bias + (cov[j] < d.cout ? cov[j] : 0));
    }
    if constexpr (FULL) {
        if (d.res) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    res4[j][r] = ld_raw4(d.res, pixv[r] * d.res_ld + d.res_coff + (cov[j] < d.res_cmax ? cov[j] : 0), d.res_bf16);
        }
    }
    // ---- phase 2: arithmetic and stores
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!okp[r]) continue;
        const size_t pix = pixv[r];
        unsigned bits_out = 0;
        raw4 msk4[4];
        if (d.mask) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = cov[j] >= d.mask_c0 && cov[j] < d.mask_c1;
                msk4[j] = ld_raw4(d.mask, pix * d.mask_ld + d.mask_coff + (in ? cov[j] : 0), d.mask_bf16);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = cov[j];
            if (co >= d.cout_store) continue;
            float v[4] = {acc[r][4 * j], acc[r][4 * j + 1], acc[r][4 * j + 2], acc[r][4 * j + 3]};
            if (d.bias && co < d.cout) {
                v[0] += bias4[j].x; v[1] += bias4[j].y; v[2] += bias4[j].z; v[3] += bias4[j].w;
            }
            if (d.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= d.alpha;
            if (d.out2) stx4(d.out2, pix * d.out2_ld + d.out2_coff + co, d.out2_bf16, make_float4(v[0], v[1], v[2], v[3]));
            if (d.res && co < d.res_cmax) {
                if constexpr (!FULL) res4[0][0] = ld_raw4(d.res, pix * d.res_ld + d.res_coff + co, d.res_bf16);
                const float4 q = raw4_f(res4[FULL ? j : 0][FULL ? r : 0], d.res_bf16);
                v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
            }
            const size_t oi = pix * d.out_ld + d.out_coff + co;
            if (d.accumulate) {
                const float4 o = ldx4(d.out, oi, d.out_bf16);
                v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
            }
            if (d.mask && co >= d.mask_c0 && co < d.mask_c1) {
                const float4 m = raw4_f(msk4[j], d.mask_bf16);
                if (!(m.x > 0.f)) v[0] = 0.f;
                if (!(m.y > 0.f)) v[1] = 0.f;
                if (!(m.z > 0.f)) v[2] = 0.f;
                if (!(m.w > 0.f)) v[3] = 0.f;
            }
            if (d.bits_mode == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!((bits_in[r] >> ((co & 31) + e)) & 1u)) v[e] = 0.f;
            } else if (d.bits_mode == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bits_out |= (v[e] > 0.f ? 1u : 0u) << ((co & 31) + e);
            }
            if (stage)
                *reinterpret_cast<bf16x4*>(stage + (r * TW + x) * stage_px + co - stage_c0) =
                    (bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            else
                stx4(d.out, oi, d.out_bf16, make_float4(v[0], v[1], v[2], v[3]));
        }
        if (d.bits_mode == 1) {                           // OR over the two lanes of this pixel, lane h = 0 stores the word
            bits_out |= __shfl_xor(bits_out, 32, 64);
            if (h == 0) d.bits[pix * bw + wsel] = bits_out;
        }
    }
}

// cout <= 32 (one 32-channel slab), bf16 input (cin % 8 == 0), 16 x 32-pixel tiles; R rows per wave, NW = 16 / R waves.
// PF: the fragments of the next (k-step, dx) group are read one group ahead of their MFMAs (needs the registers of R = 4's
// two waves per SIMD).
// libnvq_debug.so only (tools/tile_timeline.py): dbg & 8 = wave 0 of every workgroup leaves s_memrealtime stamps (100 MHz) and
// its HW_ID / XCC_ID in d.bits[blockIdx.x * 16 ..] (bits_mode must be 0) - the per-CU-slot timeline of a launch.
#ifdef NVQ_DEBUG_TOOLS
#define NVQ_M32_STAMP(i)                                                                              \
    do {                                                                                              \
        if ((dbg & 8) && tid == 0) {                                                                  \
            const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                           \
            d.bits[blockIdx.x * 16 + 2 * (i)] = (unsigned)t_;                                         \
            d.bits[blockIdx.x * 16 + 2 * (i) + 1] = (unsigned)(t_ >> 32);                             \
        }                                                                                             \
    } while (0)
#else
#define NVQ_M32_STAMP(i) do { } while (0)
#endif

template <int R, int NW, bool PF>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 4) void conv_m32_kernel(const nvq_conv_desc d, int tilesX, int tilesY,
                                                                            int nkc, int dbg) {
    constexpr int NT = 32, KS = 3, TAPS = 9;
    constexpr int NTHR = 64 * NW;
    constexpr int TH_ = R * NW;
    constexpr int HW_ = TW + 2, HH_ = TH_ + 2, NPIX = HW_ * HH_;
    constexpr int WS_HALFS = ws_stride_halfs(TAPS, NT);
    constexpr int XITEMS = NPIX * 4;                          // (pixel, 8-channel group) pieces per chunk
    constexpr int XPER = (XITEMS + NTHR - 1) / NTHR;
    constexpr int WPIECES = TAPS * 4 * NT;                    // 16-byte pieces of the slab that carry weights
    constexpr int WPER = (WPIECES + NTHR - 1) / NTHR;
    constexpr int CT_K = (TAPS / 2) * 4 * NT / NTHR;          // the centre tap's 4 * NT pieces: register index ...
    constexpr int CT_N = 4 * NT;                              // ... and thread count
    static_assert(((TAPS / 2) * 4 * NT) % NTHR == 0 && CT_N <= NTHR, "the centre tap starts a piece row");
    static_assert(TH_ == 16, "16-row tiles");
    static_assert(NW * R * TW * STAGE_PX <= NPIX * XS32 + WS_HALFS, "the output staging tiles fit the LDS stages");

    __shared__ __attribute__((aligned(16))) __bf16 lds[NPIX * XS32 + WS_HALFS + 2 * NT];
    __bf16* xs = lds;
    __bf16* ws = lds + NPIX * XS32;
    float* biasL = reinterpret_cast<float*>(lds + NPIX * XS32 + WS_HALFS);   // 32 floats behind the stages
    const int kcl = d.center_cin / KCB;                       // leading chunks that only have a centre tap

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 31;
    const int h = lane >> 5;
    NVQ_M32_STAMP(0);                                         // entry
#ifdef NVQ_DEBUG_TOOLS
    if ((dbg & 8) && tid == 0) {
        d.bits[blockIdx.x * 16 + 12] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        d.bits[blockIdx.x * 16 + 13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    }
#endif

    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int H = d.h, W = d.w;
    // epilogue operands fetched now: the bias into LDS (visible behind the K loop's barriers), the wave's mask words into R
    // registers
    if (tid < NT) biasL[tid] = d.bias && tid < d.cout ? d.bias[tid] : 0.f;
    unsigned bits_pre[R];
    if (d.bits_mode == 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = ty * TH_ + R * wave + r, gxx = tx * TW + x;
            bits_pre[r] = d.bits[(gy < H && gxx < W ? (size_t)(n * H + gy) * W + gxx : 0) * (d.bits_words > 0 ? d.bits_words : 1)];
        }
    }

    f32x16 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    const __bf16* wp_base = reinterpret_cast<const __bf16*>(d.wpack);
    const __bf16* in16 = reinterpret_cast<const __bf16*>(d.in) + d.in_coff;

    // Staging: the rules of conv_bf16_kernel (every load unconditional, nothing between the prefetch and the commit uses a
    // loaded value, literal `light`).
    unsigned xoff[XPER];
    bool xok[XPER];
#pragma unroll
    for (int k = 0; k < XPER; ++k) {
        const int item = tid + k * NTHR;
        const int hp = item >> 2;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int gy = ty * TH_ + hy - 1, gx = tx * TW + hx - 1;
        xok[k] = item < XITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        xoff[k] = xok[k] ? (unsigned)(((size_t)(n * H + gy) * W + gx) * (d.in_plane ? 32 : d.in_ld)) : 0u;
    }
    const int nk0 = d.in_plane ? d.in_ld >> 5 : 0x7fffffff;   // chunks that live in the leading tensor (all, if interleaved)
    const int sh0 = d.in_plane ? __ffs(d.in_ld >> 5) - 1 : 0; // log2(in_ld / 32)
    const bool interior = ty * TH_ >= 1 && tx * TW >= 1 && ty * TH_ + TH_ + 1 <= H && tx * TW + TW + 1 <= W &&
                          d.cin % KCB == 0;
    const int chg = 8 * (tid & 3);
    u32x4 xr[XPER];
    u32x4 wr[WPER];
    bool cv0 = false;                                         // channel validity of the chunk held in xr
    auto fetch = [&](int kc, bool light) {
        const int ch = kc * KCB + chg;
        cv0 = ch < d.cin;
        const bool lead = kc < nk0;                           // uniform
        const int sh = lead ? sh0 : 0;
        const unsigned cbase = lead ? (unsigned)kc * KCB : (unsigned)kc * d.in_plane;
        const unsigned o0 = cv0 ? cbase + chg : 0u;
#pragma unroll
        for (int k = 0; k < XPER; ++k) xr[k] = *reinterpret_cast<const u32x4*>(in16 + ((xoff[k] << sh) + o0));
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(wp_base + (size_t)kc * WS_HALFS);
        if (light) {
            wr[CT_K] = wsrc[(tid < CT_N ? tid : 0) + CT_K * NTHR];
        } else {
#pragma unroll
            for (int k = 0; k < WPER; ++k) wr[k] = wsrc[tid + k * NTHR < WPIECES ? tid + k * NTHR : 0];
        }
    };
    auto commit = [&](bool light) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (interior) {                                       // workgroup-uniform
#pragma unroll
            for (int k = 0; k < XPER; ++k) {
                const int item = tid + k * NTHR;
                if (item < XITEMS) *reinterpret_cast<u32x4*>(xs + (item >> 2) * XS32 + 8 * (item & 3)) = xr[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < XPER; ++k) {
                const int item = tid + k * NTHR;
                if (item < XITEMS)
                    *reinterpret_cast<u32x4*>(xs + (item >> 2) * XS32 + 8 * (item & 3)) = (xok[k] && cv0) ? xr[k] : z;
            }
        }
        if (light) {
            if (tid < CT_N) reinterpret_cast<u32x4*>(ws)[tid + CT_K * NTHR] = wr[CT_K];
        } else {
#pragma unroll
            for (int k = 0; k < WPER; ++k)
                if ((k + 1) * NTHR <= WPIECES || tid + k * NTHR < WPIECES) reinterpret_cast<u32x4*>(ws)[tid + k * NTHR] = wr[k];
        }
    };
    // fragments: B = 16 channels (k-step ks, lane half h) of the 32 pixels of halo row R*wave + rr shifted by dx;
    // A = the same 16 channels of tap's weights for the 32 output channels
    const __bf16* xrow = xs + (R * wave * HW_ + x) * XS32 + 8 * h;
    const __bf16* wfrag = ws + (h * NT + x) * 8;
    auto ldB = [&](int rr, int dx, int ks) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(xrow + (rr * HW_ + dx) * XS32 + 16 * ks);
    };
    auto ldA = [&](int tap, int ks) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(wfrag + (tap * 4 + 2 * ks) * NT * 8);
    };
    auto stage_chunk = [&](int kc, bool cl, bool fl) {
        __syncthreads();
        commit(cl);
        __syncthreads();
        if (kc + 1 < nkc && !(dbg & 2)) fetch(kc + 1, fl);
    };
    auto center_stage = [&]() {                               // the MFMAs of a centre-tap-only chunk
        bf16x8 a[2], b[2][R];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a[ks] = ldA(TAPS / 2, ks);
#pragma unroll
            for (int r = 0; r < R; ++r) b[ks][r] = ldB(r + 1, 1, ks);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b[ks][r], acc[r], 0, 0, 0);
    };

    int kc = 0;
    if (kcl > 0) {                                            // (loops of their own: see conv_bf16_kernel)
        fetch(0, true);
        for (; kc + 1 < kcl; ++kc) {
            stage_chunk(kc, true, true);
            if (!(dbg & 1)) center_stage();
        }
        stage_chunk(kc, true, false);                         // the last of them fetches a full slab
        if (!(dbg & 1)) center_stage();
        ++kc;
    } else {
        fetch(0, false);
    }
    NVQ_M32_STAMP(1);                                         // first fetch issued
    for (; kc < nkc; ++kc) {
        stage_chunk(kc, false, false);
        if (kc == 0 || kc == kcl) NVQ_M32_STAMP(2);           // first full chunk staged (its loads have arrived)
        if (dbg & 1) continue;
        // six groups (k-step, dx); a group reads the R + 2 halo-row fragments of its column shift and its three weight
        // fragments (dy = 0..2) and issues 3 R MFMAs
        if constexpr (PF) {
            bf16x8 bc[R + 2], ac[3], bn[R + 2], an[3];
#pragma unroll
            for (int rr = 0; rr < R + 2; ++rr) bc[rr] = ldB(rr, 0, 0);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) ac[dy] = ldA(dy * KS, 0);
#pragma unroll
            for (int gi = 0; gi < 6; ++gi) {
                const int nks = (gi + 1) / 3, ndx = (gi + 1) % 3;
                if (gi + 1 < 6) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) an[dy] = ldA(dy * KS + ndx, nks);
#pragma unroll
                    for (int rr = 0; rr < R + 2; ++rr) bn[rr] = ldB(rr, ndx, nks);
                }
                // (the scheduler otherwise sinks every read to just before its first use: read - lgkmcnt(0) - MFMA, the LDS
                // latency exposed per fragment with only two waves per SIMD to cover it)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[dy], bc[r + dy], acc[r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (gi + 1 < 6) {
#pragma unroll
                    for (int rr = 0; rr < R + 2; ++rr) bc[rr] = bn[rr];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) ac[dy] = an[dy];
                }
            }
        } else {
#pragma unroll
            for (int gi = 0; gi < 6; ++gi) {
                const int ks = gi / 3, dx = gi % 3;
                bf16x8 b[R + 2], a[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {               // in the order the MFMAs want them
                    a[dy] = ldA(dy * KS + dx, ks);
                    if (dy == 0) {
#pragma unroll
                        for (int rr = 0; rr < R; ++rr) b[rr] = ldB(rr, dx, ks);
                    } else {
                        b[R - 1 + dy] = ldB(R - 1 + dy, dx, ks);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);            // the group's reads in flight together, then its MFMAs
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[dy], b[r + dy], acc[r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    NVQ_M32_STAMP(3);                                         // K loop done
    const int row0 = ty * TH_ + R * wave, gx = tx * TW + x;
    const bool can_stage = d.out_bf16 && d.cout_store >= NT;  // workgroup-uniform
    if (can_stage) {
        // full 32-channel bf16 output: stage the wave's rows in LDS and store whole 64-byte pixel rows
        __syncthreads();                                      // every wave is done reading xs / ws
        __bf16* stage = lds + wave * (R * TW * STAGE_PX);
        conv_epilogue_m32<R>(d, acc, n, row0, gx, x, h, stage, biasL, bits_pre);
        NVQ_M32_STAMP(4);                                     // epilogue arithmetic done (operands arrived)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __bf16* o16 = reinterpret_cast<__bf16*>(d.out);
#pragma unroll
        for (int k = 0; k < 2 * R; ++k) {
            const int item = lane + k * 64;
            const int px = item >> 2, piece = item & 3;       // wave-local pixel (R rows x 32), 8-channel piece
            const int gy = row0 + (px >> 5), gxx = tx * TW + (px & 31);
            if (gy < H && gxx < W)
                *reinterpret_cast<u32x4*>(o16 + ((size_t)(n * H + gy) * W + gxx) * d.out_ld + d.out_coff + 8 * piece) =
                    *reinterpret_cast<const u32x4*>(stage + px * STAGE_PX + 8 * piece);
        }
        NVQ_M32_STAMP(5);                                     // stores issued
        return;
    }
    conv_epilogue_m32<R>(d, acc, n, row0, gx, x, h, nullptr, biasL, bits_pre);
}

// 64 output channels per workgroup (cout >= 64: attention, flow net, upsampler, the blocks' input-gradient conv) on 8 x 32-pixel
// tiles, against the shipped channel-split kernel (conv_bf16_kernel<2,3,true,8,2>, 16x16x32 MFMA).  blockIdx.y = the 64-channel slab.
//   NW = 4, NCO = 2 (tile_rows 264): a wave = 2 rows x 32 pixels x BOTH 32-channel fragments (64 accumulators), every pixel
//                   fragment read feeds two MFMAs and every weight fragment two rows; 190 VGPRs, two waves per SIMD;
//   NW = 8, NCO = 1 (tile_rows 265): the channel split itself on 32x32x16 - wave = (row pair wave & 3, channel half wave >> 2),
//                   32 accumulators, four waves per SIMD, half the MFMA instructions of the shipped kernel.
// Measured: profiles/r04_wide_conv_m32.txt.
template <int NW, int NCO>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 4) void conv_m32w_kernel(const nvq_conv_desc d, int tilesX, int tilesY, int nkc,
                                                                             int dbg) {
    constexpr int R = 2, NT = 64, KS = 3, TAPS = 9;
    static_assert((NW == 4 && NCO == 2) || (NW == 8 && NCO == 1), "64 channels x 8 rows per workgroup");
    constexpr int NTHR = 64 * NW;
    constexpr int TH_ = 8;
    constexpr int SPXW = NCO == 2 ? STAGE_PX64 : STAGE_PX;    // halfs per pixel of a wave's staging tile
    constexpr int HW_ = TW + 2, HH_ = TH_ + 2, NPIX = HW_ * HH_;
    constexpr int WS_HALFS = ws_stride_halfs(TAPS, NT);
    constexpr int XITEMS = NPIX * 4;
    constexpr int XPER = (XITEMS + NTHR - 1) / NTHR;
    constexpr int WPIECES = TAPS * 4 * NT;
    constexpr int WPER = (WPIECES + NTHR - 1) / NTHR;
    constexpr int CT_K = (TAPS / 2) * 4 * NT / NTHR;
    constexpr int CT_N = 4 * NT;
    static_assert(((TAPS / 2) * 4 * NT) % NTHR == 0 && CT_N <= NTHR, "the centre tap starts a piece row");
    static_assert(NW * R * TW * SPXW <= NPIX * XS32 + WS_HALFS, "the output staging tiles fit the LDS stages");

    __shared__ __attribute__((aligned(16))) __bf16 lds[NPIX * XS32 + WS_HALFS];
    __bf16* xs = lds;
    __bf16* ws = lds + NPIX * XS32;
    const int kcl = d.center_cin / KCB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 31;
    const int h = lane >> 5;
    const int cz = blockIdx.y;
    const int rp = wave & 3, half = NCO == 2 ? 0 : wave >> 2;   // the wave's row pair; its 32-channel half (channel split)

    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int H = d.h, W = d.w;

    f32x16 acc[NCO][R];
#pragma unroll
    for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cb][r][e] = 0.f;

    const __bf16* wp_base = reinterpret_cast<const __bf16*>(d.wpack) + (size_t)cz * nkc * WS_HALFS;
    const __bf16* in16 = reinterpret_cast<const __bf16*>(d.in) + d.in_coff;
    unsigned xoff[XPER];
    bool xok[XPER];
#pragma unroll
    for (int k = 0; k < XPER; ++k) {
        const int item = tid + k * NTHR;
        const int hp = item >> 2;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int gy = ty * TH_ + hy - 1, gx = tx * TW + hx - 1;
        xok[k] = item < XITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        xoff[k] = xok[k] ? (unsigned)(((size_t)(n * H + gy) * W + gx) * (d.in_plane ? 32 : d.in_ld)) : 0u;
    }
    const int nk0 = d.in_plane ? d.in_ld >> 5 : 0x7fffffff;
    const int sh0 = d.in_plane ? __ffs(d.in_ld >> 5) - 1 : 0;
    const int chg = 8 * (tid & 3);
    u32x4 xr[XPER];
    u32x4 wr[WPER];
    bool cv0 = false;
    auto fetch = [&](int kc, bool light) {
        const int ch = kc * KCB + chg;
        cv0 = ch < d.cin;
        const bool lead = kc < nk0;
        const int sh = lead ? sh0 : 0;
        const unsigned cbase = lead ? (unsigned)kc * KCB : (unsigned)kc * d.in_plane;
        const unsigned o0 = cv0 ? cbase + chg : 0u;
#pragma unroll
        for (int k = 0; k < XPER; ++k) xr[k] = *reinterpret_cast<const u32x4*>(in16 + ((xoff[k] << sh) + o0));
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(wp_base + (size_t)kc * WS_HALFS);
        if (light) {
            wr[CT_K] = wsrc[(tid < CT_N ? tid : 0) + CT_K * NTHR];
        } else {
#pragma unroll
            for (int k = 0; k < WPER; ++k) wr[k] = wsrc[tid + k * NTHR < WPIECES ? tid + k * NTHR : 0];
        }
    };
    auto commit = [&](bool light) {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid + k * NTHR;
            if (item < XITEMS)
                *reinterpret_cast<u32x4*>(xs + (item >> 2) * XS32 + 8 * (item & 3)) = (xok[k] && cv0) ? xr[k] : z;
        }
        if (light) {
            if (tid < CT_N) reinterpret_cast<u32x4*>(ws)[tid + CT_K * NTHR] = wr[CT_K];
        } else {
#pragma unroll
            for (int k = 0; k < WPER; ++k)
                if ((k + 1) * NTHR <= WPIECES || tid + k * NTHR < WPIECES) reinterpret_cast<u32x4*>(ws)[tid + k * NTHR] = wr[k];
        }
    };
    const __bf16* xrow = xs + (R * rp * HW_ + x) * XS32 + 8 * h;
    const __bf16* wfrag = ws + (h * NT + half * 32 + x) * 8;
    auto ldB = [&](int rr, int dx, int ks) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(xrow + (rr * HW_ + dx) * XS32 + 16 * ks);
    };
    auto ldA = [&](int tap, int ks, int cb) -> bf16x8 {       // slab [tap][g = 2 ks + h][64 channels][8]
        return *reinterpret_cast<const bf16x8*>(wfrag + ((tap * 4 + 2 * ks) * NT + cb * 32) * 8);
    };
    auto stage_chunk = [&](int kc, bool cl, bool fl) {
        __syncthreads();
        commit(cl);
        __syncthreads();
        if (kc + 1 < nkc && !(dbg & 2)) fetch(kc + 1, fl);
    };
    auto center_stage = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[NCO], b[R];
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb) a[cb] = ldA(TAPS / 2, ks, cb);
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = ldB(r + 1, 1, ks);
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
                for (int r = 0; r < R; ++r)
                    acc[cb][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb], b[r], acc[cb][r], 0, 0, 0);
        }
    };

    int kc = 0;
    if (kcl > 0) {
        fetch(0, true);
        for (; kc + 1 < kcl; ++kc) {
            stage_chunk(kc, true, true);
            if (!(dbg & 1)) center_stage();
        }
        stage_chunk(kc, true, false);
        if (!(dbg & 1)) center_stage();
        ++kc;
    } else {
        fetch(0, false);
    }
    for (; kc < nkc; ++kc) {
        stage_chunk(kc, false, false);
        if (dbg & 1) continue;
        // six groups (k-step, dx): R + 2 pixel fragments, 3 x NCO weight fragments, 3 R NCO MFMAs
#pragma unroll
        for (int gi = 0; gi < 6; ++gi) {
            const int ks = gi / 3, dx = gi % 3;
            bf16x8 b[R + 2], a[3][NCO];
#pragma unroll
            for (int rr = 0; rr < R + 2; ++rr) b[rr] = ldB(rr, dx, ks);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int cb = 0; cb < NCO; ++cb) a[dy][cb] = ldA(dy * KS + dx, ks, cb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        acc[cb][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[dy][cb], b[r + dy], acc[cb][r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue: the wave's fragment(s) through its LDS staging tile [R x 32 px][32 NCO channels], whole 64 / 128-byte pixel rows
    // out (host: bf16 output, all 64 channels of the slab stored)
    const int row0 = ty * TH_ + R * rp, gx = tx * TW + x;
    const int c0 = cz * NT + half * 32;                       // first channel of the wave's staging tile
    __syncthreads();                                          // every wave is done reading xs / ws
    __bf16* stage = lds + wave * (R * TW * SPXW);
#pragma unroll
    for (int cb = 0; cb < NCO; ++cb)
        conv_epilogue_m32<R, NW == 4>(d, acc[cb], n, row0, gx, x, h, stage, nullptr, nullptr, c0 + cb * 32, c0, SPXW);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __bf16* o16 = reinterpret_cast<__bf16*>(d.out);
    constexpr int PPP = 4 * NCO;                              // 16-byte pieces per pixel of the wave's channels
#pragma unroll
    for (int k = 0; k < R * TW * PPP / 64; ++k) {
        const int item = lane + k * 64;
        const int px = item / PPP, piece = item % PPP;        // wave-local pixel (R rows x 32)
        const int gy = row0 + (px >> 5), gxx = tx * TW + (px & 31);
        if (gy < H && gxx < W)
            *reinterpret_cast<u32x4*>(o16 + ((size_t)(n * H + gy) * W + gxx) * d.out_ld + d.out_coff + c0 + 8 * piece) =
                *reinterpret_cast<const u32x4*>(stage + px * SPXW + 8 * piece);
    }
}

// variant: 2 = two rows per wave, 8 waves; 4 = four rows per wave, 4 waves (fragments read one group ahead); 64 / 65 = the
// 64-channel-per-workgroup kernels (whole-slab waves / channel split)
int conv_forward_m32(const nvq_conv_desc& d, int variant, int dbg, hipStream_t s) {
    NVQ_REQUIRE((size_t)d.n * d.h * d.w * d.in_ld < ((size_t)1 << 32),
                "conv_forward(bf16, 32x32x16): input tensor of %d x %d x %d x %d elements exceeds the 32-bit offsets of the kernels",
                d.n, d.h, d.w, d.in_ld);
    const int tilesX = (d.w + TW - 1) / TW, tilesY = (d.h + 15) / 16;
    const int nkc = (d.cin + KCB - 1) / KCB;
    const dim3 grid((unsigned)((long)tilesX * tilesY * d.n));
    if (variant == 64 || variant == 65) {                     // 64 output channels per workgroup, 8 x 32 tiles (the caller checks)
        const int tY8 = (d.h + 7) / 8;
        const dim3 g8((unsigned)((long)tilesX * tY8 * d.n), (d.cout_store + 63) / 64);
        if (variant == 64) hipLaunchKernelGGL((conv_m32w_kernel<4, 2>), g8, dim3(256), 0, s, d, tilesX, tY8, nkc, dbg);
        else hipLaunchKernelGGL((conv_m32w_kernel<8, 1>), g8, dim3(512), 0, s, d, tilesX, tY8, nkc, dbg);
    } else if (variant == 2)
        hipLaunchKernelGGL((conv_m32_kernel<2, 8, false>), grid, dim3(512), 0, s, d, tilesX, tilesY, nkc, dbg);
    else
        hipLaunchKernelGGL((conv_m32_kernel<4, 4, true>), grid, dim3(256), 0, s, d, tilesX, tilesY, nkc, dbg);
    return check_launch("conv_forward_m32");
}

void conv_occupancy_m32(int* out) {
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[0], conv_m32_kernel<2, 8, false>, 512, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[1], conv_m32_kernel<4, 4, true>, 256, 0);
}

}  // namespace nvq
