// Shared pieces of the fp32 and bf16 implicit-GEMM convolution kernels.
#pragma once
#include "common.h"

namespace nvq {

constexpr int TH = 8;      // tile rows
constexpr int TW = 32;     // tile cols
constexpr int NVQ_SMALL_IMAGE_TILES = 32;  // images of fewer 16x32-pixel tiles take the 8x32-tile conv form (conv_bf16.hip)
constexpr int WG_C = 32;   // wgrad: channels per ci / co chunk
constexpr int WGRAD_MAX_WG = 512;      // workgroups per wgrad launch (2 per CU)
constexpr int WGRAD_MAX_SLABS = 1536;  // partial slabs (split x 32-ci chunk x 32-co chunk) the workspace holds (256 x 6)

static inline int choose_nt(int cout) { return cout <= 16 ? 16 : (cout <= 32 ? 32 : 64); }

#ifdef NVQ_DEBUG_TOOLS
void set_conv_debug_mode(int m);
#endif
void conv_occupancy_bf16(int* out);

// ---- shared by the NVQ_MATH_BF16 kernels (conv_bf16.hip: v_mfma_f32_16x16x32_bf16; conv_m32.hip: v_mfma_f32_32x32x16_bf16)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // 16-byte piece

constexpr int KCB = 32;    // input channels per K chunk

// Per-(cz, kc) weight slab in the packed buffer / in LDS, padded so that 256 threads move it as a whole number
// of 16-byte pieces each.
__host__ __device__ constexpr int ws_stride_halfs(int taps, int NT) { return ((taps * 4 * NT * 8 + 2047) / 2048) * 2048; }

__device__ __forceinline__ float4 as_f4(u32x4 v) {
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
__device__ __forceinline__ bf16x4 cvt4(float4 v) {
    return (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}
__device__ __forceinline__ bf16x8 cvt8(float4 a, float4 b) {
    return (bf16x8){(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }          // even channel
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }  // odd channel

// NVQ_MATH_BF16 variants (conv_bf16.hip)
size_t pack_floats_bf16(int cout, int cin_store, int ksize);
int pack_bf16(const float* w, int cout_w, int cin_w, int ksize, int transpose, int cin_store, int cout_keep,
              float* wpack, hipStream_t s);
// One job of a batched pack launch (nvq_conv_pack_batch), in kernel-argument form; total = elements of the packed buffer
// (floats in the fp32 layout, bf16 values in the bf16 layout).
struct PackJobDev { const float* w; float* wp; int cout_w, cin_w, taps, transpose, cout_keep, NT, ncz, nkc; long total; };
constexpr int PACK_BATCH = 48;                      // jobs per launch (the table travels as a kernel argument, < 4 KB)
struct PackJobTable { PackJobDev j[PACK_BATCH]; };
int pack_batch_bf16(const PackJobTable& t, int n, hipStream_t s);
PackJobDev pack_job_bf16(const nvq_pack_job& j);
int conv_forward_bf16(const nvq_conv_desc& d, int vec_ok, hipStream_t s);
int conv_wgrad_bf16(const nvq_wgrad_desc& d, int nsplit, int nci, int nco, int tilesX, int tilesY, int ntiles,
                    hipStream_t s);
int upsampler_tail_bf16(const nvq_conv_desc& d, const float* frames, int T, int t_center, int Cimg, int s, float* out,
                        unsigned char* pass, hipStream_t st);
int rdb_tail_bf16(const nvq_conv_desc& d3, const nvq_conv_desc& dl, int vec3, int vecl, hipStream_t s);
// v_mfma_f32_32x32x16_bf16 kernels (conv_m32.hip)
int conv_forward_m32(const nvq_conv_desc& d, int variant, int dbg, hipStream_t s);
void conv_occupancy_m32(int* out);
// all-input-channel 3x3 weight gradient (wgrad_m32.hip)
bool wgrad_m32_takes(const nvq_wgrad_desc& d);
int conv_wgrad_m32(const nvq_wgrad_desc& d, hipStream_t s);

// Bias-gradient partials of the wgrad kernels: every thread holds the column sums of the dy pieces it staged
// (channels 4*(tid&7)..+3 of its co chunk).  The 32 threads that share tid&7 are summed through LDS and the
// ci-chunk-0 workgroups write bias_part[split][coc][32] behind the weight partial slabs.
__device__ __forceinline__ void wgrad_bias_partial(const nvq_wgrad_desc& d, float4 bsum, float* lds, int nco, int cic,
                                                   int coc) {
    if (cic != 0 || d.dbias == nullptr) return;     // uniform per workgroup
    __syncthreads();
    st4(lds + 4 * threadIdx.x, bsum);
    __syncthreads();
    if (threadIdx.x < 32) {
        const int q = threadIdx.x >> 2, e = threadIdx.x & 3;   // channel 4q+e of the chunk
        float s = 0.f;
        for (int k = 0; k < 32; ++k) s += lds[4 * (8 * k + q) + e];
        float* bp = d.workspace + (size_t)WGRAD_MAX_SLABS * 9 * WG_C * WG_C;
        bp[((size_t)blockIdx.x * nco + coc) * WG_C + threadIdx.x] = s;
    }
}

// Epilogue of the forward / input-gradient kernels.  acc[cb][pb] is the 16x16 MFMA result with
// M = output channel, N = pixel: lane (c = lane & 15, g = lane >> 4) holds channels
// cz*NB*16 + cb*16 + 4g .. +3 of pixel (row = 2*wave + (pb >> 1), x = (pb & 1)*16 + c) of the tile.
// stage != nullptr (bf16 output, vector path): the final value is written as bf16 to the wave's LDS staging tile
// [2 rows x 32 px][stage_px halfs] instead of global memory; the caller then stores it as whole pixel rows (16 B per lane).
// A lane's 8-byte pieces of 16 pixels at the tensor's pixel stride are the expensive way to write.
constexpr int STAGE_PX = 40;   // halfs per staged pixel for 32 output channels (80 B: 16-byte aligned rows)
constexpr int STAGE_PX64 = 72; // ... for 64 output channels (144 B)

// 4 consecutive channels of an fp32 or bf16 tensor as they come from memory (16 or 8 bytes): kept raw until every operand
// load of the epilogue is in flight - a conversion would be a use, i.e. a wait for that one load.
typedef unsigned raw4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ raw4 ld_raw4(const float* base, size_t idx, int is_bf16) {
    raw4 r;
    if (is_bf16) {
        const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(base) + idx);
        r[0] = t.x; r[1] = t.y;
    } else {
        r = *reinterpret_cast<const raw4*>(base + idx);
    }
    return r;
}
__device__ __forceinline__ float4 raw4_f(raw4 r, int is_bf16) {
    if (is_bf16)
        return make_float4(__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                           __uint_as_float(r[1] & 0xffff0000u));
    return make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
}

// FULL: bias, ReLU bits and the residual of all four pixel blocks are in flight before the first use (needs the registers:
// 4 NB + 16 NB); otherwise, like the mask always, they are loaded per (pixel block, channel block).
// barrier: a workgroup barrier between the operand loads and the first write to `stage` - the caller's "every wave is done with
// the K loop's LDS images" barrier, placed here so that the loads are in flight while the waves wait for each other (a tile's
// last barrier costs wave 0 several microseconds: tools/tile_timeline.py).  Every thread of the workgroup must then make this
// call, or a __syncthreads() of its own in its place.
template <int NB, bool FULL = (NB <= 2)>
__device__ __forceinline__ void conv_epilogue(const nvq_conv_desc& d, f32x4 (&acc)[NB][4], int n, int ty, int tx,
                                              int cz, int wave, int c, int g, int vec_ok, int th = TH,
                                              __bf16* stage = nullptr, int stage_px = STAGE_PX, bool barrier = false) {
    constexpr int NT = NB * 16;
    const int H = d.h, W = d.w;
    if (vec_ok) {
        // ---- phase 1: every operand load (bias, ReLU bits, residual, mask) back to back.  Written load - use - store per
        // (pixel block, channel block), the stores (which may alias) pin every load behind the previous store: 4 .. 16
        // memory round trips in a row per tile.  Pixels / channels outside the tensor read element 0 and are not used.
        bool okp[4];
        size_t pixv[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const int gy = ty * th + 2 * wave + (pb >> 1), gx = tx * TW + (pb & 1) * 16 + c;
            okp[pb] = gy < H && gx < W;                   // (the 4 lanes g = 0..3 of a pixel leave together)
            pixv[pb] = okp[pb] ? (size_t)(n * H + gy) * W + gx : 0;
        }
        int cov[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) cov[cb] = cz * NT + cb * 16 + 4 * g;
        constexpr int CG = FULL ? NB : 1;                 // channel blocks per load group
        constexpr int RP = FULL ? 4 : 1;
        float4 bias4[CG];
        unsigned bits_in[4];
        raw4 res4[CG][RP], msk4[CG];
        auto bias_load = [&](int cb0) {
#pragma unroll
            for (int k = 0; k < CG; ++k) bias4[k] = ld4(d.bias + (cov[cb0 + k] < d.cout ? cov[cb0 + k] : 0));
        };
        auto res_load = [&](int pb, int cb0, int slot) {
#pragma unroll
            for (int k = 0; k < CG; ++k)
                res4[k][slot] = ld_raw4(d.res, pixv[pb] * d.res_ld + d.res_coff + (cov[cb0 + k] < d.res_cmax ? cov[cb0 + k] : 0),
                                        d.res_bf16);
        };
        // one-bit ReLU masks: word (cz * NT) / 32 of the pixel's bits_words words (a wave's channels lie in one word: NT = 32, or
        // NT = 16 with a single channel chunk), bit = channel % 32
        const int bw = d.bits_words > 0 ? d.bits_words : 1, wsel = (cz * NT) >> 5;
        if constexpr (NB <= 2) {
            if (d.bits_mode == 2) {
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) bits_in[pb] = d.bits[pixv[pb] * bw + wsel];
            }
        }
        if constexpr (FULL) {
            if (d.bias) bias_load(0);
            if (d.res) {
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) res_load(pb, 0, pb);
            }
        }
        if (barrier) __syncthreads();
        // ---- phase 2: arithmetic and stores
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            if (!okp[pb]) continue;
            const size_t pix = pixv[pb];
            unsigned bits_out = 0;
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                const int k = cb % CG;
                if (k == 0) {                              // a new load group
                    if constexpr (!FULL) {
                        if (d.bias) bias_load(cb);
                        if (d.res) res_load(pb, cb, 0);
                    }
                    if (d.mask) {
#pragma unroll
                        for (int j = 0; j < CG; ++j) {
                            const bool in = cov[cb + j] >= d.mask_c0 && cov[cb + j] < d.mask_c1;
                            msk4[j] = ld_raw4(d.mask, pix * d.mask_ld + d.mask_coff + (in ? cov[cb + j] : 0), d.mask_bf16);
                        }
                    }
                }
                const int co = cov[cb];
                if (co >= d.cout_store) continue;
                float v[4] = {acc[cb][pb][0], acc[cb][pb][1], acc[cb][pb][2], acc[cb][pb][3]};
                if (d.bias && co < d.cout) {
                    v[0] += bias4[k].x; v[1] += bias4[k].y; v[2] += bias4[k].z; v[3] += bias4[k].w;
                }
                if (d.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= d.alpha;
                if (d.out2)
                    stx4(d.out2, pix * d.out2_ld + d.out2_coff + co, d.out2_bf16, make_float4(v[0], v[1], v[2], v[3]));
                if (d.res && co < d.res_cmax) {
                    const float4 r = raw4_f(res4[k][RP == 4 ? pb : 0], d.res_bf16);
                    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                }
                const size_t oi = pix * d.out_ld + d.out_coff + co;
                if (d.accumulate) {
                    const float4 o = ldx4(d.out, oi, d.out_bf16);
                    v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
                }
                if (d.mask && co >= d.mask_c0 && co < d.mask_c1) {
                    const float4 m = raw4_f(msk4[k], d.mask_bf16);
                    if (!(m.x > 0.f)) v[0] = 0.f;
                    if (!(m.y > 0.f)) v[1] = 0.f;
                    if (!(m.z > 0.f)) v[2] = 0.f;
                    if (!(m.w > 0.f)) v[3] = 0.f;
                }
                if constexpr (NB <= 2) {
                    if (d.bits_mode == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (!((bits_in[pb] >> ((co & 31) + e)) & 1u)) v[e] = 0.f;
                    } else if (d.bits_mode == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) bits_out |= (v[e] > 0.f ? 1u : 0u) << ((co & 31) + e);
                    }
                }
                if (stage)
                    *reinterpret_cast<bf16x4*>(stage + ((pb >> 1) * TW + (pb & 1) * 16 + c) * stage_px + co - cz * NT) =
                        (bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                else
                    stx4(d.out, oi, d.out_bf16, make_float4(v[0], v[1], v[2], v[3]));
            }
            if constexpr (NB <= 2) {
                if (d.bits_mode == 1) {                // OR over the 4 lanes (g) of this pixel, lane g = 0 stores the word
                    bits_out |= __shfl_xor(bits_out, 16, 64);
                    bits_out |= __shfl_xor(bits_out, 32, 64);
                    if (g == 0) d.bits[pix * bw + wsel] = bits_out;
                }
            }
        }
        return;
    }
    // ---- scalar path (channel counts / alignments the 16-byte path cannot take)
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        const int row = 2 * wave + (pb >> 1);
        const int gy = ty * th + row;
        const int gx = tx * TW + (pb & 1) * 16 + c;
        if (gy >= H || gx >= W) continue;
        const size_t pix = (size_t)(n * H + gy) * W + gx;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const int co = cz * NT + cb * 16 + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ce = co + e;
                if (ce >= d.cout_store) continue;
                float x = acc[cb][pb][e];
                if (d.bias && ce < d.cout) x += d.bias[ce];
                if (d.relu) x = fmaxf(x, 0.f);
                x *= d.alpha;
                if (d.out2) d.out2[pix * d.out2_ld + d.out2_coff + ce] = x;
                if (d.res && ce < d.res_cmax) x += d.res[pix * d.res_ld + d.res_coff + ce];
                float* op = d.out + pix * d.out_ld + d.out_coff + ce;
                if (d.accumulate) x += *op;
                if (d.mask && ce >= d.mask_c0 && ce < d.mask_c1 &&
                    !(d.mask[pix * d.mask_ld + d.mask_coff + ce] > 0.f))
                    x = 0.f;
                *op = x;
            }
        }
    }
}

}  // namespace nvq
