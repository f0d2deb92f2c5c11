// 9x9 local correlation on the matrix cores (NVQ_MATH_BF16): forward and both gradients.
//
// For a block of 16 pixels P of one image row and one displacement row i, the 16 x 9 products
//   corr[p, i*9 + j] = (1/C) x1[p] . x2[(y+i-4, x_p+j-4)]
// are the band 0 <= q-p <= 8 of the 16 x 24 matrix  S = X1[P] . X2row[Q]^T  (Q = the 24 halo pixels of row y+i-4):
// one 16x16x32 bf16 MFMA per 16 q and 32 channels.  Only 28 % of S is used, but the whole correlation is 43 GFLOP
// per pass at 540p x 8 images - on the matrix cores that is noise, while the per-pixel fp32 loop of motion.hip is
// bound by its LDS reads (324 ds_read_b128 per pixel and 16 channels).  The gradients are the same band turned into a
// GEMM operand:  dX[P] (+)= Band_i . Yrow[Q]  with K = q.
//
// Workgroup = 8 rows x 16 pixels, 8 waves, wave w = tile row w.  The 16 x 24-pixel halo of the "other" tensor sits in
// LDS as bf16 [384 px][C + 16]: the +16 halfs (pixel stride 160 B at C = 64, 96 B at C = 32) make the b128 row reads of
// the forward conflict-free.  fp32 inputs are rounded to bf16 when staged; accumulation is fp32.
#include <type_traits>
#include "common.h"

namespace nvq {

constexpr int MT_H = 8, MT_W = 16;              // pixel tile
constexpr int MD = 4, MN = 9;                    // displacement radius, displacements per row
constexpr int MHH = MT_H + 2 * MD;              // 16 halo rows
constexpr int MHW = MT_W + 2 * MD;              // 24 halo columns
constexpr int MHP = MHH * MHW;                  // 384 halo pixels
constexpr int M_T = 512;                        // threads
constexpr int M_DSTR = 96;                      // halfs per pixel of a staged dcorr row (12 pieces of 16 B)
constexpr int M_OSTR = 84;                      // floats per pixel of the forward's output staging rows

typedef unsigned m_u32x4 __attribute__((ext_vector_type(4)));
typedef short m_s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ m_u32x4 cvt_f8_bf16(float4 a, float4 b) {
    return (m_u32x4){pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w)};
}

// Stage the halo of fp32 `src` image `n` (C channels from src, origin (ty0-4, tx0-4)) as bf16 [384][YS].
// YS = C + 16 makes the forward's b128 row reads conflict-free; C + 8 costs a 2-way conflict on half of them but lets two
// workgroups share a CU (these kernels stage, synchronise and compute one tile per workgroup, so occupancy is what
// overlaps their phases).
template <int C, int YS>
__device__ __forceinline__ void m_stage_halo(const float* __restrict__ src, int ld, int n, int H, int W, int ty0, int tx0,
                                             __bf16* ys, int src_bf16) {
    constexpr int PPP = C / 8;                   // 16-byte bf16 pieces per pixel
    constexpr int ITEMS = MHP * PPP;
    constexpr int PER = ITEMS / M_T;             // exact: 384 * {4, 8} / 512
    static_assert(ITEMS % M_T == 0, "halo pieces must divide evenly");
    if (src_bf16) {                              // (uniform) bf16-stored source: the pieces are copied as they are
        const __bf16* s16 = reinterpret_cast<const __bf16*>(src);
        // every load first (clamped addresses), the out-of-image mask applied when the piece is stored: a select right
        // behind a load is a use of it, i.e. a wait per load - PER memory round trips in a row
        m_u32x4 v[PER];
        unsigned okm = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int item = threadIdx.x + k * M_T;
            const int hp = item / PPP, q = item - hp * PPP;
            const int hy = hp / MHW, hx = hp - hy * MHW;
            const int gy = ty0 + hy - MD, gx = tx0 + hx - MD;
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            okm |= (ok ? 1u : 0u) << k;
            v[k] = *reinterpret_cast<const m_u32x4*>(s16 + (ok ? ((size_t)(n * H + gy) * W + gx) * ld + 8 * q : 0));
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int item = threadIdx.x + k * M_T;
            const int hp = item / PPP, q = item - hp * PPP;
            *reinterpret_cast<m_u32x4*>(ys + hp * YS + 8 * q) = (okm >> k) & 1 ? v[k] : (m_u32x4){0u, 0u, 0u, 0u};
        }
        return;
    }
    float4 a[PER], b[PER];
    unsigned okm = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {              // loads first (clamped addresses), mask + conversion + stores after
        const int item = threadIdx.x + k * M_T;
        const int hp = item / PPP, q = item - hp * PPP;
        const int hy = hp / MHW, hx = hp - hy * MHW;
        const int gy = ty0 + hy - MD, gx = tx0 + hx - MD;
        const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
        okm |= (ok ? 1u : 0u) << k;
        const float* p = src + (ok ? ((size_t)(n * H + gy) * W + gx) * ld + 8 * q : 0);
        a[k] = ld4(p);
        b[k] = ld4(p + 4);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int item = threadIdx.x + k * M_T;
        const int hp = item / PPP, q = item - hp * PPP;
        const m_u32x4 v = cvt_f8_bf16(a[k], b[k]);
        *reinterpret_cast<m_u32x4*>(ys + hp * YS + 8 * q) = (okm >> k) & 1 ? v : (m_u32x4){0u, 0u, 0u, 0u};
    }
}

// ---------------------------------------------------------------- forward
// out[n,p,i*9+j] = (1/C) sum_c x1[n,p,c] * x2[n % x2_images, p + (i-4, j-4), c]; channels 81..out_ld-1 zeroed.
template <int C, bool OUT_BF16>
__global__ __launch_bounds__(M_T) void corr_fwd_mfma_kernel(const float* __restrict__ x1, int x1_ld,
                                                            const float* __restrict__ x2, int x2_ld, int x2_images, int H,
                                                            int W, int tilesX, int tilesY, float* __restrict__ out,
                                                            int out_ld, int in_bf16) {
    constexpr int YS = OUT_BF16 ? C + 8 : C + 16;
    constexpr int KS = C / 32;                   // MFMA k-steps
    typedef typename std::conditional<OUT_BF16, __bf16, float>::type stage_t;   // staged in the output's type
    __shared__ __attribute__((aligned(16))) __bf16 ys[MHP * YS];
    __shared__ __attribute__((aligned(16))) stage_t stage[MT_H * MT_W * M_OSTR];
    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    const int gy = ty * MT_H + r, gx = tx * MT_W + p;
    const bool inside = gy < H && gx < W;

    // this lane's x1 operand: pixel p of row r, channels 32s + 8g .. +7   (B: k = channel, n = pixel)
    bf16x8 xb[KS];
    {
        const size_t off = inside ? ((size_t)(n * H + gy) * W + gx) * x1_ld : 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (in_bf16) {                                    // (uniform)
                m_u32x4 v = *reinterpret_cast<const m_u32x4*>(reinterpret_cast<const __bf16*>(x1) + off + 32 * s + 8 * g);
                if (!inside) v = (m_u32x4){0u, 0u, 0u, 0u};
                xb[s] = __builtin_bit_cast(bf16x8, v);
            } else {
                float4 a = ld4(x1 + off + 32 * s + 8 * g), b = ld4(x1 + off + 32 * s + 8 * g + 4);
                if (!inside) { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
                xb[s] = __builtin_bit_cast(bf16x8, cvt_f8_bf16(a, b));
            }
        }
    }
    m_stage_halo<C, YS>(x2, x2_ld, n % x2_images, H, W, ty * MT_H, tx * MT_W, ys, in_bf16);
    stage_t* srow = stage + (r * MT_W) * M_OSTR;
    if (lane < 16) { srow[lane * M_OSTR + 81] = (stage_t)0.f; srow[lane * M_OSTR + 82] = (stage_t)0.f; srow[lane * M_OSTR + 83] = (stage_t)0.f; }
    __syncthreads();

    const float inv = 1.f / (float)C;
#pragma unroll 1
    for (int i = 0; i < MN; ++i) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // A: m = halo pixel qb*8 + (lane & 15) of halo row r + i, k = channel
            const __bf16* arow = ys + ((r + i) * MHW + qb * 8 + p) * YS + 8 * g;
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(arow + 32 * s), xb[s], acc, 0,
                                                              0, 0);
            // D[m = 4g + e][n = p]: halo column q = qb*8 + 4g + e, displacement j = q - p.
            // qb = 0 serves q <= 15, qb = 1 the columns 16..23 (its m >= 8); both only inside the band 0 <= j <= 8.
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * g + e;
                const int j = qb * 8 + m - p;
                if (j >= 0 && j <= 8 && (qb == 0 || m >= 8)) srow[p * M_OSTR + i * MN + j] = (stage_t)(acc[e] * inv);
            }
        }
    }
    __syncthreads();
    // row r of the tile: 16 pixels x out_ld channels, written as whole 16-byte pieces
    if (gy >= H) return;
    if constexpr (OUT_BF16) {
        __bf16* o16 = reinterpret_cast<__bf16*>(out);
        const int ppp = out_ld / 8;              // pieces per pixel (out_ld % 8 == 0 checked on the host)
        for (int item = lane; item < MT_W * ppp; item += 64) {
            const int px = item / ppp, q = item - px * ppp;
            if (tx * MT_W + px >= W) continue;
            // staged row: 84 bf16 = 168 B per pixel, so pieces are read as 8-byte halves (8-B aligned)
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 lo = {0u, 0u}, hi = {0u, 0u};
            if (8 * q < M_OSTR) lo = *reinterpret_cast<const u2*>(srow + px * M_OSTR + 8 * q);
            if (8 * q + 4 < M_OSTR) hi = *reinterpret_cast<const u2*>(srow + px * M_OSTR + 8 * q + 4);
            *reinterpret_cast<m_u32x4*>(o16 + ((size_t)(n * H + gy) * W + tx * MT_W + px) * out_ld + 8 * q) =
                (m_u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    } else {
        const int ppp = out_ld / 4;
        for (int item = lane; item < MT_W * ppp; item += 64) {
            const int px = item / ppp, q = item - px * ppp;
            if (tx * MT_W + px >= W) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * q < M_OSTR) v = ld4(srow + px * M_OSTR + 4 * q);
            st4(out + ((size_t)(n * H + gy) * W + tx * MT_W + px) * out_ld + 4 * q, v);
        }
    }
}

// ---------------------------------------------------------------- forward, strip form (bf16-stored inputs)
// The tile kernel above stages a 16 x 24-pixel halo for 8 x 16 pixels - three times its pixels - and does so with nothing
// else to do: stage, barrier, compute, barrier, store.  Here a workgroup walks DOWN a strip of tiles (one 16-pixel column
// block of one image, a segment of its tile rows): the halo rows live in LDS as a ring of 16 rows, a step fetches only the 8
// new ones (1.5x the tile's pixels), and they - and the next tile's x1 operand - are fetched into registers while the current
// tile is computed.  Row hy of tile ty sits in ring slot (hy + 8 (ty & 1)) & 15: the lower half of a tile's halo is the upper
// half of the next one's.
constexpr int MS_SEG = 4;                       // strips are cut into this many vertical segments (jobs = N x tilesX x MS_SEG)
constexpr int MS_MAXWG = 512;                   // two workgroups per CU

template <int C, bool OUT_BF16>
__global__ __launch_bounds__(M_T) void corr_fwd_strip_kernel(const __bf16* __restrict__ x1, int x1_ld,
                                                             const __bf16* __restrict__ x2, int x2_ld, int x2_images, int H,
                                                             int W, int tilesX, int tilesY, int njobs,
                                                             float* __restrict__ out, int out_ld) {
    constexpr int YS = OUT_BF16 ? C + 8 : C + 16;
    constexpr int KS = C / 32;
    constexpr int PPP = C / 8;                   // 16-byte pieces per pixel
    constexpr int HITEMS = 8 * MHW * PPP;        // pieces of 8 halo rows
    constexpr int HPER = (HITEMS + M_T - 1) / M_T;
    typedef typename std::conditional<OUT_BF16, __bf16, float>::type stage_t;
    __shared__ __attribute__((aligned(16))) __bf16 ys[MHP * YS];
    __shared__ __attribute__((aligned(16))) stage_t stage[MT_H * MT_W * M_OSTR];
    const int tid = threadIdx.x;
    const int lane = tid & 63, r = tid >> 6;
    const int p = lane & 15, g = lane >> 4;
    stage_t* srow = stage + (r * MT_W) * M_OSTR;
    if (lane < 16) { srow[lane * M_OSTR + 81] = (stage_t)0.f; srow[lane * M_OSTR + 82] = (stage_t)0.f; srow[lane * M_OSTR + 83] = (stage_t)0.f; }
    const float inv = 1.f / (float)C;
    const int seg_rows = (tilesY + MS_SEG - 1) / MS_SEG;

    m_u32x4 hv[HPER], xv[KS];
    unsigned hokm = 0;
    bool xin = false;
    // halo rows 8 half .. 8 half + 7 of tile (n2, ty, tx) -> registers (clamped addresses; masked at commit)
    auto fetch_rows = [&](int n2, int ty, int tx, int half) {
        hokm = 0;
        int tid_o = tid;                                      // (opaque copy: keeps the per-piece coordinates out of hoisted registers)
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int k = 0; k < HPER; ++k) {
            const int item = tid_o + k * M_T;
            const int hp = item / PPP, q = item - hp * PPP;
            const int hyl = hp / MHW, hx = hp - hyl * MHW;
            const int gy = ty * MT_H + 8 * half + hyl - MD, gx = tx * MT_W + hx - MD;
            const bool ok = item < HITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            hokm |= (ok ? 1u : 0u) << k;
            hv[k] = *reinterpret_cast<const m_u32x4*>(x2 + (ok ? ((size_t)(n2 * H + gy) * W + gx) * x2_ld + 8 * q : 0));
        }
    };
    auto commit_rows = [&](int ty, int half) {
#pragma unroll
        for (int k = 0; k < HPER; ++k) {
            const int item = tid + k * M_T;
            const int hp = item / PPP, q = item - hp * PPP;
            const int hyl = hp / MHW, hx = hp - hyl * MHW;
            const int slot = (8 * half + hyl + 8 * (ty & 1)) & 15;
            if (item < HITEMS)
                *reinterpret_cast<m_u32x4*>(ys + (slot * MHW + hx) * YS + 8 * q) = (hokm >> k) & 1 ? hv[k] : (m_u32x4){0u, 0u, 0u, 0u};
        }
    };
    // this lane's x1 operand of tile (n, ty, tx): pixel p of row r, channels 32 s + 8 g .. + 7
    auto fetch_x1 = [&](int n, int ty, int tx) {
        const int gy = ty * MT_H + r, gx = tx * MT_W + p;
        xin = gy < H && gx < W;
        const size_t off = xin ? ((size_t)(n * H + gy) * W + gx) * x1_ld : 0;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) xv[s2] = *reinterpret_cast<const m_u32x4*>(x1 + off + 32 * s2 + 8 * g);
    };

    for (int job = blockIdx.x; job < njobs; job += gridDim.x) {
        int jt = job;
        const int seg = jt % MS_SEG; jt /= MS_SEG;
        const int tx = jt % tilesX;
        const int n = jt / tilesX;
        const int n2 = n % x2_images;
        const int ty0 = seg * seg_rows, ty1 = min(ty0 + seg_rows, tilesY);
        if (ty0 >= ty1) continue;                             // (uniform)
        // the segment's first tile: both halves of its halo
        __syncthreads();                                      // the previous job's last reads of ys
        fetch_rows(n2, ty0, tx, 0);
        commit_rows(ty0, 0);
        fetch_rows(n2, ty0, tx, 1);
        fetch_x1(n, ty0, tx);
        for (int ty = ty0; ty < ty1; ++ty) {
            bf16x8 xb[KS];
            commit_rows(ty, 1);
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) xb[s2] = __builtin_bit_cast(bf16x8, xin ? xv[s2] : (m_u32x4){0u, 0u, 0u, 0u});
            __syncthreads();                                  // the tile's 16 halo rows are in the ring
            if (ty + 1 < ty1) { fetch_rows(n2, ty + 1, tx, 1); fetch_x1(n, ty + 1, tx); }
            const int rot = 8 * (ty & 1);
#pragma unroll 1
            for (int i = 0; i < MN; ++i) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    // A: m = halo pixel qb*8 + (lane & 15) of halo row r + i, k = channel
                    const __bf16* arow = ys + ((((r + i) + rot) & 15) * MHW + qb * 8 + p) * YS + 8 * g;
#pragma unroll
                    for (int s2 = 0; s2 < KS; ++s2)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(arow + 32 * s2), xb[s2], acc,
                                                                      0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {             // D[m = 4g + e][n = p]: halo column q = qb*8 + m, displacement j = q - p
                        const int m = 4 * g + e;
                        const int j = qb * 8 + m - p;
                        if (j >= 0 && j <= 8 && (qb == 0 || m >= 8)) srow[p * M_OSTR + i * MN + j] = (stage_t)(acc[e] * inv);
                    }
                }
            }
            // row r of the tile: 16 pixels x out_ld channels as whole 16-byte pieces (srow is this wave's own)
            const int gy = ty * MT_H + r;
            if (gy < H) {
                if constexpr (OUT_BF16) {
                    __bf16* o16 = reinterpret_cast<__bf16*>(out);
                    const int ppp = out_ld / 8;
                    for (int item = lane; item < MT_W * ppp; item += 64) {
                        const int px = item / ppp, q = item - px * ppp;
                        if (tx * MT_W + px >= W) continue;
                        typedef unsigned u2 __attribute__((ext_vector_type(2)));
                        u2 lo = {0u, 0u}, hi = {0u, 0u};
                        if (8 * q < M_OSTR) lo = *reinterpret_cast<const u2*>(srow + px * M_OSTR + 8 * q);
                        if (8 * q + 4 < M_OSTR) hi = *reinterpret_cast<const u2*>(srow + px * M_OSTR + 8 * q + 4);
                        *reinterpret_cast<m_u32x4*>(o16 + ((size_t)(n * H + gy) * W + tx * MT_W + px) * out_ld + 8 * q) =
                            (m_u32x4){lo[0], lo[1], hi[0], hi[1]};
                    }
                } else {
                    const int ppp = out_ld / 4;
                    for (int item = lane; item < MT_W * ppp; item += 64) {
                        const int px = item / ppp, q = item - px * ppp;
                        if (tx * MT_W + px >= W) continue;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (4 * q < M_OSTR) v = ld4(srow + px * M_OSTR + 4 * q);
                        st4(out + ((size_t)(n * H + gy) * W + tx * MT_W + px) * out_ld + 4 * q, v);
                    }
                }
            }
            __syncthreads();                                  // everyone is done with the rows the next commit replaces
        }
    }
}

// ---------------------------------------------------------------- gradients
// WHICH == 1: dx[n,p,c] (+)= (1/C) sum_d dcorr[n,p,d]          * other[n % oi, p + off(d), c]
// WHICH == 2: dx[n,q,c] (+)= (1/C) sum_d dcorr[n,q - off(d),d] * other[n,      q - off(d), c]
// Both are  D[c][p] = sum_{q'} Y[(r+i, q')][c] * B_i[q'][p]  over the 24 (padded to 32) halo columns q' of halo row r+i:
//   WHICH 1: B_i[q'][p] = dcorr[p, i*9 + (q'-p)]                                   (dcorr of the tile's own pixels)
//   WHICH 2: B_i[q'][p] = dcorr[(r+i, q'), (8-i)*9 + 8 - (q'-p)]                   (dcorr of the halo pixel itself)
// inside the band 0 <= q'-p <= 8, zero outside.  A (m = channel, k = q') comes from the [pixel][channel] halo image by
// ds_read_b64_tr_b16; B is gathered element-wise (8 bf16 per lane) from the staged dcorr rows.
template <int C, int WHICH, bool D_BF16>
__global__ __launch_bounds__(M_T) void corr_bwd_mfma_kernel(const float* __restrict__ dcorr, int dcorr_ld,
                                                            const float* __restrict__ other, int other_ld,
                                                            int other_images, int H, int W, int tilesX, int tilesY,
                                                            float* __restrict__ dx, int dx_ld, int dx_coff,
                                                            int accumulate, int other_bf16, int groups, int group_images,
                                                            __bf16* __restrict__ dx16, int dx16_ld, const nvq_corr_addends ad) {
    constexpr int YS = WHICH == 1 ? C + 8 : C + 16;
    constexpr int NCB = C / 16;
    constexpr int DPX = WHICH == 1 ? MT_H * MT_W : MHP;     // staged dcorr pixels: the tile / its halo
    __shared__ __attribute__((aligned(16))) __bf16 ys[MHP * YS];
    __shared__ __attribute__((aligned(16))) __bf16 ds[DPX * M_DSTR];
    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n0 = bt / tilesY;
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // groups > 1 (WHICH == 2): dx of image n0 collects from the images n0 + k * group_images of dcorr / other (the T - 1
    // reference frames of a clip) in one pass: one read-modify-write of dx instead of one per frame, fixed order
#pragma unroll 1
    for (int gi = 0; gi < groups; ++gi) {
    const int n = n0 + gi * group_images;
    if (gi > 0) __syncthreads();                             // everyone is done reading the previous group's tiles

    // ---- stage dcorr (channels 0..95 of each pixel; 12 pieces of 8)
    {
        constexpr int ITEMS = DPX * 12;
        constexpr int PER = (ITEMS + M_T - 1) / M_T;
        const __bf16* d16 = reinterpret_cast<const __bf16*>(dcorr);
        if constexpr (D_BF16) {
            // bf16 dcorr: all of a thread's pieces in flight, then the stores (written load - store per piece, every load is a
            // memory round trip of its own: 9 in a row for the halo image)
            m_u32x4 v[PER];
            unsigned okm = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int item = threadIdx.x + k * M_T;
                const int dp = item / 12, q = item - dp * 12;
                int gy, gx;
                if (WHICH == 1) { gy = ty * MT_H + dp / MT_W; gx = tx * MT_W + dp % MT_W; }
                else { gy = ty * MT_H + dp / MHW - MD; gx = tx * MT_W + dp % MHW - MD; }
                const bool ok = item < ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
                okm |= (ok ? 1u : 0u) << k;
                v[k] = *reinterpret_cast<const m_u32x4*>(d16 + (ok ? ((size_t)(n * H + gy) * W + gx) * dcorr_ld + 8 * q : 0));
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int item = threadIdx.x + k * M_T;
                const int dp = item / 12, q = item - dp * 12;
                if (item < ITEMS)
                    *reinterpret_cast<m_u32x4*>(ds + dp * M_DSTR + 8 * q) = (okm >> k) & 1 ? v[k] : (m_u32x4){0u, 0u, 0u, 0u};
            }
        } else {
#pragma unroll 3
            for (int k = 0; k < PER; ++k) {
                const int item = threadIdx.x + k * M_T;
                if (item >= ITEMS) break;
                const int dp = item / 12, q = item - dp * 12;
                int gy, gx;
                if (WHICH == 1) { gy = ty * MT_H + dp / MT_W; gx = tx * MT_W + dp % MT_W; }
                else { gy = ty * MT_H + dp / MHW - MD; gx = tx * MT_W + dp % MHW - MD; }
                const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
                m_u32x4 v = {0u, 0u, 0u, 0u};
                if (ok) {
                    const size_t idx = ((size_t)(n * H + gy) * W + gx) * dcorr_ld + 8 * q;
                    v = cvt_f8_bf16(ld4(dcorr + idx), ld4(dcorr + idx + 4));
                }
                *reinterpret_cast<m_u32x4*>(ds + dp * M_DSTR + 8 * q) = v;
            }
        }
    }
    m_stage_halo<C, YS>(other, other_ld, n % other_images, H, W, ty * MT_H, tx * MT_W, ys, other_bf16);
    __syncthreads();

    typedef m_s16x4 __attribute__((address_space(3))) * lds_tr_ptr;
    const int trq = p >> 2, trp = p & 3;                     // tr-read role inside the 16-lane group
    const int gq = g < 3 ? g : 2;                            // halo columns 24..31 do not exist; their B rows are zero
    const unsigned short* dsu = reinterpret_cast<const unsigned short*>(ds);
#pragma unroll 1
    for (int i = 0; i < MN; ++i) {
        // B: n = pixel p, k = halo column q' = 8g + t
        unsigned bw[4];
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            unsigned half[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int q = 8 * g + 2 * t2 + h;
                const int j = q - p;
                const bool band = j >= 0 && j <= 8;
                int idx;
                if (WHICH == 1) idx = (r * MT_W + p) * M_DSTR + i * MN + (band ? j : 0);
                else idx = ((r + i) * MHW + (band ? q : 0)) * M_DSTR + (8 - i) * MN + (band ? 8 - j : 0);
                const unsigned v = dsu[idx];
                half[h] = band ? v : 0u;
            }
            bw[t2] = half[0] | (half[1] << 16);
        }
        const bf16x8 bfrag = __builtin_bit_cast(bf16x8, (m_u32x4){bw[0], bw[1], bw[2], bw[3]});
        const __bf16* yrow = ys + ((r + i) * MHW + 8 * gq + trq) * YS + 4 * trp;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            // A: m = channel cb*16 + (lane & 15), k = q' = 8g + e: two transposed 4-pixel x 16-channel reads
            const m_s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(yrow + cb * 16));
            const m_s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(yrow + 4 * YS + cb * 16));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), bfrag, acc[cb], 0, 0, 0);
        }
    }
    }   // groups
    // D[m = channel cb*16 + 4g + e][n = pixel p]
    const int gy = ty * MT_H + r, gx = tx * MT_W + p;
    if (gy >= H || gx >= W) return;
    const float inv = 1.f / (float)C;
    float* op = dx + ((size_t)(n0 * H + gy) * W + gx) * dx_ld + dx_coff + 4 * g;
    float4 old[NCB];
    if (accumulate) {                                        // every read of the read-modify-write before the first store
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) old[cb] = ld4(op + cb * 16);
    }
    // bf16 addends (the other terms of the gradient this pass finishes): raw 8-byte loads, all in flight before the first use
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    u2_t ra[NCB], rb[NCB];
    const size_t apix = (size_t)(n0 * H + gy) * W + gx;
    if (ad.a) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
            ra[cb] = *reinterpret_cast<const u2_t*>(reinterpret_cast<const __bf16*>(ad.a) + apix * ad.a_ld + ad.a_coff + cb * 16 + 4 * g);
    }
    if (ad.b) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
            rb[cb] = *reinterpret_cast<const u2_t*>(reinterpret_cast<const __bf16*>(ad.b) + apix * ad.b_ld + ad.b_coff + cb * 16 + 4 * g);
    }
    auto add_raw = [](float4& v, u2_t w) {
        v.x += __uint_as_float(w[0] << 16); v.y += __uint_as_float(w[0] & 0xffff0000u);
        v.z += __uint_as_float(w[1] << 16); v.w += __uint_as_float(w[1] & 0xffff0000u);
    };
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        float4 v = make_float4(acc[cb][0] * inv, acc[cb][1] * inv, acc[cb][2] * inv, acc[cb][3] * inv);
        if (accumulate) { v.x += old[cb].x; v.y += old[cb].y; v.z += old[cb].z; v.w += old[cb].w; }
        if (ad.a) add_raw(v, ra[cb]);
        if (ad.b) add_raw(v, rb[cb]);
        if (dx16)   // the last pass over an accumulated gradient: the sum leaves as bf16 (dx is only read)
            *reinterpret_cast<bf16x4*>(dx16 + ((size_t)(n0 * H + gy) * W + gx) * dx16_ld + cb * 16 + 4 * g) =
                (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        else
            st4(op + cb * 16, v);
    }
}

// ---------------------------------------------------------------- gradient w.r.t. x1 (WHICH == 1), strip form (bf16 tensors)
// As corr_fwd_strip_kernel: a workgroup walks down a column block; the 16 halo rows of `other` (the centre features) are a
// ring in LDS of which a tile fetches only the 8 new rows, and they and the tile's dcorr rows are fetched into registers under
// the MFMAs of the tile before.  (The tile form stages 16 x 24 halo pixels and 8 x 16 dcorr rows per 128 pixels and does nothing
// else meanwhile.)
template <int C>
__global__ __launch_bounds__(M_T) void corr_bwd1_strip_kernel(const __bf16* __restrict__ dcorr, int dcorr_ld,
                                                              const __bf16* __restrict__ other, int other_ld, int other_images,
                                                              int H, int W, int tilesX, int tilesY, int njobs,
                                                              float* __restrict__ dx, int dx_ld, int dx_coff, int accumulate,
                                                              __bf16* __restrict__ dx16, int dx16_ld, const nvq_corr_addends ad) {
    constexpr int YS = C + 8;
    constexpr int NCB = C / 16;
    constexpr int PPP = C / 8;
    constexpr int HITEMS = 8 * MHW * PPP;        // pieces of 8 halo rows
    constexpr int HPER = (HITEMS + M_T - 1) / M_T;
    constexpr int DITEMS = MT_H * MT_W * 12;     // pieces of the tile's dcorr rows (channels 0..95)
    constexpr int DPER = (DITEMS + M_T - 1) / M_T;
    __shared__ __attribute__((aligned(16))) __bf16 ys[MHP * YS];
    __shared__ __attribute__((aligned(16))) __bf16 ds[MT_H * MT_W * M_DSTR];
    const int tid = threadIdx.x;
    const int lane = tid & 63, r = tid >> 6;
    const int p = lane & 15, g = lane >> 4;
    const int seg_rows = (tilesY + MS_SEG - 1) / MS_SEG;
    const float inv = 1.f / (float)C;

    m_u32x4 hv[HPER], dv[DPER];
    unsigned hokm = 0, dokm = 0;
    auto fetch_rows = [&](int n2, int ty, int tx, int half) {
        hokm = 0;
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int k = 0; k < HPER; ++k) {
            const int item = tid_o + k * M_T;
            const int hp = item / PPP, q = item - hp * PPP;
            const int hyl = hp / MHW, hx = hp - hyl * MHW;
            const int gy = ty * MT_H + 8 * half + hyl - MD, gx = tx * MT_W + hx - MD;
            const bool ok = item < HITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            hokm |= (ok ? 1u : 0u) << k;
            hv[k] = *reinterpret_cast<const m_u32x4*>(other + (ok ? ((size_t)(n2 * H + gy) * W + gx) * other_ld + 8 * q : 0));
        }
    };
    auto commit_rows = [&](int ty, int half) {
#pragma unroll
        for (int k = 0; k < HPER; ++k) {
            const int item = tid + k * M_T;
            const int hp = item / PPP, q = item - hp * PPP;
            const int hyl = hp / MHW, hx = hp - hyl * MHW;
            const int slot = (8 * half + hyl + 8 * (ty & 1)) & 15;
            if (item < HITEMS)
                *reinterpret_cast<m_u32x4*>(ys + (slot * MHW + hx) * YS + 8 * q) = (hokm >> k) & 1 ? hv[k] : (m_u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto fetch_d = [&](int n, int ty, int tx) {
        dokm = 0;
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int k = 0; k < DPER; ++k) {
            const int item = tid_o + k * M_T;
            const int dp = item / 12, q = item - dp * 12;
            const int gy = ty * MT_H + dp / MT_W, gx = tx * MT_W + dp % MT_W;
            const bool ok = item < DITEMS && gy < H && gx < W;
            dokm |= (ok ? 1u : 0u) << k;
            dv[k] = *reinterpret_cast<const m_u32x4*>(dcorr + (ok ? ((size_t)(n * H + gy) * W + gx) * dcorr_ld + 8 * q : 0));
        }
    };
    auto commit_d = [&]() {
#pragma unroll
        for (int k = 0; k < DPER; ++k) {
            const int item = tid + k * M_T;
            const int dp = item / 12, q = item - dp * 12;
            if (item < DITEMS)
                *reinterpret_cast<m_u32x4*>(ds + dp * M_DSTR + 8 * q) = (dokm >> k) & 1 ? dv[k] : (m_u32x4){0u, 0u, 0u, 0u};
        }
    };
    typedef m_s16x4 __attribute__((address_space(3))) * lds_tr_ptr;
    const int trq = p >> 2, trp = p & 3;                     // tr-read role inside the 16-lane group
    const int gq = g < 3 ? g : 2;                            // halo columns 24..31 do not exist; their B rows are zero
    const unsigned short* dsu = reinterpret_cast<const unsigned short*>(ds);

    for (int job = blockIdx.x; job < njobs; job += gridDim.x) {
        int jt = job;
        const int seg = jt % MS_SEG; jt /= MS_SEG;
        const int tx = jt % tilesX;
        const int n = jt / tilesX;
        const int n2 = n % other_images;
        const int ty0 = seg * seg_rows, ty1 = min(ty0 + seg_rows, tilesY);
        if (ty0 >= ty1) continue;                             // (uniform)
        __syncthreads();                                      // the previous job's last reads of ys / ds
        fetch_rows(n2, ty0, tx, 0);
        commit_rows(ty0, 0);
        fetch_rows(n2, ty0, tx, 1);
        fetch_d(n, ty0, tx);
        for (int ty = ty0; ty < ty1; ++ty) {
            commit_rows(ty, 1);
            commit_d();
            __syncthreads();                                  // the tile's halo ring and dcorr rows are in LDS
            if (ty + 1 < ty1) { fetch_rows(n2, ty + 1, tx, 1); fetch_d(n, ty + 1, tx); }
            const int rot = 8 * (ty & 1);
            f32x4 acc[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int i = 0; i < MN; ++i) {
                // B: n = pixel p, k = halo column q' = 8g + t: dcorr[p, i*9 + (q' - p)] inside the band
                unsigned bw[4];
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    unsigned half[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int q = 8 * g + 2 * t2 + h;
                        const int j = q - p;
                        const bool band = j >= 0 && j <= 8;
                        const unsigned v = dsu[(r * MT_W + p) * M_DSTR + i * MN + (band ? j : 0)];
                        half[h] = band ? v : 0u;
                    }
                    bw[t2] = half[0] | (half[1] << 16);
                }
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, (m_u32x4){bw[0], bw[1], bw[2], bw[3]});
                const __bf16* yrow = ys + ((((r + i) + rot) & 15) * MHW + 8 * gq + trq) * YS + 4 * trp;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const m_s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(yrow + cb * 16));
                    const m_s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(yrow + 4 * YS + cb * 16));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), bfrag, acc[cb], 0, 0, 0);
                }
            }
            // D[m = channel cb*16 + 4g + e][n = pixel p]: the epilogue of the tile form
            const int gy = ty * MT_H + r, gx = tx * MT_W + p;
            if (gy < H && gx < W) {
                const size_t apix = (size_t)(n * H + gy) * W + gx;
                float* op = dx + apix * dx_ld + dx_coff + 4 * g;
                float4 old[NCB];
                if (accumulate) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) old[cb] = ld4(op + cb * 16);
                }
                typedef unsigned u2_t __attribute__((ext_vector_type(2)));
                u2_t ra[NCB], rb[NCB];
                if (ad.a) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        ra[cb] = *reinterpret_cast<const u2_t*>(reinterpret_cast<const __bf16*>(ad.a) + apix * ad.a_ld + ad.a_coff + cb * 16 + 4 * g);
                }
                if (ad.b) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        rb[cb] = *reinterpret_cast<const u2_t*>(reinterpret_cast<const __bf16*>(ad.b) + apix * ad.b_ld + ad.b_coff + cb * 16 + 4 * g);
                }
                auto add_raw = [](float4& v, u2_t w) {
                    v.x += __uint_as_float(w[0] << 16); v.y += __uint_as_float(w[0] & 0xffff0000u);
                    v.z += __uint_as_float(w[1] << 16); v.w += __uint_as_float(w[1] & 0xffff0000u);
                };
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    float4 v = make_float4(acc[cb][0] * inv, acc[cb][1] * inv, acc[cb][2] * inv, acc[cb][3] * inv);
                    if (accumulate) { v.x += old[cb].x; v.y += old[cb].y; v.z += old[cb].z; v.w += old[cb].w; }
                    if (ad.a) add_raw(v, ra[cb]);
                    if (ad.b) add_raw(v, rb[cb]);
                    if (dx16)
                        *reinterpret_cast<bf16x4*>(dx16 + apix * dx16_ld + cb * 16 + 4 * g) =
                            (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                    else
                        st4(op + cb * 16, v);
                }
            }
            __syncthreads();                                  // everyone is done with the rows / dcorr the next commit replaces
        }
    }
}

// ---------------------------------------------------------------- gradient w.r.t. x2 (WHICH == 2), prefetching form (bf16)
// The tile form stages 123 KB per (tile, frame group) - the halo of `other` AND of dcorr - with one workgroup per CU and
// nothing in flight meanwhile.  Same tiles, same arithmetic and order here, but a persistent workgroup walks (tile, group)
// steps and fetches the next step's two halos into registers under the current step's MFMAs.  (A ring of halo rows as in
// the strip kernels would need one ring per frame group: 270 KB of LDS.)
template <int C>
__global__ __launch_bounds__(M_T) void corr_bwd2_pf_kernel(const __bf16* __restrict__ dcorr, int dcorr_ld,
                                                           const __bf16* __restrict__ other, int other_ld, int other_images,
                                                           int H, int W, int tilesX, int tilesY, int ntiles,
                                                           float* __restrict__ dx, int dx_ld, int dx_coff, int accumulate,
                                                           int groups, int group_images, __bf16* __restrict__ dx16, int dx16_ld,
                                                           const nvq_corr_addends ad) {
    constexpr int YS = C + 16;
    constexpr int NCB = C / 16;
    constexpr int PPP = C / 8;
    constexpr int YITEMS = MHP * PPP, YPER = YITEMS / M_T;   // exact: 384 * {4, 8} / 512
    constexpr int DITEMS = MHP * 12, DPER = DITEMS / M_T;    // exact: 4608 / 512 = 9
    static_assert(YITEMS % M_T == 0 && DITEMS % M_T == 0, "halo pieces divide evenly");
    __shared__ __attribute__((aligned(16))) __bf16 ys[MHP * YS];
    __shared__ __attribute__((aligned(16))) __bf16 ds[MHP * M_DSTR];
    const int tid = threadIdx.x;
    const int lane = tid & 63, r = tid >> 6;
    const int p = lane & 15, g = lane >> 4;
    const float inv = 1.f / (float)C;

    m_u32x4 yv[YPER], dv[DPER];
    unsigned yokm = 0, dokm = 0;
    auto locate = [&](int tile, int& n0, int& ty, int& tx) {
        int bt = xcd_tile(tile, ntiles);
        tx = bt % tilesX; bt /= tilesX;
        ty = bt % tilesY;
        n0 = bt / tilesY;
    };
    // both halos of image n (origin (ty*8 - 4, tx*16 - 4)) -> registers (clamped addresses; masked at commit)
    auto fetch = [&](int n, int ty, int tx) {
        yokm = 0; dokm = 0;
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const int n2 = n % other_images;
#pragma unroll
        for (int k = 0; k < YPER; ++k) {
            const int item = tid_o + k * M_T;
            const int hp = item / PPP, q = item - hp * PPP;
            const int hy = hp / MHW, hx = hp - hy * MHW;
            const int gy = ty * MT_H + hy - MD, gx = tx * MT_W + hx - MD;
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            yokm |= (ok ? 1u : 0u) << k;
            yv[k] = *reinterpret_cast<const m_u32x4*>(other + (ok ? ((size_t)(n2 * H + gy) * W + gx) * other_ld + 8 * q : 0));
        }
#pragma unroll
        for (int k = 0; k < DPER; ++k) {
            const int item = tid_o + k * M_T;
            const int dp = item / 12, q = item - dp * 12;
            const int gy = ty * MT_H + dp / MHW - MD, gx = tx * MT_W + dp % MHW - MD;
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            dokm |= (ok ? 1u : 0u) << k;
            dv[k] = *reinterpret_cast<const m_u32x4*>(dcorr + (ok ? ((size_t)(n * H + gy) * W + gx) * dcorr_ld + 8 * q : 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < YPER; ++k) {
            const int item = tid + k * M_T;
            const int hp = item / PPP, q = item - hp * PPP;
            *reinterpret_cast<m_u32x4*>(ys + hp * YS + 8 * q) = (yokm >> k) & 1 ? yv[k] : (m_u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < DPER; ++k) {
            const int item = tid + k * M_T;
            const int dp = item / 12, q = item - dp * 12;
            *reinterpret_cast<m_u32x4*>(ds + dp * M_DSTR + 8 * q) = (dokm >> k) & 1 ? dv[k] : (m_u32x4){0u, 0u, 0u, 0u};
        }
    };
    typedef m_s16x4 __attribute__((address_space(3))) * lds_tr_ptr;
    const int trq = p >> 2, trp = p & 3;
    const int gq = g < 3 ? g : 2;
    const unsigned short* dsu = reinterpret_cast<const unsigned short*>(ds);

    int tile = blockIdx.x;
    if (tile < ntiles) { int n0, ty, tx; locate(tile, n0, ty, tx); fetch(n0, ty, tx); }
    for (; tile < ntiles; tile += gridDim.x) {
        int n0, ty, tx;
        locate(tile, n0, ty, tx);
        f32x4 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int gi = 0; gi < groups; ++gi) {
            __syncthreads();                                  // everyone is done reading the previous step's tiles
            commit();
            __syncthreads();
            // the next step's halos: the next frame group of this tile, or the first group of the workgroup's next tile
            if (gi + 1 < groups) {
                fetch(n0 + (gi + 1) * group_images, ty, tx);
            } else if (tile + (int)gridDim.x < ntiles) {
                int n1, ty1, tx1;
                locate(tile + gridDim.x, n1, ty1, tx1);
                fetch(n1, ty1, tx1);
            }
#pragma unroll 1
            for (int i = 0; i < MN; ++i) {
                // B: n = pixel p, k = halo column q' = 8g + t: dcorr[(r + i, q'), (8 - i)*9 + 8 - (q' - p)] inside the band
                unsigned bw[4];
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    unsigned half[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int q = 8 * g + 2 * t2 + h;
                        const int j = q - p;
                        const bool band = j >= 0 && j <= 8;
                        const unsigned v = dsu[((r + i) * MHW + (band ? q : 0)) * M_DSTR + (8 - i) * MN + (band ? 8 - j : 0)];
                        half[h] = band ? v : 0u;
                    }
                    bw[t2] = half[0] | (half[1] << 16);
                }
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, (m_u32x4){bw[0], bw[1], bw[2], bw[3]});
                const __bf16* yrow = ys + ((r + i) * MHW + 8 * gq + trq) * YS + 4 * trp;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const m_s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(yrow + cb * 16));
                    const m_s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(yrow + 4 * YS + cb * 16));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), bfrag, acc[cb], 0, 0, 0);
                }
            }
        }
        // D[m = channel cb*16 + 4g + e][n = pixel p]: the epilogue of the tile form
        const int gy = ty * MT_H + r, gx = tx * MT_W + p;
        if (gy < H && gx < W) {
            const size_t apix = (size_t)(n0 * H + gy) * W + gx;
            float* op = dx + apix * dx_ld + dx_coff + 4 * g;
            float4 old[NCB];
            if (accumulate) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) old[cb] = ld4(op + cb * 16);
            }
            typedef unsigned u2_t __attribute__((ext_vector_type(2)));
            u2_t ra[NCB], rb[NCB];
            if (ad.a) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    ra[cb] = *reinterpret_cast<const u2_t*>(reinterpret_cast<const __bf16*>(ad.a) + apix * ad.a_ld + ad.a_coff + cb * 16 + 4 * g);
            }
            if (ad.b) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    rb[cb] = *reinterpret_cast<const u2_t*>(reinterpret_cast<const __bf16*>(ad.b) + apix * ad.b_ld + ad.b_coff + cb * 16 + 4 * g);
            }
            auto add_raw = [](float4& v, u2_t w) {
                v.x += __uint_as_float(w[0] << 16); v.y += __uint_as_float(w[0] & 0xffff0000u);
                v.z += __uint_as_float(w[1] << 16); v.w += __uint_as_float(w[1] & 0xffff0000u);
            };
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                float4 v = make_float4(acc[cb][0] * inv, acc[cb][1] * inv, acc[cb][2] * inv, acc[cb][3] * inv);
                if (accumulate) { v.x += old[cb].x; v.y += old[cb].y; v.z += old[cb].z; v.w += old[cb].w; }
                if (ad.a) add_raw(v, ra[cb]);
                if (ad.b) add_raw(v, rb[cb]);
                if (dx16)
                    *reinterpret_cast<bf16x4*>(dx16 + apix * dx16_ld + cb * 16 + 4 * g) =
                        (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                else
                    st4(op + cb * 16, v);
            }
        }
    }
}

// ---------------------------------------------------------------- host side (called from motion.hip's entry points)
bool corr_mfma_supported(int C) { return C == 32 || C == 64; }

int corr_forward_mfma(const float* x1, int x1_ld, const float* x2, int x2_ld, int x2_images, int C, int N, int H, int W,
                      float* out, int out_ld, int out_bf16, int in_bf16, hipStream_t s) {
    NVQ_REQUIRE(!in_bf16 || (x1_ld % 8 == 0 && x2_ld % 8 == 0), "correlation_forward(bf16): bf16 inputs need ld %% 8 == 0");
    NVQ_REQUIRE(out_ld >= 84 && out_ld % (out_bf16 ? 8 : 4) == 0, "correlation_forward(bf16): out_ld %d", out_ld);
    const int tilesX = (W + MT_W - 1) / MT_W, tilesY = (H + MT_H - 1) / MT_H;
    if (in_bf16) {                                            // bf16-stored features: the strip form (1.12 -> 1.03 ms at 16 x 540 x 960)
        const int njobs = N * tilesX * MS_SEG;
        int nwg = njobs < MS_MAXWG ? njobs : MS_MAXWG;
#define NVQ_CS(CC, OB) \
    hipLaunchKernelGGL((corr_fwd_strip_kernel<CC, OB>), dim3(nwg), dim3(M_T), 0, s, reinterpret_cast<const __bf16*>(x1), x1_ld, \
                       reinterpret_cast<const __bf16*>(x2), x2_ld, x2_images, H, W, tilesX, tilesY, njobs, out, out_ld)
        if (C == 64) { if (out_bf16) NVQ_CS(64, true); else NVQ_CS(64, false); }
        else { if (out_bf16) NVQ_CS(32, true); else NVQ_CS(32, false); }
#undef NVQ_CS
        return check_launch("correlation_forward(bf16, strip)");
    }
    const dim3 grid((unsigned)((long)tilesX * tilesY * N));
#define NVQ_CF(CC, OB) \
    hipLaunchKernelGGL((corr_fwd_mfma_kernel<CC, OB>), grid, dim3(M_T), 0, s, x1, x1_ld, x2, x2_ld, x2_images, H, W, tilesX, tilesY, out, out_ld, in_bf16)
    if (C == 64) { if (out_bf16) NVQ_CF(64, true); else NVQ_CF(64, false); }
    else { if (out_bf16) NVQ_CF(32, true); else NVQ_CF(32, false); }
#undef NVQ_CF
    return check_launch("correlation_forward(bf16)");
}

int corr_backward_mfma(int which, const float* dcorr, int dcorr_ld, int dcorr_bf16, const float* other, int other_ld,
                       int other_images, int C, int N, int H, int W, float* dx, int dx_ld, int dx_coff, int accumulate,
                       int other_bf16, int groups, float* dx16, int dx16_ld, const nvq_corr_addends* addends, hipStream_t s) {
    const nvq_corr_addends ad = addends ? *addends : nvq_corr_addends{nullptr, 0, 0, nullptr, 0, 0};
    NVQ_REQUIRE((!ad.a || (ad.a_ld % 4 == 0 && ad.a_coff % 4 == 0 && (reinterpret_cast<uintptr_t>(ad.a) & 7) == 0)) &&
                    (!ad.b || (ad.b_ld % 4 == 0 && ad.b_coff % 4 == 0 && (reinterpret_cast<uintptr_t>(ad.b) & 7) == 0)),
                "correlation_backward(bf16): addends need 8-byte aligned rows");
    NVQ_REQUIRE(!other_bf16 || other_ld % 8 == 0, "correlation_backward(bf16): a bf16 `other` needs ld %% 8 == 0");
    NVQ_REQUIRE(dcorr_ld >= 96 && dcorr_ld % (dcorr_bf16 ? 8 : 4) == 0,
                "correlation_backward(bf16): dcorr must be readable up to channel 96 (ld %d)", dcorr_ld);
    const int tilesX = (W + MT_W - 1) / MT_W, tilesY = (H + MT_H - 1) / MT_H;
    if (which == 1 && dcorr_bf16 && other_bf16 && dcorr_ld % 8 == 0) {          // all-bf16 tensors: the strip form
        const int njobs = N * tilesX * MS_SEG;
        const int nwg = njobs < MS_MAXWG ? njobs : MS_MAXWG;
#define NVQ_CB1(CC) hipLaunchKernelGGL(corr_bwd1_strip_kernel<CC>, dim3(nwg), dim3(M_T), 0, s, reinterpret_cast<const __bf16*>(dcorr), \
                                       dcorr_ld, reinterpret_cast<const __bf16*>(other), other_ld, other_images, H, W, tilesX, tilesY, \
                                       njobs, dx, dx_ld, dx_coff, accumulate, reinterpret_cast<__bf16*>(dx16), dx16_ld, ad)
        if (C == 64) NVQ_CB1(64); else NVQ_CB1(32);
#undef NVQ_CB1
        return check_launch("correlation_backward(bf16, strip)");
    }
    if (which == 2 && dcorr_bf16 && other_bf16 && dcorr_ld % 8 == 0) {          // all-bf16 tensors: the prefetching form
        const int ntiles = tilesX * tilesY * N;
        int nwg = ntiles < 256 ? ntiles : 256;                // one workgroup per CU (135 KB of LDS)
        if (nwg >= 8) nwg &= ~7;                              // multiple of the XCD count, see xcd_tile()
#define NVQ_CB2P(CC) hipLaunchKernelGGL(corr_bwd2_pf_kernel<CC>, dim3(nwg), dim3(M_T), 0, s, reinterpret_cast<const __bf16*>(dcorr), \
                                        dcorr_ld, reinterpret_cast<const __bf16*>(other), other_ld, other_images, H, W, tilesX, tilesY, \
                                        ntiles, dx, dx_ld, dx_coff, accumulate, groups, N, reinterpret_cast<__bf16*>(dx16), dx16_ld, ad)
        if (C == 64) NVQ_CB2P(64); else NVQ_CB2P(32);
#undef NVQ_CB2P
        return check_launch("correlation_backward(bf16, prefetching)");
    }
    const dim3 grid((unsigned)((long)tilesX * tilesY * N));
#define NVQ_CB(CC, WH, DB) \
    hipLaunchKernelGGL((corr_bwd_mfma_kernel<CC, WH, DB>), grid, dim3(M_T), 0, s, dcorr, dcorr_ld, other, other_ld, other_images, H, W, tilesX, tilesY, dx, dx_ld, dx_coff, accumulate, other_bf16, groups, N, reinterpret_cast<__bf16*>(dx16), dx16_ld, ad)
#define NVQ_CB2(CC) \
    do { if (which == 1) { if (dcorr_bf16) NVQ_CB(CC, 1, true); else NVQ_CB(CC, 1, false); } \
         else { if (dcorr_bf16) NVQ_CB(CC, 2, true); else NVQ_CB(CC, 2, false); } } while (0)
    if (C == 64) NVQ_CB2(64); else NVQ_CB2(32);
#undef NVQ_CB2
#undef NVQ_CB
    return check_launch("correlation_backward(bf16)");
}

}  // namespace nvq
