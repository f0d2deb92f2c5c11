// NVQ_MATH_BF16 convolution kernels: operands are bf16 in LDS (activations stored as fp32 are rounded while
// they are staged, activations stored as bf16 are copied), v_mfma_f32_16x16x32_bf16 with fp32 accumulation,
// fp32 epilogue arithmetic, output stored as fp32 or bf16 per tensor (nvq_conv_desc::*_bf16).
//
// At 16x the fp32 MFMA rate these kernels are HBM-bound, so the structure is built around keeping loads in
// flight: the next K chunk / pixel tile is fetched into registers while the current one is consumed from LDS.
// Rules the prefetch obeys (each one was a measured stall):
//   * every prefetch load is unconditional (invalid pieces read element 0 and are masked at commit time):
//     branches around loads serialise them;
//   * nothing between the prefetch and the next commit may USE a loaded value (not even a select);
//   * register pieces are ext_vector types: HIP's uint4 struct is not scalar-replaced and lands in scratch;
//   * the packed weight slab is padded to a whole number of 16-byte pieces per thread.
#include "conv_common.h"

namespace nvq {

// Diagnostic switch of libnvq_debug.so (tools/kernel_phases.py; -DNVQ_DEBUG_TOOLS): 1 = skip the MFMA section, 2 = skip the
// per-chunk global loads after the first chunk.  Results are wrong in both modes.  The shipped library has no such state:
// there the kernels' `dbg` argument is the literal 0.
#ifdef NVQ_DEBUG_TOOLS
static int g_debug_mode = 0;
void set_conv_debug_mode(int m) { g_debug_mode = m; }
#else
constexpr int g_debug_mode = 0;
#endif

constexpr int XSB = 48;    // bf16 per staged pixel: 32 data + 16 pad (96 B: the 16 pixels x 4 k-groups of one
                           // ds_read_b128 / ds_read_b64_tr_b16 instruction land on distinct 16-B slots)

// ---------------------------------------------------------------- weight packing (bf16)
// wpack[cz][kc][tap][g][n][j] (g = 0..3, n = 0..NT-1, j = 0..7) = bf16(W[cout = cz*NT + n][ch = kc*32 + 8g + j][tap])
__device__ __forceinline__ __bf16 pack_bf16_elem(const float* __restrict__ w, int cout_w, int cin_w, int taps, int transpose,
                                                 int cout_keep, int NT, int nkc, long idx) {
    const int stride = ws_stride_halfs(taps, NT);
    long t = idx % stride;
    const long slab = idx / stride;
    if (t >= (long)taps * 4 * NT * 8) return (__bf16)0.f;
    const int j = t & 7; t >>= 3;
    const int n = t % NT; t /= NT;
    const int g = t & 3; t >>= 2;
    const int tap = (int)t;
    const int kc = slab % nkc;
    const int cz = (int)(slab / nkc);
    const int co = cz * NT + n;
    const int ch = kc * KCB + 8 * g + j;
    float v = 0.f;
    if (!transpose) {
        if (co < cout_w && ch < cin_w) v = w[((long)co * cin_w + ch) * taps + tap];
    } else {
        if (co < cout_keep && ch < cout_w) v = w[((long)ch * cin_w + co) * taps + (taps - 1 - tap)];
    }
    return (__bf16)v;
}

__global__ void pack_bf16_kernel(const float* __restrict__ w, int cout_w, int cin_w, int taps, int transpose,
                                 int cout_keep, int NT, int ncz, int nkc, __bf16* __restrict__ wp) {
    const long total = (long)ncz * nkc * ws_stride_halfs(taps, NT);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x)
        wp[idx] = pack_bf16_elem(w, cout_w, cin_w, taps, transpose, cout_keep, NT, nkc, idx);
}

// grid (blocks, jobs): nvq_conv_pack_batch
__global__ void pack_batch_bf16_kernel(const PackJobTable t) {
    const PackJobDev j = t.j[blockIdx.y];
    __bf16* wp = reinterpret_cast<__bf16*>(j.wp);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < j.total; idx += (long)gridDim.x * blockDim.x)
        wp[idx] = pack_bf16_elem(j.w, j.cout_w, j.cin_w, j.taps, j.transpose, j.cout_keep, j.NT, j.nkc, idx);
}

// ---------------------------------------------------------------- forward / input gradient
// INB: the input activation tensor is stored as bf16.
// NW: waves per workgroup; each wave owns two tile rows, so the tile is 2*NW x 32 pixels.  NW = 8 (16 rows) stages the
// weight slab once per 512 pixels instead of 256 and carries 19 % halo instead of 33 %: the kernel is bound by its
// staging side, and for cout <= 32 the slab is as many bytes as the activations of a 256-pixel tile.
// CS = 2 (NW = 8): the workgroup's waves split in two halves that compute output channels [0, NT) and [NT, 2 NT) of the SAME
// 8 x 32 pixel tile (a 64-channel 3x3 conv at the register budget and the four waves per SIMD of the 32-channel kernel: with
// all 64 channels in one wave the accumulators and the 37 KB weight slab's prefetch registers leave two waves per SIMD).
// TS > 0: the upsampler tail (PixelShuffleUpsampler.conv efficient_layers.py:94-100 + PixelShuffle(TS) :101-106 + the bicubic
// skip and the clamp of super_resolution.py:378-382) as this kernel's epilogue: the tile's 3 TS^2 conv outputs go through LDS,
// are read back as TS x TS pixel blocks, and the frame is written (no intermediate tensor); `tail` says where.
struct TailArgs {
    const float* frames;   // fp32 NCHW clip (B, T, Cimg, H, W)
    float* out;            // fp32 NCHW (B, Cimg, H*TS, W*TS)
    unsigned char* pass;   // 1 where 0 <= pre-clamp <= 1
    int T, t_center, Cimg;
};
// Keys cubic convolution, A = -0.75 (PyTorch upsample_bicubic2d, align_corners=False); the same arithmetic as
// shuffle_bicubic_clamp_kernel (upsample.hip)
__device__ __forceinline__ float tcc1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float tcc2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__device__ __forceinline__ void tail_cubic_taps(int dst, float scale, int size, int idx[4], float w[4]) {
    const float real = scale * ((float)dst + 0.5f) - 0.5f;
    const float fl = floorf(real);
    const float t = real - fl;
    const int i0 = (int)fl;
    w[0] = tcc2(t + 1.f);
    w[1] = tcc1(t);
    w[2] = tcc1(1.f - t);
    w[3] = tcc2((1.f - t) + 1.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) idx[k] = min(max(i0 - 1 + k, 0), size - 1);
}

template <int NB, int KS, bool INB, int NW = 4, int CS = 1, int TS = 0>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 4) void conv_bf16_kernel(const nvq_conv_desc d, int tilesX,
                                                                               int tilesY, int nkc, int vec_ok, int dbg,
                                                                               const TailArgs tail) {
    constexpr int NT = NB * 16;                               // output channels of one wave
    constexpr int NTW = NT * CS;                              // ... of the workgroup (= the packed slab's width)
    constexpr int NTHR = 64 * NW;
    constexpr int TH_ = 2 * NW / CS;
    constexpr int HALO = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int HW_ = TW + 2 * HALO;
    constexpr int HH_ = TH_ + 2 * HALO;
    constexpr int NPIX = HW_ * HH_;
    constexpr int WS_HALFS = ws_stride_halfs(TAPS, NTW);    // padded slab (see ws_stride_halfs)
    constexpr int XITEMS = NPIX * 4;                          // (pixel, 8-channel group) pieces per chunk
    constexpr int XPER = (XITEMS + NTHR - 1) / NTHR;
    constexpr int WTHR = NTHR;                                // every thread moves its share of the weight slab
    constexpr int WPIECES = TAPS * 4 * NTW;                   // its 16-byte pieces that carry weights (the packed slab is padded)
    constexpr int WPER = (WPIECES + WTHR - 1) / WTHR;         // pieces per thread (the last one only where tid + k*WTHR < WPIECES)
    constexpr int XREGS = INB ? 1 : 2;                        // 16-byte registers per 8-channel piece
    constexpr int CT_K = (TAPS / 2) * 4 * NTW / WTHR;         // the centre tap's 4 * NTW pieces: register index and
    constexpr int CT_N = 4 * NTW;                             // thread count (NT = 16: pieces 256..319, 32: 512..639, 64: 1024..1279)
    static_assert(((TAPS / 2) * 4 * NTW) % WTHR == 0 && CT_N <= 256, "the centre tap starts a WTHR-piece row");
    static_assert(CS == 1 || (CS == 2 && NW == 8 && NB == 2), "channel-split variant: 8 waves, 2 x 32 channels");

    __shared__ __attribute__((aligned(16))) __bf16 lds[NPIX * XSB + WS_HALFS];
    __bf16* xs = lds;
    __bf16* ws = lds + NPIX * XSB;
    const int kcl = KS == 3 ? d.center_cin / KCB : 0;         // leading chunks that only have a centre tap

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // (wave-uniform, so kept in scalar registers: as vector values the channel-split variant spilled them around its main loop)
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = CS == 1 ? wid : wid & (NW / CS - 1);     // the wave's row pair in the tile
    const int half = CS == 1 ? 0 : wid >> 2;                  // ... and its half of the workgroup's output channels
    const int c = lane & 15;
    const int g = lane >> 4;

    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int cz = blockIdx.y * CS + half;                    // in units of NT channels (epilogue)
    const int H = d.h, W = d.w;

    f32x4 acc[NB][4];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const __bf16* wp_base = reinterpret_cast<const __bf16*>(d.wpack) + (size_t)blockIdx.y * nkc * WS_HALFS;
    const float* in32 = d.in + d.in_coff;
    const __bf16* in16 = reinterpret_cast<const __bf16*>(d.in) + d.in_coff;

    // Per-thread element offsets of the activation pieces (chunk independent).  Out-of-image pieces point at
    // element 0 and are zeroed by a select, so every load below is unconditional (no branches, no spills).
    unsigned xoff[XPER];
    bool xok[XPER];
#pragma unroll
    for (int k = 0; k < XPER; ++k) {
        const int item = tid + k * NTHR;
        const int hp = item >> 2;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int gy = ty * TH_ + hy - HALO, gx = tx * TW + hx - HALO;
        xok[k] = item < XITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        // pixel offset without the channel group (added per chunk); out-of-image pieces use pixel 0 and are masked at commit.
        // Slice-planar input: the offset inside a 32-channel plane; chunks of the leading in_ld-channel tensor shift it.
        xoff[k] = xok[k] ? (unsigned)(((size_t)(n * H + gy) * W + gx) * (d.in_plane ? 32 : d.in_ld)) : 0u;
    }
    const int nk0 = d.in_plane ? d.in_ld >> 5 : 0x7fffffff;   // chunks that live in the leading tensor (all, if interleaved)
    const int sh0 = d.in_plane ? __ffs(d.in_ld >> 5) - 1 : 0; // log2(in_ld / 32)
    // Whole halo inside the image and a whole number of chunks: no piece of this tile is ever masked, and the commit can
    // store the registers as they are (with two waves per SIMD every VALU instruction of the staging code competes with
    // the other wave's MFMA issue, so the selects are kept out of the common case).
    const bool interior = ty * TH_ >= HALO && tx * TW >= HALO && ty * TH_ + TH_ + HALO <= H && tx * TW + TW + HALO <= W &&
                          d.cin % KCB == 0;
    const int chg = 8 * (tid & 3);                            // channel group of this thread's pieces (256 % 4 == 0)
    u32x4 xr[XPER][XREGS];
    u32x4 wr[WPER];

    bool cv0 = false, cv1 = false;                            // channel validity of the chunk held in xr
    // `light`: the chunk only has a centre tap (kc < kcl), one tap's 4 * NTW pieces of the slab.  A LITERAL at every call
    // site: as a run-time condition it makes the weight registers a phi of "loaded" and "kept" values, which the compiler
    // resolves by loading into temporaries and copying them behind a vmcnt(0) - right after the loads, in front of the MFMA
    // section: the whole prefetch exposed.  For the same reason every load is unconditional for every thread (a piece past
    // the slab reads piece 0 and is not committed).
    auto fetch = [&](int kc, bool light) {                    // raw loads only: nothing here may USE a loaded value
        const int ch = kc * KCB + chg;
        cv0 = ch < d.cin;
        cv1 = ch + 4 < d.cin;
        // a channel group past the slice reads the pixel's group 0 instead (masked at commit)
        const bool lead = kc < nk0;                           // uniform
        const int sh = lead ? sh0 : 0;
        const unsigned cbase = lead ? (unsigned)kc * KCB : (unsigned)kc * d.in_plane;
        const unsigned o0 = cv0 ? cbase + chg : 0u, o1 = cv1 ? cbase + chg + 4 : 0u;
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            if constexpr (INB) {
                xr[k][0] = *reinterpret_cast<const u32x4*>(in16 + ((xoff[k] << sh) + o0));
            } else {
                xr[k][0] = *reinterpret_cast<const u32x4*>(in32 + ((xoff[k] << sh) + o0));
                xr[k][XREGS - 1] = *reinterpret_cast<const u32x4*>(in32 + ((xoff[k] << sh) + o1));
            }
        }
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(wp_base + (size_t)kc * WS_HALFS);
        if (light) {
            wr[CT_K] = wsrc[(tid < CT_N ? tid : 0) + CT_K * WTHR];
        } else {
#pragma unroll
            // (a piece past the slab: every such lane reads piece 0 - ONE 16-byte request per wave; what a workgroup ingests per
            // chunk is what bounds these kernels, and distinct dummy addresses were 10 % of it)
            for (int k = 0; k < WPER; ++k) wr[k] = wsrc[tid + k * WTHR < WPIECES ? tid + k * WTHR : 0];
        }
    };
    auto commit = [&](int kc, bool light) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (interior) {                                      // workgroup-uniform
#pragma unroll
            for (int k = 0; k < XPER; ++k) {
                const int item = tid + k * NTHR;
                if (item < XITEMS) {
                    __bf16* dst = xs + (item >> 2) * XSB + 8 * (item & 3);
                    if constexpr (INB) *reinterpret_cast<u32x4*>(dst) = xr[k][0];
                    else *reinterpret_cast<bf16x8*>(dst) = cvt8(as_f4(xr[k][0]), as_f4(xr[k][XREGS - 1]));
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < XPER; ++k) {
                const int item = tid + k * NTHR;
                const bool v0 = xok[k] && cv0, v1 = xok[k] && cv1;
                if (item < XITEMS) {
                    __bf16* dst = xs + (item >> 2) * XSB + 8 * (item & 3);
                    if constexpr (INB)                       // cin % 8 == 0: a piece is valid or invalid as a whole
                        *reinterpret_cast<u32x4*>(dst) = v0 ? xr[k][0] : z;
                    else
                        *reinterpret_cast<bf16x8*>(dst) = cvt8(as_f4(v0 ? xr[k][0] : z), as_f4(v1 ? xr[k][XREGS - 1] : z));
                }
            }
        }
        if (light) {
            if (tid < CT_N) reinterpret_cast<u32x4*>(ws)[tid + CT_K * WTHR] = wr[CT_K];
        } else {
#pragma unroll
            for (int k = 0; k < WPER; ++k)
                if ((k + 1) * WTHR <= WPIECES || tid + k * WTHR < WPIECES) reinterpret_cast<u32x4*>(ws)[tid + k * WTHR] = wr[k];
        }
    };

    auto ldP = [&](int rr, int xh, int dx) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(xs + ((2 * wave + rr) * HW_ + xh * 16 + c + dx) * XSB + 8 * g);
    };
    auto ldW = [&](int tap, int cb) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(ws + ((tap * 4 + g) * NTW + half * NT + cb * 16 + c) * 8);
    };
    // chunk kc: registers -> LDS, chunk kc + 1: memory -> registers.  cl / fl: chunk kc / kc + 1 is a centre-tap-only chunk
    // (literals, see fetch)
    auto stage_chunk = [&](int kc, bool cl, bool fl) {
        __syncthreads();
        commit(kc, cl);
        __syncthreads();
        if (kc + 1 < nkc && !(dbg & 2)) fetch(kc + 1, fl);
    };
    auto center_stage = [&]() {                               // the MFMAs of a centre-tap-only chunk
        bf16x8 p0[2], p1[2], wc[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) wc[cb] = ldW(TAPS / 2, cb);
#pragma unroll
        for (int xh = 0; xh < 2; ++xh) { p0[xh] = ldP(1, xh, 1); p1[xh] = ldP(2, xh, 1); }
#pragma unroll
        for (int xh = 0; xh < 2; ++xh)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                acc[cb][xh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[cb], p0[xh], acc[cb][xh], 0, 0, 0);
                acc[cb][2 + xh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[cb], p1[xh], acc[cb][2 + xh], 0, 0, 0);
            }
    };

    // CS = 2: a half whose 32 channels lie beyond the stored ones (cout_store = 96: the second half of the second slab of the
    // flow net's 128 -> 81 input gradient) stages with the others but leaves the matrix cores to the other waves
    const bool dead = CS == 2 && cz * NT >= d.cout_store;     // (wave-uniform)
    int kc = 0;
    // Leading chunks whose weights are zero outside the centre tap (nvq_conv_desc::center_cin): one stage instead of nine.
    // Loops of their own - as a branch inside the main loop the same code cost the main path 60 VGPRs.
    if constexpr (KS == 3) {
        if (kcl > 0) {
            fetch(0, true);
            for (; kc + 1 < kcl; ++kc) {
                stage_chunk(kc, true, true);
                if (!(dbg & 1) && !dead) center_stage();
            }
            stage_chunk(kc, true, false);                    // the last of them fetches a full slab
            if (!(dbg & 1) && !dead) center_stage();
            ++kc;
        } else {
            fetch(0, false);
        }
    } else {
        fetch(0, false);
    }
    for (; kc < nkc; ++kc) {
        stage_chunk(kc, false, false);
        if ((dbg & 1) || dead) continue;
        // Fragment reads software-pipelined against the MFMAs (the compiler otherwise emits read -> lgkmcnt(0) ->
        // 4 MFMAs, exposing the LDS latency 2*TAPS times per chunk).  Stages run dx-major, dy-minor: going from dy to
        // dy+1 the wave's upper output row reuses the fragments of the lower one, so a stage needs only the two
        // fragments of one new halo row; those, and the next stage's weights, are read one stage ahead.
        bf16x8 lo[2], hi[2], wa[NB], wn[NB];
        lo[0] = ldP(0, 0, 0); lo[1] = ldP(0, 1, 0);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) wa[cb] = ldW(0, cb);
        hi[0] = ldP(1, 0, 0); hi[1] = ldP(1, 1, 0);
#pragma unroll
        for (int s = 0; s < TAPS; ++s) {
            const int dy = s % KS;
            const bool last = s + 1 == TAPS;
            const int ndx = (s + 1) / KS, ndy = (s + 1) % KS;
            const bool same_dx = !last && ndy != 0;
            bf16x8 nlo[2], nhi[2];
            if (!last) {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) wn[cb] = ldW(ndy * KS + ndx, cb);
            }
#pragma unroll
            for (int xh = 0; xh < 2; ++xh) {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb)
                    acc[cb][xh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb], lo[xh], acc[cb][xh], 0, 0, 0);
                if (same_dx) nhi[xh] = ldP(dy + 2, xh, ndx);
                else if (!last) nlo[xh] = ldP(0, xh, ndx);
            }
#pragma unroll
            for (int xh = 0; xh < 2; ++xh) {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb)
                    acc[cb][2 + xh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb], hi[xh], acc[cb][2 + xh], 0, 0, 0);
                if (same_dx) nlo[xh] = hi[xh];
                else if (!last) nhi[xh] = ldP(1, xh, ndx);
            }
            if (!last) {
#pragma unroll
                for (int xh = 0; xh < 2; ++xh) { lo[xh] = nlo[xh]; hi[xh] = nhi[xh]; }
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) wa[cb] = wn[cb];
            }
        }
    }
    if constexpr (TS > 0) {
        // ---- upsampler tail.  acc[cb][pb]: channels cb*16 + 4g .. +3 of pixel (row 2 wave + (pb >> 1), x = (pb & 1)*16 + c).
        static_assert(NW == 4 && CS == 1, "8 x 32-pixel tiles");
        constexpr int CP = NT + 1;                           // floats per pixel of the LDS tile (odd: spreads the banks)
        static_assert(TH * TW * CP * 4 <= (NPIX * XSB + WS_HALFS) * 2, "the conv output tile fits the LDS stages");
        float* ut = reinterpret_cast<float*>(lds);
        const int U = tail.Cimg * TS * TS;                   // real conv channels
        __syncthreads();                                      // every wave is done reading xs / ws
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const int px = (2 * wave + (pb >> 1)) * TW + (pb & 1) * 16 + c;
#pragma unroll
            for (int cb = 0; cb < NB; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = cb * 16 + 4 * g + e;
                    if (co < U) ut[px * CP + co] = acc[cb][pb][e] + (d.bias ? d.bias[co] : 0.f);
                }
        }
        __syncthreads();
        const int OW = W * TS, OH = H * TS;
        const float scale = 1.f / (float)TS;
        constexpr int OWt = TW * TS, OHt = TH * TS;
        const float* frame = tail.frames + (size_t)(n * tail.T + tail.t_center) * tail.Cimg * H * W;
        for (int idx = tid; idx < OHt * OWt; idx += NTHR) {  // one thread per HR pixel of the tile, all image channels
            const int oyl = idx / OWt, oxl = idx - oyl * OWt;
            const int oy = ty * OHt + oyl, ox = tx * OWt + oxl;
            if (oy >= OH || ox >= OW) continue;
            const int hl = oyl / TS, i = oyl - hl * TS, wl = oxl / TS, j = oxl - wl * TS;
            int xi[4], yi[4];
            float wx[4], wy[4];
            tail_cubic_taps(ox, scale, W, xi, wx);
            tail_cubic_taps(oy, scale, H, yi, wy);
            const float* up = ut + (hl * TW + wl) * CP + i * TS + j;
            for (int ch = 0; ch < tail.Cimg; ++ch) {
                const float* img = frame + (size_t)ch * H * W;
                float bic = 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float* row = img + (size_t)yi[a] * W;
                    const float r = wx[0] * row[xi[0]] + wx[1] * row[xi[1]] + wx[2] * row[xi[2]] + wx[3] * row[xi[3]];
                    bic += wy[a] * r;
                }
                const float pre = bic + up[ch * TS * TS];
                const size_t o = ((size_t)(n * tail.Cimg + ch) * OH + oy) * OW + ox;
                tail.pass[o] = (pre >= 0.f && pre <= 1.f) ? 1 : 0;
                tail.out[o] = fminf(fmaxf(pre, 0.f), 1.f);
            }
        }
        return;
    }
    constexpr int SPX = NB == 2 ? STAGE_PX : STAGE_PX64;
    if constexpr (NB >= 2 && 2 * TW * SPX * NW <= NPIX * XSB + WS_HALFS) {
        // full 32 / 64-channel bf16 outputs: stage the tile's output in LDS and store whole pixel rows (64 / 128 B)
        constexpr int PPP = NT / 8;                          // 16-byte pieces per pixel
        const bool can_stage = d.out_bf16 && vec_ok && (CS == 2 || d.cout_store - cz * NT >= NT);   // workgroup-uniform
        // "every wave is done reading xs / ws": the barrier sits inside the epilogue, behind its operand loads
        if (can_stage && !(d.cout_store - cz * NT >= NT)) __syncthreads();   // (CS = 2: a half that stores nothing)
        if (can_stage && d.cout_store - cz * NT >= NT) {     // (CS = 2: per half of the workgroup)
            __bf16* stage = lds + (tid >> 6) * (2 * TW * SPX);
            conv_epilogue<NB>(d, acc, n, ty, tx, cz, wave, c, g, vec_ok, TH_, stage, SPX, true);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __bf16* o16 = reinterpret_cast<__bf16*>(d.out);
#pragma unroll
            for (int k = 0; k < PPP; ++k) {
                const int item = lane + k * 64;
                const int px = item / PPP, piece = item % PPP;  // wave-local pixel (2 rows x 32), 8-channel piece
                const int gy = ty * TH_ + 2 * wave + (px >> 5), gx = tx * TW + (px & 31);
                if (gy < H && gx < W)
                    *reinterpret_cast<u32x4*>(o16 + ((size_t)(n * H + gy) * W + gx) * d.out_ld + d.out_coff + cz * NT + 8 * piece) =
                        *reinterpret_cast<const u32x4*>(stage + px * SPX + 8 * piece);
            }
            return;
        }
    }
    conv_epilogue<NB>(d, acc, n, ty, tx, cz, wave, c, g, vec_ok, TH_);
}

// ---------------------------------------------------------------- dense-block tail: last 3x3 layer + 1x1 lff, fused
// ResidualDenseBlock.forward (super_resolution.py:245-253) ends with  y4 = relu(conv3x3(cat[0:C])),  out = 0.2 *
// lff(cat[0:C+32]) + x.  Run separately, the 1x1 reads the whole concat buffer again (448 B per pixel for 64 + 128 B of
// residual and output).  Here the 3x3 kernel's K loop also feeds every staged 32-channel chunk to the 1x1 (the centre-tap
// fragments of the 3x3 ARE the 1x1's operands), and the tile's own y4 goes from the accumulators through LDS into the
// 1x1's last K step, so lff costs 16 extra MFMAs per chunk and no extra activation read.
// d3: the 3x3 layer (cout 32, bf16 in/out, bias + ReLU, optional bit-mask output); dl: the lff conv (cout 64, cin = d3.cin
// + 32 over the same buffer; epilogue alpha / residual / output as described by dl).  8x32-pixel tiles, 4 waves.
__global__ __launch_bounds__(256, 2) void rdb_tail_kernel(const nvq_conv_desc d3, const nvq_conv_desc dl, int tilesX,
                                                          int tilesY, int nkc, int vec3, int vecl) {
    constexpr int NB = 2, NT = 32, KS = 3, TAPS = 9, NBL = 4, NTL = 64;
    constexpr int HW_ = TW + 2, HH_ = TH + 2, NPIX = HW_ * HH_;
    constexpr int WS3 = ws_stride_halfs(TAPS, NT);            // 10240 halfs
    constexpr int WSL = ws_stride_halfs(1, NTL);              // 2048 halfs
    constexpr int XITEMS = NPIX * 4;
    constexpr int XPER = (XITEMS + 255) / 256;                // 6
    constexpr int WPER = WS3 / 8 / 256;                       // 5
    constexpr int T4S = 40;                                   // halfs per pixel of the staged y4 tile (80 B)
    static_assert(WSL / 8 == 256, "one lff weight piece per thread");
    static_assert(TH * TW * T4S <= WS3, "the y4 tile reuses the 3x3 weight stage");
    __shared__ __attribute__((aligned(16))) __bf16 lds[NPIX * XSB + WS3 + WSL];
    __bf16* xs = lds;
    __bf16* ws = lds + NPIX * XSB;
    __bf16* wl = ws + WS3;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = lane & 15;
    const int g = lane >> 4;
    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int H = d3.h, W = d3.w;

    f32x4 acc[NB][4], lacc[NBL][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int a = 0; a < NB; ++a) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NBL; ++a) lacc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const u32x4* w3p = reinterpret_cast<const u32x4*>(d3.wpack);
    const u32x4* wlp = reinterpret_cast<const u32x4*>(dl.wpack);
    const __bf16* in16 = reinterpret_cast<const __bf16*>(d3.in) + d3.in_coff;

    unsigned xoff[XPER];
    bool xok[XPER];
#pragma unroll
    for (int k = 0; k < XPER; ++k) {
        const int item = tid + k * 256;
        const int hp = item >> 2;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int gy = ty * TH + hy - 1, gx = tx * TW + hx - 1;
        xok[k] = item < XITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        xoff[k] = xok[k] ? (unsigned)(((size_t)(n * H + gy) * W + gx) * (d3.in_plane ? 32 : d3.in_ld)) : 0u;
    }
    const int nk0 = d3.in_plane ? d3.in_ld >> 5 : 0x7fffffff;  // slice-planar input: see conv_bf16_kernel
    const int sh0 = d3.in_plane ? __ffs(d3.in_ld >> 5) - 1 : 0;
    const unsigned chg = 8 * (tid & 3);
    u32x4 xr[XPER], wr[WPER], lr;
    auto fetch = [&](int kc) {                                // raw loads only (d3.cin % 32 == 0: no channel masks)
        const bool lead = kc < nk0;
        const int sh = lead ? sh0 : 0;
        const unsigned o0 = (lead ? (unsigned)kc * KCB : (unsigned)kc * d3.in_plane) + chg;
#pragma unroll
        for (int k = 0; k < XPER; ++k) xr[k] = *reinterpret_cast<const u32x4*>(in16 + ((xoff[k] << sh) + o0));
#pragma unroll
        for (int k = 0; k < WPER; ++k) wr[k] = w3p[(size_t)kc * (WS3 / 8) + tid + k * 256];
        lr = wlp[(size_t)kc * (WSL / 8) + tid];
    };
    auto fetch_tail = [&]() { lr = wlp[(size_t)nkc * (WSL / 8) + tid]; };   // lff weights of the y4 channels
    auto commit = [&]() {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid + k * 256;
            if (item < XITEMS) *reinterpret_cast<u32x4*>(xs + (item >> 2) * XSB + 8 * (item & 3)) = xok[k] ? xr[k] : z;
        }
#pragma unroll
        for (int k = 0; k < WPER; ++k) reinterpret_cast<u32x4*>(ws)[tid + k * 256] = wr[k];
        reinterpret_cast<u32x4*>(wl)[tid] = lr;
    };
    auto ldL = [&](int cb) -> bf16x8 {                       // lff weights: m = cout cb*16 + c, k = channel 8g..
        return *reinterpret_cast<const bf16x8*>(wl + (g * NTL + cb * 16 + c) * 8);
    };

    fetch(0);
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();
        commit();
        __syncthreads();
        if (kc + 1 < nkc) fetch(kc + 1);
        else fetch_tail();
        auto ldP = [&](int rr, int xh, int dx) -> bf16x8 {
            return *reinterpret_cast<const bf16x8*>(xs + ((2 * wave + rr) * HW_ + xh * 16 + c + dx) * XSB + 8 * g);
        };
        auto ldW = [&](int tap, int cb) -> bf16x8 {
            return *reinterpret_cast<const bf16x8*>(ws + ((tap * 4 + g) * NT + cb * 16 + c) * 8);
        };
        bf16x8 lo[2], hi[2], wa[NB], wn[NB];
        lo[0] = ldP(0, 0, 0); lo[1] = ldP(0, 1, 0);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) wa[cb] = ldW(0, cb);
        hi[0] = ldP(1, 0, 0); hi[1] = ldP(1, 1, 0);
#pragma unroll
        for (int s = 0; s < TAPS; ++s) {                      // dx-major, dy-minor (see conv_bf16_kernel)
            const int dy = s % KS;
            const bool last = s + 1 == TAPS;
            const int ndx = (s + 1) / KS, ndy = (s + 1) % KS;
            const bool same_dx = !last && ndy != 0;
            bf16x8 nlo[2], nhi[2];
            if (!last) {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) wn[cb] = ldW(ndy * KS + ndx, cb);
            }
            if (s == 4) {                                     // centre tap: lo / hi are the tile's own pixels (rows 0 / 1)
#pragma unroll
                for (int cb = 0; cb < NBL; ++cb) {
                    const bf16x8 wf = ldL(cb);
#pragma unroll
                    for (int xh = 0; xh < 2; ++xh) {
                        lacc[cb][xh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, lo[xh], lacc[cb][xh], 0, 0, 0);
                        lacc[cb][2 + xh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hi[xh], lacc[cb][2 + xh], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int xh = 0; xh < 2; ++xh) {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb)
                    acc[cb][xh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb], lo[xh], acc[cb][xh], 0, 0, 0);
                if (same_dx) nhi[xh] = ldP(dy + 2, xh, ndx);
                else if (!last) nlo[xh] = ldP(0, xh, ndx);
            }
#pragma unroll
            for (int xh = 0; xh < 2; ++xh) {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb)
                    acc[cb][2 + xh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb], hi[xh], acc[cb][2 + xh], 0, 0, 0);
                if (same_dx) nlo[xh] = hi[xh];
                else if (!last) nhi[xh] = ldP(1, xh, ndx);
            }
            if (!last) {
#pragma unroll
                for (int xh = 0; xh < 2; ++xh) { lo[xh] = nlo[xh]; hi[xh] = nhi[xh]; }
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) wa[cb] = wn[cb];
            }
        }
    }
    // ---- y4 = relu(acc + bias) as bf16: (a) to the concat buffer through the regular epilogue, (b) into LDS as the 1x1's
    // last operand.  Lane (c, g) holds channels cb*16 + 4g..+3 of pixel (row 2*wave + (pb>>1), x = (pb&1)*16 + c).
    __syncthreads();                                          // every wave is done with xs / ws / wl
    reinterpret_cast<u32x4*>(wl)[tid] = lr;                   // lff weights of channels [cin, cin + 32)
    __bf16* t4 = ws;                                          // [8 x 32 px][T4S]
    static_assert(T4S == STAGE_PX, "the y4 tile is the epilogue's staging tile");
    // the regular epilogue (bias, ReLU, bit masks) with the wave's rows of t4 as its staging tile
    conv_epilogue<NB>(d3, acc, n, ty, tx, 0, wave, c, g, vec3, TH, t4 + wave * (2 * TW * T4S));
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < NBL; ++cb) {
        const bf16x8 wf = ldL(cb);
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const int px = (2 * wave + (pb >> 1)) * TW + (pb & 1) * 16 + c;
            const bf16x8 yf = *reinterpret_cast<const bf16x8*>(t4 + px * T4S + 8 * g);
            lacc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, yf, lacc[cb][pb], 0, 0, 0);
        }
    }
    {   // y4 to the concat buffer as whole 64-byte pixel rows
        __bf16* o16 = reinterpret_cast<__bf16*>(d3.out);
        const __bf16* stage = t4 + wave * (2 * TW * T4S);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int item = lane + k * 64;
            const int px = item >> 2, piece = item & 3;
            const int gy = ty * TH + 2 * wave + (px >> 5), gx = tx * TW + (px & 31);
            if (gy < H && gx < W)
                *reinterpret_cast<u32x4*>(o16 + ((size_t)(n * H + gy) * W + gx) * d3.out_ld + d3.out_coff + 8 * piece) =
                    *reinterpret_cast<const u32x4*>(stage + px * T4S + 8 * piece);
        }
    }
    if (dl.out_bf16) {                                        // lff output as whole 128-byte pixel rows (uniform)
        static_assert(2 * TW * STAGE_PX64 * 4 <= NPIX * XSB + WS3, "lff staging tiles fit the LDS stages");
        __syncthreads();                                      // every wave is done with t4 / wl
        __bf16* stage = lds + wave * (2 * TW * STAGE_PX64);
        conv_epilogue<NBL, true>(dl, lacc, n, ty, tx, 0, wave, c, g, vecl, TH, stage, STAGE_PX64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __bf16* o16 = reinterpret_cast<__bf16*>(dl.out);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int item = lane + k * 64;
            const int px = item >> 3, piece = item & 7;
            const int gy = ty * TH + 2 * wave + (px >> 5), gx = tx * TW + (px & 31);
            if (gy < H && gx < W)
                *reinterpret_cast<u32x4*>(o16 + ((size_t)(n * H + gy) * W + gx) * dl.out_ld + dl.out_coff + 8 * piece) =
                    *reinterpret_cast<const u32x4*>(stage + px * STAGE_PX64 + 8 * piece);
        }
        return;
    }
    conv_epilogue<NBL, true>(dl, lacc, n, ty, tx, 0, wave, c, g, vecl);
}

// The same tile as eight waves with two roles (the automatic choice): waves 0-3 are the 3x3 layer's waves of the kernel above
// (two tile rows each, 72 MFMAs per chunk) and stage only the 3x3 weights; waves 4-7 own the lff accumulators (64 channels x
// the same two rows each, 16 MFMAs per chunk on the centre pixels), stage the halo tile and the lff weights, and write the
// lff output.  With the 64 lff accumulators in the same waves as the 3x3's the kernel above needs ~190 VGPRs, i.e. two waves
// per SIMD, each of which also issues 12 loads + 12 LDS stores per chunk beside its MFMAs; split like this both roles fit 128
// registers (four waves per SIMD) and the 3x3 waves' instruction stream is fragment reads and MFMAs only.
__global__ __launch_bounds__(512, 4) void rdb_tail8_kernel(const nvq_conv_desc d3, const nvq_conv_desc dl, int tilesX,
                                                           int tilesY, int nkc, int vec3, int vecl) {
    // libnvq_debug.so only (tools/rdb_tail_ab.py): tile_rows >> 8 = 1 no 3x3 MFMAs, 2 no lff MFMAs, 4 fetch once, 8 no LDS stores,
    // 16 no lff stores, 32 no y4 epilogue, 64 no first fetch, 128 no residual loads; the shipped library's dbg is the literal 0
#ifdef NVQ_DEBUG_TOOLS
    const int dbg = d3.tile_rows >> 8;
#else
    constexpr int dbg = 0;
#endif
    constexpr int NB = 2, NT = 32, KS = 3, TAPS = 9, NBL = 4, NTL = 64;
    constexpr int HW_ = TW + 2, HH_ = TH + 2, NPIX = HW_ * HH_;
    constexpr int WS3 = ws_stride_halfs(TAPS, NT);            // 10240 halfs
    constexpr int WSL = ws_stride_halfs(1, NTL);              // 2048 halfs
    constexpr int XITEMS = NPIX * 4;
    constexpr int XPER = XITEMS / 256;                        // 5 (lff waves); the last XITEMS - 1280 = 80 pieces: 3x3 waves
    constexpr int XREST = XITEMS - XPER * 256;
    static_assert(XREST >= 0 && XREST <= 256, "one leftover halo piece per 3x3-wave thread at most");
    constexpr int WPER = WS3 / 8 / 256;                       // 5 (3x3 waves)
    constexpr int T4S = 40;
    static_assert(WSL / 8 == 256 && T4S == STAGE_PX && TH * TW * T4S <= WS3, "see rdb_tail_kernel");
    static_assert(2 * TW * STAGE_PX64 * 4 <= NPIX * XSB + WS3, "lff staging tiles fit the LDS stages");
    __shared__ __attribute__((aligned(16))) __bf16 lds[NPIX * XSB + WS3 + WSL];
    __bf16* xs = lds;
    __bf16* ws = lds + NPIX * XSB;
    __bf16* wl = ws + WS3;

    const int tid = threadIdx.x;
    const int rtid = tid & 255;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool lffw = wave >= 4;                              // (wave-uniform)
    const int w4 = wave & 3;
    const int c = lane & 15;
    const int g = lane >> 4;
    int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int H = d3.h, W = d3.w;
    const u32x4* w3p = reinterpret_cast<const u32x4*>(d3.wpack);
    const u32x4* wlp = reinterpret_cast<const u32x4*>(dl.wpack);
    const __bf16* in16 = reinterpret_cast<const __bf16*>(d3.in) + d3.in_coff;
    auto ldP = [&](int rr, int xh, int dx) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(xs + ((2 * w4 + rr) * HW_ + xh * 16 + c + dx) * XSB + 8 * g);
    };

    const int nk0 = d3.in_plane ? d3.in_ld >> 5 : 0x7fffffff;  // slice-planar input: see conv_bf16_kernel
    const int sh0 = d3.in_plane ? __ffs(d3.in_ld >> 5) - 1 : 0;
    const unsigned chg = 8 * (rtid & 3);
    auto halo_off = [&](int item, bool& ok) -> unsigned {     // element offset of halo piece `item`'s pixel (0 and !ok outside)
        const int hp = item >> 2;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int gy = ty * TH + hy - 1, gx = tx * TW + hx - 1;
        ok = item < XITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        return ok ? (unsigned)(((size_t)(n * H + gy) * W + gx) * (d3.in_plane ? 32 : d3.in_ld)) : 0u;
    };
    auto chunk_off = [&](int kc, int& sh) -> unsigned {       // channel offset of chunk kc, shift of the pixel offsets
        const bool lead = kc < nk0;
        sh = lead ? sh0 : 0;
        return (lead ? (unsigned)kc * KCB : (unsigned)kc * d3.in_plane) + chg;
    };
    if (!lffw) {
        // ------------------------------------------------------------ 3x3 waves
        f32x4 acc[NB][4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < NB; ++a) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        u32x4 wr[WPER], lr, xr5;
        bool xok5;
        const unsigned xoff5 = halo_off(XPER * 256 + rtid, xok5);
        auto fetch = [&](int kc) {                            // the chunk's 3x3 and lff weights, the last 80 halo pieces
#pragma unroll
            for (int k = 0; k < WPER; ++k) wr[k] = w3p[(size_t)kc * (WS3 / 8) + rtid + k * 256];
            lr = wlp[(size_t)kc * (WSL / 8) + rtid];
            int sh;
            const unsigned o0 = chunk_off(kc, sh);
            if (rtid < XREST) xr5 = *reinterpret_cast<const u32x4*>(in16 + ((xoff5 << sh) + o0));
        };
        auto ldW = [&](int tap, int cb) -> bf16x8 {
            return *reinterpret_cast<const bf16x8*>(ws + ((tap * 4 + g) * NT + cb * 16 + c) * 8);
        };
        if (!(dbg & 64)) fetch(0);
        for (int kc = 0; kc < nkc; ++kc) {
            __syncthreads();
            if (!(dbg & 8)) {
#pragma unroll
            for (int k = 0; k < WPER; ++k) reinterpret_cast<u32x4*>(ws)[rtid + k * 256] = wr[k];
            reinterpret_cast<u32x4*>(wl)[rtid] = lr;
            if (rtid < XREST) {
                const int item = XPER * 256 + rtid;
                *reinterpret_cast<u32x4*>(xs + (item >> 2) * XSB + 8 * (item & 3)) = xok5 ? xr5 : (u32x4){0u, 0u, 0u, 0u};
            } }
            __syncthreads();
            if (!(dbg & 4)) { if (kc + 1 < nkc) fetch(kc + 1);
            else lr = wlp[(size_t)nkc * (WSL / 8) + rtid]; }   // lff weights of the y4 channels
            if (dbg & 1) continue;
            bf16x8 lo[2], hi[2], wa[NB], wn[NB];
            lo[0] = ldP(0, 0, 0); lo[1] = ldP(0, 1, 0);
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) wa[cb] = ldW(0, cb);
            hi[0] = ldP(1, 0, 0); hi[1] = ldP(1, 1, 0);
#pragma unroll
            for (int s = 0; s < TAPS; ++s) {                  // dx-major, dy-minor (see conv_bf16_kernel)
                const int dy = s % KS;
                const bool last = s + 1 == TAPS;
                const int ndx = (s + 1) / KS, ndy = (s + 1) % KS;
                const bool same_dx = !last && ndy != 0;
                bf16x8 nlo[2], nhi[2];
                if (!last) {
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb) wn[cb] = ldW(ndy * KS + ndx, cb);
                }
#pragma unroll
                for (int xh = 0; xh < 2; ++xh) {
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb)
                        acc[cb][xh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb], lo[xh], acc[cb][xh], 0, 0, 0);
                    if (same_dx) nhi[xh] = ldP(dy + 2, xh, ndx);
                    else if (!last) nlo[xh] = ldP(0, xh, ndx);
                }
#pragma unroll
                for (int xh = 0; xh < 2; ++xh) {
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb)
                        acc[cb][2 + xh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb], hi[xh], acc[cb][2 + xh], 0, 0, 0);
                    if (same_dx) nlo[xh] = hi[xh];
                    else if (!last) nhi[xh] = ldP(1, xh, ndx);
                }
                if (!last) {
#pragma unroll
                    for (int xh = 0; xh < 2; ++xh) { lo[xh] = nlo[xh]; hi[xh] = nhi[xh]; }
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb) wa[cb] = wn[cb];
                }
            }
        }
        // y4 = relu(acc + bias) as bf16 into the wave's rows of the y4 tile (bias, ReLU, bit masks: the regular epilogue)
        __syncthreads();                                      // every wave is done with xs / ws / wl
        reinterpret_cast<u32x4*>(wl)[rtid] = lr;              // lff weights of channels [cin, cin + 32)
        __bf16* t4 = ws;
        if (!(dbg & 32)) conv_epilogue<NB>(d3, acc, n, ty, tx, 0, w4, c, g, vec3, TH, t4 + w4 * (2 * TW * T4S));
        __syncthreads();                                      // the lff waves read the y4 tile
        if (!(dbg & 32)) {   // y4 to the concat buffer as whole 64-byte pixel rows
            __bf16* o16 = reinterpret_cast<__bf16*>(d3.out);
            const __bf16* stage = t4 + w4 * (2 * TW * T4S);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int item = lane + k * 64;
                const int px = item >> 2, piece = item & 3;
                const int gy = ty * TH + 2 * w4 + (px >> 5), gx = tx * TW + (px & 31);
                if (gy < H && gx < W)
                    *reinterpret_cast<u32x4*>(o16 + ((size_t)(n * H + gy) * W + gx) * d3.out_ld + d3.out_coff + 8 * piece) =
                        *reinterpret_cast<const u32x4*>(stage + px * T4S + 8 * piece);
            }
        }
        __syncthreads();                                      // the lff waves reuse the stages for their output
        return;
    }
    // ---------------------------------------------------------------- lff waves
    f32x4 lacc[NBL][4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < NBL; ++a) lacc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned xoff[XPER];
    bool xok[XPER];
#pragma unroll
    for (int k = 0; k < XPER; ++k) xoff[k] = halo_off(rtid + k * 256, xok[k]);
    u32x4 xr[XPER];
    auto fetch = [&](int kc) {
        int sh;
        const unsigned o0 = chunk_off(kc, sh);
#pragma unroll
        for (int k = 0; k < XPER; ++k) xr[k] = *reinterpret_cast<const u32x4*>(in16 + ((xoff[k] << sh) + o0));
    };
    auto ldL = [&](int cb) -> bf16x8 {                       // lff weights: m = cout cb*16 + c, k = channel 8g..
        return *reinterpret_cast<const bf16x8*>(wl + (g * NTL + cb * 16 + c) * 8);
    };
    if (!(dbg & 64)) fetch(0);
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();
        if (!(dbg & 8)) {
            const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < XPER; ++k) {
                const int item = rtid + k * 256;
                *reinterpret_cast<u32x4*>(xs + (item >> 2) * XSB + 8 * (item & 3)) = xok[k] ? xr[k] : z;
            }
        }
        __syncthreads();
        if (kc + 1 < nkc && !(dbg & 4)) fetch(kc + 1);
        if (dbg & 2) continue;
        bf16x8 px[4];                                         // the tile's own pixels: halo rows 2*w4 + 1 / + 2, dx = 1
        px[0] = ldP(1, 0, 1); px[1] = ldP(1, 1, 1); px[2] = ldP(2, 0, 1); px[3] = ldP(2, 1, 1);
#pragma unroll
        for (int cb = 0; cb < NBL; ++cb) {
            const bf16x8 wf = ldL(cb);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
                lacc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, px[pb], lacc[cb][pb], 0, 0, 0);
        }
    }
    // Epilogue operands (the host side sends only lff descriptors with bias, a bf16 residual over all 64 channels and bf16
    // output here): every load is issued here, before the barriers and the y4 step, so that the tile's tail is one memory round trip and not one per (pixel
    // block, channel block) - with the generic epilogue's per-block loads the tail was 60 % of this kernel's time.
    bool okp[4];
    uint2 rr[4][NBL];
    float4 bb[NBL];
    // (32-bit element offsets from a uniform base: the host side checks n*h*w*res_ld < 2^32)
    const __bf16* r16 = reinterpret_cast<const __bf16*>(dl.res) + dl.res_coff;
    auto res_load = [&](int pb) {
        const int gy = ty * TH + 2 * w4 + (pb >> 1), gx = tx * TW + (pb & 1) * 16 + c;
        okp[pb] = gy < H && gx < W;
        const unsigned off = okp[pb] ? ((unsigned)(n * H + gy) * (unsigned)W + gx) * (unsigned)dl.res_ld + 4u * g : 0u;
#pragma unroll
        for (int cb = 0; cb < NBL; ++cb) rr[pb][cb] = *reinterpret_cast<const uint2*>(r16 + (off + cb * 16u));
    };
    if (!(dbg & 128)) { res_load(0); res_load(1); }           // the first tile row now, the second behind the y4 step
    __syncthreads();                                          // every wave is done with xs / ws / wl
    __syncthreads();                                          // the y4 tile and the last lff weights (3x3 waves) are in LDS
    {
        const __bf16* t4 = ws;
#pragma unroll
        for (int cb = 0; cb < NBL; ++cb) {
            const bf16x8 wf = ldL(cb);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int px = (2 * w4 + (pb >> 1)) * TW + (pb & 1) * 16 + c;
                const bf16x8 yf = *reinterpret_cast<const bf16x8*>(t4 + px * T4S + 8 * g);
                lacc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, yf, lacc[cb][pb], 0, 0, 0);
            }
        }
    }
    if (!(dbg & 128)) { res_load(2); res_load(3); }
#pragma unroll
    for (int cb = 0; cb < NBL; ++cb) bb[cb] = ld4(dl.bias + cb * 16 + 4 * g);   // (cache hits, under the barrier)
    __syncthreads();                                      // every wave is done with t4 / wl
    __bf16* stage = lds + w4 * (2 * TW * STAGE_PX64);
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        if (!okp[pb]) continue;
#pragma unroll
        for (int cb = 0; cb < NBL; ++cb) {                // the generic epilogue's arithmetic, in its order
            float v[4] = {lacc[cb][pb][0] + bb[cb].x, lacc[cb][pb][1] + bb[cb].y, lacc[cb][pb][2] + bb[cb].z,
                          lacc[cb][pb][3] + bb[cb].w};
            {   // product and sum rounded separately, as in the generic epilogue (whose conditional add is not contracted)
#pragma clang fp contract(off)
                v[0] = v[0] * dl.alpha + __uint_as_float(rr[pb][cb].x << 16);
                v[1] = v[1] * dl.alpha + __uint_as_float(rr[pb][cb].x & 0xffff0000u);
                v[2] = v[2] * dl.alpha + __uint_as_float(rr[pb][cb].y << 16);
                v[3] = v[3] * dl.alpha + __uint_as_float(rr[pb][cb].y & 0xffff0000u);
            }
            *reinterpret_cast<bf16x4*>(stage + ((pb >> 1) * TW + (pb & 1) * 16 + c) * STAGE_PX64 + cb * 16 + 4 * g) =
                (bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (dbg & 16) return;
    __bf16* o16 = reinterpret_cast<__bf16*>(dl.out);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int item = lane + k * 64;
        const int px = item >> 3, piece = item & 7;
        const int gy = ty * TH + 2 * w4 + (px >> 5), gx = tx * TW + (px & 31);
        if (gy < H && gx < W)
            *reinterpret_cast<u32x4*>(o16 + ((size_t)(n * H + gy) * W + gx) * dl.out_ld + dl.out_coff + 8 * piece) =
                *reinterpret_cast<const u32x4*>(stage + px * STAGE_PX64 + 8 * piece);
    }
}

// ---------------------------------------------------------------- weight gradient
// grid = (pixel split, ci chunk, 32-co chunk); K = the 32 pixels of one tile row per MFMA.  The LDS images stay
// [pixel][channels]; ds_read_b64_tr_b16 delivers, per 16-lane group, 4 pixels x 16 channels transposed, i.e.
// exactly the k-major operand the MFMA wants.  k order inside a row: group g, element e -> x = 4g + e (e < 4),
// 16 + 4g + e - 4 (e >= 4), identical for both operands, so that the two 16-lane groups of a 32-lane half read 8
// consecutive pixels = 8 distinct 32-B bank ranges (pixel stride 96 B, or 160 B for the 64-channel image).
// XB / YB: x / dy are stored as bf16.  Staging pieces are 16 bytes: 4 fp32 or 8 bf16 channels.
// CIC: input channels per workgroup.  32: wave = one 16x16 (ci, co) block for all taps.  64 (bf16 x only): wave =
// one 16-ci block x both 16-co blocks, i.e. twice the MFMA work per staged dy tile and dy re-read half as often -
// the per-tile iteration is latency-bound, so doing more per iteration is what pays.
// COC: output channels per workgroup (32, or 64 for the 1x1 kernel with bf16 dy: its accumulators are one fragment per
// (ci, co) block, so a wave takes all four 16-co blocks and the x tile is fetched once instead of once per 32 co).
template <int KS, bool XB, bool YB, int CIC, int COC = 32>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const nvq_wgrad_desc d, int tilesX, int tilesY,
                                                             int ntiles, int nci32, int nco) {
    static_assert(CIC == 32 || (CIC == 64 && XB), "64-channel chunks need bf16 x");
    static_assert(COC == 32 || (COC == 64 && CIC == 64 && YB && KS == 1), "64-co chunks: 1x1, bf16 x and dy");
    constexpr int HALO = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int HW_ = TW + 2 * HALO;
    constexpr int HH_ = TH + 2 * HALO;
    constexpr int NPIX = HW_ * HH_;
    constexpr int XSX = CIC == 64 ? 80 : XSB;        // bf16 per staged x pixel (160 B / 96 B)
    constexpr int XPP = (XB ? 4 : 8) * (CIC / 32);   // 16-byte pieces per pixel of the X chunk
    constexpr int YSX = COC == 64 ? 80 : XSB;        // bf16 per staged dy pixel
    constexpr int YPP = (YB ? 4 : 8) * (COC / 32);
    constexpr int XCH = CIC / XPP, YCH = COC / YPP;   // channels per piece
    constexpr int XITEMS = NPIX * XPP;
    constexpr int XPER = (XITEMS + 255) / 256;
    constexpr int YPER = TH * TW * YPP / 256;
    constexpr int NCO = (CIC == 64 ? 2 : 1) * (COC / 32);   // 16-co blocks per wave
    __shared__ __attribute__((aligned(16))) __bf16 lds[NPIX * XSX + TH * TW * YSX];
    __bf16* xs = lds;
    __bf16* dys = lds + NPIX * XSX;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 15;
    const int g = lane >> 4;
    const int q = r >> 2, p = r & 3;                 // tr-read role of this lane inside its 16-lane group
    const int cib = CIC == 64 ? wave : wave >> 1;    // 16-ci block of this wave inside the chunk
    const int cob0 = CIC == 64 ? 0 : (wave & 1);
    const int cic = blockIdx.y, coc = blockIdx.z;
    const int H = d.h, W = d.w;
    const float* x32 = d.x + d.x_coff;
    const __bf16* x16 = reinterpret_cast<const __bf16*>(d.x) + d.x_coff;
    const float* dy32 = d.dy + d.dy_coff;
    const __bf16* dy16 = reinterpret_cast<const __bf16*>(d.dy) + d.dy_coff;

    f32x4 acc[NCO][TAPS];
#pragma unroll
    for (int a = 0; a < NCO; ++a)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[a][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x4 xr[XPER], yr[YPER];
    // fp32 column sums of dy (bias gradient): channels yq..yq+3 in bsumA, yq+4..yq+7 in bsumB (bf16 pieces only)
    float4 bsumA = make_float4(0.f, 0.f, 0.f, 0.f), bsumB = bsumA;
    const int xq = XCH * (tid & (XPP - 1));          // channel offset of this thread's X pieces inside the chunk
    const int yq = YCH * (tid & (YPP - 1));
    const bool xch_ok = cic * CIC + xq < d.cin;
    // slice-planar x (nvq_wgrad_desc::x_plane): this thread's channels lie in the leading x_ld-channel tensor or in one
    // compact 32-channel plane; either way its pieces are  x + pixel * xmul + xbase
    const int xch = cic * CIC + xq;
    const bool xlead = !d.x_plane || xch < d.x_ld;
    const unsigned xmul = xlead ? d.x_ld : 32;
    const size_t xbase = xlead ? (size_t)xch : (size_t)(xch >> 5) * d.x_plane + (xch & 31);
    const int ych = coc * COC + yq;
    const bool ych_ok = ych < d.cout;
    // all loads unconditional (invalid pieces read element 0 of the slice); the validity masks are applied in
    // commit(), so that nothing between the prefetch and the next commit uses a loaded value
    unsigned xmask = 0, ymask = 0;
    auto fetch = [&](int tile) {
        int bt = xcd_tile(tile, ntiles);
        const int tx = bt % tilesX; bt /= tilesX;
        const int ty = bt % tilesY;
        const int n = bt / tilesY;
        xmask = 0; ymask = 0;
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid + k * 256;
            const int hp = item / XPP;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int gy = ty * TH + hy - HALO, gx = tx * TW + hx - HALO;
            const bool ok = item < XITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W && xch_ok;
            xmask |= (ok ? 1u : 0u) << k;
            const size_t off = ok ? ((size_t)(n * H + gy) * W + gx) * xmul + xbase : 0;
            if constexpr (XB) xr[k] = *reinterpret_cast<const u32x4*>(x16 + off);
            else xr[k] = *reinterpret_cast<const u32x4*>(x32 + off);
        }
#pragma unroll
        for (int k = 0; k < YPER; ++k) {
            const int item = tid + k * 256;
            const int pp = item / YPP;
            const int py = pp / TW, px = pp - py * TW;
            const int gy = ty * TH + py, gx = tx * TW + px;
            // the dy slice is readable up to a multiple of 4 (fp32) / 8 (bf16) channels (checked on the host);
            // channels >= cout only feed partial sums that are never written
            const bool ok = gy < H && gx < W && ych_ok;
            ymask |= (ok ? 1u : 0u) << k;
            const size_t off = ok ? ((size_t)(n * H + gy) * W + gx) * d.dy_ld + ych : 0;
            if constexpr (YB) yr[k] = *reinterpret_cast<const u32x4*>(dy16 + off);
            else yr[k] = *reinterpret_cast<const u32x4*>(dy32 + off);
        }
    };
    auto commit = [&]() {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int item = tid + k * 256;
            if (item < XITEMS) {
                const u32x4 v = (xmask >> k) & 1 ? xr[k] : z;
                __bf16* dst = xs + (item / XPP) * XSX + XCH * (item & (XPP - 1));
                if constexpr (XB) *reinterpret_cast<u32x4*>(dst) = v;
                else *reinterpret_cast<bf16x4*>(dst) = cvt4(as_f4(v));
            }
        }
#pragma unroll
        for (int k = 0; k < YPER; ++k) {
            const int item = tid + k * 256;
            const u32x4 v = (ymask >> k) & 1 ? yr[k] : z;
            __bf16* dst = dys + (item / YPP) * YSX + YCH * (item & (YPP - 1));
            if constexpr (YB) {
                *reinterpret_cast<u32x4*>(dst) = v;           // word w holds channels 2w (low half), 2w+1 (high half)
                bsumA.x += bf_lo(v[0]); bsumA.y += bf_hi(v[0]); bsumA.z += bf_lo(v[1]); bsumA.w += bf_hi(v[1]);
                bsumB.x += bf_lo(v[2]); bsumB.y += bf_hi(v[2]); bsumB.z += bf_lo(v[3]); bsumB.w += bf_hi(v[3]);
            } else {
                const float4 f = as_f4(v);
                *reinterpret_cast<bf16x4*>(dst) = cvt4(f);
                bsumA.x += f.x; bsumA.y += f.y; bsumA.z += f.z; bsumA.w += f.w;   // unrounded values
            }
        }
    };
    // a wave whose 16 input channels lie beyond the tensor's (the second half of a 64-channel chunk when cin % 64 == 32) has
    // nothing to accumulate: it keeps staging with the others but leaves the matrix cores to the workgroup sharing the CU
    const bool idle = cic * CIC + cib * 16 >= d.cin_w;
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    auto tr_read = [&](const __bf16* base, int stride, int pix, int chblock) -> s16x4 {
        // lane (q, p) supplies row `pix + q`'s address, columns 4p..4p+3 of the 16-channel block
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + (pix + q) * stride + chblock * 16 + 4 * p));
    };

    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        commit();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        if (idle) continue;                                   // (wave-uniform; the wave still staged and met the barriers)
#pragma unroll 2
        for (int py = 0; py < TH; ++py) {
            bf16x8 bfrag[NCO];
#pragma unroll
            for (int a = 0; a < NCO; ++a) {
                const s16x4 b0 = tr_read(dys, YSX, py * TW + 4 * g, cob0 + a);
                const s16x4 b1 = tr_read(dys, YSX, py * TW + 16 + 4 * g, cob0 + a);
                bfrag[a] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int ddy = tap / KS, ddx = tap - ddy * KS;
                const int hp = (py + ddy) * HW_ + ddx;
                const s16x4 a0 = tr_read(xs, XSX, hp + 4 * g, cib);
                const s16x4 a1 = tr_read(xs, XSX, hp + 16 + 4 * g, cib);
                const bf16x8 afrag = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int a = 0; a < NCO; ++a)
                    acc[a][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[a], acc[a][tap], 0, 0, 0);
            }
        }
    }

    // partial slabs are indexed in 32-ci units so the reduce kernel is independent of CIC
    const int slab = cic * (CIC / 32) + (cib * 16) / 32;
    if (slab < nci32) {                                      // wave-uniform
        const int cil0 = (cib * 16) % 32;
#pragma unroll
        for (int a = 0; a < NCO; ++a) {
            // `nco` and the slab index count 32-co chunks: block a of a 64-co workgroup lies in chunk coc*2 + a/2
            const int coc32 = coc * (COC / 32) + (cob0 + a) / 2, cob = (cob0 + a) % 2;
            if (coc32 >= nco) continue;
            float* part = d.workspace + ((size_t)(blockIdx.x * nci32 + slab) * nco + coc32) * (TAPS * WG_C * WG_C);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                for (int e = 0; e < 4; ++e) part[(tap * WG_C + cil0 + 4 * g + e) * WG_C + cob * 16 + r] = acc[a][tap][e];
        }
    }

    // bias partials: thread t's sums cover channels yq..yq+YCH-1 of the co chunk, pieces repeat every YPP threads
    if (cic == 0 && d.dbias != nullptr) {                    // uniform per workgroup
        float* scratch = reinterpret_cast<float*>(lds);      // 256 x 8 floats
        __syncthreads();
        st4(scratch + 8 * tid, bsumA);
        st4(scratch + 8 * tid + 4, bsumB);
        __syncthreads();
        if (tid < COC) {
            const int piece = tid / YCH, e = tid % YCH;      // channel `tid` of the chunk
            float s = 0.f;
            for (int k = 0; k < 256 / YPP; ++k) s += scratch[8 * (YPP * k + piece) + e];
            float* bp = d.workspace + (size_t)WGRAD_MAX_SLABS * 9 * WG_C * WG_C;
            const int coc32 = coc * (COC / 32) + tid / WG_C;
            if (coc32 < nco) bp[((size_t)blockIdx.x * nco + coc32) * WG_C + tid % WG_C] = s;
        }
    }
}

// ---------------------------------------------------------------- host side
size_t pack_floats_bf16(int cout, int cin_store, int ksize) {
    const int NT = choose_nt(cout);
    const size_t ncz = (cout + NT - 1) / NT, nkc = (cin_store + KCB - 1) / KCB;
    return ncz * nkc * (size_t)ws_stride_halfs(ksize * ksize, NT) / 2;   // bf16 pairs per float
}

int pack_bf16(const float* w, int cout_w, int cin_w, int ksize, int transpose, int cin_store, int cout_keep,
              float* wpack, hipStream_t s) {
    const int cout = transpose ? cout_keep : cout_w;
    const int NT = choose_nt(cout);
    const int ncz = (cout + NT - 1) / NT, nkc = (cin_store + KCB - 1) / KCB;
    const long total = (long)ncz * nkc * ws_stride_halfs(ksize * ksize, NT);
    int nblk = ceil_div(total, 256);
    if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(nblk), dim3(256), 0, s, w, cout_w, cin_w, ksize * ksize, transpose,
                       cout_keep, NT, ncz, nkc, reinterpret_cast<__bf16*>(wpack));
    return check_launch("conv_pack_bf16");
}

PackJobDev pack_job_bf16(const nvq_pack_job& j) {
    const int cout = j.transpose ? j.cout_keep : j.cout_w;
    const int NT = choose_nt(cout);
    const int ncz = (cout + NT - 1) / NT, nkc = (j.cin_store + KCB - 1) / KCB;
    return PackJobDev{j.w, j.wpack, j.cout_w, j.cin_w, j.ksize * j.ksize, j.transpose, j.cout_keep, NT, ncz, nkc,
                      (long)ncz * nkc * ws_stride_halfs(j.ksize * j.ksize, NT)};
}

int pack_batch_bf16(const PackJobTable& t, int n, hipStream_t s) {
    hipLaunchKernelGGL(pack_batch_bf16_kernel, dim3(32, n), dim3(256), 0, s, t);
    return check_launch("conv_pack_batch(bf16)");
}

int conv_forward_bf16(const nvq_conv_desc& d, int vec_ok, hipStream_t s) {
    // the kernels keep per-thread activation offsets as 32-bit element counts
    NVQ_REQUIRE((size_t)d.n * d.h * d.w * d.in_ld < ((size_t)1 << 32),
                "conv_forward(bf16): input tensor of %d x %d x %d x %d elements exceeds the 32-bit offsets of the kernels", d.n,
                d.h, d.w, d.in_ld);
    const int NT = choose_nt(d.cout);
    const int ncz = (d.cout_store + NT - 1) / NT;
    const int nkc = (d.cin + KCB - 1) / KCB;
    int tilesX = (d.w + TW - 1) / TW, tilesY = (d.h + TH - 1) / TH;
    // 16-row tiles (8 waves) for the cout <= 32 3x3 kernel with bf16 input, when the image fills them AND is large enough for
    // such launches to fill the device: images of fewer than NVQ_SMALL_IMAGE_TILES 16x32 tiles (the 64x64 clips of the
    // continual-learning step, the coarse levels of the recovery net) take the 8-row tiles of the 4-wave kernel, twice the
    // workgroups for the same work.  The rule looks at ONE image, not at the batch: the forms differ in their fp32 summation
    // order inside a K chunk (32x32x16 vs 16x16x32 MFMA), and an image's result must not depend on how many others share the
    // launch (tests/test_full_size_gpu.py, batch independence).  tile_rows = 16 / 162 / 164 / 8 still force one form.
    // the same tile on v_mfma_f32_32x32x16_bf16 (conv_m32.hip): tile_rows 162 = two rows per wave (8 waves), 164 = four rows
    // per wave (4 waves).  Automatic for up to 128 input channels (profiles/r04_mfma32_per_shape.txt: the two-row form is 1 - 7 %
    // faster there and 0 - 4 % slower on 160 / 192; the four-row form loses everywhere); tile_rows 16 = always the 16x16x32 form.
    const bool underfilled = tilesX * ((d.h + 2 * TH - 1) / (2 * TH)) < NVQ_SMALL_IMAGE_TILES;
    const int rows = d.tile_rows == 0 && underfilled ? 8 : d.tile_rows;
    if (d.ksize == 3 && NT == 32 && d.in_bf16 && d.h >= 2 * TH && vec_ok &&
        (rows == 162 || rows == 164 || (rows == 0 && d.cin <= 128)))
        return conv_forward_m32(d, rows == 164 ? 4 : 2, g_debug_mode & 15, s);
    if (d.ksize == 3 && NT == 32 && d.in_bf16 && d.h >= 2 * TH && rows != 8) {
        tilesY = (d.h + 2 * TH - 1) / (2 * TH);
        const dim3 grid8((unsigned)((long)tilesX * tilesY * d.n), ncz);
        hipLaunchKernelGGL((conv_bf16_kernel<2, 3, true, 8>), grid8, dim3(512), 0, s, d, tilesX, tilesY, nkc, vec_ok,
                           g_debug_mode & 3, TailArgs{});
        return check_launch("conv_forward_bf16");
    }
    // tile_rows 264: the same layers on v_mfma_f32_32x32x16_bf16, a wave = 2 rows x all 64 channels (conv_m32.hip; A/B only)
    // (265: the channel split itself on 32x32x16, eight waves)
    if (d.ksize == 3 && NT == 64 && d.in_bf16 && (d.tile_rows == 264 || d.tile_rows == 265)) {
        NVQ_REQUIRE(vec_ok && d.out_bf16 && d.cout_store % 64 == 0 && d.cin % 8 == 0,
                    "conv_forward: tile_rows 264 / 265 need bf16 output and a multiple of 64 stored channels");
        return conv_forward_m32(d, d.tile_rows == 264 ? 64 : 65, g_debug_mode & 3, s);
    }
    // 64 output channels per workgroup, 3x3, bf16 input: eight waves, each half of them 32 of the channels (see the kernel)
    if (d.ksize == 3 && NT == 64 && d.in_bf16 && d.tile_rows != 8) {
        const dim3 grid8((unsigned)((long)tilesX * tilesY * d.n), ncz);
        hipLaunchKernelGGL((conv_bf16_kernel<2, 3, true, 8, 2>), grid8, dim3(512), 0, s, d, tilesX, tilesY, nkc, vec_ok,
                           g_debug_mode & 3, TailArgs{});
        return check_launch("conv_forward_bf16");
    }
    const dim3 grid((unsigned)((long)tilesX * tilesY * d.n), ncz);
#define NVQ_LAUNCH_CONVB(NB, KS)                                                                                        \
    do {                                                                                                                 \
        if (d.in_bf16)                                                                                                   \
            hipLaunchKernelGGL((conv_bf16_kernel<NB, KS, true>), grid, dim3(256), 0, s, d, tilesX, tilesY, nkc, vec_ok, g_debug_mode & 3, TailArgs{});   \
        else                                                                                                             \
            hipLaunchKernelGGL((conv_bf16_kernel<NB, KS, false>), grid, dim3(256), 0, s, d, tilesX, tilesY, nkc, vec_ok, g_debug_mode & 3, TailArgs{});  \
    } while (0)
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

// conv 3x3 (cin -> Cimg * s * s, bf16 input) + PixelShuffle(s) + bicubic skip + clamp in one launch
int upsampler_tail_bf16(const nvq_conv_desc& d, const float* frames, int T, int t_center, int Cimg, int s, float* out,
                        unsigned char* pass, hipStream_t st) {
    NVQ_REQUIRE((size_t)d.n * d.h * d.w * d.in_ld < ((size_t)1 << 32), "upsampler_tail: tensor exceeds 32-bit offsets");
    const int tilesX = (d.w + TW - 1) / TW, tilesY = (d.h + TH - 1) / TH;
    const int nkc = (d.cin + KCB - 1) / KCB;
    const dim3 grid((unsigned)((long)tilesX * tilesY * d.n), 1);
    const TailArgs t{frames, out, pass, T, t_center, Cimg};
    if (s == 2 && Cimg * 4 <= 16)
        hipLaunchKernelGGL((conv_bf16_kernel<1, 3, true, 4, 1, 2>), grid, dim3(256), 0, st, d, tilesX, tilesY, nkc, 0, 0, t);
    else if (s == 3 && Cimg * 9 <= 32)
        hipLaunchKernelGGL((conv_bf16_kernel<2, 3, true, 4, 1, 3>), grid, dim3(256), 0, st, d, tilesX, tilesY, nkc, 0, 0, t);
    else if (s == 4 && Cimg * 16 <= 64)
        hipLaunchKernelGGL((conv_bf16_kernel<4, 3, true, 4, 1, 4>), grid, dim3(256), 0, st, d, tilesX, tilesY, nkc, 0, 0, t);
    else
        NVQ_REQUIRE(false, "upsampler_tail: scale %d with %d image channels", s, Cimg);
    return check_launch("upsampler_tail_forward");
}

// resident workgroups per CU of the three heaviest kernels as the runtime computes them (tools/kernel_phases.py)
void conv_occupancy_bf16(int* out) {
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[0], conv_bf16_kernel<2, 3, true, 8>, 512, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[1], conv_bf16_kernel<2, 3, true, 8, 2>, 512, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[2], conv_bf16_kernel<4, 3, true, 4>, 256, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[3], rdb_tail_kernel, 256, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[4], wgrad_bf16_kernel<3, true, true, 64, 32>, 256, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[5], conv_bf16_kernel<2, 3, true, 4>, 256, 0);
}

int rdb_tail_bf16(const nvq_conv_desc& d3, const nvq_conv_desc& dl, int vec3, int vecl, hipStream_t s) {
    NVQ_REQUIRE((size_t)d3.n * d3.h * d3.w * d3.in_ld < ((size_t)1 << 32) &&
                    (!dl.res || (size_t)dl.n * dl.h * dl.w * dl.res_ld < ((size_t)1 << 32)),
                "rdb_tail_forward: tensor exceeds 32-bit offsets");
    const int tilesX = (d3.w + TW - 1) / TW, tilesY = (d3.h + TH - 1) / TH;
    const int nkc = d3.cin / KCB;
    // d3.tile_rows = 4, or an lff epilogue other than the dense block's (bias, bf16 residual over all channels, bf16 output):
    // the four-wave kernel; otherwise the eight-wave, two-role one (same tiles, same results)
    const bool usual = dl.out_bf16 && dl.bias && dl.res && dl.res_bf16 && dl.res_cmax >= 64 && !dl.out2 && !dl.relu && vecl;
    if ((d3.tile_rows & 255) == 4 || !usual)
        hipLaunchKernelGGL(rdb_tail_kernel, dim3((unsigned)((long)tilesX * tilesY * d3.n)), dim3(256), 0, s, d3, dl, tilesX,
                           tilesY, nkc, vec3, vecl);
    else
        hipLaunchKernelGGL(rdb_tail8_kernel, dim3((unsigned)((long)tilesX * tilesY * d3.n)), dim3(512), 0, s, d3, dl, tilesX,
                           tilesY, nkc, vec3, vecl);
    return check_launch("rdb_tail_forward");
}

int conv_wgrad_bf16(const nvq_wgrad_desc& d, int nsplit, int nci, int nco, int tilesX, int tilesY, int ntiles,
                    hipStream_t s) {
    if ((d.variant & 15) != 1 && wgrad_m32_takes(d)) return conv_wgrad_m32(d, s);
    NVQ_REQUIRE((d.variant & 15) != 2, "conv_wgrad: variant 2 (all-input-channel kernel) does not take %d -> %d channels, ksize %d",
                d.cin, d.cout, d.ksize);
    // nsplit / nci come from the caller in 32-ci units; with bf16 x and >= 64 input channels use 64-ci workgroups
    const bool wide = d.x_bf16 && d.dy_bf16 && d.cin_w >= 64;   // (the bf16-x / fp32-dy wide variant spills)
    const int ncig = wide ? (nci + 1) / 2 : nci;
    if (wide) {
        nsplit = WGRAD_MAX_WG / (ncig * nco);
        if (nsplit < 1) nsplit = 1;
        if (nsplit > ntiles) nsplit = ntiles;
        if (nsplit >= 8) nsplit &= ~7;      // multiple of the XCD count: see xcd_tile()
    }
    NVQ_REQUIRE((size_t)nsplit * nci * nco * d.ksize * d.ksize <= (size_t)WGRAD_MAX_SLABS * 9,
                "conv_wgrad(bf16): %d splits x %d x %d chunks exceed the workspace", nsplit, nci, nco);
    // 1x1, bf16 x and dy, >= 64 output channels: one workgroup takes 64 co (x fetched once per 64 co instead of per 32)
    if (wide && d.ksize == 1 && d.cout >= 64) {
        const int nco64 = (nco + 1) / 2;
        nsplit = WGRAD_MAX_WG / (ncig * nco64);
        // the partial slabs (one per split x 32-ci chunk x 32-co chunk, taps * 32 * 32 floats each) must fit the workspace
        const int cap = WGRAD_MAX_SLABS * 9 / (nci * nco * d.ksize * d.ksize);
        if (nsplit > cap) nsplit = cap;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > ntiles) nsplit = ntiles;
        if (nsplit >= 8) nsplit &= ~7;
        NVQ_REQUIRE((size_t)nsplit * nci * nco * d.ksize * d.ksize <= (size_t)WGRAD_MAX_SLABS * 9,
                    "conv_wgrad(bf16, 64 co): %d splits x %d x %d chunks exceed the workspace", nsplit, nci, nco);
        hipLaunchKernelGGL((wgrad_bf16_kernel<1, true, true, 64, 64>), dim3(nsplit, ncig, nco64), dim3(256), 0, s, d, tilesX,
                           tilesY, ntiles, nci, nco);
        return nsplit;
    }
    const dim3 grid(nsplit, ncig, nco);
#define NVQ_LAUNCH_WG(KS, XB, YB, CIC) \
    hipLaunchKernelGGL((wgrad_bf16_kernel<KS, XB, YB, CIC>), grid, dim3(256), 0, s, d, tilesX, tilesY, ntiles, nci, nco)
    if (d.ksize == 3) {
        if (wide) NVQ_LAUNCH_WG(3, true, true, 64);
        else if (d.x_bf16 && d.dy_bf16) NVQ_LAUNCH_WG(3, true, true, 32);
        else if (d.x_bf16) NVQ_LAUNCH_WG(3, true, false, 32);
        else if (d.dy_bf16) NVQ_LAUNCH_WG(3, false, true, 32);
        else NVQ_LAUNCH_WG(3, false, false, 32);
    } else {
        if (wide) NVQ_LAUNCH_WG(1, true, true, 64);
        else if (d.x_bf16 && d.dy_bf16) NVQ_LAUNCH_WG(1, true, true, 32);
        else if (d.x_bf16) NVQ_LAUNCH_WG(1, true, false, 32);
        else if (d.dy_bf16) NVQ_LAUNCH_WG(1, false, true, 32);
        else NVQ_LAUNCH_WG(1, false, false, 32);
    }
#undef NVQ_LAUNCH_WG
    return nsplit;      // > 0: the number of pixel splits actually used (the reduce kernel needs it)
}

}  // namespace nvq
