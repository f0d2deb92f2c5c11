// 3x3 weight gradient for bf16-stored x and dy, cin >= 64 (NVQ_MATH_BF16): the weight half of convolution_backward of
// the dense layers, the flow net and the attention convs (reference super_resolution.py:74-82,168-175,236-253).
//
// What bounds this computation is the bytes a CU ingests per MFMA: a pixel tile of x with its halo
// is 1.3 x its own bytes, and the dy tile is needed once per group of input channels a workgroup owns.  So:
//   * a workgroup walks a VERTICAL STRIP of 4 x 32-pixel tiles and keeps x in LDS as a ring of 10 pixel rows: the six rows a
//     tile's 3x3 window touches plus the four rows of the next tile, which arrive while the current tile is computed.  Going
//     down one tile only 4 new rows (with their 2 halo columns) are fetched: 1.06 x the tile's own bytes;
//   * the rows arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no LDS store pass, ONE barrier per tile):
//     the LDS images are [row][16-channel plane][pixel][16 ch], so that a row is one contiguous span that whole wave
//     instructions fill 1 KiB at a time from per-lane source addresses, and the 8 consecutive pixels x 32 B a half-wave
//     reads with ds_read_b64_tr_b16 are 256 contiguous bytes = all 64 banks.  Pixels outside the image (and channels past
//     the tensor) are fetched from a 16-byte zero constant;
//   * a workgroup owns 96 input channels where the tensor allows it (cin = 96, 160, 192 = 96 | 96 + 64 | 96 + 96), 64
//     otherwise: the dy tile is fetched twice instead of three times for cin = 192, and no half-empty 64-channel chunk runs
//     for cin % 64 == 32.  Two such workgroups (81 664 B of LDS each, 384 threads) share a CU.
// Wave w of the six owns the 16-ci block w of the chunk (x both 16-co blocks x 9 taps = 72 accumulator registers; a
// 64-channel chunk leaves waves 4-5 without matrix work) and issues every sixth DMA instruction of a tile.  The bias
// gradient is one more MFMA per pixel row with an all-ones operand (wave 0 of the chunk-0 workgroups).  Partial sums go to
// the same per-(split, 32-ci, 32-co) slabs as wgrad_bf16_kernel's, reduced in double by wgrad_reduce_kernel in a fixed order:
// deterministic.
#include <stdlib.h>
#include "conv_common.h"

namespace nvq {

__device__ const uint4 nvq_zero16 = {0u, 0u, 0u, 0u};   // source of every out-of-image / out-of-tensor piece

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int SNT = 384;                      // threads
constexpr int TR = 2;                         // output rows per tile
constexpr int AHEAD = 3;                      // tiles in flight behind the one being computed
constexpr int RING = TR + 2 + AHEAD * TR;     // x rows in LDS: the tile's window (4) + the rows of the next three tiles (6)
constexpr int RW = TW + 2;                    // pixels per ring row
constexpr int YBUF = TR * 2 * TW * 16;        // halfs per dy buffer: [row][plane][32 px][16 ch]; AHEAD + 1 of them
constexpr int KDMA = 3;                       // DMA instructions per wave and tile (what the counted vmcnt waits assume)

// LDS-DMA of 16 bytes per lane: LDS[lds_dst + 16 lane] = *gsrc (lds_dst: wave-uniform LDS byte address).  Inline asm, because
// the compiler treats the builtin's LDS write as a possible alias of every later ds_read and drains vmcnt(0) in front of the
// first fragment read, i.e. it would serialise the fetch of the tiles ahead with the MFMAs of the current one.  An asm load is
// invisible to its s_waitcnt bookkeeping (guide 5.7): the kernel counts them itself (KDMA per wave and tile, always).
// M0 holds the LDS base and is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// One chunk of CIC input channels starting at channel c0, output channels [32 coc, 32 coc + 32).
template <int CIC>
__device__ __forceinline__ void strip_body(const nvq_wgrad_desc& d, __bf16* lds, int c0, int coc, int tilesX, int tilesYR,
                                           int nsy, int SL, int nstrips, int nci32, int nco, int dbg) {
    constexpr int NPL = CIC / 16;             // 16-channel planes = waves with matrix work
    constexpr int XSLOTS = NPL * RW * 2;      // 16-byte slots of a ring row (408 / 272)
    constexpr int IPR = (XSLOTS + 63) / 64;   // DMA instructions per row (7 / 5); the last one is partial
    constexpr int LASTL = XSLOTS - 64 * (IPR - 1);   // its active lanes (24 / 16)
    constexpr int ROWH = NPL * RW * 16;       // halfs per ring row
    constexpr int NXI = TR * IPR;             // x instructions per tile (14 / 10)
    constexpr int NI = NXI + TR * 2;          // + dy instructions (18 / 14)
    constexpr int HPW = (2 * IPR + 5) / 6;    // strip head (two rows), per wave (3 / 2)
    static_assert(NI <= 6 * KDMA, "KDMA instructions per wave cover a tile");
    __bf16* xs = lds;
    __bf16* dys = lds + 6 * RW * 16 * RING;   // behind a 96-channel ring in either case

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int q = r >> 2, p = r & 3;          // tr-read role inside the 16-lane group
    const int H = d.h, W = d.w;
    const __bf16* x16 = reinterpret_cast<const __bf16*>(d.x) + d.x_coff;
    const __bf16* dy16 = reinterpret_cast<const __bf16*>(d.dy) + d.dy_coff;
    const __bf16* zsrc = reinterpret_cast<const __bf16*>(&nvq_zero16);
    // LDS byte addresses of the two images (wave-uniform); a dummy instruction (64-channel chunk: the waves whose third
    // instruction of a tile does not exist) lands behind the 64-channel ring, in the part only a 96-channel ring uses
    const unsigned xs_b = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) __bf16*)xs);
    const unsigned dys_b = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) __bf16*)dys);
    const unsigned dummy_b = xs_b + 2 * (4 * RW * 16 * RING) + 1024 * wave;
    static_assert(CIC == 96 || 2 * (4 * RW * 16 * RING) + 6 * 1024 <= 2 * (6 * RW * 16 * RING), "room for the dummies");

    // ---- DMA role: instruction j = wave + 6 m of a tile.  j < NXI: slots 64 i .. 64 i + 63 of x row j / IPR (i = j % IPR);
    // j < NI: dy row (j - NXI) / 2, plane (j - NXI) % 2; else a dummy.  This lane's slot -> (plane, pixel, half) -> channel and
    // column.  x channels lie in the leading x_ld-channel tensor or in a compact 32-channel plane (nvq_wgrad_desc::x_plane).
    auto xplace = [&](int ch, unsigned& mul, unsigned& base) {
        const bool lead = !d.x_plane || ch < d.x_ld;
        mul = lead ? d.x_ld : 32;             // elements per pixel where this channel lives
        base = lead ? (unsigned)ch : (unsigned)(ch >> 5) * d.x_plane + (ch & 31);
    };
    int dcol[KDMA];                           // image column relative to the tile's first column
    unsigned dmul[KDMA], dbase[KDMA];         // elements per pixel, channel offset
    unsigned chok = 0;                        // bit m: the piece's channels exist
#pragma unroll
    for (int m = 0; m < KDMA; ++m) {
        const int j = wave + 6 * m;
        if (j < NXI) {
            const int slot = 64 * (j % IPR) + lane;
            const int pl = slot / (2 * RW), rem = slot - pl * (2 * RW);
            const int ch = c0 + 16 * pl + 8 * (rem & 1);
            dcol[m] = (rem >> 1) - 1;
            xplace(ch, dmul[m], dbase[m]);
            chok |= (slot < XSLOTS && ch < d.cin ? 1u : 0u) << m;
        } else {
            const int ch = coc * 32 + 16 * ((j - NXI) & 1) + 8 * (lane & 1);
            dcol[m] = lane >> 1;
            dmul[m] = d.dy_ld;
            dbase[m] = ch;
            chok |= (j < NI && ch < d.cout ? 1u : 0u) << m;
        }
    }
    // the strip head: instruction j = wave + 6 m of rows 0 and 1
    int hcol[HPW];
    unsigned hmul[HPW], hbase[HPW], hchok = 0;
#pragma unroll
    for (int m = 0; m < HPW; ++m) {
        const int j = wave + 6 * m;
        const int slot = 64 * (j % IPR) + lane;
        const int pl = slot / (2 * RW), rem = slot - pl * (2 * RW);
        const int ch = c0 + 16 * pl + 8 * (rem & 1);
        hcol[m] = (rem >> 1) - 1;
        xplace(ch, hmul[m], hbase[m]);
        hchok |= (j < 2 * IPR && slot < XSLOTS && ch < d.cin ? 1u : 0u) << m;
    }

    // ---- MFMA role
    const int cib = wave;                                          // 16-ci block of the chunk (waves < NPL)
    const bool idle = wave >= NPL || c0 + cib * 16 >= d.cin_w;
    const bool bias_wave = wave == 0 && c0 == 0 && d.dbias != nullptr;
    f32x4 acc[2][9], accb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[a][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const __bf16 one = (__bf16)1.0f;
    const bf16x8 ones = {one, one, one, one, one, one, one, one};
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    const __bf16* xlane = xs + cib * (RW * 16) + (4 * g + q) * 16 + 4 * p;   // this lane's tr-read base in its x plane
    const __bf16* ylane = dys + (4 * g + q) * 16 + 4 * p;

    for (int s = blockIdx.x; s < nstrips; s += gridDim.x) {
        int sid = xcd_tile(s, nstrips);
        const int tx = sid % tilesX; sid /= tilesX;
        const int sy = sid % nsy;
        const int n = sid / nsy;
        const int t0 = sy * SL, t1 = min(t0 + SL, tilesYR);
        const int nt = t1 - t0;
        const unsigned img = (unsigned)n * H;
        // column validity of this lane's pieces in this strip
        unsigned ok = 0, hok = 0;
#pragma unroll
        for (int m = 0; m < KDMA; ++m) {
            const int gx = tx * TW + dcol[m];
            ok |= (((chok >> m) & 1) && gx >= 0 && gx < W ? 1u : 0u) << m;
        }
#pragma unroll
        for (int m = 0; m < HPW; ++m) {
            const int gx = tx * TW + hcol[m];
            hok |= (((hchok >> m) & 1) && gx >= 0 && gx < W ? 1u : 0u) << m;
        }
        // x row rho of the strip = image row t0 * TR - 1 + rho, kept in ring row rho % RING; tile t (strip-relative) computes
        // from rho = TR t .. TR t + 3; its body, fetched AHEAD tiles early, is rho = TR t + 2, TR t + 3; its dy rows go to dy
        // buffer t % (AHEAD + 1).  Tiles past the strip's end are fetched from the zero constant (same instruction count).
        auto dma_body = [&](int t) {                               // strip-relative tile t; exactly KDMA instructions
            const int gyx = (t0 + t) * TR + 1, gyy = (t0 + t) * TR;  // image row of x body row 0 / dy row 0
            const int rbase = (TR * t + 2) % RING;
            const bool live = t < nt;                              // (uniform)
#pragma unroll
            for (int m = 0; m < KDMA; ++m) {
                const int j = wave + 6 * m;                        // (uniform)
                if (j >= NI) {                                     // 64-channel chunk only
                    glds16(zsrc, __builtin_amdgcn_readfirstlane(dummy_b));
                    continue;
                }
                const bool isx = j < NXI;
                const int row = isx ? j / IPR : (j - NXI) >> 1;
                const int gy = (isx ? gyx : gyy) + row;
                const bool v = live && ((ok >> m) & 1) && gy < H;
                const __bf16* base = isx ? x16 : dy16;
                const __bf16* src = v ? base + ((size_t)((img + gy) * W + tx * TW + dcol[m]) * dmul[m] + dbase[m]) : zsrc;
                int rr = rbase + row;
                rr = rr >= RING ? rr - RING : rr;
                const unsigned dst = isx ? xs_b + 2 * (rr * ROWH + (j % IPR) * 512)
                                         : dys_b + 2 * ((t % (AHEAD + 1)) * YBUF + (j - NXI) * 512);
                // (one instruction per wave either way: the partial one runs with its first LASTL lanes)
                if (!isx || j % IPR != IPR - 1 || lane < LASTL) glds16(src, __builtin_amdgcn_readfirstlane(dst));
            }
        };
        auto dma_head = [&]() {
#pragma unroll
            for (int m = 0; m < HPW; ++m) {
                const int j = wave + 6 * m;
                if (j >= 2 * IPR) continue;
                const int row = j / IPR;
                const int gy = t0 * TR - 1 + row;
                const bool v = ((hok >> m) & 1) && gy >= 0 && gy < H;
                const __bf16* src = v ? x16 + ((size_t)((img + gy) * W + tx * TW + hcol[m]) * hmul[m] + hbase[m]) : zsrc;
                const unsigned dst = xs_b + 2 * (row * ROWH + (j % IPR) * 512);
                if (j % IPR != IPR - 1 || lane < LASTL) glds16(src, __builtin_amdgcn_readfirstlane(dst));
            }
        };
        auto rd = [&](const __bf16* base, int off) -> s16x4 {
            return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off));
        };

        // everything this workgroup still has in flight (the zero tiles behind the previous strip) lands, and the previous
        // strip's last tile has been computed, before the ring is refilled
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        dma_head();
#pragma unroll
        for (int t = 0; t < AHEAD; ++t) dma_body(t);
        for (int t = 0; t < nt; ++t) {
            // all but the newest AHEAD - 1 tiles of this wave have landed, i.e. tile t ...
            static_assert(AHEAD == 3 && KDMA == 3, "the counted wait below is vmcnt((AHEAD - 1) * KDMA)");
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __syncthreads();                                        // ... and everybody's; tile t-1 has been computed
            if (!(dbg & 2)) dma_body(t + AHEAD);
            if (idle || (dbg & 1)) continue;                        // (wave-uniform)
            // input-row stationary: row i of the tile's four x rows feeds output row 0 as tap row i and output row 1 as tap
            // row i-1; its six fragments (3 column shifts x 2 pixel halves) are read one row ahead of their MFMAs
            const int ra = (TR * t) % RING;
            const __bf16* yt = ylane + (t % (AHEAD + 1)) * YBUF;
            bf16x8 bfr[2][2];
#pragma unroll
            for (int py = 0; py < TR; ++py)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const __bf16* yb = yt + py * (2 * TW * 16) + a * (TW * 16);
                    const s16x4 b0 = rd(yb, 0), b1 = rd(yb, 16 * 16);
                    bfr[py][a] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            s16x4 ar[2][6];
            auto read_row = [&](int i, s16x4 (&dst)[6]) {
                int rr = ra + i;
                rr = rr >= RING ? rr - RING : rr;
                const __bf16* rp = xlane + rr * ROWH;
#pragma unroll
                for (int ddx = 0; ddx < 3; ++ddx) {
                    dst[2 * ddx] = rd(rp, ddx * 16);
                    dst[2 * ddx + 1] = rd(rp, (16 + ddx) * 16);
                }
            };
            read_row(0, ar[0]);
            if (bias_wave) {
#pragma unroll
                for (int py = 0; py < TR; ++py)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[py][a], accb[a], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TR + 2; ++i) {
                if (i + 1 < TR + 2) read_row(i + 1, ar[(i + 1) & 1]);
#pragma unroll
                for (int ddx = 0; ddx < 3; ++ddx) {
                    const bf16x8 afrag = __builtin_bit_cast(
                        bf16x8, __builtin_shufflevector(ar[i & 1][2 * ddx], ar[i & 1][2 * ddx + 1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int py = 0; py < TR; ++py) {
                        const int ddy = i - py;
                        if (ddy < 0 || ddy > 2) continue;
#pragma unroll
                        for (int a = 0; a < 2; ++a)
                            acc[a][ddy * 3 + ddx] =
                                __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfr[py][a], acc[a][ddy * 3 + ddx], 0, 0, 0);
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // nothing of this wave is in flight when it leaves

    // ---- partial slabs, indexed in 32-ci units: workspace[((split * nci32 + slab) * nco + coc) * 9*32*32]
    if (!idle) {
        const int ci0 = c0 + cib * 16;
        const int slab = ci0 >> 5, cil0 = ci0 & 31;
        if (slab < nci32) {
            float* part = d.workspace + ((size_t)(blockIdx.x * nci32 + slab) * nco + coc) * (9 * WG_C * WG_C);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int e = 0; e < 4; ++e) part[(tap * WG_C + cil0 + 4 * g + e) * WG_C + a * 16 + r] = acc[a][tap][e];
        }
    }
    // ---- bias partials: every row of the ones-product holds the column sums; row 0 = register 0 of the lanes g = 0
    if (bias_wave && g == 0) {
        float* bp = d.workspace + (size_t)WGRAD_MAX_SLABS * 9 * WG_C * WG_C;
#pragma unroll
        for (int a = 0; a < 2; ++a) bp[((size_t)blockIdx.x * nco + coc) * WG_C + a * 16 + r] = accb[a][0];
    }
}

}  // namespace

// grid (pixel split, channel chunk, 32-co chunk); chunks [0, n96) are 96 channels wide, the rest 64
__global__ __launch_bounds__(SNT, 3) void wgrad_strip_kernel(const nvq_wgrad_desc d, int tilesX, int tilesYR, int nsy, int SL,
                                                               int nstrips, int nci32, int nco, int n96, int dbg) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[6 * RW * 16 * RING + (AHEAD + 1) * YBUF];
    static_assert(sizeof(lds) <= 81920, "two workgroups per CU");
    const int c = blockIdx.y;
    if (c < n96)
        strip_body<96>(d, lds, 96 * c, blockIdx.z, tilesX, tilesYR, nsy, SL, nstrips, nci32, nco, dbg);
    else
        strip_body<64>(d, lds, 96 * n96 + 64 * (c - n96), blockIdx.z, tilesX, tilesYR, nsy, SL, nstrips, nci32, nco, dbg);
}

int wgrad_strip_occupancy() {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_strip_kernel, SNT, 0);
    return n;
}

// Returns the pixel-split count used (> 0) or a negative error code.  nci32 / nco: 32-channel units of the partial slabs.
int conv_wgrad_strip_bf16(const nvq_wgrad_desc& d, int nci32, int nco, hipStream_t s) {
    const int tilesX = (d.w + TW - 1) / TW, tilesY = (d.h + TR - 1) / TR;     // tiles of TR rows
    // channel chunks: as many 96s as possible, 64s for a remainder of 32 or 64 channels (cin = 128: 64 + 64)
    const int u = (d.cin_w + 31) / 32;                              // 32-channel units that receive a gradient
    int n64 = u < 3 ? 1 : (3 - u % 3) % 3;
    int n96 = u < 3 ? 0 : (u - 2 * n64) / 3;
    if (u >= 3 && 3 * n96 + 2 * n64 < u) ++n96;
    const int nchunks = n96 + n64;
    int nsplit = WGRAD_MAX_WG / (nchunks * nco);
    const int cap = WGRAD_MAX_SLABS / (nci32 * nco);                // partial slabs the workspace holds
    if (nsplit > cap) nsplit = cap;
    if (nsplit < 1) nsplit = 1;
    // strips of ~68 tiles, shorter when the image is small, so that every workgroup gets a few
    int nsy = (tilesY + 67) / 68;
    const int want = (2 * nsplit + d.n * tilesX - 1) / (d.n * tilesX);
    if (nsy < want) nsy = want < tilesY ? want : tilesY;
    const int SL = (tilesY + nsy - 1) / nsy;
    nsy = (tilesY + SL - 1) / SL;
    const int nstrips = d.n * tilesX * nsy;
    if (nsplit > nstrips) nsplit = nstrips;
    if (nsplit >= 8) nsplit &= ~7;                                  // multiple of the XCD count: see xcd_tile()
    NVQ_REQUIRE((size_t)nsplit * nci32 * nco <= (size_t)WGRAD_MAX_SLABS, "conv_wgrad(strip): %d splits x %d x %d chunks exceed the workspace",
                nsplit, nci32, nco);
    // the kernel keeps element offsets in 32 bits
    const size_t xel = d.x_plane ? (size_t)((d.cin + 31) / 32) * d.x_plane + (size_t)d.n * d.h * d.w * d.x_ld
                                 : (size_t)d.n * d.h * d.w * d.x_ld;
    NVQ_REQUIRE(xel < ((size_t)1 << 32) && (size_t)d.n * d.h * d.w * d.dy_ld < ((size_t)1 << 32),
                "conv_wgrad(strip): tensor of %d x %d x %d pixels exceeds the 32-bit offsets of the kernel", d.n, d.h, d.w);
    hipLaunchKernelGGL(wgrad_strip_kernel, dim3(nsplit, nchunks, nco), dim3(SNT), 0, s, d, tilesX, tilesY, nsy, SL, nstrips,
                       nci32, nco, n96, getenv("NVQ_WS_DBG") ? atoi(getenv("NVQ_WS_DBG")) : 0);
    return nsplit;
}

}  // namespace nvq
