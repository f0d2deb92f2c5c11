// NVQ_MATH_BF16 weight gradient of a 3x3 convolution with ALL input channels in one workgroup, on
// v_mfma_f32_32x32x16_bf16 (the weight half of aten::convolution_backward for the dense layers of ResidualDenseBlock,
// super_resolution.py:236-253).
//
// wgrad_bf16_kernel (conv_bf16.hip) splits a launch into (pixel split, 64-ci chunk, 32-co chunk) workgroups: every ci chunk
// re-reads the dy tile, and the x tile carries a one-pixel halo on the BIG tensor (8x32 tiles: +33 %): 1.5x the algorithmic
// bytes (profiles/r03_cfg2_traffic_pmc.json).  Here
//     dW[co][ci][ky][kx] = sum_q x[ci][q] * dy[co][q - (ky-1, kx-1)]
// is evaluated over the pixels q of x UNITS (2 rows x 32 pixels) WITHOUT halo - whole, aligned 128-B lines of the big tensor,
// each read exactly once - against the unit's dy rows WITH a one-pixel halo (32 channels, zero outside the image), and the
// workgroup owns every input channel, so dy is not re-read per channel chunk: (cin + 2.1 * 32) / (cin + 32) = 1.08 .. 1.16x the
// algorithmic bytes issued, of which the dy rows two consecutive units share are L2 hits.
//
// One persistent workgroup per CU (XCD-ordered tiles, two units per 4 x 32 tile), eight waves, one s_barrier per unit:
//   * every wave holds accumulators: the 9 taps x 2 groups of CB 32-channel blocks are 18 (group, tap) jobs of CB accumulator
//     blocks (M = ci, N = co, K = 16 pixels of a unit row; both operands transposed out of [pixel][channel] LDS images by
//     ds_read_b64_tr_b16).  Waves w and w + 4 share a SIMD: six waves take two jobs, waves 2 and 3 three (12 / 12 / 15 / 15 jobs
//     per SIMD).  The fragments of K step k + 1 are read under the MFMAs of step k - across units too: the barrier that hands
//     unit u + 1 over sits in front of unit u's last MFMA batch;
//   * ALL eight waves stage x (three 16-byte loads and three LDS stores per wave and unit), the six two-job waves also dy (with
//     its halo; they sum it for the bias gradient, fp32, from the staged values): two units ahead in two register sets, unit
//     u + 2 stored into the slot unit u was read from right behind barrier u + 1's last MFMA batch, then the loads of unit u + 4.
// Loads are inline asm (uniform base + per-lane offset: one instruction) with counted s_waitcnt vmcnt by hand: with register
// sets that rotate over loop iterations the compiler's own bookkeeping drains vmcnt(0) in front of every store pass.
//
// How it got here (cin = 192, 8 x 540 x 960; profiles/r04_wgrad_all_ci.txt): (1) every wave staging a whole 4-row tile through
// registers spills behind 144 accumulators; (2) both tiles by LDS-DMA (global_load_lds_dwordx4, three units ahead, counted
// vmcnt): correct, 1.08x the bytes, but DMA alone 318 us + MFMAs alone 392 us = 582 us together - a DMA piece costs the issuing
// wave ~200 cycles in an MFMA phase whether its data comes from HBM or L2; (3) two dedicated producer waves (registers, two or
// three units ahead) + six MFMA waves: s_memtime stamps show the producers taking 3400 cycles per unit for 13 loads + 13 LDS
// stores each while the consumers compute for 1500 and wait; (4) the memory instructions spread over all eight waves: 496 us,
// 482 - 492 with the dy waves at s_setprio 3; (5) the MFMA jobs spread over all eight waves as above: 473 us against 565 - 595
// of the split kernel.  What bounds it: the staging of a unit - 40 loads and 40 LDS stores of 1 KiB through the CU's one
// address pipe and LDS store path - takes the waves 700 - 1900 cycles of a 3300-cycle unit wherever it is placed (in the middle
// of the unit between other waves' MFMAs, or with all waves at once where no MFMA is in flight), next to ~1900 cycles of MFMAs
// on the busiest SIMD at the 1.5 - 1.75 GHz the chip holds under this load.
// Partial slabs / bias partials in the layout of wgrad_bf16_kernel; the same reduce kernel finishes.
//
// LDS images.  x unit: [32-channel block][64 pixels][64 B] (the four pixels x two 16-channel blocks of a half-wave's transposing
// read are 256 contiguous bytes = all 64 banks); a load instruction covers 16 pixels of one block (a compact plane: 1 KiB
// contiguous in memory; a block of the leading tensor: 64 B of every pixel).  dy of a unit: [4 x 34 halo pixels][64 B].
#include <type_traits>
#include "conv_common.h"

namespace nvq {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

constexpr int WM_TR = 4;                 // tile rows
constexpr int WM_NTHR = 512;             // eight waves
constexpr int WM_HW = TW + 2, WM_HH = 2 + 2, WM_NPY = WM_HW * WM_HH;       // dy of a unit with halo: 34 x 4
constexpr int WM_YBYTES = WM_NPY * 64;
constexpr int WM_UPX = 2 * TW;           // pixels of a unit (two tile rows)
constexpr int WM_DYT = 384;              // threads (six waves) that stage dy: 544 pieces per unit -> two each (the second mostly a dummy)

// Producer loads and their waits by hand (cdna_hip_programming.md 5.7): with register sets that rotate over the iterations of
// a loop the compiler's own s_waitcnt bookkeeping drains vmcnt(0) in front of every LDS write pass - i.e. every unit in flight -
// and throttles the loads behind write-after-write waits.  An asm load is invisible to it; vm_wait<N>() + hold() make the
// loaded registers usable: N = loads issued after the ones needed (vmcnt counts in issue order).
__device__ __forceinline__ void gload16(u32x4& dst, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
// ... with a uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: one instruction, no vector address arithmetic
__device__ __forceinline__ void gload16s(u32x4& dst, const void* sbase, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void hold(u32x4& r) { asm volatile("" : "+v"(r)); }   // uses of r stay behind the wait in front

template <int NCI>
struct WmCfg {
    static constexpr int CB = (NCI + 1) / 2;                            // 32-channel blocks per wave (two block groups)
    static constexpr int UBYTES = WM_UPX * NCI * 64;                    // x bytes of a unit
    static constexpr int XPIECES = WM_UPX * NCI * 4;                    // its 16-byte pieces ...
    static constexpr int XPT = (XPIECES + WM_NTHR - 1) / WM_NTHR;       // ... per thread (ALL eight waves stage x)
    static constexpr int LDS = 2 * UBYTES + 2 * WM_YBYTES;
};

}  // namespace

// cout = 32, cin = 32 NCI (3 <= NCI <= 6), slice-planar bf16 x (nvq_wgrad_desc::x_plane), bf16 dy.
template <int NCI>
__global__ __launch_bounds__(WM_NTHR, 2) void wgrad_m32_kernel(const nvq_wgrad_desc d, int tilesX, int tilesY, int ntiles) {
    using C = WmCfg<NCI>;
    constexpr int CB = C::CB, XPT = C::XPT;
    static_assert(C::LDS <= 160 * 1024 && C::LDS >= WM_DYT * 32 && NCI >= 2 && NCI <= 6, "configuration");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[C::LDS];
    constexpr int YOFF = 2 * C::UBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d.h, W = d.w;
    const int G = gridDim.x;
    const int mytiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / G + 1 : 0;
    const int U = 2 * mytiles;                                // units of this workgroup: rows 2 (u & 1), + 1 of its tile u >> 1
    const __bf16* x16 = reinterpret_cast<const __bf16*>(d.x);
    const __bf16* dy16 = reinterpret_cast<const __bf16*>(d.dy);
    const int LU = d.x_ld >> 5;                               // 32-channel blocks of the leading tensor

    // ---- roles.  The 9 taps x 2 block groups are 18 (group, tap) jobs of CB accumulator blocks each.  Waves w and w + 4 share
    // a SIMD: waves 0, 1, 4, 5 (two MFMA waves per SIMD) and 6, 7 take two jobs, waves 2, 3 (next to 6, 7) three:
    //   group 0 (blocks [0, CB)):   wave 0: taps 0, 1   wave 1: 2, 3   wave 6: 4, 5   wave 2: 6, 7, 8
    //   group 1 (blocks [CB, NCI)): wave 4: taps 0, 1   wave 5: 2, 3   wave 7: 4, 5   wave 3: 6, 7, 8
    // = 12 / 12 / 15 / 15 jobs' worth of MFMAs per SIMD instead of the 18 / 18 / 9 / 9 of six equal MFMA waves.  The six
    // two-job waves also stage dy (with its halo) and sum it for the bias gradient; all eight stage x.
    const bool three = wave == 2 || wave == 3;
    const int grp = (wave == 0 || wave == 1 || wave == 2 || wave == 6) ? 0 : 1;
    const int tap0 = three ? 6 : (wave == 6 || wave == 7) ? 4 : 2 * (wave & 1);
    const int dyw = wave < 2 ? wave : (wave < 6 ? wave - 2 : wave - 2);     // index among the dy-staging waves (waves 2, 3: unused)
    const int cb0 = grp * CB;
    const int ncb = NCI - cb0 < CB ? NCI - cb0 : CB;

    // ---- x staging, by every wave: piece j = tid + 512 k of a unit image [block][64 px][4 pieces] -> block (tid >> 8) + 2 k
    // (wave-uniform), unit row (tid >> 7) & 1 (wave-uniform), column (tid >> 2) & 31, piece tid & 3; its LDS address is linear
    // in j.  A load is one instruction: uniform base + one of two per-lane byte offsets.
    const int xcol = (tid >> 2) & 31, xsub = tid & 3, xrow = (wave >> 1) & 1, xblk0 = wave >> 2;   // (row, block: from the SGPR)
    const unsigned v_lead = (unsigned)((xcol * d.x_ld + 8 * xsub) * 2), v_plane = (unsigned)((xcol * 32 + 8 * xsub) * 2);
    // dy staging, by the six two-job waves: piece j = dt + 384 k (k < 2) of the unit's [4 x 34 halo pixels][4 pieces]
    const int dt = (wave < 2 ? wave : wave - 2) * 64 + lane;  // 0 .. 383 (waves 2, 3 never use it)
    struct Set { u32x4 x[XPT]; u32x4 y[2]; bool xok; unsigned char yok; };
    // Units are loaded in order; the tile of the next one is kept decoded: consecutive tiles of a workgroup are G / 8 apart in the
    // XCD order (xcd_tile(bid + k G) = xcd_tile(bid) + k G / 8 for G % 8 == 0), so (n, ty, tx) advance by carries - the two
    // divisions of a from-scratch decode are ~80 scalar instructions per unit in every wave.
    int c_n = 0, c_ty = 0, c_tx = 0, c_unit = 0;             // the tile of unit c_unit (wave-uniform)
    {
        int bt = xcd_tile((int)blockIdx.x, ntiles);
        c_tx = bt % tilesX; bt /= tilesX;
        c_ty = bt % tilesY;
        c_n = bt / tilesY;
    }
    const int c_step = G >> 3;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
#ifdef NVQ_WM_STAMPS
    // (development build: s_memtime stamps of workgroup 0, wave w's k-th stamp at stamps[w * 2048 + k]; tools/wgrad_stamps2.py)
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(d.workspace + (size_t)WGRAD_MAX_SLABS * 9 * WG_C * WG_C + 512 * 256);
    int nst = 0;
    const bool stamping = (d.variant & 128) && blockIdx.x == 0 && lane == 0;
    auto stamp = [&]() {
        if (stamping && nst < 2048) stamps[wave * 2048 + nst++] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [&]() {};
#endif

    // One role = one instantiation: NT taps per wave, DY: the wave stages dy.  (Separate top-level branches: accumulators that
    // are live across two copies of a loop in ONE branch make the register allocator spill.)
    auto role = [&](auto nt_tag, auto dy_tag) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr bool DY = decltype(dy_tag)::value;
        constexpr int S = XPT + (DY ? 2 : 0);                 // loads per unit and thread (always exactly S: counted waits)
        // dy pieces of this thread: halo pixel hp (row hr, column hc), 16-byte piece dt & 3
        int yhr[2] = {0, 0}, yhc[2] = {0, 0};
        unsigned yvoff[2] = {0u, 0u}, yin = 0, yreal = 0;
        if constexpr (DY) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = dt + WM_DYT * k, hp = j >> 2;
                const bool real = hp < WM_NPY;
                yhr[k] = real ? hp / WM_HW : 0;
                yhc[k] = real ? hp - yhr[k] * WM_HW : 0;
                yvoff[k] = (unsigned)(((yhr[k] * W + yhc[k]) * d.dy_ld + 8 * (dt & 3)) * 2);   // from the halo origin (y0 - 1, x0 - 1)
                yreal |= (real ? 1u : 0u) << k;
                yin |= ((real && yhr[k] >= 1 && yhr[k] <= 2 && yhc[k] >= 1 && yhc[k] <= TW) ? 1u : 0u) << k;
            }
        }
        auto load = [&](int u, Set& st) {                     // exactly S loads (a piece that does not exist: a dummy)
            if ((u >> 1) != (c_unit >> 1)) {
                if ((G & 7) == 0) {
                    c_tx += c_step;
                    while (c_tx >= tilesX) { c_tx -= tilesX; ++c_ty; }
                    while (c_ty >= tilesY) { c_ty -= tilesY; ++c_n; }
                } else {                                      // (a handful of workgroups: decode from scratch)
                    int bt = xcd_tile((int)blockIdx.x + (u >> 1) * G, ntiles);
                    c_tx = bt % tilesX; bt /= tilesX;
                    c_ty = bt % tilesY;
                    c_n = bt / tilesY;
                }
            }
            c_unit = u;
            const int y0 = c_ty * WM_TR + 2 * (u & 1), x0 = c_tx * TW;
            {
                const bool row_ok = y0 + xrow < H;            // (wave-uniform)
                const unsigned rel = (unsigned)((c_n * H + (row_ok ? y0 + xrow : y0)) * W + x0);
                st.xok = row_ok && xcol < W - x0;
                const unsigned vl = st.xok ? v_lead : 0u, vp = st.xok ? v_plane : 0u;
#pragma unroll
                for (int k = 0; k < XPT; ++k) {
                    const int blk = xblk0 + 2 * k < NCI ? xblk0 + 2 * k : 0;   // (wave-uniform; past the unit: block 0 again, not stored)
                    if (blk < LU) gload16s(st.x[k], x16 + (size_t)(rel * (unsigned)d.x_ld + 32u * blk), vl);
                    else gload16s(st.x[k], x16 + ((size_t)blk * d.x_plane + rel * 32u), vp);
                }
            }
            if constexpr (DY) {
                // the unit's dy rows y0 - 1 .. y0 + 2 with a column of halo on either side (the two rows it shares with the
                // neighbouring unit are fetched by both: L2 hits one unit later)
                const long pix0 = (long)(c_n * H + y0) * W + x0;
                if (x0 >= 1 && x0 + TW + 1 <= W && y0 >= 1 && y0 + 3 <= H) {   // (uniform) the whole halo lies inside the image
                    st.yok = (unsigned char)yreal;
                    const __bf16* ybase = dy16 + ((pix0 - W - 1) * d.dy_ld + d.dy_coff);
#pragma unroll
                    for (int k = 0; k < 2; ++k) gload16s(st.y[k], ybase, (yreal >> k) & 1 ? yvoff[k] : 0u);
                } else {
                    st.yok = 0;
                    const __bf16* ybase = dy16 + d.dy_coff;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int gy = y0 + yhr[k] - 1, gx = x0 + yhc[k] - 1;
                        const bool ok = ((yreal >> k) & 1) && gy >= 0 && gy < H && gx >= 0 && gx < W;
                        st.yok |= (ok ? 1u : 0u) << k;
                        gload16s(st.y[k], ybase, ok ? (unsigned)((((long)(c_n * H + gy) * W + gx) * d.dy_ld + 8 * (dt & 3)) * 2) : 0u);
                    }
                }
            }
        };
        auto write = [&](int u, Set& st) {                    // (behind a vm_wait that covers st's loads)
            const u32x4 z = {0u, 0u, 0u, 0u};
            unsigned char* slot = lds + (u & 1) * C::UBYTES + tid * 16;
#pragma unroll
            for (int k = 0; k < XPT; ++k) {
                hold(st.x[k]);
                if (xblk0 + 2 * k < NCI) *reinterpret_cast<u32x4*>(slot + k * (WM_NTHR * 16)) = st.xok ? st.x[k] : z;
            }
            if constexpr (DY) {
                unsigned char* ys = lds + YOFF + (u & 1) * WM_YBYTES + dt * 16;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    hold(st.y[k]);
                    if (!((yreal >> k) & 1)) continue;
                    const u32x4 v = (st.yok >> k) & 1 ? st.y[k] : z;
                    *reinterpret_cast<u32x4*>(ys + k * (WM_DYT * 16)) = v;
                    if ((yin >> k) & 1) {                     // the unit's own pixels: what the bias gradient sums, once per pixel
                        bsum[0] += bf_lo(v[0]); bsum[1] += bf_hi(v[0]); bsum[2] += bf_lo(v[1]); bsum[3] += bf_hi(v[1]);
                        bsum[4] += bf_lo(v[2]); bsum[5] += bf_hi(v[2]); bsum[6] += bf_lo(v[3]); bsum[7] += bf_hi(v[3]);
                    }
                }
            }
        };
        // ---- MFMA side.  A transposing read: lane (r = (q, p), hb) supplies the address of image row pix + q, columns 4p ..
        // 4p+3 of a 16-channel block and receives channel r of that block for pixels pix .. pix + 3; two reads (pix, pix + 4) =
        // the 8 k values 8h .. 8h + 7 of an operand.  A wave of group 1 may own fewer than CB blocks (NCI = 3, 5): it runs the
        // MFMAs of a phantom block on block 0's fragments and never writes that accumulator - the K loop stays free of branches.
        const int r = lane & 15, hb = (lane >> 4) & 1, h = lane >> 5;
        const int q = r >> 2, p = r & 3;
        f32x16 acc[CB * NT];
#pragma unroll
        for (int a = 0; a < CB * NT; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
        unsigned abase[CB], bbase[NT];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            abase[cb] = (unsigned)((cb < ncb ? cb0 + cb : 0) * (WM_UPX * 64) + (8 * h + q) * 64 + hb * 32 + 8 * p);
#pragma unroll
        for (int j = 0; j < NT; ++j) {                        // dy pixel of x pixel (row, c) under tap (ky, kx): (row + 2 - ky, c + 2 - kx)
            const int t = tap0 + j, ky = t / 3, kx = t - 3 * ky;
            bbase[j] = (unsigned)(YOFF + ((2 - ky) * WM_HW + 2 - kx + 8 * h + q) * 64 + hb * 32 + 8 * p);
        }
        auto tr2 = [&](const unsigned char* a) -> bf16x8 {
            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a);
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + 256));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        // The four K steps (unit row rl, half kq) of a unit, unrolled; the fragments of step k + 1 are read before the MFMAs of
        // step k (pinned: the scheduler otherwise sinks every read to its first use) - ACROSS units too: the barrier that hands
        // unit u + 1 over sits in front of the last step's MFMAs of unit u, whose fragments are in registers by then (every read
        // of unit u has been waited for), and the first fragments of unit u + 1 are read under those MFMAs.
        bf16x8 a[2][CB], b[2][NT];
        auto load_step = [&](int u, int k, int buf) {
            const unsigned char* xs = lds + (u & 1) * C::UBYTES + ((k >> 1) * TW + 16 * (k & 1)) * 64;
            const unsigned char* ys = lds + (u & 1) * WM_YBYTES + ((k >> 1) * WM_HW + 16 * (k & 1)) * 64;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) a[buf][cb] = tr2(xs + abase[cb]);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[buf][j] = tr2(ys + bbase[j]);
        };
        auto unit = [&](int u, Set& st) {                     // st holds this wave's pieces of unit u + 2
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cur = k & 1;
                if (k + 1 < 4) {
                    load_step(u, k + 1, cur ^ 1);
                } else if (u + 1 < U) {                       // (uniform)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's reads of unit u (and its stores of unit u + 1) are done ...
                    stamp();
                    __builtin_amdgcn_s_barrier();             // ... barrier u + 1: unit u + 1 is written, unit u's slot is free
                    stamp();
                    load_step(u + 1, 0, cur ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[cb * NT + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][cb], b[cur][j], acc[cb * NT + j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // Staging right behind the barrier's last MFMA batch, in EVERY wave at the same time: a memory instruction issued
            // while the SIMD streams MFMAs costs its wave 100+ cycles (s_memtime stamps: ten of them 800 - 1400 cycles per
            // wave and unit when placed in the middle of the unit, where the waves have drifted apart), so the eight waves' stores
            // and loads go where no MFMA is in flight: unit u + 2 into the slot unit u was read from, then the loads of unit u + 4.
            if (u + 2 < U) {                                  // (uniform)
                stamp();
                if (u + 3 < U) vm_wait<S>(); else vm_wait<0>();
                stamp();
                write(u + 2, st);
                if (u + 4 < U) load(u + 4, st);
                stamp();
            }
        };
        Set sa, sb;                                           // even / odd units
        if (U > 0) {
            load(0, sa);
            load(1, sb);
            vm_wait<S>();
            write(0, sa);
            if (2 < U) load(2, sa);
            if (2 < U) vm_wait<S>(); else vm_wait<0>();
            write(1, sb);
            if (3 < U) load(3, sb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // barrier 0: units 0 and 1 are in their slots
            load_step(0, 0, 0);
        }
#pragma unroll 1
        for (int u = 0; u < U; u += 2) {
            unit(u, sa);
            unit(u + 1, sb);
        }
        // ---- partial slabs [split][ci block][tap][ci 32][co 32] (the layout wgrad_reduce_kernel sums)
        // D[m][n]: n = co = lane & 31, m = ci = (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if (cb >= ncb) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float* part = d.workspace + ((size_t)(blockIdx.x * NCI + cb0 + cb) * 9 + tap0 + j) * (WG_C * WG_C);
                const f32x16 v = acc[cb * NT + j];
#pragma unroll
                for (int e = 0; e < 16; ++e) part[((e & 3) + 8 * (e >> 2) + 4 * h) * WG_C + (lane & 31)] = v[e];
            }
        }
    };
    if (three) role(std::integral_constant<int, 3>{}, std::false_type{});
    else role(std::integral_constant<int, 2>{}, std::true_type{});

    // ---- bias partials bias_part[split][32]: the dy-staging threads' sums (thread dt: channels 8 (dt & 3) .. + 7) meet in LDS
    if (d.dbias != nullptr) {                                 // (uniform)
        float* scratch = reinterpret_cast<float*>(lds);       // [384][8]
        __syncthreads();                                      // every wave is done with the LDS images
        if (!three) {
#pragma unroll
            for (int e = 0; e < 8; ++e) scratch[8 * dt + e] = bsum[e];
        }
        __syncthreads();
        if (tid < 32) {
            const int piece = tid >> 3, e = tid & 7;          // channel tid = 8 piece + e
            float sum = 0.f;
            for (int k = piece; k < WM_DYT; k += 4) sum += scratch[8 * k + e];
            float* bp = d.workspace + (size_t)WGRAD_MAX_SLABS * 9 * WG_C * WG_C;
            bp[(size_t)blockIdx.x * WG_C + tid] = sum;
        }
    }
    (void)dyw;
}

// ---------------------------------------------------------------- 1x1 (the blocks' local feature fusion, lff)
// The same idea for a 1x1 convolution (ResidualDenseBlock.lff, super_resolution.py:245-253: 224 -> 64): the (pixel split, 64-ci
// chunk) kernel re-reads the 64-channel dy once per chunk - 1.54x the algorithmic bytes at an HBM-saturated 6.5 TB/s
// (profiles/r03_cfg2_traffic_pmc.json) - and here every input channel meets dy in one workgroup: x and dy are each read exactly
// once, no halo.  Few MFMAs (one tap), so all eight waves are alike: each stages its share of the 64-pixel unit (x: [block][64
// px][64 B] as above, dy: [co block][64 px][64 B]) two units ahead in two register sets and, if it owns a 32-channel input
// block (wave w < NCI), multiplies it with every co block.  Two workgroups per CU.  bias: every thread sums the dy piece it staged.
template <int NCI, int NCO>
__global__ __launch_bounds__(WM_NTHR, 4) void wgrad1_m32_kernel(const nvq_wgrad_desc d, int tilesX, int tilesY, int ntiles) {
    constexpr int UBYTES = WM_UPX * NCI * 64, YB = WM_UPX * NCO * 64;
    constexpr int XPT = (WM_UPX * NCI * 4 + WM_NTHR - 1) / WM_NTHR;   // x pieces per thread and unit
    constexpr int YTHR = WM_UPX * NCO * 4;                              // threads that stage one dy piece each (256 / 512)
    static_assert(NCI <= 8 && NCO <= 2 && 2 * (UBYTES + YB) <= 80 * 1024, "1x1 configuration");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (UBYTES + YB)];
    constexpr int YOFF = 2 * UBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d.h, W = d.w;
    const int G = gridDim.x;
    const int mytiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / G + 1 : 0;
    const int U = 2 * mytiles;
    const __bf16* x16 = reinterpret_cast<const __bf16*>(d.x);
    const __bf16* dy16 = reinterpret_cast<const __bf16*>(d.dy);
    const int LU = d.x_plane ? d.x_ld >> 5 : NCI;            // blocks of the leading tensor (interleaved x: all of them)

    // x piece j = tid + 512 k: block (tid >> 8) + 2 k, unit row (tid >> 7) & 1 (both wave-uniform), column (tid >> 2) & 31
    const int xcol = (tid >> 2) & 31, xsub = tid & 3, xrow = (wave >> 1) & 1, xblk0 = wave >> 2;
    const unsigned v_lead = (unsigned)((xcol * d.x_ld + 8 * xsub) * 2), v_plane = (unsigned)((xcol * 32 + 8 * xsub) * 2);
    // dy piece tid (< YTHR): pixel tid / (4 NCO) -> row (wave-uniform), column; 16-byte piece tid % (4 NCO) of its NCO * 64 B
    const int ypx = tid / (4 * NCO), yq = tid % (4 * NCO), yrow = NCO == 2 ? wave >> 2 : (wave >> 1) & 1, ycol = ypx & 31;
    const bool ythr = tid < YTHR;                             // (wave-uniform)
    const unsigned v_y = (unsigned)((ycol * d.dy_ld + 8 * yq) * 2);
    struct Set { u32x4 x[XPT]; u32x4 y; bool xok, yok; };
    constexpr int S = XPT + 1;                                // loads per unit and thread
    int c_n = 0, c_ty = 0, c_tx = 0, c_unit = 0;             // (see wgrad_m32_kernel)
    {
        int bt = xcd_tile((int)blockIdx.x, ntiles);
        c_tx = bt % tilesX; bt /= tilesX;
        c_ty = bt % tilesY;
        c_n = bt / tilesY;
    }
    const int c_step = G >> 3;
    auto load = [&](int u, Set& st) {
        if ((u >> 1) != (c_unit >> 1)) {
            if ((G & 7) == 0) {
                c_tx += c_step;
                while (c_tx >= tilesX) { c_tx -= tilesX; ++c_ty; }
                while (c_ty >= tilesY) { c_ty -= tilesY; ++c_n; }
            } else {
                int bt = xcd_tile((int)blockIdx.x + (u >> 1) * G, ntiles);
                c_tx = bt % tilesX; bt /= tilesX;
                c_ty = bt % tilesY;
                c_n = bt / tilesY;
            }
        }
        c_unit = u;
        const int y0 = c_ty * WM_TR + 2 * (u & 1), x0 = c_tx * TW;
        {
            const bool row_ok = y0 + xrow < H;                // (wave-uniform)
            const unsigned rel = (unsigned)((c_n * H + (row_ok ? y0 + xrow : y0)) * W + x0);
            st.xok = row_ok && xcol < W - x0;
            const unsigned vl = st.xok ? v_lead : 0u, vp = st.xok ? v_plane : 0u;
#pragma unroll
            for (int k = 0; k < XPT; ++k) {
                const int blk = xblk0 + 2 * k < NCI ? xblk0 + 2 * k : 0;
                if (blk < LU) gload16s(st.x[k], x16 + ((size_t)rel * (unsigned)d.x_ld + d.x_coff + 32u * blk), vl);
                else gload16s(st.x[k], x16 + ((size_t)blk * d.x_plane + rel * 32u), vp);
            }
        }
        {
            const bool row_ok = y0 + yrow < H;
            const size_t rel = (size_t)((c_n * H + (row_ok ? y0 + yrow : y0)) * W + x0);
            st.yok = ythr && row_ok && ycol < W - x0;
            gload16s(st.y, dy16 + (rel * d.dy_ld + d.dy_coff), st.yok ? v_y : 0u);
        }
    };
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    auto write = [&](int u, Set& st) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        unsigned char* slot = lds + (u & 1) * UBYTES + tid * 16;
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            hold(st.x[k]);
            if (xblk0 + 2 * k < NCI) *reinterpret_cast<u32x4*>(slot + k * (WM_NTHR * 16)) = st.xok ? st.x[k] : z;
        }
        hold(st.y);
        if (ythr) {                                           // (wave-uniform)
            const u32x4 v = st.yok ? st.y : z;
            *reinterpret_cast<u32x4*>(lds + YOFF + (u & 1) * YB + ((yq >> 2) * WM_UPX + ypx) * 64 + (yq & 3) * 16) = v;
            bsum[0] += bf_lo(v[0]); bsum[1] += bf_hi(v[0]); bsum[2] += bf_lo(v[1]); bsum[3] += bf_hi(v[1]);
            bsum[4] += bf_lo(v[2]); bsum[5] += bf_hi(v[2]); bsum[6] += bf_lo(v[3]); bsum[7] += bf_hi(v[3]);
        }
    };
    const int r = lane & 15, hb = (lane >> 4) & 1, h = lane >> 5;
    const int q = r >> 2, p = r & 3;
    const unsigned fbase = (unsigned)((8 * h + q) * 64 + hb * 32 + 8 * p);
    f32x16 acc[NCO];
#pragma unroll
    for (int a = 0; a < NCO; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    auto tr2 = [&](const unsigned char* a) -> bf16x8 {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a);
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + 256));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    Set sa, sb;                                               // even / odd units
    auto unit = [&](int u, Set& st) {                         // st holds this thread's pieces of unit u + 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // barrier u: unit u is in its slot, unit u - 1's slot is free
        if (wave < NCI) {                                     // (wave-uniform)
            const unsigned char* xs = lds + (u & 1) * UBYTES + wave * (WM_UPX * 64) + fbase;
            const unsigned char* ys = lds + YOFF + (u & 1) * YB + fbase;
            bf16x8 a[4], b[4][NCO];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] = tr2(xs + k * (16 * 64));
#pragma unroll
                for (int co = 0; co < NCO; ++co) b[k][co] = tr2(ys + co * (WM_UPX * 64) + k * (16 * 64));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int co = 0; co < NCO; ++co) acc[co] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k][co], acc[co], 0, 0, 0);
        }
        if (u + 1 < U) {
            if (u + 2 < U) vm_wait<S>(); else vm_wait<0>();
            write(u + 1, st);
            if (u + 3 < U) load(u + 3, st);
        }
    };
    if (U > 0) {
        load(0, sa);
        load(1, sb);
        vm_wait<S>();
        write(0, sa);
        if (2 < U) load(2, sa);
    }
#pragma unroll 1
    for (int u = 0; u < U; u += 2) {
        unit(u, sb);
        unit(u + 1, sa);
    }
    // partial slabs [split][ci block][co block][ci 32][co 32]
    if (wave < NCI) {
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
            float* part = d.workspace + ((size_t)(blockIdx.x * NCI + wave) * NCO + co) * (WG_C * WG_C);
#pragma unroll
            for (int e = 0; e < 16; ++e) part[((e & 3) + 8 * (e >> 2) + 4 * h) * WG_C + (lane & 31)] = acc[co][e];
        }
    }
    if (d.dbias != nullptr) {                                 // thread tid < YTHR: channels 8 yq .. + 7
        float* scratch = reinterpret_cast<float*>(lds);       // [YTHR][8]
        __syncthreads();
        if (ythr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) scratch[8 * tid + e] = bsum[e];
        }
        __syncthreads();
        if (tid < NCO * 32) {
            const int piece = tid >> 3, e = tid & 7;
            float sum = 0.f;
            for (int k = piece; k < YTHR; k += 4 * NCO) sum += scratch[8 * k + e];
            float* bp = d.workspace + (size_t)WGRAD_MAX_SLABS * 9 * WG_C * WG_C;
            bp[((size_t)blockIdx.x * NCO + (tid >> 5)) * WG_C + (tid & 31)] = sum;
        }
    }
}

// Does an all-ci kernel take this launch?  bf16 x and dy, whole 32-channel blocks that all get a gradient: 3x3 with cout = 32,
// 64 .. 192 input channels and a slice-planar x; 1x1 with 64 .. 256 input and 32 / 64 output channels.
bool wgrad_m32_takes(const nvq_wgrad_desc& d) {
    if (!(d.x_bf16 && d.dy_bf16) || d.cin_w != d.cin || d.cin % 32 != 0 || d.cout % 32 != 0) return false;
    const int nci = d.cin / 32, nco = d.cout / 32;
    // 32-bit element offsets: the whole x buffer, and dy
    if ((size_t)(d.x_plane ? (size_t)nci * d.x_plane : (size_t)d.n * d.h * d.w * d.x_ld) >= ((size_t)1 << 32) ||
        (size_t)d.n * d.h * d.w * d.dy_ld >= ((size_t)1 << 31))
        return false;
    // persistent workgroups, each leaving its own partial slabs: with fewer than four tiles per workgroup (the 64x64 clips of
    // the continual-learning loop: 256 tiles) the prologue and the reduce over 256 / 512 slabs cost more than the kernel saves
    // (4.19 -> 4.70 ms per step there), and the split kernel with its few pixel splits stays
    const long ntiles = (long)((d.w + TW - 1) / TW) * ((d.h + WM_TR - 1) / WM_TR) * d.n;
    if ((d.variant & 15) != 2 && ntiles < 4 * (d.ksize == 3 ? 256 : 512)) return false;
    // (3x3 with 64 input channels: equal in isolation, 15 % slower than the split kernel inside the training step)
    if (d.ksize == 3) return d.x_plane && d.cout == 32 && nci >= 3 && nci <= 6 && d.cin >= d.x_ld;
    // 1x1: 64 .. 256 input channels, 32 / 64 output channels; slice-planar or interleaved x
    return nci >= 2 && nci <= 8 && nco <= 2 && (!d.x_plane || d.cin >= d.x_ld);
}

// returns the number of pixel splits (= workgroups) used, < 0 on error
int conv_wgrad_m32(const nvq_wgrad_desc& d, hipStream_t s) {
    const int tilesX = (d.w + TW - 1) / TW, tilesY = (d.h + WM_TR - 1) / WM_TR;
    const int ntiles = tilesX * tilesY * d.n;
    const int nci = d.cin / 32, nco = d.cout / 32;
    int wgs = d.ksize == 3 ? 256 : 512;                       // one / two persistent workgroups per CU
    if (wgs > ntiles) wgs = ntiles;
    if (wgs >= 8) wgs &= ~7;                                  // multiple of the XCD count: see xcd_tile()
    NVQ_REQUIRE((size_t)wgs * nci * nco * d.ksize * d.ksize <= (size_t)WGRAD_MAX_SLABS * 9 && wgs <= 512,
                "conv_wgrad(bf16, all-ci): %d workgroups x %d x %d blocks exceed the workspace", wgs, nci, nco);
    if (d.ksize == 1) {
#define NVQ_LAUNCH_W1(NCI, NCO) \
    hipLaunchKernelGGL((wgrad1_m32_kernel<NCI, NCO>), dim3(wgs), dim3(WM_NTHR), 0, s, d, tilesX, tilesY, ntiles)
#define NVQ_W1_ROW(NCI) do { if (nco == 1) NVQ_LAUNCH_W1(NCI, 1); else NVQ_LAUNCH_W1(NCI, 2); } while (0)
        switch (nci) {
            case 2: NVQ_W1_ROW(2); break;
            case 3: NVQ_W1_ROW(3); break;
            case 4: NVQ_W1_ROW(4); break;
            case 5: NVQ_W1_ROW(5); break;
            case 6: NVQ_W1_ROW(6); break;
            case 7: NVQ_W1_ROW(7); break;
            default: NVQ_W1_ROW(8); break;
        }
#undef NVQ_W1_ROW
#undef NVQ_LAUNCH_W1
        return wgs;
    }
#define NVQ_LAUNCH_WM(NCI) hipLaunchKernelGGL((wgrad_m32_kernel<NCI>), dim3(wgs), dim3(WM_NTHR), 0, s, d, tilesX, tilesY, ntiles)
    switch (nci) {
        case 2: NVQ_LAUNCH_WM(2); break;
        case 3: NVQ_LAUNCH_WM(3); break;
        case 4: NVQ_LAUNCH_WM(4); break;
        case 5: NVQ_LAUNCH_WM(5); break;
        default: NVQ_LAUNCH_WM(6); break;
    }
#undef NVQ_LAUNCH_WM
    return wgs;
}

}  // namespace nvq
