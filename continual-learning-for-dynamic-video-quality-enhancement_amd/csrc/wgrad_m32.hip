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
//   * six CONSUMER waves hold the accumulators (cin x 32 x 9 fp32 = 221 KB for cin = 192): wave (grp, ky) owns the three kx
//     taps of row ky for CB 32-channel blocks; M = ci, N = co, K = 16 pixels of a unit row, both operands transposed out of
//     [pixel][channel] LDS images by ds_read_b64_tr_b16, the fragments of K step k + 1 read under the MFMAs of step k - across
//     units too: the barrier that hands unit u + 1 over sits in front of unit u's last MFMA batch;
//   * two more waves stage dy (with its halo) and sum it for the bias gradient (fp32, from the staged values);
//   * ALL eight waves stage x: three 16-byte loads and three LDS stores per wave and unit, two units ahead in two register sets
//     (unit u + 1 is stored into the slot the consumers left one barrier ago, then the loads of unit u + 3 are issued).
// Loads are inline asm (uniform base + per-lane offset: one instruction) with counted s_waitcnt vmcnt by hand: with register
// sets that rotate over loop iterations the compiler's own bookkeeping drains vmcnt(0) in front of every store pass.
//
// How it got here (cin = 192, 8 x 540 x 960; profiles/r04_wgrad_all_ci.txt): (1) every wave staging a whole 4-row tile through
// registers spills behind 144 accumulators; (2) both tiles by LDS-DMA (global_load_lds_dwordx4, three units ahead, counted
// vmcnt): correct, 1.08x the bytes, but DMA alone 318 us + MFMAs alone 392 us = 582 us together - a DMA piece costs the issuing
// wave ~200 cycles in an MFMA phase whether its data comes from HBM or L2; (3) two dedicated producer waves (registers, two or
// three units ahead): s_memtime stamps show them taking 3400 cycles per unit for 13 loads + 13 LDS stores each while the
// consumers compute for 1500 and wait - ANY memory instruction issued next to a dense MFMA stream costs its wave 100 - 200
// cycles, so the instruction count per wave is what matters: spread over all eight waves 496 us, the dy waves at s_setprio 3
// 482 - 492 us against 565 - 595 us of the split kernel.  What bounds it now: the two SIMDs that hold two consumers each run
// 2 x 36 MFMAs = 2304 cycles per unit at the ~1.5 - 1.75 GHz the chip holds under this load (MFMAs alone, nothing staged: 372 us).
// Partial slabs / bias partials in the layout of wgrad_bf16_kernel; the same reduce kernel finishes.
//
// LDS images.  x unit: [32-channel block][64 pixels][64 B] (the four pixels x two 16-channel blocks of a half-wave's transposing
// read are 256 contiguous bytes = all 64 banks); a load instruction covers 16 pixels of one block (a compact plane: 1 KiB
// contiguous in memory; a block of the leading tensor: 64 B of every pixel).  dy of a unit: [4 x 34 halo pixels][64 B].
#include "conv_common.h"

namespace nvq {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

constexpr int WM_TR = 4;                 // tile rows
constexpr int WM_NCONS = 6, WM_NPROD = 2;
constexpr int WM_NTHR = 64 * (WM_NCONS + WM_NPROD);
constexpr int WM_PT = 64 * WM_NPROD;     // producer threads
constexpr int WM_HW = TW + 2, WM_HH = 2 + 2, WM_NPY = WM_HW * WM_HH;       // dy of a unit with halo: 34 x 4
constexpr int WM_YBYTES = WM_NPY * 64;
constexpr int WM_UPX = 2 * TW;           // pixels of a unit (two tile rows)
constexpr int WM_YPP = (WM_NPY * 4 + WM_PT - 1) / WM_PT;                    // dy pieces per producer thread and unit (5)

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
    static constexpr int CB = NCI <= 2 ? 1 : (NCI <= 4 ? 2 : 3);       // 32-channel blocks per consumer wave
    static constexpr int UBYTES = WM_UPX * NCI * 64;                    // x bytes of a unit
    static constexpr int XPIECES = WM_UPX * NCI * 4;                    // its 16-byte pieces ...
    static constexpr int XPT = (XPIECES + WM_NTHR - 1) / WM_NTHR;       // ... per thread (ALL eight waves stage x)
    static constexpr int LDS = 2 * UBYTES + 2 * WM_YBYTES;
};

}  // namespace

// cout = 32, cin = 32 NCI, slice-planar bf16 x (nvq_wgrad_desc::x_plane), bf16 dy.
template <int NCI>
__global__ __launch_bounds__(WM_NTHR, 2) void wgrad_m32_kernel(const nvq_wgrad_desc d, int tilesX, int tilesY, int ntiles) {
    using C = WmCfg<NCI>;
    constexpr int CB = C::CB, XPT = C::XPT;
    static_assert(C::LDS <= 160 * 1024 && C::LDS >= WM_PT * 32, "LDS");
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
    const int LU = d.x_ld >> 5;                               // 32-channel blocks of the leading tensor

    // ---- x staging, by every wave: piece j = tid + 512 k of a unit image [block][64 px][4 pieces] -> block (tid >> 8) + 2 k
    // (wave-uniform), unit row (tid >> 7) & 1 (wave-uniform), column (tid >> 2) & 31, piece tid & 3; its LDS address is linear
    // in j.  A load is one instruction: uniform base + one of two per-lane byte offsets.  (A memory instruction issued next to
    // a dense MFMA stream costs its wave ~100 cycles - s_memtime stamps: two dedicated producer waves needed 3400 cycles per
    // unit for 13 loads + 13 LDS stores each while the consumers computed for 1500 and waited - so the instructions are spread
    // over all eight waves: three loads and three stores per wave and unit.)
    const int xcol = (tid >> 2) & 31, xsub = tid & 3, xrow = (wave >> 1) & 1, xblk0 = wave >> 2;   // (row, block: from the SGPR)
    const unsigned v_lead = (unsigned)((xcol * d.x_ld + 8 * xsub) * 2), v_plane = (unsigned)((xcol * 32 + 8 * xsub) * 2);
    struct XSet { u32x4 x[XPT]; bool full, ok; };          // ok: this lane's pixel exists (always, in a full unit)
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
    auto unit_origin = [&](int u, int& n, int& y0, int& x0) {  // u = c_unit or c_unit + 1 (monotonic callers)
        if ((u >> 1) != (c_unit >> 1)) {
            if ((G & 7) == 0) {
                c_tx += c_step;
                while (c_tx >= tilesX) { c_tx -= tilesX; ++c_ty; }
                while (c_ty >= tilesY) { c_ty -= tilesY; ++c_n; }
            } else {                                          // (a handful of workgroups: decode from scratch)
                int bt = xcd_tile((int)blockIdx.x + (u >> 1) * G, ntiles);
                c_tx = bt % tilesX; bt /= tilesX;
                c_ty = bt % tilesY;
                c_n = bt / tilesY;
            }
        }
        c_unit = u;
        n = c_n; y0 = c_ty * WM_TR + 2 * (u & 1); x0 = c_tx * TW;
    };
    auto load_x = [&](int u, XSet& st) {                      // exactly XPT loads (a piece past the image / the unit: a dummy)
        int n, y0, x0;
        unit_origin(u, n, y0, x0);
        const bool row_ok = y0 + xrow < H;                    // (wave-uniform)
        const unsigned rel = (unsigned)((n * H + (row_ok ? y0 + xrow : y0)) * W + x0);
        st.full = x0 + TW <= W && y0 + 2 <= H;                // (uniform)
        st.ok = row_ok && xcol < W - x0;
        const unsigned vl = st.ok ? v_lead : 0u, vp = st.ok ? v_plane : 0u;
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int blk = xblk0 + 2 * k < NCI ? xblk0 + 2 * k : 0;       // (wave-uniform; past the unit: block 0 again, not stored)
            if (blk < LU) gload16s(st.x[k], x16 + (size_t)(rel * (unsigned)d.x_ld + 32u * blk), vl);
            else gload16s(st.x[k], x16 + ((size_t)blk * d.x_plane + rel * 32u), vp);
        }
    };
    auto write_x = [&](int u, XSet& st) {                     // (behind a vm_wait that covers st's loads)
        unsigned char* slot = lds + (u & 1) * C::UBYTES + tid * 16;
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            hold(st.x[k]);
            if (xblk0 + 2 * k < NCI) *reinterpret_cast<u32x4*>(slot + k * (WM_NTHR * 16)) = st.ok ? st.x[k] : z;
        }
    };
    XSet xa, xb;                                              // even / odd units
#ifdef NVQ_WM_STAMPS
    // (development build: s_memtime stamps of workgroup 0, wave w's k-th stamp at stamps[w * 2048 + k]; tools/wgrad_stamps2.py)
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(d.workspace + (size_t)WGRAD_MAX_SLABS * 9 * WG_C * WG_C + 512 * 256);   // (behind the official workspace)
    int nst = 0;
    const bool stamping = (d.variant & 128) && blockIdx.x == 0 && lane == 0;
    auto stamp = [&]() {
        if (stamping && nst < 2048) stamps[wave * 2048 + nst++] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [&]() {};
#endif
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

    if (wave >= WM_NCONS) {
        // ================================================================ the two waves that also stage dy and sum it
        const int pt = tid - 64 * WM_NCONS;                   // 0 .. 127
        __builtin_amdgcn_s_setprio(3);                        // their memory instructions ahead of the co-resident consumer's MFMAs (-4 %)
        const __bf16* dy16 = reinterpret_cast<const __bf16*>(d.dy);
        const int sub = pt & 3;
        // dy piece k: j = pt + 128 k -> halo pixel j >> 2 (row hr, column hc), piece pt & 3
        unsigned short ypos[WM_YPP];
        unsigned yvoff[WM_YPP];                               // byte offset of the piece from the unit's halo origin (y0 - 1, x0 - 1)
        unsigned yin = 0;                                     // bit k: piece k is one of the unit's own pixels (not halo)
#pragma unroll
        for (int k = 0; k < WM_YPP; ++k) {
            const int hp = (pt >> 2) + 32 * k;
            const int hr = hp / WM_HW, hc = hp - hr * WM_HW;
            ypos[k] = (unsigned short)(hp < WM_NPY ? hr << 8 | hc : 0xffff);
            yvoff[k] = hp < WM_NPY ? (unsigned)(((hr * W + hc) * d.dy_ld + 8 * sub) * 2) : 0u;
            yin |= ((hp < WM_NPY && hr >= 1 && hr <= 2 && hc >= 1 && hc <= TW) ? 1u : 0u) << k;
        }
        const int ysub = d.dy_coff + 8 * sub;
        struct YSet { u32x4 y[WM_YPP]; unsigned ymask, ymask_in; bool full; };
        YSet ya, yb;
        constexpr int S = XPT + WM_YPP;                       // loads per unit and thread (always exactly S: counted waits)
        // the unit's dy rows y0 - 1 .. y0 + 2 with a column of halo on either side (the two rows it shares with the
        // neighbouring unit are fetched by both: L2 hits one unit later); full: the whole halo lies inside the image
        auto load_y = [&](int u, YSet& st) {
            int n, y0, x0;
            unit_origin(u, n, y0, x0);
            const unsigned pix0 = (unsigned)((n * H + y0) * W + x0);
            st.full = x0 >= 1 && x0 + TW + 1 <= W && y0 >= 1 && y0 + 3 <= H;
            if (st.full) {
                const __bf16* ybase = dy16 + ((long)(pix0 - W - 1) * d.dy_ld + d.dy_coff);
#pragma unroll
                for (int k = 0; k < WM_YPP; ++k) gload16s(st.y[k], ybase, yvoff[k]);
            } else {
                const long ypix0 = (long)pix0 * d.dy_ld;
                st.ymask = 0; st.ymask_in = 0;
#pragma unroll
                for (int k = 0; k < WM_YPP; ++k) {
                    const int hr = ypos[k] >> 8, hc = ypos[k] & 0xff;
                    const int gy = y0 + hr - 1, gx = x0 + hc - 1;
                    const bool ok = ypos[k] != 0xffff && gy >= 0 && gy < H && gx >= 0 && gx < W;
                    st.ymask |= (ok ? 1u : 0u) << k;
                    st.ymask_in |= ((ok && ((yin >> k) & 1)) ? 1u : 0u) << k;
                    gload16(st.y[k], dy16 + (ok ? ypix0 + (((hr - 1) * W + (hc - 1)) * d.dy_ld + ysub) : (long)d.dy_coff));
                }
            }
        };
        auto write_y = [&](int u, YSet& st) {
            unsigned char* ys = lds + YOFF + (u & 1) * WM_YBYTES + pt * 16;
            const u32x4 z = {0u, 0u, 0u, 0u};
            const unsigned vmask = st.full ? 0xffffffffu : st.ymask, imask = st.full ? yin : st.ymask_in;
#pragma unroll
            for (int k = 0; k < WM_YPP; ++k) {
                hold(st.y[k]);
                if (ypos[k] == 0xffff) continue;
                const u32x4 v = (vmask >> k) & 1 ? st.y[k] : z;
                *reinterpret_cast<u32x4*>(ys + k * (WM_PT * 16)) = v;
                if ((imask >> k) & 1) {                       // the unit's own pixels: what the bias gradient sums, once per pixel
                    bsum[0] += bf_lo(v[0]); bsum[1] += bf_hi(v[0]); bsum[2] += bf_lo(v[1]); bsum[3] += bf_hi(v[1]);
                    bsum[4] += bf_lo(v[2]); bsum[5] += bf_hi(v[2]); bsum[6] += bf_lo(v[3]); bsum[7] += bf_hi(v[3]);
                }
            }
        };
        auto hand_over = [&]() {                              // this wave's LDS stores are done; meet the others
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        // while the consumers compute unit u: unit u + 1 -> the slot they left one barrier ago, then the loads of unit u + 3
        auto body = [&](int u, XSet& xs, YSet& ys) {          // the sets hold unit u + 1; behind them unit u + 2 was loaded
            if (u + 1 >= U) return;
            if (u + 2 < U) vm_wait<S>(); else vm_wait<0>();
            stamp();
            write_x(u + 1, xs);
            write_y(u + 1, ys);
            stamp();
            if (u + 3 < U) { load_x(u + 3, xs); load_y(u + 3, ys); }
            stamp();
            hand_over();                                      // barrier u + 1
            stamp();
        };
        if (U > 0) {
            load_x(0, xa); load_y(0, ya);
            load_x(1, xb); load_y(1, yb);
            vm_wait<S>();
            write_x(0, xa); write_y(0, ya);
            if (2 < U) { load_x(2, xa); load_y(2, ya); }
            hand_over();                                      // barrier 0: unit 0 is in its slot
#pragma unroll 1
            for (int u = 0; u < U; u += 2) {
                body(u, xb, yb);
                body(u + 1, xa, ya);
            }
        }
    } else {
        // ================================================================ consumers
        const int r = lane & 15, hb = (lane >> 4) & 1, h = lane >> 5;
        const int q = r >> 2, p = r & 3;                      // tr-read role of this lane inside its 16-lane group
        // wave (grp, ky) = the three kx taps of row ky for ci blocks [cb0, cb0 + ncb)
        const int ky = wave % 3;
        const int cb0 = (wave / 3) * CB;
        const int ncb = NCI - cb0 < CB ? NCI - cb0 : CB;
        f32x16 acc[CB * 3];
#pragma unroll
        for (int a = 0; a < CB * 3; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
        // A transposing read: lane (r = (q, p), hb) supplies the address of image row pix + q, columns 4p .. 4p+3 of a 16-channel
        // block and receives channel r of that block for pixels pix .. pix + 3; two reads (pix, pix + 4) = the 8 k values
        // 8h .. 8h + 7 of an operand.
        // A wave of the last group may own fewer than CB blocks (NCI = 3, 5): it runs the MFMAs of a phantom block on block 0's
        // fragments and never writes that accumulator - the K loop stays free of branches (with them the compiler's schedule is
        // read - wait - MFMA per fragment, the LDS latency exposed 24 times per tile row).
        unsigned abase[CB];                                   // byte offset of block cb's fragment for pixel 8h + q of a unit image
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            abase[cb] = (unsigned)((cb < ncb ? cb0 + cb : 0) * (WM_UPX * 64) + (8 * h + q) * 64 + hb * 32 + 8 * p);
        const unsigned bbase = (unsigned)(YOFF + ((2 - ky) * WM_HW + 8 * h + q + 2) * 64 + hb * 32 + 8 * p);
        auto tr2 = [&](const unsigned char* a, int stride4) -> bf16x8 {
            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a);
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + stride4));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        // The four K steps (unit row rl, half kq) of a unit, unrolled; the fragments of step k + 1 are read before the MFMAs of
        // step k (pinned: the scheduler otherwise sinks every read to its first use) - ACROSS units too: the barrier that hands
        // unit u + 1 over sits in front of the last step's MFMAs of unit u, whose fragments are in registers by then (every read
        // of unit u has been waited for), and the first fragments of unit u + 1 are read under those MFMAs.  This wave's share of
        // the x staging (unit u + 1 into its slot, the loads of unit u + 3) goes behind the MFMAs of step 1.
        bf16x8 a[2][CB], b[2][3];
        auto load_step = [&](int u, int k, int buf) {
            const unsigned char* xs = lds + (u & 1) * C::UBYTES;
            const unsigned char* ys = lds + (u & 1) * WM_YBYTES;   // (+ bbase)
            const int rl = k >> 1, kq = k & 1;
            const int pxo = rl * TW + 16 * kq;                // first pixel of the K step (+ 8h + q per lane: in abase)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) a[buf][cb] = tr2(xs + abase[cb] + pxo * 64, 256);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)                    // dy pixel of x pixel (row, c) under tap (ky, kx): (row + 2 - ky, c + 2 - kx)
                b[buf][kx] = tr2(ys + bbase + (rl * WM_HW + 16 * kq - kx) * 64, 256);
        };
        auto unit = [&](int u, XSet& xs) {                    // xs holds this wave's pieces of unit u + 1
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
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[cb * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][cb], b[cur][kx], acc[cb * 3 + kx], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (k == 1 && u + 1 < U) {                    // (uniform)
                    if (u + 2 < U) vm_wait<XPT>(); else vm_wait<0>();
                    write_x(u + 1, xs);
                    if (u + 3 < U) load_x(u + 3, xs);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (U > 0) {
            load_x(0, xa);
            load_x(1, xb);
            vm_wait<XPT>();
            write_x(0, xa);
            if (2 < U) load_x(2, xa);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // barrier 0: unit 0 is in its slot
            load_step(0, 0, 0);
        }
#pragma unroll 1
        for (int u = 0; u < U; u += 2) {
            unit(u, xb);
            unit(u + 1, xa);
        }
        // ---- partial slabs [split][ci block][tap][ci 32][co 32] (the layout wgrad_reduce_kernel sums)
        // D[m][n]: n = co = lane & 31, m = ci = (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if (cb >= ncb) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float* part = d.workspace + ((size_t)(blockIdx.x * NCI + cb0 + cb) * 9 + ky * 3 + kx) * (WG_C * WG_C);
                const f32x16 v = acc[cb * 3 + kx];
#pragma unroll
                for (int e = 0; e < 16; ++e) part[((e & 3) + 8 * (e >> 2) + 4 * h) * WG_C + (lane & 31)] = v[e];
            }
        }
    }
    // ---- bias partials bias_part[split][32]: the dy waves' sums (thread pt: channels 8 (pt & 3) .. + 7) meet in LDS
    if (d.dbias != nullptr) {                                 // (uniform)
        float* scratch = reinterpret_cast<float*>(lds);       // [128][8]
        __syncthreads();                                      // every wave is done with the LDS images
        if (wave >= WM_NCONS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) scratch[8 * (tid - 64 * WM_NCONS) + e] = bsum[e];
        }
        __syncthreads();
        if (tid < 32) {
            const int piece = tid >> 3, e = tid & 7;          // channel tid = 8 piece + e
            float sum = 0.f;
            for (int k = piece; k < WM_PT; k += 4) sum += scratch[8 * k + e];
            float* bp = d.workspace + (size_t)WGRAD_MAX_SLABS * 9 * WG_C * WG_C;
            bp[(size_t)blockIdx.x * WG_C + tid] = sum;
        }
    }
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
