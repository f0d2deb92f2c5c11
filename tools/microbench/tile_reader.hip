// Diagnostic only (not part of libnvq): how fast can a CU-filling grid READ the activation bytes of the 3x3 bf16 conv in
// the conv's own access pattern, with nothing else going on?
//   mode 0: NHWC tile pattern - workgroup = 16x32-pixel tile (+halo), per 32-channel chunk each thread loads 16-B pieces,
//           4 lanes per pixel (64 B per pixel, pixel stride ld*2 bytes), chunk loop, XOR-reduce so nothing is elided.
//   mode 1: the same number of bytes per workgroup read as one contiguous block per chunk.
//   mode 2: NHWC tile pattern, all chunks of a pixel in one pass (8 lanes x 16 B x (cin/64) per pixel: whole lines).
// usage: tile_reader <mode> <N> <H> <W> <ld> <cin>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void reader(const unsigned short* __restrict__ x, int H, int W, int ld, int cin, int tilesX,
                                              int tilesY, unsigned* __restrict__ sink) {
    const int tid = threadIdx.x;
    int bt = blockIdx.x;
    const int n8 = gridDim.x;
    { const int q = n8 >> 3, r = n8 & 7, xx = bt & 7; bt = xx * q + (xx < r ? xx : r) + (bt >> 3); }
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    u32x4 acc = {0u, 0u, 0u, 0u};
    const int nkc = cin / 32;
    if (MODE == 0) {
        size_t off[5]; bool ok[5];
        for (int k = 0; k < 5; ++k) {
            const int item = tid + k * 512, hp = item >> 2, hy = hp / 34, hx = hp - hy * 34;
            const int gy = ty * 16 + hy - 1, gx = tx * 32 + hx - 1;
            ok[k] = item < 612 * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            off[k] = ok[k] ? ((size_t)(n * H + gy) * W + gx) * ld + 8 * (item & 3) : 0;
        }
        for (int kc = 0; kc < nkc; ++kc) {
            u32x4 v[5];
            for (int k = 0; k < 5; ++k) v[k] = *reinterpret_cast<const u32x4*>(x + off[k] + kc * 32);
            for (int k = 0; k < 5; ++k) acc ^= v[k];
        }
    } else if (MODE >= 3) {
        // 3: mode 0 + the weight slab of every chunk (20 KB, shared by all workgroups: L2 hits)
        // 4: 3 + everything stored to LDS, one barrier pair per chunk (loads of chunk kc+1 issued before the stores of kc)
        // 5: 4 without the weights
        __shared__ __attribute__((aligned(16))) u32x4 lds[612 * 6 + 1280];
        size_t off[5]; bool ok[5];
        for (int k = 0; k < 5; ++k) {
            const int item = tid + k * 512, hp = item >> 2, hy = hp / 34, hx = hp - hy * 34;
            const int gy = ty * 16 + hy - 1, gx = tx * 32 + hx - 1;
            ok[k] = item < 612 * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            off[k] = ok[k] ? ((size_t)(n * H + gy) * W + gx) * ld + 8 * (item & 3) : 0;
        }
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(x) + ((size_t)1 << 22);     // some L2-resident 120 KB
        u32x4 v[5], w[3];
        auto fetch = [&](int kc) {
            for (int k = 0; k < 5; ++k) v[k] = *reinterpret_cast<const u32x4*>(x + off[k] + kc * 32);
            if (MODE != 5) { w[0] = wsrc[kc * 1280 + tid]; w[1] = wsrc[kc * 1280 + 512 + tid]; if (tid < 256) w[2] = wsrc[kc * 1280 + 1024 + tid]; }
        };
        fetch(0);
        for (int kc = 0; kc < nkc; ++kc) {
            if (MODE >= 4) {
                __syncthreads();
                for (int k = 0; k < 5; ++k) { const int item = tid + k * 512; if (item < 2448) lds[(item >> 2) * 6 + (item & 3)] = v[k]; }
                if (MODE != 5) { lds[612 * 6 + tid] = w[0]; lds[612 * 6 + 512 + tid] = w[1]; if (tid < 256) lds[612 * 6 + 1024 + tid] = w[2]; }
                __syncthreads();
                acc ^= lds[(tid * 7) % (612 * 6)];
            } else {
                for (int k = 0; k < 5; ++k) acc ^= v[k];
                if (MODE != 5) acc ^= w[0] ^ w[1] ^ w[2];
            }
            if (kc + 1 < nkc) fetch(kc + 1);
        }
        // 6: + the conv's output stores: per wave 2 rows x 32 px x 32 bf16 channels as 8 stores of 8 B per lane
        //    (lane (c, g): pixel c of a 16-pixel block, channels 4g..4g+3 of a 16-channel block) into channels [cin, cin+32)
        // 7: the same bytes as 4 stores of 16 B per lane (4 lanes per pixel = 64 contiguous bytes)
        if (MODE == 6 || MODE == 7 || MODE == 8 || MODE == 9) {
            const int lane = tid & 63, wave = tid >> 6;
            unsigned short* o = const_cast<unsigned short*>(x);
            if (MODE == 6) {
                const int c = lane & 15, g = lane >> 4;
                for (int pb = 0; pb < 4; ++pb)
                    for (int cb = 0; cb < 2; ++cb) {
                        const int gy = ty * 16 + 2 * wave + (pb >> 1), gx = tx * 32 + (pb & 1) * 16 + c;
                        if (gy < H && gx < W) {
                            typedef unsigned u2 __attribute__((ext_vector_type(2)));
                            *reinterpret_cast<u2*>(o + ((size_t)(n * H + gy) * W + gx) * ld + cin + cb * 16 + 4 * g) = (u2){acc[0], acc[1]};
                        }
                    }
            } else if (MODE == 9) {                       // 9: the bytes of mode 7 written as one contiguous 4 KB block per wave
                unsigned short* ob = o + (size_t)gridDim.x * 0 + ((size_t)1 << 28) + ((size_t)blockIdx.x * 8 + wave) * 2048;
                for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(ob + (lane + k * 64) * 8) = acc;
            } else if (MODE == 8) {                       // 8: whole 128-B lines: 8 lanes per pixel, 64 channels from `cin`
                for (int k = 0; k < 8; ++k) {
                    const int item = lane + k * 64, px = item >> 3, piece = item & 7;
                    const int gy = ty * 16 + 2 * wave + (px >> 5), gx = tx * 32 + (px & 31);
                    if (gy < H && gx < W) *reinterpret_cast<u32x4*>(o + ((size_t)(n * H + gy) * W + gx) * ld + cin + 8 * piece) = acc;
                }
            } else {
                for (int k = 0; k < 4; ++k) {
                    const int item = lane + k * 64, px = item >> 2, piece = item & 3;
                    const int gy = ty * 16 + 2 * wave + (px >> 5), gx = tx * 32 + (px & 31);
                    if (gy < H && gx < W) *reinterpret_cast<u32x4*>(o + ((size_t)(n * H + gy) * W + gx) * ld + cin + 8 * piece) = acc;
                }
            }
        }
    } else if (MODE == 1) {
        const size_t base = (size_t)blockIdx.x * nkc * 612 * 32;       // contiguous 612*64 B per chunk
        for (int kc = 0; kc < nkc; ++kc) {
            u32x4 v[5];
            for (int k = 0; k < 5; ++k) {
                const int item = tid + k * 512;
                v[k] = *reinterpret_cast<const u32x4*>(x + base + (size_t)kc * 612 * 32 + (item < 2448 ? item : 0) * 8);
            }
            for (int k = 0; k < 5; ++k) acc ^= v[k];
        }
    } else {
        for (int item = tid; item < 612 * 8; item += 512) {
            const int hp = item >> 3, hy = hp / 34, hx = hp - hy * 34;
            const int gy = ty * 16 + hy - 1, gx = tx * 32 + hx - 1;
            if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
            const size_t o = ((size_t)(n * H + gy) * W + gx) * ld + 8 * (item & 7);
            for (int c0 = 0; c0 + 64 <= cin; c0 += 64) acc ^= *reinterpret_cast<const u32x4*>(x + o + c0);
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
    const int mode = atoi(argv[1]), N = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]), ld = atoi(argv[5]), cin = atoi(argv[6]);
    const size_t elems = std::max((size_t)N * H * W * ld, ((size_t)1 << 28) + (size_t)N * H * W * 64) + (1 << 20);
    unsigned short* x; unsigned* sink;
    hipMalloc(&x, elems * 2); hipMalloc(&sink, 4);
    std::vector<unsigned short> h(1 << 20);
    for (auto& v : h) v = (unsigned short)(rand() & 0x7fff);
    for (size_t o = 0; o < elems; o += h.size()) hipMemcpy(x + o, h.data(), std::min(h.size(), elems - o) * 2, hipMemcpyHostToDevice);
    const int tilesX = (W + 31) / 32, tilesY = (H + 15) / 16, ntiles = tilesX * tilesY * N;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(reader<0>, dim3(ntiles), dim3(512), 0, 0, x, H, W, ld, cin, tilesX, tilesY, sink);
        else if (mode == 3) hipLaunchKernelGGL(reader<3>, dim3(ntiles), dim3(512), 0, 0, x, H, W, ld, cin, tilesX, tilesY, sink);
        else if (mode == 4) hipLaunchKernelGGL(reader<4>, dim3(ntiles), dim3(512), 0, 0, x, H, W, ld, cin, tilesX, tilesY, sink);
        else if (mode == 6) hipLaunchKernelGGL(reader<6>, dim3(ntiles), dim3(512), 0, 0, x, H, W, ld, cin, tilesX, tilesY, sink);
        else if (mode == 9) hipLaunchKernelGGL(reader<9>, dim3(ntiles), dim3(512), 0, 0, x, H, W, ld, cin, tilesX, tilesY, sink);
        else if (mode == 8) hipLaunchKernelGGL(reader<8>, dim3(ntiles), dim3(512), 0, 0, x, H, W, ld, cin, tilesX, tilesY, sink);
        else if (mode == 7) hipLaunchKernelGGL(reader<7>, dim3(ntiles), dim3(512), 0, 0, x, H, W, ld, cin, tilesX, tilesY, sink);
        else if (mode == 5) hipLaunchKernelGGL(reader<5>, dim3(ntiles), dim3(512), 0, 0, x, H, W, ld, cin, tilesX, tilesY, sink);
        else if (mode == 1) hipLaunchKernelGGL(reader<1>, dim3(ntiles), dim3(512), 0, 0, x, H, W, ld, cin, tilesX, tilesY, sink);
        else hipLaunchKernelGGL(reader<2>, dim3(ntiles), dim3(512), 0, 0, x, H, W, ld, cin, tilesX, tilesY, sink);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / 20 * 1e3;
    const double alg = (double)N * H * W * cin * 2, staged = (double)ntiles * 612 * cin * 2;
    printf("mode %d N%d %dx%d ld%d cin%d: %.1f us  algorithmic %.2f TB/s  staged (with halo) %.2f TB/s\n", mode, N, H, W, ld, cin, us,
           alg / us / 1e6, staged / us / 1e6);
    return 0;
}
