// Diagnostic only (not part of libnvq): how many bytes per second does ONE workgroup per CU pull from HBM as a function of the
// number of waves that issue the loads and of the 16-byte loads each lane keeps in flight?  Every wave instruction reads 1 KiB
// contiguous; a wave streams its own 1/(waves * workgroups) share of a buffer far larger than the Infinity Cache, D loads deep
// (a ring of D registers: issue load i + D, then consume load i).
// usage: wave_loader            -> table over waves in {1, 2, 4, 8} x D in {4, 8, 16, 24, 32}
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ void loader(const u32x4* __restrict__ src, size_t per_wave_vecs, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t wid = (size_t)blockIdx.x * nw + wave;
    const u32x4* p = src + wid * per_wave_vecs + lane;
    const int iters = (int)(per_wave_vecs / 64);
    u32x4 r[D];
    u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < D; ++i) r[i] = p[(size_t)i * 64];
    for (int i = 0; i + D < iters; i += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            acc ^= r[j];
            r[j] = p[(size_t)(i + D + j) * 64];
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) acc ^= r[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main() {
    const size_t bytes = (size_t)6 << 30;
    u32x4* buf; unsigned* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 256;
    printf("one workgroup per CU (256), 6 GiB streamed once; GB/s per CU and TB/s total\n");
    for (int nw : {1, 2, 4, 8}) {
        for (int D : {4, 8, 16, 24, 32}) {
            const size_t per_wave_vecs = (bytes / 16 / ((size_t)wgs * nw)) / 64 * 64;
            auto launch = [&]() {
                switch (D) {
                    case 4: hipLaunchKernelGGL(loader<4>, dim3(wgs), dim3(64 * nw), 0, 0, buf, per_wave_vecs, sink); break;
                    case 8: hipLaunchKernelGGL(loader<8>, dim3(wgs), dim3(64 * nw), 0, 0, buf, per_wave_vecs, sink); break;
                    case 16: hipLaunchKernelGGL(loader<16>, dim3(wgs), dim3(64 * nw), 0, 0, buf, per_wave_vecs, sink); break;
                    case 24: hipLaunchKernelGGL(loader<24>, dim3(wgs), dim3(64 * nw), 0, 0, buf, per_wave_vecs, sink); break;
                    default: hipLaunchKernelGGL(loader<32>, dim3(wgs), dim3(64 * nw), 0, 0, buf, per_wave_vecs, sink); break;
                }
            };
            launch();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double moved = (double)per_wave_vecs * 16 * wgs * nw;
            printf("waves %d  loads in flight per lane %2d (%5.1f KB per CU): %6.1f GB/s per CU  %5.2f TB/s\n", nw, D,
                   nw * 64.0 * D * 16 / 1024, moved / ms / 1e6 / wgs, moved / ms / 1e9);
        }
    }
    return 0;
}
