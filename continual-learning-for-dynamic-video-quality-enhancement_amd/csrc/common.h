// Internal helpers shared by the libnvq translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/nvq.h"

namespace nvq {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return NVQ_ELAUNCH;
    }
    return NVQ_OK;
}

#define NVQ_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            nvq::set_error(__VA_ARGS__);  \
            return NVQ_EINVAL;            \
        }                                 \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// 4 consecutive channels at element index `idx` of an fp32 or bf16 activation buffer
__device__ __forceinline__ float4 ldx4(const float* base, size_t idx, int is_bf16) {
    if (is_bf16) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(base) + idx);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
    return ld4(base + idx);
}
__device__ __forceinline__ void stx4(float* base, size_t idx, int is_bf16, float4 v) {
    if (is_bf16)
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + idx) =
            (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    else
        st4(base + idx, v);
}

// Index decomposition without 64-bit divisions.  A 64-bit '/' or '%' by a runtime value is a ~150-instruction software
// routine on gfx9; a launch of fewer than 2^32 work items (`total`, uniform) splits its linear id with 32-bit unsigned
// arithmetic instead.  idiv(a, d, total) = a / d for 0 <= a < total.
__device__ __forceinline__ long idiv(long a, int d, long total) {
    return total <= 0xffffffffL ? (long)((unsigned)a / (unsigned)d) : a / d;
}
__device__ __forceinline__ long idiv(long a, long d, long total) {
    return (total <= 0xffffffffL && d <= 0xffffffffL) ? (long)((unsigned)a / (unsigned)d) : a / d;
}

// Sum over the lanes of a wave that share the same (lane / width) group, width = power of two <= 64.
__device__ __forceinline__ float group_sum(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum of `v` over 256 threads; result valid in every thread. scratch: >= 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* scratch) {
    v = group_sum(v, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// Workgroups are dispatched round-robin over the 8 XCDs (linear id % 8), each with its own 4 MiB L2.  Tile kernels map
// workgroup `bid` of `n` to tile xcd_tile(bid, n): a bijection on [0, n) that gives every XCD one contiguous range of
// tiles, so the halo a tile shares with its neighbours is an L2 hit instead of a second HBM read.  Persistent kernels
// pass bid + k * gridDim.x (gridDim.x a multiple of 8): a workgroup then walks its own XCD's range.
__device__ __forceinline__ int xcd_tile(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7;
    return x * q + (x < r ? x : r) + (bid >> 3);
}

// Second stage of every two-stage reduction in the library:
// out[k] = alpha * sum_b part[b*K + k] (+ out[k]); summed in double, fixed order => deterministic.
int launch_reduce_partials(const float* part, int nblk, int K, float alpha, float* out,
                           int accumulate, hipStream_t s);

}  // namespace nvq
