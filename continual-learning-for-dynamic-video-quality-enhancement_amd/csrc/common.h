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

// Second stage of every two-stage reduction in the library:
// out[k] = alpha * sum_b part[b*K + k] (+ out[k]); summed in double, fixed order => deterministic.
int launch_reduce_partials(const float* part, int nblk, int K, float alpha, float* out,
                           int accumulate, hipStream_t s);

}  // namespace nvq
