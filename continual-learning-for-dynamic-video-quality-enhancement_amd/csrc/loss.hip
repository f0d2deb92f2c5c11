// Mean-squared-error loss on flat fp32 tensors (nn.MSELoss / F.mse_loss of the training scripts:
// experiments/train_baseline.py:64,86, train_continual.py:31,55, ewc.py:125).  HBM-bound: two reads per element forward,
// two reads + one write backward.
#include "common.h"

namespace nvq {

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                          float* __restrict__ part) {
    __shared__ float scratch[4];
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = ld4(a + 4 * i), y = ld4(b + 4 * i);
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        s += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float d = a[4 * n4 + threadIdx.x] - b[4 * n4 + threadIdx.x];
        s += d * d;
    }
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void mse_final_kernel(const float* __restrict__ part, int nblk, double inv_n, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += (double)part[k];
        *out = (float)(s * inv_n);
    }
}

// dA = go * 2 (a - b) / n
__global__ __launch_bounds__(256) void mse_backward_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                           const float* __restrict__ go, float two_over_n,
                                                           float* __restrict__ da) {
    const float sc = two_over_n * (go ? *go : 1.f);
    const long n4 = n >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = ld4(a + 4 * i), y = ld4(b + 4 * i);
        st4(da + 4 * i, make_float4(sc * (x.x - y.x), sc * (x.y - y.y), sc * (x.z - y.z), sc * (x.w - y.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = 4 * n4 + threadIdx.x;
        da[i] = sc * (a[i] - b[i]);
    }
}

static int mse_blocks(long n) {
    int nb = ceil_div(n, 256L * 16);
    if (nb > 2048) nb = 2048;
    return nb < 1 ? 1 : nb;
}

}  // namespace nvq

using namespace nvq;

extern "C" {

int nvq_mse_forward(const float* a, const float* b, long n, float* out, float* workspace, size_t workspace_bytes,
                    void* stream) {
    NVQ_REQUIRE(n > 0 && aligned16(a) && aligned16(b), "mse_forward: n > 0 and 16-byte aligned tensors");
    const int nb = mse_blocks(n);
    if ((size_t)nb * sizeof(float) > workspace_bytes) { set_error("mse_forward: workspace"); return NVQ_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(nb), dim3(256), 0, s, a, b, n, workspace);
    int rc = check_launch("mse_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, s, workspace, nb, 1.0 / (double)n, out);
    return check_launch("mse_final");
}

int nvq_mse_backward(const float* a, const float* b, long n, const float* grad_out_dev, float* da, void* stream) {
    NVQ_REQUIRE(n > 0 && aligned16(a) && aligned16(b) && aligned16(da), "mse_backward: n > 0 and 16-byte aligned tensors");
    hipLaunchKernelGGL(mse_backward_kernel, dim3(mse_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, grad_out_dev,
                       (float)(2.0 / (double)n), da);
    return check_launch("mse_backward");
}

}  // extern "C"
