// Elastic Weight Consolidation on flat parameter buckets: one launch instead of the
// reference's ~5 tiny ops per parameter tensor (ewc.py:225-232, 139-141).
#include "common.h"

namespace nvq {

__global__ __launch_bounds__(256) void ewc_penalty_kernel(const float* __restrict__ theta,
                                                          const float* __restrict__ star,
                                                          const float* __restrict__ fisher, long n,
                                                          float* __restrict__ part) {
    __shared__ float scratch[4];
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = theta[i] - star[i];
        s += fisher[i] * d * d;
    }
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void ewc_penalty_final_kernel(const float* __restrict__ part, int nblk, float half_lambda,
                                         float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += (double)part[b];
        *out = (float)((double)half_lambda * s);
    }
}

__global__ __launch_bounds__(256) void ewc_grad_kernel(const float* __restrict__ theta, const float* __restrict__ star,
                                                       const float* __restrict__ fisher, long n, float lambda,
                                                       const float* __restrict__ scale_dev, float* __restrict__ grad,
                                                       int accumulate) {
    const float sc = lambda * (scale_dev ? *scale_dev : 1.f);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float g = sc * fisher[i] * (theta[i] - star[i]);
        grad[i] = accumulate ? grad[i] + g : g;
    }
}

__global__ __launch_bounds__(256) void fisher_acc_kernel(const float* __restrict__ grad, long n,
                                                         float* __restrict__ fisher) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float g = grad[i];
        fisher[i] += g * g;
    }
}

// SynapticIntelligence.update_importance (ewc.py:343-354) over a flat bucket: W += -(grad * (theta - p_old)); p_old = theta.
// Separate multiply and add (no fma contraction), the reference's two torch ops.
__global__ __launch_bounds__(256) void si_update_kernel(const float* __restrict__ theta, const float* __restrict__ grad,
                                                        long n, float* __restrict__ p_old, float* __restrict__ W) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float t = theta[i];
        W[i] = __fadd_rn(W[i], -__fmul_rn(grad[i], __fsub_rn(t, p_old[i])));
        p_old[i] = t;
    }
}

// SynapticIntelligence.register_task (ewc.py:356-368): omega += W / ((theta - p_old)^2 + damping); W = 0; p_old = theta.
__global__ __launch_bounds__(256) void si_consolidate_kernel(const float* __restrict__ theta, long n, float damping,
                                                             float* __restrict__ p_old, float* __restrict__ W,
                                                             float* __restrict__ omega) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float t = theta[i];
        const float d = __fsub_rn(t, p_old[i]);
        omega[i] = __fadd_rn(omega[i], __fdiv_rn(W[i], __fadd_rn(__fmul_rn(d, d), damping)));
        W[i] = 0.f;
        p_old[i] = t;
    }
}

static int flat_blocks(long n) {
    int nb = ceil_div(n, 256L * 4);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return nb;
}

}  // namespace nvq

using namespace nvq;

extern "C" {

int nvq_ewc_penalty(const float* theta, const float* theta_star, const float* fisher, long n, float lambda,
                    float* out, float* workspace, size_t workspace_bytes, void* stream) {
    const int nb = flat_blocks(n);
    if ((size_t)nb * sizeof(float) > workspace_bytes) { set_error("ewc_penalty: workspace"); return NVQ_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ewc_penalty_kernel, dim3(nb), dim3(256), 0, s, theta, theta_star, fisher, n, workspace);
    int rc = check_launch("ewc_penalty");
    if (rc) return rc;
    hipLaunchKernelGGL(ewc_penalty_final_kernel, dim3(1), dim3(64), 0, s, workspace, nb, 0.5f * lambda, out);
    return check_launch("ewc_penalty_final");
}

int nvq_ewc_penalty_grad(const float* theta, const float* theta_star, const float* fisher, long n, float lambda,
                         const float* scale_dev, float* grad, int accumulate, void* stream) {
    hipLaunchKernelGGL(ewc_grad_kernel, dim3(flat_blocks(n)), dim3(256), 0, (hipStream_t)stream, theta, theta_star,
                       fisher, n, lambda, scale_dev, grad, accumulate);
    return check_launch("ewc_penalty_grad");
}

int nvq_si_update(const float* theta, const float* grad, long n, float* p_old, float* W, void* stream) {
    hipLaunchKernelGGL(si_update_kernel, dim3(flat_blocks(n)), dim3(256), 0, (hipStream_t)stream, theta, grad, n, p_old, W);
    return check_launch("si_update");
}

int nvq_si_consolidate(const float* theta, long n, float damping, float* p_old, float* W, float* omega, void* stream) {
    hipLaunchKernelGGL(si_consolidate_kernel, dim3(flat_blocks(n)), dim3(256), 0, (hipStream_t)stream, theta, n, damping,
                       p_old, W, omega);
    return check_launch("si_consolidate");
}

int nvq_fisher_accumulate(const float* grad, long n, float* fisher, void* stream) {
    hipLaunchKernelGGL(fisher_acc_kernel, dim3(flat_blocks(n)), dim3(256), 0, (hipStream_t)stream, grad, n, fisher);
    return check_launch("fisher_accumulate");
}

}  // extern "C"
