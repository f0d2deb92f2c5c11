/*
 * nvq.h — C ABI of libnvq.so, the MI355X (gfx950) kernels behind
 * nerve_cl.models.SuperResolutionNet forward/backward and nerve_cl.continual.EWC.
 *
 * The reference (manikya7022/Continual-Learning-for-Dynamic-Video-Quality-Enhancement)
 * has no native / FFI boundary: its hot path is a chain of torch.nn calls.  Each entry
 * point below therefore names the reference torch call(s) it replaces (file:line,
 * relative to the reference checkout).  The Python host that binds these symbols is
 * nerve_cl/_nvq.py (ctypes); INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator);
 *    the library never allocates, frees, synchronises or changes device: calls are
 *    stream-ordered on `stream` (a hipStream_t passed as void*) and graph-capturable.
 *  - activations are fp32 "NHWC with leading dimension": image n, pixel (y,x), channel c
 *    lives at base[((n*H + y)*W + x)*ld + coff + c].  A channel slice of a wider buffer
 *    (dense-block concat, frame-major aligned stack) is addressed by (base, ld, coff);
 *    ld and coff are in elements.  Tensors at the module boundary (frames in, SR frame
 *    out, parameters and their gradients) are fp32 NCHW / PyTorch layout.
 *  - return value: 0 on success, a negative NVQ_E* code otherwise; nvq_last_error()
 *    returns a thread-local description.  Nothing throws.
 */
#ifndef NVQ_H
#define NVQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVQ_OK 0
#define NVQ_EINVAL (-1)   /* bad argument / unsupported shape */
#define NVQ_ELAUNCH (-2)  /* hipLaunch / hipGetLastError failure */
#define NVQ_EWORKSPACE (-3)

#define NVQ_MATH_F32 0    /* v_mfma_f32_16x16x4_f32: exact fp32 products and sums */
#define NVQ_MATH_BF16 1   /* operands rounded to bf16 in LDS, fp32 accumulate (v_mfma_f32_16x16x32_bf16) */

#define NVQ_MAX_T 8

int nvq_version(void);
const char* nvq_last_error(void);

/* ------------------------------------------------------------------ convolution
 * Dense k x k (k = 1 or 3), stride 1, zero "same" padding, cross-correlation
 * convention: replaces nn.Conv2d forward and, with a transposed pack, its input
 * gradient.  Reference call sites: flow_net super_resolution.py:74-82, attention
 * :168-175, ResidualDenseBlock :236-253, gff :308-311, upsampler conv
 * efficient_layers.py:94-100, pointwise :49-56. */

/* Packed-weight size in floats for a conv with `cout` outputs reading `cin_store`
 * stored input channels (cin_store % 4 == 0), for the given NVQ_MATH_* mode (the pack is
 * mode-specific: fp32 values in 16-channel chunks, or bf16 values in 32-channel chunks). */
size_t nvq_conv_pack_floats(int cout, int cin_store, int ksize, int math);

/* w: PyTorch layout [cout_w][cin_w][k][k].
 * transpose == 0: forward pack; output channels = cout_w, input channels = cin_w
 *                 (zero-padded up to cin_store).
 * transpose == 1: input-gradient pack; output channels = cin_w (only the first
 *                 `cout_keep` are kept), input channels = cout_w (padded to cin_store),
 *                 taps flipped. */
int nvq_conv_pack(const float* w, int cout_w, int cin_w, int ksize, int transpose,
                  int cin_store, int cout_keep, int math, float* wpack, void* stream);

/* Several nvq_conv_pack calls in one launch: a training step packs ~120 weights, and at small frame sizes every launch costs
 * ~18 us of device-side latency whatever its size.  Fields as the arguments of nvq_conv_pack; `jobs` is a host array. */
typedef struct nvq_pack_job {
    const float* w; int cout_w; int cin_w; int ksize; int transpose; int cin_store; int cout_keep; float* wpack;
} nvq_pack_job;
int nvq_conv_pack_batch(const nvq_pack_job* jobs, int njobs, int math, void* stream);

typedef struct nvq_conv_desc {
    const float* in;  int in_ld;  int in_coff;  int cin;       /* cin % 4 == 0 stored channels */
    const float* wpack;                                          /* from nvq_conv_pack */
    const float* bias;                                           /* [cout] or NULL */
    float* out;       int out_ld; int out_coff; int cout;      /* real output channels */
    int cout_store;                                              /* >= cout: channels [cout,cout_store) are written as the epilogue of a zero accumulator */
    float* out2;      int out2_ld; int out2_coff;               /* optional: value before residual/accumulate/mask */
    const float* res; int res_ld; int res_coff; int res_cmax;   /* residual added to out channels < res_cmax */
    const float* mask; int mask_ld; int mask_coff; int mask_c0; int mask_c1; /* out channels in [c0,c1) are zeroed where mask <= 0 */
    int n, h, w;
    int ksize;        /* 1 or 3 */
    int relu;         /* max(.,0) after bias */
    float alpha;      /* scale after relu */
    int accumulate;   /* out += result */
    int math;         /* NVQ_MATH_* */
    /* Storage type of each activation tensor: 0 = fp32, 1 = bf16 (ld / coff then count bf16 elements and the
     * pointer, although typed float*, addresses bf16 data).  bf16 tensors need NVQ_MATH_BF16, a bf16 input
     * needs cin, in_ld, in_coff % 8 == 0, and every slice must be 8-byte addressable. */
    int in_bf16, out_bf16, out2_bf16, res_bf16, mask_bf16;
    /* One-bit ReLU masks (NVQ_MATH_BF16, vector epilogue; cout <= 32, or a 3x3 conv of a bf16 input with more output channels):
     * word c / 32 of the bits_words (0 = 1) words of pixel (n*h + y)*w + x holds bit c % 32 = "output channel c of this pixel
     * is > 0".  bits_mode 1: written from the value stored to `out` (a conv + ReLU forward); bits_mode 2: read as the mask of
     * channels [0, cout_store) instead of a `mask` tensor (the input-gradient conv of the layer behind it: 4 bytes per pixel
     * and 32 channels instead of 64); 0: unused. */
    unsigned* bits;
    int bits_mode;
    /* 3x3 only, a hint: the packed weights of input channels [0, center_cin) are zero outside the centre tap (the
     * 0.2*lff^T part of the mirror-form dense-block gradient convs, nvq_rdb_backward_weights), so the kernel may skip the
     * other eight taps of those channels.  Multiple of 32, <= cin; 0 = no such channels.  Results do not depend on it. */
    int center_cin;
    /* Slice-planar input (NVQ_MATH_BF16, bf16 input, cin % 32 == 0, in_coff == 0): in_plane = elements of one 32-channel
     * plane = n*h*w*32.  The first in_ld channels (in_ld in {32, 64, 128}) are an ordinary [n][h][w][in_ld] tensor at `in`;
     * every further 32-channel chunk kc (>= in_ld/32) is a compact [n][h][w][32] tensor at in + kc*in_plane - the layout in
     * which a dense block's buffer is [x | y_0 | y_1 | ...] as separate tensors in one allocation, so that every 64-byte
     * pixel row a layer writes shares its 128-byte line with the next pixel, not with another layer.  0 = the usual
     * interleaved buffer.  Outputs, residuals and masks are ordinary (ld, coff) slices of those tensors in either case. */
    unsigned in_plane;
    /* Kernel-variant hint, results do not depend on it (beyond the fp32 summation order inside a 32-channel K chunk): 0 =
     * automatic; 8 = the 3x3 NVQ_MATH_BF16 kernels use their 8x32-pixel, four-wave form (the automatic choice for bf16 input is
     * an eight-wave form: 16x32 tiles for cout <= 32 - for images of fewer than 32 such tiles the 8x32 form -, two
     * 32-channel halves per workgroup for cout >= 64); for cout <= 32 also
     * 16 = the 16x32-tile kernel on v_mfma_f32_16x16x32_bf16, 162 / 164 = the same tile on v_mfma_f32_32x32x16_bf16 with two /
     * four tile rows per wave (automatic: 162 up to 128 input channels, 16 above); for cout >= 64 (a multiple of 64 stored channels,
     * bf16 output) 264 / 265 = v_mfma_f32_32x32x16_bf16 forms (a wave of 2 rows x 64 channels / eight channel-split waves;
     * measured slower or equal, never automatic).  Lets a caller A/B the forms without any
     * library state. */
    int tile_rows;
    /* words per pixel of `bits` (0 or 1: one word, cout <= 32) */
    int bits_words;
} nvq_conv_desc;
/* epilogue: v = acc + bias; if relu v = max(v,0); v *= alpha; out2 = v;
 *           if c < res_cmax v += res; if accumulate v += out; if mask<=0 on [c0,c1) v = 0; out = v */
int nvq_conv_forward(const nvq_conv_desc* d, void* stream);
/* Tail of ResidualDenseBlock.forward (super_resolution.py:245-253) in one launch: d3 = the last dense layer
 * (3x3, cin = F+128 -> 32 channels written in place at [cin, cin+32) of the bf16 concat buffer, bias + ReLU, optional
 * bits_mode 1), dl = the local feature fusion (1x1 over channels [0, cin+32) of the same buffer -> 64 channels, with
 * dl's alpha / res / out2 / output).  Same results as nvq_conv_forward(d3) followed by nvq_conv_forward(dl); the concat
 * buffer is read once instead of twice.  NVQ_MATH_BF16 only. */
int nvq_rdb_tail_forward(const nvq_conv_desc* d3, const nvq_conv_desc* dl, void* stream);
/* (The library keeps no mutable state: the diagnostic switches of tools/ - nvq_debug_* - exist only in the separate
 * libnvq_debug.so that `NVQ_DEBUG_TOOLS=1 build.sh` makes; tools/nvq_debug.h declares them.) */
size_t nvq_sizeof_conv_desc(void);

/* Combined weights for the backward of one ResidualDenseBlock (super_resolution.py:245-253) in "mirror"
 * form.  With the block's gradient buffer laid out [gout(F) | dy_4 | dy_3 | dy_2 | dy_1 | dy_0], the gradient
 * of growth slice y_i is ONE 3x3 convolution over the channel prefix [0, F+32(4-i)) (the 0.2*lff^T term on the
 * centre tap, the transposed/flipped dense-layer weights elsewhere), and the gradient w.r.t. the block input
 * is one more over all F+160 channels - no read-modify-write accumulation.
 * lff: [F][F+160][1][1]; w_i: [32][F+32i][3][3].  out holds, in PyTorch layout and in this order,
 * Wb_4 [32][F], Wb_3 [32][F+32], ..., Wb_0 [32][F+128], Wb_x [F][F+160]  (each [..][3][3]). */
size_t nvq_rdb_backward_weights_floats(int F);
int nvq_rdb_backward_weights(const float* lff, const float* w0, const float* w1, const float* w2,
                             const float* w3, const float* w4, int F, float* out, void* stream);

/* Weight (+ bias) gradient of the same convolution: replaces the parameter half of
 * aten::convolution_backward.  dw is PyTorch layout [cout][cin_w][k][k]; only the
 * first cin_w of the `cin` stored input channels receive a gradient.
 * dw = alpha * sum_pixels x (*) dy  (+ dw if accumulate), same for dbias. */
typedef struct nvq_wgrad_desc {
    const float* x;  int x_ld;  int x_coff;  int cin;  int cin_w;
    const float* dy; int dy_ld; int dy_coff; int cout;
    float* dw; float* dbias;            /* dbias may be NULL */
    float* workspace; size_t workspace_bytes;
    int n, h, w, ksize;
    float alpha; int accumulate; int math;
    int x_bf16, dy_bf16;                /* storage type of x / dy (see nvq_conv_desc); need NVQ_MATH_BF16 */
    unsigned x_plane;                   /* slice-planar x (see nvq_conv_desc::in_plane; bf16 x, x_coff == 0); 0 = interleaved */
    /* Kernel-variant hint, results do not depend on it (up to the summation order): 0 = automatic; 1 = always the
     * (pixel split, ci chunk, co chunk) kernels; 2 = always the all-input-channel kernels (an error for a shape they do not
     * take).  The all-input-channel kernels (bf16 x and dy; 3x3: cout = 32, slice-planar x, cin in {96, .., 192}; 1x1: cin in
     * {64, .., 256}, cout <= 64) read x without halo and dy once per launch, as persistent workgroups; automatic from four
     * 4x32-pixel tiles per workgroup on (smaller launches: the split kernels).  Lets a caller A/B the two forms. */
    int variant;
} nvq_wgrad_desc;
size_t nvq_wgrad_workspace_bytes(void);   /* upper bound valid for every shape */
int nvq_conv_wgrad(const nvq_wgrad_desc* d, void* stream);
/* The same in two halves, for steps that are bound by their launch count (the 64x64 continual-learning step): the weight-gradient
 * kernel alone leaves its partial sums in d->workspace and describes the reduce that finishes it in *job; nvq_wgrad_reduce_batch
 * runs up to 16 such reduces per launch (any n).  Every job needs a workspace of its own until its batch has run.  Results equal
 * nvq_conv_wgrad's bit for bit (same sums, same order). */
typedef struct {
    const float* part; const float* bias_part; float* dw; float* dbias;
    int nsplit, nci, nco, taps, cout, cin_w; float alpha; int accumulate;
} nvq_wgrad_reduce_job;
int nvq_conv_wgrad_partial(const nvq_wgrad_desc* d, nvq_wgrad_reduce_job* job, void* stream);
int nvq_wgrad_reduce_batch(const nvq_wgrad_reduce_job* jobs, int n, void* stream);
size_t nvq_sizeof_wgrad_desc(void);
size_t nvq_sizeof_wgrad_reduce_job(void);

/* Backward of  pointwise 1x1 conv (no bias) -> BatchNorm2d -> ReLU  of a DepthwiseSeparableConv (efficient_layers.py:49-66) in
 * the bf16 mode, 64 channels in and out: what nvq_bn_relu_backward + nvq_conv_forward (transposed pack) + nvq_conv_wgrad do
 * in three passes, after the BatchNorm sums in ONE pass over the tensors (dp = the BatchNorm-input gradient is formed in LDS
 * and never stored).  dy [N,H,W,dy_ld]: gradient w.r.t. relu(bn(p)), fp32 or bf16; p: the conv output = BatchNorm input, d: the
 * conv input (both bf16); groups of `group_images` images have their own statistics mean / invstd [G][64];
 * weight [64 co][64 ci] fp32.  Outputs: dd (bf16) = gradient w.r.t. d; dgamma, dbeta [64]; dweight [64][64] (all overwritten).
 * sums_in != NULL: the BatchNorm sums [G][2][64] already computed (nvq_dwconv_backward's bn_sums for the same dy and p):
 * the reduce pass is skipped and dgamma / dbeta are left alone (that call wrote them).
 * workspace: nvq_wgrad_workspace_bytes() is enough. */
int nvq_pw_bn_backward(const float* dy, int dy_ld, int dy_bf16, const float* p, int p_ld, const float* d, int d_ld,
                       int N, int group_images, int H, int W, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, int training, const float* weight, float* dd, int dd_ld,
                       float* dgamma, float* dbeta, float* dweight, const float* sums_in, float* workspace,
                       size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ feature extractor
 * FeatureExtractor.head, super_resolution.py:40-43: relu(conv3x3(frame; W[F,Cin,3,3], b)).
 * frames: fp32 NCHW clip (B,T,Cin,H,W) contiguous.  Output image index = slot*B + b
 * holds frame t = t_of_slot[slot]. */
int nvq_head_forward(const float* frames, int B, int T, int Cin, int H, int W,
                     const int* t_of_slot_host, int nslots,
                     const float* weight, const float* bias, int F,
                     float* out, int out_ld, int out_bf16, float* img8, int math, void* stream);
/* img8 (optional, bf16 [nslots*B][H][W][8]): the frames themselves in slot order, channels 0..Cin-1 (rest zero) - the
 * x operand with which nvq_conv_wgrad computes the head's weight gradient on the matrix cores in bf16 mode.
 * math: NVQ_MATH_F32 = exact fp32 FMAs; NVQ_MATH_BF16 = weights rounded to bf16, the frame taken as a hi + lo pair of bf16
 * values (~16 bits), fp32 accumulation on the matrix cores (Cin = 3, F in {16, 32, 64}; other shapes use the fp32 kernel
 * in either mode). */
/* dweight[F][Cin][3][3], dbias[F] (+= if accumulate) from (dout + dout2) masked by (act > 0); dout2 may be NULL
 * (it is the skip path of `features = body(h) + h`, summed here instead of in a separate pass). */
int nvq_head_wgrad(const float* frames, int B, int T, int Cin, int H, int W,
                   const int* t_of_slot_host, int nslots,
                   const float* dout, int dout_ld, const float* dout2, int dout2_ld,
                   const float* act, int act_ld, int F,
                   float* dweight, float* dbias, float* workspace, size_t workspace_bytes,
                   int accumulate, int act_bf16, void* stream);

/* Depthwise 3x3 (groups = C, no bias), efficient_layers.py:38-46.  weight [C][1][3][3].
 * flip == 1 gives the input gradient.  The *_bf16 flags of this section give the storage type of the
 * corresponding activation tensor (0 = fp32, 1 = bf16; arithmetic and statistics stay fp32). */
/* bn != NULL (bf16 input, C % 64 == 0): the kernel's input is relu(batchnorm(in)) evaluated on the fly with the given
 * per-(group, channel) statistics - the BatchNorm + ReLU of DepthwiseSeparableConv k feeding the depthwise conv of k+1
 * (efficient_layers.py:62-67) without a pass of its own.  mean/invstd: [N/group_images][C]; gamma/beta: [C]. */
typedef struct nvq_bn_input {
    const float* mean; const float* invstd; const float* gamma; const float* beta;
    int group_images;
} nvq_bn_input;
/* epi != NULL (same kernels): out = (conv + add) where mask > 0, else 0.  add: fp32 or bf16 (add_bf16) [.., add_ld] or NULL;
 * mask: fp32 or bf16 [.., mask_ld] or NULL.  Used for the last depthwise input gradient of the feature extractor, which so leaves the
 * kernel as the ReLU-masked gradient of the head conv (skip path added). */
typedef struct nvq_dw_epilogue {
    const float* add; int add_ld; const float* mask; int mask_ld; int mask_bf16; int add_bf16;
} nvq_dw_epilogue;
int nvq_dwconv_forward(const float* in, int in_ld, const float* weight, int C,
                       float* out, int out_ld, int N, int H, int W, int flip,
                       int in_bf16, int out_bf16, const nvq_bn_input* bn, const nvq_dw_epilogue* epi,
                       void* stream);
int nvq_dwconv_wgrad(const float* x, int x_ld, const float* dy, int dy_ld, int C,
                     int N, int H, int W, float* dweight, float* workspace,
                     size_t workspace_bytes, int accumulate, int x_bf16, int dy_bf16,
                     const nvq_bn_input* bn, void* stream);
/* Backward of the depthwise 3x3 conv of a DepthwiseSeparableConv (efficient_layers.py:49-66) in the bf16 mode, 64 channels:
 * what nvq_dwconv_wgrad + nvq_dwconv_forward(flip = 1) do in two launches, from one staged tile (x, dy -> dx; dy read once).
 * x [N,H,W,x_ld] bf16: the conv's input; bn != NULL: the input was relu(bn(x)) (evaluated while x is staged, as in
 * nvq_dwconv_wgrad); dy: gradient w.r.t. the conv output, bf16; weight [64][3][3] fp32.  Outputs: dx (bf16, overwritten) =
 * gradient w.r.t. the conv input, through the optional epilogue dx = (dx + add) where mask > 0 (add fp32, mask bf16);
 * dweight [64][3][3] (overwritten).  bn_sums != NULL (needs bn, no epi): the kernel holds both operands of the BACKWARD of that
 * BatchNorm + ReLU - its input x and the gradient dx of its output - and also returns the per-group sums bn_sums [G][2][64]
 * = {sum g, sum g xhat} (g = dx where relu(bn(x)) > 0) and bn_dgamma / bn_dbeta [64] (overwritten): what the first half of
 * nvq_bn_relu_backward / nvq_pw_bn_backward computes in a pass of its own; hand bn_sums to nvq_pw_bn_backward(sums_in).
 * workspace >= 512*(576+128) floats. */
int nvq_dwconv_backward(const float* x, int x_ld, const nvq_bn_input* bn, const float* dy, int dy_ld, const float* weight,
                        float* dx, int dx_ld, const nvq_dw_epilogue* epi, int N, int H, int W, float* dweight,
                        float* bn_sums, float* bn_dgamma, float* bn_dbeta, float* workspace, size_t workspace_bytes,
                        void* stream);
/* Forward of  depthwise 3x3 -> pointwise 1x1 (no bias) -> BatchNorm2d statistics  of a DepthwiseSeparableConv
 * (efficient_layers.py:49-66) in the bf16 mode, 64 channels: what nvq_dwconv_forward + nvq_conv_forward (1x1) + nvq_bn_stats
 * do in three launches (five tensor passes), in one pass x -> d, p.  in [N,H,W,in_ld] bf16; bn != NULL: x := relu(bn(x)) is
 * applied while the halo is staged (the previous layer's BatchNorm + ReLU, as in nvq_dwconv_forward; its group_images must be
 * `group_images`); dw_weight [64][3][3], pw_weight [64 co][64 ci] fp32.  Outputs (bf16, overwritten): d = the depthwise
 * result (the backward's operand), p = the pointwise result = BatchNorm input.  stats != 0: the batch statistics of p (of
 * the stored bf16 values) per group of `group_images` images go to mean / invstd [G][64] and are folded into running_mean /
 * running_var (either may be NULL) in the order order_host[0..G-1] (host array; NULL: 0..G-1), exactly as nvq_bn_stats does; workspace >= 512*128 floats.
 * stats == 0: p and d only (eval mode: the caller takes mean / invstd from nvq_bn_eval_stats). */
int nvq_dwpw_forward(const float* in, int in_ld, const nvq_bn_input* bn, const float* dw_weight, const float* pw_weight,
                     float* d, int d_ld, float* p, int p_ld, int N, int group_images, int H, int W, int stats, float eps,
                     float momentum, const int* order_host, float* mean, float* invstd, float* running_mean, float* running_var,
                     float* workspace, size_t workspace_bytes, void* stream);

/* BatchNorm2d, efficient_layers.py:59,65.  The N images form G = N/group_images groups
 * (one per feature-extractor call); statistics are per (group, channel) over
 * group_images*H*W pixels.
 * nvq_bn_stats: training statistics.  mean/invstd: [G][C] (saved for backward).
 *   running_* are updated once per group in the order order_host[0..G-1] (the reference
 *   calls the extractor for t = 0..T-1, super_resolution.py:347-349): momentum 0.1,
 *   unbiased variance.  running_* may be NULL. */
int nvq_bn_stats(const float* x, int x_ld, int C, int N, int group_images, int H, int W,
                 float eps, float momentum, const int* order_host,
                 float* mean, float* invstd, float* running_mean, float* running_var,
                 float* workspace, size_t workspace_bytes, int x_bf16, void* stream);
/* Evaluation: fill mean/invstd [G][C] from the running statistics. */
int nvq_bn_eval_stats(const float* running_mean, const float* running_var, int C, int G,
                      float eps, float* mean, float* invstd, void* stream);
/* y = relu(gamma*(x-mean)*invstd + beta) (+ res).  Images [0, split_images) go to outA,
 * the rest to outB (image index rebased): lets the centre frame land directly in its
 * slot of the frame-major aligned stack (super_resolution.py:352,357). */
int nvq_bn_apply_relu(const float* x, int x_ld, int C, int N, int group_images, int H, int W,
                      const float* mean, const float* invstd, const float* gamma,
                      const float* beta, const float* res, int res_ld,
                      float* outA, int outA_ld, int outA_coff, int split_images,
                      float* outB, int outB_ld, int outB_coff, int x_bf16, int out_bf16, int res_bf16,
                      void* stream);
/* Backward of y = relu(bn(x)).  dy is the gradient w.r.t. y; the ReLU mask is recomputed
 * from x and the statistics (gamma*(x-mean)*invstd + beta > 0), so y itself is not needed.
 * training != 0: batch-statistics backward; else running-statistics backward.
 * dgamma/dbeta [C]: summed over all groups, (+)= if accumulate. */
int nvq_bn_relu_backward(const float* dy, int dy_ld, const float* x, int x_ld, int C, int N,
                         int group_images, int H, int W, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, int training,
                         float* dx, int dx_ld, float* dgamma, float* dbeta,
                         float* workspace, size_t workspace_bytes, int accumulate,
                         int dy_bf16, int x_bf16, int dx_bf16, void* stream);

/* ------------------------------------------------------------------ motion
 * LiteFlowNetCorrelation(d=4).forward, efficient_layers.py:313-343.
 * out[n,p, i*9+j] = (1/C) sum_c x1[n,p,c] * x2[n, p + (i-4, j-4), c]; channels 81..out_ld-1
 * of each pixel are written as zero.  x2 image index = n % x2_images (centre-frame
 * features are shared by the T-1 reference frames).
 * math == NVQ_MATH_BF16 (and C in {32, 64}): the products run on the bf16 matrix cores (x1 / x2 rounded to bf16,
 * fp32 accumulation), out may be stored as bf16 (out_bf16; out_ld then a multiple of 8) and x1 / x2 may be bf16-stored
 * feature tensors (in_bf16, both; their ld then count bf16 elements and are multiples of 8).  Otherwise exact fp32. */
int nvq_correlation_forward(const float* x1, int x1_ld, const float* x2, int x2_ld,
                            int x2_images, int C, int N, int H, int W,
                            float* out, int out_ld, int math, int out_bf16, int in_bf16, void* stream);
/* which == 1: dx[n,p,c] (+)= (1/C) sum_d dcorr[n,p,d] * other[n % other_images, p+off(d), c]
 *             (gradient w.r.t. x1; other = x2)
 * which == 2: dx[n,q,c] (+)= (1/C) sum_d dcorr[n,q-off(d),d] * other[n, q-off(d), c]
 *             (gradient w.r.t. x2 contributed by image n; other = x1, other_images = N).
 *             groups > 1: dcorr and other hold groups * N images (frame-major: the T - 1 reference frames of every clip);
 *             dx[n] collects from images n, n + N, ... in that order in ONE pass (one read-modify-write of the shared
 *             centre-frame gradient instead of one per frame)
 * math / dcorr_bf16 / other_bf16 as above; the bf16 path reads dcorr up to channel 96 (dcorr_ld >= 96).  dx is fp32.
 * dx_bf16_out != NULL (matrix-core path): this is the LAST pass over an accumulated gradient - the result (dx + the new
 * term when accumulate) is written as bf16 to dx_bf16_out [N,H,W,dx_bf16_ld] channels [0,C) instead of back to dx, which
 * is only read: the readers of the finished gradient then move half the bytes.  addends (optional, with dx_bf16_out): up to
 * two more terms of that gradient, bf16 tensors [N,H,W,ld] read at channels [coff, coff + C), added in the same pass (with
 * accumulate == 0, dx is not touched at all - the whole sum is formed here, no fp32 accumulator, no separate add kernel). */
typedef struct nvq_corr_addends {
    const void* a; int a_ld; int a_coff;
    const void* b; int b_ld; int b_coff;
} nvq_corr_addends;
int nvq_correlation_backward(int which, const float* dcorr, int dcorr_ld,
                             const float* other, int other_ld, int other_images, int C, int N,
                             int H, int W, float* dx, int dx_ld, int dx_coff, int accumulate,
                             int math, int dcorr_bf16, int other_bf16, int groups, float* dx_bf16_out, int dx_bf16_ld,
                             const nvq_corr_addends* addends, void* stream);

/* warp_features, super_resolution.py:104-143 (F.grid_sample bilinear, zeros,
 * align_corners=True at pixel coordinates (x+flow_x, y+flow_y)). flow: [N,H,W,flow_ld>=2] fp32.
 * feat_bf16 / out_bf16: storage type of the two feature tensors (fp32 interpolation arithmetic either way). */
int nvq_warp_forward(const float* feat, int feat_ld, const float* flow, int flow_ld,
                     int C, int N, int H, int W, float* out, int out_ld, int out_coff,
                     int feat_bf16, int out_bf16, void* stream);
/* overwrite == 0: dfeat must be pre-initialised (the gradient is ADDED to it); overwrite != 0 (gather form only, 16 more
 * bytes of records): dfeat [N,H,W,0..C) is WRITTEN - no zero fill before and no read-modify-write in the call.  dflow
 * [N,H,W,dflow_ld] gets channels 0,1 written and 2..dflow_ld-1 zeroed.  records == NULL: scatter form, 4 float atomics per (pixel, channel).  records != NULL (20 bytes
 * per pixel of scratch, 16-byte aligned): gather form - a per-source pass writes the flow gradient and a {corner offset,
 * 4 weights} record, a per-destination pass collects the contributions from the 9x9 window around each pixel; no atomics
 * and a fixed summation order for every source whose flow is shorter than 4 pixels (longer ones are still scattered).
 * feat_bf16 / dout_bf16: `feat` (read for the flow gradient) / `dout` are stored as bf16 - gather form only; dfeat and
 * dflow are fp32 - except dfeat in the overwrite mode with dfeat_bf16 != 0: the gather pass then writes bf16 (the far sources
 * and the hit-list overflow add with a compare-and-swap loop per 32-bit word). */
int nvq_warp_backward(const float* dout, int dout_ld, int dout_coff, const float* feat,
                      int feat_ld, const float* flow, int flow_ld, int C, int N, int H, int W,
                      float* dfeat, int dfeat_ld, float* dflow, int dflow_ld,
                      float* records, size_t records_bytes, int feat_bf16, int dout_bf16, int overwrite, int dfeat_bf16,
                      void* stream);

/* ------------------------------------------------------------------ temporal aggregation
 * TemporalAggregator.forward softmax + weighted sum, super_resolution.py:174,203-204:
 * attn = softmax_t(logits[n,p,0..T-1]); weighted[n,p,c] = sum_t aligned[n,p,t*C+c]*attn_t.
 * Also emits per-block channel sums of `weighted` for the CBAM global average pool:
 * gap_partial [N][nblk][C] with nblk = nvq_tsum_blocks(H,W). */
int nvq_tsum_blocks(int H, int W);
int nvq_tsum_forward(const float* aligned, int aligned_ld, const float* logits, int logits_ld,
                     int T, int C, int N, int H, int W, float* attn, int attn_ld,
                     float* weighted, int weighted_ld, float* gap_partial, int aligned_bf16, int weighted_bf16,
                     void* stream);
/* aligned_bf16 (here and below): `aligned` is stored as bf16 (aligned_ld counts bf16 elements).  weighted_bf16 /
 * dweighted_bf16 / x_bf16 (here and in the CBAM calls below): the aggregated feature tensor `weighted` (= the CBAM's x), the
 * gradient that enters the CBAM backward (dout) and the one that leaves it (dx = dweighted) are stored as bf16 - the bf16
 * activation mode of the network; the pooled sums (gap_partial, sm, dca_partial) are formed from unrounded values either way. */
/* dw = dweighted + dgap_pix[n][c] (dgap_pix may be NULL);
 * daligned[n,p,t*C+c] = dw*attn_t ; dlogits = softmax backward of sum_c dw*aligned_t. */
int nvq_tsum_backward(const float* dweighted, int dweighted_ld, const float* dgap_pix,
                      const float* aligned, int aligned_ld, const float* attn, int attn_ld,
                      int T, int C, int N, int H, int W, float* daligned, int daligned_ld,
                      float* dlogits, int dlogits_ld, int aligned_bf16, int daligned_bf16, int dweighted_bf16,
                      void* stream);

/* CBAM, efficient_layers.py:154-228.
 * cbam_channel: gap = mean(weighted); hid = relu(W1 gap); ca = sigmoid(W2 hid).
 *   w1 [R][C], w2 [C][R]; outputs gap [N][C], hid [N][R], ca [N][C]. */
int nvq_cbam_channel(const float* gap_partial, int nblk, int C, int R, int N, int HW,
                     const float* w1, const float* w2, float* gap, float* hid, float* ca,
                     void* stream);
/* sm[n,p,0] = mean_c(x*ca), sm[n,p,1] = max_c(x*ca); amax = argmax channel. sm ld = 2. */
int nvq_cbam_pool(const float* x, int x_ld, const float* ca, int C, int N, int H, int W,
                  float* sm, int* amax, int x_bf16, void* stream);
/* sa = sigmoid(conv7x7(sm; w[1][2][7][7], pad 3)); out = x*ca*sa written at (out,ld,coff); out_bf16 != 0
 * stores bf16 (the destination is the first dense block's concat buffer). */
int nvq_cbam_spatial_apply(const float* x, int x_ld, const float* ca, const float* sm,
                           const float* w7, int C, int N, int H, int W, float* sa,
                           float* out, int out_ld, int out_coff, int out_bf16, int x_bf16, void* stream);
/* Backward through out = x*ca*sa:
 * step1: dpre[n,p] = (sum_c dout*x*ca) * sa*(1-sa)                                   */
int nvq_cbam_bwd_spatial_pre(const float* dout, int dout_ld, int dout_coff, const float* x,
                             int x_ld, const float* ca, const float* sa, int C, int N,
                             int H, int W, float* dpre, int x_bf16, void* stream);
/* step2: dsm = conv7x7^T(dpre) [N,H,W,2]; dw7[1][2][7][7] (+)= sum sm (*) dpre */
int nvq_cbam_bwd_spatial_conv(const float* dpre, const float* sm, const float* w7, int N,
                              int H, int W, float* dsm, float* dw7, float* workspace,
                              size_t workspace_bytes, int accumulate, void* stream);
/* step3: dxc = dout*sa + dsm0/C + [c==amax] dsm1 ; dx = dxc*ca ;
 *        dca_partial[n][blk][c] = block sums of dxc*x   (nblk = nvq_tsum_blocks) */
int nvq_cbam_bwd_scale(const float* dout, int dout_ld, int dout_coff, const float* x, int x_ld,
                       const float* ca, const float* sa, const float* dsm, const int* amax,
                       int C, int N, int H, int W, float* dx, int dx_ld, float* dca_partial,
                       int x_bf16, void* stream);
/* step4: through sigmoid / FC / relu / FC / mean: dw1, dw2 (+)=; dgap_pix[n][c] = dgap/HW */
int nvq_cbam_bwd_channel(const float* dca_partial, int nblk, int C, int R, int N, int HW,
                         const float* w1, const float* w2, const float* gap, const float* hid,
                         const float* ca, float* dw1, float* dw2, float* dgap_pix,
                         int accumulate, void* stream);

/* The whole upsampler tail in one launch (NVQ_MATH_BF16, bf16 input): PixelShuffleUpsampler.conv (3x3, cin -> Cimg*s*s, bias;
 * efficient_layers.py:94-100) whose epilogue puts the tile's conv outputs through LDS, reads them back as s x s pixel blocks
 * (PixelShuffle, :101-106), adds the bicubic skip of frames[:, t_center] and clamps (super_resolution.py:378-382): the
 * conv output never exists as a tensor.  d: in / wpack / bias / cin / cout = Cimg*s*s / n, h, w (output fields unused);
 * out, pass as nvq_shuffle_bicubic_clamp below, to which (after nvq_conv_forward) the result is bit-identical.  s in {2, 3, 4}. */
int nvq_upsampler_tail_forward(const nvq_conv_desc* d, const float* frames, int T, int t_center, int Cimg, int s,
                               float* out, uint8_t* pass, void* stream);

/* ------------------------------------------------------------------ upsampler tail
 * PixelShuffle(s) + bicubic skip + clamp, efficient_layers.py:101-106 and
 * super_resolution.py:378-382: out[b,c,h*s+i,w*s+j] =
 *   clamp(u[b,h,w,c*s*s+i*s+j] + bicubic(frames[b,t_center])[b,c,h*s+i,w*s+j], 0, 1).
 * out: fp32 NCHW (B,Cimg,H*s,W*s); pass: uint8 1 where 0 <= pre-clamp <= 1. */
int nvq_shuffle_bicubic_clamp(const float* u, int u_ld, const float* frames, int B, int T,
                              int t_center, int Cimg, int H, int W, int s, float* out,
                              uint8_t* pass, void* stream);
/* EnhancementEngine strength blend, enhancement_engine.py:172-180:
 * out = strength * sr + (1 - strength) * F.interpolate(frames[:, t_center], bicubic, align_corners=False).
 * sr / out: fp32 NCHW (B,Cimg,H*s,W*s); frames: fp32 (B,T,Cimg,H,W). */
int nvq_bicubic_blend(const float* sr, const float* frames, int B, int T, int t_center, int Cimg,
                      int H, int W, int s, float strength, float* out, void* stream);
/* du[b,h,w,c*s*s+i*s+j] = dout[b,c,h*s+i,w*s+j] * pass ; channels up to u_ld zero-filled */
int nvq_shuffle_clamp_backward(const float* dout, const uint8_t* pass, int B, int Cimg, int H,
                               int W, int s, float* du, int du_ld, void* stream);

/* nn.PixelShuffle(s) alone (stand-alone PixelShuffleUpsampler, efficient_layers.py:101-106):
 * img[b,c,h*s+i,w*s+j] = u[b,h,w,c*s*s+i*s+j] (backward != 0: the other direction, padding channels of u zeroed) */
int nvq_pixel_shuffle(float* u, int u_ld, int B, int C, int H, int W, int s, float* img, int backward,
                      void* stream);

/* ------------------------------------------------------------------ loss
 * nn.MSELoss() of the training loops (experiments/train_baseline.py:64,86, train_continual.py:31,55) and F.mse_loss of
 * EWC.compute_fisher (ewc.py:125): *out = mean((a - b)^2) over n floats; workspace >= 8 KiB. */
int nvq_mse_forward(const float* a, const float* b, long n, float* out, float* workspace,
                    size_t workspace_bytes, void* stream);
/* da = (*grad_out_dev) * 2 (a - b) / n  (grad_out_dev: device scalar, NULL for 1) */
int nvq_mse_backward(const float* a, const float* b, long n, const float* grad_out_dev, float* da,
                     void* stream);

/* ------------------------------------------------------------------ FrameRecoveryNet layers (csrc/fr_ops.hip)
 * Generic NHWC kernels for reference nerve_cl/models/frame_recovery.py:23-446 (+ efficient_layers.py:109-151,
 * 231-294).  Tensors are [N,H,W,ld], logical channel count C <= ld, ld % 4 == 0, channels [C, ld) kept 0; fp32, or -
 * where an entry point has a `bf16` flag and it is set - bf16 for every activation tensor of that call (statistics,
 * parameters and their gradients are always fp32). */
/* dst[n,p,dst_coff+c] = src[n*src_nstride + c*H*W + p] (c < C), 0 for C <= c < czero: NCHW image (or frame t of a
 * (B,T,C,H,W) clip: src + t*C*H*W, src_nstride = T*C*H*W) into an NHWC slice */
int nvq_nchw_to_nhwc(const float* src, long src_nstride, int N, int C, int H, int W, float* dst,
                     int dst_ld, int dst_coff, int czero, void* stream);
int nvq_nhwc_to_nchw(const float* src, int src_ld, int src_coff, int N, int C, int H, int W,
                     float* dst, long dst_nstride, void* stream);
/* nn.BatchNorm2d / BatchNorm3d (frame_recovery.py:45,73,285-305; efficient_layers.py:139,266,279) for any C: statistics
 * over all npix pixels (train: batch mean / biased variance, running statistics updated with the unbiased variance;
 * running_mean may be NULL), then y = bn(x) (+ res) (ReLU if relu).  workspace >= nvq_bn2_workspace_bytes(C)
 * (+ 2*C floats for the backward). */
size_t nvq_bn2_workspace_bytes(int C);
int nvq_bn2_stats(const float* x, int x_ld, int C, long npix, float eps, float momentum, float* mean,
                  float* invstd, float* running_mean, float* running_var, float* workspace,
                  size_t workspace_bytes, int bf16, void* stream);
int nvq_bn2_eval_stats(const float* running_mean, const float* running_var, int C, float eps,
                       float* mean, float* invstd, void* stream);
int nvq_bn2_apply(const float* x, int x_ld, int C, long npix, const float* mean, const float* invstd,
                  const float* gamma, const float* beta, const float* res, int res_ld, int relu,
                  float* out, int out_ld, int bf16, void* stream);
/* g = dy masked by the forward ReLU (recomputed from x, res); dgamma = sum g*xhat, dbeta = sum g; dx as BatchNorm's
 * backward (training) or gamma*invstd*g (eval); dres = g when res != NULL */
int nvq_bn2_backward(const float* dy, int dy_ld, const float* x, int x_ld, int C, long npix,
                     const float* mean, const float* invstd, const float* gamma, const float* beta,
                     const float* res, int res_ld, int relu, int training, float* dx, int dx_ld,
                     float* dres, int dres_ld, float* dgamma, float* dbeta, float* workspace,
                     size_t workspace_bytes, int bf16, void* stream);
/* nn.MaxPool2d(k, s, pad) (frame_recovery.py:46) and F.max_pool3d(x, (1,2,2)) (:155,158): first maximum in scan order
 * wins (PyTorch's tie rule, matters after ReLU); idx = one byte per element; the backward is a gather (no atomics) */
int nvq_maxpool_forward(const float* x, int ld, int N, int H, int W, int k, int s, int pad, float* out,
                        uint8_t* idx, int bf16, void* stream);
int nvq_maxpool_backward(const float* dy, const uint8_t* idx, int ld, int N, int H, int W, int k, int s,
                         int pad, float* dx, int bf16, void* stream);
/* input side of nn.Conv2d(k=1, stride=2) (frame_recovery.py:71): out[n,y,x] = in[n,2y,2x]; backward = zero insertion
 * (in = gradient at the subsampled size, out = gradient at H x W) */
int nvq_subsample2(const float* in, int ld, int N, int H, int W, float* out, int backward, int bf16, void* stream);
/* F.interpolate(mode='bilinear', align_corners=False) (frame_recovery.py:224-229,433-436); backward (gather form):
 * in = dy [N,OH,OW], out = dx [N,H,W] */
int nvq_bilinear_resize(const float* in, int ld, int N, int H, int W, int OH, int OW, float* out,
                        int backward, void* stream);
/* nn.ConvTranspose2d(k=4, s=2, p=1) (frame_recovery.py:283-304) = 3x3 conv to 4*Co phase-major channels followed by
 * depth-to-space: nvq_convt_pack builds the [4*Co, Ci, 3, 3] conv weight from [Ci, Co, 4, 4], nvq_convt_unpack_grad
 * maps its gradient back, nvq_depth_space2 moves [N,H,W,4*Co] <-> [N,2H,2W,Co] */
int nvq_convt_pack(const float* w, int Ci, int Co, float* w3, void* stream);
int nvq_convt_unpack_grad(const float* dw3, int Ci, int Co, float* dw, void* stream);
int nvq_depth_space2(const float* in, float* out, int N, int H, int W, int Co, int to_depth, int bf16, void* stream);
/* TemporalConv3D's nn.Conv3d(Ci, Co, (3,1,1)) weight [Co,Ci,3,1,1] (efficient_layers.py:271-278) -> three 1x1 conv
 * weights [3][Co][Ci] (to_taps = 1) and back (its gradient, to_taps = 0): the temporal conv runs as three accumulating
 * 1x1 convolutions over time-shifted image ranges */
int nvq_tconv_relayout(const float* in, float* out, int Co, int Ci, int to_taps, void* stream);
/* The same convolution over a time-in-channels activation [B,H,W,T*Cp] (frame t = channels [t*Cp, t*Cp + C)): frame t of the
 * output is ONE 1x1 convolution over the channels of frames t-1..t+1 with the taps side by side - no accumulating passes.
 * nvq_tconv_cat lays taps k0 .. k0+nk-1 out as [Co][nk*Cp] (transpose = 0: forward / weight-gradient form) or as
 * [Ci][nk*Cp] with the tap order reversed (transpose = 1: input-gradient form over dy frames); padding columns are 0.
 * nvq_tconv_grad_combine folds the gradients of the first-frame (taps 1,2), middle (taps 0..2, may be NULL) and last-frame
 * (taps 0,1) forms back into dw [Co,Ci,3,1,1]. */
int nvq_tconv_cat(const float* w, int Co, int Ci, int Cp, int k0, int nk, int transpose, float* out, void* stream);
int nvq_tconv_grad_combine(const float* g_first, const float* g_mid, const float* g_last, int Co, int Ci, int Cp,
                           float* dw, void* stream);
/* per-image partial channel sums [N][nvq_gap_blocks(H,W)][C]: the input of nvq_cbam_channel for a stand-alone CBAM */
int nvq_gap_blocks(int H, int W);
int nvq_gap_partial(const float* x, int ld, int C, int N, int H, int W, float* part, void* stream);
/* x[n,p,c] += v[n,c] (gradient of the global average pool) */
int nvq_add_image_channel(float* x, int ld, int C, int N, int H, int W, const float* v, void* stream);
/* nn.Tanh (frame_recovery.py:309): out = tanh(a) | backward: out = a * (1 - y^2) */
int nvq_tanh(const float* a, const float* y, long n, float* out, int backward, void* stream);
/* FusionModule (frame_recovery.py:239-254): out = aligned + a0*mean_c(sp) + a1*mean_c(tp), (a0,a1) = softmax(logits[0:2]);
 * attn / means: [npix][2] saved for the backward; C a power of two in [16,256]; aligned / out / dy have ld = C */
int nvq_fusion_mix_forward(const float* aligned, const float* logits, int logits_ld, const float* sp,
                           int sp_ld, const float* tp, int tp_ld, int C, long npix, float* out,
                           float* attn, float* means, void* stream);
int nvq_fusion_mix_backward(const float* dy, const float* attn, const float* means, int C, long npix,
                            float* dlogits, int logits_ld, float* dsp, int sp_ld, float* dtp, int tp_ld,
                            void* stream);
/* FrameRecoveryNet blend (frame_recovery.py:439-440): out = frame*(1-m) + rec*m; frame/out NCHW, rec NHWC, m [N,1,H,W] */
int nvq_mask_blend(const float* frame, const float* rec, int rec_ld, const float* mask, int N, int C,
                   int H, int W, float* out, void* stream);
int nvq_mask_blend_backward(const float* dout, const float* mask, int N, int C, int H, int W,
                            float* drec, int rec_ld, void* stream);
/* SpatialEncoder stem nn.Conv2d(4, Co, 7, 2, 3, bias=False) (frame_recovery.py:42-44) on a 4-channel NHWC image;
 * w / dw in PyTorch layout [Co,4,7,7]; wgrad workspace >= 256*64*196 floats */
int nvq_stem7_forward(const float* x, const float* w, int N, int H, int W, int Co, float* out, int out_ld,
                      int out_bf16, void* stream);
int nvq_stem7_wgrad(const float* x, const float* dy, int dy_ld, int N, int H, int W, int Co, float* dw,
                    float* workspace, size_t workspace_bytes, int dy_bf16, void* stream);
/* dst[p, dst_coff + c] (+)= alpha * src[p, src_coff + c], c < C, each side fp32 or bf16 ([npix, ld] tensors): storage-type
 * boundary of FrameRecoveryNet's bf16 stages, mean / broadcast over the time groups (frame_recovery.py:164-165) */
int nvq_cast_slice(float* dst, int dst_ld, int dst_coff, int dst_bf16, const float* src, int src_ld,
                   int src_coff, int src_bf16, int C, long npix, float alpha, int accumulate, void* stream);

/* ------------------------------------------------------------------ small helpers */
/* dst[n,p,dst_coff+c] (+)= alpha*src[n,p,src_coff+c] [* (mask[n,p,mask_coff+c] > 0)] for c < C; src may be stored as
 * bf16 (src_bf16), dst and mask are fp32 */
int nvq_axpy_slice(float* dst, int dst_ld, int dst_coff, const float* src, int src_ld,
                   int src_coff, const float* mask, int mask_ld, int mask_coff, int C,
                   long npix, float alpha, int accumulate, int src_bf16, void* stream);
/* out[c] (+)= alpha * sum_pixels x[p, coff + c] */
int nvq_colsum(const float* x, int x_ld, int x_coff, int C, long npix, float alpha,
               float* out, float* workspace, size_t workspace_bytes, int accumulate,
               void* stream);

/* ------------------------------------------------------------------ EWC (flat buckets)
 * EWC.penalty, ewc.py:195-232: penalty = lambda/2 * sum F*(theta-theta_star)^2 over a flat
 * bucket of n floats; result written to *out (device scalar). */
int nvq_ewc_penalty(const float* theta, const float* theta_star, const float* fisher,
                    long n, float lambda, float* out, float* workspace,
                    size_t workspace_bytes, void* stream);
/* grad (+)= (*scale_dev) * lambda * F * (theta - theta_star): gradient of the penalty
 * times the upstream scalar gradient (device pointer, may be NULL for 1). */
int nvq_ewc_penalty_grad(const float* theta, const float* theta_star, const float* fisher,
                         long n, float lambda, const float* scale_dev, float* grad,
                         int accumulate, void* stream);
/* EWC.compute_fisher accumulation, ewc.py:139-141: fisher += grad^2 */
int nvq_fisher_accumulate(const float* grad, long n, float* fisher, void* stream);
/* SynapticIntelligence on flat buckets (theta, grad, p_old, W, omega in one layout, n floats each).
 * update_importance, ewc.py:343-354:  W += -grad * (theta - p_old);  p_old = theta.
 * register_task, ewc.py:356-368:      omega += W / ((theta - p_old)^2 + damping);  W = 0;  p_old = theta.
 * (The penalty si_lambda * sum omega (theta - p_old)^2, ewc.py:370-379, is nvq_ewc_penalty with lambda = 2 si_lambda.) */
int nvq_si_update(const float* theta, const float* grad, long n, float* p_old, float* W, void* stream);
int nvq_si_consolidate(const float* theta, long n, float damping, float* p_old, float* W, float* omega, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NVQ_H */
