/*
 * nerf_hip.h -- C ABI of libnerf_hip.so: NeRF volume rendering for AMD MI355X (gfx950).
 *
 * The reference (nerf-pytorch inside Jiankai-Sun/nerf-loc) has no FFI on this path: every
 * step is a PyTorch eager op.  Each entry point below therefore replaces a *function* of the
 * reference, cited as file:line under /root/reference; nerf_loc_amd/hip.py binds them with
 * ctypes and the Python host code keeps the reference's function names and signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - row-major, contiguous, float32 unless stated;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued asynchronously on it, nothing here synchronises or allocates;
 *   - return value: 0 on success, a negative NERF_E_* code otherwise; nerf_last_error()
 *     then returns a message for the calling thread.  NaN/Inf in data is never an error
 *     (the reference only prints a warning: run_nerf.py:420-422).
 *   - inputs are never written; outputs are fully overwritten.
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_OK 0
#define NERF_E_INVALID (-1)   /* bad size, null pointer, unsupported combination */
#define NERF_E_HIP (-2)       /* a HIP runtime call or launch failed */
#define NERF_E_UNSUPPORTED (-3)

#define NERF_PREC_F32 0       /* exact float32 MFMA (v_mfma_f32_32x32x2_f32), fmaf-chain numerics */
#define NERF_PREC_BF16 1      /* bf16 operands, float32 accumulation (v_mfma_f32_16x16x32_bf16 for   */
                              /* inference, v_mfma_f32_32x32x16_bf16 in the training kernels)        */
#define NERF_PREC_BF16X3 2    /* float32 accuracy on the bf16 matrix cores: every operand a sum of   */
                              /* three bf16 terms, the six leading products per element, float32    */
                              /* accumulation (inference; training: the forward pass, see below)    */
#define NERF_PREC_BF16X2 3    /* two bf16 terms, three products: operands good to 2^-16              */

/* what nerf_mlp_forward reads as its input */
#define NERF_IN_FEATURES 0    /* in0 = already encoded features [M, 63(+27)] (NeRF.forward)        */
#define NERF_IN_POINTS 1      /* in0 = pts [R*S, 3], in1 = viewdirs [R, 3] or NULL (run_network)   */
#define NERF_IN_RAYS 2        /* in0 = rays [R, ray_cols], in1 = z_vals [R, S]: pts = o + d*z made */
                              /* in registers, never written to memory (render_rays)               */

/* Version of this interface (2: the training entry points take a `precision` argument; 3: adds
 * nerf_mlp_backward_static; 4: adds nerf_mlp_forward_objects; 5: adds nerf_make_rays_views; 6: the inference entry
 * points accept NERF_PREC_BF16X3 / NERF_PREC_BF16X2; adds nerf_render_rays; nerf_raw2outputs_backward takes g_weights;
 * 7: the training entry points accept NERF_PREC_BF16X3). */
#define NERF_HIP_ABI_VERSION 7
int nerf_hip_abi_version(void);
const char* nerf_last_error(void);

/* Verifies on the current device that the MFMA operand/result lane maps this library is
 * built on hold (exact small-integer products).  Synchronises the device.  0 = ok. */
int nerf_selftest_mfma(void);

/* ---- rays ------------------------------------------------------------------------------
 * get_rays (run_nerf_helpers.py:153-162), the view-direction normalisation, optional NDC
 * warp (ndc_rays, run_nerf_helpers.py:175-192) and the [N, 8|11] packing of render()
 * (run_nerf.py:95-123), in one pass.
 *   o_in/d_in  [N,3] each, or both NULL to generate the H x W pixel grid (N = H*W) from
 *              c2w [3,4] and the intrinsics fx, fy, cx, cy;
 *   c2w_view   optional second pose [3,4]: view directions come from it while origins and
 *              directions come from c2w (the c2w_staticcam case, run_nerf.py:105-107);
 *   near_arr/far_arr optional per-ray bounds [N]; otherwise the scalars are used;
 *   rays_out   [N, 8] or, with use_viewdirs, [N, 11].                                     */
int nerf_make_rays(int n_rays, int H, int W, const float* o_in, const float* d_in,
                   const float* c2w, const float* c2w_view,
                   float fx, float fy, float cx, float cy,
                   float near, float far, const float* near_arr, const float* far_arr,
                   int use_viewdirs, int ndc, float ndc_focal, float ndc_near,
                   float* rays_out, void* stream);

/* The pixel grids of n_views poses in one launch: what render_path (run_nerf.py:137-181, nerf_utils.py:107-151) and
 * Model3DETR.forward (model_3detr.py:619-724) obtain from one get_rays + render() packing per pose.  views [n_views,16]
 * on the device: c2w (3x4, row-major) then fx, fy, cx, cy; no NDC warp, no second pose.  rays_out [n_views*H*W, 8|11],
 * view after view.                                                                          */
int nerf_make_rays_views(int n_views, int H, int W, const float* views, float near, float far, int use_viewdirs,
                         float* rays_out, void* stream);

/* Stratified depths, run_nerf.py:363-385.  t_vals [S] = linspace(0,1,S) as the host's torch
 * computes it; t_rand [R,S] uniform variates or NULL (perturb = 0).  z_out [R,S].          */
int nerf_z_vals(const float* rays, int ray_cols, int R, int S, const float* t_vals,
                const float* t_rand, int lindisp, float* z_out, void* stream);

/* Embedder.embed as a standalone op (run_nerf_helpers.py:15-45): x [M,3] ->
 * out [M, (include_input ? 3 : 0) + 6*multires] = [x, sin(2^0 x), cos(2^0 x), ...].  The fused
 * network kernel below never calls it; it exists for callers that use get_embedder directly. */
int nerf_embed(const float* x, long long M, int multires, int include_input, float* out,
               void* stream);

/* How many NaN / Inf values each of up to 16 tensors holds, in one launch: the census behind the reference's
 * "! [Numerical Error] ... contains nan or inf." warnings (run_nerf.py:420-422; nerf_utils.py:336-340 checks every
 * render).  ptrs_host / numel_host: HOST arrays of device pointers and element counts; counts_dev: DEVICE int32
 * [n_tensors], zeroed and filled by the call.                                                               */
int nerf_count_nonfinite(int n_tensors, const float* const* ptrs_host, const long long* numel_host,
                         int* counts_dev, void* stream);

/* ---- network ---------------------------------------------------------------------------
 * NeRF(D=8, W=256, input_ch=63, input_ch_views=27, skips=[4]) of run_nerf_helpers.py:67-119,
 * together with the positional encoding (run_nerf_helpers.py:15-63) and the flatten /
 * expand / concatenate / chunk logic of run_network and batchify (run_nerf.py:27-51).
 *
 * Weights are first re-laid-out into the MFMA fragment stream the kernel consumes.
 *   tensors_host: HOST array of device pointers in state_dict order: with use_viewdirs 24
 *   entries (pts_linears.0..7 weight,bias; views_linears.0; feature_linear; alpha_linear;
 *   rgb_linear), otherwise 20 (pts_linears.0..7; views_linears.0 (ignored, may be NULL);
 *   output_linear with output_ch rows, 4 or 5).                                            */
size_t nerf_mlp_packed_bytes(int precision, int use_viewdirs);
int nerf_mlp_pack(const float* const* tensors_host, int use_viewdirs, int output_ch,
                  int precision, void* packed_out, void* stream);

/* raw_out [M, output_ch] with M = R*S (NERF_IN_FEATURES: M = R, S = 1).  With use_viewdirs
 * the output is [rgb(3), sigma] (run_nerf_helpers.py:115).
 * Domain of NERF_PREC_BF16: network inputs (features, coordinates) of magnitude <= 2^24 (the
 * pack-time activation scales are derived from that bound; beyond it hidden activations may
 * saturate), and the sin/cos features keep bf16 accuracy for |coordinate| up to about 100
 * (fast range reduction).  NERF_PREC_F32 has neither limit.
 * NERF_PREC_BF16X3 evaluates the same network with every operand written as a sum of three bf16 numbers and the
 * six leading products per element, float32 accumulation: results within the float32 kernel's own error of the
 * reference (measured 1e-6 of the output scale) at about twice its speed; NERF_PREC_BF16X2 (two terms, three
 * products) is good to 2^-16.  Both take inputs of any magnitude (library sine / cosine, no activation scaling) and
 * serve nerf_mlp_packed_bytes, nerf_mlp_pack, nerf_mlp_forward and nerf_mlp_forward_objects; the training entry points
 * take NERF_PREC_F32 or NERF_PREC_BF16 only.                                                  */
int nerf_mlp_forward(const void* packed, int precision, int use_viewdirs, int output_ch,
                     int input_mode, const float* in0, const float* in1,
                     int R, int S, int ray_cols, float* raw_out, void* stream);

/* run_network for SEVERAL independent networks in one launch: the multi-NeRF workload (one network per object,
 * run_multi_nerf.py:560-572) as NeRF-Loc consumes it, a batch of scans each rendered with its own network
 * (model_3detr.py:619-724).  The flat batch holds n_objects equal parts: rays [o R/n, (o+1) R/n) (and their samples)
 * are evaluated with network o.  packed_table_dev: DEVICE array of n_objects pointers to packed streams
 * (nerf_mlp_pack; same precision, use_viewdirs and output_ch for all).  R must divide evenly; n_objects at most the number of compute units.  Every other argument as for nerf_mlp_forward; results are bit-identical to n_objects separate
 * nerf_mlp_forward calls.                                                                                   */
int nerf_mlp_forward_objects(const void* const* packed_table_dev, int n_objects, int precision, int use_viewdirs,
                             int output_ch, int input_mode, const float* in0, const float* in1,
                             int R, int S, int ray_cols, float* raw_out, void* stream);

/* ---- compositing -----------------------------------------------------------------------
 * raw2outputs, run_nerf.py:268-311 (and nerf_utils.py:344-395).  raw [R,S,raw_ch] (raw_ch 4
 * or 5; channels 0-2 colour, 3 density), z_vals [R,S], rays_d read from d_ptr with row
 * stride d_stride floats (so rays+3 with stride ray_cols works), noise [R,S] already scaled
 * or NULL.  Outputs rgb [R,3], disp [R], acc [R], depth [R], weights [R,S]; any may be NULL.
 * disp is NaN where acc == 0, like the reference.  S = 1 is the reference's degenerate case:
 * its interval list is then empty (run_nerf.py:283-284), nothing is composited, acc = depth = 0,
 * disp = NaN, rgb = the background and `weights` (zero columns there) is left untouched.       */
int nerf_raw2outputs(const float* raw, int raw_ch, const float* z_vals, const float* d_ptr,
                     int d_stride, const float* noise, int R, int S, int white_bkgd,
                     float* rgb, float* disp, float* acc, float* depth, float* weights,
                     void* stream);

/* ---- hierarchical sampling -------------------------------------------------------------
 * sample_pdf (run_nerf_helpers.py:196-239) plus the surrounding mid-point, sort(cat) and
 * z_std code of render_rays (run_nerf.py:397-401, 418).
 *   bins    [R, nb] with row stride bins_stride; if bins_are_z != 0 the pointer holds
 *           z_vals [R, nb+1] instead and bins = .5*(z[1:]+z[:-1]) are formed on the fly;
 *   weights [R, nb-1] with row stride w_stride (pass weights+1 and stride S for [...,1:-1]);
 *   u       [n_u] shared by all rays (u_per_ray = 0) or [R, n_u] (u_per_ray = 1);
 *   z_samples [R, n_u], inds [R, n_u] (int32 searchsorted results), z_std [R]: optional;
 *   z_sorted  [R, n_z + n_u]: optional, sorted union of z_vals [R, n_z] and the samples.
 * The float32 arithmetic follows ATen's CPU kernels operation by operation, so indices and
 * samples are bit-identical to the reference on CPU for identical inputs.                  */
int nerf_sample_pdf(const float* bins, int bins_stride, int bins_are_z, int nb,
                    const float* weights, int w_stride,
                    const float* u, int n_u, int u_per_ray, int R,
                    const float* z_vals, int n_z,
                    float* z_samples, int32_t* inds, float* z_sorted, float* z_std,
                    void* stream);

/* ---- the whole per-ray pipeline in one call ----------------------------------------------
 * render_rays (run_nerf.py:314-424; the coarse-only flavour with a depth map: nerf_utils.py:248-342): stratified
 * depths, coarse network, compositing and, with n_importance > 0, inverse-CDF sampling, the sorted union of depths,
 * the fine network and the final compositing.  A convenience over the entry points above (it enqueues exactly their
 * kernels, in the reference's order, on `stream`; results are bit-identical to calling them one by one), for callers
 * that want one call per ray batch.  All pointers are device pointers; the caller owns every buffer; optional
 * outputs may be NULL.  rays [R, ray_cols] as render() packs them (run_nerf.py:116-123).                       */
typedef struct {
    int n_samples;          /* S: coarse samples per ray                                                      */
    int n_importance;       /* extra samples of the fine pass, 0 = coarse pass only                           */
    int precision;          /* NERF_PREC_* of both networks                                                   */
    int use_viewdirs;       /* rays then have 11 columns and the networks their view branch                   */
    int output_ch;          /* 4, or 5 without view directions                                                */
    int ray_cols;           /* row stride of rays, >= 8 (11 with view directions)                             */
    int lindisp;            /* depths linear in disparity                                                     */
    int white_bkgd;
    int u_per_ray;          /* u is [R, n_importance] instead of [n_importance]                               */
    const float* t_vals;    /* [S] = linspace(0, 1, S) as the host's torch computes it                        */
    const float* t_rand;    /* [R, S] uniform variates for the stratified jitter, or NULL (perturb = 0)       */
    const float* u;         /* inverse-CDF variates (n_importance > 0)                                        */
    const float* noise0;    /* [R, S] scaled density noise of the coarse pass, or NULL                        */
    const float* noise1;    /* [R, S + n_importance] of the fine pass, or NULL                                */
} nerf_render_cfg;
typedef struct {
    float *rgb, *disp, *acc, *depth;      /* maps of the last pass: [R,3], [R], [R], [R]; rgb required        */
    float *rgb0, *disp0, *acc0;           /* coarse-pass maps when n_importance > 0                           */
    float *z_std;                         /* [R] standard deviation of the importance samples                 */
    float *raw;                           /* [R, S + n_importance, output_ch]: the last pass's network output  */
    float *z_vals;                        /* [R, S + n_importance]: the last pass's depths                     */
    float *z_coarse, *raw_coarse;         /* n_importance > 0: [R, S], [R, S, output_ch] work buffers          */
    float *weights;                       /* [R, S] coarse weights: required when n_importance > 0, else optional */
} nerf_render_buffers;
int nerf_render_rays(const void* packed_coarse, const void* packed_fine_or_null, const float* rays, int R,
                     const nerf_render_cfg* cfg, const nerf_render_buffers* buf, void* stream);

/* ---- training ---------------------------------------------------------------------------
 * The reference trains through PyTorch autograd (run_nerf.py:770-796: render -> img2mse ->
 * loss.backward() -> Adam).  These entry points produce what autograd would: gradients of a
 * scalar loss with respect to raw (compositing) and to the 24 parameter tensors (network), for
 * the D=8 / W=256 architecture (with or without view directions).  `precision` selects the
 * arithmetic of the network kernels: NERF_PREC_F32 (exact float32 MFMA), NERF_PREC_BF16 (bf16
 * operands, float32 accumulation; parameters and gradients stay float32) or NERF_PREC_BF16X3:
 * forward pass and input-gradient chain evaluate the float32 network in three bf16 terms (float32
 * accuracy at about 0.67 of the exact kernels' time) on the float32 tapes, whose sizes are the
 * float32 ones; the weight gradients are the float32 kernels'; the backward stream is its own
 * (nerf_mlp_backward_packed_bytes); at most 4 194 304 samples per call.  Tapes, packed streams and
 * entry points of one training step must all use the same precision.
 *
 * nerf_raw2outputs_backward: backward of raw2outputs (run_nerf.py:268-311).  g_rgb [R,3], g_disp,
 * g_acc, g_depth [R] and g_weights [R,S] are the upstream gradients of the maps and of the
 * per-sample weights (any may be NULL = zero); d_raw [R,S,raw_ch] receives dL/draw.  S <= 256.
 * No gradient flows to z_vals or rays_d.                                                      */
int nerf_raw2outputs_backward(const float* raw, int raw_ch, const float* z_vals, const float* d_ptr,
                              int d_stride, const float* noise, int R, int S, int white_bkgd,
                              const float* g_rgb, const float* g_disp, const float* g_acc,
                              const float* g_depth, const float* g_weights, float* d_raw, void* stream);

/* Buffers the caller owns: the activation tape written by the training forward and read by
 * the backward, the gradient tape (scratch of the backward), and a fixed-size workspace.     */
size_t nerf_train_tape_bytes(long long M, int use_viewdirs, int precision);
size_t nerf_train_grad_tape_bytes(long long M, int use_viewdirs, int precision);
size_t nerf_train_workspace_bytes(void);

/* nerf_mlp_forward that also fills `tape` (NeRF.forward with autograd recording,
 * run_nerf_helpers.py:96-119).  `packed` comes from nerf_mlp_pack for the same precision,
 * use_viewdirs and output_ch.                                                                */
int nerf_mlp_forward_train(const void* packed, int precision, int use_viewdirs, int output_ch,
                           int input_mode, const float* in0, const float* in1, int R, int S,
                           int ray_cols, float* raw_out, void* tape, void* stream);

/* Transposed weights for the input-gradient chain; same tensors_host as nerf_mlp_pack (biases
 * ignored).  Re-pack whenever the weights change.                                             */
size_t nerf_mlp_backward_packed_bytes(int use_viewdirs, int precision);
int nerf_mlp_pack_backward(const float* const* tensors_host, int use_viewdirs, int output_ch,
                           int precision, void* packed_out, void* stream);

/* Gradients of the parameter tensors given d_raw [M, output_ch] = dL/draw (M = R*S of the
 * forward).  grads_host: HOST array of device pointers in state_dict order (24 with view
 * directions, 20 without: entries 16, 17 = views_linears.0 are then unused and may be NULL),
 * each shaped like its parameter; every element is overwritten (not accumulated).
 * Deterministic: partial sums are combined in a fixed order.  No gradient is produced for the
 * network inputs.  packed_fwd: the forward stream of nerf_mlp_pack the tape was made with
 * (read in NERF_PREC_BF16 only, for its activation scales; may be NULL in NERF_PREC_F32).     */
int nerf_mlp_backward(const void* packed_bwd, const void* packed_fwd, int precision, int use_viewdirs,
                      int output_ch, const float* d_raw, long long M, const void* tape,
                      void* grad_tape, float* workspace, size_t workspace_bytes,
                      float* const* grads_host, void* stream);

/* nerf_mlp_backward for callers that cannot allocate device memory: code under HIP graph capture, tight
 * training / serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference (which
 * allocates freely through PyTorch).  Same arguments and results as nerf_mlp_backward, bit for bit, minus the two
 * scratch buffers: the gradient tape and the partial-sum workspace live in a static device array of the
 * library (NERF_STATIC_SCRATCH_BYTES = 163 840 000 bytes, one copy on every GPU the library is used on, held
 * until the process ends).  Limit: M <= NERF_STATIC_MAX_SAMPLES per call; larger M is refused with
 * NERF_E_INVALID.  Calls neither hipMalloc nor hipFree nor any allocating routine.  All calls on one GPU share
 * that GPU's array: calls that overlap in time on one GPU (different streams or threads) race; serialise them.
 * The first call on a device looks the array's address up (hipGetSymbolAddress); make that call before starting
 * a graph capture.  The device the stream belongs to must be the current device.                          */
#define NERF_STATIC_MAX_SAMPLES 8192
#define NERF_STATIC_SCRATCH_BYTES 163840000
int nerf_mlp_backward_static(const void* packed_bwd, const void* packed_fwd, int precision, int use_viewdirs,
                             int output_ch, const float* d_raw, long long M, const void* tape,
                             float* const* grads_host, void* stream);

/* One Adam update of up to 64 parameter tensors in a single launch: torch.optim.Adam as the
 * reference constructs it (run_nerf.py:225, betas (0.9, 0.999), eps 1e-8, no amsgrad), the
 * optimizer.step() of run_nerf.py:787.  The four HOST arrays hold device pointers per tensor
 * (parameter, gradient, exp_avg, exp_avg_sq), numel_host their element counts; step counts
 * from 1.  Parameters and both moment buffers are updated in place.                          */
int nerf_adam_step(int n_tensors, float* const* params_host, const float* const* grads_host,
                   float* const* exp_avg_host, float* const* exp_avg_sq_host,
                   const long long* numel_host, long long step, float lr, float beta1, float beta2,
                   float eps, float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_HIP_H */
