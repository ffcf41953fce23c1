/* atlasfit.h — C ABI of libatlasfit.so: the MI355X-native (gfx950) implementation of the stage-1
 * neural-atlas optimisation loop of All-In-One-Deflicker.
 *
 * The reference has no plugin/operator API for this path; its seams are the stage-1 script and the
 * Python objects it drives.  Each entry point below names the reference code it replaces
 * (paths relative to the reference repository root).  A handle owns ONE video on ONE device with ONE
 * stream (reference: one process per GPU via CUDA_VISIBLE_DEVICES, src/stage1_neural_atlas.py:267-268).
 *
 * Conventions: every function returns 0 on success and a negative af_status on failure (message via
 * af_last_error); nothing throws across the boundary; host buffers are borrowed for the duration of
 * the call; the handle owns all device memory.  All tensors are fp32 unless stated.
 */
#ifndef ATLASFIT_H
#define ATLASFIT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct af_handle af_handle;

enum af_status { AF_OK = 0, AF_EINVAL = -1, AF_EHIP = -2, AF_ENOMEM = -3, AF_ENAN = -4, AF_ESTATE = -5,
                 AF_ERANGE = -6   /* af_set_mlp_mode(h, 3) only: a hidden-layer weight left the range its fp16 images are scaled for (|w| >= 8) */ };
enum af_net { AF_MAPPING1 = 0, AF_ATLAS = 1, AF_MAPPING2 = 2, AF_ALPHA = 3 };

/* Mirrors the keys of src/config/config_flow_100.json that the loop reads
 * (src/stage1_neural_atlas.py:28-90) plus the sizes main() derives (resx, resy, number_of_frames). */
typedef struct af_config {
  int32_t resx, resy, number_of_frames;          /* stage1_neural_atlas.py:31-38,108 */
  int32_t samples_batch;                         /* config :7 */
  /* number_of_channels_*: 1..256 (a narrower net runs exactly, zero-padded inside the 256-wide kernels; all flat parameter / Adam-state / gradient
   * buffers of this ABI are in the CONFIGURED net's state_dict order and size); number_of_layers_*: 2..8 */
  int32_t number_of_channels_mapping1, number_of_layers_mapping1;   /* config :24-25  (256, 6) */
  int32_t number_of_channels_atlas, number_of_layers_atlas;         /* config :19-20  (256, 8) */
  int32_t positional_encoding_num_atlas;         /* config :31 (10) */
  int32_t use_positional_encoding_mapping1;      /* config :32 (false) */
  int32_t derivative_amount;                     /* config :10 */
  int32_t include_global_rigidity_loss;          /* config :39 */
  int32_t global_rigidity_derivative_amount_fg;  /* config :40 */
  int32_t stop_global_rigidity;                  /* config :44 */
  int32_t use_gradient_loss;                     /* config :29 (true); false: the gradient term is 0 and carries no gradient (stage1_neural_atlas.py:185-190) */
  float rgb_coeff, gradient_loss_coeff, rigidity_coeff, optical_flow_coeff;   /* config :11,28,12,8 */
  float global_rigidity_coeff_fg;                /* config :42 */
  float uv_mapping_scale;                        /* config :13 */
  float lr;                                      /* 1e-4, hard-coded at stage1_neural_atlas.py:134 */
  int32_t pretrain_batch;                        /* 10000, hard-coded at unwrap_utils.py:182-183 */
  /* ---- fg/bg dual-atlas path (src/stage1_neural_atlas_seg.py:27-107); read only when two_layer != 0 ---- */
  int32_t two_layer;                             /* 0: stage1_neural_atlas.py   1: stage1_neural_atlas_seg.py */
  int32_t number_of_channels_mapping2, number_of_layers_mapping2;   /* config :26-27  (256, 4) */
  int32_t number_of_channels_alpha, number_of_layers_alpha;         /* config :21-22  (256, 8) */
  int32_t positional_encoding_num_alpha;         /* config :18 (5) */
  int32_t use_positional_encoding_mapping2;      /* config :34 (false) */
  int32_t global_rigidity_derivative_amount_bg;  /* config :41 */
  int32_t stop_bootstrapping_iteration;          /* config :23 */
  float global_rigidity_coeff_bg;                /* config :43 */
  float alpha_bootstrapping_factor, alpha_flow_factor, sparsity_coeff;   /* config :16,17,30 */
  int32_t number_of_positional_encoding_mapping1;   /* config :33 (4): frequencies K of mapping1's PE (3 -> 6K), 1..5, read when use_positional_encoding_mapping1 */
  int32_t number_of_positional_encoding_mapping2;   /* config :35 (2): the same for mapping2 */
  int32_t reserved[2];
} af_config;

/* Replaces model construction + optimizer construction (stage1_neural_atlas.py:112-134).  Parameters
 * start at zero; load them with af_set_params. */
int af_create(const af_config* cfg, int device_ordinal, af_handle** out);
size_t af_config_size(void);                     /* sizeof(af_config) of this build: bindings check their mirror against it */
void af_destroy(af_handle* h);
const char* af_last_error(const af_handle* h);   /* h may be NULL: message of the last failed af_create */

/* Replaces the tensors returned by load_input_data_single (src/models/stage_1/unwrap_utils.py:105-163),
 * the dx/dy construction (:132-133) and get_tuples (:166-173; the index table is arithmetic here).
 * Layouts are the reference's: frames (resy,resx,3,F); flows (resy,resx,2,F[,1]); masks (resy,resx,F[,1]);
 * mask_fg (resy,resx,F) or NULL.  on_device != 0: the pointers are device pointers on this handle's GPU. */
int af_upload_video(af_handle* h, const float* frames, const float* flow_fwd, const float* flow_bwd,
                    const float* mask_fwd, const float* mask_bwd, const float* mask_fg, int on_device);

/* ---- input builder (the device side of load_input_data / load_input_data_single, unwrap_utils.py:40-163) ----------
 * Stateless utilities on `device_ordinal`; with on_device != 0 the pointers are device pointers, else host buffers.
 *
 * af_resize_bilinear: cv2.resize(src, (dw, dh)) with the default INTER_LINEAR geometry (half-pixel centres, edge clamp,
 * no anti-aliasing; unwrap_utils.py:35,131).  src is HWC contiguous, float32 (src_u8 = 0) or uint8 (src_u8 = 1: divided
 * by 255 first, :128).  dst element (y, x, c) is written at dst[(y*dw + x)*pix_stride + c*ch_stride + offset] — e.g.
 * frame f of video_frames (resy,resx,3,F): pix_stride 3F, ch_stride F, offset f.  scale0/scale1 multiply channels 0/1
 * (resize_flow's newh/oldh and neww/oldw, :36-37); pass 1 for images.
 * af_flow_consistency: out[(y*w + x)*pix_stride + offset] = || f12 + remap(f21, f12) ||_2 (:10-23, bilinear, zero
 * border) if thresh <= 0, else 1.0 / 0.0 for norm < thresh (the mask of :151-159 with thresh = 1). */
int af_resize_bilinear(int device_ordinal, const void* src, int src_u8, int sh, int sw, int ch, float* dst, int dh, int dw,
                       int64_t pix_stride, int64_t ch_stride, int64_t offset, double scale0, double scale1, int on_device);
int af_flow_consistency(int device_ordinal, const float* f12, const float* f21, int h, int w, float* out,
                        int64_t pix_stride, int64_t offset, float thresh, int on_device);

/* IMLP.state_dict() order: hidden.0.weight (out,in) row-major, hidden.0.bias, hidden.1.weight, ...
 * (src/models/stage_1/implicit_neural_networks.py:37-52). */
size_t af_param_count(const af_handle* h, int net);
int af_set_params(af_handle* h, int net, const float* flat, size_t n);
int af_get_params(af_handle* h, int net, float* flat, size_t n);
/* torch.optim.Adam state of optimizer_all (exp_avg, exp_avg_sq, step) for one net's parameters. */
int af_get_adam_state(af_handle* h, int net, float* exp_avg, float* exp_avg_sq, int64_t* step);
int af_set_adam_state(af_handle* h, int net, const float* exp_avg, const float* exp_avg_sq, int64_t step);

/* pre_train_mapping (src/models/stage_1/unwrap_utils.py:176-198): pretrain_iters x number_of_frames Adam
 * steps (own optimizer, lr 1e-4) on `net` (AF_MAPPING1/2).  ys/xs: [pretrain_iters*F][pretrain_batch]
 * row / column draws in the reference's order (i_s then j_s), or NULL for the device sampler(seed).
 * losses_out: [pretrain_iters*F] mean loss per step, or NULL. */
int af_pretrain(af_handle* h, int net, int pretrain_iters, const int64_t* ys, const int64_t* xs,
                uint64_t seed, float* losses_out);

/* The loop body (src/stage1_neural_atlas.py:151-231, or src/stage1_neural_atlas_seg.py:191-315 for a
 * two_layer handle) for iterations first_iter .. first_iter+n_iters-1.
 * inds: [n_iters][samples_batch] values of inds_foreground (:159-160), or NULL for the device sampler.
 * losses_out: [n_iters][af_loss_width(h)] or NULL.
 *   single (width 8): rgb, gradient, rigidity, global rigidity, flow, total, #valid fwd, #valid bwd
 *     (the un-weighted terms of :186-218 and the weighted sum of :220-227);
 *   two_layer (width 16): rgb, gradient, rigidity1, rigidity2, global rigidity1, global rigidity2, flow1, flow2,
 *     alpha-flow, alpha bootstrapping (BCE), sparsity, total, #valid fwd, #valid bwd, 0, 0
 *     (stage1_neural_atlas_seg.py:237-311). */
int af_train_steps(af_handle* h, int first_iter, int n_iters, const int64_t* inds, uint64_t seed,
                   float* losses_out);
int af_loss_width(const af_handle* h);

/* Forward-only reconstruction of one frame (src/models/stage_1/evaluate.py:640-661):
 * rgb_out (resy,resx,3) host buffer or NULL; sse_out: sum of squared error vs the input frame (fp64). */
int af_render_frame(af_handle* h, int frame, float* rgb_out, double* sse_out);
/* Mean over frames of skimage PSNR(data_range=1) (evaluate.py:740-743,775); per_frame[F] optional. */
int af_psnr(af_handle* h, double* mean_psnr, double* per_frame);

int af_sync(af_handle* h);

/* ---- test / measurement hooks (not part of the reference surface) --------------------------------- */
/* Run one net forward on caller rows: in [rows][4] host -> out [rows][4] host. */
int af_debug_forward(af_handle* h, int net, const float* in, int rows, float* out);
/* Debug: the tensors the LAST training step left in HBM for the weight-gradient GEMMs, as the kernels wrote them (per row tile of 32 rows):
 * which = 0 activation plane `layer` (relu(Z_layer) = X_{layer+1}, [256 features][32 rows]), 1 gradient plane (dZ_layer), 2 its sign-bit words
 * (copied as 256 x 32-bit per tile), 3 the PE features [64][32], 4 dZ of the output layer [32][32], 5 the xyt rows [32][32].  nt_stride = row tiles per
 * plane of that step (ceil(rows of the net's batch / 32)); `ntiles` tiles from `tile0` go to out.  What tests/test_gpu_gemm_error.py measures the
 * per-layer error of each arithmetic on: one layer's product recomputed in fp64 from the kernel's OWN inputs against the kernel's output. */
int af_debug_tiles(af_handle* h, int net, int which, int layer, int nt_stride, int tile0, int ntiles, float* out);
/* The launch plan of the single-atlas step for a chip of `ncu` compute units (pure arithmetic, no GPU needed):
 * out3 = {T1, T2, NT}: mapping row tiles [0,T1) form launch 1, [T1,T2) lead and [T2,NT) trail the atlas part of
 * launch 2 (DESIGN.md §2.1 "Packed launches"). */
int af_debug_plan(int ncu, int rows_map, int rows_atlas, int dep_rows, int out3[3]);
/* Balance diagnostics of k_dw: enable != 0 makes every later k_dw launch record s_memrealtime (100 MHz) at the start and
 * end of each workgroup; out (nullable) receives [min(cap_wg, #CUs)][2] values of the most recent launch.  Returns #CUs. */
int af_debug_dw_clocks(af_handle* h, int enable, uint64_t* out, int cap_wg);
/* The clock each hot kernel runs at INSIDE the training step: enable != 0 makes the five hot launches of every later step (forward 1, 2,
 * backward 1, 2 of the chains in any mlp_mode, k_dw) record per workgroup {s_memrealtime, s_memtime} at its start and at its end; out
 * (nullable) receives [5][min(cap_wg, 4096)][4] uint64 of the most recent step (zeros for workgroups a launch did not have) and the
 * buffer is cleared.  Only the first 4096 workgroups of a launch are stamped (samples_batch 100 000 launches ~7 000 - 18 000; the rest
 * are skipped, nothing is written past a launch's region).  Ticks over the 100 MHz span = the shader clock under that launch's load
 * (tools/step_clock.py).  Returns 4096 (the cap, not a grid size). */
int af_debug_step_clocks(af_handle* h, int enable, uint64_t* out, int cap_wg);
/* The static split-K schedule of k_dw: which = 0 (9 row segments), 1 (7), 2 / 3 (pre-train of mapping1 / mapping2).
 * out (nullable) [min(cap_wg, #workgroups)][16][4] int32 = {job shape 0..4 (8x8, 8x2, 8x1, 1x8, 1x2; -1 ends a list), first row tile,
 * one past the last, job index}.  Returns the number of workgroups.  Used by tools/dw_fit.py to fit the schedule's cost model. */
int af_debug_dw_schedule(af_handle* h, int which, int32_t* out, int cap_wg);
/* Read back n 64-byte pixel records of the packed table: out [n][16] = rgb(3), d/dx rgb(3), d/dy rgb(3), fwd flow(2),
 * bwd flow(2), fwd mask, bwd mask, fg mask, for pixel-frame indices inds[n] (the k of get_tuples' column k). */
int af_debug_records(af_handle* h, const int64_t* inds, int n, float* out);
/* Arithmetic of the weight-gradient GEMMs (k_dw), for the loop AND for pre_train_mapping's dW: 1 (default) = fp32-faithful
 * "bf16x6" — each fp32 operand split in registers into three bf16 values, the six leading partial products accumulated in
 * fp32 on the bf16 matrix pipe (dropped terms <= 2^-23 relative: tests/test_split_precision.py).  2 (opt-in) = "bf16x3" —
 * two bf16 values per operand (16 mantissa bits), three partial products: NARROWER than the reference's fp32, faster,
 * measured within 3x of torch-fp32's own gradient error against an fp64 twin at full size (tests/test_gpu_fullsize.py);
 * never the default and never bench.py's headline value.  0 = the fp32 matrix pipe (v_mfma_f32_32x32x2_f32).  The three
 * are held against each other in tests/test_gpu_dw_modes.py.  The library reads no environment: the Python mirror maps
 * AF_DW_MODE=<m> / AF_DW_FP32=1 onto this call for the A/B tools.
 * A switch re-cuts all split-K schedules (their tile costs belong to the arithmetic). */
int af_set_dw_mode(af_handle* h, int mode);
/* Experiments and the partition-sensitivity tests: replace the per-shape tile costs k_dw's static split-K schedule is cut with
 * (cost5 = 8x8, 8x2, 8x1, 1x8, 1x2 tiles; every value finite and > 0, ratios <= 1000:1; seg_cost <= 0 keeps the shipped per-segment
 * cost) and re-cut all schedules; cost5 == NULL returns to the shipped row of the current arithmetic.  Another row = another
 * partition of the row batch over workgroups = another summation ORDER of the same partial products, nothing else; results stay
 * bit-reproducible for a given row.  AF_EINVAL (handle unchanged) for a row that is not valid or cannot be scheduled. */
int af_debug_set_dw_cost(af_handle* h, const double* cost5, double seg_cost);
/* The same choice for the 256x256 hidden-layer products of the forward / backward chains: 3 (default since round 6) = "f16x3" (mlphf.hip): every
 * operand as two fp16 terms of its scaled value, three products on v_mfma_f32_32x32x16_f16, a power-of-two scale per ROW and layer on the activations /
 * gradients and a fixed 2^12 on the weights — measured from the kernels' own tiles, its per-layer error against fp64 is below an fp32 fmaf chain's
 * (tests/test_gpu_gemm_error.py); af_train_steps / af_pretrain return AF_ERANGE once a hidden-layer weight reaches |w| >= 8 (the images stay finite up
 * to 16; mode 1 has no such limit).  1 = bf16x6 (mlpbf.hip: three bf16 terms, six products; the default of rounds 2-5), 0 = fp32 matrix pipe (mlp.hip),
 * 2 = bf16x6 forward with the backward chain (dX = W^T dZ) on three products of two-bf16 operands — a measured experiment, narrower than fp32.
 * k_adam maintains the 16-bit weight streams of the mode in force only; a switch between the stream families re-emits them (synchronises the stream).
 * (Python mirror: AF_EXPERIMENT=1 AF_MLP_MODE=<m> maps onto this call; AF_MLP_FP32=1 selects 0.)
 * pre_train_mapping's MLP chains always run the fp32 16-row kernels (mlp16.hip); its weight-gradient GEMM follows af_set_dw_mode. */
int af_set_mlp_mode(af_handle* h, int mode);
/* The arithmetic modes in force (either pointer may be NULL): what the host side records next to its results. */
int af_get_modes(const af_handle* h, int* mlp_mode, int* dw_mode);
/* After af_train_steps / af_pretrain with debug enabled: reduced gradient of the last step, flat order. */
int af_set_debug(af_handle* h, int enable);
int af_get_last_grads(af_handle* h, int net, float* flat, size_t n);
/* Time the most recent launches: accumulated HIP-event milliseconds, launch counts and algorithmic FLOPs per
 * launch class since the last reset: [0]=prep [1]=fwd_1 [2]=fwd_2 [3]=loss [4]=bwd_1 [5]=bwd_2 [6]=dw [7]=adam.
 * A step has two forward and two backward MLP launches: fwd_1 = the mapping batch's whole rounds (two_layer:
 * alpha + both mappings), fwd_2 = atlas + the remainder; bwd_1 = atlas + remainder, bwd_2 = the rest.
 * ms16[16], counts16[16], flops16[16] (any may be NULL).
 * Bits 16..23 of class_mask: sample period P (0 or 1 = every step) - only the launches of every P-th step of an af_train_steps call
 * (its first step, its (P+1)-th, ...) carry events: a HIP event costs ~5 us in-stream, so timing two launches of every step of a
 * 1.1 ms step slows it by ~1.8 %; counts16 / flops16 cover exactly the launches that were timed. */
int af_set_timing(af_handle* h, int class_mask);   /* bit i (0..15) enables HIP-event timing of launch class i */
int af_get_timing(af_handle* h, double* ms16, int64_t* counts16, double* flops16, int reset);
/* Algorithmic work of ONE train step at the given iteration: MLP rows per net (indexed by af_net) and the
 * fwd+bwd FLOPs of the step. */
int af_step_work(const af_handle* h, int iter, int64_t rows4[4], double* flops);

#ifdef __cplusplus
}
#endif
#endif
