// elem.h — argument blocks of the non-GEMM kernels (see elem.hip).
#pragma once
#include "af_dev.h"

struct PackArgs {
  const float* frames;    // (resy, resx, 3, F)      reference layout, unwrap_utils.py:113
  const float* flow_f;    // (resy, resx, 2, F[,1])  frame i -> i+1 at index i
  const float* flow_b;    // (resy, resx, 2, F[,1])  frame j -> j-1 at index j
  const float* mask_f;    // (resy, resx, F[,1])
  const float* mask_b;
  const float* mask_fg;   // (resy, resx, F) or null
  float* table;           // [F*resy*resx][16]
  int resx, resy, F;
  unsigned long long* nvalid;   // [64][2] += records with a valid forward / backward flow match (64 spread counters; what the batch planner expects per sample)
};

// Input builder (load_input_data*, unwrap_utils.py:40-163): bilinear resize with cv2.resize's INTER_LINEAR geometry
// (half-pixel centres, edge clamp, no anti-aliasing; :35,131) from an HWC source into an arbitrarily strided
// destination, and the forward/backward flow-consistency field || f12 + remap(f21, f12) || (:10-23).
struct ResizeArgs {
  const void* src; int src_u8;        // HWC contiguous, float32 or uint8 (uint8 is divided by 255 first, :128)
  int sh, sw, ch;
  float* dst; int dh, dw;
  long long pix_stride, ch_stride, offset;   // dst index = (y*dw + x)*pix_stride + c*ch_stride + offset
  double scale0, scale1;              // per-channel multipliers of channels 0 / 1 (resize_flow, :36-37); 1 for images
};
struct ConsistencyArgs {
  const float* f12; const float* f21; int h, w;     // (h, w, 2) each
  float* out; long long pix_stride, offset;         // out[(y*w + x)*pix_stride + offset] = norm (thresh <= 0) or norm < thresh
  float thresh;
};

struct PrepArgs {
  const float* table;
  const int64_t* inds;    // [N] pixel-frame indices of this iteration, or null -> device Philox
  uint64_t seed; uint32_t iter;
  int N, resx, resy, F;
  float half_main, half_grad, half_frames;   // larger_dim/2, resx/2, F/2
  int d_local, d_global, nseg;
  float* coords;          // [rows_pad][4]   mapping1 rows
  float* x0_tile;         // [NT][32][32]
  float* samples;         // [N][16]
  int* counts;            // [2]
  // fg/bg path (stage1_neural_atlas_seg.py): mapping2 sees the same rows except the global-rigidity
  // neighbours (its own finite-difference distance); the alpha net sees segments 0,1,2 and, behind them, the compacted flow matches.
  float* coords2;         // null in the single-atlas path
  float* x0_tile2;
  float* coordsA;         // [5N pad][4]
  int d_global2;
  // compaction of the flow-match rows (loss_utils.py:328-335 evaluates the mapping nets on the VALID matches only): the
  // matches of the batch are ranked sample-major (fwd before bwd) by a decoupled look-back scan over the blocks of this
  // launch and written behind the fixed segments, row flow_base + rank (alpha net: 3N + rank).
  int* flow_rank;         // [N][2] rank of the sample's fwd / bwd match among the valid matches, -1 when invalid
  unsigned long long* scan;   // [gridDim.x] (epoch << 32 | matches of the block), indexed by the block's TICKET
  uint32_t epoch;         // unique per launch: a stale entry of an earlier launch can never satisfy the look-back
  unsigned long long* ticket; // [1] monotonic over the handle's k_prep launches: a block's virtual id = its ticket - ticket_base
  unsigned long long ticket_base;   // blocks launched by all earlier k_prep launches of the handle
  int* live;              // [1] valid matches of the batch = live rows behind flow_base
};

struct LossArgs {
  const float* samples; const float* out_map; const float* out_atlas;
  float* dout_map; float* dout_atlas;
  const int* counts;
  const int* flow_rank; const int* live;      // see PrepArgs: rows of the valid flow matches, and how many there are
  float* loss_part;       // [nblocks][AF_LOSS_W]: rgb, gradient, rigidity, global rigidity, flow fwd, flow bwd (sums)
  int N, nseg;
  float L, uv_scale; int d_local, d_global;
  float c_rgb, c_grad, c_rig, c_grig, c_flow;
};

// Loss stack of the fg/bg path (stage1_neural_atlas_seg.py:220-311).  Row layouts: mapping nets as in PrepArgs
// (fixed segment s, sample n -> row s*N+n; flow matches behind them by rank, see k_prep); alpha net rows {centre, (x,y+1), (x+1,y)} + the
// same ranked flow matches behind 3N;
// atlas rows {m1 centre, m1 (x,y+1), m1 (x+1,y), m2 centre, m2 (x,y+1), m2 (x+1,y)}.
// loss_part sums: 0 rgb, 1 gradient, 2 rigidity1, 3 rigidity2, 4 global rigidity1, 5 global rigidity2,
// 6 flow1 fwd, 7 flow1 bwd, 8 flow2 fwd, 9 flow2 bwd, 10 alpha-flow fwd, 11 alpha-flow bwd, 12 BCE, 13 sparsity.
struct LossSegArgs {
  const float* samples;
  const float* out_m1; const float* out_m2; const float* out_alpha; const float* out_atlas;
  float* dout_m1; float* dout_m2; float* dout_alpha; float* dout_atlas;
  const int* counts; float* loss_part;
  const int* flow_rank; const int* live;
  int N, nseg;
  float L, uv_scale; int d_local, d_global_fg, d_global_bg;
  float c_rgb, c_grad, c_rig, c_grig_fg, c_grig_bg, c_flow, c_boot, c_aflow, c_sparse;
};

struct PrePrepArgs {
  const int64_t* ys; const int64_t* xs;   // host-provided rows/cols or null
  uint64_t seed; uint32_t iter;
  int N, resx, resy;
  float half_main, t;
  float* coords; float* x0_tile;
};
struct PreLossArgs { const float* coords; const float* out_map; float* dout_map; float* loss_part; int N; float uv_scale; };

struct AdamJob {
  uint32_t part_off, part_blk, nslots, pld;
  uint32_t out_real, in_real, out_real_pad;
  uint32_t p_off;       // flat param offset of W[0][col0]
  uint32_t p_ld;        // in_features of the layer (canonical row length)
  uint32_t col0;
  int32_t  b_off;       // flat param offset of the bias, or -1 when another job of the layer owns it
  uint32_t f_off;       // forward image: float offset of this layer's image
  uint32_t f_mpad;
  int32_t  b_img_off;   // backward image (W^T) float offset, or -1
  uint32_t b_mpad;
  uint32_t bias_img_off;// float offset in the padded bias array
  uint32_t pe_kind;     // column -> slot permutation for columns >= hid_cols (0/1 identity, 2 alpha)
  uint32_t hid_cols;    // leading identity-mapped columns of the LAYER (256 for skip layers, 0 for PE first layers, p_ld otherwise)
  // weight streams of the bf16x6 chains (mlpbf.hip): byte offset of the block this job writes, its kind (0 = fp32 block packed like
  // the fp32 images with row padding *_mpad and reduction index k - sf_k0; 1 = 256x256 block as three bf16 images), -1 = none
  int32_t  sf_off; uint32_t sf_kind, sf_mpad, sf_k0;
  int32_t  sb_off; uint32_t sb_kind, sb_mpad;
  // weight streams of the f16x3 chains (mlphf.hip): the same blocks in that stream's plan (fp32 blocks as above; kind 1 = 256x256 block as two fp16 images of 2^12 W)
  int32_t  hf_off, hb_off;
};
struct AdamHyper { float step_size, bc2_sqrt, one_minus_b1, beta2, one_minus_b2, eps; };
struct AdamBufs { float* params; float* m; float* v; float* img_f; float* img_b; float* bias_img; char* sf; char* sb; char* hf; char* hb; };
struct AdamArgs {
  const AdamJob* jobs; const float* partial;
  AdamBufs bufs; AdamHyper hy;
  float* grad_out;           // optional: reduced gradient in flat param order (tests)
  const float* loss_part; float* loss_out; int* counts; int loss_nblk;
  int* nan_flag;             // bit 0: a parameter is not finite, a folded loss term is NaN, or (check_counts) a flow-match set is empty;
                             // bit 1: a hidden-layer weight with |w| >= 8 (the fixed 2^12 scale of the fp16 images covers |w| < 16: mlphf.hip)
  int check_counts;
};
