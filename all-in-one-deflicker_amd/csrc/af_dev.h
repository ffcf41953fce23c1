// af_dev.h — shared host/device definitions for the atlas-fit hot path (gfx950 / CDNA4 only).
//
// Layout vocabulary used by every kernel in this directory
// --------------------------------------------------------
// * row            one evaluation point of a coordinate MLP (a sampled (x,y,t) pixel or one of its
//                  neighbours).  Rows are grouped in ROW TILES of 32 (the N dimension of
//                  v_mfma_f32_32x32x2_f32).  One wavefront owns one row tile.
// * C-layout       how a wave holds a 32-row x 256-feature activation block in registers: lane (j,h)
//                  (j = lane&31 = row in tile, h = lane>>5) register rho = 16*T + 4*q + p holds feature
//                  f = 32*T + 8*q + 4*h + p.  This is exactly the C/D fragment layout of the MFMA when the
//                  layer is evaluated transposed (Y^T = W * X^T), and — because the K order of a dot product
//                  is free — it is also a legal B-operand layout for the NEXT layer.  Activations therefore
//                  never leave registers between layers.
// * k-group g      8 consecutive reduction indices; lane-half h supplies indices 8g+4h+p, p=0..3, to four
//                  consecutive MFMA steps.
// * packed image   weights of one layer, ordered so that the A operand of those four steps is one 16-byte
//                  LDS read: slot(g,h,m) = (2g+h)*Mpad + m holds A[m][8g+4h .. 8g+4h+3].
// * T-layout       activations / gradients in HBM for the dW GEMM: per row tile, [feature][32 rows] floats.
//
// Reference: the networks are IMLP (src/models/stage_1/implicit_neural_networks.py:15-80).
#pragma once
#include <stdint.h>

#define AF_HID        256           // hidden width (config number_of_channels_* — only 256 is built)
#define AF_TROWS      32            // rows per tile
#define AF_TILE_F     (AF_HID * AF_TROWS)       // floats per 256-feature T-layout tile
#define AF_CHUNK_MAX  65536         // bytes per LDS weight buffer
#define AF_MAX_LAYERS 8
// kernel kind of a net = its af_net id, except a mapping net WITH positional encoding (use_positional_encoding_mapping*): the alpha
// net's input stage (PE 3 -> 6K, K <= 5 frequencies in the 5-frequency slot layout, unused slots meet zero weights) with 2 outputs
#define AF_KIND_MAP_PE 4
#define AF_MAX_NETS   4
#define AF_REC_F      16            // floats per pixel record (64 B)
#define AF_LOSS_W     16            // floats per loss record: 14 partial sums + #valid fwd + #valid bwd

enum { AF_NET_MAP1 = 0, AF_NET_ATLAS = 1, AF_NET_MAP2 = 2, AF_NET_ALPHA = 3 };
enum { AF_IN_XYT = 0, AF_IN_PE2 = 1, AF_IN_PE3 = 2 };

// pixel record field offsets (floats)
enum { REC_RGB = 0, REC_DX = 3, REC_DY = 6, REC_FF = 9, REC_FB = 11, REC_MF = 13, REC_MB = 14, REC_FG = 15 };

// dW job shapes: (out tiles, in tiles) of the layer block; per-wave split in dw.hip
enum { DW_8x8 = 0, DW_8x2 = 1, DW_8x1 = 2, DW_1x8 = 3, DW_1x2 = 4 };

struct AfChunk { uint32_t off; uint32_t bytes; };   // host-side plan entry: byte offset into a net's image, size (multiple of 4096)

struct FwdArgs {
  const float* wimg;          // forward packed image (all layers of this net, contiguous in stream order)
  const float* bias;          // [NL][256] padded biases
  const float* in;            // [rows_pad][4]  xyt coords, or uv (PE nets use .x,.y[,.z])
  const float* in1;           // rows >= split_row read in1[row - split_row] (second mapping net of the fg/bg path)
  float* out;                 // [rows_pad][4]  tanh outputs
  float* acts;                // X_1..X_{NL-1}: [NL-1][NT][256][32]      (train only)
  uint32_t* masks;            // relu bits:     [NL-1][NT][64][4]        (train only)
  float* pe_tile;             // [NT][64][32] PE features in T-layout     (train only, PE nets)
  float in_scale, in_shift0, in_shift1;   // PE nets: x = v*scale + (row < split_row ? shift0 : shift1)
  int split_row;
  int tile0;                  // first row tile of this launch
  int NT;                     // one past the last row tile of this launch
  int nt_stride;              // row tiles per layer plane of acts / masks (the whole batch)
  const int* live_rows;       // null, or: rows [live_base + *live_rows, NT*32) do not exist this iteration (compacted flow matches,
  int live_base;              // written by k_prep); workgroups wholly beyond the last live row tile return at once
  int nl;                     // layers of this net (2..AF_MAX_LAYERS; number_of_layers_* of the config): the layer loops are runtime loops
};

struct BwdArgs {
  const float* wimg;          // backward packed image (W^T per layer)
  const float* out;           // [rows_pad][4] tanh outputs (forward)
  const float* dout;          // [rows_pad][4] dL/d out
  const uint32_t* masks;
  float* dz;                  // dZ_0..dZ_{NL-2}: [NL-1][NT][256][32]
  float* dz_last;             // [NT][32][32]
  const float* pe_tile;       // [NT][64][32]
  float* din0; float* din1;   // dL/d(input uv) accumulation targets ([rows][4]); rows>=split_row go to din1
  float din_scale;
  int split_row;
  int nrows;                  // rows that own an input gradient (pad rows are skipped)
  int tile0;
  int NT;
  int nt_stride;
  const int* live_rows; int live_base;      // as in FwdArgs
  int nl;                     // layers of this net
};

// one launch = up to AF_MAX_NETS independent row-tile ranges ("parts"), see mlp.hip
#define AF_STAMP_WG 4096      // workgroups per launch the stamp buffer holds; workgroups beyond it are not stamped (a launch of the shipped sizes has at most ~1200;
                              // samples_batch 100 000 has ~7 000 single / ~18 000 two-layer)
// wg_stamp (optional, af_debug_step_clocks): [min(gridDim.x, AF_STAMP_WG)][4] = s_memrealtime (100 MHz) and s_memtime (shader clock) at workgroup start, then at its end
struct MultiFwd { int n; int net[AF_MAX_NETS]; int wg_end[AF_MAX_NETS]; FwdArgs a[AF_MAX_NETS]; unsigned long long* wg_stamp; };
struct MultiBwd { int n; int net[AF_MAX_NETS]; int wg_end[AF_MAX_NETS]; BwdArgs a[AF_MAX_NETS]; int nprod; unsigned long long* wg_stamp; };      // nprod: 3 selects the three-product chain (mlpbf.hip)

struct DwJob {
  const float* A; const float* B;     // T-layout tensors (dZ_l and X_l)
  uint32_t a_stride, b_stride;        // floats per row tile
  int shape;                          // DW_*
  uint32_t part_off;                  // float offset of slot 0 in the partial buffer
  uint32_t part_blk;                  // floats per slot
  int live_base;                      // with live_rows: row tiles beyond row live_base + *live_rows hold stale data and are
  const int* live_rows;               // skipped (a segment wholly beyond them stores a zero block); null: every tile is live
};
struct DwSeg { int job, t0, t1, slot; };
#define DW_MAXSEG 16
struct DwArgs {
  const DwJob* jobs; const DwSeg* segs;   // segs: [gridDim.x][DW_MAXSEG], job<0 terminates
  float* partial;
  unsigned long long* wg_clock;           // optional [gridDim.x][2]: s_memrealtime at workgroup start / end (balance diagnostics)
  unsigned long long* wg_stamp;           // optional [gridDim.x][4]: as in MultiFwd (the clock the kernel runs at inside the real step)
};

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
// af_debug_step_clocks: both counters at workgroup start (i = 0) and end (i = 1) - s_memtime ticks over the s_memrealtime span = the
// clock the CU's issue follows while this launch runs INSIDE the step (one scalar compare per workgroup when off)
#define AF_STAMP(m, i) if ((m).wg_stamp && threadIdx.x == 0 && blockIdx.x < AF_STAMP_WG) { (m).wg_stamp[blockIdx.x * 4 + 2 * (i)] = __builtin_amdgcn_s_memrealtime(); (m).wg_stamp[blockIdx.x * 4 + 2 * (i) + 1] = __builtin_amdgcn_s_memtime(); }
typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef float    f32x4  __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4  __attribute__((ext_vector_type(4)));

#define AF_DEV __device__ __forceinline__

AF_DEV __amdgpu_buffer_rsrc_t af_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
// Same, for a wave-uniform base the compiler cannot prove uniform (e.g. carried around a loop): pin the
// descriptor words to SGPRs.  A descriptor left in VGPRs turns EVERY buffer access into a waterfall loop
// (4 v_readfirstlane + compare + saveexec + branch per instruction).
AF_DEV __amdgpu_buffer_rsrc_t af_rsrc_uniform(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
AF_DEV float af_bl32(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
AF_DEV f32x4 af_bl128(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
AF_DEV void af_bs32(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, 0);
}
// the chains' tile stores (activations / gradients written once, read once by k_dw, 1.3 GB per step: nothing to keep in a cache): cache-policy bits
// of the store (gfx940+: bit 0 sc0, bit 1 nt, bit 4 sc1).  Measured per 128-row task in core ticks (tools/hfbench.hip, profiles/r6_hfbench_tile_store_policy.txt):
// nt -2.9 % (mapping forward) / -3.6 % (mapping backward) / -1..2 % (atlas) against the default policy; sc1, sc0 sc1, nt sc1: -1.5 %.
#ifndef AF_TILE_AUX
#define AF_TILE_AUX 2
#endif
AF_DEV void af_bs32_tile(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, AF_TILE_AUX);
}
AF_DEV void af_bs128(f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16
AF_DEV void af_glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
AF_DEV void af_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Stage one weight chunk (bytes multiple of 4096) into an LDS buffer; all 256 threads participate.
AF_DEV void af_stage_chunk(const char* src, uint32_t bytes, char* dst, int tid, int wave) {
  const int nit = (int)(bytes >> 12);
  for (int it = 0; it < nit; ++it)
    af_glds16(src + it * 4096 + tid * 16, dst + it * 4096 + wave * 1024);
}

// ---- PE slot <-> reference feature permutations -------------------------------------------------
// Atlas (in_dim 2, 10 freqs): reference feature e = 4k + {sin x0, sin x1, cos x0, cos x1}
// (implicit_neural_networks.py:9-13); slot == e (lane half h owns frequencies k = 2g+h).
// Alpha (in_dim 3, 5 freqs): e = 6k + {sin x0..2, cos x0..2}; lane half h owns k in {2h, 2h+1}
// plus the sin (h=0) / cos (h=1) triple of k=4.  Register rho = slot>>3*4 + slot&3, h = (slot>>2)&1.
__host__ __device__ inline int af_pe_slot_of_feature(int pe_kind, int e) {
  if (pe_kind != 2) return e;
  int h, rho;
  if (e < 24) { int k = e / 6, c = e % 6; h = k >> 1; rho = (k & 1) * 6 + c; }
  else        { int c = e - 24; h = c / 3; rho = 12 + c % 3; }
  return ((rho >> 2) << 3) + (h << 2) + (rho & 3);
}
// packed-image float index of element (m, k) in an image with row padding mpad
__host__ __device__ inline uint32_t af_img_index(uint32_t mpad, uint32_t m, uint32_t k) {
  return ((((k >> 3) * 2 + ((k >> 2) & 1)) * mpad + m) << 2) + (k & 3);
}
#endif  // __HIPCC__
