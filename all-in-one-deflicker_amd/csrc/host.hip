// host.hip — handle, memory plan, dW schedule and the C ABI of libatlasfit.so (include/atlasfit.h).
// The per-iteration loop body of the reference (src/stage1_neural_atlas.py:151-231) becomes eight kernel
// launches on one stream: prep -> mapping fwd -> atlas fwd -> loss -> atlas bwd -> mapping bwd -> dW -> adam.
// The fg/bg dual-atlas loop (src/stage1_neural_atlas_seg.py:191-315) is the same chain over four nets:
// prep -> fwd {mapping1, mapping2, alpha, atlas} -> loss -> bwd {atlas, mapping1, mapping2, alpha} -> dW -> adam.
// No host synchronisation inside the loop; the host only enqueues.
#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/atlasfit.h"
#include "af_dev.h"
#include "elem.h"

extern "C" {
int af_launch_fwd_multi(MultiFwd* m, int train, hipStream_t s);
int af_launch_bwd_multi(MultiBwd* m, hipStream_t s);
int af_mlp_init();
int af_launch_fwd16(int net, const FwdArgs* a, hipStream_t s);
int af_launch_bwd16(int net, const BwdArgs* a, hipStream_t s);
int af_mlp16_init();
int af_mlp_chunk_bytes(int net, int which, int nl);
int af_launch_fwd_multi_bf(MultiFwd* m, int train, hipStream_t s);
int af_launch_bwd_multi_bf(MultiBwd* m, hipStream_t s);
int af_mlp_bf_init();
int af_mlp_chunk_bytes_bf(int net, int which, int nl);
int af_launch_fwd_multi_hf(MultiFwd* m, int train, hipStream_t s);
int af_launch_bwd_multi_hf(MultiBwd* m, hipStream_t s);
int af_mlp_hf_init();
int af_mlp_chunk_bytes_hf(int net, int which, int nl);
int af_launch_dw(const DwArgs* a, int nwg, int mode, hipStream_t s);
int af_dw_init();
int af_launch_pack(const PackArgs* a, hipStream_t s);
int af_launch_prep(const PrepArgs* a, hipStream_t s);
int af_launch_loss_single(const LossArgs* a, hipStream_t s);
int af_launch_loss_seg(const LossSegArgs* a, hipStream_t s);
int af_launch_resize(const ResizeArgs* a, hipStream_t s);
int af_launch_consistency(const ConsistencyArgs* a, hipStream_t s);
int af_launch_frame_finish_seg(const float* out_atlas, const float* out_alpha, size_t row2, const float* table, float* rgb_out, double* sse_part,
                               int npix, size_t rec0, hipStream_t s);
int af_launch_pre_prep(const PrePrepArgs* a, hipStream_t s);
int af_launch_pre_loss(const PreLossArgs* a, hipStream_t s);
int af_launch_adam(const AdamArgs* a, int njobs, int update, hipStream_t s);
int af_launch_frame_coords(float* coords, int resx, int resy, float half_main, float t, int npix_pad, hipStream_t s);
int af_launch_frame_finish(const float* out_atlas, const float* table, float* rgb_out, double* sse_part, int npix, size_t rec0, hipStream_t s);
}

namespace {

thread_local std::string g_create_error;     // handle-less errors (af_create, the input-builder utilities): per calling thread

struct NetDesc {
  int kern = -1;                                        // kernel kind (af_dev.h): the net id, or AF_KIND_MAP_PE for a mapping net with positional encoding
  int id = -1, NL = 0, in_kind = 0, in_feat0 = 0, out = 0, pe_feats = 0, pe_kind = 0;
  unsigned skip = 0; bool dx0 = false; bool used = false;
  int in_feat[AF_MAX_LAYERS], out_feat[AF_MAX_LAYERS];
  size_t w_off[AF_MAX_LAYERS], b_off[AF_MAX_LAYERS];   // within the net's flat params
  size_t nparams = 0, p_base = 0;                       // p_base: offset in the global flat buffer
  // Hidden width of the net as configured (number_of_channels_*).  The chains, tiles and split-K jobs are built around AF_HID = 256 units;
  // a narrower net runs EXACTLY inside them: units hid..255 carry zero weights and biases, so they output relu(0) = 0, receive the gradient
  // W^T dZ = 0, produce dW rows / columns of exact zeros, and Adam moves a zero-gradient, zero-moment parameter by lr * 0 / (0 + eps) = 0 —
  // they stay zero for ever.  Only the ABI's flat parameter order differs: lmap[i] = physical index of logical parameter i (state_dict
  // order of the hid-wide IMLP, implicit_neural_networks.py:43-51); empty when hid == AF_HID.
  int hid = AF_HID; size_t nlogical = 0; std::vector<uint32_t> lmap;
  // images (float offsets into the global image buffers)
  size_t f_off[AF_MAX_LAYERS]; int f_mpad[AF_MAX_LAYERS], f_groups[AF_MAX_LAYERS];
  long long b_off_img[AF_MAX_LAYERS]; int b_mpad[AF_MAX_LAYERS];
  size_t f_base = 0, b_base = 0, bias_base = 0;        // float offsets of this net's region (chunk offsets are relative to these)
  std::vector<AfChunk> fchunks, bchunks;               // the planned chunk sequences (checked against the kernels' ChunkBytes)
  // weight streams of the bf16x6 chains (mlpbf.hip): byte offsets of this net's stream in the stream buffers and, per layer,
  // of its 256x256 bf16x3 block (-1: none) and of its fp32 block (layer 0 / skip columns / output layer; -1: none)
  size_t sf_base = 0, sb_base = 0;
  long long sf_hid[AF_MAX_LAYERS], sf_fp[AF_MAX_LAYERS], sb_hid[AF_MAX_LAYERS], sb_fp[AF_MAX_LAYERS];
  // the same for the f16x3 chains (mlphf.hip): four 64 KB chunks of two fp16 images per 256x256 block
  size_t hf_base = 0, hb_base = 0;
  long long hf_hid[AF_MAX_LAYERS], hf_fp[AF_MAX_LAYERS], hb_hid[AF_MAX_LAYERS], hb_fp[AF_MAX_LAYERS];
  // activations
  int nt_cap = 0;
  float *coords = nullptr, *x0_tile = nullptr;         // input rows [rows_pad][4] (+ T-layout copy for the layer-0 dW of xyt nets)
  float *acts = nullptr, *dz = nullptr, *dz_last = nullptr, *pe_tile = nullptr, *out_buf = nullptr, *dout = nullptr;
  uint32_t* masks = nullptr;
};

struct Sched {
  std::vector<DwJob> jobs; std::vector<AdamJob> ajobs; std::vector<DwSeg> segs;
  int nwg = 0; size_t partial_floats = 0;
  DwJob* d_jobs = nullptr; AdamJob* d_ajobs = nullptr; DwSeg* d_segs = nullptr;
};

struct TimedEv { int cls; hipEvent_t a, b; };

}  // namespace

struct af_handle {
  af_config cfg;
  bool seg = false;                   // two_layer: four nets (stage1_neural_atlas_seg.py), else two
  int device = 0; hipStream_t stream = nullptr; int ncu = 256;
  std::string err;
  NetDesc nets[AF_MAX_NETS];
  size_t total_params = 0, img_f_floats = 0, img_b_floats = 0, bias_floats = 0, sf_bytes = 0, sb_bytes = 0;
  char *img_sf = nullptr, *img_sb = nullptr;   // bf16x6 chain streams
  char *img_hf = nullptr, *img_hb = nullptr; size_t hf_bytes = 0, hb_bytes = 0;   // f16x3 chain streams (mlphf.hip)
  int mlp_mode = 3;                            // MLP chains: 3 = f16x3, the default since round 6: hidden layers on the fp16 matrix pipe, two-term fp16 split with a scale
                                               // per row, three products, both directions (mlphf.hip); 1 = bf16x6 on the bf16 pipe (mlpbf.hip, the default of rounds 2-5);
                                               // 2 = as 1 with the backward chain on three bf16 products (experiment); 0 = fp32 MFMA (mlp.hip)
  float *params = nullptr, *adam_m = nullptr, *adam_v = nullptr, *pre_m = nullptr, *pre_v = nullptr, *grads = nullptr;
  float *img_f = nullptr, *img_b = nullptr, *bias_img = nullptr;
  long long adam_step = 0;
  // video
  float* table = nullptr; bool have_video = false;
  // batch
  int N = 0;
  float *samples = nullptr, *loss_part = nullptr, *loss_log = nullptr;
  int* counts = nullptr; int loss_nblk_cap = 0; size_t loss_log_cap = 0;
  int* nan_flag = nullptr;                    // sticky, set on the device by k_adam: a non-finite parameter, a NaN loss term or an empty flow-match set
  int cur_nseg = 0;
  // compaction of the flow-match rows (k_prep): per-sample ranks, look-back scan slots, live match count, launch epoch
  int* flow_rank = nullptr; unsigned long long* scan = nullptr; int* live = nullptr; uint32_t prep_epoch = 0; unsigned long long prep_tickets = 0;
  unsigned long long* nvalid = nullptr; double p_valid[2] = {1.0, 1.0};     // share of the video's pixels with a valid fwd / bwd match
  int plan_flow_rows = 0;                                                   // flow-match rows the launches and the dW schedule are balanced for
  // schedules: 0 = 9 segments, 1 = 7 segments, 2 = pretrain mapping1, 3 = pretrain mapping2
  Sched sched[4]; float* partial = nullptr; size_t partial_cap = 0;
  unsigned long long* dw_clock = nullptr;     // af_debug_dw_clocks: per-workgroup start/end times of the last k_dw launch
  unsigned long long* step_stamp = nullptr;   // af_debug_step_clocks: [5 launches: fwd_1, fwd_2, bwd_1, bwd_2, dw][AF_STAMP_WG][4] of the last step
  // render
  int render_rows_cap = 0; float *r_coords = nullptr, *r_uv = nullptr, *r_uv2 = nullptr, *r_al = nullptr, *r_t = nullptr, *r_rgb = nullptr; double* r_sse = nullptr;
  std::vector<double> frame_sse; std::vector<char> frame_sse_valid;
  bool debug = false; unsigned timing = 0, timing_every = 1; bool timing_live = true;   // timing_every: af_set_timing's sample period; timing_live: this step of af_train_steps is a sampled one
  double flop_fwd[AF_MAX_NETS] = {0}, flop_dx[AF_MAX_NETS] = {0};     // algorithmic FLOPs per MLP row of each net as built (forward == dW; dX chain), BASELINE.md 3
  bool dw_cost_set = false; double dw_cost[5] = {0}, dw_seg_cost = 0;      // af_debug_set_dw_cost: an explicit tile-cost row for build_sched (validated there: finite, > 0)
  int dw_mode = 1;                            // k_dw arithmetic (dw.hip): 1 = bf16x6 (fp32-faithful, the default), 2 = bf16x3 (hi + mid bf16 per operand, three products; opt-in), 0 = fp32 MFMA
  std::vector<TimedEv> evs; double t_ms[16] = {0}, t_flops[16] = {0}; long long t_cnt[16] = {0};

  int fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    err = buf;
    return code;
  }
};

#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return h->fail(AF_EHIP, #x, e_); } while (0)
#define LCHK(x) do { int r_ = (x); if (r_ != 0) return h->fail(AF_EHIP, #x, (hipError_t)r_); } while (0)

namespace {

// timing classes (include/atlasfit.h: af_get_timing): the two forward and the two backward launches of a step
enum { T_PREP = 0, T_FWD_1 = 1, T_FWD_2 = 2, T_LOSS = 3, T_BWD_1 = 4, T_BWD_2 = 5, T_DW = 6, T_ADAM = 7 };
// algorithmic FLOPs per MLP row (BASELINE.md §3 / SURVEY.md §8d), indexed by af_net: forward (== dW) and dX chain
// (the shipped architecture: forward 526848 / 829168 / 264704 / 802304, dX chain 525312 / 808448 / 263168 / 786944; af_create
// computes them for the configured layer counts: forward = 2 sum_l in_l out_l, dX = 2 (sum_{l>=1} 256 out_l + [atlas] in_0 256))

template <class T> hipError_t dalloc(T** p, size_t n) { return hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)); }

void describe_net(NetDesc& n, int id, int NL, int in_kind, int pe_freqs, int out, unsigned skip, bool dx0, int hid = AF_HID) {
  n.id = id; n.kern = (in_kind == AF_IN_PE3 && out == 2) ? AF_KIND_MAP_PE : id; n.NL = NL; n.in_kind = in_kind; n.out = out; n.skip = skip; n.dx0 = dx0; n.used = true;
  const int in_dim = in_kind == AF_IN_PE2 ? 2 : 3;
  n.pe_feats = in_kind == AF_IN_XYT ? 0 : 2 * in_dim * pe_freqs;
  n.pe_kind = in_kind == AF_IN_PE3 ? 2 : (in_kind == AF_IN_PE2 ? 1 : 0);
  n.in_feat0 = in_kind == AF_IN_XYT ? 3 : n.pe_feats;
  size_t off = 0;
  for (int l = 0; l < NL; ++l) {
    n.in_feat[l] = l == 0 ? n.in_feat0 : (AF_HID + (((skip >> l) & 1) ? n.pe_feats : 0));
    n.out_feat[l] = l == NL - 1 ? out : AF_HID;
    n.w_off[l] = off; off += (size_t)n.in_feat[l] * n.out_feat[l];
    n.b_off[l] = off; off += n.out_feat[l];
  }
  n.nparams = off;
  n.hid = hid; n.nlogical = off; n.lmap.clear();
  if (hid != AF_HID) {      // logical (hid-wide) state_dict order -> physical (AF_HID-wide) index
    for (int l = 0; l < NL; ++l) {
      const int pin = n.in_feat[l], pout = n.out_feat[l];
      const int lout = l == NL - 1 ? out : hid;
      const int extra = l == 0 ? n.in_feat0 : (pin - AF_HID);                 // layer 0: every input column; later: the skip-concat's PE columns behind the hidden ones
      const int lhid = l == 0 ? 0 : hid;
      for (int r = 0; r < lout; ++r) {
        for (int c = 0; c < lhid; ++c) n.lmap.push_back((uint32_t)(n.w_off[l] + (size_t)r * pin + c));
        for (int c = 0; c < extra; ++c) n.lmap.push_back((uint32_t)(n.w_off[l] + (size_t)r * pin + (l == 0 ? 0 : AF_HID) + c));
      }
      (void)pout;
      for (int r = 0; r < lout; ++r) n.lmap.push_back((uint32_t)(n.b_off[l] + r));
    }
    n.nlogical = n.lmap.size();
  }
}

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Lay out the forward and backward packed images of one net and their chunk tables.
void plan_images(NetDesc& n, size_t& f_cursor, size_t& b_cursor, size_t& bias_cursor) {
  n.f_base = f_cursor; n.b_base = b_cursor; n.bias_base = bias_cursor;
  bias_cursor += (size_t)n.NL * AF_HID;
  const int peg = n.in_kind == AF_IN_PE3 ? 4 : (n.in_kind == AF_IN_PE2 ? 5 : 0);     // k-groups of 8 PE slots the kernels walk: the encodings always have their shipped slot layout (3-D: five frequencies, 2-D: ten); fewer frequencies leave slots with zero weights
  size_t foff = 0;   // bytes relative to f_base
  for (int l = 0; l < n.NL; ++l) {
    const bool last = l == n.NL - 1;
    const int mpad = last ? 4 : AF_HID;            // the output layer runs on 4x4x1 MFMA blocks (mlp.hip)
    const int groups = (l == 0) ? (n.in_kind == AF_IN_XYT ? 1 : peg) : (32 + (((n.skip >> l) & 1) ? peg : 0));
    n.f_mpad[l] = mpad; n.f_groups[l] = groups;
    n.f_off[l] = n.f_base + foff / 4;
    const size_t gbytes = (size_t)2 * mpad * 16;
    if (l == 0 || last) {
      const size_t bytes = round_up(groups * gbytes, 4096);
      n.fchunks.push_back({(uint32_t)foff, (uint32_t)bytes});
      foff += bytes;
    } else {
      for (int c = 0; c < 4; ++c) { n.fchunks.push_back({(uint32_t)foff, (uint32_t)(8 * gbytes)}); foff += 8 * gbytes; }
      if ((n.skip >> l) & 1) { const size_t bytes = round_up(peg * gbytes, 4096); n.fchunks.push_back({(uint32_t)foff, (uint32_t)bytes}); foff += bytes; }
    }
  }
  f_cursor += foff / 4;
  size_t boff = 0;
  for (int l = 0; l < n.NL; ++l) { n.b_off_img[l] = -1; n.b_mpad[l] = 0; }
  for (int l = n.NL - 1; l >= (n.dx0 ? 0 : 1); --l) {
    const int mpad = l == 0 ? 64 : AF_HID;
    const int groups = l == n.NL - 1 ? 1 : 32;
    n.b_mpad[l] = mpad; n.b_off_img[l] = (long long)(n.b_base + boff / 4);
    const size_t gbytes = (size_t)2 * mpad * 16;
    if (groups * gbytes <= AF_CHUNK_MAX) { n.bchunks.push_back({(uint32_t)boff, (uint32_t)(groups * gbytes)}); boff += groups * gbytes; }
    else for (int c = 0; c < 4; ++c) { n.bchunks.push_back({(uint32_t)boff, (uint32_t)(8 * gbytes)}); boff += 8 * gbytes; }
  }
  b_cursor += boff / 4;
}

// Streams of the bf16x6 chains: fp32 blocks for layer 0, the skip columns and the output layers, eight 48 KB bf16x3 chunks
// per 256x256 hidden product, in consumption order (mlpbf.hip).  Returns false if the sizes disagree with the kernels'.
bool plan_streams_bf(NetDesc& n, size_t& f_cursor, size_t& b_cursor) {
  const int peg = n.in_kind == AF_IN_PE3 ? 4 : (n.in_kind == AF_IN_PE2 ? 5 : 0);     // k-groups of 8 PE slots the kernels walk: the encodings always have their shipped slot layout (3-D: five frequencies, 2-D: ten); fewer frequencies leave slots with zero weights
  const int cb_l0 = af_mlp_chunk_bytes_bf(n.kern, 0, n.NL), cb_hid = af_mlp_chunk_bytes_bf(n.kern, 1, n.NL), cb_skip = af_mlp_chunk_bytes_bf(n.kern, 2, n.NL);
  const int cb_last = af_mlp_chunk_bytes_bf(n.kern, 3, n.NL), cb_blast = af_mlp_chunk_bytes_bf(n.kern, 4, n.NL), cb_bl0h = af_mlp_chunk_bytes_bf(n.kern, 5, n.NL);
  for (int l = 0; l < AF_MAX_LAYERS; ++l) n.sf_hid[l] = n.sf_fp[l] = n.sb_hid[l] = n.sb_fp[l] = -1;
  n.sf_base = f_cursor; n.sb_base = b_cursor;
  size_t off = 0;
  n.sf_fp[0] = (long long)off; off += cb_l0;
  if ((size_t)cb_l0 < round_up((size_t)(n.in_kind == AF_IN_XYT ? 1 : peg) * 2 * AF_HID * 16, 4096)) return false;
  for (int l = 1; l < n.NL - 1; ++l) {
    n.sf_hid[l] = (long long)off; off += (size_t)8 * cb_hid;
    if ((n.skip >> l) & 1) { n.sf_fp[l] = (long long)off; off += cb_skip; }
  }
  n.sf_fp[n.NL - 1] = (long long)off; off += cb_last;
  f_cursor += off;
  off = 0;
  n.sb_fp[n.NL - 1] = (long long)off; off += cb_blast;
  for (int l = n.NL - 2; l >= 1; --l) { n.sb_hid[l] = (long long)off; off += (size_t)8 * cb_hid; }
  if (n.dx0) { n.sb_fp[0] = (long long)off; off += (size_t)2 * cb_bl0h; }
  b_cursor += off;
  return cb_hid == 49152 && cb_blast == 2 * AF_HID * 16 && 2 * cb_bl0h == 32 * 2 * 64 * 16;
}

// Streams of the f16x3 chains (mlphf.hip): the fp32 blocks as above, four 64 KB chunks (two fp16 images of four k-steps) per 256x256 product.
bool plan_streams_hf(NetDesc& n, size_t& f_cursor, size_t& b_cursor) {
  const int peg = n.in_kind == AF_IN_PE3 ? 4 : (n.in_kind == AF_IN_PE2 ? 5 : 0);
  const int cb_l0 = af_mlp_chunk_bytes_hf(n.kern, 0, n.NL), cb_hid = af_mlp_chunk_bytes_hf(n.kern, 1, n.NL), cb_skip = af_mlp_chunk_bytes_hf(n.kern, 2, n.NL);
  const int cb_last = af_mlp_chunk_bytes_hf(n.kern, 3, n.NL), cb_blast = af_mlp_chunk_bytes_hf(n.kern, 4, n.NL), cb_bl0h = af_mlp_chunk_bytes_hf(n.kern, 5, n.NL);
  for (int l = 0; l < AF_MAX_LAYERS; ++l) n.hf_hid[l] = n.hf_fp[l] = n.hb_hid[l] = n.hb_fp[l] = -1;
  n.hf_base = f_cursor; n.hb_base = b_cursor;
  size_t off = 0;
  n.hf_fp[0] = (long long)off; off += cb_l0;
  if ((size_t)cb_l0 < round_up((size_t)(n.in_kind == AF_IN_XYT ? 1 : peg) * 2 * AF_HID * 16, 4096)) return false;
  for (int l = 1; l < n.NL - 1; ++l) {
    n.hf_hid[l] = (long long)off; off += (size_t)4 * cb_hid;
    if ((n.skip >> l) & 1) { n.hf_fp[l] = (long long)off; off += cb_skip; }
  }
  n.hf_fp[n.NL - 1] = (long long)off; off += cb_last;
  f_cursor += off;
  off = 0;
  n.hb_fp[n.NL - 1] = (long long)off; off += cb_blast;
  for (int l = n.NL - 2; l >= 1; --l) { n.hb_hid[l] = (long long)off; off += (size_t)4 * cb_hid; }
  if (n.dx0) { n.hb_fp[0] = (long long)off; off += (size_t)2 * cb_bl0h; }
  b_cursor += off;
  return cb_hid == 65536 && cb_blast == 2 * AF_HID * 16 && 2 * cb_bl0h == 32 * 2 * 64 * 16;
}

// The kernels walk the weight stream with compile-time chunk sizes (mlp.hip ChunkBytes): the planned layout must
// be exactly that sequence, contiguous.
bool check_chunk_plan(const NetDesc& n) {
  std::vector<int> f, b;
  f.push_back(af_mlp_chunk_bytes(n.kern, 0, n.NL));
  for (int l = 1; l < n.NL - 1; ++l) { for (int c = 0; c < 4; ++c) f.push_back(af_mlp_chunk_bytes(n.kern, 1, n.NL)); if ((n.skip >> l) & 1) f.push_back(af_mlp_chunk_bytes(n.kern, 2, n.NL)); }
  f.push_back(af_mlp_chunk_bytes(n.kern, 3, n.NL));
  b.push_back(af_mlp_chunk_bytes(n.kern, 4, n.NL));
  for (int l = n.NL - 2; l >= 1; --l) for (int c = 0; c < 4; ++c) b.push_back(af_mlp_chunk_bytes(n.kern, 1, n.NL));
  if (n.dx0) b.push_back(af_mlp_chunk_bytes(n.kern, 5, n.NL));
  auto same = [](const std::vector<AfChunk>& plan, const std::vector<int>& want) {
    if (plan.size() != want.size()) return false;
    uint32_t off = 0;
    for (size_t i = 0; i < plan.size(); ++i) { if (plan[i].off != off || (int)plan[i].bytes != want[i]) return false; off += plan[i].bytes; }
    return true;
  };
  return same(n.fchunks, f) && same(n.bchunks, b);
}

int shape_tiles(int shape, int& To, int& Ti) {
  switch (shape) {
    case DW_8x8: To = 8; Ti = 8; return 256;
    case DW_8x2: To = 8; Ti = 2; return 64;
    case DW_8x1: To = 8; Ti = 1; return 32;
    case DW_1x8: To = 1; Ti = 8; return 32;
    default:     To = 1; Ti = 2; return 16;
  }
}

struct NetUse { NetDesc* n; int NT; int live_base = -1; int NT_plan = -1; };     // live_base >= 0: rows behind live_base + *h->live are dead (compacted flow matches);
                                                                          // NT_plan: the row tiles expected to be live (the schedule is balanced for them, NT is the capacity)

// Build the dW job list, the Adam job list and the cost-balanced split-K schedule for a set of nets.
bool build_sched(af_handle* h, Sched& sc, const std::vector<NetUse>& uses) {
  sc.jobs.clear(); sc.ajobs.clear(); sc.segs.clear();
  std::vector<int> job_nt, job_cap;       // row tiles the cut is balanced for / the job's last segment reaches (>= job_nt with compaction)
  int cur_live_base = -1, cur_cap = 0;
  auto add = [&](NetDesc& n, int NT, int l, int shape, const float* A, uint32_t as, const float* B, uint32_t bs,
                 int col0, int in_real, bool owns_bias) {
    int To, Ti; shape_tiles(shape, To, Ti);
    DwJob j{}; j.A = A; j.B = B; j.a_stride = as; j.b_stride = bs; j.shape = shape;
    if (cur_live_base >= 0) { j.live_base = cur_live_base; j.live_rows = h->live; }
    j.part_blk = (uint32_t)(To * 32 * Ti * 32 + To * 32);
    sc.jobs.push_back(j); job_nt.push_back(NT); job_cap.push_back(cur_cap);
    AdamJob a{};
    a.part_blk = j.part_blk; a.pld = Ti * 32; a.out_real = n.out_feat[l]; a.in_real = in_real; a.out_real_pad = To * 32;
    a.p_off = (uint32_t)(n.p_base + n.w_off[l] + col0); a.p_ld = n.in_feat[l]; a.col0 = col0;
    a.b_off = owns_bias ? (int32_t)(n.p_base + n.b_off[l]) : -1;
    a.f_off = (uint32_t)n.f_off[l]; a.f_mpad = n.f_mpad[l];
    a.b_img_off = (int32_t)n.b_off_img[l]; a.b_mpad = n.b_mpad[l];
    a.bias_img_off = (uint32_t)(n.bias_base + (size_t)l * AF_HID);
    a.pe_kind = n.pe_kind;
    a.hid_cols = l == 0 ? (n.in_kind == AF_IN_XYT ? n.in_feat[l] : 0) : AF_HID;
    // bf16x6 chain streams: which block of the net's forward / backward stream this job's weights go to
    const bool last_l = l == n.NL - 1;
    a.sf_k0 = 0; a.sf_mpad = last_l ? 4 : AF_HID; a.sb_mpad = l == 0 ? 64 : AF_HID;
    if (l == 0 || last_l)  { a.sf_kind = 0; a.sf_off = (int32_t)(n.sf_base + n.sf_fp[l]); }
    else if (col0 == 0)    { a.sf_kind = 1; a.sf_off = (int32_t)(n.sf_base + n.sf_hid[l]); }
    else                   { a.sf_kind = 0; a.sf_off = (int32_t)(n.sf_base + n.sf_fp[l]); a.sf_k0 = AF_HID; }
    if (n.b_off_img[l] < 0)      { a.sb_kind = 0; a.sb_off = -1; }
    else if (last_l || l == 0)   { a.sb_kind = 0; a.sb_off = (int32_t)(n.sb_base + n.sb_fp[l]); }
    else                         { a.sb_kind = 1; a.sb_off = (int32_t)(n.sb_base + n.sb_hid[l]); }
    // f16x3 chain streams: the same blocks (same kinds) at that plan's offsets
    a.hf_off = (int32_t)(n.hf_base + (a.sf_kind == 1 ? n.hf_hid[l] : n.hf_fp[l]));
    a.hb_off = a.sb_off < 0 ? -1 : (int32_t)(n.hb_base + (a.sb_kind == 1 ? n.hb_hid[l] : n.hb_fp[l]));
    sc.ajobs.push_back(a);
  };
  for (const NetUse& u : uses) {
    NetDesc& n = *u.n;
    const int NT = u.NT_plan >= 0 ? std::min(u.NT_plan, u.NT) : u.NT;      // the tensors' layer planes are u.NT tiles apart whatever is live
    cur_live_base = u.live_base; cur_cap = u.NT;
    const size_t ts = (size_t)u.NT * AF_TILE_F;
    for (int l = 0; l < n.NL; ++l) {
      const bool last = l == n.NL - 1, sk = (n.skip >> l) & 1;
      if (l == 0) {
        if (n.in_kind == AF_IN_XYT) add(n, NT, 0, DW_8x1, n.dz, AF_TILE_F, n.x0_tile, 1024, 0, 3, true);
        else                        add(n, NT, 0, DW_8x2, n.dz, AF_TILE_F, n.pe_tile, 2048, 0, n.pe_feats, true);
      } else if (!last) {
        add(n, NT, l, DW_8x8, n.dz + l * ts, AF_TILE_F, n.acts + (l - 1) * ts, AF_TILE_F, 0, AF_HID, true);
        if (sk) add(n, NT, l, DW_8x2, n.dz + l * ts, AF_TILE_F, n.pe_tile, 2048, AF_HID, n.pe_feats, false);
      } else {
        add(n, NT, l, DW_1x8, n.dz_last, 1024, n.acts + (l - 1) * ts, AF_TILE_F, 0, AF_HID, true);
        if (sk) add(n, NT, l, DW_1x2, n.dz_last, 1024, n.pe_tile, 2048, AF_HID, n.pe_feats, false);
      }
    }
  }
  const int nj = (int)sc.jobs.size();
  // Cost of one 32-row tile of each job shape, in units of 1/306 of an 8x8 tile — measured per-workgroup busy times of a launch
  // (s_memrealtime) against the segment lists, tools/dw_fit.py / tools/dw_dump.py.  Only the 8x8 jobs are matrix-pipe work; the
  // narrow shapes (layer 0, skip columns, output layer) are bound by the latency of their 36-40 KB operand tiles, so they do not
  // get cheaper when the 8x8 tiles move to the bf16 matrix pipe:
  //   fp32 MFMA k_dw:  8x8 7.9 us/tile, 8x2 2.4, 8x1 1.6, 1x8 1.6, 1x2 1.2      bf16x6 k_dw_bf:  8x8 5.47, 8x2 2.0, 8x1 1.44, 1x8 1.44, 1x2 0.9
  // Row 1 re-fitted in round 4 for the slotted 8x8 stage (dw.hip DW_SLOT: the 8x8 tile got 13 % cheaper in ticks, the narrow shapes did not):
  // non-negative least squares over 1024 workgroups (tools/dw_fit.py --save, gpurun_out/r4f_dwfit_sys.npz): 8x8 4.0-4.3 us per tile, ratios
  // 306 : 158 : 133 : 136 : 92 on the single- and the two-layer schedules alike; with the stale row the slowest workgroup ran 15 % over the mean.
  // A sweep on the real step (tools/dw_cost_sweep.sh, gpurun_out/r4e/r4f/r4i_sweep.txt) prefers the narrow shapes another 5 % dearer: per-workgroup
  // busy mean / max 0.95 (9 segments) and 0.92 (7 segments).
  static const double kTileCost[3][5] = {{306.0, 94.0, 62.0, 62.0, 46.0}, {306.0, 166.0, 140.0, 143.0, 97.0}, {306.0, 168.0, 157.0, 150.0, 95.0}};     // row 2 fitted in round 3 (gpurun_out/r3g_dwfit_m2.txt)
  double seg_cost = 60.0;
  double cost_row[5];
  for (int i = 0; i < 5; ++i) cost_row[i] = kTileCost[h->dw_mode][i];
  if (h->dw_cost_set) {     // af_debug_set_dw_cost (experiments and the partition-sensitivity tests): replaces the row of the current arithmetic
    for (int i = 0; i < 5; ++i) cost_row[i] = h->dw_cost[i];
    if (h->dw_seg_cost > 0) seg_cost = h->dw_seg_cost;
  }
  auto tile_cost = [&](int j) { return cost_row[sc.jobs[j].shape]; };
  double work = 0;
  for (int j = 0; j < nj; ++j) work += tile_cost(j) * job_nt[j];
  int nwg = (int)std::min<double>(h->ncu, std::max(1.0, work / (4.0 * 276.0)));     // at least ~4 full-size row tiles per workgroup
  // Cut the job sequence (largest jobs first) into nwg pieces of equal cost.  Boundaries are placed on the
  // CUMULATIVE cost line (rounding to the nearest tile), so rounding never accumulates onto the last workgroup.
  std::vector<int> order(nj);
  for (int j = 0; j < nj; ++j) order[j] = j;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tile_cost(a) * job_nt[a] > tile_cost(b) * job_nt[b]; });
  std::vector<int> nslots(nj, 0);
  std::vector<std::vector<DwSeg>> wg(nwg);
  const double total = work + seg_cost * (nj + nwg);
  const double target = total / nwg;
  int w = 0; double cum = 0;
  for (int oi = 0; oi < nj; ++oi) {
    const int j = order[oi]; int t = 0; const double tc = tile_cost(j);
    while (t < job_nt[j]) {
      const double bound = (w + 1) * target;
      int take;
      if (w == nwg - 1) take = job_nt[j] - t;
      else {
        take = (int)floor((bound - cum - seg_cost) / tc + 0.5);
        if (take <= 0 || (int)wg[w].size() >= DW_MAXSEG) { ++w; continue; }
        take = std::min(take, job_nt[j] - t);
      }
      wg[w].push_back({j, t, t + take, nslots[j]++});
      cum += seg_cost + tc * take; t += take;
      if (t == job_nt[j]) wg[w].back().t1 = job_cap[j];     // more valid matches than planned: the job's last segment takes them (k_dw clips at the live tiles)
      if (w < nwg - 1 && cum >= bound - 0.5 * tc) ++w;
    }
  }
  sc.nwg = nwg;
  for (int i = 0; i < nwg; ++i) if (wg[i].size() > DW_MAXSEG) return false;
  sc.segs.assign((size_t)nwg * DW_MAXSEG, DwSeg{-1, 0, 0, 0});
  for (int i = 0; i < nwg; ++i)
    for (size_t s = 0; s < wg[i].size() && s < DW_MAXSEG; ++s) sc.segs[(size_t)i * DW_MAXSEG + s] = wg[i][s];
  size_t off = 0;
  for (int j = 0; j < nj; ++j) {
    sc.jobs[j].part_off = (uint32_t)off; sc.ajobs[j].part_off = (uint32_t)off; sc.ajobs[j].nslots = nslots[j];
    off += (size_t)nslots[j] * sc.jobs[j].part_blk;
  }
  sc.partial_floats = off;
  return true;
}

hipError_t upload_sched(Sched& sc) {
  hipError_t e;
  if (sc.d_jobs) { (void)hipFree(sc.d_jobs); (void)hipFree(sc.d_ajobs); (void)hipFree(sc.d_segs); sc.d_jobs = nullptr; }
  if ((e = dalloc(&sc.d_jobs, sc.jobs.size())) != hipSuccess) return e;
  if ((e = dalloc(&sc.d_ajobs, sc.ajobs.size())) != hipSuccess) return e;
  if ((e = dalloc(&sc.d_segs, sc.segs.size())) != hipSuccess) return e;
  if ((e = hipMemcpy(sc.d_jobs, sc.jobs.data(), sc.jobs.size() * sizeof(DwJob), hipMemcpyHostToDevice)) != hipSuccess) return e;
  if ((e = hipMemcpy(sc.d_ajobs, sc.ajobs.data(), sc.ajobs.size() * sizeof(AdamJob), hipMemcpyHostToDevice)) != hipSuccess) return e;
  return hipMemcpy(sc.d_segs, sc.segs.data(), sc.segs.size() * sizeof(DwSeg), hipMemcpyHostToDevice);
}

int tiles_of(int rows) { return (rows + 31) / 32; }

// Flow-match rows a batch is planned for: the expectation of its valid matches (the share of valid pixels of the uploaded
// video, the sampler is uniform) + 4 sigma of the binomial, never more than 2N.  Launch splits and the dW schedule are
// balanced for it; capacity stays 2N, a batch with more valid matches is still computed completely (just less evenly).
int planned_flow_rows(const af_handle* h) {
  const double N = h->N, pf = h->p_valid[0], pb = h->p_valid[1];
  const double e = N * (pf + pb), sd = sqrt(std::max(0.0, N * (pf * (1.0 - pf) + pb * (1.0 - pb))));
  return (int)std::min(2.0 * N, ceil(e + 4.0 * sd) + 32.0);
}

// the two schedules of the loop (0: with the global-rigidity rows, 1: without)
bool build_main_scheds(af_handle* h) {
  NetDesc& M = h->nets[AF_NET_MAP1]; NetDesc& A = h->nets[AF_NET_ATLAS]; NetDesc& M2 = h->nets[AF_NET_MAP2]; NetDesc& AL = h->nets[AF_NET_ALPHA];
  const int N = h->N; const bool seg = h->seg;
  h->plan_flow_rows = planned_flow_rows(h);
  for (int v = 0; v < 2; ++v) {
    const int nseg = v == 0 ? 9 : 7, fb = (nseg - 2) * N;
    std::vector<NetUse> uses = {{&M, tiles_of(nseg * N), fb, tiles_of(fb + h->plan_flow_rows)}};
    if (seg) { uses.push_back({&M2, tiles_of(nseg * N), fb, tiles_of(fb + h->plan_flow_rows)}); uses.push_back({&AL, tiles_of(5 * N), 3 * N, tiles_of(3 * N + h->plan_flow_rows)}); }
    uses.push_back({&A, tiles_of((seg ? 6 : 3) * N)});
    if (!build_sched(h, h->sched[v], uses)) return false;
  }
  return true;
}

hipError_t alloc_net_buffers(NetDesc& n, int rows_cap, bool own_coords) {
  n.nt_cap = tiles_of(rows_cap);
  const size_t nt = n.nt_cap, rp = nt * 32;
  hipError_t e;
  if ((e = dalloc(&n.acts, (size_t)(n.NL - 1) * nt * AF_TILE_F)) != hipSuccess) return e;
  if ((e = dalloc(&n.dz, (size_t)(n.NL - 1) * nt * AF_TILE_F)) != hipSuccess) return e;
  if ((e = dalloc(&n.masks, (size_t)(n.NL - 1) * nt * 64 * 4)) != hipSuccess) return e;
  if ((e = dalloc(&n.dz_last, nt * 1024)) != hipSuccess) return e;
  if ((e = dalloc(&n.pe_tile, n.pe_feats ? nt * 2048 : 1)) != hipSuccess) return e;
  if ((e = dalloc(&n.out_buf, rp * 4)) != hipSuccess) return e;
  if ((e = dalloc(&n.dout, rp * 4)) != hipSuccess) return e;
  if ((e = hipMemset(n.dz_last, 0, nt * 1024 * 4)) != hipSuccess) return e;
  if (n.pe_feats && (e = hipMemset(n.pe_tile, 0, nt * 2048 * 4)) != hipSuccess) return e;
  if ((e = hipMemset(n.out_buf, 0, rp * 16)) != hipSuccess) return e;
  if ((e = hipMemset(n.dout, 0, rp * 16)) != hipSuccess) return e;
  if (own_coords) {
    if ((e = dalloc(&n.coords, rp * 4)) != hipSuccess) return e;
    if ((e = hipMemset(n.coords, 0, rp * 16)) != hipSuccess) return e;
    if (n.in_kind == AF_IN_XYT) {
      if ((e = dalloc(&n.x0_tile, nt * 1024)) != hipSuccess) return e;
      if ((e = hipMemset(n.x0_tile, 0, nt * 4096)) != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

void free_net(NetDesc& n) {
  (void)hipFree(n.acts); (void)hipFree(n.dz); (void)hipFree(n.masks); (void)hipFree(n.dz_last); (void)hipFree(n.pe_tile); (void)hipFree(n.out_buf); (void)hipFree(n.dout);
  (void)hipFree(n.coords); (void)hipFree(n.x0_tile);
}

// bf: the launch goes to the bf16x6 chains (mlpbf.hip), which walk the net's bf16 stream; the fp32 chains (mlp.hip, and always
// the 16-row pre-train chains of mlp16.hip) walk the fp32 images
FwdArgs fwd_args(af_handle* h, NetDesc& n, const float* in, float* out, int NT, bool train, bool bf) {
  FwdArgs a{};
  a.wimg = bf ? (h->mlp_mode == 3 ? (const float*)(h->img_hf + n.hf_base) : (const float*)(h->img_sf + n.sf_base)) : h->img_f + n.f_base; a.bias = h->bias_img + n.bias_base;
  a.in = in; a.in1 = nullptr; a.out = out; a.acts = train ? n.acts : nullptr; a.masks = train ? n.masks : nullptr; a.pe_tile = train ? n.pe_tile : nullptr;
  a.in_scale = 0.5f; a.in_shift0 = 0.5f; a.in_shift1 = -0.5f; a.split_row = 0x7fffffff;
  a.NT = NT; a.nt_stride = NT; a.nl = n.NL;
  return a;
}

BwdArgs bwd_args(af_handle* h, NetDesc& n, int NT, bool bf) {
  BwdArgs a{};
  a.wimg = bf ? (h->mlp_mode == 3 ? (const float*)(h->img_hb + n.hb_base) : (const float*)(h->img_sb + n.sb_base)) : h->img_b + n.b_base; a.out = n.out_buf; a.dout = n.dout; a.masks = n.masks;
  a.dz = n.dz; a.dz_last = n.dz_last; a.pe_tile = n.pe_tile; a.din0 = nullptr; a.din1 = nullptr; a.din_scale = 0.5f;
  a.split_row = 0x7fffffff; a.nrows = 0; a.NT = NT; a.nt_stride = NT; a.nl = n.NL;
  return a;
}

struct Timer {
  af_handle* h; int cls; hipEvent_t a = nullptr, b = nullptr;
  Timer(af_handle* h_, int c, double flops = 0.0) : h(h_), cls(c) {
    if (on()) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, h->stream); h->t_flops[cls] += flops; }
  }
  ~Timer() { if (on()) { (void)hipEventRecord(b, h->stream); h->evs.push_back({cls, a, b}); } }
  bool on() const { return h->timing_live && ((h->timing >> cls) & 1u); }
};

void drain_timers(af_handle* h) {
  for (TimedEv& e : h->evs) {
    float ms = 0; (void)hipEventSynchronize(e.b); (void)hipEventElapsedTime(&ms, e.a, e.b);
    h->t_ms[e.cls] += ms; h->t_cnt[e.cls] += 1;
    (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
  }
  h->evs.clear();
}

AdamHyper adam_hyper(double lr, long long step) {
  const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
  const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  AdamHyper hy;
  hy.step_size = (float)(lr / bc1); hy.bc2_sqrt = (float)sqrt(bc2);
  hy.one_minus_b1 = (float)(1.0 - b1); hy.beta2 = (float)b2; hy.one_minus_b2 = (float)(1.0 - b2); hy.eps = (float)eps;
  return hy;
}

// The weight views k_adam keeps current: the canonical parameters, the fp32 images (mlp.hip, and always mlp16.hip's pre-train chains) and the
// 16-bit chain streams of the arithmetic in force ONLY (bf16 h/m/l for modes 1 and 2, fp16 h/l for mode 3: 12 resp. 8 scattered 2-byte
// stores per hidden weight that the other mode never reads); af_set_mlp_mode re-emits everything when the set changes.
AdamBufs adam_bufs(af_handle* h, float* m, float* v) {
  const bool bf = h->mlp_mode == 1 || h->mlp_mode == 2, hf = h->mlp_mode == 3;
  return {h->params, m, v, h->img_f, h->img_b, h->bias_img, bf ? h->img_sf : nullptr, bf ? h->img_sb : nullptr, hf ? h->img_hf : nullptr, hf ? h->img_hb : nullptr};
}

// re-emit the GEMM weight views of the jobs of a schedule from the canonical parameters
int repack(af_handle* h, Sched& sc) {
  AdamArgs a{};
  a.jobs = sc.d_ajobs; a.partial = h->partial;
  a.bufs = adam_bufs(h, h->adam_m, h->adam_v);
  a.nan_flag = h->nan_flag;
  LCHK(af_launch_adam(&a, (int)sc.ajobs.size(), 0, h->stream));
  return 0;
}

bool glob_on(const af_config& c, int iter) { return c.include_global_rigidity_loss && iter <= c.stop_global_rigidity; }

// rows of a part that carry real data (the last tile of a net may be padded)
double part_rows(int tile0, int NT, int rows_total) { return (double)std::max(0, std::min(NT * 32, rows_total) - tile0 * 32); }

struct FwdPart { int net; FwdArgs a; int rows_total; };
struct BwdPart { int net; BwdArgs a; int rows_total; };

// One forward / backward launch over independent parts (mlp.hip k_mlp_*_multi); empty parts are dropped.
int launch_fwd(af_handle* h, int cls, std::initializer_list<FwdPart> parts, bool train) {
  MultiFwd m{}; double fl = 0;
  for (const FwdPart& p : parts) {
    if (p.a.NT <= p.a.tile0) continue;
    m.net[m.n] = h->nets[p.net].kern; m.a[m.n] = p.a; ++m.n;
    fl += part_rows(p.a.tile0, p.a.NT, p.rows_total) * h->flop_fwd[p.net];
  }
  if (m.n == 0) return 0;
  if (h->step_stamp && train && (cls == T_FWD_1 || cls == T_FWD_2)) m.wg_stamp = h->step_stamp + (size_t)(cls == T_FWD_1 ? 0 : 1) * AF_STAMP_WG * 4;
  Timer t(h, cls, fl);
  if (h->mlp_mode == 3) LCHK(af_launch_fwd_multi_hf(&m, train ? 1 : 0, h->stream));
  else if (h->mlp_mode) LCHK(af_launch_fwd_multi_bf(&m, train ? 1 : 0, h->stream));
  else             LCHK(af_launch_fwd_multi(&m, train ? 1 : 0, h->stream));
  return 0;
}
int launch_bwd(af_handle* h, int cls, std::initializer_list<BwdPart> parts) {
  MultiBwd m{}; double fl = 0;
  for (const BwdPart& p : parts) {
    if (p.a.NT <= p.a.tile0) continue;
    m.net[m.n] = h->nets[p.net].kern; m.a[m.n] = p.a; ++m.n;
    fl += part_rows(p.a.tile0, p.a.NT, p.rows_total) * h->flop_dx[p.net];
  }
  if (m.n == 0) return 0;
  if (h->step_stamp && (cls == T_BWD_1 || cls == T_BWD_2)) m.wg_stamp = h->step_stamp + (size_t)(cls == T_BWD_1 ? 2 : 3) * AF_STAMP_WG * 4;
  Timer t(h, cls, fl);
  m.nprod = h->mlp_mode == 2 ? 3 : 6;
  if (h->mlp_mode == 3) LCHK(af_launch_bwd_multi_hf(&m, h->stream));
  else if (h->mlp_mode) LCHK(af_launch_bwd_multi_bf(&m, h->stream));
  else             LCHK(af_launch_bwd_multi(&m, h->stream));
  return 0;
}
template <class Args> Args with_live(Args a, const af_handle* h, int live_base) { a.live_rows = h->live; a.live_base = live_base; return a; }
FwdArgs tile_range(FwdArgs a, int t0, int t1) { a.tile0 = t0; a.NT = t1; return a; }
BwdArgs tile_range(BwdArgs a, int t0, int t1) { a.tile0 = t0; a.NT = t1; return a; }

// The mapping batch splits into whole rounds of the chip (ncu workgroups x 4 row tiles), [0, T1), and a remainder
// that rides with the atlas chain.  The whole rounds must contain every row the atlas chain depends on (the first
// `dep_rows` rows); if they do not, the batch is not split (T1 = NT).  Workgroup i + ncu is dispatched behind
// workgroup i (measured: two workgroups over one round cost a full atlas chain when they queue behind atlas
// workgroups, nothing when they queue behind the shorter mapping ones), so the remainder's first E workgroups —
// E = the number of workgroups beyond one round — lead the atlas launch: [T1, T2) before the atlas part, [T2, NT) after.
void plan_mapping_split(int ncu, int NT_map, int NT_atlas, int dep_rows, int& T1, int& T2) {
  const int round = ncu * 4, t1 = NT_map / round * round;
  T1 = t1 * 32 >= dep_rows ? t1 : NT_map;
  const int extra = (NT_atlas + 3) / 4 + (NT_map - T1 + 3) / 4 - ncu;
  T2 = std::min(NT_map, T1 + 4 * std::max(0, extra));
}

// dW of every layer of the schedule's nets + split-K reduction / Adam / weight-view re-emission + loss fold
int finish_step(af_handle* h, Sched& sc, float* m, float* v, long long step, float* loss_out, int loss_nblk, double dw_flops, bool check_counts = true) {
  { Timer t(h, T_DW, dw_flops); DwArgs d{sc.d_jobs, sc.d_segs, h->partial, h->dw_clock, h->step_stamp ? h->step_stamp + (size_t)4 * AF_STAMP_WG * 4 : nullptr}; LCHK(af_launch_dw(&d, sc.nwg, h->dw_mode, h->stream)); }
  {
    Timer t(h, T_ADAM);
    AdamArgs a{};
    a.jobs = sc.d_ajobs; a.partial = h->partial;
    a.bufs = adam_bufs(h, m, v);
    a.hy = adam_hyper(h->cfg.lr, step);
    a.grad_out = h->debug ? h->grads : nullptr;
    a.loss_part = h->loss_part; a.loss_out = loss_out; a.counts = h->counts; a.loss_nblk = loss_nblk; a.nan_flag = h->nan_flag;
    a.check_counts = check_counts ? 1 : 0;
    LCHK(af_launch_adam(&a, (int)sc.ajobs.size(), 1, h->stream));
  }
  return 0;
}

// Fetch and clear the device-side NaN flag (the stream is idle).  Non-zero: the reference would be carrying NaNs from here on.
int take_nan_flag(af_handle* h, int& flag) {
  flag = 0;
  HCHK(hipMemcpy(&flag, h->nan_flag, 4, hipMemcpyDeviceToHost));
  if (flag) HCHK(hipMemset(h->nan_flag, 0, 4));
  return 0;
}

int ensure_loss_log(af_handle* h, size_t steps) {
  if (steps * AF_LOSS_W <= h->loss_log_cap) return 0;
  if (h->loss_log) (void)hipFree(h->loss_log);
  h->loss_log = nullptr; h->loss_log_cap = 0;
  HCHK(dalloc(&h->loss_log, steps * AF_LOSS_W));
  h->loss_log_cap = steps * AF_LOSS_W;
  return 0;
}

// One iteration of the single-atlas loop (stage1_neural_atlas.py:159-231), enqueued on the handle's stream.
int enqueue_single_step(af_handle* h, int i, const int64_t* d_inds, uint64_t seed, float* loss_out) {
  const af_config& c = h->cfg;
  NetDesc& M = h->nets[AF_NET_MAP1]; NetDesc& A = h->nets[AF_NET_ATLAS];
  const int N = h->N, L = std::max(c.resx, c.resy);
  const bool glob = glob_on(c, i);
  const int nseg = glob ? 9 : 7;
  Sched& sc = h->sched[glob ? 0 : 1];
  const int NT_map = tiles_of(nseg * N), NT_atlas = tiles_of(3 * N);
  if (nseg != h->cur_nseg) {   // pad rows of the last tile must carry zero gradient
    HCHK(hipMemsetAsync(M.dout, 0, (size_t)M.nt_cap * 32 * 16, h->stream));
    h->cur_nseg = nseg;
  }
  {
    Timer t(h, T_PREP);
    PrepArgs p{};
    p.table = h->table; p.inds = d_inds; p.seed = seed; p.iter = (uint32_t)i;
    p.N = N; p.resx = c.resx; p.resy = c.resy; p.F = c.number_of_frames;
    p.half_main = (float)(L / 2.0); p.half_grad = (float)(c.resx / 2.0); p.half_frames = (float)(c.number_of_frames / 2.0);
    p.d_local = c.derivative_amount; p.d_global = c.global_rigidity_derivative_amount_fg; p.nseg = nseg;
    p.coords = M.coords; p.x0_tile = M.x0_tile; p.samples = h->samples; p.counts = h->counts;
    p.flow_rank = h->flow_rank; p.scan = h->scan; p.live = h->live; p.epoch = ++h->prep_epoch;
    p.ticket = h->scan + (N + 255) / 256; p.ticket_base = h->prep_tickets;
    LCHK(af_launch_prep(&p, h->stream));
    h->prep_tickets += (unsigned long long)((N + 255) / 256);      // only a launch that went out takes tickets: the device counter and this base must not part
  }
  // Launch 1: the whole rounds of the mapping batch (they hold the 3N rows the atlas reads).  Launch 2: the atlas
  // chain plus the mapping remainder — rigidity / flow rows nothing in this launch depends on — in the CUs the
  // atlas workgroups leave idle.  The backward pass mirrors it (the remainder needs no atlas gradient).
  int rc, T1, T2;
  plan_mapping_split(h->ncu, std::min(NT_map, tiles_of((nseg - 2) * N + h->plan_flow_rows)), NT_atlas, 3 * N, T1, T2);
  const int flow_base = (nseg - 2) * N;      // rows behind flow_base + (valid matches of this batch) do not exist: the launches are sized for the maximum
  const FwdArgs fm = with_live(fwd_args(h, M, M.coords, M.out_buf, NT_map, true, h->mlp_mode != 0), h, flow_base);
  if ((rc = launch_fwd(h, T_FWD_1, {{AF_NET_MAP1, tile_range(fm, 0, T1), nseg * N}}, true)) != 0) return rc;
  if ((rc = launch_fwd(h, T_FWD_2, {{AF_NET_MAP1, tile_range(fm, T1, T2), nseg * N},
                                    {AF_NET_ATLAS, fwd_args(h, A, M.out_buf, A.out_buf, NT_atlas, true, h->mlp_mode != 0), 3 * N},
                                    {AF_NET_MAP1, tile_range(fm, T2, NT_map), nseg * N}}, true)) != 0) return rc;
  {
    Timer t(h, T_LOSS);
    LossArgs l{};
    l.samples = h->samples; l.out_map = M.out_buf; l.out_atlas = A.out_buf; l.dout_map = M.dout; l.dout_atlas = A.dout;
    l.counts = h->counts; l.loss_part = h->loss_part; l.N = N; l.nseg = nseg; l.flow_rank = h->flow_rank; l.live = h->live;
    l.L = (float)L; l.uv_scale = c.uv_mapping_scale; l.d_local = c.derivative_amount; l.d_global = c.global_rigidity_derivative_amount_fg;
    l.c_rgb = c.rgb_coeff; l.c_grad = c.use_gradient_loss ? c.gradient_loss_coeff : 0.f; l.c_rig = c.rigidity_coeff;
    l.c_grig = glob ? c.global_rigidity_coeff_fg : 0.f; l.c_flow = c.optical_flow_coeff;
    LCHK(af_launch_loss_single(&l, h->stream));
  }
  h->adam_step += 1;
  {
    BwdArgs ba = bwd_args(h, A, NT_atlas, h->mlp_mode != 0); ba.din0 = M.dout; ba.nrows = 3 * N;
    const BwdArgs bm = with_live(bwd_args(h, M, NT_map, h->mlp_mode != 0), h, flow_base);
    if ((rc = launch_bwd(h, T_BWD_1, {{AF_NET_MAP1, tile_range(bm, T1, T2), nseg * N}, {AF_NET_ATLAS, ba, 3 * N},
                                      {AF_NET_MAP1, tile_range(bm, T2, NT_map), nseg * N}})) != 0) return rc;
    if ((rc = launch_bwd(h, T_BWD_2, {{AF_NET_MAP1, tile_range(bm, 0, T1), nseg * N}})) != 0) return rc;
  }
  const double dwf = (double)nseg * N * h->flop_fwd[AF_NET_MAP1] + 3.0 * N * h->flop_fwd[AF_NET_ATLAS];
  return finish_step(h, sc, h->adam_m, h->adam_v, h->adam_step, loss_out, (N + 255) / 256, dwf);
}

// One iteration of the fg/bg dual-atlas loop (stage1_neural_atlas_seg.py:193-315).
int enqueue_seg_step(af_handle* h, int i, const int64_t* d_inds, uint64_t seed, float* loss_out) {
  const af_config& c = h->cfg;
  NetDesc& M1 = h->nets[AF_NET_MAP1]; NetDesc& M2 = h->nets[AF_NET_MAP2]; NetDesc& A = h->nets[AF_NET_ATLAS]; NetDesc& AL = h->nets[AF_NET_ALPHA];
  const int N = h->N, L = std::max(c.resx, c.resy);
  const bool glob = glob_on(c, i);
  const int nseg = glob ? 9 : 7;
  Sched& sc = h->sched[glob ? 0 : 1];
  const int NT_map = tiles_of(nseg * N), NT_atlas = tiles_of(6 * N), NT_alpha = tiles_of(5 * N);
  if (nseg != h->cur_nseg) {
    HCHK(hipMemsetAsync(M1.dout, 0, (size_t)M1.nt_cap * 32 * 16, h->stream));
    HCHK(hipMemsetAsync(M2.dout, 0, (size_t)M2.nt_cap * 32 * 16, h->stream));
    h->cur_nseg = nseg;
  }
  {
    Timer t(h, T_PREP);
    PrepArgs p{};
    p.table = h->table; p.inds = d_inds; p.seed = seed; p.iter = (uint32_t)i;
    p.N = N; p.resx = c.resx; p.resy = c.resy; p.F = c.number_of_frames;
    p.half_main = (float)(L / 2.0); p.half_grad = (float)(c.resx / 2.0); p.half_frames = (float)(c.number_of_frames / 2.0);
    p.d_local = c.derivative_amount; p.d_global = c.global_rigidity_derivative_amount_fg; p.nseg = nseg;
    p.coords = M1.coords; p.x0_tile = M1.x0_tile; p.samples = h->samples; p.counts = h->counts;
    p.coords2 = M2.coords; p.x0_tile2 = M2.x0_tile; p.coordsA = AL.coords; p.d_global2 = c.global_rigidity_derivative_amount_bg;
    p.flow_rank = h->flow_rank; p.scan = h->scan; p.live = h->live; p.epoch = ++h->prep_epoch;
    p.ticket = h->scan + (N + 255) / 256; p.ticket_base = h->prep_tickets;
    LCHK(af_launch_prep(&p, h->stream));
    h->prep_tickets += (unsigned long long)((N + 255) / 256);      // only a launch that went out takes tickets: the device counter and this base must not part
  }
  // Launch 1: alpha, mapping1, mapping2 (longest chains first, so the launch drains on the short ones).
  // Launch 2: the atlas chain — rows [0,3N) = uv1*0.5+0.5 (foreground quadrant), [3N,6N) = uv2*0.5-0.5
  // (background), :229-232 — topped up to whole rounds of the chip with the last alpha row tiles.
  int rc;
  const int wg_atlas = (NT_atlas + 3) / 4, pad = (h->ncu - wg_atlas % h->ncu) % h->ncu;
  const int T_al = std::max(0, std::min(NT_alpha, tiles_of(3 * N + h->plan_flow_rows)) - 4 * pad);     // alpha tiles [T_al, NT_alpha) ride with the atlas (the expected live ones fill its last round)
  const int flow_base = (nseg - 2) * N;      // mapping rows behind flow_base + (valid matches), alpha rows behind 3N + (valid matches) do not exist
  const FwdArgs fal = with_live(fwd_args(h, AL, AL.coords, AL.out_buf, NT_alpha, true, h->mlp_mode != 0), h, 3 * N);
  FwdArgs fat = fwd_args(h, A, M1.out_buf, A.out_buf, NT_atlas, true, h->mlp_mode != 0);
  fat.in1 = M2.out_buf; fat.split_row = 3 * N;
  if ((rc = launch_fwd(h, T_FWD_1, {{AF_NET_ALPHA, tile_range(fal, 0, T_al), 5 * N},
                                    {AF_NET_MAP1, with_live(fwd_args(h, M1, M1.coords, M1.out_buf, NT_map, true, h->mlp_mode != 0), h, flow_base), nseg * N},
                                    {AF_NET_MAP2, with_live(fwd_args(h, M2, M2.coords, M2.out_buf, NT_map, true, h->mlp_mode != 0), h, flow_base), nseg * N}}, true)) != 0) return rc;
  if ((rc = launch_fwd(h, T_FWD_2, {{AF_NET_ATLAS, fat, 6 * N}, {AF_NET_ALPHA, tile_range(fal, T_al, NT_alpha), 5 * N}}, true)) != 0) return rc;
  {
    Timer t(h, T_LOSS);
    LossSegArgs l{};
    l.samples = h->samples; l.out_m1 = M1.out_buf; l.out_m2 = M2.out_buf; l.out_alpha = AL.out_buf; l.out_atlas = A.out_buf;
    l.dout_m1 = M1.dout; l.dout_m2 = M2.dout; l.dout_alpha = AL.dout; l.dout_atlas = A.dout;
    l.counts = h->counts; l.loss_part = h->loss_part; l.N = N; l.nseg = nseg; l.flow_rank = h->flow_rank; l.live = h->live;
    l.L = (float)L; l.uv_scale = c.uv_mapping_scale; l.d_local = c.derivative_amount;
    l.d_global_fg = c.global_rigidity_derivative_amount_fg; l.d_global_bg = c.global_rigidity_derivative_amount_bg;
    l.c_rgb = c.rgb_coeff; l.c_grad = c.use_gradient_loss ? c.gradient_loss_coeff : 0.f; l.c_rig = c.rigidity_coeff;
    l.c_grig_fg = glob ? c.global_rigidity_coeff_fg : 0.f; l.c_grig_bg = glob ? c.global_rigidity_coeff_bg : 0.f;
    l.c_flow = c.optical_flow_coeff;
    l.c_boot = i > c.stop_bootstrapping_iteration ? 0.f : c.alpha_bootstrapping_factor;     // :193-194
    l.c_aflow = c.alpha_flow_factor; l.c_sparse = c.sparsity_coeff;
    LCHK(af_launch_loss_seg(&l, h->stream));
  }
  h->adam_step += 1;
  {   // the mapping chains need the atlas chain's input gradient (rows < 3N): atlas (+ alpha top-up) first
    BwdArgs ba = bwd_args(h, A, NT_atlas, h->mlp_mode != 0);
    ba.din0 = M1.dout; ba.din1 = M2.dout; ba.split_row = 3 * N; ba.nrows = 6 * N;
    const BwdArgs bal = with_live(bwd_args(h, AL, NT_alpha, h->mlp_mode != 0), h, 3 * N);
    if ((rc = launch_bwd(h, T_BWD_1, {{AF_NET_ATLAS, ba, 6 * N}, {AF_NET_ALPHA, tile_range(bal, T_al, NT_alpha), 5 * N}})) != 0) return rc;
    if ((rc = launch_bwd(h, T_BWD_2, {{AF_NET_ALPHA, tile_range(bal, 0, T_al), 5 * N}, {AF_NET_MAP1, with_live(bwd_args(h, M1, NT_map, h->mlp_mode != 0), h, flow_base), nseg * N},
                                      {AF_NET_MAP2, with_live(bwd_args(h, M2, NT_map, h->mlp_mode != 0), h, flow_base), nseg * N}})) != 0) return rc;
  }
  const double dwf = (double)nseg * N * (h->flop_fwd[AF_NET_MAP1] + h->flop_fwd[AF_NET_MAP2]) + 6.0 * N * h->flop_fwd[AF_NET_ATLAS] + 5.0 * N * h->flop_fwd[AF_NET_ALPHA];
  return finish_step(h, sc, h->adam_m, h->adam_v, h->adam_step, loss_out, (N + 255) / 256, dwf);
}

}  // namespace

// =================================================================================================
extern "C" {

size_t af_config_size(void) { return sizeof(af_config); }

// ---- input builder utilities (stateless; default stream of the device)
namespace {
struct Staged {   // host buffer mirrored on the device for the duration of a call
  void* d = nullptr; bool own = false;
  hipError_t in(const void* p, size_t bytes, bool on_device) {
    if (on_device) { d = const_cast<void*>(p); return hipSuccess; }
    hipError_t e = hipMalloc(&d, std::max<size_t>(bytes, 1)); if (e != hipSuccess) return e;
    own = true;
    return p ? hipMemcpy(d, p, bytes, hipMemcpyHostToDevice) : hipSuccess;
  }
  ~Staged() { if (own) (void)hipFree(d); }
};
int util_fail(const char* what, hipError_t e) { g_create_error = std::string(what) + ": " + hipGetErrorString(e); return AF_EHIP; }
}  // namespace

int af_resize_bilinear(int device_ordinal, const void* src, int src_u8, int sh, int sw, int ch, float* dst, int dh, int dw,
                       int64_t pix_stride, int64_t ch_stride, int64_t offset, double scale0, double scale1, int on_device) {
  if (!src || !dst || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || ch <= 0 || pix_stride <= 0) { g_create_error = "af_resize_bilinear: arguments"; return AF_EINVAL; }
  hipError_t e = hipSetDevice(device_ordinal); if (e != hipSuccess) return util_fail("hipSetDevice", e);
  if (!on_device && (pix_stride != ch || ch_stride != 1 || offset != 0)) { g_create_error = "af_resize_bilinear: host destinations must be HWC contiguous"; return AF_EINVAL; }
  Staged s, d;
  const size_t sbytes = (size_t)sh * sw * ch * (src_u8 ? 1 : 4), dbytes = (size_t)dh * dw * ch * 4;
  if ((e = s.in(src, sbytes, on_device)) != hipSuccess) return util_fail("stage source", e);
  if ((e = d.in(on_device ? (const void*)dst : nullptr, dbytes, on_device)) != hipSuccess) return util_fail("stage destination", e);
  ResizeArgs a{s.d, src_u8, sh, sw, ch, (float*)d.d, dh, dw, pix_stride, ch_stride, offset, scale0, scale1};
  int r = af_launch_resize(&a, nullptr); if (r) return util_fail("k_resize_bilinear", (hipError_t)r);
  if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return util_fail("k_resize_bilinear", e);
  if (!on_device && (e = hipMemcpy(dst, d.d, dbytes, hipMemcpyDeviceToHost)) != hipSuccess) return util_fail("copy back", e);
  return AF_OK;
}

int af_flow_consistency(int device_ordinal, const float* f12, const float* f21, int h, int w, float* out,
                        int64_t pix_stride, int64_t offset, float thresh, int on_device) {
  if (!f12 || !f21 || !out || h <= 0 || w <= 0 || pix_stride <= 0) { g_create_error = "af_flow_consistency: arguments"; return AF_EINVAL; }
  hipError_t e = hipSetDevice(device_ordinal); if (e != hipSuccess) return util_fail("hipSetDevice", e);
  if (!on_device && (pix_stride != 1 || offset != 0)) { g_create_error = "af_flow_consistency: host destinations must be contiguous"; return AF_EINVAL; }
  Staged a12, a21, o;
  const size_t fb = (size_t)h * w * 8, ob = (size_t)h * w * 4;
  if ((e = a12.in(f12, fb, on_device)) != hipSuccess || (e = a21.in(f21, fb, on_device)) != hipSuccess) return util_fail("stage flows", e);
  if ((e = o.in(on_device ? (const void*)out : nullptr, ob, on_device)) != hipSuccess) return util_fail("stage output", e);
  ConsistencyArgs a{(const float*)a12.d, (const float*)a21.d, h, w, (float*)o.d, pix_stride, offset, thresh};
  int r = af_launch_consistency(&a, nullptr); if (r) return util_fail("k_flow_consistency", (hipError_t)r);
  if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return util_fail("k_flow_consistency", e);
  if (!on_device && (e = hipMemcpy(out, o.d, ob, hipMemcpyDeviceToHost)) != hipSuccess) return util_fail("copy back", e);
  return AF_OK;
}

const char* af_last_error(const af_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int af_create(const af_config* cfg, int device_ordinal, af_handle** out) {
  if (!cfg || !out) { g_create_error = "af_create: null argument"; return AF_EINVAL; }
  *out = nullptr;
  auto bad = [&](const char* m) { g_create_error = std::string("af_create: ") + m; return (int)AF_EINVAL; };
  if (cfg->resx <= 1 || cfg->resy <= 1 || cfg->number_of_frames <= 0) return bad("resx/resy/number_of_frames");
  if (cfg->samples_batch <= 0) return bad("samples_batch");
  auto width_ok = [](int w) { return w >= 1 && w <= AF_HID; };       // narrower nets run zero-padded inside the 256-wide chains (NetDesc::hid); wider ones are not built
  if (!width_ok(cfg->number_of_channels_mapping1) || !width_ok(cfg->number_of_channels_atlas)) return bad("number_of_channels_mapping1 / number_of_channels_atlas must be 1..256 (config_flow_100.json:19,24)");
  auto layers_ok = [](int n) { return n >= 2 && n <= AF_MAX_LAYERS; };
  if (!layers_ok(cfg->number_of_layers_mapping1) || !layers_ok(cfg->number_of_layers_atlas)) return bad("number_of_layers_mapping1 / number_of_layers_atlas must be 2..8");
  if (cfg->positional_encoding_num_atlas < 1 || cfg->positional_encoding_num_atlas > 10) return bad("positional_encoding_num_atlas must be 1..10");
  auto pe_ok = [](int k) { return k >= 1 && k <= 5; };
  if (cfg->use_positional_encoding_mapping1 && !pe_ok(cfg->number_of_positional_encoding_mapping1)) return bad("number_of_positional_encoding_mapping1 must be 1..5 when use_positional_encoding_mapping1 is set");
  if (cfg->derivative_amount <= 0 || cfg->global_rigidity_derivative_amount_fg <= 0) return bad("derivative amounts");
  const bool seg = cfg->two_layer != 0;
  if (seg) {
    if (!width_ok(cfg->number_of_channels_mapping2) || !width_ok(cfg->number_of_channels_alpha)) return bad("number_of_channels_mapping2 / number_of_channels_alpha must be 1..256 (config_flow_100.json:21,26)");
    if (!layers_ok(cfg->number_of_layers_mapping2) || !layers_ok(cfg->number_of_layers_alpha)) return bad("number_of_layers_mapping2 / number_of_layers_alpha must be 2..8");
    if (cfg->positional_encoding_num_alpha < 1 || cfg->positional_encoding_num_alpha > 5) return bad("positional_encoding_num_alpha must be 1..5");
    if (cfg->use_positional_encoding_mapping2 && !pe_ok(cfg->number_of_positional_encoding_mapping2)) return bad("number_of_positional_encoding_mapping2 must be 1..5 when use_positional_encoding_mapping2 is set");
    if (cfg->global_rigidity_derivative_amount_bg <= 0) return bad("global_rigidity_derivative_amount_bg");
  }
  hipError_t e = hipSetDevice(device_ordinal);
  if (e != hipSuccess) { g_create_error = std::string("af_create: hipSetDevice: ") + hipGetErrorString(e); return AF_EHIP; }
  af_handle* h = new af_handle();
  h->cfg = *cfg; h->device = device_ordinal; h->seg = seg;
  // no environment is read here: the arithmetic modes and the tile-cost row are set through af_set_mlp_mode / af_set_dw_mode / af_debug_set_dw_cost
  // (the Python mirror maps AF_MLP_MODE, AF_DW_MODE, AF_MLP_FP32, AF_DW_FP32, AF_DW_COST onto those calls for the A/B tools)
  if (h->cfg.pretrain_batch <= 0) h->cfg.pretrain_batch = 10000;
  if (h->cfg.lr <= 0) h->cfg.lr = 1e-4f;
  auto die = [&](int code) { g_create_error = h->err; af_destroy(h); return code; };
#define CCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { h->fail(AF_EHIP, #x, e_); return die(e_ == hipErrorOutOfMemory ? AF_ENOMEM : AF_EHIP); } } while (0)
  hipDeviceProp_t prop; CCHK(hipGetDeviceProperties(&prop, device_ordinal));
  h->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  CCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CCHK((hipError_t)af_mlp_init()); CCHK((hipError_t)af_mlp16_init()); CCHK((hipError_t)af_dw_init()); CCHK((hipError_t)af_mlp_bf_init()); CCHK((hipError_t)af_mlp_hf_init());

  // layer counts come from the config (stage1_neural_atlas.py:112-128, _seg.py:127-161); the atlas net's skip_layers=[4, 7] apply to the
  // layers it has (implicit_neural_networks.py:40-44: `if i in skip_layers` for i < num_layers, the output layer included)
  const int nl_atlas = cfg->number_of_layers_atlas;
  const unsigned atlas_skip = (nl_atlas > 4 ? (1u << 4) : 0u) | (nl_atlas > 7 ? (1u << 7) : 0u);
  // a mapping net with positional encoding (IMLP(use_positional=True, positional_dim=K), implicit_neural_networks.py:9-13,28-33): PE 3 -> 6K
  if (cfg->use_positional_encoding_mapping1) describe_net(h->nets[AF_NET_MAP1], AF_NET_MAP1, cfg->number_of_layers_mapping1, AF_IN_PE3, cfg->number_of_positional_encoding_mapping1, 2, 0u, false, cfg->number_of_channels_mapping1);
  else describe_net(h->nets[AF_NET_MAP1], AF_NET_MAP1, cfg->number_of_layers_mapping1, AF_IN_XYT, 0, 2, 0u, false, cfg->number_of_channels_mapping1);
  describe_net(h->nets[AF_NET_ATLAS], AF_NET_ATLAS, nl_atlas, AF_IN_PE2, cfg->positional_encoding_num_atlas, 3, atlas_skip, true, cfg->number_of_channels_atlas);
  if (seg) {
    if (cfg->use_positional_encoding_mapping2) describe_net(h->nets[AF_NET_MAP2], AF_NET_MAP2, cfg->number_of_layers_mapping2, AF_IN_PE3, cfg->number_of_positional_encoding_mapping2, 2, 0u, false, cfg->number_of_channels_mapping2);
    else describe_net(h->nets[AF_NET_MAP2], AF_NET_MAP2, cfg->number_of_layers_mapping2, AF_IN_XYT, 0, 2, 0u, false, cfg->number_of_channels_mapping2);
    describe_net(h->nets[AF_NET_ALPHA], AF_NET_ALPHA, cfg->number_of_layers_alpha, AF_IN_PE3, cfg->positional_encoding_num_alpha, 1, 0u, false, cfg->number_of_channels_alpha);
  }
  for (NetDesc& n : h->nets) if (n.used) {
    // ALGORITHMIC work = what the configured (hid-wide) net needs, not what the 256-wide machinery executes for a narrower one
    auto lw = [&](int w) { return w == AF_HID ? n.hid : (w > AF_HID ? w - AF_HID + n.hid : w); };     // physical width -> logical (hidden part narrowed, PE columns kept)
    double f = 0, d = n.dx0 ? (double)n.in_feat0 * n.hid : 0.0;
    for (int l = 0; l < n.NL; ++l) {
      const double lin = l == 0 ? n.in_feat0 : lw(n.in_feat[l]), lout = l == n.NL - 1 ? n.out : n.hid;
      f += lin * lout; if (l >= 1) d += (double)n.hid * lout;
    }
    h->flop_fwd[n.id] = 2.0 * f; h->flop_dx[n.id] = 2.0 * d;
  }
  size_t fc = 0, bc = 0, biasc = 0, pc = 0;
  for (NetDesc& n : h->nets) if (n.used) {
    n.p_base = pc; pc += n.nparams; plan_images(n, fc, bc, biasc);
    if (!check_chunk_plan(n)) { h->fail(AF_EINVAL, "weight-image plan does not match the kernels' chunk sequence"); return die(AF_EINVAL); }
    if (!plan_streams_bf(n, h->sf_bytes, h->sb_bytes)) { h->fail(AF_EINVAL, "bf16 stream plan does not match the kernels' chunk sizes"); return die(AF_EINVAL); }
    if (!plan_streams_hf(n, h->hf_bytes, h->hb_bytes)) { h->fail(AF_EINVAL, "fp16 stream plan does not match the kernels' chunk sizes"); return die(AF_EINVAL); }
  }
  h->sf_bytes += 49152 + 65536; h->sb_bytes += 49152 + 65536;   // every LDS stage copies a full slot: keep the over-read in bounds
  if (h->sf_bytes >= ((size_t)1 << 31) || h->sb_bytes >= ((size_t)1 << 31)) { h->fail(AF_EINVAL, "stream images exceed 2 GB"); return die(AF_EINVAL); }
  CCHK(hipMalloc((void**)&h->img_sf, h->sf_bytes)); CCHK(hipMalloc((void**)&h->img_sb, h->sb_bytes));
  CCHK(hipMemset(h->img_sf, 0, h->sf_bytes)); CCHK(hipMemset(h->img_sb, 0, h->sb_bytes));
  h->hf_bytes += 2 * 65536; h->hb_bytes += 2 * 65536;          // every LDS stage copies a full 64 KB slot
  if (h->hf_bytes >= ((size_t)1 << 31) || h->hb_bytes >= ((size_t)1 << 31)) { h->fail(AF_EINVAL, "stream images exceed 2 GB"); return die(AF_EINVAL); }
  CCHK(hipMalloc((void**)&h->img_hf, h->hf_bytes)); CCHK(hipMalloc((void**)&h->img_hb, h->hb_bytes));
  CCHK(hipMemset(h->img_hf, 0, h->hf_bytes)); CCHK(hipMemset(h->img_hb, 0, h->hb_bytes));
  fc += AF_CHUNK_MAX / 4; bc += AF_CHUNK_MAX / 4;   // every LDS stage copies a full 64 KB buffer: keep the over-read in bounds
  h->total_params = pc; h->img_f_floats = fc; h->img_b_floats = bc; h->bias_floats = biasc;
  CCHK(dalloc(&h->params, pc)); CCHK(dalloc(&h->adam_m, pc)); CCHK(dalloc(&h->adam_v, pc));
  CCHK(dalloc(&h->pre_m, pc)); CCHK(dalloc(&h->pre_v, pc)); CCHK(dalloc(&h->grads, pc));
  CCHK(dalloc(&h->img_f, fc)); CCHK(dalloc(&h->img_b, bc)); CCHK(dalloc(&h->bias_img, biasc));
  CCHK(hipMemset(h->params, 0, pc * 4)); CCHK(hipMemset(h->adam_m, 0, pc * 4)); CCHK(hipMemset(h->adam_v, 0, pc * 4));
  CCHK(hipMemset(h->pre_m, 0, pc * 4)); CCHK(hipMemset(h->pre_v, 0, pc * 4)); CCHK(hipMemset(h->grads, 0, pc * 4));
  // INVARIANT: image slots no parameter maps to stay EXACTLY zero for the life of the handle.  The PE stages always walk
  // their shipped slot layout (`peg` above: five 3-D / ten 2-D frequencies) and padded K rows; k_adam, load_state_dict and the
  // repack of a mode switch only ever write the slots af_img_index() gives a real parameter, so the zeros written here are what
  // carries positional_encoding_num_* below the shipped count and the K padding.  Guard: tests/test_gpu_arch.py trains such
  // configurations for several Adam steps against the CPU restatement (a slot that picked up a value would show in the second step's losses).
  CCHK(hipMemset(h->img_f, 0, fc * 4)); CCHK(hipMemset(h->img_b, 0, bc * 4)); CCHK(hipMemset(h->bias_img, 0, biasc * 4));
  // batch buffers
  h->N = cfg->samples_batch;
  const int N = h->N, rows_pre = h->cfg.pretrain_batch;
  NetDesc& M = h->nets[AF_NET_MAP1]; NetDesc& A = h->nets[AF_NET_ATLAS]; NetDesc& M2 = h->nets[AF_NET_MAP2]; NetDesc& AL = h->nets[AF_NET_ALPHA];
  CCHK(alloc_net_buffers(M, std::max(9 * N, rows_pre), true));
  CCHK(alloc_net_buffers(A, (seg ? 6 : 3) * N, false));
  if (seg) {
    CCHK(alloc_net_buffers(M2, std::max(9 * N, rows_pre), true));
    CCHK(alloc_net_buffers(AL, 5 * N, true));
  }
  CCHK(dalloc(&h->samples, (size_t)N * AF_REC_F));
  h->loss_nblk_cap = (std::max(N, rows_pre) + 255) / 256;
  CCHK(dalloc(&h->loss_part, (size_t)h->loss_nblk_cap * AF_LOSS_W)); CCHK(hipMemset(h->loss_part, 0, (size_t)h->loss_nblk_cap * AF_LOSS_W * 4));
  CCHK(dalloc(&h->counts, 2)); CCHK(hipMemset(h->counts, 0, 8));
  CCHK(dalloc(&h->nan_flag, 1)); CCHK(hipMemset(h->nan_flag, 0, 4));
  CCHK(dalloc(&h->flow_rank, (size_t)2 * N)); CCHK(hipMemset(h->flow_rank, 0xff, (size_t)2 * N * 4));
  CCHK(dalloc(&h->scan, (size_t)(N + 255) / 256 + 1)); CCHK(hipMemset(h->scan, 0, (size_t)((N + 255) / 256 + 1) * 8));   // + the ticket counter of k_prep
  CCHK(dalloc(&h->live, 1)); CCHK(hipMemset(h->live, 0, 4));
  CCHK(dalloc(&h->nvalid, 128));
  // schedules
  bool ok = build_main_scheds(h);
  ok = ok && build_sched(h, h->sched[2], {{&M, tiles_of(rows_pre)}});
  if (seg) ok = ok && build_sched(h, h->sched[3], {{&M2, tiles_of(rows_pre)}});
  if (!ok) { h->fail(AF_EINVAL, "dW schedule needs more than DW_MAXSEG segments per workgroup"); return die(AF_EINVAL); }
  size_t pf = 0;
  for (int i = 0; i < (seg ? 4 : 3); ++i) { CCHK(upload_sched(h->sched[i])); pf = std::max(pf, h->sched[i].partial_floats); }
  CCHK(dalloc(&h->partial, pf)); h->partial_cap = pf;
  h->frame_sse.assign(cfg->number_of_frames, 0.0); h->frame_sse_valid.assign(cfg->number_of_frames, 0);
  if (repack(h, h->sched[0]) != 0) return die(AF_EHIP);
  CCHK(hipStreamSynchronize(h->stream));
#undef CCHK
  *out = h;
  return AF_OK;
}

void af_destroy(af_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  drain_timers(h);
  for (NetDesc& n : h->nets) if (n.used) free_net(n);
  for (Sched& s : h->sched) { (void)hipFree(s.d_jobs); (void)hipFree(s.d_ajobs); (void)hipFree(s.d_segs); }
  (void)hipFree(h->params); (void)hipFree(h->adam_m); (void)hipFree(h->adam_v); (void)hipFree(h->pre_m); (void)hipFree(h->pre_v); (void)hipFree(h->grads);
  (void)hipFree(h->img_f); (void)hipFree(h->img_b); (void)hipFree(h->bias_img); (void)hipFree(h->table); (void)hipFree(h->img_sf); (void)hipFree(h->img_sb); (void)hipFree(h->img_hf); (void)hipFree(h->img_hb);
  (void)hipFree(h->samples); (void)hipFree(h->loss_part); (void)hipFree(h->loss_log); (void)hipFree(h->counts); (void)hipFree(h->nan_flag); (void)hipFree(h->flow_rank); (void)hipFree(h->scan); (void)hipFree(h->live); (void)hipFree(h->nvalid);
  (void)hipFree(h->partial); (void)hipFree(h->dw_clock); (void)hipFree(h->step_stamp); (void)hipFree(h->r_coords); (void)hipFree(h->r_uv); (void)hipFree(h->r_uv2); (void)hipFree(h->r_al);
  (void)hipFree(h->r_t); (void)hipFree(h->r_rgb); (void)hipFree(h->r_sse);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int af_sync(af_handle* h) {
  if (!h) return AF_EINVAL;
  HCHK(hipSetDevice(h->device)); HCHK(hipStreamSynchronize(h->stream));
  return AF_OK;
}

int af_upload_video(af_handle* h, const float* frames, const float* flow_fwd, const float* flow_bwd,
                    const float* mask_fwd, const float* mask_bwd, const float* mask_fg, int on_device) {
  if (!h) return AF_EINVAL;
  if (!frames || !flow_fwd || !flow_bwd || !mask_fwd || !mask_bwd) return h->fail(AF_EINVAL, "af_upload_video: null tensor");
  if (h->seg && !mask_fg) return h->fail(AF_EINVAL, "af_upload_video: a two_layer handle needs mask_fg (mask_frames, unwrap_utils.py:68-70)");
  HCHK(hipSetDevice(h->device));
  const size_t P2 = (size_t)h->cfg.resx * h->cfg.resy, F = h->cfg.number_of_frames, P = P2 * F;
  if (!h->table) { hipError_t e = dalloc(&h->table, P * AF_REC_F); if (e != hipSuccess) return h->fail(AF_ENOMEM, "record table", e); }
  const float* src[6] = {frames, flow_fwd, flow_bwd, mask_fwd, mask_bwd, mask_fg};
  const size_t cnt[6] = {P * 3, P * 2, P * 2, P, P, P};
  float* tmp[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const float* dev[6];
  int rc = AF_OK;
  for (int i = 0; i < 6 && rc == AF_OK; ++i) {
    if (!src[i]) { dev[i] = nullptr; continue; }
    if (on_device) { dev[i] = src[i]; continue; }
    hipError_t e = dalloc(&tmp[i], cnt[i]);
    if (e == hipSuccess) e = hipMemcpy(tmp[i], src[i], cnt[i] * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = h->fail(e == hipErrorOutOfMemory ? AF_ENOMEM : AF_EHIP, "af_upload_video staging", e);
    dev[i] = tmp[i];
  }
  if (rc == AF_OK) {
    PackArgs a{dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], h->table, h->cfg.resx, h->cfg.resy, h->cfg.number_of_frames, h->nvalid};
    hipError_t e = hipMemsetAsync(h->nvalid, 0, 128 * 8, h->stream);
    int r = e == hipSuccess ? af_launch_pack(&a, h->stream) : (int)e;
    e = r ? (hipError_t)r : hipStreamSynchronize(h->stream);
    unsigned long long nv[2] = {0, 0}, nvs[128];
    if (e == hipSuccess) e = hipMemcpy(nvs, h->nvalid, sizeof nvs, hipMemcpyDeviceToHost);
    if (e == hipSuccess) for (int i = 0; i < 64; ++i) { nv[0] += nvs[2 * i]; nv[1] += nvs[2 * i + 1]; }
    if (e != hipSuccess) rc = h->fail(AF_EHIP, "pack table", e);
    else {
      // re-balance launches and dW schedule for the share of valid flow matches of THIS video (the rows are compacted on the device)
      h->p_valid[0] = (double)nv[0] / (double)P; h->p_valid[1] = (double)nv[1] / (double)P;
      if (planned_flow_rows(h) != h->plan_flow_rows) {
        if (!build_main_scheds(h)) rc = h->fail(AF_EINVAL, "dW schedule needs more than DW_MAXSEG segments per workgroup");
        for (int i = 0; i < 2 && rc == AF_OK; ++i) {
          if ((e = upload_sched(h->sched[i])) != hipSuccess) rc = h->fail(AF_EHIP, "schedule upload", e);
          if (rc == AF_OK && h->sched[i].partial_floats > h->partial_cap) {
            (void)hipFree(h->partial); h->partial = nullptr;
            if ((e = dalloc(&h->partial, h->sched[i].partial_floats)) != hipSuccess) rc = h->fail(AF_ENOMEM, "partial dW buffer", e);
            else h->partial_cap = h->sched[i].partial_floats;
          }
        }
      }
    }
  }
  for (int i = 0; i < 6; ++i) if (tmp[i]) (void)hipFree(tmp[i]);
  if (rc == AF_OK) { h->have_video = true; std::fill(h->frame_sse_valid.begin(), h->frame_sse_valid.end(), 0); }
  return rc;
}

size_t af_param_count(const af_handle* h, int net) {
  if (!h || net < 0 || net >= AF_MAX_NETS || !h->nets[net].used) return 0;
  return h->nets[net].nlogical;
}

static int check_net(af_handle* h, int net, size_t n) {
  if (!h) return AF_EINVAL;
  if (net < 0 || net >= AF_MAX_NETS || !h->nets[net].used) return h->fail(AF_EINVAL, "unknown net");
  if (n != h->nets[net].nlogical) return h->fail(AF_EINVAL, "parameter count mismatch");
  return AF_OK;
}
// The ABI's flat order <-> the device buffers (params, Adam moments, gradients): a straight copy for a 256-wide net, a scatter into /
// gather out of the zero-padded physical layout for a narrower one (NetDesc::lmap).  put zeroes the padding: it must BE zero (see NetDesc::hid).
static int put_flat(af_handle* h, const NetDesc& n, float* dev, const float* flat) {
  if (n.lmap.empty()) { HCHK(hipMemcpy(dev + n.p_base, flat, n.nparams * 4, hipMemcpyHostToDevice)); return AF_OK; }
  std::vector<float> phys(n.nparams, 0.f);
  for (size_t i = 0; i < n.nlogical; ++i) phys[n.lmap[i]] = flat[i];
  HCHK(hipMemcpy(dev + n.p_base, phys.data(), n.nparams * 4, hipMemcpyHostToDevice));
  return AF_OK;
}
static int get_flat(af_handle* h, const NetDesc& n, const float* dev, float* flat) {
  if (n.lmap.empty()) { HCHK(hipMemcpy(flat, dev + n.p_base, n.nparams * 4, hipMemcpyDeviceToHost)); return AF_OK; }
  std::vector<float> phys(n.nparams);
  HCHK(hipMemcpy(phys.data(), dev + n.p_base, n.nparams * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n.nlogical; ++i) flat[i] = phys[n.lmap[i]];
  return AF_OK;
}

int af_set_params(af_handle* h, int net, const float* flat, size_t n) {
  int rc = check_net(h, net, n); if (rc) return rc;
  if (!flat) return h->fail(AF_EINVAL, "af_set_params: null");
  HCHK(hipSetDevice(h->device));
  HCHK(hipStreamSynchronize(h->stream));
  rc = put_flat(h, h->nets[net], h->params, flat); if (rc) return rc;
  rc = repack(h, h->sched[0]); if (rc) return rc;
  HCHK(hipStreamSynchronize(h->stream));
  std::fill(h->frame_sse_valid.begin(), h->frame_sse_valid.end(), 0);
  return AF_OK;
}

int af_get_params(af_handle* h, int net, float* flat, size_t n) {
  int rc = check_net(h, net, n); if (rc) return rc;
  if (!flat) return h->fail(AF_EINVAL, "af_get_params: null");
  HCHK(hipSetDevice(h->device)); HCHK(hipStreamSynchronize(h->stream));
  return get_flat(h, h->nets[net], h->params, flat);
}

int af_get_adam_state(af_handle* h, int net, float* m, float* v, int64_t* step) {
  if (!h) return AF_EINVAL;
  if (net < 0 || net >= AF_MAX_NETS || !h->nets[net].used) return h->fail(AF_EINVAL, "unknown net");
  HCHK(hipSetDevice(h->device)); HCHK(hipStreamSynchronize(h->stream));
  const NetDesc& n = h->nets[net];
  int rc = AF_OK;
  if (m && (rc = get_flat(h, n, h->adam_m, m)) != AF_OK) return rc;
  if (v && (rc = get_flat(h, n, h->adam_v, v)) != AF_OK) return rc;
  if (step) *step = h->adam_step;
  return AF_OK;
}

int af_set_adam_state(af_handle* h, int net, const float* m, const float* v, int64_t step) {
  if (!h) return AF_EINVAL;
  if (net < 0 || net >= AF_MAX_NETS || !h->nets[net].used) return h->fail(AF_EINVAL, "unknown net");
  if (step < 0) return h->fail(AF_EINVAL, "negative step");
  HCHK(hipSetDevice(h->device)); HCHK(hipStreamSynchronize(h->stream));
  const NetDesc& n = h->nets[net];
  int rc = AF_OK;
  if (m && (rc = put_flat(h, n, h->adam_m, m)) != AF_OK) return rc;
  if (v && (rc = put_flat(h, n, h->adam_v, v)) != AF_OK) return rc;
  h->adam_step = step;
  return AF_OK;
}

// Re-cut every split-K schedule (loop and pre-train) for the current arithmetic / cost row and upload it.
static int rebuild_dw_scheds(af_handle* h) {
  HCHK(hipSetDevice(h->device)); HCHK(hipStreamSynchronize(h->stream));
  bool ok = build_main_scheds(h);
  ok = ok && build_sched(h, h->sched[2], {{&h->nets[AF_NET_MAP1], tiles_of(h->cfg.pretrain_batch)}});
  if (h->seg) ok = ok && build_sched(h, h->sched[3], {{&h->nets[AF_NET_MAP2], tiles_of(h->cfg.pretrain_batch)}});
  if (!ok) return h->fail(AF_EINVAL, "dW schedule needs more than DW_MAXSEG segments per workgroup");
  for (int i = 0; i < (h->seg ? 4 : 3); ++i) {
    HCHK(upload_sched(h->sched[i]));
    if (h->sched[i].partial_floats > h->partial_cap) {
      (void)hipFree(h->partial); h->partial = nullptr; h->partial_cap = 0;
      HCHK(dalloc(&h->partial, h->sched[i].partial_floats)); h->partial_cap = h->sched[i].partial_floats;
    }
  }
  return AF_OK;
}
int af_set_dw_mode(af_handle* h, int mode) {
  if (!h) return AF_EINVAL;
  if (mode < 0 || mode > 2) return h->fail(AF_EINVAL, "af_set_dw_mode: 0 (fp32 MFMA), 1 (bf16x6) or 2 (bf16x3)");
  if (mode == h->dw_mode) return AF_OK;
  h->dw_mode = mode;                       // the tile costs of the split-K schedule belong to the arithmetic
  return rebuild_dw_scheds(h);
}
int af_debug_set_dw_cost(af_handle* h, const double* cost5, double seg_cost) {
  if (!h) return AF_EINVAL;
  if (!cost5) { if (!h->dw_cost_set) return AF_OK; h->dw_cost_set = false; h->dw_seg_cost = 0; return rebuild_dw_scheds(h); }
  for (int i = 0; i < 5; ++i)
    if (!(cost5[i] > 0.0) || !std::isfinite(cost5[i])) return h->fail(AF_EINVAL, "af_debug_set_dw_cost: every tile cost must be finite and > 0");
  if (std::isnan(seg_cost) || std::isinf(seg_cost)) return h->fail(AF_EINVAL, "af_debug_set_dw_cost: seg_cost must be finite (<= 0 keeps the shipped one)");
  const double r = *std::max_element(cost5, cost5 + 5) / *std::min_element(cost5, cost5 + 5);
  if (r > 1.0e3) return h->fail(AF_EINVAL, "af_debug_set_dw_cost: cost ratios beyond 1000:1 cut segments the DW_MAXSEG lists cannot hold");
  double keep[5]; const bool was = h->dw_cost_set; const double keep_seg = h->dw_seg_cost;
  for (int i = 0; i < 5; ++i) { keep[i] = h->dw_cost[i]; h->dw_cost[i] = cost5[i]; }
  h->dw_cost_set = true; h->dw_seg_cost = seg_cost > 0 ? seg_cost : 0;
  const int rc = rebuild_dw_scheds(h);
  if (rc != AF_OK) {                      // a row the segment lists cannot hold: back to what was there, the handle stays usable
    const std::string msg = h->err;
    for (int i = 0; i < 5; ++i) h->dw_cost[i] = keep[i];
    h->dw_cost_set = was; h->dw_seg_cost = keep_seg;
    (void)rebuild_dw_scheds(h);
    h->err = msg;
  }
  return rc;
}
int af_set_mlp_mode(af_handle* h, int mode) {
  if (!h) return AF_EINVAL;
  if (mode < 0 || mode > 3) return h->fail(AF_EINVAL, "af_set_mlp_mode: 0 (fp32 MFMA), 1 (bf16x6), 2 (bf16x6 forward, three-product bf16 backward chain) or 3 (f16x3: two-term fp16 split with a scale per row)");
  const int cls_old = h->mlp_mode == 3 ? 2 : (h->mlp_mode ? 1 : 0), cls_new = mode == 3 ? 2 : (mode ? 1 : 0);
  h->mlp_mode = mode;
  if (cls_new != cls_old && cls_new != 0) {      // another set of 16-bit streams comes into use: bring it up to date (k_adam only maintains the set in force)
    HCHK(hipSetDevice(h->device));
    const int rc = repack(h, h->sched[0]); if (rc) return rc;
    HCHK(hipStreamSynchronize(h->stream));
  }
  return AF_OK;
}
int af_get_modes(const af_handle* h, int* mlp_mode, int* dw_mode) {
  if (!h) return AF_EINVAL;
  if (mlp_mode) *mlp_mode = h->mlp_mode;
  if (dw_mode) *dw_mode = h->dw_mode;
  return AF_OK;
}
int af_set_debug(af_handle* h, int enable) { if (!h) return AF_EINVAL; h->debug = enable != 0; return AF_OK; }
int af_set_timing(af_handle* h, int class_mask) {
  if (!h) return AF_EINVAL;
  h->timing = (unsigned)class_mask & 0xFFFFu;
  const unsigned every = ((unsigned)class_mask >> 16) & 0xFFu;      // an event costs ~5 us in-stream: a caller timing a region may sample every P-th step
  h->timing_every = every ? every : 1;
  return AF_OK;
}
int af_get_timing(af_handle* h, double* ms16, int64_t* counts16, double* flops16, int reset) {
  if (!h) return AF_EINVAL;
  (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); drain_timers(h);
  for (int i = 0; i < 16; ++i) {
    if (ms16) ms16[i] = h->t_ms[i];
    if (counts16) counts16[i] = h->t_cnt[i];
    if (flops16) flops16[i] = h->t_flops[i];
    if (reset) { h->t_ms[i] = 0; h->t_cnt[i] = 0; h->t_flops[i] = 0; }
  }
  return AF_OK;
}

int af_get_last_grads(af_handle* h, int net, float* flat, size_t n) {
  int rc = check_net(h, net, n); if (rc) return rc;
  HCHK(hipSetDevice(h->device)); HCHK(hipStreamSynchronize(h->stream));
  return get_flat(h, h->nets[net], h->grads, flat);
}

int af_loss_width(const af_handle* h) { return h && h->seg ? 16 : 8; }

int af_pretrain(af_handle* h, int net, int pretrain_iters, const int64_t* ys, const int64_t* xs, uint64_t seed, float* losses_out) {
  if (!h) return AF_EINVAL;
  if (!(net == AF_MAPPING1 || (net == AF_MAPPING2 && h->seg))) return h->fail(AF_EINVAL, "af_pretrain: net must be a mapping net of this handle");
  if (pretrain_iters < 0 || ((ys == nullptr) != (xs == nullptr))) return h->fail(AF_EINVAL, "af_pretrain: arguments");
  HCHK(hipSetDevice(h->device));
  const int F = h->cfg.number_of_frames, NB = h->cfg.pretrain_batch;
  const size_t steps = (size_t)pretrain_iters * F;
  if (steps == 0) return AF_OK;
  NetDesc& M = h->nets[net];
  Sched& sc = h->sched[net == AF_MAPPING1 ? 2 : 3];
  int rc = ensure_loss_log(h, steps); if (rc) return rc;
  int64_t *d_ys = nullptr, *d_xs = nullptr;
  if (ys) {
    HCHK(dalloc(&d_ys, steps * NB)); HCHK(dalloc(&d_xs, steps * NB));
    HCHK(hipMemcpy(d_ys, ys, steps * NB * 8, hipMemcpyHostToDevice));
    HCHK(hipMemcpy(d_xs, xs, steps * NB * 8, hipMemcpyHostToDevice));
  }
  const int NT = tiles_of(NB);
  HCHK(hipMemsetAsync(M.dout, 0, (size_t)M.nt_cap * 32 * 16, h->stream));
  HCHK(hipMemsetAsync(h->pre_m, 0, h->total_params * 4, h->stream));     // pre_train_mapping builds its own Adam (unwrap_utils.py:178)
  HCHK(hipMemsetAsync(h->pre_v, 0, h->total_params * 4, h->stream));
  h->cur_nseg = 0;
  const float half_main = (float)(std::max(h->cfg.resx, h->cfg.resy) / 2.0);
  size_t s = 0;
  for (int it = 0; it < pretrain_iters && rc == 0; ++it)
    for (int f = 0; f < F && rc == 0; ++f, ++s) {
      PrePrepArgs p{};
      p.ys = d_ys ? d_ys + s * NB : nullptr; p.xs = d_xs ? d_xs + s * NB : nullptr;
      p.seed = seed; p.iter = (uint32_t)s; p.N = NB; p.resx = h->cfg.resx; p.resy = h->cfg.resy;
      p.half_main = half_main; p.t = (float)((double)f / (F / 2.0) - 1.0);
      p.coords = M.coords; p.x0_tile = M.x0_tile;
      if (af_launch_pre_prep(&p, h->stream)) { rc = h->fail(AF_EHIP, "pre_prep"); break; }
      // The batch is smaller than one round of the chip, so a step is bound by the LATENCY of one tile chain.  With the f16x3 arithmetic the 32-row chains
      // (mlphf.hip: 79 workgroups, a 128-row task in ~45 us at the idle chip's clock) are shorter than the 16-row fp32-MFMA chains built for this case
      // (mlp16.hip: half the rows per wave, 74 us per direction); the other arithmetics keep those.
      const bool hf = h->mlp_mode == 3;
      if (hf) { if ((rc = launch_fwd(h, T_FWD_1, {{net, fwd_args(h, M, M.coords, M.out_buf, NT, true, true), NB}}, true)) != 0) break; }
      else { Timer t(h, T_FWD_1, (double)NB * h->flop_fwd[net]); const FwdArgs fa = fwd_args(h, M, M.coords, M.out_buf, NT, true, false);
        if (af_launch_fwd16(M.kern, &fa, h->stream)) { rc = h->fail(AF_EHIP, "fwd16"); break; } }
      PreLossArgs l{M.coords, M.out_buf, M.dout, h->loss_part, NB, h->cfg.uv_mapping_scale};
      if (af_launch_pre_loss(&l, h->stream)) { rc = h->fail(AF_EHIP, "pre_loss"); break; }
      if (hf) { if ((rc = launch_bwd(h, T_BWD_2, {{net, bwd_args(h, M, NT, true), NB}})) != 0) break; }
      else { Timer t(h, T_BWD_2, (double)NB * h->flop_dx[net]); const BwdArgs ba = bwd_args(h, M, NT, false);
        if (af_launch_bwd16(M.kern, &ba, h->stream)) { rc = h->fail(AF_EHIP, "bwd16"); break; } }
      rc = finish_step(h, sc, h->pre_m, h->pre_v, (long long)s + 1, h->loss_log + s * AF_LOSS_W, (NB + 255) / 256, (double)NB * h->flop_fwd[net], false);
    }
  hipError_t e = hipStreamSynchronize(h->stream);
  drain_timers(h);
  if (d_ys) { (void)hipFree(d_ys); (void)hipFree(d_xs); }
  if (rc) return rc;
  if (e != hipSuccess) return h->fail(AF_EHIP, "af_pretrain sync", e);
  { int dev_nan = 0; int r2 = take_nan_flag(h, dev_nan); if (r2) return r2; if (dev_nan & 1) return h->fail(AF_ENAN, "af_pretrain: NaN loss or non-finite parameter");
    if ((dev_nan & 2) && h->mlp_mode == 3) return h->fail(AF_ERANGE, "af_pretrain: a hidden-layer weight reached |w| >= 8, beyond what the fp16 weight images of af_set_mlp_mode(h, 3) are scaled for; use mode 1 (bf16x6)"); }
  if (losses_out) {
    std::vector<float> tmp(steps * AF_LOSS_W);
    HCHK(hipMemcpy(tmp.data(), h->loss_log, steps * AF_LOSS_W * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < steps; ++i) losses_out[i] = tmp[i * AF_LOSS_W] / (float)NB;
  }
  std::fill(h->frame_sse_valid.begin(), h->frame_sse_valid.end(), 0);
  return AF_OK;
}

int af_train_steps(af_handle* h, int first_iter, int n_iters, const int64_t* inds, uint64_t seed, float* losses_out) {
  if (!h) return AF_EINVAL;
  if (!h->have_video) return h->fail(AF_ESTATE, "af_train_steps: no video uploaded");
  if (n_iters < 0 || first_iter < 0) return h->fail(AF_EINVAL, "af_train_steps: arguments");
  if (n_iters == 0) return AF_OK;
  HCHK(hipSetDevice(h->device));
  const af_config& c = h->cfg;
  const int N = h->N;
  int rc = ensure_loss_log(h, n_iters); if (rc) return rc;
  int64_t* d_inds = nullptr;
  if (inds) {
    HCHK(dalloc(&d_inds, (size_t)n_iters * N));
    HCHK(hipMemcpy(d_inds, inds, (size_t)n_iters * N * 8, hipMemcpyHostToDevice));
  }
  HCHK(hipMemsetAsync(h->counts, 0, 8, h->stream));
  for (int k = 0; k < n_iters && rc == 0; ++k) {
    const int64_t* di = d_inds ? d_inds + (size_t)k * N : nullptr;
    float* lo = h->loss_log + (size_t)k * AF_LOSS_W;
    h->timing_live = (k % (int)h->timing_every) == 0;
    rc = h->seg ? enqueue_seg_step(h, first_iter + k, di, seed, lo) : enqueue_single_step(h, first_iter + k, di, seed, lo);
  }
  h->timing_live = true;
  hipError_t e = hipStreamSynchronize(h->stream);
  drain_timers(h);
  if (d_inds) (void)hipFree(d_inds);
  if (rc) return rc;
  if (e != hipSuccess) return h->fail(AF_EHIP, "af_train_steps sync", e);
  std::fill(h->frame_sse_valid.begin(), h->frame_sse_valid.end(), 0);
  int dev_nan = 0;
  if ((rc = take_nan_flag(h, dev_nan)) != 0) return rc;
  if (losses_out) {
    std::vector<float> tmp((size_t)n_iters * AF_LOSS_W);
    HCHK(hipMemcpy(tmp.data(), h->loss_log, (size_t)n_iters * AF_LOSS_W * 4, hipMemcpyDeviceToHost));
    bool nan = false;
    const float invN = 1.f / (float)N;
    for (int k = 0; k < n_iters; ++k) {
      const int i = first_iter + k;
      const bool glob = glob_on(c, i);
      const float* s = &tmp[(size_t)k * AF_LOSS_W];
      const float nf = s[AF_LOSS_W - 2], nb = s[AF_LOSS_W - 1];
      float total;
      // mean over an empty set is NaN in the reference (loss_utils.py:317-320)
      if (!h->seg) {
        float* o = losses_out + (size_t)k * 8;
        o[0] = s[0] * invN; o[1] = c.use_gradient_loss ? s[1] * invN : 0.f; o[2] = s[2] * invN; o[3] = glob ? s[3] * invN : 0.f;
        o[4] = 0.5f * (s[5] / nb) + 0.5f * (s[4] / nf);
        o[5] = c.rigidity_coeff * o[2] + (glob ? c.global_rigidity_coeff_fg * o[3] : 0.f) + c.rgb_coeff * o[0] + c.optical_flow_coeff * o[4] + c.gradient_loss_coeff * o[1];
        o[6] = nf; o[7] = nb;
        total = o[5];
      } else {
        float* o = losses_out + (size_t)k * 16;
        const float boot = i > c.stop_bootstrapping_iteration ? 0.f : c.alpha_bootstrapping_factor;
        o[0] = s[0] * invN; o[1] = c.use_gradient_loss ? s[1] * invN : 0.f; o[2] = s[2] * invN; o[3] = s[3] * invN;
        o[4] = glob ? s[4] * invN : 0.f; o[5] = glob ? s[5] * invN : 0.f;
        o[6] = 0.5f * (s[7] / nb) + 0.5f * (s[6] / nf);
        o[7] = 0.5f * (s[9] / nb) + 0.5f * (s[8] / nf);
        o[8] = (s[10] / nf + s[11] / nb) * 0.5f;
        o[9] = s[12] * invN; o[10] = s[13] * invN;
        o[11] = c.rigidity_coeff * (o[2] + o[3]) + c.rgb_coeff * o[0] + c.optical_flow_coeff * (o[6] + o[7]) + boot * o[9]
              + c.alpha_flow_factor * o[8] + c.sparsity_coeff * o[10] + c.gradient_loss_coeff * o[1]
              + (glob ? c.global_rigidity_coeff_fg * o[4] + c.global_rigidity_coeff_bg * o[5] : 0.f);
        o[12] = nf; o[13] = nb; o[14] = 0.f; o[15] = 0.f;
        total = o[11];
      }
      if (!(total == total)) nan = true;
    }
    if (nan) return h->fail(AF_ENAN, "af_train_steps: NaN loss (a batch without valid flow pixels, as in the reference, or divergence)");
  }
  // also without a loss buffer: a non-finite parameter, a NaN loss term or an empty flow-match set seen by k_adam
  if (dev_nan & 1) return h->fail(AF_ENAN, "af_train_steps: NaN loss or non-finite parameter (a batch without valid flow pixels, as in the reference, or divergence)");
  if ((dev_nan & 2) && h->mlp_mode == 3) return h->fail(AF_ERANGE, "af_train_steps: a hidden-layer weight reached |w| >= 8, beyond what the fp16 weight images of af_set_mlp_mode(h, 3) are scaled for; use mode 1 (bf16x6)");
  return AF_OK;
}

int af_step_work(const af_handle* h, int iter, int64_t rows4[4], double* flops) {
  if (!h) return AF_EINVAL;
  const int64_t nseg = glob_on(h->cfg, iter) ? 9 : 7, N = h->N;
  int64_t r[4] = {nseg * N, (h->seg ? 6 : 3) * N, h->seg ? nseg * N : 0, h->seg ? 5 * N : 0};
  double f = 0;
  for (int i = 0; i < 4; ++i) { if (rows4) rows4[i] = r[i]; f += (double)r[i] * (2.0 * h->flop_fwd[i] + h->flop_dx[i]); }
  if (flops) *flops = f;
  return AF_OK;
}

static int ensure_render(af_handle* h, int rows) {
  if (rows <= h->render_rows_cap) return 0;
  (void)hipFree(h->r_coords); (void)hipFree(h->r_uv); (void)hipFree(h->r_uv2); (void)hipFree(h->r_al); (void)hipFree(h->r_t); (void)hipFree(h->r_rgb); (void)hipFree(h->r_sse);
  h->r_coords = h->r_uv = h->r_uv2 = h->r_al = h->r_t = h->r_rgb = nullptr; h->r_sse = nullptr; h->render_rows_cap = 0;
  const size_t rp = (size_t)tiles_of(rows) * 32;
  HCHK(dalloc(&h->r_coords, rp * 4)); HCHK(dalloc(&h->r_uv, rp * 4)); HCHK(dalloc(&h->r_t, rp * 4 * (h->seg ? 2 : 1)));
  if (h->seg) { HCHK(dalloc(&h->r_uv2, rp * 4)); HCHK(dalloc(&h->r_al, rp * 4)); }
  HCHK(dalloc(&h->r_rgb, rp * 3)); HCHK(dalloc(&h->r_sse, (rp + 255) / 256));
  h->render_rows_cap = rows;
  return 0;
}

int af_debug_plan(int ncu, int rows_map, int rows_atlas, int dep_rows, int out3[3]) {
  if (!out3 || ncu <= 0) return AF_EINVAL;
  const int NT_map = tiles_of(rows_map), NT_atlas = tiles_of(rows_atlas);
  plan_mapping_split(ncu, NT_map, NT_atlas, dep_rows, out3[0], out3[1]);
  out3[2] = NT_map;
  return AF_OK;
}

int af_debug_dw_clocks(af_handle* h, int enable, uint64_t* out, int cap_wg) {
  if (!h) return AF_EINVAL;
  HCHK(hipSetDevice(h->device)); HCHK(hipStreamSynchronize(h->stream));
  if (enable && !h->dw_clock) { HCHK(dalloc(&h->dw_clock, (size_t)h->ncu * 2)); HCHK(hipMemset(h->dw_clock, 0, (size_t)h->ncu * 16)); }
  if (out && h->dw_clock) HCHK(hipMemcpy(out, h->dw_clock, (size_t)std::min(cap_wg, h->ncu) * 16, hipMemcpyDeviceToHost));
  if (!enable && h->dw_clock) { (void)hipFree(h->dw_clock); h->dw_clock = nullptr; }
  return h->ncu;
}

int af_debug_step_clocks(af_handle* h, int enable, uint64_t* out, int cap_wg) {
  if (!h) return AF_EINVAL;
  HCHK(hipSetDevice(h->device)); HCHK(hipStreamSynchronize(h->stream));
  const size_t n = (size_t)5 * AF_STAMP_WG * 4;
  if (enable && !h->step_stamp) { HCHK(dalloc(&h->step_stamp, n)); HCHK(hipMemset(h->step_stamp, 0, n * 8)); }
  if (out && h->step_stamp) {
    const int w = std::min(cap_wg, AF_STAMP_WG);
    for (int l = 0; l < 5; ++l) HCHK(hipMemcpy(out + (size_t)l * w * 4, h->step_stamp + (size_t)l * AF_STAMP_WG * 4, (size_t)w * 32, hipMemcpyDeviceToHost));
  }
  if (h->step_stamp) HCHK(hipMemset(h->step_stamp, 0, n * 8));      // a launch smaller than the last one leaves zeros, not old stamps
  if (!enable && h->step_stamp) { (void)hipFree(h->step_stamp); h->step_stamp = nullptr; }
  return AF_STAMP_WG;      // the cap, NOT the grid size: workgroups beyond it are not stamped (the kernels test blockIdx.x < AF_STAMP_WG), rows of zeros are workgroups a launch did not have
}

int af_debug_dw_schedule(af_handle* h, int which, int32_t* out, int cap_wg) {
  // out [min(cap_wg, nwg)][DW_MAXSEG][4] = {job shape (DW_* enum), first row tile, one past the last, job index}, shape -1 = end of list;
  // segments of compacted jobs end where k_dw's last launch clipped them (the live row tiles of the last iteration)
  if (!h || which < 0 || which > 3) return AF_EINVAL;
  const Sched& sc = h->sched[which];
  int live = 0;
  if (out) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); (void)hipMemcpy(&live, h->live, 4, hipMemcpyDeviceToHost); }
  if (out) {
    for (int w = 0; w < std::min(cap_wg, sc.nwg); ++w)
      for (int k = 0; k < DW_MAXSEG; ++k) {
        const DwSeg& sg = sc.segs[(size_t)w * DW_MAXSEG + k];
        int32_t* o = out + ((size_t)w * DW_MAXSEG + k) * 4;
        o[0] = sg.job < 0 ? -1 : sc.jobs[sg.job].shape; o[1] = sg.t0; o[2] = sg.t1; o[3] = sg.job;
        if (sg.job >= 0 && sc.jobs[sg.job].live_rows) o[2] = std::max(sg.t0, std::min(sg.t1, tiles_of(sc.jobs[sg.job].live_base + live)));
      }
  }
  return sc.nwg;
}

int af_debug_records(af_handle* h, const int64_t* inds, int n, float* out) {
  if (!h || !inds || !out || n <= 0) return AF_EINVAL;
  if (!h->have_video) return h->fail(AF_ESTATE, "af_debug_records: no video uploaded");
  HCHK(hipSetDevice(h->device)); HCHK(hipStreamSynchronize(h->stream));
  const int64_t P = (int64_t)h->cfg.resx * h->cfg.resy * h->cfg.number_of_frames;
  for (int i = 0; i < n; ++i) {
    if (inds[i] < 0 || inds[i] >= P) return h->fail(AF_EINVAL, "af_debug_records: index out of range");
    HCHK(hipMemcpy(out + (size_t)i * AF_REC_F, h->table + (size_t)inds[i] * AF_REC_F, AF_REC_F * 4, hipMemcpyDeviceToHost));
  }
  return AF_OK;
}

int af_debug_forward(af_handle* h, int net, const float* in, int rows, float* out) {
  if (!h || !in || !out || rows <= 0) return AF_EINVAL;
  if (net < 0 || net >= AF_MAX_NETS || !h->nets[net].used) return h->fail(AF_EINVAL, "unknown net");
  HCHK(hipSetDevice(h->device));
  int rc = ensure_render(h, rows); if (rc) return rc;
  const int NT = tiles_of(rows);
  HCHK(hipMemsetAsync(h->r_coords, 0, (size_t)NT * 32 * 16, h->stream));
  HCHK(hipMemcpyAsync(h->r_coords, in, (size_t)rows * 16, hipMemcpyHostToDevice, h->stream));
  FwdArgs fa = fwd_args(h, h->nets[net], h->r_coords, h->r_uv, NT, false, h->mlp_mode != 0);
  if (h->nets[net].in_kind != AF_IN_XYT) { fa.in_scale = 1.f; fa.in_shift0 = 0.f; fa.in_shift1 = 0.f; }
  { int rc2 = launch_fwd(h, T_FWD_1, {{net, fa, rows}}, false); if (rc2) return rc2; }
  HCHK(hipMemcpyAsync(out, h->r_uv, (size_t)rows * 16, hipMemcpyDeviceToHost, h->stream));
  HCHK(hipStreamSynchronize(h->stream));
  return AF_OK;
}

int af_debug_tiles(af_handle* h, int net, int which, int layer, int nt_stride, int tile0, int ntiles, float* out) {
  if (!h || !out || ntiles <= 0 || tile0 < 0 || layer < 0 || nt_stride <= 0) return AF_EINVAL;
  if (net < 0 || net >= AF_MAX_NETS || !h->nets[net].used) return h->fail(AF_EINVAL, "unknown net");
  const NetDesc& n = h->nets[net];
  if (tile0 + ntiles > nt_stride || nt_stride > n.nt_cap) return h->fail(AF_EINVAL, "af_debug_tiles: tile range");
  const float* src = nullptr; size_t per = 0; bool planes = false;
  switch (which) {
    case 0: src = n.acts; per = AF_TILE_F; planes = true; break;                 // plane l = relu(Z_l) = X_{l+1}, T-layout [256][32]
    case 1: src = n.dz; per = AF_TILE_F; planes = true; break;                   // plane l = dZ_l
    case 2: src = (const float*)n.masks; per = 64 * 4; planes = true; break;     // plane l = sign bits of X_{l+1}, [64 lanes][4 words]
    case 3: src = n.pe_tile; per = 2048; break;                                  // PE features [64][32]
    case 4: src = n.dz_last; per = 1024; break;                                  // dZ of the output layer [32][32]
    case 5: src = n.x0_tile; per = 1024; break;                                  // xyt rows [32][32] (mapping nets without PE)
    default: return h->fail(AF_EINVAL, "af_debug_tiles: which");
  }
  if (!src || (which == 3 && !n.pe_feats)) return h->fail(AF_EINVAL, "af_debug_tiles: this net has no such tensor");
  if (planes && layer >= n.NL - 1) return h->fail(AF_EINVAL, "af_debug_tiles: layer");
  HCHK(hipSetDevice(h->device)); HCHK(hipStreamSynchronize(h->stream));
  const size_t off = ((planes ? (size_t)layer * nt_stride : 0) + tile0) * per;
  HCHK(hipMemcpy(out, src + off, (size_t)ntiles * per * 4, hipMemcpyDeviceToHost));
  return AF_OK;
}

int af_render_frame(af_handle* h, int frame, float* rgb_out, double* sse_out) {
  if (!h) return AF_EINVAL;
  if (frame < 0 || frame >= h->cfg.number_of_frames) return h->fail(AF_EINVAL, "af_render_frame: frame index");
  if (!h->have_video) return h->fail(AF_ESTATE, "af_render_frame: no video uploaded");
  HCHK(hipSetDevice(h->device));
  const int npix = h->cfg.resx * h->cfg.resy;
  int rc = ensure_render(h, npix); if (rc) return rc;
  const int NT = tiles_of(npix), F = h->cfg.number_of_frames;
  const float half_main = (float)(std::max(h->cfg.resx, h->cfg.resy) / 2.0);
  const float t = (float)((double)frame / (F / 2.0) - 1.0);     // evaluate.py:656 computes t in Python floats
  LCHK(af_launch_frame_coords(h->r_coords, h->cfg.resx, h->cfg.resy, half_main, t, NT * 32, h->stream));
  FwdArgs fm = fwd_args(h, h->nets[AF_NET_MAP1], h->r_coords, h->r_uv, NT, false, h->mlp_mode != 0);
  if (!h->seg) {
    if ((rc = launch_fwd(h, T_FWD_1, {{AF_NET_MAP1, fm, npix}}, false)) != 0) return rc;
    FwdArgs fa = fwd_args(h, h->nets[AF_NET_ATLAS], h->r_uv, h->r_t, NT, false, h->mlp_mode != 0);
    if ((rc = launch_fwd(h, T_FWD_2, {{AF_NET_ATLAS, fa, npix}}, false)) != 0) return rc;
    LCHK(af_launch_frame_finish(h->r_t, h->table, h->r_rgb, h->r_sse, npix, (size_t)frame * npix, h->stream));
  } else {   // evaluate.py:302-337
    FwdArgs f2 = fwd_args(h, h->nets[AF_NET_MAP2], h->r_coords, h->r_uv2, NT, false, h->mlp_mode != 0);
    FwdArgs fl = fwd_args(h, h->nets[AF_NET_ALPHA], h->r_coords, h->r_al, NT, false, h->mlp_mode != 0);
    if ((rc = launch_fwd(h, T_FWD_1, {{AF_NET_ALPHA, fl, npix}, {AF_NET_MAP1, fm, npix}, {AF_NET_MAP2, f2, npix}}, false)) != 0) return rc;
    FwdArgs fa = fwd_args(h, h->nets[AF_NET_ATLAS], h->r_uv, h->r_t, 2 * NT, false, h->mlp_mode != 0);
    fa.in1 = h->r_uv2; fa.split_row = NT * 32;
    if ((rc = launch_fwd(h, T_FWD_2, {{AF_NET_ATLAS, fa, 2 * NT * 32}}, false)) != 0) return rc;
    LCHK(af_launch_frame_finish_seg(h->r_t, h->r_al, (size_t)NT * 32, h->table, h->r_rgb, h->r_sse, npix, (size_t)frame * npix, h->stream));
  }
  const int nblk = (npix + 255) / 256;
  std::vector<double> part(nblk);
  HCHK(hipMemcpyAsync(part.data(), h->r_sse, (size_t)nblk * 8, hipMemcpyDeviceToHost, h->stream));
  if (rgb_out) HCHK(hipMemcpyAsync(rgb_out, h->r_rgb, (size_t)npix * 12, hipMemcpyDeviceToHost, h->stream));
  HCHK(hipStreamSynchronize(h->stream));
  double sse = 0; for (double v : part) sse += v;
  h->frame_sse[frame] = sse; h->frame_sse_valid[frame] = 1;
  if (sse_out) *sse_out = sse;
  return AF_OK;
}

int af_psnr(af_handle* h, double* mean_psnr, double* per_frame) {
  if (!h) return AF_EINVAL;
  const int F = h->cfg.number_of_frames; const double cnt = (double)h->cfg.resx * h->cfg.resy * 3.0;
  double acc = 0;
  for (int f = 0; f < F; ++f) {
    if (!h->frame_sse_valid[f]) { int rc = af_render_frame(h, f, nullptr, nullptr); if (rc) return rc; }
    const double mse = h->frame_sse[f] / cnt;
    const double p = 10.0 * log10(1.0 / mse);
    if (per_frame) per_frame[f] = p;
    acc += p;
  }
  if (mean_psnr) *mean_psnr = acc / F;
  return AF_OK;
}

}  // extern "C"
