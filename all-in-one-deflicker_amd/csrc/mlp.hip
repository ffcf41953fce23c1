// mlp.hip — register-chained fused coordinate-MLP kernels for gfx950 (forward, and backward dX chain).
//
// Replaces, for the stage-1 hot path, every `model_F_*(x)` module call and the dX half of
// `loss.backward()` of the reference (src/stage1_neural_atlas.py:174,181,230;
// src/models/stage_1/loss_utils.py:154-159,235,305,312; IMLP.forward in
// src/models/stage_1/implicit_neural_networks.py:62-80).
//
// One wavefront owns 32 rows.  A layer is evaluated transposed, Y^T = W * X^T, with
// v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains): A = packed weight image read from LDS with one
// ds_read_b128 per four MFMA steps, B = the previous layer's output, which is ALREADY in the right
// registers (see "C-layout" in af_dev.h).  The four waves of a workgroup share the weight stream, which is
// double-buffered in LDS in 64 KB chunks by global_load_lds (one barrier per chunk, 16 K MFMA cycles apart).
// LDS: 2 x 64 KB weight buffers + 8 KB bias rows.  Registers: 128 (activations) + 128 (accumulators, AGPR) + 64
// (A fragments) -> 1 wave / SIMD.  One kernel carries the chains of all four nets (k_mlp_*_multi, below): a launch is a
// list of independent (net, row-tile range) parts.
#include "mlp_common.h"

template <class NS, bool TRAIN, bool HID>      // HID: see mlp_fwd_body_bf (mlpbf.hip)
AF_DEV void mlp_fwd_body(const FwdArgs& a, int wg, char* smem) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, h = lane >> 5;
  int tile = a.tile0 + wg * 4 + wave;
  const int NT = live_tiles(a);
  if (a.tile0 + wg * 4 >= NT) return;                  // a workgroup of rows that do not exist this iteration (uniform: before any barrier)
  const bool live = tile < NT;
  if (!live) tile = NT - 1;
  const int row = tile * 32 + j;

  using CB = ChunkBytes<NS>;
  ChunkStream cs{nullptr, smem, wave, 0, nullptr, 0};
  cs.start(a.wimg, tid);

  const int nl = a.nl;
  stage_bias(nl, a.bias, smem + AF_BIAS_LDS, tid);
  constexpr int NPE = NS::PEG > 0 ? NS::PEG * 4 : 4;
  float pe[NPE];            // first-layer / skip B operand (PE features, or xyt for the mapping nets)
  {
    const f32x4 v = row < a.split_row ? *(const f32x4*)(a.in + (size_t)row * 4) : *(const f32x4*)(a.in1 + (size_t)(row - a.split_row) * 4);
    if constexpr (NS::IN == AF_IN_XYT) {
#pragma unroll
      for (int p = 0; p < 4; ++p) pe[p] = (h == 0 && p < 3) ? v[p] : 0.f;
    } else if constexpr (NS::IN == AF_IN_PE2) {
      const float sh = row < a.split_row ? a.in_shift0 : a.in_shift1;
      const float x0 = v[0] * a.in_scale + sh, x1 = v[1] * a.in_scale + sh;
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        const float b = h ? __builtin_ldexpf(3.14159265358979323846f, 2 * g + 1) : __builtin_ldexpf(3.14159265358979323846f, 2 * g);
        const float p0 = x0 * b, p1 = x1 * b;
        pe[g * 4 + 0] = sinf(p0); pe[g * 4 + 1] = sinf(p1); pe[g * 4 + 2] = cosf(p0); pe[g * 4 + 3] = cosf(p1);
      }
    } else {   // AF_IN_PE3: lane half h owns k in {2h, 2h+1} (+ sin/cos triple of k = 4)
      const float x[3] = {v[0], v[1], v[2]};
      const float bA = __builtin_ldexpf(3.14159265358979323846f, 2 * h), bB = __builtin_ldexpf(3.14159265358979323846f, 2 * h + 1);
      const float b4 = __builtin_ldexpf(3.14159265358979323846f, 4);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        pe[d] = sinf(x[d] * bA); pe[3 + d] = cosf(x[d] * bA);
        pe[6 + d] = sinf(x[d] * bB); pe[9 + d] = cosf(x[d] * bB);
        pe[12 + d] = h ? cosf(x[d] * b4) : sinf(x[d] * b4);
      }
      pe[15] = 0.f;
    }
    if constexpr (TRAIN && NS::PEG > 0) {
      if (live) {   // PE features in reference feature order, T-layout [64][32], for the dW GEMMs
        const auto r = af_rsrc_uniform(a.pe_tile + (size_t)tile * 64 * 32, 64 * 32 * 4);
        if constexpr (NS::IN == AF_IN_PE2) {
#pragma unroll
          for (int g = 0; g < 5; ++g)
#pragma unroll
            for (int p = 0; p < 4; ++p) af_bs32(pe[g * 4 + p], r, (4 * h * 32 + j) * 4, (8 * g + p) * 128);
        } else {
#pragma unroll
          for (int rho = 0; rho < 15; ++rho) {
            if (rho < 12) af_bs32(pe[rho], r, (12 * h * 32 + j) * 4, rho * 128);
            else          af_bs32(pe[rho], r, (3 * h * 32 + j) * 4, (24 + rho - 12) * 128);
          }
        }
      }
    }
  }

  const int a_off8 = (h * 256 + j) * 16;     // lane offset inside a Mpad=256 image chunk
  const int voff_t = (4 * h * 32 + j) * 4;
  f32x16 acc[8];
  float in[128];
  TileStore ts{af_rsrc(a.acts, 0), voff_t};

  auto relu_out = [&](int l) {               // acc -> in[] = relu(Z_l) = X_{l+1}; its stores are deferred
    // ReLU sign bits, 32 per word: element e = (T&1)*16 + r of word T>>1 sits at bit 31-e.  One v_alignbit
    // shifts the sign of (0 - v) in: set exactly when v > 0 (0 - (+0) = +0), no VCC round trip.
    uint32_t mk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = af_relu(acc[T][r]);
        in[T * 16 + r] = v;
        if (TRAIN) mk[T >> 1] = __builtin_amdgcn_alignbit(mk[T >> 1], __builtin_bit_cast(uint32_t, 0.f - v), 31);
      }
    if constexpr (TRAIN) {
      if (live) {
        u32x4 m4 = {mk[0], mk[1], mk[2], mk[3]};
        *(u32x4*)(a.masks + (((size_t)l * a.nt_stride + tile) * 64 + lane) * 4) = m4;
      }
      ts.r = af_rsrc_uniform(a.acts + ((size_t)l * a.nt_stride + tile) * AF_TILE_F, live ? AF_TILE_F * 4 : 0);
    }
    // the element-wise ops above are inline asm (af_relu / bf_mask_keep): gfx950 needs TWO wait states between a VALU write of a VGPR and an MFMA
    // reading it as SrcA / SrcB (tools/hazardprobe.hip); hipcc pads its own VALU ops and cannot see these.  Left to the scheduler they sink to just in
    // front of the output layer's MFMAs (a two-layer net has nothing else behind them): stale operands, a corrupted chain.  isa_check.py rule (d)
    // proves on every build that no such pair exists
    AF_ELEMWISE_FENCE();
  };
  auto hook_dma = [&](auto gi) { if constexpr (decltype(gi)::value < 8) cs.issue2(); };
  auto hook_dma_store = [&](auto gi) {
    if constexpr (decltype(gi)::value < 8) cs.issue2();
    if constexpr (TRAIN) ts.template part<decltype(gi)::value>(in);
  };

  // ---- layer 0
  {
    const char* buf = cs.next<CB::L0>();            // its barrier also publishes the bias rows
    init_bias(acc, smem + AF_BIAS_LDS, 0, h);
    mm_block<8, NS::K0G, 0, 4>(acc, pe, buf + a_off8, hook_dma);
  }
  relu_out(0);

  // ---- hidden layers 1 .. NL-2
  if constexpr (HID) {
  int l = 1;
  do {
    init_bias(acc, smem + AF_BIAS_LDS, l, h);
    { const char* buf = cs.next<CB::HID>(); mm_block<8, 8, 0, 4>(acc, in, buf + a_off8, hook_dma_store); }
    { const char* buf = cs.next<CB::HID>(); mm_block<8, 8, 32, 4>(acc, in, buf + a_off8, hook_dma); }
    { const char* buf = cs.next<CB::HID>(); mm_block<8, 8, 64, 4>(acc, in, buf + a_off8, hook_dma); }
    { const char* buf = cs.next<CB::HID>(); mm_block<8, 8, 96, 4>(acc, in, buf + a_off8, hook_dma); }
    if constexpr (NS::SKIP != 0) {
      if ((NS::SKIP >> l) & 1) { const char* buf = cs.next<CB::SKIP>(); mm_block<8, NS::PEG, 0, 4>(acc, pe, buf + a_off8, hook_dma); }
    }
    relu_out(l);
  } while (++l <= nl - 2);
  }

  // ---- output layer (1..3 real outputs), tanh.  A 32-wide MFMA tile would spend 128+ full-rate MFMAs on 2 or 3
  // useful rows (3 % of the whole chain); v_mfma_f32_4x4x1_16B_f32 does the same dot products in 4-output blocks:
  // lane l = block (l >> 2) = (k-half h, row quad), column l & 3 = row within the quad, so the B operand is the
  // activation register as it stands (lane = row, register = feature 8g+4h+p) and the A operand is W[l & 3][8g+4h+p]
  // — the usual packed image with Mpad = 4.  Each k-half accumulates its own partial; one cross-half shuffle adds them.
  {
    const char* buf = cs.next_rt(CB::last_bytes(nl));
    if constexpr (TRAIN) {     // the last hidden layer's activation tile (after the barrier: its wait must not cover them)
      ts.template part<0>(in); ts.template part<1>(in); ts.template part<2>(in); ts.template part<3>(in);
      ts.template part<4>(in); ts.template part<5>(in); ts.template part<6>(in); ts.template part<7>(in);
    }
    const char* al = buf + (h * 4 + (lane & 3)) * 16;
    f32x4 o4[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) o4[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      const f32x4 w = *(const f32x4*)(al + g * 2 * 4 * 16);
#pragma unroll
      for (int p = 0; p < 4; ++p) o4[p] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[p], in[4 * g + p], o4[p], 0, 0, 0);
    }
    if constexpr (NS::SKIP != 0) {
      if (CB::out_skip(nl)) {
#pragma unroll
        for (int g = 0; g < NS::PEG; ++g) {
          const f32x4 w = *(const f32x4*)(al + (32 + g) * 2 * 4 * 16);
#pragma unroll
          for (int p = 0; p < 4; ++p) o4[p] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[p], pe[4 * g + p], o4[p], 0, 0, 0);
        }
      }
    }
    const f32x4 bias = *(const f32x4*)(smem + AF_BIAS_LDS + (nl - 1) * AF_HID * 4);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float z = (o4[0][i] + o4[1][i]) + (o4[2][i] + o4[3][i]);
      z += __shfl_xor(z, 32);
      o[i] = i < NS::OUT ? tanhf(z + bias[i]) : 0.f;
    }
    if (live && h == 0) *(f32x4*)(a.out + (size_t)row * 4) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Backward dX chain: dZ_{l-1} = (W_l^T dZ_l) . relu'(Z_{l-1}); writes every dZ_l in T-layout for the dW
// GEMMs (dw.hip).  For the atlas net the chain continues through layer 0 into the positional encoding
// and accumulates dL/d(uv) onto the mapping net's output gradient (the detached skip inputs carry no
// gradient: implicit_neural_networks.py:69).
template <class NS>
AF_DEV void mlp_bwd_body(const BwdArgs& a, int wg, char* smem) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, h = lane >> 5;
  int tile = a.tile0 + wg * 4 + wave;
  const int NT = live_tiles(a);
  if (a.tile0 + wg * 4 >= NT) return;                  // a workgroup of rows that do not exist this iteration (uniform: before any barrier)
  const bool live = tile < NT;
  if (!live) tile = NT - 1;
  const int row = tile * 32 + j;

  using CB = ChunkBytes<NS>;
  ChunkStream cs{nullptr, smem, wave, 0, nullptr, 0};
  cs.start(a.wimg, tid);
  const int nl = a.nl;

  float dzl[4];
  {
    const f32x4 o = *(const f32x4*)(a.out + (size_t)row * 4);
    const f32x4 d = *(const f32x4*)(a.dout + (size_t)row * 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) dzl[p] = (h == 0 && p < NS::OUT) ? d[p] * (1.f - o[p] * o[p]) : 0.f;
    if (live && h == 0) {
#pragma unroll
      for (int p = 0; p < NS::OUT; ++p) a.dz_last[((size_t)tile * 32 + p) * 32 + j] = dzl[p];
    }
  }

  const int a_off8 = (h * 256 + j) * 16;
  const int voff_t = (4 * h * 32 + j) * 4;
  f32x16 acc[8];
  float in[128];
  TileStore ts{af_rsrc(a.dz, 0), voff_t};

  auto mask_out = [&](int l) {      // acc = dX_l; mask with sign bits of X_l (masks[l-1]) -> in[] = dZ_{l-1}
    const u32x4 m4 = *(const u32x4*)(a.masks + (((size_t)(l - 1) * a.nt_stride + tile) * 64 + lane) * 4);
    const uint32_t mk[4] = {m4[0], m4[1], m4[2], m4[3]};
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        in[T * 16 + r] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, (float)acc[T][r]) &
                                                   (uint32_t)__builtin_amdgcn_sbfe((int)mk[T >> 1], 31 - ((T & 1) * 16 + r), 1));
    ts.r = af_rsrc_uniform(a.dz + ((size_t)(l - 1) * a.nt_stride + tile) * AF_TILE_F, live ? AF_TILE_F * 4 : 0);
    // the element-wise ops above are inline asm (af_relu / bf_mask_keep): gfx950 needs TWO wait states between a VALU write of a VGPR and an MFMA
    // reading it as SrcA / SrcB (tools/hazardprobe.hip); hipcc pads its own VALU ops and cannot see these.  Left to the scheduler they sink to just in
    // front of the output layer's MFMAs (a two-layer net has nothing else behind them): stale operands, a corrupted chain.  isa_check.py rule (d)
    // proves on every build that no such pair exists
    AF_ELEMWISE_FENCE();
  };
  auto hook_dma = [&](auto gi) { if constexpr (decltype(gi)::value < 8) cs.issue2(); };
  auto hook_dma_store = [&](auto gi) { if constexpr (decltype(gi)::value < 8) cs.issue2(); ts.template part<decltype(gi)::value>(in); };

  // ---- output layer: K = 8 (one group), only p < OUT non-zero
  { const char* buf = cs.next<CB::BLAST>(); mm_block<8, 1, 0, NS::OUT, true>(acc, dzl, buf + a_off8, hook_dma); }
  mask_out(nl - 1);

  for (int l = nl - 2; l >= 1; --l) {
    { const char* buf = cs.next<CB::HID>(); mm_block<8, 8, 0, 4, true>(acc, in, buf + a_off8, hook_dma_store); }
    { const char* buf = cs.next<CB::HID>(); mm_block<8, 8, 32, 4>(acc, in, buf + a_off8, hook_dma); }
    { const char* buf = cs.next<CB::HID>(); mm_block<8, 8, 64, 4>(acc, in, buf + a_off8, hook_dma); }
    { const char* buf = cs.next<CB::HID>(); mm_block<8, 8, 96, 4>(acc, in, buf + a_off8, hook_dma); }
    mask_out(l);
  }

  if constexpr (NS::DX0) {
    // dPE = W_0^T dZ_0  (M = 64 padded PE features, K = 256), then chain through sin/cos to the 2-D input
    static_assert(NS::IN == AF_IN_PE2, "input gradient is only needed for the atlas net");
    f32x16 acc2[2];
    { const char* buf = cs.next<CB::BL0>(); mm_block<2, 32, 0, 4, true>(acc2, in, buf + (h * 64 + j) * 16, hook_dma_store); }
    const auto r = af_rsrc_uniform(a.pe_tile + (size_t)tile * 64 * 32, 64 * 32 * 4);
    float dx0 = 0.f, dx1 = 0.f;
#pragma unroll
    for (int g = 0; g < 5; ++g) {
      float pv[4], dv[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        pv[p] = af_bl32(r, (4 * h * 32 + j) * 4, (8 * g + p) * 128);
        dv[p] = acc2[g >> 2][(g & 3) * 4 + p];
      }
      const float b = h ? __builtin_ldexpf(3.14159265358979323846f, 2 * g + 1) : __builtin_ldexpf(3.14159265358979323846f, 2 * g);
      dx0 += b * (pv[2] * dv[0] - pv[0] * dv[2]);
      dx1 += b * (pv[3] * dv[1] - pv[1] * dv[3]);
    }
    dx0 += __shfl_xor(dx0, 32);
    dx1 += __shfl_xor(dx1, 32);
    if (live && h == 0 && row < a.nrows) {
      float* dst = row < a.split_row ? a.din0 + (size_t)row * 4 : a.din1 + (size_t)(row - a.split_row) * 4;
      dst[0] += a.din_scale * dx0;
      dst[1] += a.din_scale * dx1;
    }
  } else {
    // dZ_0 of a net whose input needs no gradient: nothing left to hide the stores behind
    ts.template part<0>(in); ts.template part<1>(in); ts.template part<2>(in); ts.template part<3>(in);
    ts.template part<4>(in); ts.template part<5>(in); ts.template part<6>(in); ts.template part<7>(in);
  }
}

// ------------------------------------------------------------------------------------------------
// One launch, up to AF_MAX_NETS row-tile ranges of different nets back to back ("parts").  A workgroup finds
// its part by its index and runs that net's chain.  Packing several nets (or the odd last round of one net
// next to another net) into one grid removes the idle tail of separate launches: 2188 row tiles of the
// 7-segment mapping batch are 2.14 rounds of the 1024 SIMDs but cost 3 as a launch of their own.
// Parts of one launch must be independent of each other (the host orders dependent work across launches).
template <bool TRAIN>
__global__ __launch_bounds__(256, 1) void k_mlp_fwd_multi(MultiFwd m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AF_STAMP(m, 0);
  int s = 0, base = 0;
  const int wg = blockIdx.x;
  while (s + 1 < m.n && wg >= m.wg_end[s]) { base = m.wg_end[s]; ++s; }
  switch (m.net[s]) {
    case AF_NET_MAP1:  if (m.a[s].nl > 2) mlp_fwd_body<NsMap1, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body<NsMap1, TRAIN, false>(m.a[s], wg - base, smem); break;
    case AF_NET_MAP2:  if (m.a[s].nl > 2) mlp_fwd_body<NsMap2, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body<NsMap2, TRAIN, false>(m.a[s], wg - base, smem); break;
    case AF_NET_ATLAS: if (m.a[s].nl > 2) mlp_fwd_body<NsAtlas, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body<NsAtlas, TRAIN, false>(m.a[s], wg - base, smem); break;
    case AF_KIND_MAP_PE: if (m.a[s].nl > 2) mlp_fwd_body<NsMapPe, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body<NsMapPe, TRAIN, false>(m.a[s], wg - base, smem); break;
    default:           if (m.a[s].nl > 2) mlp_fwd_body<NsAlpha, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body<NsAlpha, TRAIN, false>(m.a[s], wg - base, smem); break;
  }
  AF_STAMP(m, 1);
}

__global__ __launch_bounds__(256, 1) void k_mlp_bwd_multi(MultiBwd m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AF_STAMP(m, 0);
  int s = 0, base = 0;
  const int wg = blockIdx.x;
  while (s + 1 < m.n && wg >= m.wg_end[s]) { base = m.wg_end[s]; ++s; }
  switch (m.net[s]) {
    case AF_NET_MAP1:  mlp_bwd_body<NsMap1>(m.a[s], wg - base, smem); break;
    case AF_NET_MAP2:  mlp_bwd_body<NsMap2>(m.a[s], wg - base, smem); break;
    case AF_NET_ATLAS: mlp_bwd_body<NsAtlas>(m.a[s], wg - base, smem); break;
    case AF_KIND_MAP_PE: mlp_bwd_body<NsMapPe>(m.a[s], wg - base, smem); break;
    default:           mlp_bwd_body<NsAlpha>(m.a[s], wg - base, smem); break;
  }
  AF_STAMP(m, 1);
}

// wg_end[] is filled here from the parts' tile ranges
extern "C" int af_launch_fwd_multi(MultiFwd* m, int train, hipStream_t s) {
  int tot = 0;
  for (int i = 0; i < m->n; ++i) { tot += (m->a[i].NT - m->a[i].tile0 + 3) / 4; m->wg_end[i] = tot; }
  if (tot <= 0) return 0;
  const size_t lds = AF_LDS_BYTES;
  if (train) hipLaunchKernelGGL((k_mlp_fwd_multi<true>), dim3(tot), dim3(256), lds, s, *m);
  else       hipLaunchKernelGGL((k_mlp_fwd_multi<false>), dim3(tot), dim3(256), lds, s, *m);
  return (int)hipGetLastError();
}

extern "C" int af_launch_bwd_multi(MultiBwd* m, hipStream_t s) {
  int tot = 0;
  for (int i = 0; i < m->n; ++i) { tot += (m->a[i].NT - m->a[i].tile0 + 3) / 4; m->wg_end[i] = tot; }
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(k_mlp_bwd_multi, dim3(tot), dim3(256), AF_LDS_BYTES, s, *m);
  return (int)hipGetLastError();
}

// The chunk sizes the kernels assume, for the host planner to check its layout against:
// which = 0 fwd layer 0, 1 hidden quarter, 2 skip columns, 3 fwd output layer, 4 bwd output layer, 5 bwd layer 0.
extern "C" int af_mlp_chunk_bytes(int net, int which, int nl) {     // nl: layers of the net (the output-layer chunk is longer when it carries skip columns)
  auto pick = [&](auto ns) -> int {
    using CB = ChunkBytes<decltype(ns)>;
    const int v[6] = {CB::L0, CB::HID, CB::SKIP, CB::last_bytes(nl), CB::BLAST, CB::BL0};
    return which >= 0 && which < 6 ? v[which] : -1;
  };
  switch (net) {
    case AF_NET_MAP1:  return pick(NsMap1{});
    case AF_NET_MAP2:  return pick(NsMap2{});
    case AF_NET_ATLAS: return pick(NsAtlas{});
    case AF_NET_ALPHA: return pick(NsAlpha{});
    case AF_KIND_MAP_PE: return pick(NsMapPe{});
    default: return -1;
  }
}

extern "C" int af_mlp_init() {   // opt in to 128 KB dynamic LDS
  hipError_t e = hipSuccess;
#define AF_ATTR(K) do { hipError_t r = hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, AF_LDS_BYTES); if (r != hipSuccess) e = r; } while (0)
  AF_ATTR((k_mlp_fwd_multi<true>)); AF_ATTR((k_mlp_fwd_multi<false>)); AF_ATTR(k_mlp_bwd_multi);
#undef AF_ATTR
  return (int)e;
}
