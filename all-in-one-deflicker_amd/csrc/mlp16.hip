// mlp16.hip — 16-row variant of the fused mapping-MLP chains, for batches too small to fill the chip.
//
// `pre_train_mapping` (src/models/stage_1/unwrap_utils.py:176-198) runs 100 x F sequential Adam steps on 10 000 rows:
// 313 row tiles of 32 occupy 313 of the 1024 SIMDs for a whole tile chain per step — the step is bound by the LATENCY
// of one chain, not by throughput.  Here one wavefront owns 16 rows and uses v_mfma_f32_16x16x4_f32 (exact fp32, same
// 64 FLOP/clk/SIMD as the 32x32x2 form): a chain takes half as long, 10 000 rows become 157 workgroups of 4 waves
// (64 rows) — one round of the chip — and a pre-train step drops from 0.35 to ~0.24 ms.
//
// Layouts: lane (j, q) = (row j of 16, k-quad q); register 4T+r of a 16x256 block = feature 16T + 4q + r.  That is the
// C/D fragment of the transposed product Y^T = W X^T AND a valid B operand of the next layer for the SAME packed
// weight image the 32-row kernels read (slot k>>2 of an Mpad-row image holds W[m][4(k>>2) .. +3]); one 64 KB chunk is
// 4 groups of 16 k-values.  Activation / gradient tiles, dz_last and x0 tiles are the usual 32-row T-layout tiles (a
// 16-row tile fills one half), so k_dw and k_adam are unchanged.  The ReLU sign masks are private to this pair of
// kernels: 64 bits per lane and layer, element e = 4T + r at bit 31 - (e & 31) of word e >> 5.
// Mapping nets only (no skips): the nets pre_train_mapping touches — xyt input, or (use_positional_encoding_mapping*) the PE 3 -> 6K
// input stage in the slot order of the 32-row kernels: image slot 8g + 4h + p of lane half h there is k = 16G + 4q + r here with
// g = 2G + (q >> 1), h = q & 1, so lane quad q computes the features of lane half q & 1 and feeds the halves (q >> 1) of them.
#include "mlp_common.h"

// acc[T] += A * b over NG groups of 16 k-values.  a_lds includes the lane offset (q*MPAD + j)*16.  Each group runs in
// two halves of MT/2 output tiles: the A fragments of one half (<= 32 registers) are fetched while the other half's
// MFMAs run, and consecutive MFMAs always hit distinct accumulators.
template <int MT, int MPAD, int NG, int B0, int NP, bool ZI, int NB, class Hook, int... Gs>
AF_DEV void mm16_impl(f32x4 (&acc)[MT], const float (&b)[NB], const char* a_lds, Hook& hook, std::integer_sequence<int, Gs...>) {
  constexpr int HT = MT / 2;
  f32x4 a[2][HT];
#pragma unroll
  for (int T = 0; T < HT; ++T) a[0][T] = *(const f32x4*)(a_lds + T * 16 * 16);
  auto step = [&](auto gi) {
    constexpr int g = decltype(gi)::value;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int cur = hf;                                   // two halves per group: buffer index = half
      if (hf == 0) {
#pragma unroll
        for (int T = 0; T < HT; ++T) a[1][T] = *(const f32x4*)(a_lds + (g * 4 * MPAD + (HT + T) * 16) * 16);
        hook(gi);
      } else if (g + 1 < NG) {
#pragma unroll
        for (int T = 0; T < HT; ++T) a[0][T] = *(const f32x4*)(a_lds + ((g + 1) * 4 * MPAD + T * 16) * 16);
      }
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int T = 0; T < HT; ++T) {
          if constexpr (ZI && g == 0) {
            if (p == 0) { const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                          acc[hf * HT + T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][T][0], b[B0], z, 0, 0, 0); continue; }
          }
          acc[hf * HT + T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][T][p], b[B0 + g * 4 + p], acc[hf * HT + T], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  (step(GIdx<Gs>{}), ...);
}
template <int MT, int MPAD, int NG, int B0, int NP, bool ZI = false, int NB, class Hook>
AF_DEV void mm16(f32x4 (&acc)[MT], const float (&b)[NB], const char* a_lds, Hook&& hook) {
  mm16_impl<MT, MPAD, NG, B0, NP, ZI>(acc, b, a_lds, hook, std::make_integer_sequence<int, NG>{});
}

// one feature tile (16 features, registers 4T..4T+3 of this lane) of a 16x256 block -> T-layout tile rows
template <int T>
AF_DEV void store16_part(const float (&v)[64], __amdgpu_buffer_rsrc_t r, int voff) {
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) af_bs32(v[T * 4 + rr], r, voff + rr * 128, 16 * T * 128);
}
struct TileStore16 {
  __amdgpu_buffer_rsrc_t r; int voff;
  template <int G> AF_DEV void part(const float (&v)[64]) {       // group G of a 4-group block carries feature tiles 4G..4G+3
    if constexpr (G < 4) { store16_part<4 * G>(v, r, voff); store16_part<4 * G + 1>(v, r, voff); store16_part<4 * G + 2>(v, r, voff); store16_part<4 * G + 3>(v, r, voff); }
  }
};

template <class NS>
__global__ __launch_bounds__(256, 1) void k_mlp16_fwd(FwdArgs a) {
  static_assert((NS::IN == AF_IN_XYT || NS::IN == AF_IN_PE3) && NS::SKIP == 0, "mapping nets only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CB = ChunkBytes<NS>;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 15, q = lane >> 4;
  int t16 = 2 * a.tile0 + blockIdx.x * 4 + wave;             // 16-row tile index; a.NT counts 32-row tiles
  const bool live = t16 < 2 * a.NT;
  if (!live) t16 = 2 * a.NT - 1;
  const int row = t16 * 16 + j;
  const int tile = t16 >> 1, roff = (t16 & 1) * 16 + j;       // 32-row T-layout tile and the row inside it

  ChunkStream cs{nullptr, smem, wave, 0, nullptr, 0};
  cs.start(a.wimg, tid);
  const int nl = a.nl;
  stage_bias(nl, a.bias, smem + AF_BIAS_LDS, tid);

  constexpr int NG0 = NS::IN == AF_IN_XYT ? 1 : 2;              // groups of 16 k-values of the first layer
  float x0[4 * NG0];
  {
    const f32x4 v = *(const f32x4*)(a.in + (size_t)row * 4);
    if constexpr (NS::IN == AF_IN_XYT) {
#pragma unroll
      for (int p = 0; p < 4; ++p) x0[p] = (q == 0 && p < 3) ? v[p] : 0.f;     // k = 4q + p; only k < 3 is real
    } else {      // PE 3 -> 30 slots exactly as mlp.hip computes them for lane half hq (implicit_neural_networks.py:9-13, accurate sinf / cosf)
      const int hq = q & 1;
      const float x[3] = {v[0], v[1], v[2]};
      const float bA = __builtin_ldexpf(3.14159265358979323846f, 2 * hq), bB = __builtin_ldexpf(3.14159265358979323846f, 2 * hq + 1);
      const float b4 = __builtin_ldexpf(3.14159265358979323846f, 4);
      float pe[16];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        pe[d] = sinf(x[d] * bA); pe[3 + d] = cosf(x[d] * bA);
        pe[6 + d] = sinf(x[d] * bB); pe[9 + d] = cosf(x[d] * bB);
        pe[12 + d] = hq ? cosf(x[d] * b4) : sinf(x[d] * b4);
      }
      pe[15] = 0.f;
      if (live && a.pe_tile && (q >> 1) == 0) {   // PE features in reference feature order, T-layout [64][32], for the layer-0 dW
        const auto r = af_rsrc_uniform(a.pe_tile + (size_t)tile * 64 * 32, 64 * 32 * 4);
#pragma unroll
        for (int rho = 0; rho < 15; ++rho) {
          if (rho < 12) af_bs32(pe[rho], r, (12 * hq * 32 + roff) * 4, rho * 128);
          else          af_bs32(pe[rho], r, (3 * hq * 32 + roff) * 4, (24 + rho - 12) * 128);
        }
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) x0[4 * g + pp] = (q >> 1) ? pe[8 * g + 4 + pp] : pe[8 * g + pp];
    }
  }
  const int a_off = (q * 256 + j) * 16;
  const int voff_t = (4 * q * 32 + roff) * 4;
  f32x4 acc[16];
  float in[64];
  TileStore16 ts{af_rsrc(a.acts, 0), voff_t};

  auto init_bias = [&](int l) {
#pragma unroll
    for (int T = 0; T < 16; ++T) acc[T] = *(const f32x4*)(smem + AF_BIAS_LDS + (l * AF_HID + 16 * T + 4 * q) * 4);
  };
  auto relu_out = [&](int l) {
    uint32_t mk[2] = {0u, 0u};
#pragma unroll
    for (int T = 0; T < 16; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = af_relu(acc[T][r]);
        in[T * 4 + r] = v;
        mk[T >> 3] = __builtin_amdgcn_alignbit(mk[T >> 3], __builtin_bit_cast(uint32_t, 0.f - v), 31);
      }
    if (live) { uint2 m2 = make_uint2(mk[0], mk[1]); *(uint2*)(a.masks + (((size_t)l * a.nt_stride * 2 + t16) * 64 + lane) * 2) = m2; }
    ts.r = af_rsrc_uniform(a.acts + ((size_t)l * a.nt_stride + tile) * AF_TILE_F, live ? AF_TILE_F * 4 : 0);
    // the element-wise ops above are inline asm (af_relu / bf_mask_keep): gfx950 needs TWO wait states between a VALU write of a VGPR and an MFMA
    // reading it as SrcA / SrcB (tools/hazardprobe.hip); hipcc pads its own VALU ops and cannot see these.  Left to the scheduler they sink to just in
    // front of the output layer's MFMAs (a two-layer net has nothing else behind them): stale operands, a corrupted chain.  isa_check.py rule (d)
    // proves on every build that no such pair exists
    AF_ELEMWISE_FENCE();
  };
  auto hook_dma = [&](auto gi) { if constexpr (decltype(gi)::value < 4) { cs.issue2(); cs.issue2(); } };
  auto hook_dma_store = [&](auto gi) { if constexpr (decltype(gi)::value < 4) { cs.issue2(); cs.issue2(); } ts.template part<decltype(gi)::value>(in); };

  // ---- layer 0: one group of 16 k-values, of which k < 3 carry data.  Slots 2, 3 of that group lie past the 8 KB
  // image (the stage always copies 64 KB: they hold the next layer's weights, finite) and meet B = 0.
  { const char* buf = cs.next<CB::L0>(); init_bias(0); mm16<16, 256, NG0, 0, 4>(acc, x0, buf + a_off, hook_dma); }
  relu_out(0);

  for (int l = 1; l <= nl - 2; ++l) {
    init_bias(l);
    { const char* buf = cs.next<CB::HID>(); mm16<16, 256, 4, 0, 4>(acc, in, buf + a_off, hook_dma_store); }
    { const char* buf = cs.next<CB::HID>(); mm16<16, 256, 4, 16, 4>(acc, in, buf + a_off, hook_dma); }
    { const char* buf = cs.next<CB::HID>(); mm16<16, 256, 4, 32, 4>(acc, in, buf + a_off, hook_dma); }
    { const char* buf = cs.next<CB::HID>(); mm16<16, 256, 4, 48, 4>(acc, in, buf + a_off, hook_dma); }
    relu_out(l);
  }

  // ---- output layer on 4x4x1 MFMA blocks (Mpad-4 image, see mlp.hip): lane l = block (l >> 2) = (k-quad q, row quad),
  // column l & 3; A = W[l & 3][16T + 4q + r] = slot 4T + q of the image, B = in[4T + r]; four k-quads -> two shuffles.
  {
    const char* buf = cs.next_rt(CB::last_bytes(nl));
    ts.template part<0>(in); ts.template part<1>(in); ts.template part<2>(in); ts.template part<3>(in);
    f32x4 o4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < 16; ++T) {
      const f32x4 w = *(const f32x4*)(buf + ((4 * T + q) * 4 + (lane & 3)) * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[r], in[4 * T + r], o4[r], 0, 0, 0);
    }
    const f32x4 bias = *(const f32x4*)(smem + AF_BIAS_LDS + (nl - 1) * AF_HID * 4);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float z = (o4[0][i] + o4[1][i]) + (o4[2][i] + o4[3][i]);
      z += __shfl_xor(z, 16);
      z += __shfl_xor(z, 32);
      o[i] = i < NS::OUT ? tanhf(z + bias[i]) : 0.f;
    }
    if (live && q == 0) *(f32x4*)(a.out + (size_t)row * 4) = o;
  }
}

template <class NS>
__global__ __launch_bounds__(256, 1) void k_mlp16_bwd(BwdArgs a) {
  static_assert((NS::IN == AF_IN_XYT || NS::IN == AF_IN_PE3) && NS::SKIP == 0 && !NS::DX0, "mapping nets only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CB = ChunkBytes<NS>;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 15, q = lane >> 4;
  int t16 = 2 * a.tile0 + blockIdx.x * 4 + wave;
  const bool live = t16 < 2 * a.NT;
  if (!live) t16 = 2 * a.NT - 1;
  const int row = t16 * 16 + j;
  const int tile = t16 >> 1, roff = (t16 & 1) * 16 + j;

  ChunkStream cs{nullptr, smem, wave, 0, nullptr, 0};
  cs.start(a.wimg, tid);
  const int nl = a.nl;

  float dzl[4];
  {
    const f32x4 o = *(const f32x4*)(a.out + (size_t)row * 4);
    const f32x4 d = *(const f32x4*)(a.dout + (size_t)row * 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) dzl[p] = (q == 0 && p < NS::OUT) ? d[p] * (1.f - o[p] * o[p]) : 0.f;
    if (live && q == 0) {
#pragma unroll
      for (int p = 0; p < NS::OUT; ++p) a.dz_last[((size_t)tile * 32 + p) * 32 + roff] = dzl[p];
    }
  }
  const int a_off = (q * 256 + j) * 16;
  const int voff_t = (4 * q * 32 + roff) * 4;
  f32x4 acc[16];
  float in[64];
  TileStore16 ts{af_rsrc(a.dz, 0), voff_t};

  auto mask_out = [&](int l) {      // acc = dX_l; mask with the sign bits of X_l (masks[l-1]) -> in[] = dZ_{l-1}
    const uint2 m2 = *(const uint2*)(a.masks + (((size_t)(l - 1) * a.nt_stride * 2 + t16) * 64 + lane) * 2);
    const uint32_t mk[2] = {m2.x, m2.y};
#pragma unroll
    for (int T = 0; T < 16; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        in[T * 4 + r] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, (float)acc[T][r]) &
                                                  (uint32_t)__builtin_amdgcn_sbfe((int)mk[T >> 3], 31 - ((T & 7) * 4 + r), 1));
    ts.r = af_rsrc_uniform(a.dz + ((size_t)(l - 1) * a.nt_stride + tile) * AF_TILE_F, live ? AF_TILE_F * 4 : 0);
    // the element-wise ops above are inline asm (af_relu / bf_mask_keep): gfx950 needs TWO wait states between a VALU write of a VGPR and an MFMA
    // reading it as SrcA / SrcB (tools/hazardprobe.hip); hipcc pads its own VALU ops and cannot see these.  Left to the scheduler they sink to just in
    // front of the output layer's MFMAs (a two-layer net has nothing else behind them): stale operands, a corrupted chain.  isa_check.py rule (d)
    // proves on every build that no such pair exists
    AF_ELEMWISE_FENCE();
  };
  auto hook_dma = [&](auto gi) { if constexpr (decltype(gi)::value < 4) { cs.issue2(); cs.issue2(); } };
  auto hook_dma_store = [&](auto gi) { if constexpr (decltype(gi)::value < 4) { cs.issue2(); cs.issue2(); } ts.template part<decltype(gi)::value>(in); };

  // output layer: W_last^T image (Mpad 256), one group of 16 k = out features of which OUT (k-quad 0) are real;
  // k-quads 2, 3 read past the 8 KB image (next layer's weights, finite) against B = 0
  { const char* buf = cs.next<CB::BLAST>(); mm16<16, 256, 1, 0, NS::OUT, true>(acc, dzl, buf + a_off, hook_dma); }
  mask_out(nl - 1);

  for (int l = nl - 2; l >= 1; --l) {
    { const char* buf = cs.next<CB::HID>(); mm16<16, 256, 4, 0, 4, true>(acc, in, buf + a_off, hook_dma_store); }
    { const char* buf = cs.next<CB::HID>(); mm16<16, 256, 4, 16, 4>(acc, in, buf + a_off, hook_dma); }
    { const char* buf = cs.next<CB::HID>(); mm16<16, 256, 4, 32, 4>(acc, in, buf + a_off, hook_dma); }
    { const char* buf = cs.next<CB::HID>(); mm16<16, 256, 4, 48, 4>(acc, in, buf + a_off, hook_dma); }
    mask_out(l);
  }
  ts.template part<0>(in); ts.template part<1>(in); ts.template part<2>(in); ts.template part<3>(in);     // dZ_0
}

// ------------------------------------------------------------------------------------------------
extern "C" int af_launch_fwd16(int net, const FwdArgs* a, hipStream_t s) {
  const dim3 grid((2 * (a->NT - a->tile0) + 3) / 4), block(256);
  switch (net) {
    case AF_NET_MAP1: hipLaunchKernelGGL((k_mlp16_fwd<NsMap1>), grid, block, AF_LDS_BYTES, s, *a); break;
    case AF_NET_MAP2: hipLaunchKernelGGL((k_mlp16_fwd<NsMap2>), grid, block, AF_LDS_BYTES, s, *a); break;
    case AF_KIND_MAP_PE: hipLaunchKernelGGL((k_mlp16_fwd<NsMapPe>), grid, block, AF_LDS_BYTES, s, *a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
extern "C" int af_launch_bwd16(int net, const BwdArgs* a, hipStream_t s) {
  const dim3 grid((2 * (a->NT - a->tile0) + 3) / 4), block(256);
  switch (net) {
    case AF_NET_MAP1: hipLaunchKernelGGL((k_mlp16_bwd<NsMap1>), grid, block, AF_LDS_BYTES, s, *a); break;
    case AF_NET_MAP2: hipLaunchKernelGGL((k_mlp16_bwd<NsMap2>), grid, block, AF_LDS_BYTES, s, *a); break;
    case AF_KIND_MAP_PE: hipLaunchKernelGGL((k_mlp16_bwd<NsMapPe>), grid, block, AF_LDS_BYTES, s, *a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
extern "C" int af_mlp16_init() {
  hipError_t e = hipSuccess;
#define AF_ATTR(K) do { hipError_t r = hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, AF_LDS_BYTES); if (r != hipSuccess) e = r; } while (0)
  AF_ATTR((k_mlp16_fwd<NsMap1>)); AF_ATTR((k_mlp16_fwd<NsMap2>)); AF_ATTR((k_mlp16_bwd<NsMap1>)); AF_ATTR((k_mlp16_bwd<NsMap2>)); AF_ATTR((k_mlp16_fwd<NsMapPe>)); AF_ATTR((k_mlp16_bwd<NsMapPe>));
#undef AF_ATTR
  return (int)e;
}
