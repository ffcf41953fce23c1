// mlp16.hip — 16-row variant of the register-chained fused MLP kernels (see mlp.hip for the idea).
//
// One wavefront owns 16 rows and uses v_mfma_f32_16x16x4_f32 (exact fp32, same 64 FLOP/clk/SIMD rate as the
// 32x32x2 form).  Activations: 64 registers (lane (j,q) = row j, k-quad q; register 4T+r = feature 16T+4q+r),
// accumulators: 64, A fragments: 64  ->  <= 256 registers, so TWO waves share a SIMD: one wave's epilogue /
// barrier / LDS latency hides behind the other's MFMAs, and the unit of work is half as large (a 70 000-row
// batch is 4.27 tiles per SIMD instead of 2.14, i.e. 85 % instead of 71 % tail efficiency).
// The packed weight image is the SAME as for the 32-row kernels: slot(k>>2, m) holds W[m][4*(k>>2) .. +3].
// Workgroup = 8 waves = 128 rows, one 64 KB weight chunk = 4 groups of 16 k-values = 256 MFMAs per wave.
#include <utility>

#include "../../all-in-one-deflicker_amd/csrc/af_dev.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct Ns16Map1  { static constexpr int NL = 6, IN = AF_IN_XYT, K0G = 1, PEG = 0, OUT = 2; static constexpr unsigned SKIP = 0;                     static constexpr bool DX0 = false; };
struct Ns16Atlas { static constexpr int NL = 8, IN = AF_IN_PE2, K0G = 3, PEG = 3, OUT = 3; static constexpr unsigned SKIP = (1u << 4) | (1u << 7); static constexpr bool DX0 = true;  };

template <int G> struct GI { static constexpr int value = G; };

// acc[T] += A * b over NG groups of 16 k-values.  a_lds includes the lane offset (q*MPAD + i)*16.
// Each group is done in two halves of MT/2 output tiles so that the A fragments of one half (<= 32 registers)
// are prefetched while the other half's MFMAs run; consecutive MFMAs always hit distinct accumulators.
template <int MT, int MPAD, int NG, int B0, int NP, int NB, class Hook, int... Gs>
AF_DEV void mm16_impl(f32x4v (&acc)[MT], const float (&b)[NB], const char* a_lds, Hook& hook, std::integer_sequence<int, Gs...>) {
  constexpr int HT = MT >= 2 ? MT / 2 : 1, NH = MT >= 2 ? 2 : 1;
  f32x4v a[2][HT];
#pragma unroll
  for (int T = 0; T < HT; ++T) a[0][T] = *(const f32x4v*)(a_lds + T * 16 * 16);
  auto step = [&](auto gi) {
    constexpr int g = decltype(gi)::value;
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      const int cur = (g * NH + hf) & 1;
      // prefetch the next half (same group, or first half of the next group)
      if (hf + 1 < NH) {
#pragma unroll
        for (int T = 0; T < HT; ++T) a[cur ^ 1][T] = *(const f32x4v*)(a_lds + (g * 4 * MPAD + (HT + T) * 16) * 16);
      } else if (g + 1 < NG) {
#pragma unroll
        for (int T = 0; T < HT; ++T) a[cur ^ 1][T] = *(const f32x4v*)(a_lds + ((g + 1) * 4 * MPAD + T * 16) * 16);
      }
      if (hf == 0) hook(gi);
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int T = 0; T < HT; ++T)
          acc[hf * HT + T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][T][p], b[B0 + g * 4 + p], acc[hf * HT + T], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  (step(GI<Gs>{}), ...);
}
template <int MT, int MPAD, int NG, int B0, int NP, int NB, class Hook>
AF_DEV void mm16(f32x4v (&acc)[MT], const float (&b)[NB], const char* a_lds, Hook&& hook) {
  mm16_impl<MT, MPAD, NG, B0, NP>(acc, b, a_lds, hook, std::make_integer_sequence<int, NG>{});
}

// 64 KB stages issued by 512 threads: 8 iterations of 8 KB; two per k-group of a 4-group chunk.
struct ChunkStream16 {
  const char* img; const AfChunk* tab; char* smem; int tid, wave, cidx, n;
  const char* p_src; char* p_dst; int p_it;
  AF_DEV void begin_stage(int c) {
    const AfChunk d = tab[c < n ? c : 0];
    p_src = img + d.off + tid * 16; p_dst = smem + (c & 1) * AF_CHUNK_MAX + wave * 1024;
    p_it = 0;
  }
  AF_DEV void issue2() {
    af_glds16(p_src + p_it * 8192, p_dst + p_it * 8192);
    af_glds16(p_src + p_it * 8192 + 8192, p_dst + p_it * 8192 + 8192);
    p_it += 2;
  }
  AF_DEV void start() { cidx = 0; begin_stage(0); }
  AF_DEV const char* next() {
    while (p_it < 8) issue2();
    af_wait_vm0();
    __syncthreads();
    const int cur = cidx;
    cidx = cur + 1;
    begin_stage(cidx);
    return smem + (cur & 1) * AF_CHUNK_MAX;
  }
};

AF_DEV void init_bias16(f32x4v (&acc)[16], __amdgpu_buffer_rsrc_t rb, int layer, int q) {
#pragma unroll
  for (int T = 0; T < 16; ++T) acc[T] = af_bl128(rb, q * 16, (layer * AF_HID + 16 * T) * 4);
}

// registers 4T..4T+3 of a C-layout block -> T-layout tile rows; one feature tile (16 features) per call
template <int T>
AF_DEV void store16_part(const float (&v)[64], __amdgpu_buffer_rsrc_t r, int voff) {
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) af_bs32(v[T * 4 + rr], r, voff, (16 * T + rr) * 128);
}
struct TileStore16 {
  __amdgpu_buffer_rsrc_t r; int voff;
  // group g of a 4-group block carries feature tiles 4g..4g+3
  template <int G> AF_DEV void part(const float (&v)[64]) {
    if constexpr (G < 4) { store16_part<4 * G>(v, r, voff); store16_part<4 * G + 1>(v, r, voff); store16_part<4 * G + 2>(v, r, voff); store16_part<4 * G + 3>(v, r, voff); }
  }
};

#define AF_PI 3.14159265358979323846f

template <class NS, bool TRAIN>
__global__ __launch_bounds__(512, 2) void k_mlp16_fwd(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 15, q = lane >> 4;
  int t16 = (a.tile0 << 1) + blockIdx.x * 8 + wave;          // 16-row tile index; a.NT counts 32-row tiles
  const bool live = t16 < 2 * a.NT;
  if (!live) t16 = 2 * a.NT - 1;
  const int row = t16 * 16 + j;
  const int tile = t16 >> 1, roff = (t16 & 1) * 16 + j;       // 32-row T-layout tile and row inside it

  ChunkStream16 cs{(const char*)a.wimg, a.chunks, smem, tid, wave, 0, a.nchunks, nullptr, nullptr, 0};
  cs.start();

  const auto rb = af_rsrc(a.bias, NS::NL * AF_HID * 4);
  constexpr int NPE = NS::PEG > 0 ? NS::PEG * 4 : 4;
  float pe[NPE];
  {
    const f32x4v v = *(const f32x4v*)(a.in + (size_t)row * 4);
    if constexpr (NS::IN == AF_IN_XYT) {
#pragma unroll
      for (int p = 0; p < 4; ++p) pe[p] = (q == 0 && p < 3) ? v[p] : 0.f;
    } else {
      const float sh = row < a.split_row ? a.in_shift0 : a.in_shift1;
      const float x0 = v[0] * a.in_scale + sh, x1 = v[1] * a.in_scale + sh;
#pragma unroll
      for (int g = 0; g < 3; ++g) {                           // lane quad q owns frequencies 4g+q (< 10)
        const int k = 4 * g + q;
        const float b = __builtin_ldexpf(AF_PI, k);
        const float p0 = x0 * b, p1 = x1 * b;
        const bool ok = k < 10;
        pe[g * 4 + 0] = ok ? sinf(p0) : 0.f; pe[g * 4 + 1] = ok ? sinf(p1) : 0.f;
        pe[g * 4 + 2] = ok ? cosf(p0) : 0.f; pe[g * 4 + 3] = ok ? cosf(p1) : 0.f;
      }
      if constexpr (TRAIN) {
        const auto r = af_rsrc_uniform(a.pe_tile + (size_t)tile * 64 * 32, live ? 64 * 32 * 4 : 0);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int p = 0; p < 4; ++p)
            if (g < 2 || q < 2) af_bs32(pe[g * 4 + p], r, (4 * q * 32 + roff) * 4, (16 * g + p) * 128);
      }
    }
  }

  const int a_off = (q * 256 + j) * 16;       // lane offset inside a Mpad = 256 image plane set
  const int voff_t = (4 * q * 32 + roff) * 4;
  f32x4v acc[16];
  float in[64];
  TileStore16 ts{af_rsrc(a.acts, 0), voff_t};

  auto relu_out = [&](int l) {
    uint32_t mk[2] = {0u, 0u};
#pragma unroll
    for (int T = 0; T < 16; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = fmaxf(acc[T][r], 0.f);
        in[T * 4 + r] = v;
        if (TRAIN) mk[T >> 3] |= (v > 0.f ? 1u : 0u) << ((T & 7) * 4 + r);
      }
    if constexpr (TRAIN) {
      if (live) *(uint2*)(a.masks + (((size_t)l * a.nt_stride * 2 + t16) * 64 + lane) * 2) = make_uint2(mk[0], mk[1]);
      ts.r = af_rsrc_uniform(a.acts + ((size_t)l * a.nt_stride + tile) * AF_TILE_F, live ? AF_TILE_F * 4 : 0);
    }
  };
  auto hook_dma = [&](auto gi) { if constexpr (decltype(gi)::value < 4) cs.issue2(); };
  auto hook_dma_store = [&](auto gi) {
    if constexpr (decltype(gi)::value < 4) cs.issue2();
    if constexpr (TRAIN) ts.template part<decltype(gi)::value>(in);
  };

  // ---- layer 0
  init_bias16(acc, rb, 0, q);
  { const char* buf = cs.next(); mm16<16, 256, NS::K0G, 0, 4>(acc, pe, buf + a_off, hook_dma); }
  relu_out(0);

  for (int l = 1; l <= NS::NL - 2; ++l) {
    init_bias16(acc, rb, l, q);
    { const char* buf = cs.next(); mm16<16, 256, 4, 0, 4>(acc, in, buf + a_off, hook_dma_store); }
    { const char* buf = cs.next(); mm16<16, 256, 4, 16, 4>(acc, in, buf + a_off, hook_dma); }
    { const char* buf = cs.next(); mm16<16, 256, 4, 32, 4>(acc, in, buf + a_off, hook_dma); }
    { const char* buf = cs.next(); mm16<16, 256, 4, 48, 4>(acc, in, buf + a_off, hook_dma); }
    if constexpr (NS::SKIP != 0) {
      if ((NS::SKIP >> l) & 1) { const char* buf = cs.next(); mm16<16, 256, NS::PEG, 0, 4>(acc, pe, buf + a_off, hook_dma); }
    }
    relu_out(l);
  }

  {   // output layer: one 16-feature tile (Mpad = 32 image), tanh
    f32x4v acc1[1];
    acc1[0] = af_bl128(rb, q * 16, ((NS::NL - 1) * AF_HID) * 4);
    const char* buf = cs.next();
    const char* al = buf + (q * 32 + j) * 16;
    mm16<1, 32, 16, 0, 4>(acc1, in, al, hook_dma_store);
    if constexpr ((NS::SKIP >> (NS::NL - 1)) & 1) mm16<1, 32, NS::PEG, 0, 4>(acc1, pe, al + 16 * 4 * 32 * 16, hook_dma);
    if (live && q == 0) {
      f32x4v o;
      o[0] = tanhf(acc1[0][0]);
      o[1] = NS::OUT > 1 ? tanhf(acc1[0][1]) : 0.f;
      o[2] = NS::OUT > 2 ? tanhf(acc1[0][2]) : 0.f;
      o[3] = 0.f;
      *(f32x4v*)(a.out + (size_t)row * 4) = o;
    }
  }
}

template <class NS>
__global__ __launch_bounds__(512, 2) void k_mlp16_bwd(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 15, q = lane >> 4;
  int t16 = (a.tile0 << 1) + blockIdx.x * 8 + wave;
  const bool live = t16 < 2 * a.NT;
  if (!live) t16 = 2 * a.NT - 1;
  const int row = t16 * 16 + j;
  const int tile = t16 >> 1, roff = (t16 & 1) * 16 + j;

  ChunkStream16 cs{(const char*)a.wimg, a.chunks, smem, tid, wave, 0, a.nchunks, nullptr, nullptr, 0};
  cs.start();

  float dzl[4];
  {
    const f32x4v o = *(const f32x4v*)(a.out + (size_t)row * 4);
    const f32x4v d = *(const f32x4v*)(a.dout + (size_t)row * 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) dzl[p] = (q == 0 && p < NS::OUT) ? d[p] * (1.f - o[p] * o[p]) : 0.f;
    if (live && q == 0) {
#pragma unroll
      for (int p = 0; p < NS::OUT; ++p) a.dz_last[((size_t)tile * 32 + p) * 32 + roff] = dzl[p];
    }
  }

  const int a_off = (q * 256 + j) * 16;
  const int voff_t = (4 * q * 32 + roff) * 4;
  f32x4v acc[16];
  float in[64];
  TileStore16 ts{af_rsrc(a.dz, 0), voff_t};

  auto zero_acc = [&]() {
#pragma unroll
    for (int T = 0; T < 16; ++T) { f32x4v z = {0.f, 0.f, 0.f, 0.f}; acc[T] = z; }
  };
  auto mask_out = [&](int l) {
    const uint2 m2 = *(const uint2*)(a.masks + (((size_t)(l - 1) * a.nt_stride * 2 + t16) * 64 + lane) * 2);
    const uint32_t mk[2] = {m2.x, m2.y};
#pragma unroll
    for (int T = 0; T < 16; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        in[T * 4 + r] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, (float)acc[T][r]) &
                                                  (uint32_t)__builtin_amdgcn_sbfe((int)mk[T >> 3], (T & 7) * 4 + r, 1));
    ts.r = af_rsrc_uniform(a.dz + ((size_t)(l - 1) * a.nt_stride + tile) * AF_TILE_F, live ? AF_TILE_F * 4 : 0);
  };
  auto hook_dma = [&](auto gi) { if constexpr (decltype(gi)::value < 4) cs.issue2(); };
  auto hook_dma_store = [&](auto gi) { if constexpr (decltype(gi)::value < 4) cs.issue2(); ts.template part<decltype(gi)::value>(in); };

  // output layer: W_last^T image, M = 256, one group of 16 out features of which OUT are real (lane quad 0)
  zero_acc();
  { const char* buf = cs.next(); mm16<16, 256, 1, 0, NS::OUT>(acc, dzl, buf + a_off, hook_dma); }
  mask_out(NS::NL - 1);

  for (int l = NS::NL - 2; l >= 1; --l) {
    zero_acc();
    { const char* buf = cs.next(); mm16<16, 256, 4, 0, 4>(acc, in, buf + a_off, hook_dma_store); }
    { const char* buf = cs.next(); mm16<16, 256, 4, 16, 4>(acc, in, buf + a_off, hook_dma); }
    { const char* buf = cs.next(); mm16<16, 256, 4, 32, 4>(acc, in, buf + a_off, hook_dma); }
    { const char* buf = cs.next(); mm16<16, 256, 4, 48, 4>(acc, in, buf + a_off, hook_dma); }
    mask_out(l);
  }

  if constexpr (NS::DX0) {
    f32x4v acc2[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) { f32x4v z = {0.f, 0.f, 0.f, 0.f}; acc2[T] = z; }
    { const char* buf = cs.next(); mm16<4, 64, 16, 0, 4>(acc2, in, buf + (q * 64 + j) * 16, hook_dma_store); }
    const auto r = af_rsrc_uniform(a.pe_tile + (size_t)tile * 64 * 32, 64 * 32 * 4);
    float dx0 = 0.f, dx1 = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int k = 4 * g + q;
      float pv[4], dv[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        pv[p] = (g < 2 || q < 2) ? af_bl32(r, (4 * q * 32 + roff) * 4, (16 * g + p) * 128) : 0.f;
        dv[p] = acc2[g][p];
      }
      const float b = __builtin_ldexpf(AF_PI, k);
      dx0 += b * (pv[2] * dv[0] - pv[0] * dv[2]);
      dx1 += b * (pv[3] * dv[1] - pv[1] * dv[3]);
    }
    dx0 += __shfl_xor(dx0, 16); dx0 += __shfl_xor(dx0, 32);
    dx1 += __shfl_xor(dx1, 16); dx1 += __shfl_xor(dx1, 32);
    if (live && q == 0 && row < a.nrows) {
      float* dst = row < a.split_row ? a.din0 + (size_t)row * 4 : a.din1 + (size_t)(row - a.split_row) * 4;
      dst[0] += a.din_scale * dx0;
      dst[1] += a.din_scale * dx1;
    }
  } else {
    ts.template part<0>(in); ts.template part<1>(in); ts.template part<2>(in); ts.template part<3>(in);
  }
}

extern "C" int af_launch_fwd16(int net, int train, const FwdArgs* a, hipStream_t s) {
  const dim3 grid((2 * (a->NT - a->tile0) + 7) / 8), block(512);
  const size_t lds = 2 * AF_CHUNK_MAX;
#define AF_FWD16(NS)                                                                   \
  do {                                                                                 \
    if (train) hipLaunchKernelGGL((k_mlp16_fwd<NS, true>), grid, block, lds, s, *a);   \
    else       hipLaunchKernelGGL((k_mlp16_fwd<NS, false>), grid, block, lds, s, *a);  \
  } while (0)
  switch (net) {
    case AF_NET_MAP1:  AF_FWD16(Ns16Map1);  break;
    case AF_NET_ATLAS: AF_FWD16(Ns16Atlas); break;
    default: return -1;
  }
#undef AF_FWD16
  return (int)hipGetLastError();
}
extern "C" int af_launch_bwd16(int net, const BwdArgs* a, hipStream_t s) {
  const dim3 grid((2 * (a->NT - a->tile0) + 7) / 8), block(512);
  const size_t lds = 2 * AF_CHUNK_MAX;
  switch (net) {
    case AF_NET_MAP1:  hipLaunchKernelGGL((k_mlp16_bwd<Ns16Map1>),  grid, block, lds, s, *a); break;
    case AF_NET_ATLAS: hipLaunchKernelGGL((k_mlp16_bwd<Ns16Atlas>), grid, block, lds, s, *a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
extern "C" int af_mlp16_init() {
  hipError_t e = hipSuccess;
#define AF_ATTR(K) do { hipError_t r = hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * AF_CHUNK_MAX); if (r != hipSuccess) e = r; } while (0)
  AF_ATTR((k_mlp16_fwd<Ns16Map1, true>));  AF_ATTR((k_mlp16_fwd<Ns16Map1, false>));
  AF_ATTR((k_mlp16_fwd<Ns16Atlas, true>)); AF_ATTR((k_mlp16_fwd<Ns16Atlas, false>));
  AF_ATTR((k_mlp16_bwd<Ns16Map1>)); AF_ATTR((k_mlp16_bwd<Ns16Atlas>));
#undef AF_ATTR
  return (int)e;
}
