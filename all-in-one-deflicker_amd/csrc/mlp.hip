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
// LDS: 2 x 64 KB.  Registers: 128 (activations) + 128 (accumulators) + 64 (A fragments) -> 1 wave / SIMD.
#include "af_dev.h"

struct NsMap1  { static constexpr int NL = 6, IN = AF_IN_XYT, K0G = 1, PEG = 0, OUT = 2; static constexpr unsigned SKIP = 0;                       static constexpr bool DX0 = false; };
struct NsMap2  { static constexpr int NL = 4, IN = AF_IN_XYT, K0G = 1, PEG = 0, OUT = 2; static constexpr unsigned SKIP = 0;                       static constexpr bool DX0 = false; };
struct NsAtlas { static constexpr int NL = 8, IN = AF_IN_PE2, K0G = 5, PEG = 5, OUT = 3; static constexpr unsigned SKIP = (1u << 4) | (1u << 7);   static constexpr bool DX0 = true;  };
struct NsAlpha { static constexpr int NL = 8, IN = AF_IN_PE3, K0G = 4, PEG = 4, OUT = 1; static constexpr unsigned SKIP = 0;                       static constexpr bool DX0 = false; };

// acc[T] += A(image in LDS) * b[B0 + 4*g + p]  for NG k-groups; a_lds already includes the lane offset
// (h*MPAD + j)*16.  NP < 4 skips reduction indices that are structurally zero.
template <int MT, int NG, int B0, int NP, int NB>
AF_DEV void mm_block(f32x16 (&acc)[MT], const float (&b)[NB], const char* a_lds) {
  constexpr int MPAD = MT * 32;
  f32x4 a[2][MT];
#pragma unroll
  for (int T = 0; T < MT; ++T) a[0][T] = *(const f32x4*)(a_lds + T * 32 * 16);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
#pragma unroll
      for (int T = 0; T < MT; ++T) a[(g + 1) & 1][T] = *(const f32x4*)(a_lds + ((g + 1) * 2 * MPAD + 32 * T) * 16);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int T = 0; T < MT; ++T)
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][T][p], b[B0 + g * 4 + p], acc[T], 0, 0, 0);
    }
  }
}

struct ChunkStream {
  const char* img; const AfChunk* tab; char* smem; int tid, wave, cidx, n;
  AF_DEV void issue(int c) {
    const AfChunk d = tab[c];
    af_stage_chunk(img + d.off, d.bytes, smem + (c & 1) * AF_CHUNK_MAX, tid, wave);
  }
  // Wait until chunk `cidx` has landed for every wave, and every wave is done with chunk cidx-1;
  // start fetching chunk cidx+1 into the buffer chunk cidx-1 used; return the LDS base of chunk cidx.
  AF_DEV const char* next() {
    af_wait_vm0();
    __syncthreads();
    const int cur = cidx;
    if (cur + 1 < n) issue(cur + 1);
    cidx = cur + 1;
    return smem + (cur & 1) * AF_CHUNK_MAX;
  }
};

AF_DEV void init_bias(f32x16 (&acc)[8], __amdgpu_buffer_rsrc_t rb, int layer, int h) {
#pragma unroll
  for (int T = 0; T < 8; ++T) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = af_bl128(rb, h * 16, (layer * AF_HID + 32 * T + 8 * q) * 4);
      acc[T][q * 4 + 0] = b4[0]; acc[T][q * 4 + 1] = b4[1]; acc[T][q * 4 + 2] = b4[2]; acc[T][q * 4 + 3] = b4[3];
    }
  }
}

// Store a C-layout block (reg = 16T+4q+p <-> feature 32T+8q+4h+p) as a T-layout tile [256][32].
AF_DEV void store_tile(const float (&v)[128], float* tile_base, int j, int h) {
  const auto r = af_rsrc(tile_base, AF_TILE_F * 4);
  const int voff = (4 * h * 32 + j) * 4;
#pragma unroll
  for (int T = 0; T < 8; ++T)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr)
      af_bs32(v[T * 16 + rr], r, voff, (32 * T + (rr & 3) + 8 * (rr >> 2)) * 128);
}

template <class NS, bool TRAIN>
__global__ __launch_bounds__(256, 1) void k_mlp_fwd(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, h = lane >> 5;
  int tile = blockIdx.x * 4 + wave;
  const bool live = tile < a.NT;
  if (!live) tile = a.NT - 1;
  const int row = tile * 32 + j;

  ChunkStream cs{(const char*)a.wimg, a.chunks, smem, tid, wave, 0, a.nchunks};
  cs.issue(0);

  const auto rb = af_rsrc(a.bias, NS::NL * AF_HID * 4);
  constexpr int NPE = NS::PEG > 0 ? NS::PEG * 4 : 4;
  float pe[NPE];            // first-layer / skip B operand (PE features, or xyt for the mapping nets)
  {
    const f32x4 v = *(const f32x4*)(a.in + (size_t)row * 4);
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
        const auto r = af_rsrc(a.pe_tile + (size_t)tile * 64 * 32, 64 * 32 * 4);
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
  f32x16 acc[8];
  float in[128];

  auto epilogue = [&](int l) {               // relu -> in[], optional stores of X_{l+1} and its sign bits
    uint32_t mk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaxf(acc[T][r], 0.f);
        in[T * 16 + r] = v;
        if (TRAIN) mk[T >> 1] |= (v > 0.f ? 1u : 0u) << ((T & 1) * 16 + r);
      }
    if constexpr (TRAIN) {
      if (live) {
        store_tile(in, a.acts + ((size_t)l * a.NT + tile) * AF_TILE_F, j, h);
        u32x4 m4 = {mk[0], mk[1], mk[2], mk[3]};
        *(u32x4*)(a.masks + (((size_t)l * a.NT + tile) * 64 + lane) * 4) = m4;
      }
    }
  };

  // ---- layer 0
  init_bias(acc, rb, 0, h);
  {
    const char* buf = cs.next();
    mm_block<8, NS::K0G, 0, 4>(acc, pe, buf + a_off8);
  }
  epilogue(0);

  // ---- hidden layers 1 .. NL-2
  for (int l = 1; l <= NS::NL - 2; ++l) {
    init_bias(acc, rb, l, h);
    { const char* buf = cs.next(); mm_block<8, 8, 0, 4>(acc, in, buf + a_off8); }
    { const char* buf = cs.next(); mm_block<8, 8, 32, 4>(acc, in, buf + a_off8); }
    { const char* buf = cs.next(); mm_block<8, 8, 64, 4>(acc, in, buf + a_off8); }
    { const char* buf = cs.next(); mm_block<8, 8, 96, 4>(acc, in, buf + a_off8); }
    if constexpr (NS::SKIP != 0) {
      if ((NS::SKIP >> l) & 1) { const char* buf = cs.next(); mm_block<8, NS::PEG, 0, 4>(acc, pe, buf + a_off8); }
    }
    epilogue(l);
  }

  // ---- output layer (one 32-wide tile, OUT real rows), tanh
  {
    f32x16 acc1[1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = af_bl128(rb, h * 16, ((NS::NL - 1) * AF_HID + 8 * q) * 4);
      acc1[0][q * 4 + 0] = b4[0]; acc1[0][q * 4 + 1] = b4[1]; acc1[0][q * 4 + 2] = b4[2]; acc1[0][q * 4 + 3] = b4[3];
    }
    const char* buf = cs.next();
    const char* al = buf + (h * 32 + j) * 16;
    mm_block<1, 32, 0, 4>(acc1, in, al);
    if constexpr ((NS::SKIP >> (NS::NL - 1)) & 1) mm_block<1, NS::PEG, 0, 4>(acc1, pe, al + 32 * 2 * 32 * 16);
    if (live && h == 0) {
      f32x4 o;
      o[0] = tanhf(acc1[0][0]);
      o[1] = NS::OUT > 1 ? tanhf(acc1[0][1]) : 0.f;
      o[2] = NS::OUT > 2 ? tanhf(acc1[0][2]) : 0.f;
      o[3] = 0.f;
      *(f32x4*)(a.out + (size_t)row * 4) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward dX chain: dZ_{l-1} = (W_l^T dZ_l) . relu'(Z_{l-1}); writes every dZ_l in T-layout for the dW
// GEMMs (dw.hip).  For the atlas net the chain continues through layer 0 into the positional encoding
// and accumulates dL/d(uv) onto the mapping net's output gradient (the detached skip inputs carry no
// gradient: implicit_neural_networks.py:69).
template <class NS>
__global__ __launch_bounds__(256, 1) void k_mlp_bwd(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, h = lane >> 5;
  int tile = blockIdx.x * 4 + wave;
  const bool live = tile < a.NT;
  if (!live) tile = a.NT - 1;
  const int row = tile * 32 + j;

  ChunkStream cs{(const char*)a.wimg, a.chunks, smem, tid, wave, 0, a.nchunks};
  cs.issue(0);

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
  f32x16 acc[8];
  float in[128];

  auto zero_acc = [&]() {
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[T][r] = 0.f;
  };
  auto epilogue = [&](int l) {      // acc = dX_l; mask with sign bits of X_l (masks[l-1]) -> dZ_{l-1}
    const u32x4 m4 = *(const u32x4*)(a.masks + (((size_t)(l - 1) * a.NT + tile) * 64 + lane) * 4);
    const uint32_t mk[4] = {m4[0], m4[1], m4[2], m4[3]};
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        in[T * 16 + r] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, (float)acc[T][r]) &
                                                   (uint32_t)__builtin_amdgcn_sbfe((int)mk[T >> 1], (T & 1) * 16 + r, 1));
    if (live) store_tile(in, a.dz + ((size_t)(l - 1) * a.NT + tile) * AF_TILE_F, j, h);
  };

  // ---- output layer: K = 8 (one group), only p < OUT non-zero
  zero_acc();
  { const char* buf = cs.next(); mm_block<8, 1, 0, NS::OUT>(acc, dzl, buf + a_off8); }
  epilogue(NS::NL - 1);

  for (int l = NS::NL - 2; l >= 1; --l) {
    zero_acc();
    { const char* buf = cs.next(); mm_block<8, 8, 0, 4>(acc, in, buf + a_off8); }
    { const char* buf = cs.next(); mm_block<8, 8, 32, 4>(acc, in, buf + a_off8); }
    { const char* buf = cs.next(); mm_block<8, 8, 64, 4>(acc, in, buf + a_off8); }
    { const char* buf = cs.next(); mm_block<8, 8, 96, 4>(acc, in, buf + a_off8); }
    epilogue(l);
  }

  if constexpr (NS::DX0) {
    // dPE = W_0^T dZ_0  (M = 64 padded PE features, K = 256), then chain through sin/cos to the 2-D input
    static_assert(NS::IN == AF_IN_PE2, "input gradient is only needed for the atlas net");
    f32x16 acc2[2];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[T][r] = 0.f;
    { const char* buf = cs.next(); mm_block<2, 32, 0, 4>(acc2, in, buf + (h * 64 + j) * 16); }
    const auto r = af_rsrc(a.pe_tile + (size_t)tile * 64 * 32, 64 * 32 * 4);
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
  }
}

// ------------------------------------------------------------------------------------------------
extern "C" int af_launch_fwd(int net, int train, const FwdArgs* a, hipStream_t s) {
  const dim3 grid((a->NT + 3) / 4), block(256);
  const size_t lds = 2 * AF_CHUNK_MAX;
#define AF_FWD(NS)                                                                 \
  do {                                                                             \
    if (train) hipLaunchKernelGGL((k_mlp_fwd<NS, true>), grid, block, lds, s, *a); \
    else       hipLaunchKernelGGL((k_mlp_fwd<NS, false>), grid, block, lds, s, *a);\
  } while (0)
  switch (net) {
    case AF_NET_MAP1:  AF_FWD(NsMap1);  break;
    case AF_NET_MAP2:  AF_FWD(NsMap2);  break;
    case AF_NET_ATLAS: AF_FWD(NsAtlas); break;
    case AF_NET_ALPHA: AF_FWD(NsAlpha); break;
    default: return -1;
  }
#undef AF_FWD
  return (int)hipGetLastError();
}

extern "C" int af_launch_bwd(int net, const BwdArgs* a, hipStream_t s) {
  const dim3 grid((a->NT + 3) / 4), block(256);
  const size_t lds = 2 * AF_CHUNK_MAX;
  switch (net) {
    case AF_NET_MAP1:  hipLaunchKernelGGL((k_mlp_bwd<NsMap1>),  grid, block, lds, s, *a); break;
    case AF_NET_MAP2:  hipLaunchKernelGGL((k_mlp_bwd<NsMap2>),  grid, block, lds, s, *a); break;
    case AF_NET_ATLAS: hipLaunchKernelGGL((k_mlp_bwd<NsAtlas>), grid, block, lds, s, *a); break;
    case AF_NET_ALPHA: hipLaunchKernelGGL((k_mlp_bwd<NsAlpha>), grid, block, lds, s, *a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

extern "C" int af_mlp_init() {   // opt in to 128 KB dynamic LDS for every instantiation
  hipError_t e = hipSuccess;
#define AF_ATTR(K) do { hipError_t r = hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * AF_CHUNK_MAX); if (r != hipSuccess) e = r; } while (0)
  AF_ATTR((k_mlp_fwd<NsMap1, true>));  AF_ATTR((k_mlp_fwd<NsMap1, false>));
  AF_ATTR((k_mlp_fwd<NsMap2, true>));  AF_ATTR((k_mlp_fwd<NsMap2, false>));
  AF_ATTR((k_mlp_fwd<NsAtlas, true>)); AF_ATTR((k_mlp_fwd<NsAtlas, false>));
  AF_ATTR((k_mlp_fwd<NsAlpha, true>)); AF_ATTR((k_mlp_fwd<NsAlpha, false>));
  AF_ATTR((k_mlp_bwd<NsMap1>)); AF_ATTR((k_mlp_bwd<NsMap2>)); AF_ATTR((k_mlp_bwd<NsAtlas>)); AF_ATTR((k_mlp_bwd<NsAlpha>));
#undef AF_ATTR
  return (int)e;
}
