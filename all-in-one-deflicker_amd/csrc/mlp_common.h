// mlp_common.h — pieces shared by the 32-row chains (mlp.hip) and the 16-row chains (mlp16.hip): net shapes, the
// compile-time chunk plan, the LDS weight-chunk stream, the bias staging and the one-instruction ReLU.
#pragma once
#include <utility>

#include "af_dev.h"

#ifndef AF_ABL
#define AF_ABL 0   // tools/ablate.hip timing probes: bit0 no tile stores, bit1 no LDS-DMA, bit2 no barriers, bit3 no LDS fragment reads, bit4 no
                   // sched_barrier, bit5 no operand split, bit6 only wave 0 issues the LDS-DMA, bit8 atlas PE without sin / cos, bit9 with sincosf
#endif

// NL is the SHIPPED layer count (config_flow_100.json); the kernels take the actual count from FwdArgs::nl / BwdArgs::nl.
struct NsMap1  { static constexpr int NL = 6, IN = AF_IN_XYT, K0G = 1, PEG = 0, OUT = 2; static constexpr unsigned SKIP = 0;                       static constexpr bool DX0 = false; };
struct NsMap2  { static constexpr int NL = 4, IN = AF_IN_XYT, K0G = 1, PEG = 0, OUT = 2; static constexpr unsigned SKIP = 0;                       static constexpr bool DX0 = false; };
struct NsAtlas { static constexpr int NL = 8, IN = AF_IN_PE2, K0G = 5, PEG = 5, OUT = 3; static constexpr unsigned SKIP = (1u << 4) | (1u << 7);   static constexpr bool DX0 = true;  };
struct NsMapPe { static constexpr int NL = 6, IN = AF_IN_PE3, K0G = 4, PEG = 4, OUT = 2; static constexpr unsigned SKIP = 0;                       static constexpr bool DX0 = false; };
struct NsAlpha { static constexpr int NL = 8, IN = AF_IN_PE3, K0G = 4, PEG = 4, OUT = 1; static constexpr unsigned SKIP = 0;                       static constexpr bool DX0 = false; };

// Byte sizes of the weight chunks, in stream order (host.hip plan_images lays the images out contiguously in
// exactly this order, so a chunk's address is the previous chunk's address plus its size: no table lookups
// inside the kernels).  A k-group of an Mpad-row image is 2*Mpad*16 bytes; sizes round up to 4 KB.
constexpr int af_round4k(int b) { return (b + 4095) / 4096 * 4096; }
template <class NS> struct ChunkBytes {
  static constexpr int L0   = af_round4k(NS::K0G * 2 * AF_HID * 16);                                            // forward layer 0
  static constexpr int HID  = 8 * 2 * AF_HID * 16;                                                            // a quarter of a 256x256 layer = 64 KB
  static constexpr int SKIP = af_round4k(NS::PEG * 2 * AF_HID * 16);                                            // PE columns of a skip layer
  static constexpr int LASTN = af_round4k(32 * 2 * 4 * 16);                                                   // forward output layer (Mpad 4) ...
  static constexpr int LASTS = af_round4k((32 + NS::PEG) * 2 * 4 * 16);                                       // ... with the PE columns of a skip-concat in front of it
  static constexpr bool out_skip(int nl) { return ((NS::SKIP >> (nl - 1)) & 1u) != 0; }                       // implicit_neural_networks.py:40-44: layer i in skip_layers, i == num_layers - 1
  static constexpr int last_bytes(int nl) { return out_skip(nl) ? LASTS : LASTN; }
  static constexpr int LAST = last_bytes(NS::NL);
  static constexpr int BLAST = 2 * AF_HID * 16;                                                               // backward output layer: one k-group
  static constexpr int BL0  = 32 * 2 * 64 * 16;                                                               // backward layer 0 (Mpad 64 PE slots)
};
static_assert(ChunkBytes<NsMap1>::HID == AF_CHUNK_MAX && ChunkBytes<NsAtlas>::BL0 == AF_CHUNK_MAX, "chunk = LDS buffer");

template <int G> struct GIdx { static constexpr int value = G; };

// Double-buffered LDS stream of weight chunks shared by the four waves of the workgroup.  Every stage moves
// a full 64 KB buffer (16 x 1 KB per wave) whatever the chunk's real size — the image buffers are padded so
// the over-read stays in bounds — which keeps the issue sites branch-free: two LDS-DMA instructions per
// k-group ride in the shadow of that group's 32 MFMAs.
struct ChunkStream {
  const char* src;                                     // this lane's 16-B column of the chunk being staged
  char* smem; int wave, cidx;
  char* p_dst; int p_it;                               // chunk being staged (issued incrementally)
  AF_DEV void begin_stage(int c) { p_dst = smem + (c & 1) * AF_CHUNK_MAX + wave * 1024; p_it = 0; }
  AF_DEV void issue2() {
    if constexpr (AF_ABL & 2) { p_it += 2; return; }
    af_glds16(src + p_it * 4096, p_dst + p_it * 4096);
    af_glds16(src + p_it * 4096 + 4096, p_dst + p_it * 4096 + 4096);
    p_it += 2;
  }
  AF_DEV void start(const void* img, int tid) { src = (const char*)img + tid * 16; cidx = 0; begin_stage(0); }
  // Make chunk `cidx` (BYTES long) visible to every wave (and know every wave is done with chunk cidx-1), then
  // arm the staging of chunk cidx+1 — which starts BYTES further on — into the buffer chunk cidx-1 used.
  // Returns the LDS base of chunk cidx.  After the last chunk the stream stages 64 KB of whatever follows
  // (the image buffers are padded for that) into the idle buffer: harmless, and the issue sites stay branch-free.
  template <int BYTES> AF_DEV const char* next() { return next_rt(BYTES); }
  AF_DEV const char* next_rt(int BYTES) {
    while (p_it < 16) issue2();
    if constexpr (!(AF_ABL & 4)) { af_wait_vm0(); __syncthreads(); }
    const int cur = cidx;
    cidx = cur + 1;
    src += BYTES;
    begin_stage(cidx);
    return smem + (cur & 1) * AF_CHUNK_MAX;
  }
};

// max(z, 0) as ONE v_max_f32 (the C-level fmaxf adds a canonicalising v_max in front).  v_max_f32 returns 0 for a NaN input
// where torch.relu propagates it: a NaN pre-activation needs a non-finite parameter or input (Adam's steps are bounded by
// lr), and k_adam raises AF_ENAN for any non-finite parameter (elem.hip), so the condition is reported, not healed silently.
AF_DEV float af_relu(float z) { float v; asm("v_max_f32 %0, 0, %1" : "=v"(v) : "v"(z)); return v; }
#ifdef AF_NO_ELEMWISE_FENCE      // experiment: the build that corrupted two-layer nets in round 3 (tools/experiments/README.md, "element-wise fence")
#define AF_ELEMWISE_FENCE() do { } while (0)
#else
#define AF_ELEMWISE_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// Accumulator initialisation = bias.  The net's padded bias rows ([NL][256] floats, <= 8 KB) are copied once per
// workgroup into LDS behind the two weight buffers: 32 ds_read_b128 per layer cost a fraction of the 32 VMEM loads
// they replace (each VMEM instruction steals ~40 cycles of MFMA issue from the only wave of its SIMD).
#define AF_BIAS_LDS (2 * AF_CHUNK_MAX)          // byte offset of the bias rows in dynamic LDS
#define AF_LDS_BYTES (2 * AF_CHUNK_MAX + AF_MAX_LAYERS * AF_HID * 4)
// Row tiles a launch really has: the static count, or — when the batch's flow-match rows are compacted on the device
// (k_prep, elem.hip) — the tiles up to the last live row.
template <class Args> AF_DEV int live_tiles(const Args& a) {
  if (!a.live_rows) return a.NT;
  const int nt = (a.live_base + *a.live_rows + 31) >> 5;
  return nt < a.NT ? nt : a.NT;
}

AF_DEV void stage_bias(int nl, const float* bias, char* bias_lds, int tid) {
  float* dst = (float*)bias_lds;
  for (int i = 0; i < nl; ++i) dst[i * AF_HID + tid] = bias[i * AF_HID + tid];     // 256 threads x nl rows; visible after the first barrier
}

// ---- blocks shared by the fp32 chains (mlp.hip) and the bf16x6 chains (mlpbf.hip) ----------------------------------------
#ifndef AF_SGB
#define AF_SGB 1     // explicit MFMA / memory-instruction interleave (sched_group_barrier) inside every k-group
#endif
#ifndef AF_AFRAG
#define AF_AFRAG 1   // A fragments in AGPRs (see lds_frag)
#endif

// One A fragment (four consecutive k of one output row) from the LDS weight image, pinned to the accumulator half of
// the register file: ds_read_b128 writes AGPRs directly and the MFMA takes its A operand from there, so the 64
// fragment registers do not compete with the 128 activation registers for the 256 architectural VGPRs (with all of
// them in VGPRs the allocator is full and sinks every group's reads to the end of the previous group, where their
// latency is exposed behind an s_waitcnt lgkmcnt(0)).
AF_DEV f32x4 lds_frag(const char* p) { return *(const f32x4*)p; }
// The pin sits at the fragment's first USE (top of its k-group), not at the load: the compiler's s_waitcnt for the
// read lands there too, a whole group (32 MFMAs) after the read was issued.
AF_DEV void pin_acc(f32x4& v) {
#if AF_AFRAG
  asm("" : "+a"(v));
#endif
}

// acc[T] += A(image in LDS) * b[B0 + 4*g + p]  for NG k-groups; a_lds already includes the lane offset
// (h*MPAD + j)*16.  NP < 4 skips reduction indices that are structurally zero.  hook(g) is called once per
// k-group right after that group's A-fragment reads were issued: work placed there (LDS-DMA issue of the
// next weight chunk, stores of the previous layer's activations) runs in the shadow of the group's MFMAs
// instead of in front of an empty matrix pipe.
template <int MT, int NG, int B0, int NP, bool ZI, int NB, class Hook, int... Gs>
AF_DEV void mm_block_impl(f32x16 (&acc)[MT], const float (&b)[NB], const char* a_lds, Hook& hook, std::integer_sequence<int, Gs...>) {
  constexpr int MPAD = MT * 32;
  f32x4 a[2][MT];
#pragma unroll
  for (int T = 0; T < MT; ++T) a[0][T] = lds_frag(a_lds + T * 32 * 16);
  if constexpr (AF_ABL & 8) {
#pragma unroll
    for (int T = 0; T < MT; ++T) a[1][T] = a[0][T];
  }
  auto step = [&](auto gi) {
    constexpr int g = decltype(gi)::value;
#pragma unroll
    for (int T = 0; T < MT; ++T) pin_acc(a[g & 1][T]);
    if constexpr (g + 1 < NG && !(AF_ABL & 8)) {
#pragma unroll
      for (int T = 0; T < MT; ++T) a[(g + 1) & 1][T] = lds_frag(a_lds + ((g + 1) * 2 * MPAD + 32 * T) * 16);
    }
    hook(gi);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int T = 0; T < MT; ++T) {
        if constexpr (ZI && g == 0) {
          if (p == 0) { const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][T][0], b[B0], z, 0, 0, 0); continue; }
        }
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][T][p], b[B0 + g * 4 + p], acc[T], 0, 0, 0);
      }
    }
#if AF_SGB
    // Issue order inside the group: one LDS fragment read and one VMEM instruction (LDS-DMA piece / tile store) behind
    // each MFMA, so that every memory instruction issues in the shadow of a 64-cycle MFMA instead of in one burst
    // behind which the matrix pipe drains (hipcc's own order: 24 MFMAs, then 8 reads + 2 DMA + up to 16 stores).
#pragma unroll
    for (int i = 0; i < NP * MT; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    }
#endif
    if constexpr (!(AF_ABL & 16)) __builtin_amdgcn_sched_barrier(0);     // keep each group's DMA / stores inside its own MFMA shadow
  };
  (step(GIdx<Gs>{}), ...);
}
// ZI: the accumulators start at zero — the first MFMA of each takes an inline-constant 0 as C instead of
// a previously zeroed register block (saves MT*16 v_accvgpr_write per layer in the backward chain).
template <int MT, int NG, int B0, int NP, bool ZI = false, int NB, class Hook>
AF_DEV void mm_block(f32x16 (&acc)[MT], const float (&b)[NB], const char* a_lds, Hook&& hook) {
  mm_block_impl<MT, NG, B0, NP, ZI>(acc, b, a_lds, hook, std::make_integer_sequence<int, NG>{});
}

AF_DEV void init_bias(f32x16 (&acc)[8], const char* bias_lds /* the staged [NL][256] rows */, int layer, int h) {
  const char* b = bias_lds + (layer * AF_HID + 4 * h) * 4;
#pragma unroll
  for (int T = 0; T < 8; ++T) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *(const f32x4*)(b + (32 * T + 8 * q) * 4);
      acc[T][q * 4 + 0] = b4[0]; acc[T][q * 4 + 1] = b4[1]; acc[T][q * 4 + 2] = b4[2]; acc[T][q * 4 + 3] = b4[3];
    }
  }
}

// Store 1/8 (feature tile T) of a C-layout block (reg = 16T+4q+p <-> feature 32T+8q+4h+p) into a T-layout
// tile [256][32].
template <int T>
AF_DEV void store_tile_part(const float (&v)[128], __amdgpu_buffer_rsrc_t r, int voff) {
  // the 128-B steps between p = 0..3 ride in the instruction's immediate offset: one soffset per (T, q)
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) af_bs32_tile(v[T * 16 + rr], r, voff + (rr & 3) * 128, (32 * T + 8 * (rr >> 2)) * 128);
}

// Deferred stores of one 32x256 block: one feature tile per k-group of the following GEMM block.
// A dead wave (tile past the end) carries a zero-length buffer descriptor: its stores are dropped by the
// hardware bounds check, so the store sites need no branch.
struct TileStore {
  __amdgpu_buffer_rsrc_t r; int voff;
  template <int G> AF_DEV void part(const float (&v)[128]) const { if constexpr (G < 8 && !(AF_ABL & 1)) store_tile_part<G>(v, r, voff); }
};

