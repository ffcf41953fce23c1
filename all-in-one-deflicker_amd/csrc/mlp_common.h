// mlp_common.h — pieces shared by the 32-row chains (mlp.hip) and the 16-row chains (mlp16.hip): net shapes, the
// compile-time chunk plan, the LDS weight-chunk stream, the bias staging and the one-instruction ReLU.
#pragma once
#include <utility>

#include "af_dev.h"

#ifndef AF_ABL
#define AF_ABL 0   // bit0 no tile stores, bit1 no LDS-DMA, bit2 no barriers, bit3 no LDS fragment reads, bit4 no sched_barrier
#endif

struct NsMap1  { static constexpr int NL = 6, IN = AF_IN_XYT, K0G = 1, PEG = 0, OUT = 2; static constexpr unsigned SKIP = 0;                       static constexpr bool DX0 = false; };
struct NsMap2  { static constexpr int NL = 4, IN = AF_IN_XYT, K0G = 1, PEG = 0, OUT = 2; static constexpr unsigned SKIP = 0;                       static constexpr bool DX0 = false; };
struct NsAtlas { static constexpr int NL = 8, IN = AF_IN_PE2, K0G = 5, PEG = 5, OUT = 3; static constexpr unsigned SKIP = (1u << 4) | (1u << 7);   static constexpr bool DX0 = true;  };
struct NsAlpha { static constexpr int NL = 8, IN = AF_IN_PE3, K0G = 4, PEG = 4, OUT = 1; static constexpr unsigned SKIP = 0;                       static constexpr bool DX0 = false; };

// Byte sizes of the weight chunks, in stream order (host.hip plan_images lays the images out contiguously in
// exactly this order, so a chunk's address is the previous chunk's address plus its size: no table lookups
// inside the kernels).  A k-group of an Mpad-row image is 2*Mpad*16 bytes; sizes round up to 4 KB.
constexpr int af_round4k(int b) { return (b + 4095) / 4096 * 4096; }
template <class NS> struct ChunkBytes {
  static constexpr int L0   = af_round4k(NS::K0G * 2 * AF_HID * 16);                                            // forward layer 0
  static constexpr int HID  = 8 * 2 * AF_HID * 16;                                                            // a quarter of a 256x256 layer = 64 KB
  static constexpr int SKIP = af_round4k(NS::PEG * 2 * AF_HID * 16);                                            // PE columns of a skip layer
  static constexpr int LAST = af_round4k((32 + (((NS::SKIP >> (NS::NL - 1)) & 1) ? NS::PEG : 0)) * 2 * 4 * 16);    // forward output layer (Mpad 4)
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
  template <int BYTES> AF_DEV const char* next() {
    while (p_it < 16) issue2();
    if constexpr (!(AF_ABL & 4)) { af_wait_vm0(); __syncthreads(); }
    const int cur = cidx;
    cidx = cur + 1;
    src += BYTES;
    begin_stage(cidx);
    return smem + (cur & 1) * AF_CHUNK_MAX;
  }
};

// max(z, 0) as ONE v_max_f32 (the C-level fmaxf adds a canonicalising v_max in front)
AF_DEV float af_relu(float z) { float v; asm("v_max_f32 %0, 0, %1" : "=v"(v) : "v"(z)); return v; }

// Accumulator initialisation = bias.  The net's padded bias rows ([NL][256] floats, <= 8 KB) are copied once per
// workgroup into LDS behind the two weight buffers: 32 ds_read_b128 per layer cost a fraction of the 32 VMEM loads
// they replace (each VMEM instruction steals ~40 cycles of MFMA issue from the only wave of its SIMD).
#define AF_BIAS_LDS (2 * AF_CHUNK_MAX)          // byte offset of the bias rows in dynamic LDS
#define AF_LDS_BYTES (2 * AF_CHUNK_MAX + AF_MAX_LAYERS * AF_HID * 4)
template <int NL> AF_DEV void stage_bias(const float* bias, char* smem, int tid) {
  float* dst = (float*)(smem + AF_BIAS_LDS);
#pragma unroll
  for (int i = 0; i < NL; ++i) dst[i * AF_HID + tid] = bias[i * AF_HID + tid];     // 256 threads x NL rows; visible after the first barrier
}
