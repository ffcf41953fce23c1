// dw.hip — weight/bias gradient GEMMs  dW_l = dZ_l^T X_l,  db_l = sum_rows dZ_l  (the dW half of
// `loss.backward()`, src/stage1_neural_atlas.py:230) for every layer of every net in ONE launch.
//
// The reduction dimension is the row batch (up to 90 000 rows), the output is at most 256x256, so the
// rows are split over workgroups (split-K) by a static, cost-balanced schedule built on the host: each
// workgroup walks a short list of segments (job, row-tile range) and writes one partial block per
// segment; k_adam sums the partials in a fixed order (deterministic, no float atomics).
//
// Per segment a workgroup keeps the WHOLE dW block in accumulators (4 waves x up to 16 tiles of 32x32 = all 256
// AGPRs), so both operands are streamed exactly once: 2 KB of HBM per row per 256x256 layer -> 64 FLOP/B,
// i.e. ~2.4 TB/s at the FP32-MFMA peak.
//
// Operand stream: a ring of DW_STAGES LDS slots, one STAGE = 16 rows (half a T-layout tile [feature][32 rows]) of
// both operands, filled by global_load_lds (16 B per lane, LDS image lane-linear, the bank swizzle applied on the
// SOURCE address).  Stage s+3 is issued while stage s is computed, so the `s_waitcnt vmcnt(N)` that publishes a
// stage only covers pieces issued one and a half stages (~12 K MFMA cycles for an 8x8 job) earlier — it never drains
// the younger stages (a one-tile-ahead double buffer with vmcnt(0) per tile left the matrix pipe idle 16 % of the
// time, profiles/r1c_pmc_sq.txt).  One s_barrier per stage, placed mid-stage (see dw_segment); inside a stage every
// LDS fragment read and every LDS-DMA issue sits behind its own MFMA (sched_group_barrier), never in a burst.
// MFMA: A[m = out feature][k = row], B[k = row][n = in feature]; four consecutive k of one lane half are
// one ds_read_b128.  db falls out of the A fragments for free.
#include <utility>
#include "af_dev.h"

#ifndef DW_ABL
#define DW_ABL 0     // tools/dwbench.hip timing ablations of k_dw_bf: bit0 no operand split (raw bits fed to the MFMAs), bit1 one MFMA per
#endif               // product instead of six, bit2 no explicit interleave (sched_group_barrier), bit3 packed-f32 subtract avoided
#ifndef DW_STAGES
#define DW_STAGES 4
#endif
#define DW_LDS (DW_STAGES * 32768)
#ifndef DW_PIPE
#define DW_PIPE 1    // software-pipelined operand split in k_dw_bf (0: the round-2 stage, kept for A/B timing in tools/dwbench.hip)
#endif

template <int N> AF_DEV void dw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// s_barrier without the release/acquire fences of __syncthreads(): a fence makes hipcc drain vmcnt to 0 in front of the
// barrier, which is exactly what the ring avoids.  Visibility of the LDS-DMA data is given by each wave's own counted
// vmcnt wait in front of the barrier; the "memory" clobbers keep the compiler from moving LDS reads across it.
AF_DEV void dw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int TO, int TI, int TOW, int TIW>
AF_DEV void dw_segment(const DwJob& jb, const DwSeg& sg, float* partial, char* smem, int tid, int wave, int lane,
                       int a0, int b0, bool store_w, bool store_db) {
  constexpr int A_B = TO * 2048;                         // bytes of the A half-tile (TO x 32 features x 16 rows)
  constexpr int NP = (TO + TI) * 128;                    // 16-B pieces per stage, A pieces first
  constexpr int NI = (NP + 255) / 256;                   // LDS-DMA instructions per wave per stage
  constexpr int SLOT = NI * 4096;                        // ring slot (>= the stage; the tail of an odd stage is padding)
  static_assert(DW_STAGES * SLOT <= DW_LDS, "ring does not fit");
  static_assert((DW_STAGES - 2) * NI < 64 && DW_STAGES >= 3, "vmcnt is a 6-bit counter");
  const int m = lane & 31, h = lane >> 5;
  f32x16 acc[TOW][TIW];
  float dbacc[TOW];
#pragma unroll
  for (int x = 0; x < TOW; ++x) {
    dbacc[x] = 0.f;
#pragma unroll
    for (int y = 0; y < TIW; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  }
  // Per-lane source of each of the NI pieces this lane moves per stage.  Piece i = 256 k + tid of a stage lands at LDS
  // byte 16 i of the slot: feature row f = li >> 2 (li = index inside its operand), physical 16-B slot li & 3, which
  // holds logical row group c = (li & 3) ^ ((f >> 2) & 3) — so the 16 lanes of a ds_read_b128 group hit 16 banks sets.
  const char* src[NI]; uint32_t tstr[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    int i = k * 256 + tid;
    bool is_a = i < TO * 128;
    int li = is_a ? i : i - TO * 128;
    if (i >= NP) { li = 0; is_a = true; }               // padding lanes of an odd stage: re-read piece 0 into the pad
    const int f = li >> 2, c = (li & 3) ^ ((f >> 2) & 3);
    src[k] = (const char*)(is_a ? jb.A : jb.B) + f * 128 + c * 16;
    tstr[k] = (is_a ? jb.a_stride : jb.b_stride) * 4u;
  }
  const int S = 2 * (sg.t1 - sg.t0);                     // stages of this segment
  auto issue = [&](int s, int k) {                       // piece k of stage s -> ring slot s & 3
    const int sc = s < S ? s : S - 1;                    // past the end: harmless re-stage (keeps the vmcnt arithmetic uniform)
    const char* g = src[k] + (size_t)(sg.t0 + (sc >> 1)) * tstr[k] + (sc & 1) * 64;
    af_glds16(g, smem + (s % DW_STAGES) * SLOT + k * 4096 + wave * 1024);
  };
#pragma unroll
  for (int s = 0; s < DW_STAGES - 1; ++s)
#pragma unroll
    for (int k = 0; k < NI; ++k) issue(s, k);

  const int sw = (m >> 2) & 3;
  int goff[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) goff[g] = m * 64 + (((2 * g + h) ^ sw) << 4);
  const int abase = a0 * 2048, bbase = A_B + b0 * 2048;

  // Software pipeline: the barrier that publishes stage s+1 sits in the MIDDLE of stage s, between its two k-groups.
  // Behind it the second group prefetches the first fragments of stage s+1 and issues the DMA of stage s+3 (into the
  // slot stage s-1 left: every wave is past stage s-1 once it is past this barrier), so neither the LDS latency of a
  // stage's first fragment reads nor the barrier skew between the four waves is ever in front of an idle matrix pipe.
  f32x4 af[2][TOW], bf[2][TIW];
  auto read_frags = [&](int buf, const char* slot, int g) {
#pragma unroll
    for (int x = 0; x < TOW; ++x) af[buf][x] = *(const f32x4*)(slot + abase + x * 2048 + goff[g]);
#pragma unroll
    for (int y = 0; y < TIW; ++y) bf[buf][y] = *(const f32x4*)(slot + bbase + y * 2048 + goff[g]);
  };
  auto group = [&](int g) {
#pragma unroll
    for (int x = 0; x < TOW; ++x) dbacc[x] += (af[g][x][0] + af[g][x][1]) + (af[g][x][2] + af[g][x][3]);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int x = 0; x < TOW; ++x)
#pragma unroll
        for (int y = 0; y < TIW; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][x][p], bf[g][y][p], acc[x][y], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4 * TOW * TIW; ++i) {            // one LDS read and one LDS-DMA issue behind each MFMA
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  dw_wait_vm<(DW_STAGES - 2) * NI>();                                    // stage 0 has landed
  dw_barrier();
  read_frags(0, smem, 0);
  for (int s = 0; s < S; ++s) {
    const char* slot = smem + (s % DW_STAGES) * SLOT;
    const char* nslot = smem + ((s + 1) % DW_STAGES) * SLOT;
    read_frags(1, slot, 1);
    group(0);
    dw_wait_vm<(DW_STAGES - 3) * NI>();                                      // stage s+1 has landed (stage s+2 may still be in flight) ...
    dw_barrier();                                        // ... for every wave; and every wave is done with stage s-1
    read_frags(0, nslot, 0);                             // past the last stage: a harmless read of a re-staged slot
#pragma unroll
    for (int k = 0; k < NI; ++k) issue(s + DW_STAGES - 1, k);
    group(1);
  }
  dw_wait_vm<0>();
  dw_barrier();      // everyone done reading LDS (and the trailing re-stages landed) before the next segment restages

  float* blk = partial + jb.part_off + (size_t)sg.slot * jb.part_blk;
  constexpr int pld = TI * 32;
  const auto rblk = af_rsrc_uniform(blk, jb.part_blk * 4);
  if (store_w) {
    const int voff = ((a0 * 32 + 4 * h) * pld + b0 * 32 + m) * 4;
#pragma unroll
    for (int x = 0; x < TOW; ++x)
#pragma unroll
      for (int y = 0; y < TIW; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          af_bs32(acc[x][y][r], rblk, voff, ((x * 32 + (r & 3) + 8 * (r >> 2)) * pld + y * 32) * 4);
  }
#pragma unroll
  for (int x = 0; x < TOW; ++x) {
    const float tot = dbacc[x] + __shfl_xor(dbacc[x], 32);
    if (store_db && h == 0) af_bs32(tot, rblk, (TO * 32 * pld + a0 * 32 + m) * 4, x * 128);
  }
}

// ------------------------------------------------------------------------------------------------
// The same contraction on the bf16 matrix pipe with fp32-faithful operands ("bf16x6"): every fp32 operand element is
// split in registers into three bf16 values hi + mid + lo (8 + 8 + 8 mantissa bits; the two residuals are exact fp32
// subtractions, so hi + mid + lo == x to within 2^-24 |x|), and a product a*b is accumulated as the six partial
// products hh + hm + mh + mm + hl + lh in the MFMA's fp32 accumulator.  The dropped terms (ml, lm, ll) are <= 2^-23
// |ab| — the size of the rounding of ONE fp32 multiply — so the result carries fp32-level round-off (measured against
// fp64 on a 90 000-row contraction: 1.5e-7 relative, vs 2.3e-7 for a plain fp32 GEMM; tests/test_split_precision.py
// restates the arithmetic on the CPU).  v_mfma_f32_32x32x16_bf16 retires K = 16 in 32 cycles where the fp32 MFMA needs
// 8 x 64: six of them per 16 rows are 2.7x faster, which moves this kernel from the matrix pipe onto HBM (2 KB per row
// per 256x256 layer, read once).
// One stage of the LDS ring (16 rows) is exactly one k-step: lane (m, h) holds rows 8h..8h+7 of feature m — two
// ds_read_b128 of the T-layout half tile per operand tile.
#include "bfsplit.h"

// NM MFMAs, each followed by its share of NV VALU instructions, at most one LDS read and one VMEM instruction
template <int NM, int NV, int... I>
AF_DEV void dw_sgb_spread(std::integer_sequence<int, I...>) {
  ((__builtin_amdgcn_sched_group_barrier(0x008, 1, 0),
    __builtin_amdgcn_sched_group_barrier(0x002, (NV * (I + 1)) / NM - (NV * I) / NM, 0),
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0),
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0)), ...);
}

template <int TO, int TI, int TOW, int TIW, int NPROD = 6>
AF_DEV void dw_segment_bf(const DwJob& jb, const DwSeg& sg, float* partial, char* smem, int tid, int wave, int lane,
                          int a0, int b0, bool store_w, bool store_db) {
  constexpr int LV = NPROD == 6 ? 3 : 2;                  // split levels: six products need hi + mid + lo, three only hi + mid
  constexpr int A_B = TO * 2048;
  constexpr int NP = (TO + TI) * 128;
  constexpr int NI = (NP + 255) / 256;
  constexpr int SLOT = NI * 4096;
  static_assert(DW_STAGES * SLOT <= DW_LDS, "ring does not fit");
  static_assert((DW_STAGES - 2) * NI < 64 && DW_STAGES >= 3, "vmcnt is a 6-bit counter");
  const int m = lane & 31, h = lane >> 5;
  f32x16 acc[TOW][TIW];
  float dbacc[TOW];
#pragma unroll
  for (int x = 0; x < TOW; ++x) {
    dbacc[x] = 0.f;
#pragma unroll
    for (int y = 0; y < TIW; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  }
  // same ring / source swizzle as dw_segment; per lane only a 32-bit offset per piece (the operand an instruction reads
  // is wave-uniform except for the one instruction that straddles A|B when TO is odd)
  uint32_t soff[NI]; bool sel_a[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    int i = k * 256 + tid;
    bool is_a = i < TO * 128;
    int li = is_a ? i : i - TO * 128;
    if (i >= NP) { li = 0; is_a = true; }
    const int f = li >> 2, c = (li & 3) ^ ((f >> 2) & 3);
    soff[k] = (uint32_t)(f * 128 + c * 16);
    sel_a[k] = is_a;
  }
  const int S = 2 * (sg.t1 - sg.t0);
  auto issue = [&](int s, int k) {
    const int sc = s < S ? s : S - 1;
    const size_t t = (size_t)(sg.t0 + (sc >> 1));
    const char* ga = (const char*)jb.A + t * jb.a_stride * 4u + (sc & 1) * 64;      // wave-uniform
    const char* gb = (const char*)jb.B + t * jb.b_stride * 4u + (sc & 1) * 64;
    const bool all_a = (k + 1) * 256 <= TO * 128, all_b = k * 256 >= TO * 128 && (k + 1) * 256 <= NP;
    const char* g = all_a ? ga : (all_b ? gb : (sel_a[k] ? ga : gb));
    af_glds16(g + soff[k], smem + (s % DW_STAGES) * SLOT + k * 4096 + wave * 1024);
  };
#pragma unroll
  for (int s = 0; s < DW_STAGES - 1; ++s)
#pragma unroll
    for (int k = 0; k < NI; ++k) issue(s, k);

  const int sw = (m >> 2) & 3;
  const int off0 = m * 64 + (((2 * h) ^ sw) << 4), off1 = m * 64 + (((2 * h + 1) ^ sw) << 4);     // rows 8h..8h+3, 8h+4..8h+7
  const int abase = a0 * 2048, bbase = A_B + b0 * 2048;
#if DW_PIPE
  // Software-pipelined stage: every VALU instruction of the operand split sits in the shadow of an MFMA (one wave per SIMD
  // issues in order: a clump of VALU in front of the products is time the matrix pipe idles).  Entering stage s a wave holds
  // the SPLIT A operands of the stage (sa, produced during stage s-1), the raw B fragments of the stage (rb) and the split
  // of B column 0 (sb).  The stage publishes stage s+1 (counted wait + barrier), reads its raw A fragments and issues the
  // DMA of stage s+3, then runs column by column: behind the NPROD x TOW products of column y go the split of B column y+1
  // (column 0 of stage s+1 for the last), the split of A fragment y of stage s+1 (all of them spread over the columns when
  // TOW != TIW) and the refresh of column y's raw registers for stage s+1.
  f32x4 ran[TOW][2];                                     // raw A fragments of the NEXT stage
  f32x4 rb[TIW][2];                                      // raw B fragments: column y is refreshed for the next stage while column y runs
  DwSplit sa[2][TOW];                                    // split A operands: [cur] this stage, [cur ^ 1] the next (filled during this one)
  DwSplit sb;
  auto read_an = [&](const char* slot) {
#pragma unroll
    for (int x = 0; x < TOW; ++x) { ran[x][0] = *(const f32x4*)(slot + abase + x * 2048 + off0); ran[x][1] = *(const f32x4*)(slot + abase + x * 2048 + off1); }
  };
  auto read_b = [&](int y, const char* slot) {
    rb[y][0] = *(const f32x4*)(slot + bbase + y * 2048 + off0); rb[y][1] = *(const f32x4*)(slot + bbase + y * 2048 + off1);
  };
  auto split_a = [&](int buf, int x, bool real) {      // real == false: the stage behind the last one (a re-staged copy): its rows must not reach db
    sa[buf][x] = dw_split8<LV>(ran[x][0], ran[x][1]);
    const f32x4 t = ran[x][0] + ran[x][1];
    const float rs = (t[0] + t[1]) + (t[2] + t[3]);
    dbacc[x] += real ? rs : 0.f;
  };
  constexpr int APC = (TOW + TIW - 1) / TIW;             // A fragments split per column
  auto stage = [&](int s, int cur) {
    const char* nslot = smem + ((s + 1) % DW_STAGES) * SLOT;      // past the last stage: harmless reads of a re-staged slot
    dw_wait_vm<(DW_STAGES - 3) * NI>();                  // stage s+1 has landed (stage s+2 may still be in flight) ...
    dw_barrier();                                        // ... for every wave; every wave holds what it needs of stage s-1
    read_an(nslot);
#pragma unroll
    for (int k = 0; k < NI; ++k) issue(s + DW_STAGES - 1, k);      // into the slot stage s-1 left
#pragma unroll
    for (int y = 0; y < TIW; ++y) {
      DwSplit sbn;
      read_b(y, nslot);                                  // column y of stage s was split one column ago: its registers take stage s+1
      if (y + 1 < TIW) sbn = dw_split8<LV>(rb[y + 1][0], rb[y + 1][1]);
#pragma unroll
      for (int i = 0; i < APC; ++i) if (y * APC + i < TOW) split_a(cur ^ 1, y * APC + i, s + 1 < S);
      if (y + 1 == TIW) sbn = dw_split8<LV>(rb[0][0], rb[0][1]);      // stage s+1's column 0 (read behind column 0 of this stage)
      if constexpr (!(DW_ABL & 2)) {
        if constexpr (NPROD == 6) {
#pragma unroll
          for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[cur][x].h, sb.l, acc[x][y]);
#pragma unroll
          for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[cur][x].l, sb.h, acc[x][y]);
#pragma unroll
          for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[cur][x].m, sb.m, acc[x][y]);
        }
#pragma unroll
        for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[cur][x].h, sb.m, acc[x][y]);
#pragma unroll
        for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[cur][x].m, sb.h, acc[x][y]);
      }
#pragma unroll
      for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[cur][x].h, sb.h, acc[x][y]);
      if constexpr (!(DW_ABL & 4)) {
        // per MFMA: itself, then its share of the column's VALU (split of one B column + APC A fragments, ~45 each at three levels),
        // at most one LDS read and one DMA issue
        constexpr int NM = NPROD * TOW, NV = (LV == 3 ? 46 : 28) * (1 + APC);
        dw_sgb_spread<NM, NV>(std::make_integer_sequence<int, NM>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      sb = sbn;
    }
  };
  dw_wait_vm<(DW_STAGES - 2) * NI>();                                    // stage 0 has landed
  dw_barrier();
  read_an(smem);
#pragma unroll
  for (int y = 0; y < TIW; ++y) read_b(y, smem);
#pragma unroll
  for (int x = 0; x < TOW; ++x) split_a(0, x, true);
  sb = dw_split8<LV>(rb[0][0], rb[0][1]);
  for (int s = 0; s < S; s += 2) {                       // S is even (two stages per 32-row tile): no register copies between stages
    stage(s, 0);
    stage(s + 1, 1);
  }
#else
  f32x4 ra[2][TOW][2];                                   // raw A fragments, double-buffered across stages
  f32x4 rb[TIW][2];                                      // raw B fragments: column y is refreshed for the next stage while column y runs
  auto read_a = [&](int buf, const char* slot) {
#pragma unroll
    for (int x = 0; x < TOW; ++x) { ra[buf][x][0] = *(const f32x4*)(slot + abase + x * 2048 + off0); ra[buf][x][1] = *(const f32x4*)(slot + abase + x * 2048 + off1); }
  };
  auto read_b = [&](int y, const char* slot) {
    rb[y][0] = *(const f32x4*)(slot + bbase + y * 2048 + off0); rb[y][1] = *(const f32x4*)(slot + bbase + y * 2048 + off1);
  };
  // One stage: split this stage's A operands, publish stage s+1 (counted wait + barrier), start the reads of its raw A
  // fragments and the DMA of stage s+3, then column by column: the 6 x TOW products of column y run with the split of
  // column y+1 and the refresh of column y's raw registers (stage s+1) in their shadow.
  auto stage = [&](int s, int cur) {
    const char* nslot = smem + ((s + 1) % DW_STAGES) * SLOT;      // past the last stage: harmless reads of a re-staged slot
    DwSplit sa[TOW];
#pragma unroll
    for (int x = 0; x < TOW; ++x) {
      sa[x] = dw_split8<LV>(ra[cur][x][0], ra[cur][x][1]);
      const f32x4 t = ra[cur][x][0] + ra[cur][x][1];
      dbacc[x] += (t[0] + t[1]) + (t[2] + t[3]);
    }
    DwSplit sb = dw_split8<LV>(rb[0][0], rb[0][1]);
    dw_wait_vm<(DW_STAGES - 3) * NI>();                                      // stage s+1 has landed (stage s+2 may still be in flight) ...
    dw_barrier();                                        // ... for every wave; every wave holds what it needs of stage s-1
    read_a(cur ^ 1, nslot);
#pragma unroll
    for (int k = 0; k < NI; ++k) issue(s + DW_STAGES - 1, k);      // into the slot stage s-1 left
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int y = 0; y < TIW; ++y) {
      DwSplit sbn = sb;
      if (y + 1 < TIW) sbn = dw_split8<LV>(rb[y + 1][0], rb[y + 1][1]);
      read_b(y, nslot);                                  // column y of stage s was split one region ago: its registers take stage s+1
      if constexpr (!(DW_ABL & 2)) {
        if constexpr (NPROD == 6) {
#pragma unroll
          for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[x].h, sb.l, acc[x][y]);
#pragma unroll
          for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[x].l, sb.h, acc[x][y]);
#pragma unroll
          for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[x].m, sb.m, acc[x][y]);
        }
#pragma unroll
        for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[x].h, sb.m, acc[x][y]);
#pragma unroll
        for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[x].m, sb.h, acc[x][y]);
      }
#pragma unroll
      for (int x = 0; x < TOW; ++x) acc[x][y] = dw_mfma_bf(sa[x].h, sb.h, acc[x][y]);
      if constexpr (!(DW_ABL & 4)) {
#pragma unroll
        for (int i = 0; i < NPROD * TOW; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      sb = sbn;
    }
  };
  dw_wait_vm<(DW_STAGES - 2) * NI>();                                    // stage 0 has landed
  dw_barrier();
  read_a(0, smem);
#pragma unroll
  for (int y = 0; y < TIW; ++y) read_b(y, smem);
  for (int s = 0; s < S; s += 2) {                       // S is even (two stages per 32-row tile): no register copies between stages
    stage(s, 0);
    stage(s + 1, 1);
  }
#endif
  dw_wait_vm<0>();
  dw_barrier();

  float* blk = partial + jb.part_off + (size_t)sg.slot * jb.part_blk;
  constexpr int pld = TI * 32;
  const auto rblk = af_rsrc_uniform(blk, jb.part_blk * 4);
  if (store_w) {
    const int voff = ((a0 * 32 + 4 * h) * pld + b0 * 32 + m) * 4;
#pragma unroll
    for (int x = 0; x < TOW; ++x)
#pragma unroll
      for (int y = 0; y < TIW; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          af_bs32(acc[x][y][r], rblk, voff, ((x * 32 + (r & 3) + 8 * (r >> 2)) * pld + y * 32) * 4);
  }
#pragma unroll
  for (int x = 0; x < TOW; ++x) {
    const float tot = dbacc[x] + __shfl_xor(dbacc[x], 32);
    if (store_db && h == 0) af_bs32(tot, rblk, (TO * 32 * pld + a0 * 32 + m) * 4, x * 128);
  }
}

// Compacted batches (DwJob::live_rows, written by k_prep): row tiles behind the last live row hold what an earlier iteration
// left there.  Clip the segment to the live tiles; a segment with nothing left stores the zero block k_adam expects in its slot.
AF_DEV bool dw_clip(const DwJob& jb, DwSeg& sg, float* partial, int tid) {
  if (!jb.live_rows) return true;
  const int nt = (jb.live_base + *jb.live_rows + 31) >> 5;
  if (sg.t1 > nt) sg.t1 = nt;
  if (sg.t0 < sg.t1) return true;
  float* blk = partial + jb.part_off + (size_t)sg.slot * jb.part_blk;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (uint32_t i = (uint32_t)tid * 4u; i < jb.part_blk; i += 1024u) *(f32x4*)(blk + i) = z;
  return false;
}

template <int NPROD>
__global__ __launch_bounds__(256, 1) void k_dw_bf(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const DwSeg* segs = a.segs + (size_t)blockIdx.x * DW_MAXSEG;
#ifdef DW_CLK      // tools/dwbench.hip: core clock ticks AND the 100 MHz counter ([4] per workgroup: ticks0, ticks1, real0, real1)
  if (a.wg_clock && tid == 0) { a.wg_clock[blockIdx.x * 4] = __builtin_amdgcn_s_memtime(); a.wg_clock[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime(); }
#else
  if (a.wg_clock && tid == 0) a.wg_clock[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
#endif
  for (int s = 0; s < DW_MAXSEG; ++s) {
    DwSeg sg = segs[s];
    if (sg.job < 0) break;
    const DwJob jb = a.jobs[sg.job];
    if (!dw_clip(jb, sg, a.partial, tid)) continue;
    switch (jb.shape) {      // 8x8: each wave a 4x4 block of output tiles (8 operand tiles to read and split per stage, the minimum)
      case DW_8x8: dw_segment_bf<8, 8, 4, 4, NPROD>(jb, sg, a.partial, smem, tid, wave, lane, 4 * (wave & 1), 4 * (wave >> 1), true, wave < 2); break;
      case DW_8x2: dw_segment_bf<8, 2, 2, 2, NPROD>(jb, sg, a.partial, smem, tid, wave, lane, 2 * wave, 0, true, true); break;
      case DW_8x1: dw_segment_bf<8, 1, 2, 1, NPROD>(jb, sg, a.partial, smem, tid, wave, lane, 2 * wave, 0, true, true); break;
      case DW_1x8: dw_segment_bf<1, 8, 1, 2, NPROD>(jb, sg, a.partial, smem, tid, wave, lane, 0, 2 * wave, true, wave == 0); break;
      case DW_1x2: dw_segment_bf<1, 2, 1, 1, NPROD>(jb, sg, a.partial, smem, tid, wave, lane, 0, wave & 1, wave < 2, wave == 0); break;
      default: break;
    }
  }
#ifdef DW_CLK
  if (a.wg_clock && tid == 0) { a.wg_clock[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); a.wg_clock[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#else
  if (a.wg_clock && tid == 0) a.wg_clock[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

__global__ __launch_bounds__(256, 1) void k_dw(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const DwSeg* segs = a.segs + (size_t)blockIdx.x * DW_MAXSEG;
  if (a.wg_clock && tid == 0) a.wg_clock[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < DW_MAXSEG; ++s) {
    DwSeg sg = segs[s];
    if (sg.job < 0) break;
    const DwJob jb = a.jobs[sg.job];
    if (!dw_clip(jb, sg, a.partial, tid)) continue;
    switch (jb.shape) {
      case DW_8x8: dw_segment<8, 8, 2, 8>(jb, sg, a.partial, smem, tid, wave, lane, 2 * wave, 0, true, true); break;
      case DW_8x2: dw_segment<8, 2, 2, 2>(jb, sg, a.partial, smem, tid, wave, lane, 2 * wave, 0, true, true); break;
      case DW_8x1: dw_segment<8, 1, 2, 1>(jb, sg, a.partial, smem, tid, wave, lane, 2 * wave, 0, true, true); break;
      case DW_1x8: dw_segment<1, 8, 1, 2>(jb, sg, a.partial, smem, tid, wave, lane, 0, 2 * wave, true, wave == 0); break;
      case DW_1x2: dw_segment<1, 2, 1, 1>(jb, sg, a.partial, smem, tid, wave, lane, 0, wave & 1, wave < 2, wave == 0); break;
      default: break;
    }
  }
  if (a.wg_clock && tid == 0) a.wg_clock[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
}

// mode 0: fp32 matrix pipe (v_mfma_f32_32x32x2_f32); mode 1: bf16x6 split operands on the bf16 matrix pipe; mode 2: bf16x3 (hi + mid, three products)
extern "C" int af_launch_dw(const DwArgs* a, int nwg, int mode, hipStream_t s) {
  if (mode == 0)      hipLaunchKernelGGL(k_dw, dim3(nwg), dim3(256), DW_LDS, s, *a);
  else if (mode == 1) hipLaunchKernelGGL(k_dw_bf<6>, dim3(nwg), dim3(256), DW_LDS, s, *a);
  else                hipLaunchKernelGGL(k_dw_bf<3>, dim3(nwg), dim3(256), DW_LDS, s, *a);
  return (int)hipGetLastError();
}
extern "C" int af_dw_init() {
  hipError_t e = hipFuncSetAttribute((const void*)k_dw, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute((const void*)k_dw_bf<6>, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
  if (e != hipSuccess) return (int)e;
  return (int)hipFuncSetAttribute((const void*)k_dw_bf<3>, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
}
