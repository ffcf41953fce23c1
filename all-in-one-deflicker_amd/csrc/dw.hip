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
// The operand tiles are read once per step (1.3 GB, written by the chains with nt stores) and the partial blocks are written once and read once by
// k_adam: the non-temporal policy on both (gfx940+ nt bit) is an experiment switch measured on the real step (tools/experiments/README.md).
#ifndef DW_LOAD_NT
#define DW_LOAD_NT 0
#endif
#ifndef DW_STORE_NT
#define DW_STORE_NT 1
#endif
AF_DEV void dw_glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, DW_LOAD_NT ? 2 : 0);
}
AF_DEV void dw_bs32(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, DW_STORE_NT ? 2 : 0);
}

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
    dw_glds16(g, smem + (s % DW_STAGES) * SLOT + k * 4096 + wave * 1024);
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
          dw_bs32(acc[x][y][r], rblk, voff, ((x * 32 + (r & 3) + 8 * (r >> 2)) * pld + y * 32) * 4);
  }
#pragma unroll
  for (int x = 0; x < TOW; ++x) {
    const float tot = dbacc[x] + __shfl_xor(dbacc[x], 32);
    if (store_db && h == 0) dw_bs32(tot, rblk, (TO * 32 * pld + a0 * 32 + m) * 4, x * 128);
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

// ---- the 8x8 six-product stage as 96 hand-placed issue slots (round 4) -----------------------------------------------------------
// One wave per SIMD issues in order; an MFMA holds the matrix pipe for 32 cycles and hides at most five light instructions behind
// it (MI355X_MICROARCH.md: a hand-placed stream with exactly five fillers per gap runs 32.4 cycles per MFMA, the floor; six cost a
// whole extra issue slot).  The compiler-scheduled stage (dw_sgb_spread below, kept for the other shapes) came out at 4.4 VALU per
// MFMA on average but unevenly — 100 of 192 gaps with five VALU plus SALU / LDS / DMA on top, clumps of 11-26 behind the barrier,
// eight DMA pieces in three gaps, ~30 VALU of per-lane 64-bit address arithmetic per stage — and ran 40.8 cycles per MFMA.  Here
// every slot is written out: slot = one MFMA + its fillers from dw_slots.h (tools/dw_slots_gen.py), fenced by sched_barrier(0):
//   * the split is a stream of single-instruction micro-ops (dw_split_op), two streams per column (B column y+1, A fragment y of the
//     next stage + its bias-gradient row sum), interleaved A/B inside a slot so that no instruction waits on its predecessor;
//   * raw-fragment LDS reads one per slot; LDS-DMA pieces one per slot, almost alone, in the `voff, s[base]` form (inline asm: the
//     builtin takes a per-lane 64-bit address) — two per column instead of eight behind the barrier;
//   * the split B operand ping-pongs between two register sets (no copies at the column end).
// The MFMA order per accumulator is the compiler-scheduled kernel's: results are bit-identical to it (tools/dwbench.hip checks).
#ifndef DW_SLOT
#define DW_SLOT 1
#endif
#include "dw_slots.h"
struct DwStream { float t[2][2]; float u; };           // residuals of the two pairs a split stream has in flight (by pair parity) + one widened half
#define DW_PIN(x) asm volatile("" : "+v"(x))
// Micro-op K (0..43) of dw_split8<3>(lo4, hi4): one VALU instruction, tied to where it is written.  Per pair of values (a, b) the
// split is  A h = cvt_pk(a, b) | B r0 = h << 16 | C r1 = h & 0xffff0000 | D r0 = a - r0 | E r1 = b - r1 | F m = cvt_pk(r0, r1) |
// G u = m << 16 | H r0 -= u | I u = m & 0xffff0000 | J r1 -= u | K l = cvt_pk(r0, r1).  v_cvt_pk_bf16_f32 needs a wait state on
// either side (hipcc pads a cvt next to its producer / consumer with s_nop - an issue slot gone), so a stream runs the second half of
// pair i in lock-step with the first half of pair i + 1:  A0 B0 C0 D0 E0 | F0 A1 G0 B1 H0 C1 I0 D1 J0 E1 K0 | F1 A2 ... K2 | F3 G3 H3 I3 J3 K3.
struct DwOp { int pair, ph; };
constexpr DwOp dw_op_of(int K) {
  if (K < 5) return DwOp{0, K};
  if (K >= 38) return DwOp{3, 5 + (K - 38)};
  const int i = (K - 5) / 11, j = (K - 5) % 11;
  return (j & 1) ? DwOp{i + 1, j / 2} : DwOp{i, 5 + j / 2};
}
template <int K> AF_DEV void dw_split_op(const f32x4& lo4, const f32x4& hi4, DwSplit& o, DwStream& st) {
  constexpr DwOp op = dw_op_of(K);
  constexpr int i = op.pair, ph = op.ph, e = i & 1;
  const float a = i < 2 ? lo4[2 * i] : hi4[2 * i - 4], b = i < 2 ? lo4[2 * i + 1] : hi4[2 * i - 3];
  float (&t)[2] = st.t[e];
  if constexpr (ph == 0)       { uint32_t h = dw_pk(a, b); DW_PIN(h); o.h[i] = h; }
  else if constexpr (ph == 1)  { float v = __builtin_bit_cast(float, o.h[i] << 16); DW_PIN(v); t[0] = v; }
  else if constexpr (ph == 2)  { float v = __builtin_bit_cast(float, o.h[i] & 0xffff0000u); DW_PIN(v); t[1] = v; }
  else if constexpr (ph == 3)  { float v = a - t[0]; DW_PIN(v); t[0] = v; }
  else if constexpr (ph == 4)  { float v = b - t[1]; DW_PIN(v); t[1] = v; }
  else if constexpr (ph == 5)  { uint32_t m = dw_pk(t[0], t[1]); DW_PIN(m); o.m[i] = m; }
  else if constexpr (ph == 6)  { float v = __builtin_bit_cast(float, o.m[i] << 16); DW_PIN(v); st.u = v; }
  else if constexpr (ph == 7)  { float v = t[0] - st.u; DW_PIN(v); t[0] = v; }
  else if constexpr (ph == 8)  { float v = __builtin_bit_cast(float, o.m[i] & 0xffff0000u); DW_PIN(v); st.u = v; }
  else if constexpr (ph == 9)  { float v = t[1] - st.u; DW_PIN(v); t[1] = v; }
  else                         { uint32_t l = dw_pk(t[0], t[1]); DW_PIN(l); o.l[i] = l; }
}
// The A stream = the 44 split micro-ops with the nine micro-ops of the bias-gradient row sum placed between the pair blocks (left at the
// end they are one serial chain in slots no other stream fills: a dependent VALU instruction right behind its producer costs a wait state):
// db += ((t0 + t1) + (t2 + t3)) with t = lo4 + hi4 (the 8 rows of the raw fragment) — dw_segment_bf's association.
// position -> (kind 0: split op K, 1: row-sum op K)
struct DwAOp { int kind, k; };
constexpr DwAOp dw_aop_of(int P) {
  //  0..4 A0..E0 | 5,6 s0 s1 | 7..17 block 0 | 18,19 s2 s3 | 20..30 block 1 | 31,32 s4 s5 | 33..43 block 2 | 44 s6 | 45 F3 | 46 s7 | 47..50 G3..J3 | 51 s8 | 52 K3
  if (P < 5) return DwAOp{0, P};
  if (P < 7) return DwAOp{1, P - 5};
  if (P < 18) return DwAOp{0, P - 2};
  if (P < 20) return DwAOp{1, P - 16};
  if (P < 31) return DwAOp{0, P - 4};
  if (P < 33) return DwAOp{1, P - 27};
  if (P < 44) return DwAOp{0, P - 6};
  if (P == 44) return DwAOp{1, 6};
  if (P == 45) return DwAOp{0, 38};
  if (P == 46) return DwAOp{1, 7};
  if (P < 51) return DwAOp{0, P - 8};
  if (P == 51) return DwAOp{1, 8};
  return DwAOp{0, 43};
}
struct DwSum { float e0, e1, e2; };
template <int K> AF_DEV void dw_db_op(const f32x4& lo4, const f32x4& hi4, float& db, DwSum& e, bool real) {
  if constexpr (K == 0)      { float v = lo4[0] + hi4[0]; DW_PIN(v); e.e0 = v; }                  // t0
  else if constexpr (K == 1) { float v = lo4[1] + hi4[1]; DW_PIN(v); e.e1 = v; }                  // t1
  else if constexpr (K == 2) { float v = e.e0 + e.e1; DW_PIN(v); e.e0 = v; }                      // t0 + t1
  else if constexpr (K == 3) { float v = lo4[2] + hi4[2]; DW_PIN(v); e.e1 = v; }                  // t2
  else if constexpr (K == 4) { float v = lo4[3] + hi4[3]; DW_PIN(v); e.e2 = v; }                  // t3
  else if constexpr (K == 5) { float v = e.e1 + e.e2; DW_PIN(v); e.e1 = v; }                      // t2 + t3
  else if constexpr (K == 6) { float v = e.e0 + e.e1; DW_PIN(v); e.e0 = v; }
  else if constexpr (K == 7) { float v = real ? e.e0 : 0.f; DW_PIN(v); e.e0 = v; }                // the stage behind the last one is a re-staged copy: its rows must not reach db
  else                       { float v = db + e.e0; DW_PIN(v); db = v; }
}
struct Dw88 {
  f32x4 ran[4][2];                                     // raw A fragments of the NEXT stage
  f32x4 rb[4][2];                                      // raw B fragments: column y takes the next stage's while column y runs
  DwSplit sa[2][4];                                    // split A operands: [cur] this stage, [cur ^ 1] the next
  DwSplit sb[2];                                       // split B operand of column y in sb[y & 1]; column y fills sb[(y + 1) & 1]
  float db[4];
  DwStream ta, tb;
  DwSum es;
  uint32_t voff[2][4];                                 // per-lane source offset of piece k & 3 of a stage (the same for the A and the B operand), [half of the row tile]
  const char* ga; const char* gb;                      // wave-uniform source (row tile) of the stage being staged three ahead, A and B operand
  uint32_t sa4, sb4;                                   // bytes per row tile of the two operands
  int tiles_left;                                      // row tiles the source may still advance by (0: it stays on the segment's last tile - the re-stages past the end)
  uint32_t lds0, ring;                                 // wave-uniform LDS address of piece 0 of ring slot 0; byte offset of the ring slot being staged
  int abase, bbase, off0, off1;
};
// one LDS-DMA piece: 16 B per lane from s[base] + voff to LDS m0 + lane * 16 (one wait state between the m0 write and the DMA)
// (no immediate offset: the instruction adds it to the LDS address as well as to the global one; m0 is named as clobbered on purpose —
// hipcc keeps its own LDS-DMA destinations there in the other job shapes — which clang flags as a reserved register)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int IMM> AF_DEV void dw_glds_s(uint32_t voff, const char* sbase, uint32_t lds) {
#if DW_LOAD_NT
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(voff), "s"(sbase), "s"(lds), "n"(IMM) : "memory", "scc", "m0");
#else
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds), "n"(IMM) : "memory", "scc", "m0");
#endif
}
#pragma clang diagnostic pop
template <int CUR, int Y, int I>
AF_DEV void dw_slot88(Dw88& q, f32x16 (&acc)[4][4], const char* nslot, bool real) {
  constexpr int x = I & 3, p = I >> 2;
  constexpr const int* NA = Y == 0 ? DW_NA_Z : (Y == 3 ? DW_NA_H : DW_NA_G);
  constexpr const int* NB = Y == 0 ? DW_NB_Z : (Y == 3 ? DW_NB_H : DW_NB_G);
  constexpr const int* LD = Y == 0 ? DW_LDS_Z : (Y == 3 ? DW_LDS_H : DW_LDS_G);
  constexpr const int* DM = Y == 0 ? DW_DMA_Z : (Y == 3 ? DW_DMA_H : DW_DMA_G);
  constexpr int a0 = [] { int s = 0; for (int i = 0; i < I; ++i) s += (Y == 0 ? DW_NA_Z : (Y == 3 ? DW_NA_H : DW_NA_G))[i]; return s; }();
  constexpr int b0 = [] { int s = 0; for (int i = 0; i < I; ++i) s += (Y == 0 ? DW_NB_Z : (Y == 3 ? DW_NB_H : DW_NB_G))[i]; return s; }();
  constexpr int na = NA[I], nb = NB[I];
  const DwSplit& A = q.sa[CUR][x];
  const DwSplit& B = q.sb[Y & 1];
  // ---- the MFMA: products in the order h*l, l*h, m*m, h*m, m*h, h*h (smallest first), four output rows of tiles each
  if constexpr (p == 0)      acc[x][Y] = dw_mfma_bf(A.h, B.l, acc[x][Y]);
  else if constexpr (p == 1) acc[x][Y] = dw_mfma_bf(A.l, B.h, acc[x][Y]);
  else if constexpr (p == 2) acc[x][Y] = dw_mfma_bf(A.m, B.m, acc[x][Y]);
  else if constexpr (p == 3) acc[x][Y] = dw_mfma_bf(A.h, B.m, acc[x][Y]);
  else if constexpr (p == 4) acc[x][Y] = dw_mfma_bf(A.m, B.h, acc[x][Y]);
  else                       acc[x][Y] = dw_mfma_bf(A.h, B.h, acc[x][Y]);
  // An MFMA whose result has no further use inside the loop body (the last product of a stage's last visit to its accumulator) has
  // no successor in the block's DAG, and the instruction selector's bottom-up list scheduler then emits it - and the products in
  // front of it on the same accumulator - at the BOTTOM of the block, below every fence: 88 MFMAs in one run.  Tie the result to the slot.
  asm volatile("" : "+a"(acc[x][Y]));
  __builtin_amdgcn_sched_barrier(0);                   // the MFMA leads its slot: fillers that drift in front of it leave two MFMAs back to back further down
  if constexpr (Y == 0 && I == 0) {                    // the stage's first MFMA is in the pipe: now publish the next stage
    dw_wait_vm<(DW_STAGES - 3) * 8>();                 // stage s+1 has landed (stage s+2 may still be in flight) ...
    dw_barrier();                                      // ... for every wave; every wave holds what it needs of stage s-1
  }
  if constexpr (Y == 0 && I == 6) {                    // source / destination of the stage staged by this one (three ahead), before its first piece (slot 8)
    // Stage n = s + 3 is half (n & 1) of row tile min(n >> 1, last): the half lives in the per-lane offset (CUR == 0 stages an odd
    // stage, CUR == 1 an even one; a re-stage past the end may take either half), the tile advances once per two stages.
    if constexpr (CUR == 1) {
      const bool adv = q.tiles_left > 0;
      q.ga += adv ? q.sa4 : 0u; q.gb += adv ? q.sb4 : 0u;
      q.tiles_left -= 1;
    }
    q.ring = (q.ring + 0x8000u) & 0x18000u;
  }
  // ---- one raw-fragment read of the next stage
  if constexpr (LD[I] == 1) q.ran[0][0] = *(const f32x4*)(nslot + q.abase + q.off0);
  if constexpr (LD[I] == 2) q.ran[0][1] = *(const f32x4*)(nslot + q.abase + q.off1);
  if constexpr (LD[I] == 3) q.rb[Y][0] = *(const f32x4*)(nslot + q.bbase + Y * 2048 + q.off0);
  if constexpr (LD[I] == 4) q.rb[Y][1] = *(const f32x4*)(nslot + q.bbase + Y * 2048 + q.off1);
  if constexpr (LD[I] == 5) q.ran[(Y + 1) & 3][0] = *(const f32x4*)(nslot + q.abase + ((Y + 1) & 3) * 2048 + q.off0);
  if constexpr (LD[I] == 6) q.ran[(Y + 1) & 3][1] = *(const f32x4*)(nslot + q.abase + ((Y + 1) & 3) * 2048 + q.off1);
  // ---- one LDS-DMA piece of the stage three ahead: column y moves pieces 2y and 2y + 1 (0..3: the A half, 4..7: the B half)
  if constexpr (DM[I] != 0) {
    constexpr int k = 2 * Y + DM[I] - 1;
    dw_glds_s<k * 4096>(q.voff[CUR == 0 ? 1 : 0][k & 3], k < 4 ? q.ga : q.gb, q.lds0 + q.ring);
  }
  // ---- split micro-ops: A stream (fragment Y of the next stage -> sa[CUR ^ 1][Y], then its row sum -> db[Y]) and B stream
  // (column Y + 1, or the next stage's column 0 -> sb[(Y + 1) & 1]), interleaved
  constexpr int YN = (Y + 1) & 3;
  auto a_op = [&](auto kc) {
    constexpr DwAOp o = dw_aop_of(decltype(kc)::value);
    if constexpr (o.kind == 0) dw_split_op<o.k>(q.ran[Y][0], q.ran[Y][1], q.sa[CUR ^ 1][Y], q.ta);
    else                       dw_db_op<o.k>(q.ran[Y][0], q.ran[Y][1], q.db[Y], q.es, real);
  };
  auto b_op = [&](auto kc) { dw_split_op<decltype(kc)::value>(q.rb[YN][0], q.rb[YN][1], q.sb[(Y + 1) & 1], q.tb); };
  auto both = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < na) a_op(std::integral_constant<int, a0 + j>{});
    if constexpr (j < nb) b_op(std::integral_constant<int, b0 + j>{});
  };
  both(std::integral_constant<int, 0>{}); both(std::integral_constant<int, 1>{}); both(std::integral_constant<int, 2>{});
  both(std::integral_constant<int, 3>{}); both(std::integral_constant<int, 4>{});
  static_assert(na <= 5 && nb <= 5, "slot plan");
  __builtin_amdgcn_sched_barrier(0);
}
template <int CUR, int Y, int... I>
AF_DEV void dw_col88(Dw88& q, f32x16 (&acc)[4][4], const char* nslot, bool real, std::integer_sequence<int, I...>) {
  (dw_slot88<CUR, Y, I>(q, acc, nslot, real), ...);
}

// one segment of an 8x8 job on the slotted stage; prologue, ring and epilogue as in dw_segment_bf<8, 8, 4, 4, 6>
AF_DEV void dw_segment_88(const DwJob& jb, const DwSeg& sg, float* partial, char* smem, int tid, int wave, int lane, int a0, int b0, bool store_db) {
  constexpr int NI = 8, SLOT = NI * 4096, A_B = 8 * 2048;
  static_assert(DW_STAGES == 4, "the slotted stage is written for the four-slot ring");
  const int m = lane & 31, h = lane >> 5;
  f32x16 acc[4][4];
  Dw88 q;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    q.db[x] = 0.f;
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  }
  // piece k (0..7) of a stage: lane tid moves 16 B of feature row f = 64 (k & 3) + (tid >> 2), A half for k < 4, B half behind it;
  // logical row group c = (tid & 3) ^ ((f >> 2) & 3) sits in physical 16-B slot tid & 3 (the bank swizzle applied on the SOURCE)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int f = 64 * k + (tid >> 2), c = (tid & 3) ^ ((f >> 2) & 3);
    q.voff[0][k] = (uint32_t)(f * 128 + c * 16); q.voff[1][k] = q.voff[0][k] + 64u;
  }
  const int S = 2 * (sg.t1 - sg.t0);
  q.lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)smem + wave * 1024);
  q.sa4 = jb.a_stride * 4u; q.sb4 = jb.b_stride * 4u;
  // prologue: stages 0, 1, 2 = (tile 0, half 0), (tile 0, half 1), (tile min(1, last), half 0)
  const int ntile = sg.t1 - sg.t0;
  q.ga = (const char*)jb.A + (size_t)sg.t0 * q.sa4; q.gb = (const char*)jb.B + (size_t)sg.t0 * q.sb4;
  auto whole_stage = [&](int half, uint32_t ring) {
    const uint32_t (&vo)[4] = q.voff[half];
    dw_glds_s<0>(vo[0], q.ga, q.lds0 + ring); dw_glds_s<4096>(vo[1], q.ga, q.lds0 + ring); dw_glds_s<8192>(vo[2], q.ga, q.lds0 + ring); dw_glds_s<12288>(vo[3], q.ga, q.lds0 + ring);
    dw_glds_s<16384>(vo[0], q.gb, q.lds0 + ring); dw_glds_s<20480>(vo[1], q.gb, q.lds0 + ring); dw_glds_s<24576>(vo[2], q.gb, q.lds0 + ring); dw_glds_s<28672>(vo[3], q.gb, q.lds0 + ring);
  };
  whole_stage(0, 0u);
  whole_stage(1, 0x8000u);
  if (ntile > 1) { q.ga += q.sa4; q.gb += q.sb4; }     // now on tile min(1, last): what stage 2 and the loop's first staged stage (3) read
  whole_stage(0, 0x10000u);
  q.tiles_left = ntile - 2;                            // advances left after tile 1
  q.ring = 0x10000u;                                   // ring slot of the last stage staged (2); stage n goes to slot n & 3
  const int sw = (m >> 2) & 3;
  q.off0 = m * 64 + (((2 * h) ^ sw) << 4); q.off1 = m * 64 + (((2 * h + 1) ^ sw) << 4);     // rows 8h..8h+3, 8h+4..8h+7
  q.abase = a0 * 2048; q.bbase = A_B + b0 * 2048;
  dw_wait_vm<(DW_STAGES - 2) * NI>();                  // stage 0 has landed
  dw_barrier();
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    q.ran[x][0] = *(const f32x4*)(smem + q.abase + x * 2048 + q.off0); q.ran[x][1] = *(const f32x4*)(smem + q.abase + x * 2048 + q.off1);
    q.rb[x][0] = *(const f32x4*)(smem + q.bbase + x * 2048 + q.off0); q.rb[x][1] = *(const f32x4*)(smem + q.bbase + x * 2048 + q.off1);
  }
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    q.sa[0][x] = dw_split8<3>(q.ran[x][0], q.ran[x][1]);
    const f32x4 t = q.ran[x][0] + q.ran[x][1];
    q.db[x] += (t[0] + t[1]) + (t[2] + t[3]);
  }
  q.sb[0] = dw_split8<3>(q.rb[0][0], q.rb[0][1]);
  __builtin_amdgcn_sched_barrier(0);
  for (int s = 0; s < S; s += 2) {                      // S is even (two stages per 32-row tile)
    {
      const char* nslot = smem + ((s + 1) % DW_STAGES) * SLOT;
      const bool real = s + 1 < S;
      dw_col88<0, 0>(q, acc, nslot, real, std::make_integer_sequence<int, 24>{});
      dw_col88<0, 1>(q, acc, nslot, real, std::make_integer_sequence<int, 24>{});
      dw_col88<0, 2>(q, acc, nslot, real, std::make_integer_sequence<int, 24>{});
      dw_col88<0, 3>(q, acc, nslot, real, std::make_integer_sequence<int, 24>{});
    }
    {
      const char* nslot = smem + ((s + 2) % DW_STAGES) * SLOT;      // past the last stage: harmless reads of a re-staged slot
      const bool real = s + 2 < S;
      dw_col88<1, 0>(q, acc, nslot, real, std::make_integer_sequence<int, 24>{});
      dw_col88<1, 1>(q, acc, nslot, real, std::make_integer_sequence<int, 24>{});
      dw_col88<1, 2>(q, acc, nslot, real, std::make_integer_sequence<int, 24>{});
      dw_col88<1, 3>(q, acc, nslot, real, std::make_integer_sequence<int, 24>{});
    }
  }
  dw_wait_vm<0>();
  dw_barrier();

  float* blk = partial + jb.part_off + (size_t)sg.slot * jb.part_blk;
  constexpr int pld = 8 * 32;
  const auto rblk = af_rsrc_uniform(blk, jb.part_blk * 4);
  const int voff = ((a0 * 32 + 4 * h) * pld + b0 * 32 + m) * 4;
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dw_bs32(acc[x][y][r], rblk, voff, ((x * 32 + (r & 3) + 8 * (r >> 2)) * pld + y * 32) * 4);
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const float tot = q.db[x] + __shfl_xor(q.db[x], 32);
    if (store_db && h == 0) dw_bs32(tot, rblk, (8 * 32 * pld + a0 * 32 + m) * 4, x * 128);
  }
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
  // The source of a piece = a wave-uniform stage base (SGPR pair, made opaque so that hipcc does not re-associate it into the
  // per-lane part) + the lane's 32-bit offset: the `global_load_lds_dwordx4 voff, s[base]` form, no 64-bit VALU address per piece.
  auto sgpr = [](const char* p) {
    uint32_t lo = (uint32_t)(uint64_t)p, hi = (uint32_t)((uint64_t)p >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (const char*)(((uint64_t)hi << 32) | lo);
  };
  auto issue = [&](int s, int k) {
    const int sc = s < S ? s : S - 1;
    const size_t t = (size_t)(sg.t0 + (sc >> 1));
    const char* ga = sgpr((const char*)jb.A + t * jb.a_stride * 4u + (sc & 1) * 64);      // wave-uniform
    const char* gb = sgpr((const char*)jb.B + t * jb.b_stride * 4u + (sc & 1) * 64);
    const bool all_a = (k + 1) * 256 <= TO * 128, all_b = k * 256 >= TO * 128 && (k + 1) * 256 <= NP;
    const char* g = all_a ? ga : (all_b ? gb : (sel_a[k] ? ga : gb));
    dw_glds16(g + soff[k], smem + (s % DW_STAGES) * SLOT + k * 4096 + wave * 1024);
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
          dw_bs32(acc[x][y][r], rblk, voff, ((x * 32 + (r & 3) + 8 * (r >> 2)) * pld + y * 32) * 4);
  }
#pragma unroll
  for (int x = 0; x < TOW; ++x) {
    const float tot = dbacc[x] + __shfl_xor(dbacc[x], 32);
    if (store_db && h == 0) dw_bs32(tot, rblk, (TO * 32 * pld + a0 * 32 + m) * 4, x * 128);
  }
}

// Compacted batches (DwJob::live_rows, written by k_prep): row tiles behind the last live row hold what an earlier iteration
// left there.  Clip the segment to the live tiles; a segment with nothing left stores the zero block k_adam expects in its slot.
AF_DEV bool dw_clip(const DwJob& jb, DwSeg& sg, float* partial, int tid) {
  if (!jb.live_rows) return true;
  // (the count comes through a vector load: without the readfirstlane the segment's tile range — and with it every address of the
  // operand stream — is computed per lane, ~30 VALU instructions per stage that compete with the split for the MFMAs' shadows)
  const int nt = __builtin_amdgcn_readfirstlane((jb.live_base + *jb.live_rows + 31) >> 5);
  if (sg.t1 > nt) sg.t1 = nt;
  if (sg.t0 < sg.t1) return true;
  float* blk = partial + jb.part_off + (size_t)sg.slot * jb.part_blk;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (uint32_t i = (uint32_t)tid * 4u; i < jb.part_blk; i += 1024u) *(f32x4*)(blk + i) = z;
  return false;
}

// Segment and job descriptors are read after the previous segment's stores, so hipcc fetches them with VECTOR loads (the scalar
// cache is not coherent with them) and every address derived from them — the operand stream's source pointers above all — is then
// computed per lane.  They are wave-uniform by construction: pin them to SGPRs.
template <class T> AF_DEV T* dw_uniform_ptr(T* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));      // (the builtin returns int: no sign extension into the high word)
  return (T*)(((uint64_t)hi << 32) | lo);
}
AF_DEV DwSeg dw_uniform(const DwSeg& g) {
  return DwSeg{__builtin_amdgcn_readfirstlane(g.job), __builtin_amdgcn_readfirstlane(g.t0), __builtin_amdgcn_readfirstlane(g.t1), __builtin_amdgcn_readfirstlane(g.slot)};
}
AF_DEV DwJob dw_uniform(const DwJob& j) {
  DwJob u;
  u.A = dw_uniform_ptr(j.A); u.B = dw_uniform_ptr(j.B);
  u.a_stride = (uint32_t)__builtin_amdgcn_readfirstlane(j.a_stride); u.b_stride = (uint32_t)__builtin_amdgcn_readfirstlane(j.b_stride);
  u.shape = __builtin_amdgcn_readfirstlane(j.shape);
  u.part_off = (uint32_t)__builtin_amdgcn_readfirstlane(j.part_off); u.part_blk = (uint32_t)__builtin_amdgcn_readfirstlane(j.part_blk);
  u.live_base = __builtin_amdgcn_readfirstlane(j.live_base); u.live_rows = dw_uniform_ptr(j.live_rows);
  return u;
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
  if (a.wg_stamp && tid == 0 && blockIdx.x < AF_STAMP_WG) { a.wg_stamp[blockIdx.x * 4] = __builtin_amdgcn_s_memrealtime(); a.wg_stamp[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); }
  for (int s = 0; s < DW_MAXSEG; ++s) {
    DwSeg sg = dw_uniform(segs[s]);
    if (sg.job < 0) break;
    const DwJob jb = dw_uniform(a.jobs[sg.job]);
    if (!dw_clip(jb, sg, a.partial, tid)) continue;
    switch (jb.shape) {      // 8x8: each wave a 4x4 block of output tiles (8 operand tiles to read and split per stage, the minimum)
      case DW_8x8:
        if constexpr (DW_SLOT && NPROD == 6 && !DW_ABL) dw_segment_88(jb, sg, a.partial, smem, tid, wave, lane, 4 * (wave & 1), 4 * (wave >> 1), wave < 2);
        else dw_segment_bf<8, 8, 4, 4, NPROD>(jb, sg, a.partial, smem, tid, wave, lane, 4 * (wave & 1), 4 * (wave >> 1), true, wave < 2);
        break;
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
  if (a.wg_stamp && tid == 0 && blockIdx.x < AF_STAMP_WG) { a.wg_stamp[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime(); a.wg_stamp[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime(); }
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
