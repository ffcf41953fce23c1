// mlpbf.hip — the register-chained fused coordinate-MLP kernels of mlp.hip with their 256x256 hidden-layer products on
// the bf16 matrix pipe, fp32-faithful ("bf16x6", bfsplit.h): weights arrive pre-split into three bf16 images (hi, mid,
// lo — emitted by k_adam next to the fp32 views), activations are split in registers one k-step (16 features) at a
// time, and every product block accumulates the six leading partial products in the fp32 accumulators.  K = 16 retires in
// 6 x 32 cycles instead of the 8 x 64 of v_mfma_f32_32x32x2_f32: 2.7x the rate of the fp32 matrix pipe with fp32-level
// round-off (tests/test_split_precision.py).  Same tiles, same C-layout chaining, same launch structure as mlp.hip
// (src/models/stage_1/implicit_neural_networks.py:62-80 forward; the dX half of loss.backward(), stage1_neural_atlas.py:230).
//
// What stays on the fp32 pipe: the narrow blocks — layer 0 (K = 3 / 30 / 40), the positional-encoding columns of the skip
// layers, the 1..3-row output layer and, backward, the output layer and the atlas net's layer-0 (dPE) block.
//
// Weight stream: one contiguous image per net in consumption order; fp32 blocks as in mlp.hip, a hidden layer as eight
// 48 KB chunks of two k-steps: chunk-local index ((s*8 + T)*3 + level)*1024 + (h*32 + m)*16 holds the eight bf16
// W[32T + m][k(h, 0..7)] of k-step s, with k(h, i) = 32(s>>1) + 8(2(s&1) + i/4) + 4h + i%4 — exactly the eight features
// lane half h holds in registers 8s .. 8s+7 of the C-layout (one ds_read_b128 per tile, level and k-step).
// Two LDS slots of 48 KB; the barrier that publishes chunk c+1 sits 24 MFMAs before the end of chunk c (every wave then
// holds the rest of chunk c in registers), so the next chunk's first fragments and the DMA of chunk c+2 start in the
// shadow of chunk c's last product group.
#include "mlp_common.h"
#if AF_ABL & 32
#define DW_ABL 1      // timing ablation: no operand split
#endif
#include "bfsplit.h"

#define AF_SLOT_BF 49152
#define AF_BIAS_LDS_BF (2 * AF_SLOT_BF)
#define AF_LDS_BYTES_BF (2 * AF_SLOT_BF + AF_MAX_LAYERS * AF_HID * 4)

template <class NS> struct ChunkBytesBf {
  static constexpr int L0 = ChunkBytes<NS>::L0;
  static constexpr int HID = AF_SLOT_BF;                 // x8 per hidden layer
  static constexpr int SKIP = ChunkBytes<NS>::SKIP;
  static constexpr int LAST = ChunkBytes<NS>::LAST;
  static constexpr bool out_skip(int nl) { return ChunkBytes<NS>::out_skip(nl); }
  static constexpr int last_bytes(int nl) { return ChunkBytes<NS>::last_bytes(nl); }
  static constexpr int BLAST = ChunkBytes<NS>::BLAST;
  static constexpr int BL0H = 16 * 2 * 64 * 16;          // half of the backward layer-0 block (Mpad 64): two chunks of 32 KB
};
static_assert(ChunkBytesBf<NsAtlas>::L0 <= AF_SLOT_BF && ChunkBytesBf<NsAtlas>::SKIP <= AF_SLOT_BF && ChunkBytesBf<NsAlpha>::L0 <= AF_SLOT_BF, "fp32 blocks must fit a slot");

// Weight-chunk stream over two LDS slots.  Every stage copies a full slot (12 x 1 KB per wave) whatever the chunk's real
// size (the images are padded for the over-read); issue sites are branch-free (a site beyond the twelfth re-copies the
// last piece).  publish<B>() makes the chunk being staged (B bytes long) visible and starts staging the one behind it into
// the slot the previous chunk occupied — callers place it where every wave has issued its last read of that chunk.
struct BfStream {
  static constexpr int NI = AF_SLOT_BF / 4096;
  const char* src; char* smem; int wave, stg; char* p_dst; int p_it;
  AF_DEV void begin_stage() { p_dst = smem + (stg & 1) * AF_SLOT_BF + wave * 1024; p_it = 0; }
  AF_DEV void issue1() {
    const int k = p_it < NI ? p_it : NI - 1;
    if constexpr ((AF_ABL & 64) != 0) { if (wave != 0) { ++p_it; return; } }       // timing probe: only wave 0 feeds the texture addresser
    if constexpr (!(AF_ABL & 2)) af_glds16(src + k * 4096, p_dst + k * 4096);
    ++p_it;
  }
  // the five pieces a k-step issues right behind its own publish (group 3 of an odd k-step), for a stage opened elsewhere
  AF_DEV void lead5() { issue1(); issue1(); issue1(); issue1(); issue1(); }
  AF_DEV void start(const void* img, int tid, int wave_) { src = (const char*)img + tid * 16; wave = wave_; stg = 0; begin_stage(); }
  // KEEP: vector-memory instructions (tile stores) this wave has issued AFTER the last piece of the chunk being published — they
  // may stay in flight.  vmcnt counts loads and stores alike and retires them in issue order, so a counted wait covers every DMA
  // piece of the chunk without draining the youngest stores (a vmcnt(0) here waits for the write acknowledgement of stores
  // issued a few cycles earlier: ~10 % of a training chain).  Callers that cannot bound the count pass 0.
  template <int KEEP = 0>
  AF_DEV const char* publish(int BYTES) {
    if constexpr (KEEP == 0) { while (p_it < NI) issue1(); }      // (with stores behind the DMA the top-up would break the count: those callers issue all NI pieces themselves)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(KEEP) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* cur = smem + (stg & 1) * AF_SLOT_BF;
    src += BYTES; ++stg;
    begin_stage();
    return cur;
  }
};

struct BfPipe { f32x4 fl[8]; DwSplit b; };     // carried between k-steps: the lo-level fragments and the split B operand of the NEXT k-step

AF_DEV f32x4 bf_frag(const char* lane_base, int sl, int T, int lvl) {
  if constexpr (AF_ABL & 8) return f32x4{(float)sl, (float)T, (float)lvl, 1.f};
  return *(const f32x4*)(lane_base + ((sl * 8 + T) * 3 + lvl) * 1024);
}
// The six-product k-step reads its fragments straight into accumulator registers: the 96 registers of three fragment sets do not
// fit beside the 128 activation registers in the VGPR half, and a compiler-visible load lands in a VGPR first (measured: the
// slotted k-step then spills inside the split units).  The read is invisible to hipcc's waitcnt insertion - every consumer slot
// group starts with bf_lds_wait() (all fragment reads are issued at least four MFMAs before their first use).
template <int SL, int T, int LVL>
AF_DEV f32x4 bf_frag_a(uint32_t lane_addr) {
  f32x4 v;
  if constexpr (AF_ABL & 8) { v = f32x4{(float)SL, (float)T, (float)LVL, 1.f}; asm volatile("" : "+a"(v)); return v; }
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(v) : "v"(lane_addr), "n"(((SL * 8 + T) * 3 + LVL) * 1024));
  return v;
}
AF_DEV void bf_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// the same wait tied to the fragment the slot's first MFMA reads (round 6, found by isa_check.py rule (a) on mlphf.hip): an MFMA has no memory side and no
// other dependency on a bare wait, so inside its slot hipcc is free to schedule it in front of one
AF_DEV void bf_lds_wait(f32x4& frag) { asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(frag) :: "memory"); }
AF_DEV f32x16 bf_mfma(const f32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int LEVELS = 3>
AF_DEV DwSplit bf_split_in(const float (&in)[128], int s) {
  const f32x4 lo = {in[8 * s], in[8 * s + 1], in[8 * s + 2], in[8 * s + 3]}, hi = {in[8 * s + 4], in[8 * s + 5], in[8 * s + 6], in[8 * s + 7]};
  return dw_split8<LEVELS>(lo, hi);
}
template <int N> AF_DEV void bf_sgb() {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA, then at most one LDS read, two VALU, one VMEM behind it
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// One k-step (16 input features) of a 256 -> 256 product.  S: k-step of the layer (0..15), its chunk-local index is S & 1.
// ZI: the accumulators start at zero (backward chain): the very first MFMA of each takes a literal 0 as C.
// An odd k-step publishes the chunk behind its own before its last product group (after_bytes: the size of the chunk
// behind the whole block, used by k-step 15) and fetches the lo-level fragments of that chunk's first k-step — inside a
// block only: across layers nothing is carried (bf_enter), so that every layer of the runtime loop runs the same code.
// NPROD = 3 (backward chain only, af_set_mlp_mode(h, 2); an experiment, not the default): hi + mid operands, three products.
// NST: vector-memory instructions store_hook issues in a k-step S < 8 (16 tile stores, or 0 for a hook that stores nothing).
// DMA pieces of the chunk behind the current one: 5 right after its stage opens (group 3 of the odd k-step), 3 + 4 in groups
// 1 and 3 of the even k-step, none in the odd k-step's first two groups — all twelve are at least 24 MFMAs old at the publish,
// and only the odd k-step's own stores are younger (the publish's counted wait leaves exactly those in flight).
template <int S, bool ZI, int NPROD, int NST, class Hook>
AF_DEV void bf_kstep(f32x16 (&acc)[8], const float (&in)[128], BfPipe& pp, const char*& lane_base, BfStream& cs, int lane_off, int after_bytes, Hook&& store_hook) {
  constexpr bool NEXT_BF = S != 15;
  constexpr int sl = S & 1;
  constexpr bool last_in_chunk = sl == 1;
  constexpr bool X3 = NPROD == 3;
  static_assert(NPROD == 6 || (NPROD == 3 && ZI), "three products are wired for the backward chain only");
  f32x4 fm[8], fh[8];
  // ---- group 1: W_lo x B_hi (8 MFMAs); fetch W_mid
#pragma unroll
  for (int T = 0; T < 8; ++T) { if constexpr (!X3) pin_acc(pp.fl[T]); fm[T] = bf_frag(lane_base, sl, T, 1); }
  if constexpr (!X3) {
#pragma unroll
    for (int T = 0; T < 8; ++T) {
      if constexpr (ZI && S == 0) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[T] = bf_mfma(pp.fl[T], pp.b.h, z);
      } else {
        acc[T] = bf_mfma(pp.fl[T], pp.b.h, acc[T]);
      }
    }
  }
  if constexpr (!last_in_chunk) { cs.issue1(); cs.issue1(); cs.issue1(); }
  if constexpr (!X3) bf_sgb<8>();
  // ---- group 2: W_mid x (B_mid, B_hi) (16 MFMAs); fetch W_hi; the deferred tile stores of this k-step's feature tile
#pragma unroll
  for (int T = 0; T < 8; ++T) { pin_acc(fm[T]); fh[T] = bf_frag(lane_base, sl, T, 0); }
  if constexpr (!X3) {
#pragma unroll
    for (int T = 0; T < 8; ++T) acc[T] = bf_mfma(fm[T], pp.b.m, acc[T]);
  }
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    if constexpr (X3 && S == 0) {
      const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc[T] = bf_mfma(fm[T], pp.b.h, z);
    } else {
      acc[T] = bf_mfma(fm[T], pp.b.h, acc[T]);
    }
  }
  store_hook(GIdx<S>{});
  bf_sgb<X3 ? 8 : 16>();
  // ---- publish the next chunk (odd k-steps): every read of this chunk has been issued; the slot can be refilled
  const char* nxt_lane = lane_base;
  if constexpr (last_in_chunk) nxt_lane = cs.template publish<(S < 8 ? NST : 0)>(S == 15 ? after_bytes : AF_SLOT_BF) + lane_off;
  // ---- group 3: W_hi x (B_lo, B_mid, B_hi) (24 MFMAs); fetch the next k-step's W_lo and split its B operand
  DwSplit bn = pp.b;
  f32x4 fln[8];
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    pin_acc(fh[T]);
    if constexpr (X3)                  fln[T] = pp.fl[T];
    else if constexpr (!last_in_chunk) fln[T] = bf_frag(lane_base, 1, T, 2);
    else if constexpr (NEXT_BF)        fln[T] = bf_frag(nxt_lane, 0, T, 2);
    else                               fln[T] = pp.fl[T];
  }
  if constexpr (S + 1 < 16) bn = bf_split_in<X3 ? 2 : 3>(in, S + 1);
  if constexpr (!X3) {
#pragma unroll
    for (int T = 0; T < 8; ++T) acc[T] = bf_mfma(fh[T], pp.b.l, acc[T]);
  }
#pragma unroll
  for (int T = 0; T < 8; ++T) acc[T] = bf_mfma(fh[T], pp.b.m, acc[T]);
#pragma unroll
  for (int T = 0; T < 8; ++T) acc[T] = bf_mfma(fh[T], pp.b.h, acc[T]);
  cs.issue1(); cs.issue1(); cs.issue1(); cs.issue1();
  if constexpr (last_in_chunk) cs.issue1();
  bf_sgb<X3 ? 16 : 24>();
#pragma unroll
  for (int T = 0; T < 8; ++T) pp.fl[T] = fln[T];
  pp.b = bn;
  lane_base = nxt_lane;
}


// ---- the six-product k-step as 48 hand-placed issue slots --------------------------------------------------------------------
// One wave per SIMD issues in order: an MFMA holds the matrix pipe for 32 cycles, and whatever issues behind it within those
// cycles is free — up to ~5 light instructions (VALU, ds_read) or ONE vector-memory instruction (tools/issueprobe_gen.py: the
// whole k-step mix of 24 fragment reads, 44 split VALU, 6 DMA pieces and 16 stores costs 1572 ticks hand-placed, the same as 48
// bare MFMAs).  hipcc's scheduler does not produce such a stream from sched_group_barrier hints (stores and DMA pieces come out in
// clumps of 5-16, fragment reads land behind the LAST MFMAs of a group so that the next group's first use exposes the LDS latency),
// so the k-step is written slot by slot, every slot fenced by sched_barrier(0): slot I = MFMA I + the fillers listed below.
//   MFMAs    0-7 W_lo x B_hi | 8-15 W_mid x B_mid, 16-23 W_mid x B_hi | [publish, odd k-steps] | 24-31 W_hi x B_lo, 32-39 x B_mid, 40-47 x B_hi
//   reads    W_mid fragments two per slot in 0-3, W_hi in 8-11, the next k-step's W_lo in 24-27 (each a whole group ahead of its use)
//   DMA      even k-step: slots 4-6 and 32-35; odd: 28-32 (the five pieces that open the stage behind the publish)
//   stores   (k-steps 0-7 of a training chain, 16 per k-step) even: 12-23, 28-31; odd: 4-7, 12-23 — on an odd k-step all sixteen are
//            younger than the last DMA piece (even slot 35): exactly the KEEP the publish's counted vmcnt leaves in flight
//   split    of the next k-step's B operand in twelve units of 3-4 VALU: even 7, 36-46; odd 33-44
struct BfSplitState { f32x4 ra, rb; };       // residuals between the three units of a pair (vectors: constant-index lanes stay in registers)
template <int U, int S8>
AF_DEV void bf_split_unit(const float (&in)[128], DwSplit& bn, BfSplitState& st) {
  constexpr int i = U / 3, ph = U % 3;
  if constexpr (DW_ABL & 1) { if constexpr (ph == 0) { bn.h[i] = __builtin_bit_cast(uint32_t, in[S8 + 2 * i]); bn.m[i] = bn.h[i]; bn.l[i] = bn.h[i]; } return; }
  // the empty volatile asm at the end of a unit ties its results to this slot: pure VALU code has no place of its own in the
  // instruction selector's order and would otherwise be emitted next to its last use, i.e. all 44 in one clump behind slot 47
  if constexpr (ph == 0) {
    const float a = in[S8 + 2 * i], b = in[S8 + 2 * i + 1];
    uint32_t h = dw_pk(a, b);
    float ra = dw_sub(a, __builtin_bit_cast(float, h << 16)), rb = dw_sub(b, __builtin_bit_cast(float, h & 0xffff0000u));
    asm volatile("" : "+v"(h), "+v"(ra), "+v"(rb));
    bn.h[i] = h; st.ra[i] = ra; st.rb[i] = rb;
  } else if constexpr (ph == 1) {
    uint32_t m = dw_pk(st.ra[i], st.rb[i]);
    float ra = dw_sub(st.ra[i], __builtin_bit_cast(float, m << 16)), rb = dw_sub(st.rb[i], __builtin_bit_cast(float, m & 0xffff0000u));
    asm volatile("" : "+v"(m), "+v"(ra), "+v"(rb));
    bn.m[i] = m; st.ra[i] = ra; st.rb[i] = rb;
  } else {
    uint32_t l = dw_pk(st.ra[i], st.rb[i]);
    asm volatile("" : "+v"(l));
    bn.l[i] = l;
  }
}
template <int S, bool ZI, int NST, int I>
AF_DEV void bf_slot(f32x16 (&acc)[8], const float (&in)[128], BfPipe& pp, f32x4 (&fm)[8], f32x4 (&fh)[8], f32x4 (&fln)[8], DwSplit& bn, BfSplitState& st,
                    uint32_t la, uint32_t nla, BfStream& cs, const TileStore& ts) {
  constexpr int sl = S & 1, T = I & 7;
  constexpr bool odd = sl == 1, NEXT_BF = S != 15, stores = NST > 0 && S < 8;
  if constexpr (I == 0) bf_lds_wait(pp.fl[0]);                       // the fragments of this slot group have landed (read >= 4 MFMAs ago)
  if constexpr (I == 8) bf_lds_wait(fm[0]);
  if constexpr (I == 24) bf_lds_wait(fh[0]);
  // ---- the MFMA
  if constexpr (I < 8) {
    pin_acc(pp.fl[T]);
    if constexpr (ZI && S == 0) {
      const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc[T] = bf_mfma(pp.fl[T], pp.b.h, z);
    } else {
      acc[T] = bf_mfma(pp.fl[T], pp.b.h, acc[T]);
    }
  } else if constexpr (I < 16) { pin_acc(fm[T]); acc[T] = bf_mfma(fm[T], pp.b.m, acc[T]); }
  else if constexpr (I < 24) acc[T] = bf_mfma(fm[T], pp.b.h, acc[T]);
  else if constexpr (I < 32) { pin_acc(fh[T]); acc[T] = bf_mfma(fh[T], pp.b.l, acc[T]); }
  else if constexpr (I < 40) acc[T] = bf_mfma(fh[T], pp.b.m, acc[T]);
  else acc[T] = bf_mfma(fh[T], pp.b.h, acc[T]);
  // ---- fragment reads, a whole group ahead of their first use
  if constexpr (I < 4) { fm[2 * I] = bf_frag_a<sl, 2 * I, 1>(la); fm[2 * I + 1] = bf_frag_a<sl, 2 * I + 1, 1>(la); }
  if constexpr (I >= 8 && I < 12) { fh[2 * (I - 8)] = bf_frag_a<sl, 2 * (I - 8), 0>(la); fh[2 * (I - 8) + 1] = bf_frag_a<sl, 2 * (I - 8) + 1, 0>(la); }
  if constexpr (I >= 24 && I < 28) {
    constexpr int t0 = 2 * (I - 24);
    if constexpr (!odd)         { fln[t0] = bf_frag_a<1, t0, 2>(la); fln[t0 + 1] = bf_frag_a<1, t0 + 1, 2>(la); }
    else if constexpr (NEXT_BF) { fln[t0] = bf_frag_a<0, t0, 2>(nla); fln[t0 + 1] = bf_frag_a<0, t0 + 1, 2>(nla); }
    else                        { fln[t0] = pp.fl[t0]; fln[t0 + 1] = pp.fl[t0 + 1]; }
  }
  // ---- LDS-DMA pieces of the chunk behind the published one
  if constexpr ((!odd && ((I >= 4 && I < 7) || (I >= 32 && I < 36))) || (odd && I >= 28 && I < 33)) cs.issue1();
  // ---- tile stores of feature tile S (k-steps 0..7)
  if constexpr (stores) {
    constexpr int rr = !odd ? (I >= 12 && I < 24 ? I - 12 : (I >= 28 && I < 32 ? 12 + I - 28 : -1))
                            : (I >= 4 && I < 8 ? I - 4 : (I >= 12 && I < 24 ? 4 + I - 12 : -1));
    if constexpr (rr >= 0) af_bs32(in[(S & 7) * 16 + rr], ts.r, ts.voff + (rr & 3) * 128, (32 * (S & 7) + 8 * (rr >> 2)) * 128);
  }
  // ---- split of the next k-step's B operand
  if constexpr (S + 1 < 16) {
    constexpr int u = !odd ? (I == 7 ? 0 : (I >= 36 && I < 47 ? 1 + I - 36 : -1)) : (I >= 33 && I < 45 ? I - 33 : -1);
    if constexpr (u >= 0) bf_split_unit<u, 8 * (S + 1)>(in, bn, st);
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int S, bool ZI, int NST, int... I0, int... I1>
AF_DEV void bf_kstep6(f32x16 (&acc)[8], const float (&in)[128], BfPipe& pp, const char*& lane_base, BfStream& cs, int lane_off, int after_bytes, const TileStore& ts,
                      std::integer_sequence<int, I0...>, std::integer_sequence<int, I1...>) {
  f32x4 fm[8], fh[8], fln[8];
  DwSplit bn = pp.b;
  BfSplitState st{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const char* nxt_lane = lane_base;
  __builtin_amdgcn_sched_barrier(0);
  const uint32_t la = (uint32_t)(size_t)lane_base;
  (bf_slot<S, ZI, NST, I0>(acc, in, pp, fm, fh, fln, bn, st, la, la, cs, ts), ...);                              // slots 0..23
  if constexpr ((S & 1) == 1) nxt_lane = cs.template publish<(S < 8 ? NST : 0)>(S == 15 ? after_bytes : AF_SLOT_BF) + lane_off;
  const uint32_t nla = (uint32_t)(size_t)nxt_lane;
  (bf_slot<S, ZI, NST, 24 + I1>(acc, in, pp, fm, fh, fln, bn, st, la, nla, cs, ts), ...);                        // slots 24..47
  bf_lds_wait();          // the next k-step's W_lo fragments (read in slots 24..27, ~20 MFMAs ago) before anything - a register copy included - touches them
#pragma unroll
  for (int T = 0; T < 8; ++T) pp.fl[T] = fln[T];
  pp.b = bn;
  lane_base = nxt_lane;
}

// A whole 256 -> 256 product (16 k-steps = 8 chunks).  On entry the first chunk is published at lane_base (bf_enter has
// fetched its first fragments); on exit lane_base addresses the published chunk behind the block (after_bytes long).
template <bool ZI, int NPROD, int NST, int... Ss>
AF_DEV void bf_block_impl(f32x16 (&acc)[8], const float (&in)[128], BfPipe& pp, const char*& lane_base, BfStream& cs, int lane_off, int after_bytes, const TileStore& ts, std::integer_sequence<int, Ss...>) {
  if constexpr (NPROD == 6) {
    (bf_kstep6<Ss, ZI, NST>(acc, in, pp, lane_base, cs, lane_off, after_bytes, ts, std::make_integer_sequence<int, 24>{}, std::make_integer_sequence<int, 24>{}), ...);
  } else {
    auto hook = [&](auto gi) { if constexpr (NST > 0) ts.template part<decltype(gi)::value>(in); };
    (bf_kstep<Ss, ZI, NPROD, NST>(acc, in, pp, lane_base, cs, lane_off, after_bytes, hook), ...);
  }
}
// NST: 16 when the chain stores its tiles (training forward, backward), 0 otherwise (ts is then unused)
template <bool ZI, int NPROD, int NST>
AF_DEV void bf_block(f32x16 (&acc)[8], const float (&in)[128], BfPipe& pp, const char*& lane_base, BfStream& cs, int lane_off, int after_bytes, const TileStore& ts) {
  bf_block_impl<ZI, NPROD, NST>(acc, in, pp, lane_base, cs, lane_off, after_bytes, ts, std::make_integer_sequence<int, 16>{});
}
// entering a block: the lo-level fragments and the split B operand of its first k-step
AF_DEV void bf_enter(BfPipe& pp, const float (&in)[128], const char* lane_base) {
#pragma unroll
  for (int T = 0; T < 8; ++T) pp.fl[T] = bf_frag(lane_base, 0, T, 2);     // compiler-visible reads: both k-step flavours may follow
  pp.b = bf_split_in(in, 0);
}

// HID: the net has hidden 256 -> 256 layers (nl >= 3).  A two-layer net (layer 0 + output layer) takes the copy without the layer
// loop: with a possibly-empty loop the compiler keeps the pre-loop activations alive ACROSS it for the zero-trip exit (86 registers
// spilled to scratch per chain, +12 % HBM writes of the training forward, profiles/r3: found by the byte-model guard of bench.py).
template <class NS, bool TRAIN, bool HID>
AF_DEV void mlp_fwd_body_bf(const FwdArgs& a, int wg, char* smem) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, h = lane >> 5;
  int tile = a.tile0 + wg * 4 + wave;
  const int NT = live_tiles(a);
  if (a.tile0 + wg * 4 >= NT) return;                  // a workgroup of rows that do not exist this iteration (uniform: before any barrier)
  const bool live = tile < NT;
  if (!live) tile = NT - 1;
  const int row = tile * 32 + j;

  using CB = ChunkBytesBf<NS>;
  BfStream cs; cs.smem = smem;
  cs.start(a.wimg, tid, wave);
  const int nl = a.nl;
  stage_bias(nl, a.bias, smem + AF_BIAS_LDS_BF, tid);
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
        if constexpr ((AF_ABL & 256) != 0) { pe[g * 4 + 0] = p0; pe[g * 4 + 1] = p1; pe[g * 4 + 2] = -p0; pe[g * 4 + 3] = -p1; }      // timing probe: no sin / cos
        else if constexpr ((AF_ABL & 512) != 0) { sincosf(p0, &pe[g * 4 + 0], &pe[g * 4 + 2]); sincosf(p1, &pe[g * 4 + 1], &pe[g * 4 + 3]); }
        else { pe[g * 4 + 0] = sinf(p0); pe[g * 4 + 1] = sinf(p1); pe[g * 4 + 2] = cosf(p0); pe[g * 4 + 3] = cosf(p1); }
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

  const int a_off8 = (h * 256 + j) * 16;     // lane offset inside an fp32 Mpad = 256 image chunk
  const int lane_off = (h * 32 + j) * 16;    // lane offset inside a bf16 chunk
  const int voff_t = (4 * h * 32 + j) * 4;
  const char* bias_lds = smem + AF_BIAS_LDS_BF;
  f32x16 acc[8];
  float in[128];
  TileStore ts{af_rsrc(a.acts, 0), voff_t};
  BfPipe pp;

  auto relu_out = [&](int l) {               // acc -> in[] = relu(Z_l) = X_{l+1}; its stores are deferred
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
  auto hook_dma = [&](auto gi) { if constexpr (decltype(gi)::value < 6) { cs.issue1(); cs.issue1(); } };

  // ---- layer 0 (fp32 block); the chunk behind it is the first bf16 chunk of layer 1
  const char* cur = cs.publish(CB::L0);             // its barrier also publishes the bias rows
  init_bias(acc, bias_lds, 0, h);
  mm_block<8, NS::K0G, 0, 4>(acc, pe, cur + a_off8, hook_dma);
  relu_out(0);
  const char* lane_base = cs.publish(nl > 2 ? CB::HID : CB::last_bytes(nl)) + lane_off;     // what lies behind layer 0: the first hidden chunk, or the output layer
  cs.lead5();

  // ---- hidden layers 1 .. NL-2: eight bf16 chunks each (+ the fp32 block of the skip columns); every iteration runs the
  // same code — what lies behind a layer (its skip block, the next layer's first chunk, the output layer) is only a size
  if constexpr (HID) {
  int l = 1;
#pragma unroll 1
  do {
    init_bias(acc, bias_lds, l, h);
    bf_enter(pp, in, lane_base);
    const bool skip = NS::SKIP != 0 && ((NS::SKIP >> l) & 1);
    const int behind = l == nl - 2 ? CB::last_bytes(nl) : CB::HID;
    bf_block<false, 6, ((TRAIN && !(AF_ABL & 1)) ? 16 : 0)>(acc, in, pp, lane_base, cs, lane_off, skip ? CB::SKIP : behind, ts);
    if constexpr (NS::SKIP != 0) {
      if (skip) {
        mm_block<8, NS::PEG, 0, 4>(acc, pe, lane_base - lane_off + a_off8, hook_dma);
        lane_base = cs.publish(behind) + lane_off;
        cs.lead5();
      }
    }
    relu_out(l);
  } while (++l <= nl - 2);
  }
  lane_base -= lane_off;

  // ---- output layer (1..3 real outputs), tanh, on 4x4x1 fp32 MFMA blocks (see mlp.hip); lane_base = the chunk's LDS base
  {
    const char* buf = lane_base;
    if constexpr (TRAIN) {     // the last hidden layer's activation tile
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
    const f32x4 bias = *(const f32x4*)(bias_lds + (nl - 1) * AF_HID * 4);
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

// x if bit (31 - e) of the sign-mask word is set, else 0 — as ONE opaque bit-field extract + and per element, tied to the
// accumulator read: written with the C-level sbfe builtin hipcc turns the 128 tests into and / compare / select chains
// and computes all 128 bit masks up front, 128 live registers that spill the chain.
template <int E> AF_DEV float bf_mask_keep(float x, uint32_t mk) {
  uint32_t r;
  asm("v_bfe_i32 %0, %1, %2, 1\n\tv_and_b32 %0, %0, %3" : "=&v"(r) : "v"(mk), "n"(31 - ((E >> 4) & 1) * 16 - (E & 15)), "v"(x));
  return __builtin_bit_cast(float, r);
}
template <int... Es> AF_DEV void bf_mask_all(float (&in)[128], const f32x16 (&acc)[8], const uint32_t (&mk)[4], std::integer_sequence<int, Es...>) {
  ((in[Es] = bf_mask_keep<Es>(acc[Es >> 4][Es & 15], mk[Es >> 5])), ...);
}

// ------------------------------------------------------------------------------------------------
// Backward dX chain on the same scheme: dZ_{l-1} = (W_l^T dZ_l) . relu'(Z_{l-1}) with the bf16x3 image of W_l^T.
template <class NS, int NPROD = 6>
AF_DEV void mlp_bwd_body_bf(const BwdArgs& a, int wg, char* smem) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, h = lane >> 5;
  int tile = a.tile0 + wg * 4 + wave;
  const int NT = live_tiles(a);
  if (a.tile0 + wg * 4 >= NT) return;                  // a workgroup of rows that do not exist this iteration (uniform: before any barrier)
  const bool live = tile < NT;
  if (!live) tile = NT - 1;
  const int row = tile * 32 + j;

  using CB = ChunkBytesBf<NS>;
  BfStream cs; cs.smem = smem;
  cs.start(a.wimg, tid, wave);
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
  const int lane_off = (h * 32 + j) * 16;
  const int voff_t = (4 * h * 32 + j) * 4;
  f32x16 acc[8];
  float in[128];
  TileStore ts{af_rsrc(a.dz, 0), voff_t};
  BfPipe pp;

  auto mask_out = [&](int l) {      // acc = dX_l; mask with sign bits of X_l (masks[l-1]) -> in[] = dZ_{l-1}
    const u32x4 m4 = *(const u32x4*)(a.masks + (((size_t)(l - 1) * a.nt_stride + tile) * 64 + lane) * 4);
    const uint32_t mk[4] = {m4[0], m4[1], m4[2], m4[3]};
    bf_mask_all(in, acc, mk, std::make_integer_sequence<int, 128>{});
    ts.r = af_rsrc_uniform(a.dz + ((size_t)(l - 1) * a.nt_stride + tile) * AF_TILE_F, live ? AF_TILE_F * 4 : 0);
    // the element-wise ops above are inline asm (af_relu / bf_mask_keep): gfx950 needs TWO wait states between a VALU write of a VGPR and an MFMA
    // reading it as SrcA / SrcB (tools/hazardprobe.hip); hipcc pads its own VALU ops and cannot see these.  Left to the scheduler they sink to just in
    // front of the output layer's MFMAs (a two-layer net has nothing else behind them): stale operands, a corrupted chain.  isa_check.py rule (d)
    // proves on every build that no such pair exists
    AF_ELEMWISE_FENCE();
  };
  auto hook_dma = [&](auto gi) { if constexpr (decltype(gi)::value < 6) { cs.issue1(); cs.issue1(); } };
  auto hook_dma_store = [&](auto gi) { if constexpr (decltype(gi)::value < 6) { cs.issue1(); cs.issue1(); } ts.template part<decltype(gi)::value>(in); };

  // ---- output layer (fp32 block): K = 8 (one group), only p < OUT non-zero
  const char* cur = cs.publish(CB::BLAST);
  mm_block<8, 1, 0, NS::OUT, true>(acc, dzl, cur + a_off8, hook_dma);
  mask_out(nl - 1);
  const char* lane_base = cs.publish(nl > 2 ? CB::HID : (NS::DX0 ? CB::BL0H : 4096)) + lane_off;
  cs.lead5();

#pragma unroll 1
  for (int l = nl - 2; l >= 1; --l) {
    bf_enter(pp, in, lane_base);
    // behind the last hidden block: the atlas net's layer-0 block (first half), or nothing (a harmless stage of the padding)
    bf_block<true, NPROD, ((AF_ABL & 1) ? 0 : 16)>(acc, in, pp, lane_base, cs, lane_off, l > 1 ? CB::HID : (NS::DX0 ? CB::BL0H : 4096), ts);
    mask_out(l);
  }
  lane_base -= lane_off;

  if constexpr (NS::DX0) {
    // dPE = W_0^T dZ_0  (M = 64 padded PE features, K = 256) in two 16-group halves, then through sin/cos to the 2-D input
    static_assert(NS::IN == AF_IN_PE2, "input gradient is only needed for the atlas net");
    f32x16 acc2[2];
    mm_block<2, 16, 0, 4, true>(acc2, in, lane_base + (h * 64 + j) * 16, hook_dma_store);
    const char* half2 = cs.publish(CB::BL0H);
    mm_block<2, 16, 64, 4>(acc2, in, half2 + (h * 64 + j) * 16, hook_dma);
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
#ifdef AF_CLK      // tools/ablate.hip: core clock ticks (s_memtime) each workgroup spends, to read the shader clock under this load
__device__ unsigned long long* g_af_clk;
#define AF_CLK_MARK(i) if (threadIdx.x == 0 && g_af_clk) g_af_clk[blockIdx.x * 2 + (i)] = __builtin_amdgcn_s_memtime()
#else
#define AF_CLK_MARK(i)
#endif

template <bool TRAIN>
__global__ __launch_bounds__(256, 1) void k_mlp_fwd_multi_bf(MultiFwd m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AF_CLK_MARK(0);
  AF_STAMP(m, 0);
  int s = 0, base = 0;
  const int wg = blockIdx.x;
  while (s + 1 < m.n && wg >= m.wg_end[s]) { base = m.wg_end[s]; ++s; }
  switch (m.net[s]) {
    case AF_NET_MAP1:  if (m.a[s].nl > 2) mlp_fwd_body_bf<NsMap1, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body_bf<NsMap1, TRAIN, false>(m.a[s], wg - base, smem); break;
    case AF_NET_MAP2:  if (m.a[s].nl > 2) mlp_fwd_body_bf<NsMap2, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body_bf<NsMap2, TRAIN, false>(m.a[s], wg - base, smem); break;
    case AF_NET_ATLAS: if (m.a[s].nl > 2) mlp_fwd_body_bf<NsAtlas, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body_bf<NsAtlas, TRAIN, false>(m.a[s], wg - base, smem); break;
    case AF_KIND_MAP_PE: if (m.a[s].nl > 2) mlp_fwd_body_bf<NsMapPe, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body_bf<NsMapPe, TRAIN, false>(m.a[s], wg - base, smem); break;
    default:           if (m.a[s].nl > 2) mlp_fwd_body_bf<NsAlpha, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body_bf<NsAlpha, TRAIN, false>(m.a[s], wg - base, smem); break;
  }
  AF_CLK_MARK(1);
  AF_STAMP(m, 1);
}

__global__ __launch_bounds__(256, 1) void k_mlp_bwd_multi_bf(MultiBwd m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AF_CLK_MARK(0);
  AF_STAMP(m, 0);
  int s = 0, base = 0;
  const int wg = blockIdx.x;
  while (s + 1 < m.n && wg >= m.wg_end[s]) { base = m.wg_end[s]; ++s; }
  switch (m.net[s]) {
    case AF_NET_MAP1:  mlp_bwd_body_bf<NsMap1>(m.a[s], wg - base, smem); break;
    case AF_NET_MAP2:  mlp_bwd_body_bf<NsMap2>(m.a[s], wg - base, smem); break;
    case AF_NET_ATLAS: mlp_bwd_body_bf<NsAtlas>(m.a[s], wg - base, smem); break;
    case AF_KIND_MAP_PE: mlp_bwd_body_bf<NsMapPe>(m.a[s], wg - base, smem); break;
    default:           mlp_bwd_body_bf<NsAlpha>(m.a[s], wg - base, smem); break;
  }
  AF_CLK_MARK(1);
  AF_STAMP(m, 1);
}

// the same chain on three products (see bf_kstep)
__global__ __launch_bounds__(256, 1) void k_mlp_bwd_multi_bf3(MultiBwd m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AF_STAMP(m, 0);
  int s = 0, base = 0;
  const int wg = blockIdx.x;
  while (s + 1 < m.n && wg >= m.wg_end[s]) { base = m.wg_end[s]; ++s; }
  switch (m.net[s]) {
    case AF_NET_MAP1:  mlp_bwd_body_bf<NsMap1, 3>(m.a[s], wg - base, smem); break;
    case AF_NET_MAP2:  mlp_bwd_body_bf<NsMap2, 3>(m.a[s], wg - base, smem); break;
    case AF_NET_ATLAS: mlp_bwd_body_bf<NsAtlas, 3>(m.a[s], wg - base, smem); break;
    case AF_KIND_MAP_PE: mlp_bwd_body_bf<NsMapPe, 3>(m.a[s], wg - base, smem); break;
    default:           mlp_bwd_body_bf<NsAlpha, 3>(m.a[s], wg - base, smem); break;
  }
  AF_STAMP(m, 1);
}

extern "C" int af_launch_fwd_multi_bf(MultiFwd* m, int train, hipStream_t s) {
  int tot = 0;
  for (int i = 0; i < m->n; ++i) { tot += (m->a[i].NT - m->a[i].tile0 + 3) / 4; m->wg_end[i] = tot; }
  if (tot <= 0) return 0;
  if (train) hipLaunchKernelGGL((k_mlp_fwd_multi_bf<true>), dim3(tot), dim3(256), AF_LDS_BYTES_BF, s, *m);
  else       hipLaunchKernelGGL((k_mlp_fwd_multi_bf<false>), dim3(tot), dim3(256), AF_LDS_BYTES_BF, s, *m);
  return (int)hipGetLastError();
}
extern "C" int af_launch_bwd_multi_bf(MultiBwd* m, hipStream_t s) {
  int tot = 0;
  for (int i = 0; i < m->n; ++i) { tot += (m->a[i].NT - m->a[i].tile0 + 3) / 4; m->wg_end[i] = tot; }
  if (tot <= 0) return 0;
  if (m->n > 0 && m->nprod == 3) hipLaunchKernelGGL(k_mlp_bwd_multi_bf3, dim3(tot), dim3(256), AF_LDS_BYTES_BF, s, *m);
  else                           hipLaunchKernelGGL(k_mlp_bwd_multi_bf, dim3(tot), dim3(256), AF_LDS_BYTES_BF, s, *m);
  return (int)hipGetLastError();
}
// chunk sizes of the bf16 streams for the host planner: which = 0 fwd layer 0, 1 bf16 hidden chunk (x8 per layer), 2 skip columns,
// 3 fwd output layer, 4 bwd output layer, 5 half of bwd layer 0 (x2)
extern "C" int af_mlp_chunk_bytes_bf(int net, int which, int nl) {     // nl: layers of the net (the output-layer chunk is longer when it carries skip columns)
  auto pick = [&](auto ns) -> int {
    using CB = ChunkBytesBf<decltype(ns)>;
    const int v[6] = {CB::L0, CB::HID, CB::SKIP, CB::last_bytes(nl), CB::BLAST, CB::BL0H};
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
extern "C" int af_mlp_bf_init() {
  hipError_t e = hipSuccess;
#define AF_ATTR(K) do { hipError_t r = hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, AF_LDS_BYTES_BF); if (r != hipSuccess) e = r; } while (0)
  AF_ATTR((k_mlp_fwd_multi_bf<true>)); AF_ATTR((k_mlp_fwd_multi_bf<false>)); AF_ATTR(k_mlp_bwd_multi_bf); AF_ATTR(k_mlp_bwd_multi_bf3);
#undef AF_ATTR
  return (int)e;
}
