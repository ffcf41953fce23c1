// mlphf.hip — the register-chained fused coordinate-MLP kernels of mlpbf.hip with their 256x256 hidden-layer products on the
// fp16 matrix pipe in THREE products per fp32-faithful product ("f16x3", round 6; af_set_mlp_mode(h, 3)).
// (src/models/stage_1/implicit_neural_networks.py:62-80 forward; the dX half of loss.backward(), stage1_neural_atlas.py:229-230.)
//
// Arithmetic.  An fp32 operand x is split into two fp16 terms of its SCALED value: h = f16(x s), l = f16(x s - h) (round to
// nearest even, the residual exact in fp32), s a power of two; |x s - h - l| <= 2^-24 |x s| while l is a normal fp16, and
// <= 2^-25 absolute (the subnormal spacing) below that.  A product a b accumulates hh + hl + lh in the MFMA's fp32 accumulator;
// the dropped l l is <= 2^-22 |ab| in the worst case and 2^-26 typically.  What makes the split fp32-grade on EVERY row is where
// the scale comes from (profiles/r5_split_error_real_tensors.txt: one scale per tensor fails on the mapping net's dZ, whose rows
// span 2^20):
//   * activations / gradients: one scale per ROW, per layer — a lane owns a row in the C-layout, so the row's largest magnitude
//     is a 64-instruction max over the lane's registers and one exchange with the other lane half; s = 2^(15 - exponent) puts
//     it into [2^14, 2^15): every element within 2^17 of its row's maximum keeps 22 bits, smaller ones lose bits only below
//     2^-39 of that maximum.  v_mfma_f32_32x32x16_f16 honours subnormal inputs (tools/f16probe.hip, profiles/r6_f16probe.txt).
//   * weights: a FIXED 2^12 (AF_HF_WSHIFT) — |W| < 16 stays finite, a weight keeps full precision down to 2^-16 and 2^-37
//     absolute below; k_adam emits the two fp16 images next to the other views and raises the handle's range flag at |w| >= 8.
// The accumulator then holds 2^12 s (W x): the forward epilogue removes the scale and adds the bias in ONE fma (the accumulators
// start at zero in both directions), the backward epilogue in one multiply; both are exact (powers of two).
// Measured on the MI355X from this pipe (tools/f16probe.hip): rms / worst error of K = 256 dot products against fp64, in units
// of 2^-24 sum|a b|: 0.36-0.90 / 2.7-8.9 against 0.54-1.32 / 4.8-13.7 for v_mfma_f32_32x32x2_f32 (an fp32 fmaf chain) on the
// same data, rows spanning 2^24 included.
//
// Cost: 3 MFMAs of 32 cycles per K = 16 instead of bf16x6's 6; the split is 4 VALU per value pair (v_fma_mixlo/mixhi_f16:
// scale, round and pack in one instruction each; bf16x6: 11).  What does not shrink: the per-layer epilogue.
//
// What stays on the fp32 pipe: as in mlpbf.hip — layer 0, the PE columns of the skip layers (their B operand pre-multiplied by the
// row's scale so that they land in the same accumulators), the output layers, the atlas net's backward layer-0 block.
//
// Weight stream: one contiguous image per net in consumption order; fp32 blocks as in mlp.hip, a hidden layer as four 64 KB
// chunks of four k-steps: chunk-local byte ((sl*8 + T)*2 + level)*1024 + (h*32 + m)*16 holds the eight fp16
// 2^12 W[32T + m][k(h, 0..7)] of k-step sl (level 0 = h, 1 = l), k as in mlpbf.hip.  Two LDS slots of 64 KB + the bias rows.
#include "mlp_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define AF_SLOT_HF 65536
#define AF_BIAS_LDS_HF (2 * AF_SLOT_HF)
#define AF_LDS_BYTES_HF (2 * AF_SLOT_HF + AF_MAX_LAYERS * AF_HID * 4)
#define AF_HF_WSHIFT 12                 // the weight images hold 2^12 W (elem.hip emit_hf2 uses the same constant)
#define AF_HF_KEEP 8                    // tile stores younger than the last DMA piece of a chunk at its publish (see hf_slot)

#ifdef AF_HF_CLK      // tools/hfbench.hip: s_memtime at the seams of a chain (wave 0 of every workgroup): where the ticks of a task go
__device__ unsigned long long* g_hf_clk;
#define HF_MARK(i) do { if (g_hf_clk && threadIdx.x == 0) g_hf_clk[(size_t)blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HF_MARK(i) do { } while (0)
#endif

template <class NS> struct ChunkBytesHf {
  static constexpr int L0 = ChunkBytes<NS>::L0;
  static constexpr int HID = AF_SLOT_HF;                 // x4 per hidden layer
  static constexpr int SKIP = ChunkBytes<NS>::SKIP;
  static constexpr int LAST = ChunkBytes<NS>::LAST;
  static constexpr bool out_skip(int nl) { return ChunkBytes<NS>::out_skip(nl); }
  static constexpr int last_bytes(int nl) { return ChunkBytes<NS>::last_bytes(nl); }
  static constexpr int BLAST = ChunkBytes<NS>::BLAST;
  static constexpr int BL0H = 16 * 2 * 64 * 16;          // half of the backward layer-0 block (Mpad 64): two chunks of 32 KB
};
static_assert(ChunkBytesHf<NsAtlas>::L0 <= AF_SLOT_HF && ChunkBytesHf<NsAtlas>::SKIP <= AF_SLOT_HF && ChunkBytesHf<NsAlpha>::L0 <= AF_SLOT_HF, "fp32 blocks must fit a slot");

// Weight-chunk stream over two LDS slots (BfStream of mlpbf.hip with 64 KB slots: 16 x 1 KB per wave and stage).
struct HfStream {
  static constexpr int NI = AF_SLOT_HF / 4096;
  const char* src; char* smem; int wave, stg; char* p_dst; int p_it;
  AF_DEV void begin_stage() { p_dst = smem + (stg & 1) * AF_SLOT_HF + wave * 1024; p_it = 0; }
  AF_DEV void issue1() {
    const int k = p_it < NI ? p_it : NI - 1;
    if constexpr (!(AF_ABL & 2)) af_glds16(src + k * 4096, p_dst + k * 4096);
    ++p_it;
  }
  // the five pieces a k-step issues right behind its own publish, for a stage opened elsewhere
  AF_DEV void lead5() { issue1(); issue1(); issue1(); issue1(); issue1(); }
  // a whole stage at once: the head of a chain has no MFMA shadow to spread the pieces over, and a chunk issued early has landed by its publish
  // (tools/hfbench.hip: the first two publishes of a task exposed ~2 k ticks of DMA latency each).  BYTES: what the chunk really holds.
  AF_DEV void issue_bytes(int BYTES) { const int n = (BYTES + 4095) / 4096; while (p_it < n) issue1(); p_it = NI; }
  AF_DEV void start(const void* img, int tid, int wave_) { src = (const char*)img + tid * 16; wave = wave_; stg = 0; begin_stage(); }
  // KEEP: tile stores this wave has issued AFTER the last piece of the chunk being published (vmcnt retires loads and stores in
  // issue order: the counted wait covers every DMA piece and leaves the youngest stores in flight).  Callers that cannot bound the
  // count pass 0 and have the stage topped up here.
  template <int KEEP = 0>
  AF_DEV const char* publish(int BYTES) {
    if constexpr (KEEP == 0) { while (p_it < NI) issue1(); }
    if constexpr (!(AF_ABL & 4)) {          // (timing probe: no wait, no barrier)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(KEEP) : "memory");
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("" ::: "memory");
    const char* cur = smem + (stg & 1) * AF_SLOT_HF;
    src += BYTES; ++stg;
    begin_stage();
    return cur;
  }
};

struct HfB { u32x4 h, l; };                       // the split B operand of one k-step: eight fp16 pairs (hi terms, lo terms)
struct HfPipe { f32x4 fl[8]; HfB b; };            // carried between k-steps: the lo-level fragments and the split B operand of the NEXT k-step

template <int SL, int T, int LVL>
AF_DEV f32x4 hf_frag_a(uint32_t lane_addr) {      // straight into an accumulator register (mlpbf.hip bf_frag_a)
  f32x4 v;
  if constexpr (AF_ABL & 8) { v = f32x4{(float)SL, (float)T, (float)LVL, 1.f}; asm volatile("" : "+a"(v)); return v; }
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(v) : "v"(lane_addr), "n"(((SL * 8 + T) * 2 + LVL) * 1024));
  return v;
}
AF_DEV void hf_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// The same wait TIED to the fragment the slot's MFMA reads: an MFMA has no memory side and no other dependency on the wait, so inside its slot hipcc
// may schedule it in front of a bare wait (it did once the chains' row-exponent stores shifted the schedule: isa_check.py rule (a) refused the build)
AF_DEV void hf_lds_wait(f32x4& frag) { asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(frag) :: "memory"); }
AF_DEV f32x16 hf_mfma(const f32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// One value pair: h = f16(x s) packed (lo, hi), l = f16(x s - h).  v_fma_mixlo/mixhi_f16 compute the fma in fp32 (x s is exact, a
// power of two; x s - h is exact, the residual of a round-to-nearest) and round ONCE to fp16 (RNE, subnormals kept), writing one
// half of the destination.  asm volatile: the unit stays in the issue slot it is written in.
AF_DEV void hf_split_pair(float x0, float x1, float s, uint32_t& h, uint32_t& l) {
  if constexpr (AF_ABL & 32) { h = __builtin_bit_cast(uint32_t, x0); l = __builtin_bit_cast(uint32_t, x1); return; }
  asm volatile("v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
               : "=&v"(h), "=&v"(l) : "v"(x0), "v"(x1), "v"(s));
}
AF_DEV HfB hf_split_in(const float (&in)[128], int s8, float sc) {
  HfB b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { uint32_t hh, ll; hf_split_pair(in[s8 + 2 * i], in[s8 + 2 * i + 1], sc, hh, ll); b.h[i] = hh; b.l[i] = ll; }
  return b;
}
// The row's scale from the lane's maximum: mx <- max over both lane halves, s = 2^(15 - exponent(mx)) (mx s in [2^14, 2^15)), and
// the factor that takes the accumulator back: inv = 1 / (s 2^12).  Exponents beyond +-60 are clamped (a row whose largest magnitude
// is below 2^-45 loses bits it does not have; above 2^75 the chain holds non-finite values anyway and k_adam's flag is up).
// Returns the exponent e of the scale (s = 2^e), or -128 for an all-zero row.
AF_DEV int hf_row_scale(float mx, float& sc, float& inv) {
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  int e = 15 - __builtin_amdgcn_frexp_expf(mx);
  e = e > 60 ? 60 : (e < -60 ? -60 : e);
  const bool zero = !(mx > 0.f);
  if (zero) e = 0;
  sc = __builtin_ldexpf(1.f, e);
  inv = __builtin_ldexpf(1.f, -e - AF_HF_WSHIFT);
  return zero ? -128 : e;
}
// shift the bit "v > 0" (the sign of 0 - v) into a mask word, as one opaque pair: written in C the 128 subtractions of a layer are SLP-packed into
// v_pk_add_f32 on register pairs and batched ahead of their use (7 spilled registers in the atlas net's training chain)
AF_DEV void hf_sign_in(uint32_t& mk, float v) { uint32_t t; asm("v_sub_f32 %1, 0, %2\n\tv_alignbit_b32 %0, %0, %1, 31" : "+v"(mk), "=&v"(t) : "v"(v)); }
AF_DEV void hf_max3(float& m, float a, float b) { asm("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b)); }
AF_DEV void hf_max3_abs(float& m, float a, float b) { asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b)); }

// x if bit (31 - e) of the sign-mask word is set, else 0 (mlpbf.hip bf_mask_keep)
template <int E> AF_DEV float hf_mask_keep(float x, uint32_t mk) {
  uint32_t r;
  asm("v_bfe_i32 %0, %1, %2, 1\n\tv_and_b32 %0, %0, %3" : "=&v"(r) : "v"(mk), "n"(31 - ((E >> 4) & 1) * 16 - (E & 15)), "v"(x));
  return __builtin_bit_cast(float, r);
}
// ---- the element-wise tail of a layer, moved INTO the next layer's product (round 6) ------------------------------------------------
// With three products a 256 -> 256 block is 12.3 k cycles of MFMA issue and the element-wise epilogue between two blocks (bias, ReLU, sign bits /
// sign mask, the row maximum) ~2.5 k with nothing to hide behind.  Only two things are needed before the next block can start: the activations in
// VGPRs and the row's scale.  So the exposed part is a PRE-PASS — accumulator -> register (+ bias in the forward chain) and the maximum — and every
// element is FINISHED one k-step before it is split, in the shadow of that k-step's first eight MFMAs (one element per slot):
//   FIN 1  forward, training: x = relu(z), its sign bit shifted into the mask word      FIN 2  forward, inference: x = relu(z)
//   FIN 3  backward: x = (raw * invp) masked by the sign word (invp: what takes the previous block's accumulator back, a power of two)
// The row maximum is taken over the UNFINISHED values: max(0, z) is the ReLU's maximum exactly; in the backward chain the unmasked maximum bounds
// the masked one (a scale is only ever an upper bound: nothing overflows, and a row keeps its bits unless every large entry is masked out).
struct HfFin { uint32_t mk[4]; float invp; };
template <int FIN, int E> AF_DEV void hf_finish(float (&in)[128], HfFin& f) {
  if constexpr (FIN == 1) { in[E] = af_relu(in[E]); hf_sign_in(f.mk[E >> 5], in[E]); }
  else if constexpr (FIN == 2) in[E] = af_relu(in[E]);
  else if constexpr (FIN == 3) in[E] = hf_mask_keep<E>(in[E] * f.invp, f.mk[E >> 5]);
}
template <int FIN, int E0, int... Es> AF_DEV void hf_finish_range(float (&in)[128], HfFin& f, std::integer_sequence<int, Es...>) {
  (hf_finish<FIN, E0 + Es>(in, f), ...);
}

// ---- the three-product k-step as 24 hand-placed issue slots (the scheme of mlpbf.hip bf_slot) --------------------------------------
//   MFMAs    0-7 W_l x B_h | [publish, k-steps with S % 4 == 3] | 8-15 W_h x B_l, 16-23 W_h x B_h       (accumulators start at 0: S == 0, slots 0-7 take C = 0)
//   reads    W_h fragments two per slot in 0-3 (used from 8), the next k-step's W_l in 8-11 (of the chunk just published when S % 4 == 3)
//   stores   (training chains) the eight registers 8S .. 8S+7 of the block's input, one per slot in 4-7 and 20-23
//   DMA      pieces of the chunk behind the published one in slots 9, 11, 13, 15, 17 of k-steps S % 4 = 3, 0, 1 and slot 9 of S % 4 = 2:
//            all sixteen are at least 23 MFMAs old at the next publish, and exactly AF_HF_KEEP = 8 stores are younger than the last
//   finish   element 8(S+1) + I of the block's input in slot I < 8 (hf_finish: <= 3 VALU), i.e. before the split below reads it
//   split    of the next k-step's B operand: one pair (4 VALU) in each of slots 12, 14, 16, 18
template <int S, bool STORES, int FIN, int I>
AF_DEV void hf_slot(f32x16 (&acc)[8], float (&in)[128], float sc, HfPipe& pp, f32x4 (&fh)[8], f32x4 (&fln)[8], HfB& bn,
                    uint32_t la, uint32_t nla, HfStream& cs, const TileStore& ts, HfFin& fin) {
  constexpr int sl = S & 3, T = I & 7;
  constexpr bool NEXT = S != 15;
  if constexpr (I == 0) hf_lds_wait(pp.fl[0]);            // the fragments of this slot group have landed (read >= 4 MFMAs ago)
  if constexpr (I == 8 && !(AF_ABL & 128)) hf_lds_wait(fh[0]);      // (AF_ABL bit 7: timing probe, wrong results — is the W_h read latency exposed here?)
  // ---- the MFMA
  if constexpr (I < 8) {
    pin_acc(pp.fl[T]);
    if constexpr (S == 0) {
      const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc[T] = hf_mfma(pp.fl[T], pp.b.h, z);
    } else {
      acc[T] = hf_mfma(pp.fl[T], pp.b.h, acc[T]);
    }
  } else if constexpr (I < 16) { pin_acc(fh[T]); acc[T] = hf_mfma(fh[T], pp.b.l, acc[T]); }
  else acc[T] = hf_mfma(fh[T], pp.b.h, acc[T]);
  // ---- fragment reads
  if constexpr (I < 4) { fh[2 * I] = hf_frag_a<sl, 2 * I, 0>(la); fh[2 * I + 1] = hf_frag_a<sl, 2 * I + 1, 0>(la); }
  if constexpr (I >= 8 && I < 12) {
    constexpr int t0 = 2 * (I - 8);
    if constexpr (sl != 3)    { fln[t0] = hf_frag_a<(sl + 1) & 3, t0, 1>(la); fln[t0 + 1] = hf_frag_a<(sl + 1) & 3, t0 + 1, 1>(la); }
    else if constexpr (NEXT)  { fln[t0] = hf_frag_a<0, t0, 1>(nla); fln[t0 + 1] = hf_frag_a<0, t0 + 1, 1>(nla); }
    else                      { fln[t0] = pp.fl[t0]; fln[t0 + 1] = pp.fl[t0 + 1]; }
  }
  // ---- LDS-DMA pieces of the chunk behind the published one
  if constexpr ((I & 1) && I >= 9 && I <= 17 && (sl != 2 || I == 9)) cs.issue1();
  // ---- tile stores of registers 8S .. 8S+7
  if constexpr (STORES) {
    constexpr int r8 = (I >= 4 && I < 8) ? I - 4 : (I >= 20 ? 4 + I - 20 : -1);
    if constexpr (r8 >= 0) {
      constexpr int rr = 8 * (S & 1) + r8;
      af_bs32_tile(in[8 * S + r8], ts.r, ts.voff + (rr & 3) * 128, (32 * (S >> 1) + 8 * (rr >> 2)) * 128);
    }
  }
  // ---- finish one element of the next k-step's operand
  if constexpr (FIN != 0 && S + 1 < 16 && I < 8) hf_finish<FIN, 8 * (S + 1) + I>(in, fin);
  // ---- split of the next k-step's B operand
  if constexpr (S + 1 < 16 && (I == 12 || I == 14 || I == 16 || I == 18)) {
    constexpr int i = (I - 12) / 2;
    uint32_t hh, ll;
    hf_split_pair(in[8 * (S + 1) + 2 * i], in[8 * (S + 1) + 2 * i + 1], sc, hh, ll);
    bn.h[i] = hh; bn.l[i] = ll;
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int S, bool STORES, int FIN, int... I0, int... I1>
AF_DEV void hf_kstep(f32x16 (&acc)[8], float (&in)[128], float sc, HfPipe& pp, const char*& lane_base, HfStream& cs, int lane_off, int after_bytes, const TileStore& ts, HfFin& fin,
                     std::integer_sequence<int, I0...>, std::integer_sequence<int, I1...>) {
  f32x4 fh[8], fln[8];
  HfB bn = pp.b;
  const char* nxt_lane = lane_base;
  __builtin_amdgcn_sched_barrier(0);
  const uint32_t la = (uint32_t)(size_t)lane_base;
  (hf_slot<S, STORES, FIN, I0>(acc, in, sc, pp, fh, fln, bn, la, la, cs, ts, fin), ...);                           // slots 0..7
  if constexpr ((S & 3) == 3) nxt_lane = cs.template publish<(STORES ? AF_HF_KEEP : 0)>(S == 15 ? after_bytes : AF_SLOT_HF) + lane_off;
  const uint32_t nla = (uint32_t)(size_t)nxt_lane;
  (hf_slot<S, STORES, FIN, 8 + I1>(acc, in, sc, pp, fh, fln, bn, la, nla, cs, ts, fin), ...);                      // slots 8..23
  hf_lds_wait();          // the next k-step's W_l fragments (read in slots 8..11) before anything - a register copy included - touches them
#pragma unroll
  for (int T = 0; T < 8; ++T) pp.fl[T] = fln[T];
  pp.b = bn;
  lane_base = nxt_lane;
}
// A whole 256 -> 256 product (16 k-steps = 4 chunks), accumulators starting at zero.  On entry the first chunk is published at
// lane_base (hf_enter has fetched its first fragments); on exit lane_base addresses the published chunk behind the block
// (after_bytes long).
template <bool STORES, int FIN, int... Ss>
AF_DEV void hf_block_impl(f32x16 (&acc)[8], float (&in)[128], float sc, HfPipe& pp, const char*& lane_base, HfStream& cs, int lane_off, int after_bytes, const TileStore& ts, HfFin& fin,
                          std::integer_sequence<int, Ss...>) {
  (hf_kstep<Ss, STORES, FIN>(acc, in, sc, pp, lane_base, cs, lane_off, after_bytes, ts, fin, std::make_integer_sequence<int, 8>{}, std::make_integer_sequence<int, 16>{}), ...);
}
template <bool STORES, int FIN>
AF_DEV void hf_block(f32x16 (&acc)[8], float (&in)[128], float sc, HfPipe& pp, const char*& lane_base, HfStream& cs, int lane_off, int after_bytes, const TileStore& ts, HfFin& fin) {
  hf_block_impl<STORES, FIN>(acc, in, sc, pp, lane_base, cs, lane_off, after_bytes, ts, fin, std::make_integer_sequence<int, 16>{});
}
// entering a block: the lo-level fragments of its first k-step, the first eight elements of its input finished and split
template <int FIN>
AF_DEV void hf_enter(HfPipe& pp, float (&in)[128], float sc, const char* lane_base, HfFin& fin) {
#pragma unroll
  for (int T = 0; T < 8; ++T) pp.fl[T] = *(const f32x4*)(lane_base + (T * 2 + 1) * 1024);     // compiler-visible reads
  hf_finish_range<FIN, 0>(in, fin, std::make_integer_sequence<int, 8>{});
  pp.b = hf_split_in(in, 0, sc);
}

// ---- first-layer / skip B operand of a row: PE features, or xyt for the mapping nets (mlp.hip) -----------------------------------------
template <class NS, bool TRAIN, int NPE>
AF_DEV void hf_input_stage(const FwdArgs& a, float (&pe)[NPE], int row, int tile, int j, int h, bool live) {
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

// HID: the net has hidden 256 -> 256 layers (nl >= 3); see mlp_fwd_body_bf.
template <class NS, bool TRAIN, bool HID>
AF_DEV void mlp_fwd_body_hf(const FwdArgs& a, int wg, char* smem) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, h = lane >> 5;
  int tile = a.tile0 + wg * 4 + wave;
  const int NT = live_tiles(a);
  if (a.tile0 + wg * 4 >= NT) return;                  // a workgroup of rows that do not exist this iteration (uniform: before any barrier)
  const bool live = tile < NT;
  if (!live) tile = NT - 1;
  const int row = tile * 32 + j;

  using CB = ChunkBytesHf<NS>;
  HfStream cs; cs.smem = smem;
  cs.start(a.wimg, tid, wave);
  cs.issue_bytes(CB::L0);                            // the layer-0 chunk travels while the input stage computes
  const int nl = a.nl;
  stage_bias(nl, a.bias, smem + AF_BIAS_LDS_HF, tid);
  constexpr int NPE = NS::PEG > 0 ? NS::PEG * 4 : 4;
  float pe[NPE];
  hf_input_stage<NS, TRAIN>(a, pe, row, tile, j, h, live);

  const int a_off8 = (h * 256 + j) * 16;     // lane offset inside an fp32 Mpad = 256 image chunk
  const int lane_off = (h * 32 + j) * 16;    // lane offset inside an fp16 chunk
  const int voff_t = (4 * h * 32 + j) * 4;
  const char* bias_lds = smem + AF_BIAS_LDS_HF;
  f32x16 acc[8];
  float in[128];
  TileStore ts{af_rsrc(a.acts, 0), voff_t};
  HfPipe pp;
  float sc = 1.f, inv = 1.f;

  // The PRE-PASS of layer l's tail: acc -> in[] = Z_l (SCALED: acc holds 2^12 sc Z without the bias, a hidden block; else Z itself, the
  // bias-initialised fp32 layer 0), the row maximum of relu(Z_l) and from it the next block's scale.  X_{l+1} = relu(Z_l) and its sign bits are
  // finished element by element inside the next block (hf_finish), or by finish_all() in front of the output layer.
  constexpr int FIN = TRAIN ? 1 : 2;
  HfFin fin; fin.invp = 1.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) fin.mk[i] = 0u;
  auto pre_out = [&](int l, auto scaled) {
    constexpr bool SCALED = decltype(scaled)::value;
    float mx = 0.f;
    // the bias quad of the NEXT four features is fetched while a quad is processed, each feature tile fenced: left to the scheduler all 32 reads
    // of a layer are hoisted to the top (128 live registers, a spilled chain)
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SCALED) bq = *(const f32x4*)(bias_lds + (l * AF_HID + 4 * h) * 4);
#pragma unroll
    for (int T = 0; T < 8; ++T) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 bnx = bq;
        if constexpr (SCALED) { if (T * 4 + q + 1 < 32) bnx = *(const f32x4*)(bias_lds + (l * AF_HID + 8 * (T * 4 + q + 1) + 4 * h) * 4); }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int r = q * 4 + p;
          float z = acc[T][r];
          if constexpr (SCALED) z = __builtin_fmaf(z, inv, bq[p]);
          in[T * 16 + r] = z;
        }
        hf_max3(mx, in[T * 16 + q * 4], in[T * 16 + q * 4 + 1]);
        hf_max3(mx, in[T * 16 + q * 4 + 2], in[T * 16 + q * 4 + 3]);
        bq = bnx;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (TRAIN) ts.r = af_rsrc_uniform(a.acts + ((size_t)l * a.nt_stride + tile) * AF_TILE_F, live ? AF_TILE_F * 4 : 0);
    hf_row_scale(mx, sc, inv);
  };
  // the sign bits of X_{l+1} = relu(Z_l), complete once the block that consumes X_{l+1} (or finish_all) has finished every element
  auto store_masks = [&](int l) {
    if constexpr (TRAIN) {
      if (live) {
        u32x4 m4 = {fin.mk[0], fin.mk[1], fin.mk[2], fin.mk[3]};
        *(u32x4*)(a.masks + (((size_t)l * a.nt_stride + tile) * 64 + lane) * 4) = m4;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) fin.mk[i] = 0u;
    }
  };
  auto hook_dma = [&](auto gi) { if constexpr (decltype(gi)::value < 6) { cs.issue1(); cs.issue1(); } };
  auto hook_none = [&](auto) {};

  // ---- layer 0 (fp32 block); the chunk behind it is the first fp16 chunk of layer 1
  HF_MARK(1);
  const char* cur = cs.publish(CB::L0);             // its barrier also publishes the bias rows
  cs.issue_bytes(nl > 2 ? CB::HID : CB::last_bytes(nl));      // the chunk behind layer 0, whole: it lands under the layer-0 block and its pre-pass
  HF_MARK(2);
  init_bias(acc, bias_lds, 0, h);
  mm_block<8, NS::K0G, 0, 4>(acc, pe, cur + a_off8, hook_none);
  HF_MARK(3);
  pre_out(0, std::false_type{});
  HF_MARK(4);
  const char* lane_base = cs.publish(nl > 2 ? CB::HID : CB::last_bytes(nl)) + lane_off;     // what lies behind layer 0: the first hidden chunk, or the output layer
  cs.lead5();
  HF_MARK(5);

  // ---- hidden layers 1 .. NL-2: four fp16 chunks each (+ the fp32 block of the skip columns)
  if constexpr (HID) {
  int l = 1;
#pragma unroll 1
  do {
    hf_enter<FIN>(pp, in, sc, lane_base, fin);
    const bool skip = NS::SKIP != 0 && ((NS::SKIP >> l) & 1);
    const int behind = l == nl - 2 ? CB::last_bytes(nl) : CB::HID;
    hf_block<(TRAIN && !(AF_ABL & 1)), FIN>(acc, in, sc, pp, lane_base, cs, lane_off, skip ? CB::SKIP : behind, ts, fin);
    store_masks(l - 1);
    if constexpr (NS::SKIP != 0) {
      if (skip) {      // the PE columns of a skip layer on the fp32 pipe, into the SAME (scaled) accumulators: B = 2^12 sc pe
        // scaled in place and back (powers of two, |pe| <= 1: exact both ways) - a second copy of the 20 features does not fit the register file
        const float up = sc * (float)(1 << AF_HF_WSHIFT);
#pragma unroll
        for (int i = 0; i < NPE; ++i) pe[i] *= up;
        mm_block<8, NS::PEG, 0, 4>(acc, pe, lane_base - lane_off + a_off8, hook_dma);
#pragma unroll
        for (int i = 0; i < NPE; ++i) pe[i] *= inv;
        lane_base = cs.publish(behind) + lane_off;
        cs.lead5();
      }
    }
    HF_MARK(4 + 2 * l);
    pre_out(l, std::true_type{});
    HF_MARK(5 + 2 * l);
  } while (++l <= nl - 2);
  }
  lane_base -= lane_off;
  // the last hidden layer's output has no block behind it: finished here, in front of the output layer
  hf_finish_range<FIN, 0>(in, fin, std::make_integer_sequence<int, 128>{});
  store_masks(nl - 2);
  AF_ELEMWISE_FENCE();      // asm element-wise ops never next to an MFMA that reads them (mlpbf.hip relu_out; isa_check.py rule (d))

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
  HF_MARK(30);
}

template <int... Ps> AF_DEV float hf_absmax(const float (&in)[128], std::integer_sequence<int, Ps...>) {
  float mx = 0.f;
  (hf_max3_abs(mx, in[2 * Ps], in[2 * Ps + 1]), ...);
  return mx;
}

// ------------------------------------------------------------------------------------------------
// Backward dX chain on the same scheme: dZ_{l-1} = (W_l^T dZ_l) . relu'(Z_{l-1}) with the fp16 images of W_l^T and one scale per row of dZ_l.
template <class NS>
AF_DEV void mlp_bwd_body_hf(const BwdArgs& a, int wg, char* smem) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, h = lane >> 5;
  int tile = a.tile0 + wg * 4 + wave;
  const int NT = live_tiles(a);
  if (a.tile0 + wg * 4 >= NT) return;                  // a workgroup of rows that do not exist this iteration (uniform: before any barrier)
  const bool live = tile < NT;
  if (!live) tile = NT - 1;
  const int row = tile * 32 + j;

  using CB = ChunkBytesHf<NS>;
  HfStream cs; cs.smem = smem;
  cs.start(a.wimg, tid, wave);
  cs.issue_bytes(CB::BLAST);                         // the output layer's chunk travels while the seed gradient is formed
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
  HfPipe pp;
  float sc = 1.f, inv = 1.f;

  // The PRE-PASS of a block's tail: acc = dX_l (SCALED: times 2^12 sc) -> in[] raw, the sign words of X_l (masks[l-1]) and what takes the raw values
  // back (fin.invp), the row maximum of the unmasked values and from it the next block's scale.  dZ_{l-1} = (raw invp) . relu'(Z_{l-1}) is finished
  // element by element inside the next block (hf_finish<3>), or by the finish in front of the layer-0 stage.
  HfFin fin; fin.invp = 1.f;
  auto pre_mask = [&](int l, auto scaled) {
    const u32x4 m4 = *(const u32x4*)(a.masks + (((size_t)(l - 1) * a.nt_stride + tile) * 64 + lane) * 4);
    fin.mk[0] = m4[0]; fin.mk[1] = m4[1]; fin.mk[2] = m4[2]; fin.mk[3] = m4[3];
    fin.invp = decltype(scaled)::value ? inv : 1.f;
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) in[T * 16 + r] = acc[T][r];
    const float mx = hf_absmax(in, std::make_integer_sequence<int, 64>{});
    ts.r = af_rsrc_uniform(a.dz + ((size_t)(l - 1) * a.nt_stride + tile) * AF_TILE_F, live ? AF_TILE_F * 4 : 0);
    hf_row_scale(mx * fin.invp, sc, inv);
  };
  auto hook_dma = [&](auto gi) { if constexpr (decltype(gi)::value < 6) { cs.issue1(); cs.issue1(); } };
  auto hook_dma_store = [&](auto gi) { if constexpr (decltype(gi)::value < 6) { cs.issue1(); cs.issue1(); } ts.template part<decltype(gi)::value>(in); };
  auto hook_none = [&](auto) {};

  // ---- output layer (fp32 block): K = 8 (one group), only p < OUT non-zero
  HF_MARK(1);
  const char* cur = cs.publish(CB::BLAST);
  cs.issue_bytes(nl > 2 ? CB::HID : (NS::DX0 ? CB::BL0H : 4096));      // the chunk behind it, whole
  HF_MARK(2);
  mm_block<8, 1, 0, NS::OUT, true>(acc, dzl, cur + a_off8, hook_none);
  HF_MARK(3);
  pre_mask(nl - 1, std::false_type{});
  HF_MARK(4);
  const char* lane_base = cs.publish(nl > 2 ? CB::HID : (NS::DX0 ? CB::BL0H : 4096)) + lane_off;
  cs.lead5();
  HF_MARK(5);

#pragma unroll 1
  for (int l = nl - 2; l >= 1; --l) {
    hf_enter<3>(pp, in, sc, lane_base, fin);
    // behind the last hidden block: the atlas net's layer-0 block (first half), or nothing (a harmless stage of the padding)
    hf_block<!(AF_ABL & 1), 3>(acc, in, sc, pp, lane_base, cs, lane_off, l > 1 ? CB::HID : (NS::DX0 ? CB::BL0H : 4096), ts, fin);
    HF_MARK(4 + 2 * (nl - 1 - l));
    pre_mask(l, std::true_type{});
    HF_MARK(5 + 2 * (nl - 1 - l));
  }
  lane_base -= lane_off;
  // dZ_0 has no block behind it on this pipe: finished here
  hf_finish_range<3, 0>(in, fin, std::make_integer_sequence<int, 128>{});
  AF_ELEMWISE_FENCE();      // asm element-wise ops never next to an MFMA that reads them (isa_check.py rule (d))

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
  HF_MARK(30);
}

// ------------------------------------------------------------------------------------------------
template <bool TRAIN>
__global__ __launch_bounds__(256, 1) void k_mlp_fwd_multi_hf(MultiFwd m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AF_STAMP(m, 0);
  HF_MARK(0);
  int s = 0, base = 0;
  const int wg = blockIdx.x;
  while (s + 1 < m.n && wg >= m.wg_end[s]) { base = m.wg_end[s]; ++s; }
  switch (m.net[s]) {
    case AF_NET_MAP1:  if (m.a[s].nl > 2) mlp_fwd_body_hf<NsMap1, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body_hf<NsMap1, TRAIN, false>(m.a[s], wg - base, smem); break;
    case AF_NET_MAP2:  if (m.a[s].nl > 2) mlp_fwd_body_hf<NsMap2, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body_hf<NsMap2, TRAIN, false>(m.a[s], wg - base, smem); break;
    case AF_NET_ATLAS: if (m.a[s].nl > 2) mlp_fwd_body_hf<NsAtlas, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body_hf<NsAtlas, TRAIN, false>(m.a[s], wg - base, smem); break;
    case AF_KIND_MAP_PE: if (m.a[s].nl > 2) mlp_fwd_body_hf<NsMapPe, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body_hf<NsMapPe, TRAIN, false>(m.a[s], wg - base, smem); break;
    default:           if (m.a[s].nl > 2) mlp_fwd_body_hf<NsAlpha, TRAIN, true>(m.a[s], wg - base, smem); else mlp_fwd_body_hf<NsAlpha, TRAIN, false>(m.a[s], wg - base, smem); break;
  }
  AF_STAMP(m, 1);
}

__global__ __launch_bounds__(256, 1) void k_mlp_bwd_multi_hf(MultiBwd m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AF_STAMP(m, 0);
  HF_MARK(0);
  int s = 0, base = 0;
  const int wg = blockIdx.x;
  while (s + 1 < m.n && wg >= m.wg_end[s]) { base = m.wg_end[s]; ++s; }
  switch (m.net[s]) {
    case AF_NET_MAP1:  mlp_bwd_body_hf<NsMap1>(m.a[s], wg - base, smem); break;
    case AF_NET_MAP2:  mlp_bwd_body_hf<NsMap2>(m.a[s], wg - base, smem); break;
    case AF_NET_ATLAS: mlp_bwd_body_hf<NsAtlas>(m.a[s], wg - base, smem); break;
    case AF_KIND_MAP_PE: mlp_bwd_body_hf<NsMapPe>(m.a[s], wg - base, smem); break;
    default:           mlp_bwd_body_hf<NsAlpha>(m.a[s], wg - base, smem); break;
  }
  AF_STAMP(m, 1);
}

extern "C" int af_launch_fwd_multi_hf(MultiFwd* m, int train, hipStream_t s) {
  int tot = 0;
  for (int i = 0; i < m->n; ++i) { tot += (m->a[i].NT - m->a[i].tile0 + 3) / 4; m->wg_end[i] = tot; }
  if (tot <= 0) return 0;
  if (train) hipLaunchKernelGGL((k_mlp_fwd_multi_hf<true>), dim3(tot), dim3(256), AF_LDS_BYTES_HF, s, *m);
  else       hipLaunchKernelGGL((k_mlp_fwd_multi_hf<false>), dim3(tot), dim3(256), AF_LDS_BYTES_HF, s, *m);
  return (int)hipGetLastError();
}
extern "C" int af_launch_bwd_multi_hf(MultiBwd* m, hipStream_t s) {
  int tot = 0;
  for (int i = 0; i < m->n; ++i) { tot += (m->a[i].NT - m->a[i].tile0 + 3) / 4; m->wg_end[i] = tot; }
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(k_mlp_bwd_multi_hf, dim3(tot), dim3(256), AF_LDS_BYTES_HF, s, *m);
  return (int)hipGetLastError();
}
// chunk sizes of the fp16 streams for the host planner: which = 0 fwd layer 0, 1 fp16 hidden chunk (x4 per layer), 2 skip columns,
// 3 fwd output layer, 4 bwd output layer, 5 half of bwd layer 0 (x2)
extern "C" int af_mlp_chunk_bytes_hf(int net, int which, int nl) {
  auto pick = [&](auto ns) -> int {
    using CB = ChunkBytesHf<decltype(ns)>;
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
extern "C" int af_mlp_hf_init() {
  hipError_t e = hipSuccess;
#define AF_ATTR(K) do { hipError_t r = hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, AF_LDS_BYTES_HF); if (r != hipSuccess) e = r; } while (0)
  AF_ATTR((k_mlp_fwd_multi_hf<true>)); AF_ATTR((k_mlp_fwd_multi_hf<false>)); AF_ATTR(k_mlp_bwd_multi_hf);
#undef AF_ATTR
  return (int)e;
}
