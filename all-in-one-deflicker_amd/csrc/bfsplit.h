// bfsplit.h — fp32-faithful operands for the bf16 matrix pipe ("bf16x6"), shared by k_dw_bf (dw.hip) and the bf16 chains
// (mlpbf.hip): every fp32 value is split in registers into three bf16 values hi + mid + lo (8 + 8 + 8 mantissa bits,
// round-to-nearest-even at each level, the two residuals exact fp32 subtractions), and a product a*b is accumulated as
// the six leading partial products hh + hm + mh + mm + hl + lh in the MFMA's fp32 accumulator.  The dropped terms
// (ml, lm, ll) are <= 2^-23 |ab|: the size of the rounding of one fp32 multiply.
#pragma once
#include "af_dev.h"

#ifndef DW_ABL
#define DW_ABL 0
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct DwSplit { u32x4 h, m, l; };
AF_DEV uint32_t dw_pk(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }   // v_cvt_pk_bf16_f32 (RNE)
AF_DEV float dw_sub(float a, float b) {
  if constexpr (DW_ABL & 8) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }   // keeps hipcc from SLP-packing into v_pk_add_f32
  return a - b;
}
// x - float(one half of a packed bf16 pair) in ONE instruction: v_dot2(c)_f32_bf16 computes h.lo * b.lo + h.hi * b.hi + c with the
// products exact in fp32; with b = (-1, 0) or (0, -1) that is c - h.lo / c - h.hi, exactly representable (the residual of a
// round-to-nearest bf16), so no rounding takes place.  Replaces the shift / mask that widens the half plus the subtraction.
// Measured (tools/dwbench.hip, tools/ablate.hip): bit-identical (tools/splitcheck.hip) but SLOWER — 5 451 instead of 4 820 ticks per
// k_dw_bf stage, 175-181 k instead of 170 k per chain: the dot instruction is not a single-pass VALU op.  Off; kept as a probe.
#ifndef DW_SPLIT_DOT2
#define DW_SPLIT_DOT2 0
#endif
// The selector travels in an SGPR the compiler cannot see through: folded to a constant it becomes the inline operand "-1.0",
// which the instruction reads as the 32-bit pattern 0xbf800000 = (lo 0, hi -1) for BOTH selectors (tools/splitcheck.hip).
AF_DEV uint32_t dw_sel_lo() { uint32_t s; asm("s_mov_b32 %0, 0x0000bf80" : "=s"(s)); return s; }
AF_DEV uint32_t dw_sel_hi() { uint32_t s; asm("s_mov_b32 %0, 0xbf800000" : "=s"(s)); return s; }
AF_DEV float dw_res_lo(uint32_t h, float x) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), __builtin_bit_cast(bf16x2, dw_sel_lo()), x, false);
}
AF_DEV float dw_res_hi(uint32_t h, float x) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), __builtin_bit_cast(bf16x2, dw_sel_hi()), x, false);
}
// LEVELS = 3: hi + mid + lo (six products, fp32-faithful).  LEVELS = 2: hi + mid only, for the three-product variant of k_dw
// (hh + hm + mh, products of 16-bit mantissas: ~2^-17 per product — far inside the distance the fp32 reference's own weight
// gradients keep from an fp64 twin, which is the acceptance test it ships under, tests/test_gpu_dw_modes.py).
template <int LEVELS = 3>
AF_DEV DwSplit dw_split8(const f32x4& lo4, const f32x4& hi4) {
  DwSplit s;
  if constexpr (DW_ABL & 1) {
    s.h = __builtin_bit_cast(u32x4, lo4); s.m = __builtin_bit_cast(u32x4, hi4); s.l = s.h;
    return s;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = i < 2 ? lo4[2 * i] : hi4[2 * i - 4], b = i < 2 ? lo4[2 * i + 1] : hi4[2 * i - 3];
    const uint32_t h = dw_pk(a, b);
#if DW_SPLIT_DOT2
    const float ra = dw_res_lo(h, a), rb = dw_res_hi(h, b);
#else
    const float ra = dw_sub(a, __builtin_bit_cast(float, h << 16)), rb = dw_sub(b, __builtin_bit_cast(float, h & 0xffff0000u));
#endif
    const uint32_t m = dw_pk(ra, rb);
    s.h[i] = h; s.m[i] = m;
    if constexpr (LEVELS == 3) {
#if DW_SPLIT_DOT2
      const float qa = dw_res_lo(m, ra), qb = dw_res_hi(m, rb);
#else
      const float qa = dw_sub(ra, __builtin_bit_cast(float, m << 16)), qb = dw_sub(rb, __builtin_bit_cast(float, m & 0xffff0000u));
#endif
      s.l[i] = dw_pk(qa, qb);
    } else {
      s.l[i] = 0u;
    }
  }
  return s;
}
AF_DEV f32x16 dw_mfma_bf(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

