// tools/hazardprobe.hip — which instruction pair corrupted the two-layer chains without the element-wise fence? (not product code)
//
// Round 3 met a corrupted forward of a two-layer net and held it off with __builtin_amdgcn_sched_barrier(0) behind the inline-asm
// element-wise ops (af_relu = one v_max_f32, bf_mask_keep); the comment blamed "a VGPR rewritten under an in-flight global_load_lds".
// Round 5: the build without the fence (-DAF_NO_ELEMWISE_FENCE) still fails (mapping net with 2 layers, forward, 0.037 off) while
//   * the shipped build rewrites a DMA piece's address VGPRs in the very next instruction 368 times and is correct -> not that;
//   * a symbolic execution of the failing region gives the SAME 128 products in both builds -> the compiler's output is logically right;
//   * the only dependency that exists in the failing build and not in the shipped one: an inline-asm v_max_f32 whose result a
//     v_mfma_f32_4x4x1 reads as SrcB a few instructions later (distance 3; shipped: >= 39), and VALU writes next to MFMA reads.
// hipcc's hazard recogniser does not see INSIDE an asm statement: if gfx950 needs wait states between a VALU write and an MFMA read of
// that VGPR, they are inserted for compiler-visible VALU ops and silently missing behind an asm one.  This probe measures exactly that:
// one asm block = { VALU writes vB ; d-1 wait states ; MFMA reads vB as SrcB (or SrcA) } for d = 1..8, against the same MFMA issued
// 24 wait states later.  A stale read shows as a result built from the OLD content of vB.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hazardprobe.hip -o tools/bin/hazardprobe        Run: hazardprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// KIND 0: 4x4x1 f32, new value in SrcB   1: 4x4x1 f32, new value in SrcA   2: 32x32x2 f32, SrcB   3: v_accvgpr_write then MFMA SrcC (AGPR)
template <int KIND, int D> __device__ __forceinline__ void probe(float a, float oldv, float newv, f32x4& out4, f32x16& out16) {
  float b = oldv;
  if constexpr (KIND == 0) {
    asm volatile("s_nop 7\n\tv_mov_b32 %1, %2\n\ts_nop 7\n\tv_max_f32 %1, 0, %3\n\t.if %5 > 1\n\ts_nop %5 - 2\n\t.endif\n\tv_mfma_f32_4x4x1_16b_f32 %0, %4, %1, 0\n\ts_nop 7\n\ts_nop 7"
                 : "=&v"(out4), "+&v"(b) : "v"(oldv), "v"(newv), "v"(a), "n"(D));
  } else if constexpr (KIND == 1) {
    asm volatile("s_nop 7\n\tv_mov_b32 %1, %2\n\ts_nop 7\n\tv_max_f32 %1, 0, %3\n\t.if %5 > 1\n\ts_nop %5 - 2\n\t.endif\n\tv_mfma_f32_4x4x1_16b_f32 %0, %1, %4, 0\n\ts_nop 7\n\ts_nop 7"
                 : "=&v"(out4), "+&v"(b) : "v"(oldv), "v"(newv), "v"(a), "n"(D));
  } else if constexpr (KIND == 2) {
    asm volatile("s_nop 7\n\tv_mov_b32 %1, %2\n\ts_nop 7\n\tv_max_f32 %1, 0, %3\n\t.if %5 > 1\n\ts_nop %5 - 2\n\t.endif\n\tv_mfma_f32_32x32x2_f32 %0, %4, %1, 0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                 : "=&v"(out16), "+&v"(b) : "v"(oldv), "v"(newv), "v"(a), "n"(D));
  }
}

template <int KIND> __global__ void k_probe(const float* in, float* out, int* bad) {
  const int lane = threadIdx.x;
  const float a = in[lane], oldv = in[64 + lane], newv = in[128 + lane];
  f32x4 r4[9]; f32x16 r16[9];
#define P(D) probe<KIND, D>(a, oldv, newv, r4[D], r16[D])
  P(1); P(2); P(3); P(4); P(5); P(6); P(7); P(8);
  // the reference: the same MFMA with the new value long settled
  {
    float b = newv > 0.f ? newv : 0.f;
    if constexpr (KIND == 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_mfma_f32_4x4x1_16b_f32 %0, %2, %1, 0\n\ts_nop 7\n\ts_nop 7" : "=&v"(r4[0]) : "v"(b), "v"(a));
    else if constexpr (KIND == 1) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0\n\ts_nop 7\n\ts_nop 7" : "=&v"(r4[0]) : "v"(b), "v"(a));
    else asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_mfma_f32_32x32x2_f32 %0, %2, %1, 0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "=&v"(r16[0]) : "v"(b), "v"(a));
  }
  for (int d = 1; d <= 8; ++d) {
    bool diff = false;
    if constexpr (KIND < 2) { for (int i = 0; i < 4; ++i) diff |= __builtin_bit_cast(unsigned, r4[d][i]) != __builtin_bit_cast(unsigned, r4[0][i]); }
    else { for (int i = 0; i < 16; ++i) diff |= __builtin_bit_cast(unsigned, r16[d][i]) != __builtin_bit_cast(unsigned, r16[0][i]); }
    if (diff) atomicAdd(&bad[KIND * 16 + d], 1);
  }
  if (blockIdx.x == 0) out[KIND * 64 + lane] = KIND < 2 ? r4[1][0] : r16[1][0];
}

int main() {
  float *in, *out; int* bad;
  CK(hipMalloc(&in, 192 * 4)); CK(hipMalloc(&out, 4 * 64 * 4)); CK(hipMalloc(&bad, 64 * 4)); CK(hipMemset(bad, 0, 64 * 4));
  std::vector<float> h(192);
  for (int i = 0; i < 64; ++i) { h[i] = 0.25f + i * 0.03125f; h[64 + i] = 100.f + i; h[128 + i] = 1.f + i * 0.5f; }     // a, old content of vB, new content
  CK(hipMemcpy(in, h.data(), 192 * 4, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 50; ++rep) {
    hipLaunchKernelGGL(k_probe<0>, dim3(1024), dim3(64), 0, 0, in, out, bad);
    hipLaunchKernelGGL(k_probe<1>, dim3(1024), dim3(64), 0, 0, in, out, bad);
    hipLaunchKernelGGL(k_probe<2>, dim3(1024), dim3(64), 0, 0, in, out, bad);
  }
  CK(hipDeviceSynchronize());
  int hb[64]; CK(hipMemcpy(hb, bad, sizeof hb, hipMemcpyDeviceToHost));
  const char* names[3] = {"v_max_f32 (VALU write) -> v_mfma_f32_4x4x1 SrcB", "v_max_f32 (VALU write) -> v_mfma_f32_4x4x1 SrcA", "v_max_f32 (VALU write) -> v_mfma_f32_32x32x2 SrcB"};
  for (int k = 0; k < 3; ++k) {
    printf("%s: lanes with a result that differs from the settled one, of %d, by distance d (d - 1 wait states between the two instructions):\n   ", names[k], 50 * 1024 * 64);
    for (int d = 1; d <= 8; ++d) printf(" d=%d: %d ", d, hb[k * 16 + d]);
    printf("\n");
  }
  return 0;
}
