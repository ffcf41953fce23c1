// tools/splitcheck.hip — the two formulations of the three-way bf16 split (bfsplit.h: shift/mask + subtract, and v_dot2_f32_bf16)
// must agree bit for bit on every finite fp32 below the bf16 overflow threshold (they do; the dot2 form is slower and off, see
// bfsplit.h).  Usage: splitcheck
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#define DW_SPLIT_DOT2 1
#include "../all-in-one-deflicker_amd/csrc/bfsplit.h"
namespace ref {
__device__ __forceinline__ void split(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = dw_pk(a, b);
  const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
  m = dw_pk(ra, rb);
  const float qa = ra - __builtin_bit_cast(float, m << 16), qb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
  l = dw_pk(qa, qb);
}
}
__global__ void k_check(const float* x, uint32_t* out_new, uint32_t* out_ref, int n8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const f32x4 lo = *(const f32x4*)(x + (size_t)i * 8), hi = *(const f32x4*)(x + (size_t)i * 8 + 4);
  const DwSplit s = dw_split8(lo, hi);
  for (int p = 0; p < 4; ++p) {
    out_new[(size_t)i * 12 + p] = s.h[p]; out_new[(size_t)i * 12 + 4 + p] = s.m[p]; out_new[(size_t)i * 12 + 8 + p] = s.l[p];
    const float a = p < 2 ? lo[2 * p] : hi[2 * p - 4], b = p < 2 ? lo[2 * p + 1] : hi[2 * p - 3];
    uint32_t h, m, l; ref::split(a, b, h, m, l);
    out_ref[(size_t)i * 12 + p] = h; out_ref[(size_t)i * 12 + 4 + p] = m; out_ref[(size_t)i * 12 + 8 + p] = l;
  }
}
int main() {
  const int n8 = 1 << 20, n = n8 * 8;
  std::vector<float> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    uint32_t bits = ((uint32_t)rand() << 17) ^ ((uint32_t)rand() << 2) ^ (uint32_t)rand();      // every exponent, both signs, denormals, inf / NaN patterns
    if (i % 3 == 1) bits = (bits & 0x807fffffu) | ((100u + (uint32_t)(rand() % 56)) << 23);    // the magnitudes the kernels see (2^-27 .. 2^28)
    if (i % 97 == 0) bits &= 0x807fffffu;                                                       // denormals
    memcpy(&h[i], &bits, 4);
  }
  float* dx; uint32_t *da, *db;
  hipMalloc(&dx, (size_t)n * 4); hipMalloc(&da, (size_t)n8 * 48); hipMalloc(&db, (size_t)n8 * 48);
  hipMemcpy(dx, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3(n8 / 256), dim3(256), 0, 0, dx, da, db, n8);
  std::vector<uint32_t> a((size_t)n8 * 12), b((size_t)n8 * 12);
  hipMemcpy(a.data(), da, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, b.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0, bad_finite = 0;
  for (size_t i = 0; i < a.size(); ++i) if (a[i] != b[i]) {
    ++bad;
    const size_t g = i / 12, p = i % 4; const float x0 = h[g * 8 + 2 * p], x1 = h[g * 8 + 2 * p + 1];
    if (std::isfinite(x0) && std::isfinite(x1) && std::fabs(x0) < 3e38f && std::fabs(x1) < 3e38f) { if (bad_finite++ < 8) printf("mismatch level %zu: inputs %a %a -> %08x vs %08x\n", (i % 12) / 4, x0, x1, a[i], b[i]); }
  }
  printf("%zu packed words compared, %zu differ, %zu of them with finite inputs below the bf16 overflow threshold\n", a.size(), bad, bad_finite);
  return bad_finite != 0;
}
