// Shader clock under sustained MFMA load: s_memtime (core clock ticks) against s_memrealtime (100 MHz) around a loop of
// back-to-back MFMAs on every SIMD of the chip.  Usage: clockprobe [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE> __global__ __launch_bounds__(256, 1) void k_probe(int iters, unsigned long long* out, float* sink) {
  f32x16 acc[4] = {};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i); }
  float fa = threadIdx.x * 0.5f, fb = 1.25f;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if constexpr (MODE == 0) acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[x], 0, 0, 0);
        else acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[x], 0, 0, 0);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int x = 0; x < 4; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r];
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned long long* d; float* sink;
  hipMalloc(&d, 256 * 16); hipMalloc(&sink, 4);
  for (int mode = 0; mode < 2; ++mode) for (int nwg : {1, 256}) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(nwg), dim3(256), 0, 0, iters, d, sink);
      else hipLaunchKernelGGL(k_probe<1>, dim3(nwg), dim3(256), 0, 0, iters, d, sink);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(nwg * 2);
    hipMemcpy(h.data(), d, nwg * 16, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < nwg; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
    c /= nwg; r /= nwg;
    const double us = r / 100.0, n = (double)iters * 32;
    printf("%s MFMA, %3d workgroups: %.0f us, %.2f core ticks/MFMA by s_memtime, %.2f ns/MFMA -> %.0f cycles at 2.4 GHz; s_memtime rate %.1f MHz\n",
           mode ? "f32 32x32x2 " : "bf16 32x32x16", nwg, us, c / n, us * 1000 / n, us * 1000 / n * 2.4, c / us);
  }
  return 0;
}
