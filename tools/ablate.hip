// tools/ablate.hip — kernel ablation harness (not product code): times the fused mapping-net chains on synthetic buffers,
// the 32-row kernels (mlp.hip, optionally with AF_ABL ablation bits) and the 16-row pre-train kernels (mlp16.hip).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAF_ABL=<bits> tools/ablate.hip -o tools/bin/ablate_<bits>
// Run:   ablate_<bits> [row tiles of 32, default 2813]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../all-in-one-deflicker_amd/csrc/mlp.hip"
#include "../all-in-one-deflicker_amd/csrc/mlp16.hip"
#include "../all-in-one-deflicker_amd/csrc/mlpbf.hip"

// Ceiling probe: a pure v_mfma_f32_32x32x2_f32 stream in the geometry of the chains — 256 threads,
// one wave per SIMD (launch_bounds(256,1) + 128 KB of dynamic LDS so no second workgroup co-resides), 8 independent
// accumulators, operands in VGPRs, no LDS / VMEM traffic inside the loop.  FLOPs = 2*32*32*2 per MFMA per wave.
__global__ __launch_bounds__(256, 1) void k_mfma_ceiling(float* out, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  f32x16 acc[8];
#pragma unroll
  for (int T = 0; T < 8; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[T][r] = 0.f;
  float a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed * (float)(threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = seed + (float)i;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int T = 0; T < 8; ++T) acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[T], b[p], acc[T], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int T = 0; T < 8; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[T][r];
  if (s == 12345.678f) out[threadIdx.x] = s + smem_c[0];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int NT = argc > 1 ? atoi(argv[1]) : 2813;
  const int reps = 20;
  const int only_net = argc > 3 ? atoi(argv[3]) : -1;     // ablate <NT> 0 <net>: the bf16x6 chains of that net alone (0 mapping1, 1 atlas, 2 mapping2, 3 alpha)
  af_mlp_init(); af_mlp16_init(); af_mlp_bf_init();
  // mapping1 image: forward 8K + 16 x 64K + 4K, backward 8K + 16 x 64K; every stage copies 64 KB -> pad
  const size_t img_bytes = (size_t)16 << 20;         // covers the fp32 images and the bf16 stream of mapping1
  float *img, *bias, *in, *out, *acts, *dz, *dzl; uint32_t* masks;
  CK(hipMalloc(&img, img_bytes)); CK(hipMalloc(&bias, 8 * 256 * 4)); CK(hipMemset(bias, 0, 8 * 256 * 4));
  CK(hipMalloc(&in, (size_t)NT * 32 * 16)); CK(hipMemset(in, 0, (size_t)NT * 32 * 16));
  CK(hipMalloc(&out, (size_t)NT * 32 * 16)); CK(hipMemset(out, 0, (size_t)NT * 32 * 16));
  CK(hipMalloc(&acts, (size_t)7 * NT * 32768)); CK(hipMalloc(&dz, (size_t)7 * NT * 32768));
  CK(hipMalloc(&dzl, (size_t)NT * 4096)); CK(hipMalloc(&masks, (size_t)7 * NT * 1024)); CK(hipMemset(masks, 0xff, (size_t)7 * NT * 1024));
  float* pe_tile; CK(hipMalloc(&pe_tile, (size_t)NT * 8192)); CK(hipMemset(pe_tile, 0, (size_t)NT * 8192));
  std::vector<float> w(img_bytes / 4); for (auto& x : w) x = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  CK(hipMemcpy(img, w.data(), img_bytes, hipMemcpyHostToDevice));
  FwdArgs fa{}; fa.wimg = img; fa.bias = bias; fa.in = in; fa.in1 = nullptr; fa.out = out; fa.acts = acts; fa.masks = masks;
  fa.in_scale = 0.5f; fa.in_shift0 = 0.5f; fa.split_row = 1 << 30; fa.NT = NT; fa.nt_stride = NT; fa.pe_tile = pe_tile; fa.nl = only_net == 1 || only_net == 3 ? 8 : (only_net == 2 ? 4 : 6);
  BwdArgs ba{}; ba.wimg = img; ba.out = out; ba.dout = in; ba.masks = masks; ba.dz = dz; ba.dz_last = dzl;
  ba.split_row = 1 << 30; ba.NT = NT; ba.nt_stride = NT; ba.pe_tile = pe_tile; ba.nl = fa.nl;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (argc > 2 && atoi(argv[2]) == 1) {     // ablate <NT> 1: the pure-MFMA ceiling
    CK(hipFuncSetAttribute((const void*)k_mfma_ceiling, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int wgs : {256, 512, 704}) {
      const int iters = 160;                // 160 x 32 = 5120 MFMAs per wave ~ one mapping chain
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_mfma_ceiling, dim3(wgs), dim3(256), 131072, 0, out, iters, 0.f);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_mfma_ceiling, dim3(wgs), dim3(256), 131072, 0, out, iters, 0.f);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      const double fl = (double)wgs * 4 * iters * 32 * 4096.0;
      printf("mfma_ceiling wgs=%d: %.4f ms  %.1f TF (%.1f%% of 157.3; %.1f%% counting whole rounds of 256 CUs)\n", wgs, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100,
             (double)((wgs + 255) / 256 * 256) * 4 * iters * 32 * 4096.0 / ms / 1e9 / 157.3 * 100);
    }
    return 0;
  }
  if (only_net >= 0) {
    std::vector<float> hin((size_t)NT * 32 * 4); for (auto& x : hin) x = rand() / (float)RAND_MAX * 2.f - 1.f;     // uv in [-1, 1] like the tanh outputs the PE nets read
    CK(hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
    for (int dir = 0; dir < 2; ++dir) {
      auto go = [&]() {
        if (dir == 0) { MultiFwd m{}; m.n = 1; m.net[0] = only_net; m.a[0] = fa; af_launch_fwd_multi_bf(&m, 1, 0); }
        else { MultiBwd m{}; m.n = 1; m.net[0] = only_net; m.a[0] = ba; af_launch_bwd_multi_bf(&m, 0); }
      };
      for (int r = 0; r < 3; ++r) go();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) go();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      printf("net %d %s bf16x6 chain, NT=%d (%.2f rounds of 1024 SIMDs): %.1f us\n", only_net, dir ? "bwd" : "fwd", NT, NT / 1024.0, ms * 1000);
    }
    return 0;
  }
  const char* names[6] = {"fwd_map32", "bwd_map32", "fwd_map16", "bwd_map16", "fwd_map32_bf16x6", "bwd_map32_bf16x6"};
  for (int which = 0; which < 6; ++which) {
    auto go = [&]() {
      if (which == 0) { MultiFwd m{}; m.n = 1; m.net[0] = AF_NET_MAP1; m.a[0] = fa; af_launch_fwd_multi(&m, 1, 0); }
      else if (which == 1) { MultiBwd m{}; m.n = 1; m.net[0] = AF_NET_MAP1; m.a[0] = ba; af_launch_bwd_multi(&m, 0); }
      else if (which == 2) af_launch_fwd16(AF_NET_MAP1, &fa, 0);
      else if (which == 3) af_launch_bwd16(AF_NET_MAP1, &ba, 0);
      else if (which == 4) { MultiFwd m{}; m.n = 1; m.net[0] = AF_NET_MAP1; m.a[0] = fa; af_launch_fwd_multi_bf(&m, 1, 0); }
      else { MultiBwd m{}; m.n = 1; m.net[0] = AF_NET_MAP1; m.a[0] = ba; af_launch_bwd_multi_bf(&m, 0); }
    };
    for (int r = 0; r < 3; ++r) go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) go();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double fl = (double)NT * 32 * ((which & 1) == 0 ? 526848.0 : 525312.0);
    printf("ABL=%d %s NT=%d: %.4f ms  %.1f TF (%.1f%% of 157.3)\n", AF_ABL, names[which], NT, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
#ifdef AF_CLK
    if (which >= 4) {
      const int nwg = (NT + 3) / 4;
      unsigned long long* clk; CK(hipMalloc(&clk, (size_t)nwg * 16)); CK(hipMemset(clk, 0, (size_t)nwg * 16));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_af_clk), &clk, sizeof clk));
      go(); CK(hipDeviceSynchronize());
      std::vector<unsigned long long> hc((size_t)nwg * 2); CK(hipMemcpy(hc.data(), clk, (size_t)nwg * 16, hipMemcpyDeviceToHost));
      double tk = 0; for (int w = 0; w < nwg; ++w) tk += (double)(hc[2 * w + 1] - hc[2 * w]); tk /= nwg;
      unsigned long long* none = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_af_clk), &none, sizeof none));
      printf("  mean core ticks per workgroup %.0f (MFMA issue alone: %d); ticks / launch time = %.0f MHz\n", tk, 4 * 16 * 48 * 32 + 2048, tk / (ms * 1000));
    }
#endif
  }
  return 0;
}
