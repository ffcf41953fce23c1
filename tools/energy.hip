// tools/energy.hip — sustained single-kernel loops for the energy budget of the step (not product code).
//
// The board runs the training step at its 1400 W limit with 740 W idle (profiles/r4_step_clock.json), so the step time is the step's
// DYNAMIC ENERGY over (cap - idle).  This harness runs ONE kernel (or an ablated build of it) back to back for a few seconds per phase and
// prints wall-clock stamps; tools/energy_budget.py samples the board power meanwhile and turns (P - P_idle) x time-per-launch into joules
// per launch.  Component energies come out as differences between builds at EQUAL work (same MFMA count, same rows):
//   chains (mlpbf.hip): -DAF_ABL bits  1 no tile stores, 2 no LDS-DMA, 8 no LDS fragment reads, 32 no operand split, 43 = all four;
//   k_dw (dw.hip):      -DDW_ABL bits  1 no operand split, 2 one MFMA per product group instead of six; L2-resident operands = phase dw_l2.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEN_CHAINS [-DAF_ABL=n] tools/energy.hip -o tools/bin/en_c<n>
//        hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEN_DW [-DDW_ABL=n] tools/energy.hip -o tools/bin/en_d<n>
// Run:   en_x <seconds per phase> <phase> [<phase> ...]      (phases below; every phase prints one "PHASE ..." line with epoch stamps)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <functional>
#include <thread>
#include <vector>
#if defined(EN_CHAINS)
#include "../all-in-one-deflicker_amd/csrc/mlpbf.hip"
#endif
#if defined(EN_DW)
#include "../all-in-one-deflicker_amd/csrc/dw.hip"
#endif
#if !defined(EN_CHAINS) && !defined(EN_DW)
#include "../all-in-one-deflicker_amd/csrc/af_dev.h"
#endif
#ifndef AF_ABL
#define AF_ABL 0
#endif
#ifndef DW_ABL
#define DW_ABL 0
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __bf16 en_bf16x8 __attribute__((ext_vector_type(8)));
typedef float en_f32x16 __attribute__((ext_vector_type(16)));

// bare matrix-pipe load in the geometry of the chains: one wave per SIMD, 4 independent accumulators, operands in registers
template <int MODE> __global__ __launch_bounds__(256, 1) void k_en_mfma(int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char en_smem[];
  en_f32x16 acc[4] = {};
  en_bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.37f + i * 1.13f); b[i] = (__bf16)(1.0f + i * 0.71f + blockIdx.x * 0.01f); }
  float fa = threadIdx.x * 0.5f, fb = 1.25f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if constexpr (MODE == 0) acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[x], 0, 0, 0);
        else acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[x], 0, 0, 0);
      }
  }
  float s = 0;
  for (int x = 0; x < 4; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r];
  if (s == 123.456f) sink[0] = s + en_smem[0];
}
// HBM streams: read-only (sum) and copy, 16 bytes per lane per trip, grid-stride
typedef float en_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_en_read(const en_f32x4* __restrict__ src, size_t n, float* sink) {
  en_f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += __builtin_nontemporal_load(src + i);
  if (s.x + s.y + s.z + s.w == 123.456f) sink[0] = s.x;
}
__global__ __launch_bounds__(256) void k_en_copy(const en_f32x4* __restrict__ src, en_f32x4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

static double now_s() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }

// run go() back to back for `secs` seconds (batches of `batch` launches between host checks), print the phase line
static void sustain(const char* name, double secs, int batch, const std::function<void()>& go, double units_per_launch, const char* unit) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r) go();
  CK(hipDeviceSynchronize());
  const double t0 = now_s();
  long long n = 0; double ms_sum = 0;
  while (now_s() - t0 < secs) {
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < batch; ++r) go();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_sum += ms; n += batch;
  }
  const double t1 = now_s();
  printf("PHASE %s AF_ABL=%d DW_ABL=%d t0=%.3f t1=%.3f launches=%lld ms_per_launch=%.6f units_per_launch=%.6g unit=%s\n", name, AF_ABL, DW_ABL, t0, t1, n, ms_sum / n, units_per_launch, unit);
  fflush(stdout);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  if (argc < 3) { printf("usage: %s <seconds per phase> <phase>...\n", argv[0]); return 1; }
  const double secs = atof(argv[1]);
  float* sink; CK(hipMalloc(&sink, 64));
#if defined(EN_CHAINS)
  af_mlp_bf_init();
  const int NT = 8192;                                  // 8 rounds of the 1024 SIMDs per launch
  const size_t img_bytes = (size_t)16 << 20;
  float *img, *bias, *in, *out, *acts, *dz, *dzl, *pe_tile; uint32_t* masks;
  CK(hipMalloc(&img, img_bytes)); CK(hipMalloc(&bias, 8 * 256 * 4)); CK(hipMemset(bias, 0, 8 * 256 * 4));
  CK(hipMalloc(&in, (size_t)NT * 32 * 16)); CK(hipMalloc(&out, (size_t)NT * 32 * 16)); CK(hipMemset(out, 0, (size_t)NT * 32 * 16));
  CK(hipMalloc(&acts, (size_t)7 * NT * 32768)); CK(hipMalloc(&dz, (size_t)7 * NT * 32768));
  CK(hipMalloc(&dzl, (size_t)NT * 4096)); CK(hipMalloc(&masks, (size_t)7 * NT * 1024));
  CK(hipMalloc(&pe_tile, (size_t)NT * 8192)); CK(hipMemset(pe_tile, 0, (size_t)NT * 8192));
  {
    std::vector<float> w(img_bytes / 4); for (auto& x : w) x = (rand() / (float)RAND_MAX - 0.5f) * 0.15f;
    CK(hipMemcpy(img, w.data(), img_bytes, hipMemcpyHostToDevice));
    std::vector<float> hin((size_t)NT * 32 * 4); for (auto& x : hin) x = rand() / (float)RAND_MAX * 2.f - 1.f;
    CK(hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> hm((size_t)7 * NT * 256); for (auto& x : hm) x = ((uint32_t)rand() << 16) ^ (uint32_t)rand();     // ~half the units on, like a trained net
    CK(hipMemcpy(masks, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
  }
#endif
#if defined(EN_DW)
  af_dw_init();
  const int per = 66, nwg = 256, NTD = per * nwg;
  float *A, *B, *partial;
  CK(hipMalloc(&A, (size_t)NTD * AF_TILE_F * 4)); CK(hipMalloc(&B, (size_t)NTD * AF_TILE_F * 4));
  CK(hipMalloc(&partial, (size_t)nwg * (65536 + 256) * 4));
  {
    std::vector<float> h((size_t)1 << 22); for (auto& x : h) x = (rand() & 1) ? rand() / (float)RAND_MAX - 0.5f : 0.f;      // half zeros: post-ReLU activations / masked dZ
    for (size_t off = 0; off < (size_t)NTD * AF_TILE_F; off += h.size()) {
      const size_t n = std::min(h.size(), (size_t)NTD * AF_TILE_F - off);
      CK(hipMemcpy(A + off, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B + off, h.data(), n * 4, hipMemcpyHostToDevice));
    }
  }
  std::vector<DwSeg> segs((size_t)nwg * DW_MAXSEG, DwSeg{-1, 0, 0, 0});
  for (int w = 0; w < nwg; ++w) segs[(size_t)w * DW_MAXSEG] = DwSeg{0, w * per, (w + 1) * per, w};
  DwJob* dj[2]; DwSeg* ds; CK(hipMalloc(&ds, segs.size() * sizeof(DwSeg)));
  CK(hipMemcpy(ds, segs.data(), segs.size() * sizeof(DwSeg), hipMemcpyHostToDevice));
  for (int cached = 0; cached < 2; ++cached) {
    DwJob j{}; j.A = A; j.B = B; j.a_stride = cached ? 0 : AF_TILE_F; j.b_stride = cached ? 0 : AF_TILE_F; j.shape = DW_8x8; j.part_off = 0; j.part_blk = 65536 + 256;
    CK(hipMalloc(&dj[cached], sizeof j)); CK(hipMemcpy(dj[cached], &j, sizeof j, hipMemcpyHostToDevice));
  }
#endif
  for (int ai = 2; ai < argc; ++ai) {
    const char* ph = argv[ai];
    if (!strcmp(ph, "pci")) {
      char id[64] = {0}; CK(hipDeviceGetPCIBusId(id, sizeof id, 0)); printf("PCI %s\n", id); fflush(stdout);
    } else if (!strcmp(ph, "idle")) {
      CK(hipDeviceSynchronize());
      const double t0 = now_s(); std::this_thread::sleep_for(std::chrono::duration<double>(secs)); const double t1 = now_s();
      printf("PHASE idle AF_ABL=%d DW_ABL=%d t0=%.3f t1=%.3f launches=0 ms_per_launch=0 units_per_launch=0 unit=none\n", AF_ABL, DW_ABL, t0, t1); fflush(stdout);
    } else if (!strcmp(ph, "mfma_bf16") || !strcmp(ph, "mfma_f32")) {
      const bool bf = !strcmp(ph, "mfma_bf16");
      const int iters = bf ? 400 : 200;                  // 12 800 / 6 400 MFMAs per wave and launch
      CK(hipFuncSetAttribute((const void*)k_en_mfma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
      CK(hipFuncSetAttribute((const void*)k_en_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
      sustain(ph, secs, 20, [&] { if (bf) hipLaunchKernelGGL(k_en_mfma<0>, dim3(256), dim3(256), 131072, 0, iters, sink); else hipLaunchKernelGGL(k_en_mfma<1>, dim3(256), dim3(256), 131072, 0, iters, sink); },
              256.0 * 4 * iters * 32, "MFMA");
    } else if (!strcmp(ph, "hbm_read") || !strcmp(ph, "hbm_copy")) {
      static en_f32x4 *src = nullptr, *dst = nullptr; const size_t n = (size_t)1 << 27;     // 2 GiB per buffer: far beyond L2 + MALL
      if (!src) { CK(hipMalloc(&src, n * 16)); CK(hipMalloc(&dst, n * 16)); CK(hipMemset(src, 1, n * 16)); }
      if (!strcmp(ph, "hbm_read")) sustain(ph, secs, 5, [&] { hipLaunchKernelGGL(k_en_read, dim3(256 * 16), dim3(256), 0, 0, src, n, sink); }, n * 16.0 / 1e9, "GB");
      else sustain(ph, secs, 5, [&] { hipLaunchKernelGGL(k_en_copy, dim3(256 * 16), dim3(256), 0, 0, src, dst, n); }, 2.0 * n * 16.0 / 1e9, "GB");
    }
#if defined(EN_CHAINS)
    else if (!strncmp(ph, "fwd_", 4) || !strncmp(ph, "bwd_", 4) || !strncmp(ph, "bw3_", 4)) {
      const int net = !strcmp(ph + 4, "map") ? 0 : (!strcmp(ph + 4, "atlas") ? 1 : -1);
      if (net < 0) { printf("unknown phase %s\n", ph); return 1; }
      FwdArgs fa{}; fa.wimg = img; fa.bias = bias; fa.in = in; fa.in1 = nullptr; fa.out = out; fa.acts = acts; fa.masks = masks;
      fa.in_scale = 0.5f; fa.in_shift0 = 0.5f; fa.split_row = 1 << 30; fa.NT = NT; fa.nt_stride = NT; fa.pe_tile = pe_tile; fa.nl = net == 1 ? 8 : 6;
      BwdArgs ba{}; ba.wimg = img; ba.out = out; ba.dout = in; ba.masks = masks; ba.dz = dz; ba.dz_last = dzl;
      ba.split_row = 1 << 30; ba.NT = NT; ba.nt_stride = NT; ba.pe_tile = pe_tile; ba.nl = fa.nl;
      const bool fwd = ph[0] == 'f'; const int nprod = ph[2] == '3' ? 3 : 6;
      sustain(ph, secs, 4, [&] {
        if (fwd) { MultiFwd m{}; m.n = 1; m.net[0] = net; m.a[0] = fa; af_launch_fwd_multi_bf(&m, 1, 0); }
        else { MultiBwd m{}; m.n = 1; m.net[0] = net; m.a[0] = ba; m.nprod = nprod; af_launch_bwd_multi_bf(&m, 0); }
      }, (double)NT, "row_tile");
    }
#endif
#if defined(EN_DW)
    else if (!strncmp(ph, "dw", 2)) {     // dw_hbm_<mode> / dw_l2_<mode>, mode 0 fp32 MFMA, 1 bf16x6, 2 bf16x3
      const bool cached = strstr(ph, "_l2_") != nullptr; const int mode = ph[strlen(ph) - 1] - '0';
      if (mode < 0 || mode > 2) { printf("unknown phase %s\n", ph); return 1; }
      DwArgs a{dj[cached ? 1 : 0], ds, partial, nullptr, nullptr};
      sustain(ph, secs, 4, [&] { af_launch_dw(&a, nwg, mode, 0); }, (double)NTD, "row_tile_8x8");
    }
#endif
    else { printf("unknown phase %s (not compiled into this build?)\n", ph); return 1; }
  }
  return 0;
}
