// tools/dwbench.hip — k_dw micro-harness (not product code): one 8x8 job, every workgroup one segment of NT/256 row tiles.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDW_ABL=<bits> tools/dwbench.hip -o tools/bin/dwb_<bits>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <algorithm>
#include "../all-in-one-deflicker_amd/csrc/dw.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int per = argc > 1 ? atoi(argv[1]) : 66;       // row tiles per workgroup
  const int nwg = 256, NT = per * nwg, reps = argc > 3 ? atoi(argv[3]) : 10;
  const int cached = argc > 2 ? atoi(argv[2]) : 0;      // 1: every tile index maps to tile 0 (operands L2-resident): the kernel without HBM
  af_dw_init();
  float *A, *B, *partial;
  CK(hipMalloc(&A, (size_t)NT * AF_TILE_F * 4)); CK(hipMalloc(&B, (size_t)NT * AF_TILE_F * 4));
  CK(hipMalloc(&partial, (size_t)nwg * (65536 + 256) * 4));
  std::vector<float> h((size_t)1 << 22); for (auto& x : h) x = rand() / (float)RAND_MAX - 0.5f;
  for (size_t off = 0; off < (size_t)NT * AF_TILE_F; off += h.size()) {
    const size_t n = std::min(h.size(), (size_t)NT * AF_TILE_F - off);
    CK(hipMemcpy(A + off, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B + off, h.data(), n * 4, hipMemcpyHostToDevice));
  }
  DwJob j{}; j.A = A; j.B = B; j.a_stride = cached ? 0 : AF_TILE_F; j.b_stride = cached ? 0 : AF_TILE_F; j.shape = DW_8x8; j.part_off = 0; j.part_blk = 65536 + 256;
  std::vector<DwSeg> segs((size_t)nwg * DW_MAXSEG, DwSeg{-1, 0, 0, 0});
  for (int w = 0; w < nwg; ++w) segs[(size_t)w * DW_MAXSEG] = DwSeg{0, w * per, (w + 1) * per, w};
  DwJob* dj; DwSeg* ds; CK(hipMalloc(&dj, sizeof j)); CK(hipMalloc(&ds, segs.size() * sizeof(DwSeg)));
  CK(hipMemcpy(dj, &j, sizeof j, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, segs.data(), segs.size() * sizeof(DwSeg), hipMemcpyHostToDevice));
  DwArgs a{dj, ds, partial, nullptr, nullptr};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {
    for (int r = 0; r < 2; ++r) af_launch_dw(&a, nwg, mode, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) af_launch_dw(&a, nwg, mode, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double fl = (double)NT * 32 * 2.0 * 256 * 256, by = (double)NT * 2 * AF_TILE_F * 4;
#ifdef DW_CLK     // core clock ticks each workgroup spent (s_memtime): ticks / event time = the shader clock under this load
    if (mode >= 1) {
      unsigned long long* clk; CK(hipMalloc(&clk, nwg * 32));
      DwArgs ac = a; ac.wg_clock = clk;
      for (int r = 0; r < 20; ++r) af_launch_dw(&ac, nwg, mode, 0);           // a sustained run: the last launch's stamps are read
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> hc(nwg * 4); CK(hipMemcpy(hc.data(), clk, nwg * 32, hipMemcpyDeviceToHost));
      double tk = 0, rt = 0, first = 1e30, last = 0;
      for (int w = 0; w < nwg; ++w) { tk += (double)(hc[4 * w + 1] - hc[4 * w]); rt += (double)(hc[4 * w + 3] - hc[4 * w + 2]); first = std::min(first, (double)hc[4 * w + 2]); last = std::max(last, (double)hc[4 * w + 3]); }
      tk /= nwg; rt /= nwg;
      printf("mean ticks per workgroup %.0f = %.0f per stage; ticks / launch time %.0f MHz; ticks / the workgroup's own s_memrealtime span %.0f MHz "
             "(workgroup span %.1f us mean, first start to last end %.1f us, launch %.1f us)\n", tk, tk / (2 * per), tk / (ms * 1000), tk / (rt / 100.0), rt / 100.0, (last - first) / 100.0, ms * 1000);
    }
#endif
    printf("DW_ABL=%d DW_SLOT=%d%s mode %d (%s) %d tiles/WG: %.4f ms  %.1f TF-equivalent  %.2f TB/s\n", DW_ABL, DW_SLOT, cached ? " L2-resident" : "", mode, mode == 2 ? "bf16x3" : (mode ? "bf16x6" : "fp32 MFMA"), per, ms, fl / ms / 1e9, by / ms / 1e9);
    {   // FNV-1a over the bits of every partial block: the slotted stage (DW_SLOT=1) keeps the compiler-scheduled kernel's MFMA order, so the two
        // builds must print the same hash for mode 1; against mode 0 (fp32 MFMA) the blocks agree to fp32 round-off (max relative distance printed)
      std::vector<uint32_t> hp((size_t)nwg * (65536 + 256)); CK(hipMemcpy(hp.data(), partial, hp.size() * 4, hipMemcpyDeviceToHost));
      unsigned long long hsh = 1469598103934665603ull; for (uint32_t w : hp) { hsh ^= w; hsh *= 1099511628211ull; }
      static std::vector<float> ref;
      double worst = 0, scale = 0;
      if (mode == 0) ref.assign((float*)hp.data(), (float*)hp.data() + hp.size());
      else { for (size_t i = 0; i < hp.size(); ++i) { const float v = ((float*)hp.data())[i]; worst = std::max(worst, (double)fabsf(v - ref[i])); scale = std::max(scale, (double)fabsf(ref[i])); } }
      printf("   partial blocks: hash %016llx%s", hsh, mode ? "" : "\n");
      if (mode) printf("  max |x - fp32 MFMA| / max |fp32 MFMA| = %.3g\n", worst / scale);
    }
  }
  return 0;
}
