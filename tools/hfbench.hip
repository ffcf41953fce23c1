// tools/hfbench.hip — the f16x3 chains (mlphf.hip) next to the bf16x6 chains (mlpbf.hip) on synthetic buffers (not product code):
// time per 128-row task of every net in both directions, and — with the chain's s_memtime marks (-DAF_HF_CLK is set here) — where the
// ticks of an f16x3 task go: prologue, layer-0 block, every hidden block, every epilogue, the output stage.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hfbench.hip -o tools/bin/hfbench        Run: hfbench [row tiles, default 1024 = one round of the chip]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define AF_HF_CLK 1
#include "../all-in-one-deflicker_amd/csrc/mlpbf.hip"
#include "../all-in-one-deflicker_amd/csrc/mlphf.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int NT = argc > 1 ? atoi(argv[1]) : 1024;
  const int reps = 20;
  CK((hipError_t)af_mlp_bf_init()); CK((hipError_t)af_mlp_hf_init());
  const size_t img_bytes = (size_t)16 << 20;
  float *img, *bias, *in, *out, *acts, *dz, *dzl; uint32_t* masks;
  CK(hipMalloc(&img, img_bytes)); CK(hipMalloc(&bias, 8 * 256 * 4)); CK(hipMemset(bias, 0, 8 * 256 * 4));
  CK(hipMalloc(&in, (size_t)NT * 32 * 16)); CK(hipMalloc(&out, (size_t)NT * 32 * 16)); CK(hipMemset(out, 0, (size_t)NT * 32 * 16));
  CK(hipMalloc(&acts, (size_t)7 * NT * 32768)); CK(hipMalloc(&dz, (size_t)7 * NT * 32768));
  CK(hipMalloc(&dzl, (size_t)NT * 4096)); CK(hipMalloc(&masks, (size_t)7 * NT * 1024)); CK(hipMemset(masks, 0xff, (size_t)7 * NT * 1024));
  float* pe_tile; CK(hipMalloc(&pe_tile, (size_t)NT * 8192)); CK(hipMemset(pe_tile, 0, (size_t)NT * 8192));
  // weights: small fp32 values; read as bf16 / fp16 images the same bytes are finite, small numbers in every 16-bit half
  std::vector<uint32_t> w(img_bytes / 4); for (auto& x : w) x = 0x2c002c00u + ((uint32_t)rand() & 0x03ff03ffu);
  CK(hipMemcpy(img, w.data(), img_bytes, hipMemcpyHostToDevice));
  std::vector<float> hin((size_t)NT * 32 * 4); for (auto& x : hin) x = rand() / (float)RAND_MAX * 2.f - 1.f;
  CK(hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* nn[4] = {"mapping1", "atlas", "mapping2", "alpha"};
  const int nwg = (NT + 3) / 4;
  unsigned long long* clk; CK(hipMalloc(&clk, (size_t)nwg * 32 * 8));
  for (int net = 0; net < 4; ++net) {
    FwdArgs fa{}; fa.wimg = img; fa.bias = bias; fa.in = in; fa.in1 = nullptr; fa.out = out; fa.acts = acts; fa.masks = masks;
    fa.in_scale = 0.5f; fa.in_shift0 = 0.5f; fa.split_row = 1 << 30; fa.NT = NT; fa.nt_stride = NT; fa.pe_tile = pe_tile; fa.nl = net == 1 || net == 3 ? 8 : (net == 2 ? 4 : 6);
    BwdArgs ba{}; ba.wimg = img; ba.out = out; ba.dout = in; ba.masks = masks; ba.dz = dz; ba.dz_last = dzl;
    ba.split_row = 1 << 30; ba.NT = NT; ba.nt_stride = NT; ba.pe_tile = pe_tile; ba.nl = fa.nl; ba.din0 = acts; ba.din1 = acts; ba.nrows = 0;
    for (int dir = 0; dir < 2; ++dir) {
      float t[2];
      for (int hf = 0; hf < 2; ++hf) {
        auto go = [&]() {
          if (dir == 0) { MultiFwd m{}; m.n = 1; m.net[0] = net; m.a[0] = fa; if (hf) af_launch_fwd_multi_hf(&m, 1, 0); else af_launch_fwd_multi_bf(&m, 1, 0); }
          else { MultiBwd m{}; m.n = 1; m.net[0] = net; m.a[0] = ba; m.nprod = 6; if (hf) af_launch_bwd_multi_hf(&m, 0); else af_launch_bwd_multi_bf(&m, 0); }
        };
        for (int r = 0; r < 3; ++r) go();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) go();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t[hf] = ms / reps * 1000;
        if (hf) {
          CK(hipMemset(clk, 0, (size_t)nwg * 32 * 8));
          CK(hipMemcpyToSymbol(HIP_SYMBOL(g_hf_clk), &clk, sizeof clk));
          go(); CK(hipDeviceSynchronize());
          unsigned long long* none = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_hf_clk), &none, sizeof none));
        }
      }
      printf("%-8s %s, NT=%d (%.2f rounds of 1024 SIMDs): bf16x6 %.1f us   f16x3 %.1f us   (x%.2f)\n", nn[net], dir ? "bwd" : "fwd", NT, NT / 1024.0, t[0], t[1], t[0] / t[1]);
      std::vector<unsigned long long> hc((size_t)nwg * 32); CK(hipMemcpy(hc.data(), clk, (size_t)nwg * 32 * 8, hipMemcpyDeviceToHost));
      const int nl = fa.nl;
      double seg[32] = {0}; int cnt = 0;
      for (int wgi = 0; wgi < nwg; ++wgi) {
        const unsigned long long* c = &hc[(size_t)wgi * 32];
        if (!c[0] || !c[30]) continue;
        ++cnt;
        seg[0] += (double)(c[1] - c[0]); seg[1] += (double)(c[2] - c[1]); seg[2] += (double)(c[3] - c[2]); seg[3] += (double)(c[4] - c[3]); seg[4] += (double)(c[5] - c[4]);
        for (int l = 1; l <= nl - 2; ++l) { seg[5] += (double)(c[4 + 2 * l] - c[3 + 2 * l]); seg[6] += (double)(c[5 + 2 * l] - c[4 + 2 * l]); }
        seg[7] += (double)(c[30] - c[5 + 2 * (nl - 2)]);
        seg[8] += (double)(c[30] - c[0]);
      }
      if (cnt) printf("         f16x3 ticks per workgroup (mean of %d): input stage %.0f | first publish %.0f | first fp32 block %.0f | its epilogue %.0f | publish + lead %.0f | "
                      "hidden blocks %.0f (%d x %.0f; MFMA issue alone %d each) | their epilogues %.0f (%.0f each) | last stage %.0f | whole task %.0f\n", cnt,
                      seg[0] / cnt, seg[1] / cnt, seg[2] / cnt, seg[3] / cnt, seg[4] / cnt, seg[5] / cnt, nl - 2, seg[5] / cnt / (nl - 2), 16 * 24 * 32, seg[6] / cnt, seg[6] / cnt / (nl - 2), seg[7] / cnt, seg[8] / cnt);
    }
  }
  return 0;
}
