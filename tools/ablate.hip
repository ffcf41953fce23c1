// tools/ablate.hip — kernel ablation harness (not product code): times k_mlp_fwd / k_mlp_bwd / k_dw variants
// on synthetic buffers.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAF_ABL=<bits> tools/ablate.hip -o ablate_<bits>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../all-in-one-deflicker_amd/csrc/mlp.hip"
#include "experiments/mlp16.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int NT = argc > 1 ? atoi(argv[1]) : 2813;
  const int reps = 20;
  af_mlp_init(); af_mlp16_init();
  // mapping net: chunks [8K, 16 x 64K, 32K]
  std::vector<AfChunk> ch; uint32_t off = 0;
  ch.push_back({off, 16384}); off += 16384;
  for (int i = 0; i < 16; ++i) { ch.push_back({off, 65536}); off += 65536; }
  ch.push_back({off, 32768}); off += 32768;
  float *img, *bias, *in, *out, *acts, *dz, *dzl; uint32_t* masks; AfChunk* dch;
  CK(hipMalloc(&img, off + 65536)); CK(hipMemset(img, 0, off + 65536));
  CK(hipMalloc(&bias, 6 * 256 * 4)); CK(hipMemset(bias, 0, 6 * 256 * 4));
  CK(hipMalloc(&in, (size_t)NT * 32 * 16)); CK(hipMemset(in, 0, (size_t)NT * 32 * 16));
  CK(hipMalloc(&out, (size_t)NT * 32 * 16)); CK(hipMemset(out, 0, (size_t)NT * 32 * 16));
  CK(hipMalloc(&acts, (size_t)5 * NT * 32768)); CK(hipMalloc(&dz, (size_t)5 * NT * 32768));
  CK(hipMalloc(&dzl, (size_t)NT * 4096)); CK(hipMalloc(&masks, (size_t)5 * NT * 1024)); CK(hipMemset(masks, 0xff, (size_t)5 * NT * 1024));
  CK(hipMalloc(&dch, ch.size() * sizeof(AfChunk))); CK(hipMemcpy(dch, ch.data(), ch.size() * sizeof(AfChunk), hipMemcpyHostToDevice));
  std::vector<float> w(off / 4); for (auto& x : w) x = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  CK(hipMemcpy(img, w.data(), off, hipMemcpyHostToDevice));
  FwdArgs fa{}; fa.wimg = img; fa.chunks = dch; fa.in1 = nullptr; fa.bias = bias; fa.in = in; fa.out = out; fa.acts = acts; fa.masks = masks;
  fa.in_scale = 0.5f; fa.in_shift0 = 0.5f; fa.split_row = 1 << 30; fa.NT = NT; fa.nt_stride = NT; fa.nchunks = (int)ch.size();
  // backward chunk order: [8K, 16 x 64K]
  std::vector<AfChunk> bch; off = 0; bch.push_back({off, 16384}); off += 16384;
  for (int i = 0; i < 16; ++i) { bch.push_back({off, 65536}); off += 65536; }
  AfChunk* dbch; CK(hipMalloc(&dbch, bch.size() * sizeof(AfChunk))); CK(hipMemcpy(dbch, bch.data(), bch.size() * sizeof(AfChunk), hipMemcpyHostToDevice));
  BwdArgs ba{}; ba.wimg = img; ba.chunks = dbch; ba.out = out; ba.dout = in; ba.masks = masks; ba.dz = dz; ba.dz_last = dzl;
  ba.split_row = 1 << 30; ba.NT = NT; ba.nt_stride = NT; ba.nchunks = (int)bch.size();
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int which = 0; which < 4; ++which) {
    auto go = [&]() {
      if (which == 0) { MultiFwd m{}; m.n = 1; m.net[0] = AF_NET_MAP1; m.a[0] = fa; af_launch_fwd_multi(&m, 1, 0); }
      else if (which == 1) { MultiBwd m{}; m.n = 1; m.net[0] = AF_NET_MAP1; m.a[0] = ba; af_launch_bwd_multi(&m, 0); }
      else if (which == 2) af_launch_fwd16(AF_NET_MAP1, 1, &fa, 0); else af_launch_bwd16(AF_NET_MAP1, &ba, 0);
    };
    for (int r = 0; r < 3; ++r) go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) go();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double fl = (double)NT * 32 * ((which & 1) == 0 ? 526848.0 : 525312.0);
    printf("ABL=%d %s NT=%d: %.4f ms  %.1f TF (%.1f%% of 157.3)\n", AF_ABL, which == 0 ? "fwd_map32" : which == 1 ? "bwd_map32" : which == 2 ? "fwd_map16" : "bwd_map16", NT, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
  }
  return 0;
}
