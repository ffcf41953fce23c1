// dw.hip — weight/bias gradient GEMMs  dW_l = dZ_l^T X_l,  db_l = sum_rows dZ_l  (the dW half of
// `loss.backward()`, src/stage1_neural_atlas.py:230) for every layer of every net in ONE launch.
//
// The reduction dimension is the row batch (up to 90 000 rows), the output is at most 256x256, so the
// rows are split over workgroups (split-K) by a static, cost-balanced schedule built on the host: each
// workgroup walks a short list of segments (job, row-tile range) and writes one partial block per
// segment; adam.hip sums the partials in a fixed order (deterministic, no float atomics).
//
// Per segment a workgroup keeps the WHOLE dW block in accumulators (4 waves x up to 16 tiles of 32x32),
// so both operands are streamed exactly once: 2 KB of HBM per row per 256x256 layer -> 64 FLOP/B,
// i.e. ~2.4 TB/s at the FP32-MFMA peak.  Operand tiles ([feature][32 rows], T-layout) are copied to LDS
// by global_load_lds with an XOR swizzle applied on the SOURCE address (the LDS image must stay
// lane-linear), double-buffered, one barrier per 32-row tile (16 K MFMA cycles for an 8x8 job).
// MFMA: A[m = out feature][k = row], B[k = row][n = in feature]; four consecutive k of one lane half are
// one ds_read_b128.  db falls out of the A fragments for free.
#include "af_dev.h"

#define DW_BUF 65536

// One 4 KB piece (256 lanes x 16 B) of a T-layout tile -> LDS with the 16-B-slot swizzle: physical slot c of
// feature row f holds logical row-group c ^ ((f>>1)&7).  lane_off = the per-lane swizzled source offset.
AF_DEV void dw_stage_piece(const char* src, char* dst, int it, int lane_off, int wave) {
  af_glds16(src + it * 4096 + lane_off, dst + it * 4096 + wave * 1024);
}

template <int TO, int TI, int TOW, int TIW>
AF_DEV void dw_segment(const DwJob& jb, const DwSeg& sg, float* partial, char* smem, int tid, int wave, int lane,
                       int a0, int b0, bool store_w, bool store_db) {
  const int m = lane & 31, h = lane >> 5;
  f32x16 acc[TOW][TIW];
  float dbacc[TOW];
#pragma unroll
  for (int x = 0; x < TOW; ++x) {
    dbacc[x] = 0.f;
#pragma unroll
    for (int y = 0; y < TIW; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  }
  constexpr int a_bytes = TO * 4096, b_bytes = TI * 4096;
  constexpr int NPIECE = TO + TI, PER_G = (NPIECE + 3) / 4;
  const char* Ab = (const char*)jb.A;
  const char* Bb = (const char*)jb.B;
  const size_t a_ts = (size_t)jb.a_stride * 4, b_ts = (size_t)jb.b_stride * 4;
  const int lane_off = ((((tid >> 3) << 3) + ((tid & 7) ^ ((tid >> 4) & 7))) << 4);

  auto stage_piece = [&](int t, char* buf, int i) {     // piece i of tile t: first TO pieces = A, rest = B
    if (i < TO) dw_stage_piece(Ab + t * a_ts, buf, i, lane_off, wave);
    else        dw_stage_piece(Bb + t * b_ts, buf + a_bytes, i - TO, lane_off, wave);
  };
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) stage_piece(sg.t0, smem, i);

  const int swz = (m >> 1) & 7;
  int goff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) goff[g] = m * 128 + (((2 * g + h) ^ swz) << 4);

  for (int t = sg.t0; t < sg.t1; ++t) {
    af_wait_vm0();
    __syncthreads();
    const int cur = (t - sg.t0) & 1;
    const int tn = t + 1 < sg.t1 ? t + 1 : t;           // last tile: harmless re-stage into the idle buffer
    char* nb = smem + (cur ^ 1) * DW_BUF;
    const char* abuf = smem + cur * DW_BUF + a0 * 4096;
    const char* bbuf = smem + cur * DW_BUF + a_bytes + b0 * 4096;
    f32x4 af[2][TOW], bf[2][TIW];
#pragma unroll
    for (int x = 0; x < TOW; ++x) af[0][x] = *(const f32x4*)(abuf + x * 4096 + goff[0]);
#pragma unroll
    for (int y = 0; y < TIW; ++y) bf[0][y] = *(const f32x4*)(bbuf + y * 4096 + goff[0]);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g + 1 < 4) {
#pragma unroll
        for (int x = 0; x < TOW; ++x) af[(g + 1) & 1][x] = *(const f32x4*)(abuf + x * 4096 + goff[g + 1]);
#pragma unroll
        for (int y = 0; y < TIW; ++y) bf[(g + 1) & 1][y] = *(const f32x4*)(bbuf + y * 4096 + goff[g + 1]);
      }
#pragma unroll
      for (int i = g * PER_G; i < (g + 1) * PER_G && i < NPIECE; ++i) stage_piece(tn, nb, i);   // next tile, in this group's MFMA shadow
#pragma unroll
      for (int x = 0; x < TOW; ++x) dbacc[x] += (af[g & 1][x][0] + af[g & 1][x][1]) + (af[g & 1][x][2] + af[g & 1][x][3]);
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int x = 0; x < TOW; ++x)
#pragma unroll
          for (int y = 0; y < TIW; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][x][p], bf[g & 1][y][p], acc[x][y], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  af_wait_vm0();
  __syncthreads();   // everyone done reading LDS (and the trailing re-stage landed) before the next segment restages

  float* blk = partial + jb.part_off + (size_t)sg.slot * jb.part_blk;
  constexpr int pld = TI * 32;
  const auto rblk = af_rsrc_uniform(blk, jb.part_blk * 4);
  if (store_w) {
    const int voff = ((a0 * 32 + 4 * h) * pld + b0 * 32 + m) * 4;
#pragma unroll
    for (int x = 0; x < TOW; ++x)
#pragma unroll
      for (int y = 0; y < TIW; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          af_bs32(acc[x][y][r], rblk, voff, ((x * 32 + (r & 3) + 8 * (r >> 2)) * pld + y * 32) * 4);
  }
#pragma unroll
  for (int x = 0; x < TOW; ++x) {
    const float tot = dbacc[x] + __shfl_xor(dbacc[x], 32);
    if (store_db && h == 0) af_bs32(tot, rblk, (TO * 32 * pld + a0 * 32 + m) * 4, x * 128);
  }
}

__global__ __launch_bounds__(256, 1) void k_dw(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const DwSeg* segs = a.segs + (size_t)blockIdx.x * DW_MAXSEG;
  if (a.wg_clock && tid == 0) a.wg_clock[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < DW_MAXSEG; ++s) {
    const DwSeg sg = segs[s];
    if (sg.job < 0) break;
    const DwJob jb = a.jobs[sg.job];
    switch (jb.shape) {
      case DW_8x8: dw_segment<8, 8, 2, 8>(jb, sg, a.partial, smem, tid, wave, lane, 2 * wave, 0, true, true); break;
      case DW_8x2: dw_segment<8, 2, 2, 2>(jb, sg, a.partial, smem, tid, wave, lane, 2 * wave, 0, true, true); break;
      case DW_8x1: dw_segment<8, 1, 2, 1>(jb, sg, a.partial, smem, tid, wave, lane, 2 * wave, 0, true, true); break;
      case DW_1x8: dw_segment<1, 8, 1, 2>(jb, sg, a.partial, smem, tid, wave, lane, 0, 2 * wave, true, wave == 0); break;
      case DW_1x2: dw_segment<1, 2, 1, 1>(jb, sg, a.partial, smem, tid, wave, lane, 0, wave & 1, wave < 2, wave == 0); break;
      default: break;
    }
  }
  if (a.wg_clock && tid == 0) a.wg_clock[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
}

extern "C" int af_launch_dw(const DwArgs* a, int nwg, hipStream_t s) {
  hipLaunchKernelGGL(k_dw, dim3(nwg), dim3(256), 2 * DW_BUF, s, *a);
  return (int)hipGetLastError();
}
extern "C" int af_dw_init() {
  return (int)hipFuncSetAttribute((const void*)k_dw, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DW_BUF);
}
