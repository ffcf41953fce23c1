// elem.hip — the non-GEMM kernels of the atlas-fit step: pixel-record table build, batch sampling +
// gather + coordinate normalisation, the loss stack with its seed gradients, the pre-train loss,
// split-K reduction + Adam, and the render / PSNR helpers.  All fp32, no fast-math (IEEE divide/sqrt,
// accurate sinf/cosf/tanhf) — parity with the reference needs it (SURVEY.md §7 "Hard parts").
#include "af_dev.h"
#include "elem.h"

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (counter-based; one 128-bit block per (iteration, sample))
AF_DEV void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
AF_DEV uint64_t af_rand_below(uint64_t seed, uint32_t iter, uint32_t n, uint32_t stream, uint64_t bound) {
  uint32_t o[4];
  philox4x32(n, iter, stream, 0x41544c53u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  const uint64_t r = ((uint64_t)o[1] << 32) | o[0];
  return __umul64hi(r, bound);
}

// ---------------------------------------------------------------------------------------------
// Pixel-record table: 64-B records {rgb, d/dx rgb, d/dy rgb, fwd flow, bwd flow, fwd mask, bwd mask, fg mask}
// from the reference's dense layouts (unwrap_utils.py:113-122,132-133).
// record k = f*resy*resx + y*resx + x  (== column k of get_tuples' jif_all, unwrap_utils.py:166-173).
__global__ void k_pack_table(PackArgs a) {
  const size_t P2 = (size_t)a.resx * a.resy;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = idx < P2 * a.F;
  const unsigned long long vf = __ballot(in_range && a.mask_f[idx] != 0.f), vb = __ballot(in_range && a.mask_b[idx] != 0.f);   // mask layout (pix, f) == idx
  if (a.nvalid && (threadIdx.x & 63) == 0) {      // 64 counter pairs, one per residue of the wave index: no hot address
    unsigned long long* c = a.nvalid + 2 * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 63);
    if (vf) atomicAdd(c + 0, (unsigned long long)__popcll(vf));
    if (vb) atomicAdd(c + 1, (unsigned long long)__popcll(vb));
  }
  if (!in_range) return;
  const int f = (int)(idx % a.F);
  const size_t pix = idx / a.F;
  const int x = (int)(pix % a.resx), y = (int)(pix / a.resx);
  const int F = a.F;
  float rec[AF_REC_F];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = a.frames[(pix * 3 + c) * F + f];
    rec[REC_RGB + c] = v;
    rec[REC_DX + c] = (x + 1 < a.resx) ? a.frames[((pix + 1) * 3 + c) * F + f] - v : 0.f;
    rec[REC_DY + c] = (y + 1 < a.resy) ? a.frames[((pix + a.resx) * 3 + c) * F + f] - v : 0.f;
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    rec[REC_FF + c] = a.flow_f[(pix * 2 + c) * F + f];
    rec[REC_FB + c] = a.flow_b[(pix * 2 + c) * F + f];
  }
  rec[REC_MF] = a.mask_f[pix * F + f];
  rec[REC_MB] = a.mask_b[pix * F + f];
  rec[REC_FG] = a.mask_fg ? a.mask_fg[pix * F + f] : 0.f;
  f32x4* dst = (f32x4*)(a.table + ((size_t)f * P2 + pix) * AF_REC_F);
#pragma unroll
  for (int q = 0; q < 4; ++q) { f32x4 v = {rec[q * 4], rec[q * 4 + 1], rec[q * 4 + 2], rec[q * 4 + 3]}; dst[q] = v; }
}

// ---------------------------------------------------------------------------------------------
// Input builder kernels (one thread per destination pixel), in OpenCV's arithmetic (published algorithm of
// imgproc/resize.cpp and imgwarp.cpp; the tests hold an independent CPU restatement with hand-computed vectors):
//  * cv2.resize INTER_LINEAR: per axis  f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s (float), taps clamp at the
//    edges with weight 1; the two coefficients are the floats 1.f - f and f (alpha type float even for CV_64F images);
//    horizontal pass, then vertical, in double for CV_64F sources (8-bit frames / masks: `astype(float64) / 255`,
//    unwrap_utils.py:128,68) and in float for CV_32F sources (flows, :35);
//  * cv2.remap INTER_LINEAR on a CV_32FC2 map (:22): position -> fixed point with INTER_BITS = 5 (cvRound(x * 32), round
//    half to even), i.e. quantised to 1/32 px; table weights (1-fy)(1-fx) ...; float accumulation left to right;
//    constant-0 border tap by tap.
template <class WT>
AF_DEV float resize_tap(const ResizeArgs& a, int y0, int y1, int x0, int x1, float a0, float a1, float b0, float b1, int c) {
#pragma clang fp contract(off)
  auto tap = [&](int yy, int xx) -> WT {
    const size_t i = ((size_t)yy * a.sw + xx) * a.ch + c;
    return a.src_u8 ? (WT)((double)((const unsigned char*)a.src)[i] / 255.0) : (WT)((const float*)a.src)[i];
  };
  const WT t0 = tap(y0, x0) * (WT)a0 + tap(y0, x1) * (WT)a1;
  const WT t1 = tap(y1, x0) * (WT)a0 + tap(y1, x1) * (WT)a1;
  return (float)(t0 * (WT)b0 + t1 * (WT)b1);
}
AF_DEV void resize_coeff(int d, int src, int dst, int& s0, int& s1, float& c0, float& c1) {
#pragma clang fp contract(off)
  const double scale = (double)src / (double)dst;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { s = 0; f = 0.f; }
  if (s >= src - 1) { s = src - 1; f = 0.f; }
  s0 = s; s1 = min(s + 1, src - 1); c0 = 1.f - f; c1 = f;
}
__global__ void k_resize_bilinear(ResizeArgs a) {
#pragma clang fp contract(off)      // same roundings as the restatements (no FMA contraction)
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)a.dh * a.dw) return;
  const int y = (int)(idx / a.dw), x = (int)(idx - (long long)y * a.dw);
  const bool same = a.sh == a.dh && a.sw == a.dw;          // identity: cv::resize copies
  int x0, x1, y0, y1; float a0, a1, b0, b1;
  resize_coeff(x, a.sw, a.dw, x0, x1, a0, a1);
  resize_coeff(y, a.sh, a.dh, y0, y1, b0, b1);
  for (int c = 0; c < a.ch; ++c) {
    float o;
    if (same) {
      const size_t i = ((size_t)y * a.sw + x) * a.ch + c;
      o = a.src_u8 ? (float)((double)((const unsigned char*)a.src)[i] / 255.0) : ((const float*)a.src)[i];
    } else {
      o = a.src_u8 ? resize_tap<double>(a, y0, y1, x0, x1, a0, a1, b0, b1, c) : resize_tap<float>(a, y0, y1, x0, x1, a0, a1, b0, b1, c);
    }
    if (c == 0) o = o * (float)a.scale0; else if (c == 1) o = o * (float)a.scale1;     // fp32, like `flow[:, :, c] *= s`
    a.dst[idx * a.pix_stride + (long long)c * a.ch_stride + a.offset] = o;
  }
}

__global__ void k_flow_consistency(ConsistencyArgs a) {
#pragma clang fp contract(off)
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)a.h * a.w) return;
  const int y = (int)(idx / a.w), x = (int)(idx - (long long)y * a.w);
  const float u = a.f12[idx * 2], v = a.f12[idx * 2 + 1];
  const float mx = u + (float)x, my = v + (float)y;                      // sampling position in the other frame (`flow += arange`)
  const long long qx = (long long)rintf(mx * 32.f), qy = (long long)rintf(my * 32.f);   // cvRound(pos * INTER_TAB_SIZE), half to even
  const long long x0 = qx >> 5, y0 = qy >> 5;
  const float fx = (float)(qx & 31) / 32.f, fy = (float)(qy & 31) / 32.f;
  const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
  float acc[2];
  for (int c = 0; c < 2; ++c) {
    auto tap = [&](long long yy, long long xx) -> float {              // constant-0 border (cv2.remap default)
      return (xx >= 0 && xx < a.w && yy >= 0 && yy < a.h) ? a.f21[(yy * a.w + xx) * 2 + c] : 0.f;
    };
    acc[c] = ((tap(y0, x0) * w0 + tap(y0, x0 + 1) * w1) + tap(y0 + 1, x0) * w2) + tap(y0 + 1, x0 + 1) * w3;
  }
  const float du = u + acc[0], dv = v + acc[1];
  const float nrm = sqrtf(du * du + dv * dv);
  a.out[idx * a.pix_stride + a.offset] = a.thresh > 0.f ? (nrm < a.thresh ? 1.f : 0.f) : nrm;
}

// ---------------------------------------------------------------------------------------------
// Batch preparation (stage1_neural_atlas.py:159-171; loss_utils.py:137-151, 230-233, 326-351).
// Row segments of the mapping batch (N rows each): 0 centre, 1 (x,y+1), 2 (x+1,y), 3 (x,y-d), 4 (x-d,y) and, while the
// global rigidity term is on, 5 (x,y-D), 6 (x-D,y); behind them (flow_base = 5N or 7N) the flow matches of the batch,
// VALID ones only and compacted like the reference's torch.where (loss_utils.py:328-335): sample-major, a sample's
// forward match in front of its backward match.  a.nseg = 7 / 9 is the segment count of the un-compacted maximum.
AF_DEV void put_row_to(float* coords, float* x0_tile, size_t r, float x, float y, float t) {
  f32x4 v = {x, y, t, 0.f};
  *(f32x4*)(coords + r * 4) = v;
  if (x0_tile) {
    float* tl = x0_tile + (r >> 5) * 1024 + (r & 31);
    tl[0] = x; tl[32] = y; tl[64] = t;
  }
}
// a row shared by both mapping nets (and, for arow >= 0, by the alpha net as its row arow)
AF_DEV void put_row_at(const PrepArgs& a, size_t r, long long arow, float x, float y, float t) {
  put_row_to(a.coords, a.x0_tile, r, x, y, t);
  if (a.coords2) {
    put_row_to(a.coords2, a.x0_tile2, r, x, y, t);
    if (arow >= 0) put_row_to(a.coordsA, nullptr, (size_t)arow, x, y, t);
  }
}
AF_DEV void put_row(const PrepArgs& a, int seg, int aseg, int n, float x, float y, float t) {
  put_row_at(a, (size_t)seg * a.N + n, aseg >= 0 ? (long long)aseg * a.N + n : -1, x, y, t);
}

__global__ __launch_bounds__(256) void k_prep(PrepArgs a) {
  __shared__ int wsum[4];
  __shared__ int red_i[4];
  // Virtual block id from an atomic ticket, not blockIdx.x: the look-back below waits for every block with a SMALLER id, and
  // a block that holds a ticket has started, so all of its predecessors are resident or finished whatever order the
  // hardware dispatched the grid in (HIP guarantees none) and however many blocks the launch has.  The ticket counter is
  // monotonic over the handle's launches (never reset); the host passes the count at the start of this launch.
  if (threadIdx.x == 0) red_i[0] = (int)(atomicAdd(a.ticket, 1ull) - a.ticket_base);
  __syncthreads();
  const int vblk = red_i[0];
  __syncthreads();
  const int n = vblk * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int vf = 0, vb = 0;
  int x = 0, y = 0, f = 0;
  float ffu = 0.f, ffv = 0.f, fbu = 0.f, fbv = 0.f;
  const float hm = a.half_main, hg = a.half_grad, hf = a.half_frames;
  if (n < a.N) {
    const uint64_t P2 = (uint64_t)a.resx * a.resy;
    const uint64_t k = a.inds ? (uint64_t)a.inds[n] : af_rand_below(a.seed, a.iter, (uint32_t)n, 0u, P2 * a.F);
    f = (int)(k / P2);
    const uint64_t rem = k - (uint64_t)f * P2;
    y = (int)(rem / a.resx); x = (int)(rem - (uint64_t)y * a.resx);
    const f32x4* rp = (const f32x4*)(a.table + k * AF_REC_F);
    f32x4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
    f32x4* sp = (f32x4*)(a.samples + (size_t)n * AF_REC_F);
    sp[0] = r0; sp[1] = r1; sp[2] = r2; sp[3] = r3;
    ffu = r2[1]; ffv = r2[2]; fbu = r2[3]; fbv = r3[0];
    const float mf = r3[1], mb = r3[2];
    const float xc = (float)x / hm - 1.f, yc = (float)y / hm - 1.f, tc = (float)f / hf - 1.f;
    put_row(a, 0, 0, n, xc, yc, tc);
    put_row(a, 1, 1, n, (float)x / hg - 1.f, (float)(y + 1) / hg - 1.f, tc);
    put_row(a, 2, 2, n, (float)(x + 1) / hg - 1.f, (float)y / hg - 1.f, tc);
    put_row(a, 3, -1, n, xc, (float)(y - a.d_local) / hm - 1.f, tc);
    put_row(a, 4, -1, n, (float)(x - a.d_local) / hm - 1.f, yc, tc);
    vf = mf != 0.f; vb = mb != 0.f;
    if (a.nseg > 7) {
      put_row_to(a.coords, a.x0_tile, (size_t)5 * a.N + n, xc, (float)(y - a.d_global) / hm - 1.f, tc);
      put_row_to(a.coords, a.x0_tile, (size_t)6 * a.N + n, (float)(x - a.d_global) / hm - 1.f, yc, tc);
      if (a.coords2) {
        put_row_to(a.coords2, a.x0_tile2, (size_t)5 * a.N + n, xc, (float)(y - a.d_global2) / hm - 1.f, tc);
        put_row_to(a.coords2, a.x0_tile2, (size_t)6 * a.N + n, (float)(x - a.d_global2) / hm - 1.f, yc, tc);
      }
    }
  }
  // ---- rank of this sample's matches among the valid matches of the batch
  const unsigned long long bf = __ballot(vf), bb = __ballot(vb);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int in_wave = __popcll(bf & lt) + __popcll(bb & lt);
  if (lane == 0) {
    wsum[wave] = __popcll(bf) + __popcll(bb);
    if (bf) atomicAdd(a.counts + 0, __popcll(bf));
    if (bb) atomicAdd(a.counts + 1, __popcll(bb));
  }
  __syncthreads();
  int in_block = in_wave;
  for (int w = 0; w < wave; ++w) in_block += wsum[w];
  const int btot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  if (threadIdx.x == 0)
    __hip_atomic_store(a.scan + vblk, ((unsigned long long)a.epoch << 32) | (unsigned)btot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  int pre = 0;      // look back (one wave polls): every block with a smaller ticket has started, see above
  if (wave == 0) {
    for (int b = lane; b < vblk; b += 64) {
      unsigned long long v;
      do { v = __hip_atomic_load(a.scan + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while ((uint32_t)(v >> 32) != a.epoch);
      pre += (int)(uint32_t)v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o);
    if (lane == 0) red_i[0] = pre;
  }
  __syncthreads();
  pre = red_i[0];
  if (vblk == (int)gridDim.x - 1 && threadIdx.x == 0) *a.live = pre + btot;
  if (n < a.N) {
    const int rank_f = pre + in_block, rank_b = rank_f + vf;
    const size_t base = (size_t)(a.nseg - 2) * a.N, abase = (size_t)3 * a.N;
    a.flow_rank[2 * n] = vf ? rank_f : -1;
    a.flow_rank[2 * n + 1] = vb ? rank_b : -1;
    if (vf) put_row_at(a, base + rank_f, (long long)(abase + rank_f), ((float)x + ffu) / hm - 1.f, ((float)y + ffv) / hm - 1.f, (float)(f + 1) / hf - 1.f);
    if (vb) put_row_at(a, base + rank_b, (long long)(abase + rank_b), ((float)x + fbu) / hm - 1.f, ((float)y + fbv) / hm - 1.f, (float)(f - 1) / hf - 1.f);
  }
}

// ---------------------------------------------------------------------------------------------
AF_DEV float block_sum(float v, float* red /*[4]*/) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

struct Rig { float loss, du, dv, du_xm, du_ym, dv_xm, dv_ym; };
// Rigidity term and its gradient for one sample (loss_utils.py:239-278).
AF_DEV Rig rigidity(float u, float v, float u_ym, float v_ym, float u_xm, float v_xm, float L, float s, float d, float w) {
  const float j00 = ((u - u_xm) * L / 2.f) / s / d, j01 = ((u - u_ym) * L / 2.f) / s / d;
  const float j10 = ((v - v_xm) * L / 2.f) / s / d, j11 = ((v - v_ym) * L / 2.f) / s / d;
  const float g00 = j00 * j00 + j10 * j10, g01 = j00 * j01 + j10 * j11, g11 = j01 * j01 + j11 * j11;
  const float A = g00 + 0.001f, D = g11 + 0.001f, B = g01;
  const float det = A * D - B * B;
  const float i00 = D / det, i01 = -B / det, i11 = A / det;
  const float ng = sqrtf(g00 * g00 + 2.f * g01 * g01 + g11 * g11);
  const float ni = sqrtf(i00 * i00 + 2.f * i01 * i01 + i11 * i11);
  Rig r; r.loss = ng + ni;
  // S = G/|G| - B^3/|B|  (B = inverse, symmetric)
  const float b2_00 = i00 * i00 + i01 * i01, b2_01 = i00 * i01 + i01 * i11, b2_11 = i01 * i01 + i11 * i11;
  const float b3_00 = b2_00 * i00 + b2_01 * i01, b3_01 = b2_00 * i01 + b2_01 * i11, b3_11 = b2_01 * i01 + b2_11 * i11;
  const float rg = ng > 0.f ? 1.f / ng : 0.f, ri = ni > 0.f ? 1.f / ni : 0.f;
  const float s00 = g00 * rg - b3_00 * ri, s01 = g01 * rg - b3_01 * ri, s11 = g11 * rg - b3_11 * ri;
  // dl/dJ = 2 J S ; D = k * dl/dJ * w, k = (L/2)/(s d)
  const float k = (L / 2.f) / s / d * w * 2.f;
  const float d00 = k * (j00 * s00 + j01 * s01), d01 = k * (j00 * s01 + j01 * s11);
  const float d10 = k * (j10 * s00 + j11 * s01), d11 = k * (j10 * s01 + j11 * s11);
  r.du = d00 + d01; r.dv = d10 + d11; r.du_xm = -d00; r.du_ym = -d01; r.dv_xm = -d10; r.dv_ym = -d11;
  return r;
}

AF_DEV void put_d2(float* dout, size_t row, float a, float b) { f32x4 v = {a, b, 0.f, 0.f}; *(f32x4*)(dout + row * 4) = v; }

// Loss stack of the single-atlas path (stage1_neural_atlas.py:181-227; loss_utils.py:134-170,227-278,299-356)
// + seed gradients wrt every network output row.
__global__ __launch_bounds__(256) void k_loss_single(LossArgs a) {
  __shared__ float red[4];
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  float l_rgb = 0.f, l_grad = 0.f, l_rig = 0.f, l_grig = 0.f, l_ff = 0.f, l_fb = 0.f;
  if (n < a.N) {
    const size_t N = a.N;
    const float invN = 1.f / (float)a.N;
    const float* sp = a.samples + (size_t)n * AF_REC_F;
    const f32x4 uvc = *(const f32x4*)(a.out_map + (size_t)n * 4);
    const f32x4 uvy1 = *(const f32x4*)(a.out_map + (N + n) * 4);       (void)uvy1;
    float du = 0.f, dv = 0.f;
    // ---- rgb + gradient terms
    const f32x4 tc = *(const f32x4*)(a.out_atlas + (size_t)n * 4);
    const f32x4 ty = *(const f32x4*)(a.out_atlas + (N + n) * 4);
    const f32x4 tx = *(const f32x4*)(a.out_atlas + (2 * N + n) * 4);
    float dtc[3], dty[3], dtx[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float rgb = (tc[c] + 1.f) * 0.5f, rgby = (ty[c] + 1.f) * 0.5f, rgbx = (tx[c] + 1.f) * 0.5f;
      const float e = rgb - sp[REC_RGB + c];
      const float rx = sp[REC_DX + c] - (rgbx - rgb), ry = sp[REC_DY + c] - (rgby - rgb);
      l_rgb += e * e;
      l_grad += rx * rx + ry * ry;
      // d/d tanh-output = 1/2 d/d rgb
      dtc[c] = 0.5f * (a.c_rgb * 2.f * e + a.c_grad * 2.f * (rx + ry)) * invN;
      dtx[c] = 0.5f * (-a.c_grad * 2.f * rx) * invN;
      dty[c] = 0.5f * (-a.c_grad * 2.f * ry) * invN;
    }
    { f32x4 v = {dtc[0], dtc[1], dtc[2], 0.f}; *(f32x4*)(a.dout_atlas + (size_t)n * 4) = v; }
    { f32x4 v = {dty[0], dty[1], dty[2], 0.f}; *(f32x4*)(a.dout_atlas + (N + n) * 4) = v; }
    { f32x4 v = {dtx[0], dtx[1], dtx[2], 0.f}; *(f32x4*)(a.dout_atlas + (2 * N + n) * 4) = v; }
    put_d2(a.dout_map, N + n, 0.f, 0.f);        // (x,y+1) and (x+1,y) rows: gradient arrives from the atlas chain
    put_d2(a.dout_map, 2 * N + n, 0.f, 0.f);
    // ---- local rigidity
    {
      const f32x4 pym = *(const f32x4*)(a.out_map + (3 * N + n) * 4);
      const f32x4 pxm = *(const f32x4*)(a.out_map + (4 * N + n) * 4);
      const Rig r = rigidity(uvc[0], uvc[1], pym[0], pym[1], pxm[0], pxm[1], a.L, a.uv_scale, (float)a.d_local, a.c_rig * invN);
      l_rig = r.loss; du += r.du; dv += r.dv;
      put_d2(a.dout_map, 3 * N + n, r.du_ym, r.dv_ym);
      put_d2(a.dout_map, 4 * N + n, r.du_xm, r.dv_xm);
    }
    // ---- global rigidity
    if (a.nseg > 7) {
      const f32x4 pym = *(const f32x4*)(a.out_map + (5 * N + n) * 4);
      const f32x4 pxm = *(const f32x4*)(a.out_map + (6 * N + n) * 4);
      const Rig r = rigidity(uvc[0], uvc[1], pym[0], pym[1], pxm[0], pxm[1], a.L, a.uv_scale, (float)a.d_global, a.c_grig * invN);
      l_grig = r.loss; du += r.du; dv += r.dv;
      put_d2(a.dout_map, 5 * N + n, r.du_ym, r.dv_ym);
      put_d2(a.dout_map, 6 * N + n, r.du_xm, r.dv_xm);
    }
    // ---- optical flow (fwd: count[0]; bwd: count[1]; the compacted rows of the valid matches); alpha == 1 in the single path
    const float fscale = a.L / (2.f * a.uv_scale);
    const size_t flow_base = (size_t)(a.nseg - 2) * N;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
      const int rank = a.flow_rank[2 * n + dir];
      if (rank >= 0) {
        const size_t row = flow_base + rank;
        float gu = 0.f, gv = 0.f;
        const f32x4 m = *(const f32x4*)(a.out_map + row * 4);
        const float eu = m[0] - uvc[0], ev = m[1] - uvc[1];
        const float nrm = sqrtf(eu * eu + ev * ev);
        const float l = nrm * a.L / (2.f * a.uv_scale);
        if (dir) l_fb = l; else l_ff = l;
        const float cnt = (float)a.counts[dir];
        const float w = nrm > 0.f ? a.c_flow * 0.5f * fscale / (nrm * cnt) : 0.f;
        gu = w * eu; gv = w * ev;
        du -= gu; dv -= gv;
        put_d2(a.dout_map, row, gu, gv);
      }
    }
    put_d2(a.dout_map, (size_t)n, du, dv);
  }
  // rows between the last valid match and the end of its row tile are evaluated by the chains: they carry no gradient
  if (blockIdx.x == 0 && threadIdx.x < 32) {
    const size_t live_rows = (size_t)(a.nseg - 2) * a.N + *a.live, r = live_rows + threadIdx.x;
    if (r < ((live_rows + 31) & ~(size_t)31)) put_d2(a.dout_map, r, 0.f, 0.f);
  }
  float s[6] = {l_rgb, l_grad, l_rig, l_grig, l_ff, l_fb};
#pragma unroll
  for (int i = 0; i < 6; ++i) s[i] = block_sum(s[i], red);
  if (threadIdx.x == 0) {
    float* o = a.loss_part + (size_t)blockIdx.x * AF_LOSS_W;
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = s[i];
#pragma unroll
    for (int i = 6; i < AF_LOSS_W; ++i) o[i] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Loss stack of the fg/bg dual-atlas path (stage1_neural_atlas_seg.py:210-311; loss_utils.py:173-224,
// 227-278, 299-322, 385-408) + seed gradients wrt every output row of the four nets.
AF_DEV float alpha_of(float t) { float a = 0.5f * (t + 1.f); a = a * 0.99f; return a + 0.001f; }   // :224-227
#define AF_DALPHA 0.495f                                                                          // d alpha / d tanh-output

__global__ __launch_bounds__(256) void k_loss_seg(LossSegArgs a) {
  __shared__ float red[4];
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  float ls[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) ls[i] = 0.f;
  if (n < a.N) {
    const size_t N = a.N;
    const float invN = 1.f / (float)a.N;
    const float* sp = a.samples + (size_t)n * AF_REC_F;
    // ---- alpha at the centre, the +1 neighbours and the flow matches
    const float al = alpha_of(a.out_alpha[(size_t)n * 4]);
    const float aly = alpha_of(a.out_alpha[(N + n) * 4]);
    const float alx = alpha_of(a.out_alpha[(2 * N + n) * 4]);
    float dA = 0.f, dAy = 0.f, dAx = 0.f, dAf = 0.f, dAb = 0.f;
    // ---- colours: rgb = rgb1*alpha + rgb2*(1-alpha)  (:229-235), gradient loss (loss_utils.py:173-224), sparsity (:244,248)
    const f32x4 t1c = *(const f32x4*)(a.out_atlas + (size_t)n * 4);
    const f32x4 t1y = *(const f32x4*)(a.out_atlas + (N + n) * 4);
    const f32x4 t1x = *(const f32x4*)(a.out_atlas + (2 * N + n) * 4);
    const f32x4 t2c = *(const f32x4*)(a.out_atlas + (3 * N + n) * 4);
    const f32x4 t2y = *(const f32x4*)(a.out_atlas + (4 * N + n) * 4);
    const f32x4 t2x = *(const f32x4*)(a.out_atlas + (5 * N + n) * 4);
    float d1c[3], d1y[3], d1x[3], d2c[3], d2y[3], d2x[3];
    float sq1 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float r1 = (t1c[c] + 1.f) * 0.5f, r2 = (t2c[c] + 1.f) * 0.5f;
      const float r1y = (t1y[c] + 1.f) * 0.5f, r2y = (t2y[c] + 1.f) * 0.5f;
      const float r1x = (t1x[c] + 1.f) * 0.5f, r2x = (t2x[c] + 1.f) * 0.5f;
      const float rgb = r1 * al + r2 * (1.f - al);
      const float rgby = r1y * aly + r2y * (1.f - aly);
      const float rgbx = r1x * alx + r2x * (1.f - alx);
      const float e = rgb - sp[REC_RGB + c];
      const float rx = sp[REC_DX + c] - (rgbx - rgb), ry = sp[REC_DY + c] - (rgby - rgb);
      const float nf = r1 * (1.f - al);
      ls[0] += e * e;
      ls[1] += rx * rx + ry * ry;
      ls[13] += nf * nf;
      sq1 += r1 * r1;
      const float g = (a.c_rgb * 2.f * e + a.c_grad * 2.f * (rx + ry)) * invN;       // dL/d rgb
      const float gx = -a.c_grad * 2.f * rx * invN, gy = -a.c_grad * 2.f * ry * invN;   // dL/d rgb(x+1), rgb(y+1)
      // d/d tanh-output = 1/2 d/d rgb
      d1c[c] = 0.5f * (al * g + a.c_sparse * 2.f * r1 * (1.f - al) * (1.f - al) * invN);
      d2c[c] = 0.5f * (1.f - al) * g;
      d1x[c] = 0.5f * alx * gx; d2x[c] = 0.5f * (1.f - alx) * gx;
      d1y[c] = 0.5f * aly * gy; d2y[c] = 0.5f * (1.f - aly) * gy;
      dA += (r1 - r2) * g;
      dAx += (r1x - r2x) * gx;
      dAy += (r1y - r2y) * gy;
    }
    dA -= a.c_sparse * 2.f * (1.f - al) * sq1 * invN;
    { f32x4 v = {d1c[0], d1c[1], d1c[2], 0.f}; *(f32x4*)(a.dout_atlas + (size_t)n * 4) = v; }
    { f32x4 v = {d1y[0], d1y[1], d1y[2], 0.f}; *(f32x4*)(a.dout_atlas + (N + n) * 4) = v; }
    { f32x4 v = {d1x[0], d1x[1], d1x[2], 0.f}; *(f32x4*)(a.dout_atlas + (2 * N + n) * 4) = v; }
    { f32x4 v = {d2c[0], d2c[1], d2c[2], 0.f}; *(f32x4*)(a.dout_atlas + (3 * N + n) * 4) = v; }
    { f32x4 v = {d2y[0], d2y[1], d2y[2], 0.f}; *(f32x4*)(a.dout_atlas + (4 * N + n) * 4) = v; }
    { f32x4 v = {d2x[0], d2x[1], d2x[2], 0.f}; *(f32x4*)(a.dout_atlas + (5 * N + n) * 4) = v; }
    // ---- alpha bootstrapping, BCE against the (fractional) input mask (:301-302)
    {
      const float m = sp[REC_FG];
      ls[12] = -m * logf(al) - (1.f - m) * logf(1.f - al);
      dA += a.c_boot * (-m / al + (1.f - m) / (1.f - al)) * invN;
    }
    const float cntf = (float)a.counts[0], cntb = (float)a.counts[1];
    const bool vf = sp[REC_MF] != 0.f, vb = sp[REC_MB] != 0.f;
    // ---- alpha flow, L1 (loss_utils.py:385-408)
    const int rank_f = a.flow_rank[2 * n], rank_b = a.flow_rank[2 * n + 1];
    if (vf) {
      const float d = al - alpha_of(a.out_alpha[(3 * N + rank_f) * 4]);
      ls[10] = fabsf(d);
      const float g = a.c_aflow * 0.5f * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / cntf;
      dA += g; dAf = -g;
    }
    if (vb) {
      const float d = alpha_of(a.out_alpha[(3 * N + rank_b) * 4]) - al;
      ls[11] = fabsf(d);
      const float g = a.c_aflow * 0.5f * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / cntb;
      dAb = g; dA -= g;
    }
    // ---- the two mapping nets: rigidity (local, global) and alpha-weighted optical flow
    const float fscale = a.L / (2.f * a.uv_scale);
#pragma unroll
    for (int net = 0; net < 2; ++net) {
      const float* om = net ? a.out_m2 : a.out_m1;
      float* dm = net ? a.dout_m2 : a.dout_m1;
      const f32x4 uvc = *(const f32x4*)(om + (size_t)n * 4);
      float du = 0.f, dv = 0.f;
      put_d2(dm, N + n, 0.f, 0.f);        // (x,y+1) and (x+1,y) rows: gradient arrives from the atlas chain
      put_d2(dm, 2 * N + n, 0.f, 0.f);
      {
        const f32x4 pym = *(const f32x4*)(om + (3 * N + n) * 4);
        const f32x4 pxm = *(const f32x4*)(om + (4 * N + n) * 4);
        const Rig r = rigidity(uvc[0], uvc[1], pym[0], pym[1], pxm[0], pxm[1], a.L, a.uv_scale, (float)a.d_local, a.c_rig * invN);
        ls[2 + net] = r.loss; du += r.du; dv += r.dv;
        put_d2(dm, 3 * N + n, r.du_ym, r.dv_ym);
        put_d2(dm, 4 * N + n, r.du_xm, r.dv_xm);
      }
      if (a.nseg > 7) {
        const f32x4 pym = *(const f32x4*)(om + (5 * N + n) * 4);
        const f32x4 pxm = *(const f32x4*)(om + (6 * N + n) * 4);
        const Rig r = rigidity(uvc[0], uvc[1], pym[0], pym[1], pxm[0], pxm[1], a.L, a.uv_scale,
                               (float)(net ? a.d_global_bg : a.d_global_fg), (net ? a.c_grig_bg : a.c_grig_fg) * invN);
        ls[4 + net] = r.loss; du += r.du; dv += r.dv;
        put_d2(dm, 5 * N + n, r.du_ym, r.dv_ym);
        put_d2(dm, 6 * N + n, r.du_xm, r.dv_xm);
      }
      const float wrow = net ? 1.f - al : al;        // flow rows are weighted by alpha (fg) / 1-alpha (bg), :285-293
      const float wsign = net ? -1.f : 1.f;
#pragma unroll
      for (int dir = 0; dir < 2; ++dir) {
        const bool valid = dir ? vb : vf;
        if (valid) {
          const size_t row = (size_t)(a.nseg - 2) * N + (dir ? rank_b : rank_f);
          float gu = 0.f, gv = 0.f;
          const f32x4 m = *(const f32x4*)(om + row * 4);
          const float eu = m[0] - uvc[0], ev = m[1] - uvc[1];
          const float nrm = sqrtf(eu * eu + ev * ev);
          const float l = nrm * a.L / (2.f * a.uv_scale);
          const float cnt = dir ? cntb : cntf;
          ls[6 + 2 * net + dir] = l * wrow;
          const float w = nrm > 0.f ? a.c_flow * 0.5f * fscale * wrow / (nrm * cnt) : 0.f;
          gu = w * eu; gv = w * ev;
          du -= gu; dv -= gv;
          dA += wsign * a.c_flow * 0.5f * l / cnt;
          put_d2(dm, row, gu, gv);
        }
      }
      put_d2(dm, (size_t)n, du, dv);
    }
    // ---- alpha net seed gradients (tanh output -> alpha is affine with slope 0.495)
    { f32x4 v = {AF_DALPHA * dA, 0.f, 0.f, 0.f};  *(f32x4*)(a.dout_alpha + (size_t)n * 4) = v; }
    { f32x4 v = {AF_DALPHA * dAy, 0.f, 0.f, 0.f}; *(f32x4*)(a.dout_alpha + (N + n) * 4) = v; }
    { f32x4 v = {AF_DALPHA * dAx, 0.f, 0.f, 0.f}; *(f32x4*)(a.dout_alpha + (2 * N + n) * 4) = v; }
    if (vf) { f32x4 v = {AF_DALPHA * dAf, 0.f, 0.f, 0.f}; *(f32x4*)(a.dout_alpha + (3 * N + rank_f) * 4) = v; }
    if (vb) { f32x4 v = {AF_DALPHA * dAb, 0.f, 0.f, 0.f}; *(f32x4*)(a.dout_alpha + (3 * N + rank_b) * 4) = v; }
  }
  // rows between the last valid match and the end of its row tile are evaluated by the chains: they carry no gradient
  if (blockIdx.x == 0 && threadIdx.x < 32) {
    const size_t live = (size_t)*a.live;
    const size_t rm = (size_t)(a.nseg - 2) * a.N + live, ra = (size_t)3 * a.N + live;
    if (rm + threadIdx.x < ((rm + 31) & ~(size_t)31)) { put_d2(a.dout_m1, rm + threadIdx.x, 0.f, 0.f); put_d2(a.dout_m2, rm + threadIdx.x, 0.f, 0.f); }
    if (ra + threadIdx.x < ((ra + 31) & ~(size_t)31)) { f32x4 v = {0.f, 0.f, 0.f, 0.f}; *(f32x4*)(a.dout_alpha + (ra + threadIdx.x) * 4) = v; }
  }
#pragma unroll
  for (int i = 0; i < 14; ++i) ls[i] = block_sum(ls[i], red);
  if (threadIdx.x == 0) {
    float* o = a.loss_part + (size_t)blockIdx.x * AF_LOSS_W;
#pragma unroll
    for (int i = 0; i < 14; ++i) o[i] = ls[i];
    o[14] = 0.f; o[15] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Pre-train batch + loss (unwrap_utils.py:176-198): 10 000 random pixels of frame f,
// loss = mean || s*(x,y) - M(x,y,t) ||_2.
__global__ void k_pre_prep(PrePrepArgs a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  const int y = a.ys ? (int)a.ys[n] : (int)af_rand_below(a.seed, a.iter, (uint32_t)n, 1u, (uint64_t)a.resy);
  const int x = a.xs ? (int)a.xs[n] : (int)af_rand_below(a.seed, a.iter, (uint32_t)n, 2u, (uint64_t)a.resx);
  const float xc = (float)x / a.half_main - 1.f, yc = (float)y / a.half_main - 1.f;
  f32x4 v = {xc, yc, a.t, 0.f};
  *(f32x4*)(a.coords + (size_t)n * 4) = v;
  if (a.x0_tile) {      // the T-layout copy feeds the layer-0 dW of a net that reads xyt directly; a mapping net with PE stores its own PE tile
    float* tl = a.x0_tile + ((size_t)n >> 5) * 1024 + (n & 31);
    tl[0] = xc; tl[32] = yc; tl[64] = a.t;
  }
}

__global__ __launch_bounds__(256) void k_pre_loss(PreLossArgs a) {
  __shared__ float red[4];
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  float l = 0.f;
  if (n < a.N) {
    const f32x4 c = *(const f32x4*)(a.coords + (size_t)n * 4);
    const f32x4 o = *(const f32x4*)(a.out_map + (size_t)n * 4);
    const float eu = c[0] * a.uv_scale - o[0], ev = c[1] * a.uv_scale - o[1];
    l = sqrtf(eu * eu + ev * ev);
    const float w = l > 0.f ? -1.f / (l * (float)a.N) : 0.f;
    put_d2(a.dout_map, (size_t)n, w * eu, w * ev);
  }
  l = block_sum(l, red);
  if (threadIdx.x == 0) { float* o = a.loss_part + (size_t)blockIdx.x * AF_LOSS_W; o[0] = l; for (int i = 1; i < AF_LOSS_W; ++i) o[i] = 0.f; }
}

// ---------------------------------------------------------------------------------------------
// Split-K reduction + Adam (torch.optim.Adam defaults, stage1_neural_atlas.py:132-134,231) + re-emission
// of the three weight views the GEMM kernels read (canonical, forward image, backward image).
// One weight into a 256x256 bf16x3 block of a chain stream (layout: mlpbf.hip header): element (m, k) of the product's A
// matrix goes, as its three bf16 levels hi / mid / lo (round-to-nearest-even, exact fp32 residuals — bfsplit.h), to
// byte ((((s*8 + m/32)*3 + level)*2 + h)*32 + m%32)*16 + 2 i  with k = 32T + 8q + 4h + p,  s = 2T + q/2,  i = 4(q%2) + p.
AF_DEV uint16_t bf16_rne(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);          // finite inputs (parameters are checked finite below)
}
AF_DEV void emit_bf3(char* block, uint32_t m, uint32_t k, float p) {
  const uint32_t T = k >> 5, q = (k >> 3) & 3, hh = (k >> 2) & 1, pp = k & 3;
  const uint32_t s = 2 * T + (q >> 1), i = 4 * (q & 1) + pp;
  const uint16_t h = bf16_rne(p);
  const float r1 = p - __builtin_bit_cast(float, (uint32_t)h << 16);
  const uint16_t mi = bf16_rne(r1);
  const float r2 = r1 - __builtin_bit_cast(float, (uint32_t)mi << 16);
  const uint16_t lo = bf16_rne(r2);
  const size_t base = ((size_t)((s * 8 + (m >> 5)) * 3) * 2 + hh) * 32 * 16 + (size_t)(m & 31) * 16 + i * 2;
  *(uint16_t*)(block + base) = h;
  *(uint16_t*)(block + base + 1024) = mi;
  *(uint16_t*)(block + base + 2048) = lo;
}

// The same weight into a 256x256 block of an f16x3 chain stream (layout: mlphf.hip header): its two fp16 levels of 2^12 p — h = f16(2^12 p),
// l = f16(2^12 p - h), round to nearest even, the residual exact — at byte ((s*8 + m/32)*2 + level)*1024 + (h*32 + m%32)*16 + 2 i, indices as in emit_bf3.
AF_DEV void emit_hf2(char* block, uint32_t m, uint32_t k, float p) {
  const uint32_t T = k >> 5, q = (k >> 3) & 3, hh = (k >> 2) & 1, pp = k & 3;
  const uint32_t s = 2 * T + (q >> 1), i = 4 * (q & 1) + pp;
  const float ps = p * 4096.f;
  const _Float16 h = (_Float16)ps;
  const _Float16 l = (_Float16)(ps - (float)h);
  const size_t base = ((size_t)((s * 8 + (m >> 5)) * 2) * 2 + hh) * 32 * 16 + (size_t)(m & 31) * 16 + i * 2;
  *(_Float16*)(block + base) = h;
  *(_Float16*)(block + base + 1024) = l;
}

AF_DEV void emit_weight(const AdamJob& j, const AdamBufs& b, uint32_t o, uint32_t i, float p) {
  const uint32_t col = j.col0 + i;
  uint32_t k = col;
  if (col >= j.hid_cols) k = j.hid_cols + (uint32_t)af_pe_slot_of_feature((int)j.pe_kind, (int)(col - j.hid_cols));
  b.img_f[j.f_off + af_img_index(j.f_mpad, o, k)] = p;
  if (j.b_img_off >= 0 && col < (j.hid_cols ? j.hid_cols : j.p_ld)) {
    // W^T image: m = in feature (PE slot order for a PE first layer), k = out feature
    const uint32_t mrow = j.hid_cols ? col : (uint32_t)af_pe_slot_of_feature((int)j.pe_kind, (int)col);
    b.img_b[(uint32_t)j.b_img_off + af_img_index(j.b_mpad, mrow, o)] = p;
    if (b.sb && j.sb_off >= 0) {
      if (j.sb_kind == 0) ((float*)(b.sb + j.sb_off))[af_img_index(j.sb_mpad, mrow, o)] = p;
      else                emit_bf3(b.sb + j.sb_off, mrow, o, p);
    }
    if (b.hb && j.hb_off >= 0) {
      if (j.sb_kind == 0) ((float*)(b.hb + j.hb_off))[af_img_index(j.sb_mpad, mrow, o)] = p;
      else                emit_hf2(b.hb + j.hb_off, mrow, o, p);
    }
  }
  if (b.sf && j.sf_off >= 0) {
    if (j.sf_kind == 0) ((float*)(b.sf + j.sf_off))[af_img_index(j.sf_mpad, o, k - j.sf_k0)] = p;
    else                emit_bf3(b.sf + j.sf_off, o, k, p);
  }
  if (b.hf && j.hf_off >= 0) {
    if (j.sf_kind == 0) ((float*)(b.hf + j.hf_off))[af_img_index(j.sf_mpad, o, k - j.sf_k0)] = p;
    else                emit_hf2(b.hf + j.hf_off, o, k, p);
  }
}

template <bool UPDATE>
__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
  const AdamJob j = a.jobs[blockIdx.y];
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nw = j.out_real * j.in_real;
  const uint32_t nb = j.b_off >= 0 ? j.out_real : 0u;
  if (e < nw + nb) {
    const bool is_bias = e >= nw;
    const uint32_t o = is_bias ? e - nw : e / j.in_real;
    const uint32_t i = is_bias ? 0u : e - o * j.in_real;
    const uint32_t pidx = is_bias ? (uint32_t)j.b_off + o : j.p_off + o * j.p_ld + i;
    float p = a.bufs.params[pidx];
    if (UPDATE) {
      const float* pp = a.partial + j.part_off + (is_bias ? j.out_real_pad * j.pld + o : o * j.pld + i);
      // split-K reduction in slot order (fixed: runs are bit-reproducible).  Eight loads in flight per thread (as a plain loop hipcc emits
      // load / s_waitcnt vmcnt(0) / add per slot).  Measured (round 4, rocprofv3): 29.1 -> 30.6 us, i.e. nothing: with ~5 500 blocks in
      // flight the latency was already hidden across waves; the kernel moves the 66 MB of partial blocks k_dw has just written plus
      // ~35 MB of state and weight views, ~3.3 TB/s - it is bound by the partial volume (one 257 KB block per k_dw workgroup).
      float g = 0.f;
      uint32_t s = 0;
      for (; s + 8 <= j.nslots; s += 8) {
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = pp[(size_t)(s + q) * j.part_blk];
#pragma unroll
        for (int q = 0; q < 8; ++q) g += t[q];
      }
      if (s + 4 <= j.nslots) {
        float t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = pp[(size_t)(s + q) * j.part_blk];
#pragma unroll
        for (int q = 0; q < 4; ++q) g += t[q];
        s += 4;
      }
      if (s + 2 <= j.nslots) {
        const float t0 = pp[(size_t)s * j.part_blk], t1 = pp[(size_t)(s + 1) * j.part_blk];
        g += t0; g += t1; s += 2;
      }
      if (s < j.nslots) g += pp[(size_t)s * j.part_blk];
      float m = a.bufs.m[pidx], v = a.bufs.v[pidx];
      m = m + (g - m) * a.hy.one_minus_b1;
      v = v * a.hy.beta2 + (a.hy.one_minus_b2 * g) * g;
      const float denom = sqrtf(v) / a.hy.bc2_sqrt + a.hy.eps;
      p = p - a.hy.step_size * (m / denom);
      a.bufs.m[pidx] = m; a.bufs.v[pidx] = v; a.bufs.params[pidx] = p;
      if (a.grad_out) a.grad_out[pidx] = g;
    }
    // torch's relu / Linear carry a NaN or inf parameter into the loss; v_max_f32 (af_relu) would drop a NaN pre-activation
    // silently, so the condition is raised here instead (Adam's steps are bounded by lr: activations cannot overflow otherwise)
    if (a.nan_flag && !(fabsf(p) <= 3.4028235e38f)) atomicOr(a.nan_flag, 1);
    if (a.nan_flag && a.bufs.hf && !is_bias && j.sf_kind == 1 && fabsf(p) >= 8.f) atomicOr(a.nan_flag, 2);      // the fp16 images hold 2^12 w: finite below |w| = 16
    if (is_bias) a.bufs.bias_img[j.bias_img_off + o] = p;
    else emit_weight(j, a.bufs, o, i, p);
  }
  if (UPDATE && blockIdx.x == 0 && blockIdx.y == 0 && a.loss_out) {
    // fold the per-block loss partials of this step (fixed order); report and reset the flow counters
    if (threadIdx.x < AF_LOSS_W - 2) {
      float s = 0.f;
      for (int b = 0; b < a.loss_nblk; ++b) s += a.loss_part[b * AF_LOSS_W + threadIdx.x];
      a.loss_out[threadIdx.x] = s;
      if (a.nan_flag && !(s == s)) atomicOr(a.nan_flag, 1);
    } else if (threadIdx.x < AF_LOSS_W) {
      a.loss_out[threadIdx.x] = (float)a.counts[threadIdx.x - (AF_LOSS_W - 2)];
      if (a.nan_flag && a.check_counts && a.counts[threadIdx.x - (AF_LOSS_W - 2)] == 0) atomicOr(a.nan_flag, 1);     // mean over an empty set (loss_utils.py:317-320)
      a.counts[threadIdx.x - (AF_LOSS_W - 2)] = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Render helpers (evaluate.py:640-661,705-743): coordinates of every pixel of a frame, rgb = (t+1)/2,
// and the fp64 squared-error sum against the input frame for PSNR.
__global__ void k_frame_coords(float* coords, int resx, int resy, float half_main, float t, int npix_pad) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= npix_pad) return;
  const int npix = resx * resy;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (r < npix) { const int y = r / resx, x = r - y * resx; v[0] = (float)x / half_main - 1.f; v[1] = (float)y / half_main - 1.f; v[2] = t; }
  *(f32x4*)(coords + (size_t)r * 4) = v;
}

__global__ __launch_bounds__(256) void k_frame_finish(const float* out_atlas, const float* table, float* rgb_out, double* sse_part,
                                                     int npix, size_t rec0) {
  __shared__ double red[4];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  double sse = 0.0;
  if (r < npix) {
    const f32x4 t = *(const f32x4*)(out_atlas + (size_t)r * 4);
    const float* rec = table + (rec0 + r) * AF_REC_F;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = (t[c] + 1.f) * 0.5f;
      rgb_out[(size_t)r * 3 + c] = v;
      const double d = (double)rec[REC_RGB + c] - (double)v;
      sse += d * d;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sse += __shfl_xor(sse, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sse;
  __syncthreads();
  if (threadIdx.x == 0) sse_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// fg/bg path (evaluate.py:302-337): rgb = rgb1*alpha + rgb2*(1-alpha); out_atlas holds the fg rows then,
// `row2` rows later, the bg rows.
__global__ __launch_bounds__(256) void k_frame_finish_seg(const float* out_atlas, const float* out_alpha, size_t row2, const float* table,
                                                         float* rgb_out, double* sse_part, int npix, size_t rec0) {
  __shared__ double red[4];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  double sse = 0.0;
  if (r < npix) {
    const f32x4 t1 = *(const f32x4*)(out_atlas + (size_t)r * 4);
    const f32x4 t2 = *(const f32x4*)(out_atlas + (row2 + r) * 4);
    const float al = alpha_of(out_alpha[(size_t)r * 4]);
    const float* rec = table + (rec0 + r) * AF_REC_F;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = ((t1[c] + 1.f) * 0.5f) * al + ((t2[c] + 1.f) * 0.5f) * (1.f - al);
      rgb_out[(size_t)r * 3 + c] = v;
      const double d = (double)rec[REC_RGB + c] - (double)v;
      sse += d * d;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sse += __shfl_xor(sse, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sse;
  __syncthreads();
  if (threadIdx.x == 0) sse_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------
extern "C" {
int af_launch_resize(const ResizeArgs* a, hipStream_t s) {
  const long long n = (long long)a->dh * a->dw;
  hipLaunchKernelGGL(k_resize_bilinear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *a);
  return (int)hipGetLastError();
}
int af_launch_consistency(const ConsistencyArgs* a, hipStream_t s) {
  const long long n = (long long)a->h * a->w;
  hipLaunchKernelGGL(k_flow_consistency, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *a);
  return (int)hipGetLastError();
}
int af_launch_loss_seg(const LossSegArgs* a, hipStream_t s) {
  hipLaunchKernelGGL(k_loss_seg, dim3((a->N + 255) / 256), dim3(256), 0, s, *a);
  return (int)hipGetLastError();
}
int af_launch_frame_finish_seg(const float* out_atlas, const float* out_alpha, size_t row2, const float* table, float* rgb_out, double* sse_part,
                               int npix, size_t rec0, hipStream_t s) {
  hipLaunchKernelGGL(k_frame_finish_seg, dim3((npix + 255) / 256), dim3(256), 0, s, out_atlas, out_alpha, row2, table, rgb_out, sse_part, npix, rec0);
  return (int)hipGetLastError();
}
int af_launch_pack(const PackArgs* a, hipStream_t s) {
  const size_t n = (size_t)a->resx * a->resy * a->F;
  hipLaunchKernelGGL(k_pack_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *a);
  return (int)hipGetLastError();
}
int af_launch_prep(const PrepArgs* a, hipStream_t s) {
  hipLaunchKernelGGL(k_prep, dim3((a->N + 255) / 256), dim3(256), 0, s, *a);
  return (int)hipGetLastError();
}
int af_launch_loss_single(const LossArgs* a, hipStream_t s) {
  hipLaunchKernelGGL(k_loss_single, dim3((a->N + 255) / 256), dim3(256), 0, s, *a);
  return (int)hipGetLastError();
}
int af_launch_pre_prep(const PrePrepArgs* a, hipStream_t s) {
  hipLaunchKernelGGL(k_pre_prep, dim3((a->N + 255) / 256), dim3(256), 0, s, *a);
  return (int)hipGetLastError();
}
int af_launch_pre_loss(const PreLossArgs* a, hipStream_t s) {
  hipLaunchKernelGGL(k_pre_loss, dim3((a->N + 255) / 256), dim3(256), 0, s, *a);
  return (int)hipGetLastError();
}
int af_launch_adam(const AdamArgs* a, int njobs, int update, hipStream_t s) {
  const dim3 grid((AF_HID * 320 + AF_HID + 255) / 256, njobs);
  if (update) hipLaunchKernelGGL(k_adam<true>, grid, dim3(256), 0, s, *a);
  else        hipLaunchKernelGGL(k_adam<false>, grid, dim3(256), 0, s, *a);
  return (int)hipGetLastError();
}
int af_launch_frame_coords(float* coords, int resx, int resy, float half_main, float t, int npix_pad, hipStream_t s) {
  hipLaunchKernelGGL(k_frame_coords, dim3((npix_pad + 255) / 256), dim3(256), 0, s, coords, resx, resy, half_main, t, npix_pad);
  return (int)hipGetLastError();
}
int af_launch_frame_finish(const float* out_atlas, const float* table, float* rgb_out, double* sse_part, int npix, size_t rec0, hipStream_t s) {
  hipLaunchKernelGGL(k_frame_finish, dim3((npix + 255) / 256), dim3(256), 0, s, out_atlas, table, rgb_out, sse_part, npix, rec0);
  return (int)hipGetLastError();
}
}
