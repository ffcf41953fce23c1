// tools/f16probe.hip — what the fp16 matrix pipe of gfx950 does with the operands of a two-term fp16 split (not product code)
//
// Round 6: before the three-product fp16 chains are written, three questions only the hardware answers:
//   (1) does v_mfma_f32_32x32x16_f16 honour subnormal fp16 INPUTS (the low term of a split is subnormal for small elements of a row)?
//   (2) does v_cvt_pk_f16_f32 round to nearest even and produce subnormals; is v_fma_mix_f32 (x * s - f16 half) exact?
//   (3) how far from fp64 is a K = 256 dot product accumulated as hh + hl + lh on that pipe (per-row power-of-two scale on the
//       activations, a fixed power of two on the weights), next to bf16x6 on the bf16 pipe and to v_mfma_f32_32x32x2_f32 (= an fp32
//       fmaf chain) — on rows whose elements span many binades, the case a per-tensor scale failed on (profiles/r5_split_error_real_tensors.txt)?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/f16probe.hip -o tools/bin/f16probe        Run: f16probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pk_f16(float a, float b) { uint32_t r; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// x * s - (lo / hi half of a packed f16 pair), one rounding: v_fma_mix_f32 with src2 read as f16 (op_sel_hi[2] = 1), half chosen by op_sel[2]
__device__ __forceinline__ float res_lo(float x, float s, uint32_t h) { float r; asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(h)); return r; }
__device__ __forceinline__ float res_hi(float x, float s, uint32_t h) { float r; asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(h)); return r; }

// ---- (1) + (2): single values ------------------------------------------------------------------
__global__ void k_scalar(const float* in, float* out) {
  const int lane = threadIdx.x;
  // (1) A = 2^-20 (a subnormal fp16: 16 x 2^-24) in every k of lane half 0, B = 2^10: C = 8 x 2^-10 if the inputs are honoured, 0 if flushed
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = lane < 32 ? (_Float16)9.5367431640625e-07f : (_Float16)0.f; b[i] = (_Float16)1024.f; }
  f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];
  // the same with the subnormal on the B side
  c = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0);
  if (lane == 0) out[1] = c[0];
  // (2) conversions of in[lane]: h = f16(x * s) (RNE, subnormals), l = f16(x * s - h); reported as floats
  const float x = in[lane], s = in[64];
  const uint32_t h = pk_f16(x * s, -(x * s));
  const float rl = res_lo(x, s, h), rh = res_hi(-x, s, h);
  const uint32_t l = pk_f16(rl, rh);
  out[64 + lane] = (float)__builtin_bit_cast(f16x2, h)[0];
  out[128 + lane] = (float)__builtin_bit_cast(f16x2, h)[1];
  out[192 + lane] = rl; out[256 + lane] = rh;
  out[320 + lane] = (float)__builtin_bit_cast(f16x2, l)[0];
  out[384 + lane] = (float)__builtin_bit_cast(f16x2, l)[1];
}

// ---- (3): Y[32 out][32 rows] = W[32][K] * X[rows][K]^T, K = 256, one wave, four arithmetics ------
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }
struct S3 { u32x4 h, m, l; };
__device__ __forceinline__ S3 split_bf3(const float (&v)[8]) {
  S3 s;
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    const uint32_t h = pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const uint32_t m = pk_bf16(ra, rb);
    const float qa = ra - __builtin_bit_cast(float, m << 16), qb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    s.h[i] = h; s.m[i] = m; s.l[i] = pk_bf16(qa, qb);
  }
  return s;
}
struct S2 { u32x4 h, l; };
__device__ __forceinline__ S2 split_f16(const float (&v)[8], float s) {
  S2 r;
  for (int i = 0; i < 4; ++i) {
    const uint32_t h = pk_f16(v[2 * i] * s, v[2 * i + 1] * s);
    r.h[i] = h; r.l[i] = pk_f16(res_lo(v[2 * i], s, h), res_hi(v[2 * i + 1], s, h));
  }
  return r;
}
__device__ __forceinline__ f32x16 mfma_bf(const u32x4& a, const u32x4& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma_h(const u32x4& a, const u32x4& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); }

// mode 0: fp32 MFMA   1: bf16x6   2: f16 split, 3 products, per-row scale   3: the same, 4 products   4: f16 split with ONE scale for all rows (per tensor)
__global__ void k_dot(const float* W, const float* X, float* Y, int K, float wscale, float xscale_tensor, int mode) {
  const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
  const float* w = W + ((size_t)blockIdx.x * 32 + m) * K;      // A: lane holds W[m][k0 + 8h + i]
  const float* x = X + ((size_t)blockIdx.x * 32 + m) * K;      // B: lane holds X[row m][k0 + 8h + i]
  f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float xs = xscale_tensor;
  if (mode == 2 || mode == 3) {      // the row's own scale: its largest magnitude to [2^14, 2^15)
    float mx = 0.f;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(x[k]));
    int e; frexpf(mx, &e);
    xs = mx > 0.f ? ldexpf(1.f, 15 - e) : 1.f;
  }
  for (int k0 = 0; k0 < K; k0 += 16) {
    float a8[8], b8[8];
    for (int i = 0; i < 8; ++i) { a8[i] = w[k0 + 8 * h + i]; b8[i] = x[k0 + 8 * h + i]; }
    if (mode == 0) {
      for (int i = 0; i < 8; ++i) {      // 32x32x2: lane half h supplies k = h; feed the two halves' values in turn
        for (int hh = 0; hh < 2; ++hh) {
          const float av = __shfl(a8[i], m + 32 * hh), bv = __shfl(b8[i], m + 32 * hh);
          c = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? av : 0.f, h == 0 ? bv : 0.f, c, 0, 0, 0);
        }
      }
    } else if (mode == 1) {
      const S3 A = split_bf3(a8), B = split_bf3(b8);
      c = mfma_bf(A.l, B.h, c); c = mfma_bf(A.m, B.m, c); c = mfma_bf(A.m, B.h, c);
      c = mfma_bf(A.h, B.l, c); c = mfma_bf(A.h, B.m, c); c = mfma_bf(A.h, B.h, c);
    } else {
      const S2 A = split_f16(a8, wscale), B = split_f16(b8, xs);
      if (mode == 3) c = mfma_h(A.l, B.l, c);
      c = mfma_h(A.l, B.h, c); c = mfma_h(A.h, B.l, c); c = mfma_h(A.h, B.h, c);
    }
  }
  const float inv = mode >= 2 ? 1.f / (wscale * xs) : 1.f;      // a power of two: exact.  NB per-row scale = per lane column n = lane & 31 of C
  for (int r = 0; r < 16; ++r) {
    const int om = 8 * (r >> 2) + 4 * h + (r & 3);
    Y[((size_t)blockIdx.x * 32 + om) * 32 + m] = c[r] * inv;
  }
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
  // ---- scalar checks
  {
    std::vector<float> in(65), out(448);
    for (int i = 0; i < 64; ++i) in[i] = (float)(nrand() * ldexp(1.0, -(i % 32)));
    in[64] = 1.f;
    float *di, *dout;
    CK(hipMalloc(&di, 65 * 4)); CK(hipMalloc(&dout, 448 * 4));
    CK(hipMemcpy(di, in.data(), 65 * 4, hipMemcpyHostToDevice));
    k_scalar<<<1, 64>>>(di, dout);
    CK(hipMemcpy(out.data(), dout, 448 * 4, hipMemcpyDeviceToHost));
    printf("(1) MFMA f16 with subnormal A inputs: C = %g (honoured: %g; flushed: 0)   subnormal B inputs: C = %g\n", out[0], 8 * ldexp(1.0, -10), out[1]);
    int bad_h = 0, bad_l = 0, sub_h = 0, sub_l = 0; double worst = 0;
    for (int i = 0; i < 64; ++i) {
      const float x = in[i];
      const _Float16 hh = (_Float16)x;      // host RNE
      const float r = x - (float)hh; const _Float16 ll = (_Float16)r;
      if ((float)hh != out[64 + i] || (float)(_Float16)(-x) != out[128 + i]) ++bad_h;
      if (r != out[192 + i] || -r != out[256 + i]) ++bad_h;
      if ((float)ll != out[320 + i] || -(float)ll != out[384 + i]) ++bad_l;
      if (fabsf((float)hh) < 6.1035e-5f && hh != 0) ++sub_h;
      if (fabsf((float)ll) < 6.1035e-5f && ll != 0) ++sub_l;
      if (x != 0) worst = fmax(worst, fabs(((double)x - (double)out[64 + i] - (double)out[320 + i]) / x));
    }
    printf("(2) v_cvt_pk_f16_f32 / v_fma_mix_f32 against the host's RNE conversions on 64 values down to 2^-31: %d hi / residual mismatches, %d lo mismatches "
           "(%d subnormal hi, %d subnormal lo terms among them); worst |x - h - l| / |x| = %.3g\n", bad_h, bad_l, sub_h, sub_l, worst);
  }
  // ---- dot products
  const int K = 256, NB = 64;      // 64 blocks of 32 x 32 outputs
  struct Case { const char* name; int kind; };
  const Case cases[] = {{"activations: post-ReLU |N(0,1)|, half of them zero; rows scaled by 2^U(-20,4)", 0},
                        {"backward-like: every element 2^U(-24,0) x N(0,1), rows scaled by 2^U(-20,4)", 1},
                        {"all rows O(1): N(0,1)", 2}};
  for (const Case& cs : cases) {
    srand(12345);
    std::vector<float> W((size_t)NB * 32 * K), X((size_t)NB * 32 * K);
    for (auto& v : W) v = (float)((urand() * 2 - 1) * 0.0625 * (urand() < 0.02 ? 8.0 : 1.0));
    for (size_t r = 0; r < (size_t)NB * 32; ++r) {
      const double rs = cs.kind == 2 ? 1.0 : ldexp(1.0, (int)floor(urand() * 24) - 20);
      for (int k = 0; k < K; ++k) {
        double v = nrand();
        if (cs.kind == 0) v = urand() < 0.5 ? 0.0 : fabs(v);
        if (cs.kind == 1) v *= exp2(-24.0 * urand());
        X[r * K + k] = (float)(v * rs);
      }
    }
    float wmax = 0, xmax = 0;
    for (float v : W) wmax = fmaxf(wmax, fabsf(v));
    for (float v : X) xmax = fmaxf(xmax, fabsf(v));
    int e; frexpf(xmax, &e); const float xs_tensor = ldexpf(1.f, 15 - e);
    const float wscale = 4096.f;      // fixed: |W| < 16
    float *dW, *dX, *dY;
    CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dY, (size_t)NB * 1024 * 4));
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    printf("(3) %s   (max|W| %.3g, max|X| %.3g)\n", cs.name, wmax, xmax);
    const char* names[5] = {"fp32 MFMA (32x32x2_f32)", "bf16x6", "f16 split, 3 products, row scale", "f16 split, 4 products, row scale", "f16 split, 3 products, ONE scale"};
    for (int mode = 0; mode < 5; ++mode) {
      k_dot<<<NB, 64>>>(dW, dX, dY, K, wscale, xs_tensor, mode);
      std::vector<float> Y((size_t)NB * 1024);
      CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
      double s2 = 0, worst = 0; size_t n = 0;
      for (int b = 0; b < NB; ++b)
        for (int o = 0; o < 32; ++o)
          for (int r = 0; r < 32; ++r) {
            double ref = 0, den = 0;
            const float* w = &W[((size_t)b * 32 + o) * K]; const float* x = &X[((size_t)b * 32 + r) * K];
            for (int k = 0; k < K; ++k) { ref += (double)w[k] * x[k]; den += fabs((double)w[k] * x[k]); }
            if (den == 0) continue;
            const double err = fabs((double)Y[((size_t)b * 32 + o) * 32 + r] - ref) / den;
            s2 += err * err; worst = fmax(worst, err); ++n;
          }
      printf("      %-36s rms %.3g  worst %.3g   (x 2^-24: %.2f / %.2f)\n", names[mode], sqrt(s2 / n), worst, sqrt(s2 / n) * 16777216.0, worst * 16777216.0);
    }
    CK(hipFree(dW)); CK(hipFree(dX)); CK(hipFree(dY));
  }
  return 0;
}
