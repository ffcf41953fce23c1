// tools/valuprobe.hip — issue cost (core ticks, s_memtime) of the VALU instructions the bf16 split is made of, one wave per SIMD,
// back-to-back independent instructions.  Usage: valuprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int OP> __global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink, float seed) {
  float v0 = seed + threadIdx.x, v1 = seed * 2 + threadIdx.x, v2 = seed * 3, v3 = seed * 5, v4 = seed * 7, v5 = seed * 11, v6 = seed * 13, v7 = seed * 17;
  unsigned u0, u1, u2, u3;
  asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(v0), "v"(v1));
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 64; ++it) {
    if constexpr (OP == 0) asm volatile(REP64("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %6, %7\n v_cvt_pk_bf16_f32 %2, %8, %9\n v_cvt_pk_bf16_f32 %3, %10, %11\n") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    if constexpr (OP == 1) asm volatile(REP64("v_pk_add_f32 %0, %2, %3\n v_pk_add_f32 %1, %3, %2\n v_pk_add_f32 %0, %3, %3\n v_pk_add_f32 %1, %2, %2\n") : "+v"(*(double*)&u0), "+v"(*(double*)&u2) : "v"(*(double*)&v0), "v"(*(double*)&v2));
    if constexpr (OP == 2) asm volatile(REP64("v_lshlrev_b32 %0, 16, %4\n v_and_b32 %1, 0xffff0000, %5\n v_lshlrev_b32 %2, 16, %6\n v_and_b32 %3, 0xffff0000, %7\n") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
    if constexpr (OP == 3) asm volatile(REP64("v_sub_f32 %0, %4, %5\n v_sub_f32 %1, %6, %7\n v_sub_f32 %2, %5, %4\n v_sub_f32 %3, %7, %6\n") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
    if constexpr (OP == 4) asm volatile(REP64("v_perm_b32 %0, %4, %5, %8\n v_perm_b32 %1, %6, %7, %8\n v_perm_b32 %2, %5, %4, %8\n v_perm_b32 %3, %7, %6, %8\n") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4));
    if constexpr (OP == 5) asm volatile(REP64("v_dot2c_f32_bf16 %0, %4, %5\n v_dot2c_f32_bf16 %1, %6, %7\n v_dot2c_f32_bf16 %2, %5, %4\n v_dot2c_f32_bf16 %3, %7, %6\n") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
    if constexpr (OP == 6) asm volatile(REP64("v_and_or_b32 %0, %4, %8, %5\n v_and_or_b32 %1, %6, %8, %7\n v_and_or_b32 %2, %5, %8, %4\n v_and_or_b32 %3, %7, %8, %6\n") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4));
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (u0 + u1 + u2 + u3 == 12345u) sink[0] = 1.f;
  if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
}
int main() {
  unsigned long long* d; float* sink; hipMalloc(&d, 256 * 8); hipMalloc(&sink, 4);
  const char* names[7] = {"v_cvt_pk_bf16_f32", "v_pk_add_f32", "v_lshlrev_b32 / v_and_b32", "v_sub_f32", "v_perm_b32", "v_dot2c_f32_bf16", "v_and_or_b32"};
  for (int op = 0; op < 7; ++op) {
    for (int r = 0; r < 2; ++r) {
      switch (op) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, sink, 1.5f); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, sink, 1.5f); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, sink, 1.5f); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, sink, 1.5f); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, d, sink, 1.5f); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, d, sink, 1.5f); break;
        default: hipLaunchKernelGGL(k<6>, dim3(256), dim3(256), 0, 0, d, sink, 1.5f); break;
      }
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256); hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    double t = 0; for (auto x : h) t += (double)x; t /= 256;
    printf("%-28s %.2f ticks per instruction (one wave per SIMD)\n", names[op], t / (64.0 * 64 * 4));
  }
  return 0;
}
