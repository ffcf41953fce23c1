#!/usr/bin/env python
"""Generates tools/issueprobe.hip: single-wave-per-SIMD issue-cost probe for the instruction mix of the bf16x6 chains
(not product code).  Every body is ONE hand-ordered asm block — 48 v_mfma_f32_32x32x16_bf16 per iteration (one k-step of a
256x256 layer, independent accumulators) with a chosen filler list in each MFMA's shadow — timed with s_memtime on every
workgroup of a full-chip launch (256 workgroups x 4 waves, one wave per SIMD).  Prints core ticks per k-step-equivalent.

    python tools/issueprobe_gen.py > tools/issueprobe.hip && hipcc --offload-arch=gfx950 -O3 tools/issueprobe.hip -o tools/bin/issueprobe

Filler tokens: v = one VALU (v_add_f32 on scratch registers), c = v_cvt_pk_bf16_f32, D = ds_read_b128, S = buffer_store_dword
(T-layout tile pattern: 2 x 128 B), X = buffer_store_dwordx4 (1 KB contiguous), L = global_load_lds_dwordx4 (1 KB, L2-resident
source), w = s_waitcnt lgkmcnt(0)."""

NM = 48


def spread(n_m, items):
    """items: list of filler strings (each a group that stays together); distribute evenly over n_m gaps."""
    gaps = [""] * n_m
    n = len(items)
    for k, it in enumerate(items):
        gaps[(k * n_m) // n] += it
    return gaps


BODIES = {}
BODIES["mfma_only"] = [""] * NM
for k in (1, 2, 3, 4, 5, 6, 8):
    BODIES["valu%d_per_gap" % k] = ["v" * k] * NM
BODIES["cvt2_per_gap"] = ["cc"] * NM
BODIES["ds24"] = spread(NM, ["D"] * 24)
BODIES["ds48"] = ["D"] * NM
BODIES["store8_dword"] = spread(NM, ["S"] * 8)
BODIES["store16_dword"] = spread(NM, ["S"] * 16)
BODIES["store2_x4"] = spread(NM, ["X"] * 2)
BODIES["store4_x4"] = spread(NM, ["X"] * 4)
BODIES["store8_x4"] = spread(NM, ["X"] * 8)
BODIES["dma6"] = spread(NM, ["L"] * 6)
BODIES["dma12"] = spread(NM, ["L"] * 12)
# the chain's k-step mix (48 M, 24 D, 44 v, 6 L, 8 S): VMEM alone in its gap, the rest spread two or three per gap
vmem = ["L"] * 6 + ["S"] * 8
light = []
for i in range(24):
    light.append("D" + ("vv" if i < 20 else "v"))
mix = [""] * NM
vm_slots = [(k * NM) // 14 for k in range(14)]
li = 0
order = ["S", "L", "S", "S", "L", "S", "L", "S", "S", "L", "S", "L", "S", "L"]
oi = 0
rest = [g for g in range(NM) if g not in vm_slots]
for g in vm_slots:
    mix[g] = order[oi]; oi += 1
for k, it in enumerate(light):
    mix[rest[(k * len(rest)) // len(light)]] += it
BODIES["chain_mix_ideal"] = mix
# the same multiset clumped the way the round-2 kernel's stream looks: bare MFMA runs, DMA and VALU clumps
cl = [""] * NM
cl[0] = "vvv"; cl[1] = ""; cl[2] = ""
for g in range(17, 25):
    cl[g] = "D"
cl[25] = "LL" + "vvvv"; cl[26] = "L"
for g in range(27, 35):
    cl[g] = "D" + "vv"
cl[35] = "L" + "v" * 24
for g in range(36, 44):
    cl[g] = "S" if g % 2 == 0 else "SD"
cl[44] = "DDDD"; cl[45] = "LL"; cl[46] = "S" * 0
n_s = sum(x.count("S") for x in cl)
cl[47] = "S" * (8 - n_s)
BODIES["chain_mix_clumped"] = cl
# what a k-step would cost with 16-byte stores (2 instead of 8) and the DMA unchanged
vm2 = ["L"] * 6 + ["X"] * 2
mix2 = [""] * NM
vm_slots2 = [(k * NM) // 8 for k in range(8)]
rest2 = [g for g in range(NM) if g not in vm_slots2]
o2 = ["X", "L", "L", "L", "X", "L", "L", "L"]
for k, g in enumerate(vm_slots2):
    mix2[g] = o2[k]
for k, it in enumerate(light):
    mix2[rest2[(k * len(rest2)) // len(light)]] += it
BODIES["chain_mix_x4_stores"] = mix2
# no stores at all (forward without training tiles) / no DMA
BODIES["chain_mix_no_store"] = [("" if t in ("S",) else t) for t in mix]
BODIES["chain_mix_no_vmem"] = [("" if t in ("S", "L") else t) for t in mix]


def emit_body(gaps):
    out = []
    vs = 0; ds = 0; st = 0; dm = 0
    for g, fill in enumerate(gaps):
        a = (g % 8) * 16
        out.append("v_mfma_f32_32x32x16_bf16 a[%d:%d], v[0:3], v[4:7], a[%d:%d]" % (a, a + 15, a, a + 15))
        for t in fill:
            if t == "v":
                r = 8 + (vs % 24); vs += 1
                out.append("v_add_f32 v%d, v%d, v%d" % (r, 8 + ((vs + 7) % 24), 8 + ((vs + 13) % 24)))
            elif t == "c":
                r = 8 + (vs % 24); vs += 1
                out.append("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (r, 8 + ((vs + 7) % 24), 8 + ((vs + 13) % 24)))
            elif t == "D":
                d = 48 + 4 * (ds % 8); ds += 1
                out.append("ds_read_b128 v[%d:%d], v40 offset:%d" % (d, d + 3, (ds % 24) * 1024))
            elif t == "S":
                out.append("buffer_store_dword v80, v41, s[20:23], s26 offen offset:%d" % ((st % 16) * 128 % 4096)); st += 1
            elif t == "X":
                out.append("buffer_store_dwordx4 v[80:83], v42, s[20:23], s26 offen offset:%d" % ((st % 4) * 1024)); st += 1
            elif t == "L":
                out.append("s_mov_b32 m0, s27")
                out.append("global_load_lds_dwordx4 v43, s[24:25] offset:%d" % ((dm % 4) * 1024)); dm += 1
                out.append("s_add_u32 s27, s27, 0x1000"); out.append("s_and_b32 s27, s27, 0xffff")
            elif t == "w":
                out.append("s_waitcnt lgkmcnt(0)")
    return out


def main():
    print("// GENERATED by tools/issueprobe_gen.py - do not edit.  Issue-cost probe, not product code.")
    print("#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include <stdlib.h>\n#include <vector>")
    print("#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf(\"%s: %s\\n\", #x, hipGetErrorString(e)); exit(1); } } while (0)")
    names = list(BODIES)
    clob = ", ".join(['"v%d"' % i for i in range(0, 100)] + ['"a%d"' % i for i in range(128)] +
                     ['"s%d"' % i for i in range(20, 32)] + ['"m0"', '"memory"', '"scc"'])
    for n in names:
        body = emit_body(BODIES[n])
        print("__global__ __launch_bounds__(256, 1) void k_%s(float* stbuf, const float* src, unsigned long long* out, int iters) {" % n)
        print("  extern __shared__ __attribute__((aligned(16))) char smem[];")
        print("  unsigned long long t0, t1;")
        print("  float* st = stbuf + (size_t)blockIdx.x * (1u << 20);   // 4 MB per workgroup, 1 MB per wave (the wave's share is in the lane offsets)")
        print("  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;")
        print("  const unsigned lds_a = (unsigned)(size_t)smem + lane * 16, vo_t = (lane & 31) * 4 + (lane >> 5) * 512 + (wave << 20), vo_x = lane * 16 + (wave << 20), vo_g = threadIdx.x * 16;")
        print("  const unsigned lds_dma = (unsigned)(size_t)smem + 49152 + wave * 1024;")
        print("  asm volatile(")
        pre = [
            "s_mov_b32 s20, %2", "s_mov_b32 s21, %3", "s_mov_b32 s22, 0x7fffffff", "s_mov_b32 s23, 0x00020000",
            "s_mov_b32 s24, %4", "s_mov_b32 s25, %5", "s_mov_b32 s26, 0", "v_readfirstlane_b32 s27, %9", "s_mov_b32 s28, %6",
            "v_mov_b32 v40, %7", "v_mov_b32 v41, %8", "v_mov_b32 v42, %10", "v_mov_b32 v43, %11",
        ]
        for i in range(0, 40):
            pre.append("v_mov_b32 v%d, 0x3c003c00" % i if i < 8 else "v_mov_b32 v%d, 1.0" % i)
        for i in range(80, 84):
            pre.append("v_mov_b32 v%d, 2.0" % i)
        for i in range(128):
            pre.append("v_accvgpr_write_b32 a%d, 0" % i)
        pre += ["s_nop 7", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier", "s_memtime %0", "s_waitcnt lgkmcnt(0)", "1:"]
        post = ["s_add_u32 s26, s26, 0x4000", "s_and_b32 s26, s26, 0xfffff", "s_add_u32 s24, s24, 0x4000", "s_addc_u32 s25, s25, 0",
                "s_sub_u32 s28, s28, 1", "s_cmp_lg_u32 s28, 0", "s_cbranch_scc1 1b",
                "s_nop 7", "s_memtime %1", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
        # keep the DMA source inside 1 MB: the host passes a 64 MB buffer, iterations * 16 KB stays below it
        for ln in pre + body + post:
            print('    "%s\\n"' % ln)
        print('    : "=&s"(t0), "=&s"(t1)')
        print('    : "s"((unsigned)(size_t)st), "s"((unsigned)((size_t)st >> 32)), "s"((unsigned)(size_t)src), "s"((unsigned)((size_t)src >> 32)), "s"(iters),')
        print('      "v"(lds_a), "v"(vo_t), "v"(lds_dma), "v"(vo_x), "v"(vo_g)')
        print("    : %s);" % clob)
        print("  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t0; out[blockIdx.x * 2 + 1] = t1; }")
        print("}\n")
    print("typedef void (*kern_t)(float*, const float*, unsigned long long*, int);")
    print("int main(int argc, char** argv) {")
    print("  const int iters = argc > 1 ? atoi(argv[1]) : 256, nwg = argc > 2 ? atoi(argv[2]) : 256;")
    print("  float *st, *src; unsigned long long* out;")
    print("  CK(hipMalloc(&st, (size_t)nwg << 22)); CK(hipMalloc(&src, (size_t)64 << 20)); CK(hipMemset(src, 0, (size_t)64 << 20)); CK(hipMalloc(&out, nwg * 16));")
    print("  struct { const char* name; kern_t k; const char* mix; } ks[] = {")
    for n in names:
        print('    {"%s", k_%s, "%s"},' % (n, n, "|".join(BODIES[n])))
    print("  };")
    print("  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));")
    print("  for (auto& k : ks) {")
    print("    CK(hipFuncSetAttribute((const void*)k.k, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));")
    print("    hipLaunchKernelGGL(k.k, dim3(nwg), dim3(256), 98304, 0, st, src, out, iters); CK(hipDeviceSynchronize());")
    print("    CK(hipEventRecord(e0, 0));")
    print("    hipLaunchKernelGGL(k.k, dim3(nwg), dim3(256), 98304, 0, st, src, out, iters);")
    print("    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));")
    print("    float ms; CK(hipEventElapsedTime(&ms, e0, e1));")
    print("    std::vector<unsigned long long> h(nwg * 2); CK(hipMemcpy(h.data(), out, nwg * 16, hipMemcpyDeviceToHost));")
    print("    double tk = 0; for (int w = 0; w < nwg; ++w) tk += (double)(h[2 * w + 1] - h[2 * w]); tk /= nwg;")
    print('    printf("%-22s %8.1f ticks per 48 MFMAs (%.2f per MFMA; bare 1536)   %.3f ms  -> %.0f MHz\\n", k.name, tk / iters, tk / iters / 48.0, ms, tk / (ms * 1000));')
    print("  }")
    print("  return 0;\n}")


if __name__ == "__main__":
    main()
