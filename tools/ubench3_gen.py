#!/usr/bin/env python3
"""Generates tools/ubench3.hip: issue-cost model of the SG inner loop on gfx950 (development tool).

Each variant is the 12-lobe x 2-direction group of the forward kernel (per lobe: 2 instr for U, 2 for the
exponents, 2 v_exp_f32, 6 v_fmac_f32) written in inline asm with explicit physical registers, so that
operand kinds (SGPR vs VGPR), VGPR banks (index mod 4) and the order of transcendentals can be varied one
at a time.  The host side runs every variant at 1, 2 and 3 resident waves per SIMD and prints shader cycles
per wave-instruction per SIMD (s_memtime deltas, so the number does not depend on the clock).
"""
import sys

K = 12
variants = {}


def lobe_regs_compiler(k):
    # the allocation hipcc chose: lobe params at stride 7, banks rotate with k
    return dict(w0=26 + 7 * k, w1=27 + 7 * k, w2=28 + 7 * k, ax=142 - k, ay=131 - k, C=157 + k)


def lobe_regs_banked(k):
    # w0 in bank 1, w1 in bank 2, w2 in bank 3 ; ax bank 0, ay bank 1, C bank 2
    return dict(w0=25 + 4 * k, w1=26 + 4 * k, w2=27 + 4 * k, ax=80 + 4 * k, ay=81 + 4 * k, C=82 + 4 * k)


def body(regs, sgpr, banked_acc, exps=True, phased=False, vop3_acc=False, exp_sub="v_mov_b32_e32"):
    out, n = [], 0
    ca, sa, sr, nsr = ("s38", "s39", "s24", "-s24") if sgpr else ("v200", "v201", "v202", "-v202")
    if banked_acc:
        # e+ = v176 (bank 0), e- = v180 (bank 0), U = v177; acc+ {w0->v170(b2), w1->v171(b3), w2->v173(b1)}, acc- likewise
        ep, em, U = 176, 180, 177
        accp = {"w0": 170, "w1": 171, "w2": 173}
        accm = {"w0": 174, "w1": 175, "w2": 181}
    else:
        ep, em, U = 178, 176, 177
        accp = {"w0": 171, "w1": 172, "w2": 173}
        accm = {"w0": 170, "w1": 174, "w2": 175}
    ph = [[], [], []]
    for k in range(K):
        r = regs(k)
        a = [f"v_mul_f32_e32 v{U}, {ca}, v{r['ax']}", f"v_fmac_f32_e32 v{U}, {sa}, v{r['ay']}"]
        if phased:
            # phased needs separate exponent registers per lobe: v210+2k, v211+2k
            e0, e1 = 210 + 2 * k, 211 + 2 * k
        else:
            e0, e1 = ep, em
        a += [f"v_fma_f32 v{e0}, {sr}, v{U}, v{r['C']}", f"v_fma_f32 v{e1}, {nsr}, v{U}, v{r['C']}"]
        op = "v_exp_f32_e32" if exps else exp_sub
        b = [f"{op} v{e0}, v{e0}", f"{op} v{e1}, v{e1}"]
        c = []
        for e, acc in ((e0, accp), (e1, accm)):
            for w in ("w1", "w2", "w0"):
                if vop3_acc:
                    c.append(f"v_fma_f32 v{acc[w]}, v{r[w]}, v{e}, v{acc[w]}")
                else:
                    c.append(f"v_fmac_f32_e32 v{acc[w]}, v{r[w]}, v{e}")
        if phased:
            ph[0] += a; ph[1] += b; ph[2] += c
        else:
            out += a + b + c
        n += 12
    if phased:
        out = ph[0] + ph[1] + ph[2]
    return out, n


def only(kind):
    out = []
    if kind == "exp24":       # 24 independent exps
        for i in range(24):
            out.append(f"v_exp_f32_e32 v{210 + i}, v{100 + i}")
    elif kind == "exp24_dep":   # in place
        for i in range(24):
            out.append(f"v_exp_f32_e32 v{210 + i}, v{210 + i}")
    elif kind == "fmac72":    # accumulations only, banked
        for k in range(K):
            r = lobe_regs_banked(k)
            for e, acc in ((176, (170, 171, 173)), (180, (174, 175, 181))):
                for w, a in zip(("w1", "w2", "w0"), (acc[1], acc[2], acc[0])):
                    out.append(f"v_fmac_f32_e32 v{a}, v{r[w]}, v{e}")
    elif kind == "fmac72_conf":   # accumulations only, compiler banks
        for k in range(K):
            r = lobe_regs_compiler(k)
            for e, acc in ((178, (171, 172, 173)), (176, (170, 174, 175))):
                for w, a in zip(("w1", "w2", "w0"), (acc[1], acc[2], acc[0])):
                    out.append(f"v_fmac_f32_e32 v{a}, v{r[w]}, v{e}")
    elif kind == "exp_fmac_1to3":   # 1 exp : 3 fmac, all VGPR, banked, no U/arg instrs
        for k in range(K):
            r = lobe_regs_banked(k)
            for e, acc, src in ((176, (170, 171, 173), 100 + k), (180, (174, 175, 181), 120 + k)):
                out.append(f"v_exp_f32_e32 v{e}, v{src}")
                for w, a in zip(("w1", "w2", "w0"), (acc[1], acc[2], acc[0])):
                    out.append(f"v_fmac_f32_e32 v{a}, v{r[w]}, v{e}")
    elif kind == "sgpr_fma48":  # argument instrs only, SGPR operands
        for k in range(K):
            r = lobe_regs_compiler(k)
            out += [f"v_mul_f32_e32 v177, s38, v{r['ax']}", f"v_fmac_f32_e32 v177, s39, v{r['ay']}",
                    f"v_fma_f32 v178, s24, v177, v{r['C']}", f"v_fma_f32 v176, -s24, v177, v{r['C']}"]
    elif kind == "vgpr_fma48":
        for k in range(K):
            r = lobe_regs_banked(k)
            out += [f"v_mul_f32_e32 v177, v200, v{r['ax']}", f"v_fmac_f32_e32 v177, v201, v{r['ay']}",
                    f"v_fma_f32 v176, v202, v177, v{r['C']}", f"v_fma_f32 v180, -v202, v177, v{r['C']}"]
    return out, len(out)


variants["A real (sgpr, compiler banks)"] = body(lobe_regs_compiler, True, False)
variants["B vgpr operands, compiler banks"] = body(lobe_regs_compiler, False, False)
variants["C sgpr operands, conflict-free banks"] = body(lobe_regs_banked, True, True)
variants["D vgpr + conflict-free banks"] = body(lobe_regs_banked, False, True)
variants["E = D, exps replaced by v_mov"] = body(lobe_regs_banked, False, True, exps=False)
variants["F = D phased (args | exps | fmacs)"] = body(lobe_regs_banked, False, True, phased=True)
variants["G = A phased"] = body(lobe_regs_compiler, True, False, phased=True)
variants["H = D with VOP3 v_fma accumulations"] = body(lobe_regs_banked, False, True, vop3_acc=True)
variants["I = A, exps replaced by v_mov"] = body(lobe_regs_compiler, True, False, exps=False)
variants["J 24 independent v_exp_f32"] = only("exp24")
variants["K 24 in-place v_exp_f32"] = only("exp24_dep")
variants["L 72 v_fmac conflict-free"] = only("fmac72")
variants["M 72 v_fmac compiler banks"] = only("fmac72_conf")
variants["N 1 exp : 3 fmac (vgpr, banked)"] = only("exp_fmac_1to3")
variants["O 48 arg instrs, SGPR operands"] = only("sgpr_fma48")
variants["P 48 arg instrs, VGPR operands"] = only("vgpr_fma48")

clob = ",".join(f'"v{i}"' for i in range(20, 240)) + ',"s24","s38","s39","s40","s41"'

src = ['// GENERATED by tools/ubench3_gen.py -- do not edit.  hipcc -O2 --offload-arch=gfx950 tools/ubench3.hip -o tools/ubench3',
       '#include <hip/hip_runtime.h>', '#include <stdio.h>', '#include <stdlib.h>',
       '#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)',
       'constexpr int ITERS = 512;', 'extern __shared__ float dyn_lds[];']
init = "\\n".join([f"v_mov_b32 v{i}, 0x3a83126f" for i in range(20, 240)] +
                  ["s_mov_b32 s24, 0x3f000000", "s_mov_b32 s38, 0x3f000000", "s_mov_b32 s39, 0x3e800000",
                   "v_mov_b32 v200, 0x3f000000", "v_mov_b32 v201, 0x3e800000", "v_mov_b32 v202, 0x3f000000"] +
                  [f"v_mov_b32 v{i}, 0" for i in (170, 171, 172, 173, 174, 175, 181)])
for idx, (name, (ins, n)) in enumerate(variants.items()):
    text = "\\n".join(ins)
    src.append(f'''
__global__ __launch_bounds__(64) void k{idx}(unsigned long long* cyc, float* out) {{
  if (threadIdx.x == 9999) dyn_lds[threadIdx.x] = 1.0f;
  asm volatile("{init}" ::: {clob});
  unsigned long long t0 = __builtin_readcyclecounter();
  asm volatile("s_mov_b32 s40, %0\\n"
               "1:\\n"
               "{text}\\n"
               "s_sub_u32 s40, s40, 1\\n s_cmp_lg_u32 s40, 0\\n s_cbranch_scc1 1b\\n" :: "s"(ITERS) : {clob}, "scc");
  unsigned long long t1 = __builtin_readcyclecounter();
  float s;
  asm volatile("v_add_f32 %0, v170, v171\\n v_add_f32 %0, %0, v173\\n v_add_f32 %0, %0, v174\\n v_add_f32 %0, %0, v176\\n v_add_f32 %0, %0, v210\\n v_add_f32 %0, %0, v233" : "=v"(s) :: {clob});
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}}''')
src.append("struct V { const char* name; void (*fn)(unsigned long long*, float*); int n; };")
src.append("static V vs[] = {" + ",".join(f'{{"{name}", k{idx}, {n}}}' for idx, (name, (ins, n)) in enumerate(variants.items())) + "};")
src.append(r'''
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned long long* cyc; float* out;
  CHECK(hipMalloc(&cyc, 8 * cus * 16)); CHECK(hipMalloc(&out, 4));
  unsigned long long* h = (unsigned long long*)malloc(8 * cus * 16);
  printf("%-42s %6s | cycles per wave-instruction per SIMD at 1 / 2 waves per SIMD (and us per launch)\n", "variant", "instr");
  for (auto& v : vs) {
    printf("%-42s %6d |", v.name, v.n);
    for (int w = 1; w <= 2; ++w) {
      const int per_cu = 4 * w, blocks = cus * per_cu;
      const size_t lds = (size_t)(160 * 1024 / per_cu) - 512;      // caps residency at exactly per_cu workgroups per CU
      CHECK(hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(64), lds, 0, cyc, out); CHECK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(64), lds, 0, cyc, out);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost));
      double sum = 0; for (int i = 0; i < blocks; ++i) sum += (double)h[i];
      const double per_wave = sum / blocks;                        // cycles one wave needed for ITERS*n instr, sharing its SIMD with w-1 others
      printf("  %5.2f (%6.1f us)", per_wave / ((double)ITERS * v.n * w), ms * 1e3);
    }
    printf("\n");
  }
  return 0;
}''')
open(sys.argv[1] if len(sys.argv) > 1 else "tools/ubench3.hip", "w").write("\n".join(src) + "\n")
