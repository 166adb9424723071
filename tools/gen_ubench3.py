"""Generates tools/ubench_valu3.hip: transcendental grouping patterns at the deferred loop's own
ratio (16 transcendentals per 240 plain VALU instructions), with and without consumers."""
import itertools

def body(group, dep, ntrans=16, nplain=240, distinct_src=True):
    ngroups = ntrans // group
    per = nplain // ngroups
    lines = []
    t = 0
    for g in range(ngroups):
        outs = []
        for k in range(group):
            dst = 40 + (t % 16)
            src = 14 + (t % 4) if distinct_src else 17
            lines.append(f"v_rsq_f32 v{dst}, v{src}")
            outs.append(dst)
            t += 1
        lines.append("s_nop 0")
        for i in range(per):
            d = 10 + (i % 4) if i % 8 < 4 else 30 + (i % 4)
            if dep and i < len(outs):
                lines.append(f"v_mul_f32 v{60 + (i % 4)}, v{outs[i]}, v{19 + (i % 4)}")
            else:
                lines.append(f"v_mul_f32 v{d}, v{14 + (i % 4)}, v{19 + (i % 4)}")
    return "\\n ".join(lines) + "\\n"

variants = []
for group in (1, 2, 4, 8, 16):
    for dep in (0, 1):
        variants.append((f"g{group}_dep{dep}", body(group, dep)))
variants.append(("plain256", "\\n ".join(f"v_mul_f32 v{10 + (i % 4)}, v{14 + (i % 4)}, v{19 + (i % 4)}" for i in range(256)) + "\\n"))

clob = ",".join(f'"v{i}"' for i in list(range(10, 28)) + list(range(30, 34)) + list(range(40, 56)) + list(range(60, 64)))
src = ['#include <hip/hip_runtime.h>', '#include <algorithm>', '#include <cstdio>', '#include <vector>', f'#define CLOB {clob}']
for name, b in variants:
    src.append(f'''__global__ __launch_bounds__(256) void k_{name}(unsigned long long* out, int iters, float seed) {{
  asm volatile("v_mov_b32 v14, %0\\n v_add_f32 v15, 1.0, v14\\n v_add_f32 v16, 2.0, v14\\n v_add_f32 v17, 4.0, v14\\n v_mov_b32 v19, 1.0\\n v_mov_b32 v20, 1.0\\n v_mov_b32 v21, 1.0\\n v_mov_b32 v22, 1.0\\n" :: "v"(seed + threadIdx.x) : CLOB);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) asm volatile("{b}" ::: CLOB);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}}''')
src.append('''typedef void (*kern_t)(unsigned long long*, int, float);
void run(const char* name, kern_t k, int w, unsigned long long* d) {
  const int iters = 2000, blocks = 256 * w, nwaves = blocks * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwaves);
  hipMemcpy(h.data(), d, nwaves * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[nwaves / 2] / iters / w;  // SIMD cycles per body
  printf("%-12s w=%d  ticks per body: %.1f   wall ns per body per SIMD: %.1f\\n", name, w, med, ms * 1e6 / (iters * (double)w));
}
int main() {
  unsigned long long* d; hipMalloc(&d, 8 * 256 * 8 * 4 * 2);
  for (int w : {2, 4, 6, 8}) {''')
for name, _ in variants:
    src.append(f'    run("{name}", k_{name}, w, d);')
src.append('    printf("\\n");\n  }\n  return 0;\n}')
open("tools/ubench_valu3.hip", "w").write("\n".join(src) + "\n")
