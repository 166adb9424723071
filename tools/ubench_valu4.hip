// Micro-benchmark 4: do a transcendental-only wave and a plain-VALU wave on the same SIMD overlap?
// 512-thread workgroups: waves 0-3 and 4-7 share SIMDs pairwise.  role A = v_mul stream, role B = v_rsq stream.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define MULS REP8("v_mul_f32 v10, v14, v19\n v_mul_f32 v11, v15, v20\n v_mul_f32 v12, v16, v21\n v_mul_f32 v13, v17, v22\n")
#define RSQS REP8("v_rsq_f32 v10, v14\n v_rsq_f32 v11, v15\n v_rsq_f32 v12, v16\n v_rsq_f32 v13, v17\n")
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v19","v20","v21","v22"

// mode 0: all waves mul; 1: all waves rsq; 2: waves 0-3 mul, 4-7 rsq; 3: waves 0-3 mul, 4-7 idle (exit)
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters, int mode, float seed) {
  asm volatile("v_mov_b32 v14, %0\n v_add_f32 v15, 1.0, v14\n v_add_f32 v16, 2.0, v14\n v_add_f32 v17, 4.0, v14\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n" :: "v"(seed + threadIdx.x) : CLOB);
  const int wave = threadIdx.x >> 6;
  const bool trans = (mode == 1) || (mode == 2 && wave >= 4);
  const bool idle = (mode == 3 && wave >= 4);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (!idle) {
    if (trans) { for (int i = 0; i < iters; ++i) asm volatile(RSQS ::: CLOB); }
    else       { for (int i = 0; i < iters; ++i) asm volatile(MULS ::: CLOB); }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

int main() {
  unsigned long long* d; hipMalloc(&d, 8 * 256 * 8 * 8);
  const int iters = 2000;
  for (int bpc : {1, 2}) {  // 512-thread blocks per CU: 2 or 4 waves per SIMD
    const int blocks = 256 * bpc, nw = blocks * 8;
    for (int mode = 0; mode < 4; ++mode) {
      hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, mode, 1.5f);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, mode, 1.5f);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(nw);
      hipMemcpy(h.data(), d, nw * 8, hipMemcpyDeviceToHost);
      std::vector<double> a, b;
      for (int w = 0; w < nw; ++w) ((w & 7) < 4 ? a : b).push_back((double)h[w] / (iters * 32.0));
      std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
      printf("blocks/CU=%d mode=%d  waves0-3: %.2f ticks/instr   waves4-7: %.2f ticks/instr\n", bpc, mode, a[a.size() / 2], b[b.size() / 2]);
    }
  }
  return 0;
}
