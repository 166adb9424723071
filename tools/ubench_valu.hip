// Micro-benchmark: VALU issue rates on gfx950 for the instruction classes the deferred light loop uses.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float m = 1.0001f, c = 0.5f;
  const float2_ pm = {m, m}, pc = {c, c};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) {  // 8 independent v_fma_f32
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
      } else if (KIND == 1) {  // 4 independent v_pk_fma_f32 (8 fmas)
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pm), "v"(pc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pm), "v"(pc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pm), "v"(pc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pm), "v"(pc));
      } else if (KIND == 2) {  // 8 v_rsq_f32
        a0 = __builtin_amdgcn_rsqf(a0); a1 = __builtin_amdgcn_rsqf(a1); a2 = __builtin_amdgcn_rsqf(a2); a3 = __builtin_amdgcn_rsqf(a3);
        a4 = __builtin_amdgcn_rsqf(a4); a5 = __builtin_amdgcn_rsqf(a5); a6 = __builtin_amdgcn_rsqf(a6); a7 = __builtin_amdgcn_rsqf(a7);
      } else if (KIND == 3) {  // 4 fma + 4 rsq interleaved
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_amdgcn_rsqf(a1); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_amdgcn_rsqf(a3);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_amdgcn_rsqf(a5); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_amdgcn_rsqf(a7);
      } else if (KIND == 4) {  // 8 v_mul_f32
        a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m;
      } else if (KIND == 5) {  // 7 fma + 1 rsq
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_amdgcn_rsqf(a7);
      } else if (KIND == 6) {  // 8 fma with an SGPR operand
        float s;
        asm volatile("s_mov_b32 %0, 0x3f800347" : "=s"(s));
        a0 = __builtin_fmaf(a0, s, c); a1 = __builtin_fmaf(a1, s, c); a2 = __builtin_fmaf(a2, s, c); a3 = __builtin_fmaf(a3, s, c);
        a4 = __builtin_fmaf(a4, s, c); a5 = __builtin_fmaf(a5, s, c); a6 = __builtin_fmaf(a6, s, c); a7 = __builtin_fmaf(a7, s, c);
      } else if (KIND == 7) {  // 8 v_cndmask + cmp pairs
        a0 = a0 > c ? a0 * m : c; a1 = a1 > c ? a1 * m : c; a2 = a2 > c ? a2 * m : c; a3 = a3 > c ? a3 * m : c;
        a4 = a4 > c ? a4 * m : c; a5 = a5 > c ? a5 * m : c; a6 = a6 > c ? a6 * m : c; a7 = a7 > c ? a7 * m : c;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND>
void run(const char* name, int waves_per_simd, float* d, double ops_per_iter) {
  const int iters = 4096;
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: waves_per_simd * iters * 8(unroll) * ops_per_iter
  const double winstr = (double)waves_per_simd * iters * 8 * ops_per_iter;
  const double cyc = ms * 1e-3 * 2.4e9 / winstr;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cycles@2.4GHz per wave-instruction\n", name, waves_per_simd, ms, cyc);
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float) * 2);
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32 x8", w, d, 8);
    run<1>("v_pk_fma_f32 x4", w, d, 4);
    run<2>("v_rsq_f32 x8", w, d, 8);
    run<3>("fma/rsq alternating x8", w, d, 8);
    run<4>("v_mul_f32 x8", w, d, 8);
    run<5>("7 fma + 1 rsq", w, d, 8);
    run<6>("v_fma_f32 sgpr operand x8", w, d, 8);
    run<7>("cmp+mul+cndmask x8 (24 ops)", w, d, 24);
  }
  return 0;
}
