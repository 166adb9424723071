// Micro-benchmark 2: VALU issue cost on gfx950 by operand source (VGPR count / register bank) and
// for transcendental ops mixed into a plain stream.  Cycles come from s_memtime inside the kernel,
// so the result does not depend on the clock the chip happens to hold.
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/ubench_valu2.hip -o tools/ubench_valu2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP4(x) x x x x

// Each body is 32 VALU instructions on explicit registers v10..v41.
#define BODY_FMA_3V_SAMEBANK  REP8("v_fma_f32 v10, v14, v18, v22\n v_fma_f32 v11, v15, v19, v23\n v_fma_f32 v12, v16, v20, v24\n v_fma_f32 v13, v17, v21, v25\n")
#define BODY_FMA_3V_DIFFBANK  REP8("v_fma_f32 v10, v14, v19, v24\n v_fma_f32 v11, v15, v20, v25\n v_fma_f32 v12, v16, v21, v26\n v_fma_f32 v13, v17, v22, v27\n")
#define BODY_FMA_2V1S         REP8("v_fma_f32 v10, v14, s20, v24\n v_fma_f32 v11, v15, s20, v25\n v_fma_f32 v12, v16, s20, v26\n v_fma_f32 v13, v17, s20, v27\n")
#define BODY_FMAC             REP8("v_fmac_f32 v10, v14, v19\n v_fmac_f32 v11, v15, v20\n v_fmac_f32 v12, v16, v21\n v_fmac_f32 v13, v17, v22\n")
#define BODY_MUL_2V           REP8("v_mul_f32 v10, v14, v19\n v_mul_f32 v11, v15, v20\n v_mul_f32 v12, v16, v21\n v_mul_f32 v13, v17, v22\n")
#define BODY_MUL_1V1S         REP8("v_mul_f32 v10, s20, v19\n v_mul_f32 v11, s20, v20\n v_mul_f32 v12, s20, v21\n v_mul_f32 v13, s20, v22\n")
#define BODY_MUL_DEP          REP8("v_mul_f32 v10, v10, v19\n v_mul_f32 v10, v10, v20\n v_mul_f32 v10, v10, v21\n v_mul_f32 v10, v10, v22\n")
#define BODY_MUL_DEP2         REP8("v_mul_f32 v10, v10, v19\n v_mul_f32 v11, v11, v20\n v_mul_f32 v10, v10, v21\n v_mul_f32 v11, v11, v22\n")
#define BODY_RSQ              REP8("v_rsq_f32 v10, v14\n v_rsq_f32 v11, v15\n v_rsq_f32 v12, v16\n v_rsq_f32 v13, v17\n")
#define BODY_RSQ_1IN8         REP4("v_mul_f32 v10, v14, v19\n v_mul_f32 v11, v15, v20\n v_mul_f32 v12, v16, v21\n v_rsq_f32 v13, v17\n v_mul_f32 v30, v14, v19\n v_mul_f32 v31, v15, v20\n v_mul_f32 v32, v16, v21\n v_mul_f32 v33, v17, v22\n")
#define BODY_RSQ_1IN16        "v_rsq_f32 v13, v17\n" REP4("v_mul_f32 v10, v14, v19\n v_mul_f32 v11, v15, v20\n v_mul_f32 v12, v16, v21\n v_mul_f32 v30, v14, v19\n") "v_rsq_f32 v34, v17\n" REP4("v_mul_f32 v31, v15, v20\n v_mul_f32 v32, v16, v21\n v_mul_f32 v33, v17, v22\n") "v_mul_f32 v35, v17, v22\n v_mul_f32 v36, v17, v22\n"
#define BODY_SUB_SV           REP8("v_sub_f32 v10, s20, v19\n v_sub_f32 v11, s20, v20\n v_sub_f32 v12, s20, v21\n v_sub_f32 v13, s20, v22\n")
#define BODY_FMA_CLAMP        REP8("v_fma_f32 v10, v14, v19, v24 clamp\n v_fma_f32 v11, v15, v20, v25 clamp\n v_fma_f32 v12, v16, v21, v26 clamp\n v_fma_f32 v13, v17, v22, v27 clamp\n")
#define BODY_MAXMIN           REP8("v_max_f32 v10, v14, v19\n v_min_f32 v11, v15, v20\n v_max_f32 v12, v16, v21\n v_min_f32 v13, v17, v22\n")
#define BODY_ADD_2V           REP8("v_add_f32 v10, v14, v19\n v_add_f32 v11, v15, v20\n v_add_f32 v12, v16, v21\n v_add_f32 v13, v17, v22\n")

#define MUL4 "v_mul_f32 v10, v14, v19\n v_mul_f32 v11, v15, v20\n v_mul_f32 v12, v16, v21\n v_mul_f32 v30, v14, v19\n"
#define BODY_RSQ_2GRP_IN32    "v_rsq_f32 v13, v17\n v_rsq_f32 v34, v17\n" MUL4 MUL4 MUL4 MUL4 MUL4 MUL4 MUL4 "v_mul_f32 v35, v17, v22\n v_mul_f32 v36, v17, v22\n"
#define BODY_RSQ_4GRP_IN32    "v_rsq_f32 v13, v17\n v_rsq_f32 v34, v17\n v_rsq_f32 v35, v17\n v_rsq_f32 v36, v17\n" MUL4 MUL4 MUL4 MUL4 MUL4 MUL4 MUL4
#define BODY_RSQ_8GRP_IN32    "v_rsq_f32 v13, v17\n v_rsq_f32 v34, v17\n v_rsq_f32 v35, v17\n v_rsq_f32 v36, v17\n v_rsq_f32 v13, v16\n v_rsq_f32 v34, v16\n v_rsq_f32 v35, v16\n v_rsq_f32 v36, v16\n" MUL4 MUL4 MUL4 MUL4 MUL4 MUL4
#define BODY_RSQ_4SPREAD_IN32 "v_rsq_f32 v13, v17\n" MUL4 MUL4 "v_rsq_f32 v34, v17\n" MUL4 MUL4 "v_rsq_f32 v35, v17\n" MUL4 MUL4 "v_rsq_f32 v36, v17\n" MUL4
#define BODY_MIX_SQRT_RCP     "v_sqrt_f32 v13, v17\n v_rcp_f32 v34, v17\n v_rsq_f32 v35, v17\n v_sqrt_f32 v36, v17\n" MUL4 MUL4 MUL4 MUL4 MUL4 MUL4 MUL4
#define BODY_SNOP_AFTER       "v_rsq_f32 v13, v17\n s_nop 7\n" MUL4 MUL4 MUL4 MUL4 MUL4 MUL4 MUL4 "v_mul_f32 v35, v17, v22\n v_mul_f32 v36, v17, v22\n v_mul_f32 v34, v17, v22\n"
#define BODY_CVT              REP8("v_cvt_f32_u32 v10, v14\n v_cvt_f32_u32 v11, v15\n v_cvt_f32_u32 v12, v16\n v_cvt_f32_u32 v13, v17\n")
#define BODY_MUL_LIT          REP8("v_mul_f32 v10, 0x3f800347, v19\n v_mul_f32 v11, 0x3f800347, v20\n v_mul_f32 v12, 0x3f800347, v21\n v_mul_f32 v13, 0x3f800347, v22\n")
#define BODY_MUL_INLINE       REP8("v_mul_f32 v10, 0.5, v19\n v_mul_f32 v11, 0.5, v20\n v_mul_f32 v12, 0.5, v21\n v_mul_f32 v13, 0.5, v22\n")
#define BODY_FMA_2V_INL       REP8("v_fma_f32 v10, v14, v19, 1.0\n v_fma_f32 v11, v15, v20, 1.0\n v_fma_f32 v12, v16, v21, 1.0\n v_fma_f32 v13, v17, v22, 1.0\n")
#define BODY_MED3             REP8("v_med3_f32 v10, v14, v19, v24\n v_med3_f32 v11, v15, v20, v25\n v_med3_f32 v12, v16, v21, v26\n v_med3_f32 v13, v17, v22, v27\n")
#define BODY_CNDMASK          REP8("v_cndmask_b32 v10, v14, v19, vcc\n v_cndmask_b32 v11, v15, v20, vcc\n v_cndmask_b32 v12, v16, v21, vcc\n v_cndmask_b32 v13, v17, v22, vcc\n")
#define BODY_CMP              REP8("v_cmp_lt_f32 vcc, v14, v19\n v_cmp_lt_f32 vcc, v15, v20\n v_cmp_lt_f32 vcc, v16, v21\n v_cmp_lt_f32 vcc, v17, v22\n")
#define BODY_LDS_MUL          "ds_read_b128 v[40:43], v44\n ds_read_b128 v[46:49], v44 offset:16\n" MUL4 MUL4 MUL4 MUL4 MUL4 MUL4 MUL4 MUL4 "s_waitcnt lgkmcnt(0)\n"

#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v30","v31","v32","v33","v34","v35","v36","v40","v41","v42","v43","v44","v46","v47","v48","v49","vcc","s20"

#define KERNEL(NAME, BODY)                                                                 \
  __global__ __launch_bounds__(256) void NAME(unsigned long long* out, int iters) {        \
    asm volatile("s_mov_b32 s20, 0x3f800347\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n" \
                 "v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n"        \
                 "v_mov_b32 v44, 0\n v_mov_b32 v23, 0.5\n v_mov_b32 v24, 0.5\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 0.5\n v_mov_b32 v27, 0.5\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n" ::: CLOB); \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                  \
    for (int i = 0; i < iters; ++i) asm volatile(BODY ::: CLOB);                           \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                  \
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
  }

KERNEL(k_fma_same, BODY_FMA_3V_SAMEBANK)
KERNEL(k_fma_diff, BODY_FMA_3V_DIFFBANK)
KERNEL(k_fma_2v1s, BODY_FMA_2V1S)
KERNEL(k_fmac, BODY_FMAC)
KERNEL(k_mul_2v, BODY_MUL_2V)
KERNEL(k_mul_1v1s, BODY_MUL_1V1S)
KERNEL(k_mul_dep, BODY_MUL_DEP)
KERNEL(k_mul_dep2, BODY_MUL_DEP2)
KERNEL(k_rsq, BODY_RSQ)
KERNEL(k_rsq_1in8, BODY_RSQ_1IN8)
KERNEL(k_rsq_1in16, BODY_RSQ_1IN16)
KERNEL(k_sub_sv, BODY_SUB_SV)
KERNEL(k_fma_clamp, BODY_FMA_CLAMP)
KERNEL(k_maxmin, BODY_MAXMIN)
KERNEL(k_add_2v, BODY_ADD_2V)
KERNEL(k_rsq_2grp, BODY_RSQ_2GRP_IN32)
KERNEL(k_rsq_4grp, BODY_RSQ_4GRP_IN32)
KERNEL(k_rsq_8grp, BODY_RSQ_8GRP_IN32)
KERNEL(k_rsq_4spread, BODY_RSQ_4SPREAD_IN32)
KERNEL(k_mix_trans, BODY_MIX_SQRT_RCP)
KERNEL(k_snop, BODY_SNOP_AFTER)
KERNEL(k_cvt, BODY_CVT)
KERNEL(k_mul_lit, BODY_MUL_LIT)
KERNEL(k_mul_inl, BODY_MUL_INLINE)
KERNEL(k_fma_inl, BODY_FMA_2V_INL)
KERNEL(k_med3, BODY_MED3)
KERNEL(k_cndmask, BODY_CNDMASK)
KERNEL(k_cmp, BODY_CMP)
KERNEL(k_lds_mul, BODY_LDS_MUL)

typedef void (*kern_t)(unsigned long long*, int);

void run(const char* name, kern_t k, int waves_per_simd, unsigned long long* d, int ninstr) {
  const int iters = 2000;
  const int blocks = 256 * waves_per_simd;
  const int nwaves = blocks * 4;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nwaves);
  hipMemcpy(h.data(), d, nwaves * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[nwaves / 2];
  printf("%-22s w/SIMD=%d  per-wave %.2f cyc/instr  -> SIMD throughput %.2f cyc/instr\n", name, waves_per_simd,
         med / ((double)iters * ninstr), med / ((double)iters * ninstr) / waves_per_simd);
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 8 * 256 * 8 * 4 * 2);
  for (int w : {2, 4}) {
    run("fma 3v same bank", k_fma_same, w, d, 32);
    run("fma 3v diff bank", k_fma_diff, w, d, 32);
    run("fma 2v+1s", k_fma_2v1s, w, d, 32);
    run("fmac (2v + acc)", k_fmac, w, d, 32);
    run("mul 2v", k_mul_2v, w, d, 32);
    run("mul 1v+1s", k_mul_1v1s, w, d, 32);
    run("mul dependent chain", k_mul_dep, w, d, 32);
    run("mul 2 dep chains", k_mul_dep2, w, d, 32);
    run("rsq", k_rsq, w, d, 32);
    run("rsq 1 in 8 (mul)", k_rsq_1in8, w, d, 32);
    run("rsq 1 in 16 (mul)", k_rsq_1in16, w, d, 32);
    run("sub s,v", k_sub_sv, w, d, 32);
    run("fma clamp", k_fma_clamp, w, d, 32);
    run("max/min", k_maxmin, w, d, 32);
    run("add 2v", k_add_2v, w, d, 32);
    run("2 rsq grouped + 30 mul", k_rsq_2grp, w, d, 32);
    run("4 rsq grouped + 28 mul", k_rsq_4grp, w, d, 32);
    run("8 rsq grouped + 24 mul", k_rsq_8grp, w, d, 32);
    run("4 rsq spread + 28 mul", k_rsq_4spread, w, d, 32);
    run("sqrt,rcp,rsq,sqrt + 28", k_mix_trans, w, d, 32);
    run("rsq, s_nop 7, 31 mul", k_snop, w, d, 32);
    run("cvt_f32_u32", k_cvt, w, d, 32);
    run("mul literal,v", k_mul_lit, w, d, 32);
    run("mul inline 0.5,v", k_mul_inl, w, d, 32);
    run("fma v,v,inline 1.0", k_fma_inl, w, d, 32);
    run("med3 3v", k_med3, w, d, 32);
    run("cndmask", k_cndmask, w, d, 32);
    run("cmp -> vcc", k_cmp, w, d, 32);
    run("2 ds_read_b128 + 32 mul", k_lds_mul, w, d, 32);
    printf("\n");
  }
  return 0;
}
