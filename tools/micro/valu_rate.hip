// valu_rate.hip — issue rate of individual VALU instructions on gfx950 (microbenchmark behind the traversal-loop choices).
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
// 8 independent dependency chains per lane, 8 waves per SIMD: what is measured is issue throughput, not latency.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kIters = 2048, kChains = 8;

#define DEF_KERNEL(NAME, ASM)                                                                      \
  __global__ void __launch_bounds__(256) NAME(float* out, float a, float b) {                      \
    float x[kChains];                                                                              \
    for (int c = 0; c < kChains; ++c) x[c] = threadIdx.x * 1e-3f + c;                              \
    float va = a + threadIdx.x * 1e-9f, vb = b + threadIdx.x * 1e-9f;                              \
    for (int i = 0; i < kIters; ++i) {                                                             \
      _Pragma("unroll") for (int c = 0; c < kChains; ++c) asm volatile(ASM : "+v"(x[c]) : "v"(va), "v"(vb)); \
    }                                                                                              \
    float s = 0; for (int c = 0; c < kChains; ++c) s += x[c];                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                \
  }

DEF_KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
DEF_KERNEL(k_add, "v_add_f32 %0, %0, %1")
DEF_KERNEL(k_min, "v_min_f32 %0, %0, %1")
DEF_KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
DEF_KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
DEF_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(k_cndmask64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
DEF_KERNEL(k_cmp_cnd_vcc, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")          // 2 instructions
DEF_KERNEL(k_cmp_cnd_sgpr, "v_cmp_lt_f32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[10:11]")  // 2 instructions
DEF_KERNEL(k_add_cnd_vcc, "v_add_f32 %0, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")            // 2 instructions
DEF_KERNEL(k_min_add, "v_min_f32 %0, %0, %1\n v_add_f32 %0, %0, %2")                        // 2 instructions
DEF_KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")
DEF_KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")
DEF_KERNEL(k_cvt_sdwa, "v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
DEF_KERNEL(k_addu, "v_add_u32 %0, %0, %1")
DEF_KERNEL(k_and, "v_and_b32 %0, %0, %1")
DEF_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
DEF_KERNEL(k_mov, "v_mov_b32 %0, %1")
DEF_KERNEL(k_rcp, "v_rcp_f32 %0, %0")
DEF_KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
DEF_KERNEL(k_mad_u32, "v_mad_u32_u24 %0, %0, %1, %2")
DEF_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 5")
DEF_KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
DEF_KERNEL(k_sub, "v_sub_f32 %0, %0, %1")
DEF_KERNEL(k_maxf, "v_max_f32 %0, %0, %1")
DEF_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
// r04 candidates: mixed-precision fma (f16 operands read in place: no conversion), byte conversions, sign-bit mask accumulation, 32-bit integer multiplies
DEF_KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %1, %0, %2 op_sel_hi:[1,0,0]")                        // src0 = f16 (low half of va), src1 / src2 f32
DEF_KERNEL(k_fma_mix_hi, "v_fma_mix_f32 %0, %1, %0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]")      // src0 = f16 (high half)
DEF_KERNEL(k_cvt_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
DEF_KERNEL(k_cvt_ubyte3, "v_cvt_f32_ubyte3 %0, %0")
DEF_KERNEL(k_cvt_f16, "v_cvt_f32_f16 %0, %0")
DEF_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 31")
DEF_KERNEL(k_or, "v_or_b32 %0, %0, %1")
DEF_KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
DEF_KERNEL(k_addc, "v_addc_co_u32 %0, vcc, %0, %0, vcc")
DEF_KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
DEF_KERNEL(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
DEF_KERNEL(k_lshrrev, "v_lshrrev_b32 %0, %1, %0")
DEF_KERNEL(k_rsq, "v_rsq_f32 %0, %0")
DEF_KERNEL(k_sin, "v_sin_f32 %0, %0")
DEF_KERNEL(k_div_scale, "v_div_scale_f32 %0, vcc, %0, %1, %2")
DEF_KERNEL(k_div_fmas, "v_div_fmas_f32 %0, %0, %1, %2")
DEF_KERNEL(k_div_fixup, "v_div_fixup_f32 %0, %0, %1, %2")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")

template <class K> void run(const char* name, K kern) {
  const int blocks = 256 * 8;  // 256 CUs x 8 blocks of 4 waves = 8 waves per SIMD
  float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 256>>>(out, 1.0001f, 0.5f);
  hipEventRecord(e0);
  kern<<<blocks, 256>>>(out, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wave_insts = double(blocks) * 4 * kIters * kChains;
  printf("%-22s %.3f ms  %.2f wave64-instr / SIMD / ns\n", name, ms, wave_insts / ms * 1e-6 / 1024);
  hipFree(out);
}
#define RUN(K) run(#K, K)
int main() {
  RUN(k_fma); RUN(k_fmac); RUN(k_mul); RUN(k_add); RUN(k_sub); RUN(k_min); RUN(k_maxf); RUN(k_max3); RUN(k_med3); RUN(k_cndmask); RUN(k_cndmask64); RUN(k_cmp); RUN(k_cmp_cnd_vcc); RUN(k_cmp_cnd_sgpr); RUN(k_add_cnd_vcc); RUN(k_min_add);
  RUN(k_cvt); RUN(k_cvt_sdwa); RUN(k_addu); RUN(k_and); RUN(k_xor); RUN(k_lshl_add); RUN(k_mov); RUN(k_rcp); RUN(k_sqrt); RUN(k_mad_u32); RUN(k_bfe);
  RUN(k_fma_mix); RUN(k_fma_mix_hi); RUN(k_cvt_ubyte0); RUN(k_cvt_ubyte3); RUN(k_cvt_f16); RUN(k_alignbit); RUN(k_or); RUN(k_or3); RUN(k_addc); RUN(k_mul_lo_u32); RUN(k_mul_hi_u32);
  RUN(k_mul_u32_u24); RUN(k_lshrrev); RUN(k_rsq); RUN(k_sin); RUN(k_div_scale); RUN(k_div_fmas); RUN(k_div_fixup); RUN(k_perm);
  return 0;
}
