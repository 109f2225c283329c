// valu_peak.hip -- measures the sustained issue rate of the integer VALU instructions the
// verifier is built from (lane-ops/s over the whole chip), with 8 independent chains per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_peak tools/valu_peak.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096
#define CHAINS 8

#define DEFKERNEL(NAME, ASM)                                                          \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)           \
  {                                                                                   \
    uint32_t x[CHAINS], a = seed + threadIdx.x, b = seed * 3u + 7u;                    \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) x[c] = a + c;                  \
    for (int i = 0; i < ITERS; i++) {                                                 \
      _Pragma("unroll") for (int c = 0; c < CHAINS; c++)                              \
        asm volatile(ASM : "+v"(x[c]) : "v"(a), "v"(b));                              \
    }                                                                                 \
    uint32_t s = 0;                                                                   \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) s ^= x[c];                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                   \
  }

DEFKERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEFKERNEL(k_add, "v_add_u32 %0, %0, %1")
DEFKERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEFKERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
DEFKERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
DEFKERNEL(k_mulhi_u24, "v_mul_hi_u32_u24 %0, %0, %1")
DEFKERNEL(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
DEFKERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEFKERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEFKERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
DEFKERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEFKERNEL(k_min, "v_min_u32 %0, %0, %1")
DEFKERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEFKERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
DEFKERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFKERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEFKERNEL(k_permlane32, "v_permlane32_swap_b32 %0, %1")
DEFKERNEL(k_ffbh, "v_ffbh_u32 %0, %0")
DEFKERNEL(k_sub, "v_sub_u32 %0, %0, %1")
DEFKERNEL(k_and, "v_and_b32 %0, %0, %1")
DEFKERNEL(k_or, "v_or_b32 %0, %0, %1")
DEFKERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
DEFKERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0")
DEFKERNEL(k_max, "v_max_u32 %0, %0, %1")
DEFKERNEL(k_mov, "v_mov_b32 %0, %1")
DEFKERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 14")
DEFKERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEFKERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
DEFKERNEL(k_cndmask_s, "v_cndmask_b32 %0, %0, %1, s[4:5]")
DEFKERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %1\n\tv_add_u32 %0, %0, %2")
DEFKERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEFKERNEL(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
DEFKERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
DEFKERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
DEFKERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEFKERNEL(k_add_f32, "v_add_f32 %0, %0, %1")
DEFKERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1")
DEFKERNEL(k_mul_u24_sdwa, "v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEFKERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEFKERNEL(k_sub_min, "v_sub_u32 %0, %0, %1\n\tv_min_u32 %0, %0, %2")
DEFKERNEL(k_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %2")
DEFKERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
DEFKERNEL(k_mov_dpp_ror8, "v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3")
DEFKERNEL(k_add_dpp, "v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; };

int main()
{
  Entry list[] = { {"v_xor_b32", k_xor}, {"v_add_u32", k_add}, {"v_bitop3_b32", k_bitop3}, {"v_alignbit_b32", k_alignbit},
                   {"v_mul_u32_u24", k_mul_u24}, {"v_mul_hi_u32_u24", k_mulhi_u24}, {"v_mad_i32_i24", k_mad_i24},
                   {"v_mad_u32_u24", k_mad_u24}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi},
                   {"v_add3_u32", k_add3}, {"v_min_u32", k_min}, {"v_perm_b32", k_perm}, {"v_lshl_add_u32", k_lshl_add},
                   {"v_cndmask_b32", k_cndmask}, {"v_mov_b32_dpp", k_mov_dpp}, {"v_permlane32_swap", k_permlane32},
                   {"v_ffbh_u32", k_ffbh}, {"v_sub_u32", k_sub}, {"v_and_b32", k_and}, {"v_or_b32", k_or},
                   {"v_lshlrev_b32", k_lshl}, {"v_lshrrev_b32", k_lshr}, {"v_max_u32", k_max}, {"v_mov_b32", k_mov},
                   {"v_bfe_u32", k_bfe}, {"v_and_or_b32", k_and_or}, {"v_xad_u32", k_xad}, {"v_cndmask_b32(sgpr)", k_cndmask_s},
                   {"v_cmp+v_add (2 ops)", k_cmp}, {"v_pk_add_u16", k_pk_add_u16}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16},
                   {"v_pk_mad_u16", k_pk_mad_u16}, {"v_pk_min_u16", k_pk_min_u16}, {"v_fma_f32", k_fma_f32},
                   {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_mul_u32_u24_sdwa", k_mul_u24_sdwa},
                   {"v_add_u32_sdwa", k_add_sdwa}, {"v_sub+v_min (2 ops)", k_sub_min}, {"v_mad_u32_u16", k_mad_u32_u16},
                   {"v_lshl_or_b32", k_lshl_or}, {"v_mov_dpp row_ror:8 bank_mask", k_mov_dpp_ror8}, {"v_add_u32_dpp", k_add_dpp} };
  const int blocks = 256 * 8, threads = 256;
  uint32_t* d;
  hipMalloc(&d, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("{\"blocks\": %d, \"threads\": %d, \"iters\": %d, \"chains\": %d, \"rates_Tops\": {", blocks, threads, ITERS, CHAINS);
  bool first = true;
  for (auto& e : list) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 1u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 2u + rep);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double ops = (double)blocks * threads * ITERS * CHAINS;
    printf("%s\"%s\": %.2f", first ? "" : ", ", e.name, ops / (best * 1e-3) / 1e12);
    first = false;
  }
  printf("}}\n");
  return 0;
}
