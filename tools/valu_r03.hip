// valu_r03.hip -- round 3: the instruction floor of the NTT butterfly, measured before anything is built on it.
// (a) single-instruction streams for the candidates VERDICT r2 item 7 names and a few more (16-bit mads, dot2,
//     packed 16-bit, 24-bit and 32-bit multiplies, SDWA adds, packed fp32);
// (b) butterfly-shaped streams, four butterflies per group as one Falcon-512 stage has per lane, grouped by
//     instruction kind as verify_kernel.hpp emits them, at 1..8 waves per SIMD with the waves out of phase:
//       shoup      v_mul_hi_i32, v_mul_i32_i24, v_mad_i32_i24, v_sub, v_add          (5; today's butterfly)
//       plantard   v_mul_lo_u32, v_mul_hi_u32, v_sub, v_add                          (4; word-size modular
//                  multiplication after Plantard: t = hi32((x * W mod 2^32) * q), W = w * (-2^32) * q^-1)
//       plantard16 v_mul_lo_u32, v_mad_u32_u16 op_sel (high half), v_add_sdwa, v_sub_sdwa   (4; result in the high half)
//       packed16   two butterflies per lane: 2 x v_mad_u32_u16 + v_perm (the missing packed mul_hi), v_pk_mul_lo_u16,
//                  v_pk_mad_u16, v_pk_add_u16, v_pk_sub_u16                           (7 per two butterflies)
//       dot2       v_mul_lo_u32, v_bfi, v_dot2_i32_i16, v_sub, v_add                  (5)
//     The figure of merit is SIMD cycles per BUTTERFLY (at the nominal clock), not per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_r03 tools/valu_r03.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITERS 2048
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
#define DEFKERNEL64(NAME, ASM)                                                        \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)           \
  {                                                                                   \
    double x[CHAINS], a = seed + threadIdx.x, b = seed * 3.0 + 7.0;                    \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) x[c] = a + c;                  \
    for (int i = 0; i < ITERS; i++) {                                                 \
      _Pragma("unroll") for (int c = 0; c < CHAINS; c++)                              \
        asm volatile(ASM : "+v"(x[c]) : "v"(a), "v"(b));                              \
    }                                                                                 \
    double s = 0;                                                                     \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) s += x[c];                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;                         \
  }

// v_mad_u64_u32: the whole 32 x 32 -> 64 product plus a 64-bit addend; the chain runs through the low word of the result.
// ADDEND: "v" = a vector register pair, "s" = a scalar pair (wave-uniform addend, as the verify kernel's epilogue has it).
#define DEFKERNEL_MAD64(NAME, ADDEND)                                                 \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)           \
  {                                                                                   \
    uint64_t x[CHAINS];                                                               \
    uint32_t a = seed + threadIdx.x, b = seed * 3u + 7u;                              \
    uint64_t cs = ((uint64_t)seed << 32) | 12345u;                                    \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) x[c] = a + c;                  \
    for (int i = 0; i < ITERS; i++) {                                                 \
      _Pragma("unroll") for (int c = 0; c < CHAINS; c++) {                            \
        const uint32_t lo = (uint32_t)x[c];                                           \
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(x[c]) : "v"(lo), "v"(b), ADDEND(cs) : "vcc"); \
      }                                                                               \
    }                                                                                 \
    uint64_t s = 0;                                                                   \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) s ^= x[c];                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);   \
  }
DEFKERNEL_MAD64(k_mad64_v, "v")
DEFKERNEL_MAD64(k_mad64_s, "s")

DEFKERNEL(k_add, "v_add_u32 %0, %0, %1")
DEFKERNEL(k_sub, "v_sub_u32 %0, %0, %1")
DEFKERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
DEFKERNEL(k_mulhi_u24, "v_mul_hi_u32_u24 %0, %0, %1")
DEFKERNEL(k_mul_i24, "v_mul_i32_i24 %0, %0, %1")
DEFKERNEL(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
DEFKERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEFKERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
DEFKERNEL(k_mul_hi_i, "v_mul_hi_i32 %0, %0, %1")
DEFKERNEL(k_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %2")
DEFKERNEL(k_mad_u32_u16_hi, "v_mad_u32_u16 %0, %0, %1, %2 op_sel:[1,0,0,0]")
DEFKERNEL(k_mad_i32_i16, "v_mad_i32_i16 %0, %0, %1, %2")
DEFKERNEL(k_dot2_i32_i16, "v_dot2_i32_i16 %0, %0, %1, %2")
DEFKERNEL(k_dot2_u32_u16, "v_dot2_u32_u16 %0, %0, %1, %2")
DEFKERNEL(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
DEFKERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
DEFKERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEFKERNEL(k_pk_sub_u16, "v_pk_sub_u16 %0, %0, %1")
DEFKERNEL(k_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
DEFKERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
DEFKERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
DEFKERNEL(k_sub_sdwa, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
DEFKERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEFKERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
DEFKERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
DEFKERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEFKERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEFKERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1")
DEFKERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
DEFKERNEL64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
DEFKERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")

// ---- butterfly-shaped streams: NB butterflies per group, grouped by instruction kind ----
enum { SHOUP = 0, PLANTARD = 1, PLANTARD16 = 2, PACKED16 = 3, DOT2 = 4, PLANTARD_BIAS = 5, NKINDS = 6 };

template <int KIND, int NB>
__global__ __launch_bounds__(256) void k_bfly(uint32_t* out, uint32_t seed)
{
  uint32_t x[16], y[16], t[16], u[16], a = seed + threadIdx.x, b = seed * 3u + 7u;
#pragma unroll
  for (int c = 0; c < 16; c++) { x[c] = a + c; y[c] = b + c; t[c] = c; u[c] = c + 1; }
  {  // waves of one SIMD start out of phase, as in a real kernel
    const int extra = ((blockIdx.x * 4 + (threadIdx.x >> 6)) % 13) * 37;
    for (int i = 0; i < extra; i++) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(t[0]) : "v"(a));
  }
  uint32_t sq;  // a wave-uniform operand, as q is in the kernels
  asm volatile("s_mov_b32 %0, 12289" : "=s"(sq));
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int g = 0; g < 16 / NB; g++) {
      const int c0 = g * NB, c1 = (g + 1) * NB;
      if constexpr (KIND == SHOUP) {
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_mul_hi_i32 %0, %1, %2" : "+v"(t[c]) : "v"(y[c]), "v"(a));
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_mul_i32_i24 %0, %1, %2" : "+v"(u[c]) : "v"(y[c]), "v"(b));
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_mad_i32_i24 %0, %0, %2, %1" : "+v"(t[c]) : "v"(u[c]), "s"(sq));
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_sub_u32 %0, %1, %2" : "+v"(y[c]) : "v"(x[c]), "v"(t[c]));
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(t[c]));
      } else if constexpr (KIND == PLANTARD || KIND == PLANTARD_BIAS) {
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_mul_lo_u32 %0, %1, %2" : "+v"(t[c]) : "v"(y[c]), "v"(a));
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t[c]) : "s"(sq));
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_sub_u32 %0, %1, %2" : "+v"(y[c]) : "v"(x[c]), "v"(t[c]));
        if constexpr (KIND == PLANTARD_BIAS) {  // the Gentleman-Sande form: one more full-rate add per butterfly
#pragma unroll
          for (int c = c0; c < c1; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(y[c]) : "s"(sq));
        }
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(t[c]));
      } else if constexpr (KIND == PLANTARD16) {
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_mul_lo_u32 %0, %1, %2" : "+v"(t[c]) : "v"(y[c]), "v"(a));
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_mad_u32_u16 %0, %0, %1, %1 op_sel:[1,0,0,0]" : "+v"(t[c]) : "s"(sq));
#pragma unroll
        for (int c = c0; c < c1; c++)
          asm volatile("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(y[c]) : "v"(x[c]), "v"(t[c]));
#pragma unroll
        for (int c = c0; c < c1; c++)
          asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(x[c]) : "v"(t[c]));
      } else if constexpr (KIND == PACKED16) {
        // registers c and c+1 of a pair stand for ONE packed register each way: 7 instructions per pair = 2 butterflies
#pragma unroll
        for (int c = c0; c < c1; c += 2) asm volatile("v_mad_u32_u16 %0, %1, %2, 0" : "+v"(t[c]) : "v"(y[c]), "v"(a));
#pragma unroll
        for (int c = c0; c < c1; c += 2) asm volatile("v_mad_u32_u16 %0, %1, %2, 0 op_sel:[1,1,0,0]" : "+v"(t[c + 1]) : "v"(y[c]), "v"(a));
#pragma unroll
        for (int c = c0; c < c1; c += 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(t[c]) : "v"(t[c + 1]), "v"(b));
#pragma unroll
        for (int c = c0; c < c1; c += 2) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(t[c]) : "v"(b));
#pragma unroll
        for (int c = c0; c < c1; c += 2) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(t[c]) : "v"(y[c]), "v"(a));
#pragma unroll
        for (int c = c0; c < c1; c += 2) asm volatile("v_pk_sub_u16 %0, %1, %2" : "+v"(y[c]) : "v"(x[c]), "v"(t[c]));
#pragma unroll
        for (int c = c0; c < c1; c += 2) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[c]) : "v"(t[c]));
      } else {  // DOT2
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_mul_lo_u32 %0, %1, %2" : "+v"(t[c]) : "v"(y[c]), "v"(a));
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(t[c]) : "v"(y[c]), "v"(b));
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_dot2_i32_i16 %0, %0, %1, 0" : "+v"(t[c]) : "v"(a));
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_sub_u32 %0, %1, %2" : "+v"(y[c]) : "v"(x[c]), "v"(t[c]));
#pragma unroll
        for (int c = c0; c < c1; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(t[c]));
      }
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < 16; c++) s ^= x[c] ^ y[c] ^ t[c] ^ u[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; };
struct Mix { const char* name; kern_t k; double instr_per_bfly; };

static float time_kernel(kern_t k, int blocks, int threads, uint32_t* d, hipEvent_t e0, hipEvent_t e1)
{
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 1u);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 2u + rep);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main()
{
  Entry list[] = { {"v_add_u32", k_add}, {"v_sub_u32", k_sub}, {"v_mul_u32_u24", k_mul_u24}, {"v_mul_hi_u32_u24", k_mulhi_u24},
                   {"v_mul_i32_i24", k_mul_i24}, {"v_mad_i32_i24", k_mad_i24}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi},
                   {"v_mul_hi_i32", k_mul_hi_i}, {"v_mad_u32_u16", k_mad_u32_u16}, {"v_mad_u32_u16 op_sel:[1,0,0,0]", k_mad_u32_u16_hi},
                   {"v_mad_i32_i16", k_mad_i32_i16}, {"v_dot2_i32_i16", k_dot2_i32_i16}, {"v_dot2_u32_u16", k_dot2_u32_u16},
                   {"v_pk_mul_lo_u16", k_pk_mul_lo_u16}, {"v_pk_mad_u16", k_pk_mad_u16}, {"v_pk_add_u16", k_pk_add_u16},
                   {"v_pk_sub_u16", k_pk_sub_u16}, {"v_pk_sub_i16", k_pk_sub_i16}, {"v_pk_min_u16", k_pk_min_u16},
                   {"v_add_u32_sdwa src1:WORD_1", k_add_sdwa}, {"v_sub_u32_sdwa src1:WORD_1", k_sub_sdwa}, {"v_perm_b32", k_perm},
                   {"v_bfi_b32", k_bfi}, {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3}, {"v_fma_f32", k_fma_f32},
                   {"v_mul_f32", k_mul_f32}, {"v_pk_fma_f32 (2 per lane)", k_pk_fma_f32}, {"v_pk_mul_f32 (2 per lane)", k_pk_mul_f32},
                   {"v_pk_add_f32 (2 per lane)", k_pk_add_f32},
                   {"v_mad_u64_u32 (64-bit result, vector addend)", k_mad64_v}, {"v_mad_u64_u32 (64-bit result, scalar addend)", k_mad64_s} };
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const double clk = prop.clockRate * 1e3;
  const int simds = prop.multiProcessorCount * 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  uint32_t* d;
  {
    const int blocks = prop.multiProcessorCount * 8, threads = 256;
    (void)hipMalloc(&d, (size_t)blocks * threads * 4);
    printf("{\"clock_MHz\": %.0f, \"simds\": %d,\n \"single_streams_8_waves_per_simd\": {\"unit\": \"T wave-instruction-lanes/s\", \"rates\": {", clk / 1e6, simds);
    bool first = true;
    for (auto& e : list) {
      const float ms = time_kernel(e.k, blocks, threads, d, e0, e1);
      const double ops = (double)blocks * threads * ITERS * CHAINS;
      printf("%s\n  \"%s\": %.2f", first ? "" : ",", e.name, ops / (ms * 1e-3) / 1e12);
      first = false;
    }
    printf("}},\n");
    (void)hipFree(d);
  }
  Mix mixes[] = { {"shoup (mul_hi_i32, mul_i24, mad_i24, sub, add)", k_bfly<SHOUP, 4>, 5.0},
                  {"plantard (mul_lo_u32, mul_hi_u32, sub, add)", k_bfly<PLANTARD, 4>, 4.0},
                  {"plantard + bias add (Gentleman-Sande form)", k_bfly<PLANTARD_BIAS, 4>, 5.0},
                  {"plantard16 (mul_lo_u32, mad_u32_u16 hi, add_sdwa, sub_sdwa)", k_bfly<PLANTARD16, 4>, 4.0},
                  {"packed16 (2 mad_u32_u16, perm, pk_mul_lo, pk_mad, pk_sub, pk_add per TWO butterflies)", k_bfly<PACKED16, 4>, 3.5},
                  {"dot2 (mul_lo_u32, bfi, dot2_i32_i16, sub, add)", k_bfly<DOT2, 4>, 5.0},
                  {"shoup, 8 per group", k_bfly<SHOUP, 8>, 5.0},
                  {"plantard, 8 per group", k_bfly<PLANTARD, 8>, 4.0} };
  printf(" \"butterfly_streams\": {\"unit\": \"SIMD cycles per butterfly at the nominal clock\", \"results\": [");
  bool first = true;
  for (int wps : { 1, 2, 3, 4, 5, 8 }) {
    const int threads = 256, blocks = prop.multiProcessorCount * wps;
    (void)hipMalloc(&d, (size_t)blocks * threads * 4);
    for (auto& m : mixes) {
      const float ms = time_kernel(m.k, blocks, threads, d, e0, e1);
      const double bfly = (double)blocks * (threads / 64) * ITERS * 16.0 * (m.instr_per_bfly == 3.5 ? 1.0 : 1.0);
      // packed16: 16 registers stand for 8 packed pairs = 16 butterflies as well (two per pair)
      const double cyc = ms * 1e-3 * clk * simds / bfly;
      printf("%s\n  {\"waves_per_simd\": %d, \"stream\": \"%s\", \"instr_per_butterfly\": %.1f, \"cycles_per_butterfly\": %.2f, \"cycles_per_instr\": %.2f}",
             first ? "" : ",", wps, m.name, m.instr_per_bfly, cyc, cyc / m.instr_per_bfly);
      first = false;
    }
    (void)hipFree(d);
  }
  printf("\n ]}\n}\n");
  return 0;
}
