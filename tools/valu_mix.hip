// valu_mix.hip -- does a MIXED stream of half-rate (multiply) and full-rate (add/sub) vector instructions issue
// at the sum of their separate costs (4 + 2 cycles per wave64), or does every instruction take a 4-cycle slot?
// Streams shaped like the NTT butterfly of verify_kernel.hpp (mul_hi, mul_i24, mad_i24, sub, add), in the order
// the compiler emits them (chain after chain) and regrouped (all multiplies of four butterflies, then all
// adds/subs), beside pure streams. Prints lane-ops/s over the chip and cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_mix tools/valu_mix.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITERS 2048

#define HEAD                                                                       \
  uint32_t x[8], y[8], t[8], a = seed + threadIdx.x, b = seed * 3u + 7u;           \
  _Pragma("unroll") for (int c = 0; c < 8; c++) { x[c] = a + c; y[c] = b + c; t[c] = c; }
#define TAIL                                                                       \
  uint32_t s = 0;                                                                  \
  _Pragma("unroll") for (int c = 0; c < 8; c++) s ^= x[c] ^ y[c] ^ t[c];           \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;

// 8 butterflies per iteration, one after the other: 40 instructions (24 half-rate, 16 full-rate)
__global__ __launch_bounds__(256) void k_bfly_seq(uint32_t* out, uint32_t seed)
{
  HEAD
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int c = 0; c < 8; c++)
      asm volatile("v_mul_hi_i32 %2, %1, %3\n\tv_mul_i32_i24 %2, %2, %4\n\tv_mad_i32_i24 %2, %1, %3, %2\n\t"
                   "v_sub_u32 %1, %0, %2\n\tv_add_u32 %0, %0, %2"
                   : "+v"(x[c]), "+v"(y[c]), "+v"(t[c]) : "v"(a), "v"(b));
  }
  TAIL
}
// the same 40 instructions, grouped by kind: 8 mul_hi, 8 mul, 8 mad, 8 sub, 8 add
__global__ __launch_bounds__(256) void k_bfly_grouped(uint32_t* out, uint32_t seed)
{
  HEAD
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int c = 0; c < 8; c++) asm volatile("v_mul_hi_i32 %0, %1, %2" : "+v"(t[c]) : "v"(y[c]), "v"(a));
#pragma unroll
    for (int c = 0; c < 8; c++) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(t[c]) : "v"(b));
#pragma unroll
    for (int c = 0; c < 8; c++) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(t[c]) : "v"(y[c]), "v"(a));
#pragma unroll
    for (int c = 0; c < 8; c++) asm volatile("v_sub_u32 %0, %1, %2" : "+v"(y[c]) : "v"(x[c]), "v"(t[c]));
#pragma unroll
    for (int c = 0; c < 8; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(t[c]));
  }
  TAIL
}
// strict alternation of independent multiplies and adds: 8 + 8
__global__ __launch_bounds__(256) void k_alt(uint32_t* out, uint32_t seed)
{
  HEAD
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int c = 0; c < 8; c++)
      asm volatile("v_mul_i32_i24 %0, %0, %2\n\tv_add_u32 %1, %1, %3" : "+v"(x[c]), "+v"(y[c]) : "v"(a), "v"(b));
  }
  TAIL
}
// 8 multiplies then 8 adds
__global__ __launch_bounds__(256) void k_blocks(uint32_t* out, uint32_t seed)
{
  HEAD
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int c = 0; c < 8; c++) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x[c]) : "v"(a));
#pragma unroll
    for (int c = 0; c < 8; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(y[c]) : "v"(b));
  }
  TAIL
}
__global__ __launch_bounds__(256) void k_mul(uint32_t* out, uint32_t seed)
{
  HEAD
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int c = 0; c < 8; c++) asm volatile("v_mul_i32_i24 %0, %0, %2\n\tv_mul_i32_i24 %1, %1, %3" : "+v"(x[c]), "+v"(y[c]) : "v"(a), "v"(b));
  }
  TAIL
}
__global__ __launch_bounds__(256) void k_add(uint32_t* out, uint32_t seed)
{
  HEAD
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int c = 0; c < 8; c++) asm volatile("v_add_u32 %0, %0, %2\n\tv_sub_u32 %1, %1, %3" : "+v"(x[c]), "+v"(y[c]) : "v"(a), "v"(b));
  }
  TAIL
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; int per_iter; };

int main()
{
  Entry list[] = { {"mul only (16/iter)", k_mul, 16}, {"add/sub only (16/iter)", k_add, 16}, {"mul,add alternating (16/iter)", k_alt, 16},
                   {"8 mul then 8 add (16/iter)", k_blocks, 16}, {"butterflies one after the other (40/iter)", k_bfly_seq, 40},
                   {"butterflies grouped by kind (40/iter)", k_bfly_grouped, 40} };
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double clk = prop.clockRate * 1e3;  // Hz (nominal maximum)
  const int simds = prop.multiProcessorCount * 4;
  uint32_t* d;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("{\"clock_MHz\": %.0f, \"simds\": %d, \"results\": [", clk / 1e6, simds);
  bool first = true;
  for (int wps : { 1, 2, 4, 8 }) {  // waves per SIMD
    const int threads = 256, blocks = prop.multiProcessorCount * wps;
    hipMalloc(&d, (size_t)blocks * threads * 4);
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
      const double winstr = (double)blocks * (threads / 64) * ITERS * e.per_iter;  // wave-instructions
      printf("%s\n {\"waves_per_simd\": %d, \"stream\": \"%s\", \"Tlaneops\": %.2f, \"cycles_per_instr_per_simd_at_nominal_clock\": %.2f}",
             first ? "" : ",", wps, e.name, winstr * 64 / (best * 1e-3) / 1e12, best * 1e-3 * clk * simds / winstr);
      first = false;
    }
    hipFree(d);
  }
  printf("\n]}\n");
  return 0;
}
