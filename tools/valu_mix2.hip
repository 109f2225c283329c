// valu_mix2.hip -- how long must a run of full-rate instructions be to issue at 2 cycles each? Butterfly streams
// (3 multiplies + sub + add per butterfly) grouped NB butterflies at a time: NB mul_hi, NB mul, NB mad, NB sub, NB add.
// Waves per SIMD and a per-wave start stagger (so that the waves of a SIMD are out of phase, as in a real kernel)
// are varied. Prints cycles per wave-instruction per SIMD at the nominal clock.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITERS 1024

template <int NB, bool STAGGER>
__global__ __launch_bounds__(256) void k_group(uint32_t* out, uint32_t seed)
{
  uint32_t x[16], y[16], t[16], a = seed + threadIdx.x, b = seed * 3u + 7u;
#pragma unroll
  for (int c = 0; c < 16; c++) { x[c] = a + c; y[c] = b + c; t[c] = c; }
  if (STAGGER) {  // waves of one SIMD start 0, 37, 74, ... multiplies apart
    const int extra = ((blockIdx.x * 4 + (threadIdx.x >> 6)) % 13) * 37;
    for (int i = 0; i < extra; i++) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(t[0]) : "v"(a));
  }
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int g = 0; g < 16 / NB; g++) {
#pragma unroll
      for (int c = g * NB; c < (g + 1) * NB; c++) asm volatile("v_mul_hi_i32 %0, %1, %2" : "+v"(t[c]) : "v"(y[c]), "v"(a));
#pragma unroll
      for (int c = g * NB; c < (g + 1) * NB; c++) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(t[c]) : "v"(b));
#pragma unroll
      for (int c = g * NB; c < (g + 1) * NB; c++) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(t[c]) : "v"(y[c]), "v"(a));
#pragma unroll
      for (int c = g * NB; c < (g + 1) * NB; c++) asm volatile("v_sub_u32 %0, %1, %2" : "+v"(y[c]) : "v"(x[c]), "v"(t[c]));
#pragma unroll
      for (int c = g * NB; c < (g + 1) * NB; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(t[c]));
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < 16; c++) s ^= x[c] ^ y[c] ^ t[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { int nb; bool stagger; kern_t k; };

int main()
{
  Entry list[] = { {1, false, k_group<1, false>}, {2, false, k_group<2, false>}, {4, false, k_group<4, false>}, {8, false, k_group<8, false>},
                   {16, false, k_group<16, false>}, {1, true, k_group<1, true>}, {2, true, k_group<2, true>}, {4, true, k_group<4, true>},
                   {8, true, k_group<8, true>}, {16, true, k_group<16, true>} };
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const double clk = prop.clockRate * 1e3;
  const int simds = prop.multiProcessorCount * 4;
  uint32_t* d;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  printf("{\"clock_MHz\": %.0f, \"results\": [", clk / 1e6);
  bool first = true;
  for (int wps : { 2, 3, 5, 8 }) {
    const int threads = 256, blocks = prop.multiProcessorCount * wps;
    (void)hipMalloc(&d, (size_t)blocks * threads * 4);
    for (auto& e : list) {
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 1u);
      (void)hipDeviceSynchronize();
      float best = 1e30f;
      for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 2u + rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double winstr = (double)blocks * (threads / 64) * ITERS * 80.0;
      printf("%s\n {\"waves_per_simd\": %d, \"butterflies_per_group\": %d, \"stagger\": %s, \"cycles_per_instr\": %.2f}", first ? "" : ",", wps,
             e.nb, e.stagger ? "true" : "false", best * 1e-3 * clk * simds / winstr);
      first = false;
    }
    (void)hipFree(d);
  }
  printf("\n]}\n");
  return 0;
}
