// zq.hpp -- Z_q (q = 12289) arithmetic for gfx950 VALU.
//
// Replaces the reference's Barrett ff_t (ff.hpp:64-167). Any exact reduction yields the same
// canonical residues, so parity is unaffected; the form below is chosen
// for the CDNA4 vector ALU:
//   * values live in VGPRs as X = 256*x ("scaled"), x < 2^16, so X < 2^24 and both factors of
//     every product fit the full-rate 24-bit multipliers (v_mul_u32_u24, v_mul_hi_u32_u24,
//     v_mad_i32_i24); the quarter-rate v_mul_lo_u32 is never used on the butterfly path;
//   * a twiddle w is stored with its Shoup companion wp = floor(w * 2^24 / q); then
//       shoup(X, w, wp) = 256 * (w*x - floor(wp*x / 2^24) * q),   in [0, 256*(q + 48)]
//     costs exactly three VALU instructions (mul_hi_u24, mul_u24, mad_i24);
//   * sums and differences are lazy: a full reduction (shoup by 1) runs every second stage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

constexpr uint32_t Q = 12289u;
constexpr uint32_t Q256 = Q * 256u;
constexpr uint32_t ONE_WP = 1365u;  // floor(2^24 / q): Shoup companion of w = 1
constexpr uint32_t SHOUP_SLACK = 48u;  // shoup() output <= 256*(q + 48) for x < 2^16

struct Tw {
  uint32_t w;   // twiddle, canonical
  uint32_t wp;  // floor(w * 2^24 / q)
};

__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }

__device__ __forceinline__ uint32_t mulhi24(uint32_t a, uint32_t b)
{
  return (uint32_t)(((uint64_t)(a & 0xffffffu) * (uint64_t)(b & 0xffffffu)) >> 32);
}

// a * b + c with a, b signed 24-bit: hipcc lowers "c - a*Q256" to a quarter-rate v_mul_lo_u32,
// so the instruction is named explicitly. b must be wave-uniform (SGPR operand).
__device__ __forceinline__ uint32_t mad_i24(uint32_t a, int32_t b, uint32_t c)
{
  uint32_t r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}

// X scaled (< 2^24). Returns 256 * ((w * x) mod q + e*q), e in {0, 1}, value <= 256*(q + 48).
__device__ __forceinline__ uint32_t shoup(uint32_t X, uint32_t w, uint32_t wp)
{
  const uint32_t qh = mulhi24(X, wp);
  return mad_i24(qh, -(int32_t)Q256, mul24(X, w));
}

// Full lazy reduction: X scaled < 2^24  ->  <= 256*(q + 11).
__device__ __forceinline__ uint32_t reduce(uint32_t X)
{
  const uint32_t qh = mulhi24(X, ONE_WP);
  return mad_i24(qh, -(int32_t)Q256, X);
}

// Conditional subtract: X < 2*M  ->  X mod-ish M (unsigned min trick, 2 instructions).
__device__ __forceinline__ uint32_t csub(uint32_t X, uint32_t M) { return min(X, X - M); }

// Plain (unscaled) helpers ------------------------------------------------------------------

// v < 2^17  ->  v mod q, canonical. Barrett with R = floor(2^28/q) = 21843 (ff.hpp:23).
__device__ __forceinline__ uint32_t mod_q_17(uint32_t v)
{
  const uint32_t qh = mul24(v, 21843u) >> 28;
  const uint32_t r = mad_i24(qh, -(int32_t)Q, v);  // [0, 2q)
  return csub(r, Q);
}

// a, b < 2^16 unscaled -> 256 * (a*b mod q), canonical. One quarter-rate multiply
// (v_mul_hi_u32) per element; used only for the N pointwise products of a verify.
__device__ __forceinline__ uint32_t mulmod_scaled(uint32_t a, uint32_t b)
{
  const uint32_t p = mul24(a, b);
  const uint32_t qh = __umulhi(p, 349496u);  // floor(2^32 / q)
  return csub(mad_i24(qh, -(int32_t)Q, p), Q) << 8;
}

}  // namespace fa
