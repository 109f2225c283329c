// zq.hpp -- Z_q (q = 12289) arithmetic for gfx950 VALU.
//
// Replaces the reference's Barrett ff_t (ff.hpp:64-167). Any exact reduction yields the same
// canonical residues, so parity is unaffected; the form below is chosen
// for the CDNA4 vector ALU, where every integer multiply (24- or 32-bit, low or high half) issues
// at the same half rate and adds/subtracts at full rate (profiles/valu_peak_r01.json):
//   * values live in VGPRs as signed 32-bit integers that are only congruent to the residue
//     ("lazy"), with |x| < 2^19 wherever x feeds a multiplication;
//   * a twiddle w is stored centred, w in (-q/2, q/2), with its Shoup companion
//     wp = round(w * 2^32 / q); then
//       shoup(x, w, wp) = x*w - floor(x*wp / 2^32) * q   is in [0, q]   for |x| < 2^19
//     and costs exactly three VALU instructions (v_mul_hi_i32, v_mul_i32_i24, v_mad_i32_i24);
//   * a butterfly is shoup + one add + one subtract: sums and differences are never reduced in
//     the forward transform (the bound grows by q per stage, 11q after ten stages), and once in
//     the inverse transform (where sums double per stage).
// Why [0, q]: x*wp/2^32 = x*w/q + d with |d| < 2^19 * 0.5 / 2^32 = 2^-14 < 1/q, and the fractional
// part of x*w/q is a multiple of 1/q, so the floor is exact except when x*w/q is an integer and
// d < 0, which yields q instead of 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

constexpr uint32_t Q = 12289u;
constexpr int32_t QI = 12289;
constexpr int32_t RQ32 = 349496;  // round(2^32 / q): Shoup companion of w = 1

struct alignas(8) Tw {  // 8-byte aligned: one ds_read_b64 (banked mod 64) instead of a ds_read2_b32 (banked mod 32, 2-way conflicts on the per-lane tables)
  int32_t w;   // twiddle, centred in (-q/2, q/2)
  int32_t wp;  // round(w * 2^32 / q)
};

__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }

// a * b + c with a, b signed 24-bit: one v_mad_i32_i24 (the backend fuses mul24 + add).
__device__ __forceinline__ int32_t mad_i24(int32_t a, int32_t b, int32_t c) { return __mul24(a, b) + c; }

// (a * b) >> 32, signed: one v_mul_hi_i32, named. __mulhi is a 64-bit product in LLVM IR; where b is loop-invariant
// (the wave-uniform twiddles, 2^32/q) the sign extension of b is hoisted out of the loop and the product then comes
// back as a five-instruction 64-bit multiply. UNIFORM_B: b lives in an SGPR.
template <bool UNIFORM_B>
__device__ __forceinline__ int32_t mulhi_i32(int32_t a, int32_t b)
{
  int32_t r;
  if constexpr (UNIFORM_B)
    asm("v_mul_hi_i32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
  else
    asm("v_mul_hi_i32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// qh * (-q) + p with |qh| < 2^23 (a quotient estimate): one v_mad_i32_i24, named -- left to itself hipcc
// sign-extends qh from 24 bits first in places (two more instructions) or picks the 64-bit v_mad_u64_u32.
__device__ __forceinline__ int32_t msub_q(int32_t qh, int32_t p)
{
  int32_t r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(qh), "s"(-QI), "v"(p));
  return r;
}

// |x| < 2^19  ->  (x * w) mod q as a value in [0, q].
__device__ __forceinline__ int32_t shoup(int32_t x, int32_t w, int32_t wp)
{
  return msub_q(mulhi_i32<false>(x, wp), __mul24(x, w));
}

// |x| < 2^22  ->  x mod q as a value in [0, q] (the companion of w = 1 is off by only 0.04, so the
// range is wider than shoup's). The instruction is named because hipcc otherwise picks the 64-bit
// v_mad_u64_u32 here; the multiplier must be wave-uniform (SGPR operand).
__device__ __forceinline__ int32_t reduce(int32_t x)
{
  return msub_q(mulhi_i32<true>(x, RQ32), x);
}

// Conditional subtract: X < 2*M  ->  X mod M (unsigned min trick, 2 instructions).
__device__ __forceinline__ uint32_t csub(uint32_t X, uint32_t M) { return min(X, X - M); }

// v < 2^17  ->  v mod q, canonical. Barrett with R = floor(2^28/q) = 21843 (ff.hpp:23).
__device__ __forceinline__ uint32_t mod_q_17(uint32_t v)
{
  const uint32_t qh = mul24(v, 21843u) >> 28;
  const uint32_t r = (uint32_t)mad_i24((int32_t)qh, -QI, (int32_t)v);  // [0, 2q)
  return csub(r, Q);
}

// a, b in [0, q] -> (a*b) mod q, canonical (polynomial.hpp:99-110, one element).
__device__ __forceinline__ int32_t mulmod(int32_t a, int32_t b)
{
  const uint32_t p = mul24((uint32_t)a, (uint32_t)b);                   // <= q^2 < 2^28
  const uint32_t qh = __umulhi(p, 349496u);                            // floor(p/q) or one less
  return (int32_t)csub((uint32_t)mad_i24((int32_t)qh, -QI, (int32_t)p), Q);
}

// |a * b| < 2^31 (e.g. a in [0, q], |b| < 2^17)  ->  (a*b) mod q as a value in (-q/60, q + q/60): the signed
// Barrett quotient mulhi(p, round(2^32/q)) is off by at most one when p/q is within 2^31 * 0.04 / 2^32 < 1/50
// of an integer. Three instructions; good enough as input of the inverse transform, whose first Shoup
// multiplication and reductions take any |x| < 2^19.
__device__ __forceinline__ int32_t mulmod_lazy(int32_t a, int32_t b)
{
  const int32_t p = a * b;
  return msub_q(mulhi_i32<true>(p, RQ32), p);
}

}  // namespace fa
