// zq.hpp -- Z_q (q = 12289) arithmetic for gfx950 VALU.
//
// Replaces the reference's Barrett ff_t (ff.hpp:64-167). Any exact reduction yields the same
// canonical residues, so parity is unaffected; the forms below are chosen
// for the CDNA4 vector ALU. Measured on the box (tools/valu_r03.hip, profiles/valu_peak_r03.json): in a
// butterfly-shaped stream every vector instruction -- 24- or 32-bit multiply, low or high half, mad, add, sub --
// costs one issue slot of about four cycles, so the only lever is the NUMBER of instructions.
//
// Round 3: the transforms multiply by constants in Plantard's form (T. Plantard, "Efficient word size modular
// arithmetic", 2021), two instructions instead of the three of the Shoup product used until then:
//     W  = ((-w * 2^32) mod q) * q^-1 mod 2^32          (host side, one dword per twiddle)
//     plant(a, W) = floor(((a * W) mod 2^32) * q / 2^32) = (a * w) mod q,  CANONICAL, for 0 <= a < 2^32 / q
//   (v_mul_lo_u32, v_mul_hi_u32). Why: X = a*W mod 2^32 satisfies X*q = a*wt + k*2^32 with wt = (-w*2^32) mod q,
//   so k = (X*q - a*wt) / 2^32 is the integer floor(X*q / 2^32) as long as 0 <= a*wt < 2^32, and
//   k*2^32 = -a*wt = a*w*2^32 (mod q) gives k = a*w (mod q) with 0 <= k < q.
//   Values therefore live in VGPRs as UNSIGNED lazy residues below 2^32/q = 349,496 (28 q); a Cooley-Tukey
//   butterfly is plant + add + sub = 4 instructions, its difference kept non-negative by a multiple of q added
//   once per transform (verify_kernel.hpp). The same identity with two variable factors (a*b < 2^32) is
//   plant2(a, b) = -(a*b) * 2^-32 mod q; the stray factor is folded into a constant elsewhere.
// The signed Barrett forms further down (reduce, mulmod, mulmod_lazy) remain for the places outside the transforms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

constexpr uint32_t Q = 12289u;
constexpr int32_t QI = 12289;
constexpr int32_t RQ32 = 349496;  // round(2^32 / q): the signed Barrett multiplier of reduce()

typedef uint32_t Tw;  // a twiddle in Plantard's form, plant_const(w)

constexpr uint32_t QINV32 = 150982657u;              // q^-1 mod 2^32
static_assert((uint32_t)(QINV32 * Q) == 1u, "q^-1 mod 2^32");
constexpr uint32_t PLANT_LIMIT = 349496u;            // floor(2^32 / q): plant() takes 0 <= a < PLANT_LIMIT
// W for the constant w (any residue): hi32((a * W mod 2^32) * q) = a * w mod q.
__host__ __device__ constexpr uint32_t plant_const(uint32_t w)
{
  const uint64_t wt = (Q - (uint64_t)(w % Q) * ((1ull << 32) % Q) % Q) % Q;  // (-w * 2^32) mod q
  return (uint32_t)(wt * QINV32);
}
constexpr uint32_t W_ONE = plant_const(1u);                               // plant(a, W_ONE) = a mod q
constexpr uint32_t NEG_R = (Q - (uint32_t)((1ull << 32) % Q)) % Q;        // -2^32 mod q
constexpr uint32_t W_NEG_R = plant_const(NEG_R);                          // plant(a, W_NEG_R) = -a * 2^32 mod q
constexpr uint32_t POS_R = (uint32_t)((1ull << 32) % Q);                  // 2^32 mod q
constexpr uint32_t W_POS_R = plant_const(POS_R);                          // plant(a, W_POS_R) = a * 2^32 mod q: plant2 of it is the NEGATED product

// The two halves of plant(), apart so that a stage can issue them grouped by kind. UNIFORM_W: W lives in an SGPR.
template <bool UNIFORM_W>
__device__ __forceinline__ uint32_t plant_lo(uint32_t a, uint32_t W)
{
  uint32_t X;
  if constexpr (UNIFORM_W)
    asm("v_mul_lo_u32 %0, %1, %2" : "=v"(X) : "v"(a), "s"(W));
  else
    asm("v_mul_lo_u32 %0, %1, %2" : "=v"(X) : "v"(a), "v"(W));
  return X;
}
__device__ __forceinline__ uint32_t plant_hi(uint32_t X)
{
  uint32_t r;
  asm("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"(X), "s"(Q));  // named: left to itself hipcc turns a multiplication by the constant q into shifts and adds
  return r;
}
// 0 <= a < PLANT_LIMIT  ->  (a * w) mod q, canonical.
template <bool UNIFORM_W = true>
__device__ __forceinline__ uint32_t plant(uint32_t a, uint32_t W) { return plant_hi(plant_lo<UNIFORM_W>(a, W)); }
// a * b < 2^32  ->  -(a * b) * 2^-32 mod q, canonical. One factor carries a power of two beforehand: -2^32 (W_NEG_R) where
// the product itself is wanted (polymul), +2^32 (W_POS_R, prepared keys) where its negative is (the verify kernel adds it to c).
__device__ __forceinline__ uint32_t plant2(uint32_t a, uint32_t b) { return plant_hi(plant_lo<true>(a * b, QINV32)); }

// a * b + c, all 64 bits of it, in ONE v_mad_u64_u32. Measured on this chip (tools/valu_r03.hip, single-instruction
// streams at eight waves per SIMD): 31.4 T lane-instructions/s against 34.6 for v_mul_lo_u32 -- a full 32 x 32 -> 64
// product with a 64-bit addend costs 1.1 issue slots, where the low and the high product apart cost 2. The butterflies
// cannot use it (their two products are in series: low half of one, high half of the next), the places that want both
// halves of one product, or a product with something added to it, can.
// The addend is wave-uniform wherever it is used here and is taken from an SGPR pair (the instruction may name one
// scalar operand, so both factors come from vector registers).
__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c_uniform)
{
  uint64_t d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform) : "vcc");
  return d;
}
__device__ __forceinline__ uint64_t mul64(uint32_t a, uint32_t b)
{
  uint64_t d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b) : "vcc");
  return d;
}
// a, b lazy with a * b < 2^32 * q  ->  r = -(a * b) * 2^-32 (mod q) with -(a * b >> 32) <= r < q: the Plantard
// reduction of the product's low word, minus its high word. (Z = low * q^-1 mod 2^32 has Z * q = H * 2^32 + low, so
// (Z q - a b) / 2^32 = H - high is an integer congruent to -a b 2^-32.) Four instructions and neither factor needs to be
// canonical first; the result may be negative by as much as the high word (under 20 for two transform outputs) -- as a
// u32 it wraps, which the first stage of the inverse transform tolerates: it adds its bias before it subtracts.
__device__ __forceinline__ uint32_t plant2_wide(uint32_t a, uint32_t b)
{
  const uint64_t p = mul64(a, b);
  return plant_hi(plant_lo<true>((uint32_t)p, QINV32)) - (uint32_t)(p >> 32);
}

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

// |x| < 2^22  ->  x mod q as a value in [0, q]: signed Barrett, x - floor(x * round(2^32/q) / 2^32) * q (the
// multiplier is off by only 0.04, so the quotient is exact or one short). The instruction is named because hipcc
// otherwise picks the 64-bit v_mad_u64_u32 here; the multiplier must be wave-uniform (SGPR operand).
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

}  // namespace fa
