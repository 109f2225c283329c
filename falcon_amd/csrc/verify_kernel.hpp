// verify_kernel.hpp -- one wave64 per polynomial: rejection sampling, public-key unpack,
// NTT / pointwise / iNTT over Z_q, and the fused norm epilogue.
//
// Replaces, for one item per wave:
//   hashing.hpp:41-48      rejection sampling of the squeezed SHAKE256 blocks (t < 61445, t mod q)
//   decoding.hpp:18-48     decode_pkey (14-bit little-endian fields, header = log2 n)
//   ntt.hpp:120-150        forward negacyclic NTT (Cooley-Tukey, natural in, bit-reversed out)
//   polynomial.hpp:99-110  pointwise product
//   ntt.hpp:161-200        inverse NTT with the 1/N scaling
//   verification.hpp:53-79 s1 = c - s2*h, centring, squared norm, comparison with beta^2
// Only the coefficient-domain values c, s2, s1 and the norm are observable, so the transform
// uses 2 forward + 1 inverse NTT (the reference runs 3 + 1, verification.hpp:47-57) and its
// own butterflies; all arithmetic is exact mod q, hence bit-identical results.
//
// Both directions are built from ONE butterfly, the Cooley-Tukey one (t = w*b; b' = a - t; a' = a + t), four
// instructions with the Plantard product of zq.hpp. The forward transform is the reference's own (ntt.hpp:120-150:
// stages on index bits logn-1 .. 0, twiddle zeta^bitrev(k) per block). The inverse is NOT the reference's
// Gentleman-Sande network (ntt.hpp:161-200), whose butterfly multiplies a difference -- that difference would
// need one more instruction to stay non-negative, and its sums double per stage -- but the decimation-in-time
// form of the same map: stages on index bits 0 .. logn-1 with the cyclic twiddles omega^-(N/2^(b+1) * (p mod 2^b)),
// omega = zeta^2, then one multiplication per coefficient by zeta^-p / N (which also leaves the result canonical).
// Same input order (bit-reversed), same output, same values mod q; 156 instead of 200 instructions per lane for
// n = 512, and the forward transform 148 instead of 180.
//
// Values are unsigned lazy residues below 2^32/q (zq.hpp). A butterfly widens the range by q on either side, so
// the first stage of a transform adds BIAS = logn*q to its sums' inputs once; nothing is reduced in between:
//   forward: inputs < 2^16            ->  outputs in [0, 2^16 + 2 logn q) < 311,316
//   inverse: inputs in [-64, q)       ->  before the final scaling in (0, (2 logn + 2) q + 128) <= 270,486;  after it < q
//            (its bias is (logn + 1) q: the pointwise product of the packed-key path may be negative by its high word)
// (the verify kernel itself leaves the inverse transform before that scaling: its epilogue does scaling, the addition
// of c and the centring's reduction in ONE Plantard product, see verify_items)
//
// Data layout inside a wave (n = 64*R coefficients, R = 8 or 16 per lane, LB = log2 R):
//   phase A  lane = i & 63,                     regs hold index bits [6, logn)   stages on those bits
//   phase B  lane = (i>>6 << PB) | (i & (2^PB-1)), regs hold bits [PB, 6), PB = 6 - LB
//   phase C  lane = i >> LB,                    regs hold bits [0, LB)
// with two transposes through a per-wave LDS buffer in between (padded so that every
// ds_read_b32 / ds_write_b32 is bank-conflict free). The forward transform runs A, B, C, the inverse C, B, A.
// Twiddles (one dword each) come from three places. Forward: phase A's are the same for every lane and arrive as
// kernel arguments (SGPRs); phase B's sit in a small LDS table; phase C's sit in an LDS table laid out per lane
// (odd stride, conflict free). Inverse, mirrored: phase C's depend only on the register index (kernel arguments),
// phase B's and A's on the lane's low index bits: consecutive lanes read consecutive dwords of one LDS table.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "decode_kernel.hpp"
#include "zq.hpp"

namespace fa {

constexpr int VERIFY_WAVES = 4;  // waves per workgroup
constexpr int VERIFY_WG = 64 * VERIFY_WAVES;
// The verify kernel itself runs Falcon-1024 in workgroups of eight waves: its LDS is 12.3 KB of twiddle tables per
// workgroup plus 6.4 KB per wave, so eight-wave workgroups (63.5 KB) fit a CU twice = four waves per SIMD, which is
// what the kernel's 128 VGPRs allow (four-wave workgroups would fit only three times).
template <int LOGN> constexpr int verify_waves() { return LOGN == 10 ? 8 : 4; }

template <int LOGN>
struct Geo {
  static constexpr int N = 1 << LOGN;
  static constexpr int R = N / 64;       // coefficients per lane
  static constexpr int LB = LOGN - 6;    // log2 R
  static constexpr int PB = 6 - LB;      // lowest index bit held in registers during phase B
  static constexpr int PKLEN = 1 + 14 * N / 8;
  static constexpr int XBUF = N + 64;    // dwords of the padded transpose buffer
  static constexpr uint32_t BETA2 = LOGN == 9 ? 34034726u : 70265242u;  // falcon.hpp:245
  static constexpr uint32_t BIAS = (uint32_t)LOGN * Q;  // added once per transform: keeps every difference non-negative
  // The inverse transform's inputs may be negative by a few units (plant2_wide, zq.hpp: down to -(product >> 32)): one q more.
  static constexpr uint32_t BIAS_INV = BIAS + Q;
  static constexpr uint32_t INV_IN_MIN = 64;  // inputs of the inverse transform lie in [-INV_IN_MIN, q)
  static constexpr int TWB = N >> PB;    // phase-B forward twiddle indices are < N >> PB
  static constexpr int TWC = LOGN == 9 ? 7 : 13;  // per-lane phase-C entries (odd stride; 7 resp. 12 used)
  static constexpr int PKDW = ((PKLEN + 6) / 4 + 63) / 64;  // aligned dwords per lane covering a key row (any alignment)
  static constexpr int MAXHW = LOGN == 9 ? 5 : 10;    // prefetched dwords of squeezed blocks per lane (9 / 17 blocks)
  static_assert(65535u + 2u * BIAS < PLANT_LIMIT, "forward transform of any u16 input stays below 2^32/q");
  static_assert((2u * LOGN + 2u) * Q + 2u * INV_IN_MIN < PLANT_LIMIT, "inverse transform stays below 2^32/q");
  static_assert(BIAS_INV > Q + INV_IN_MIN + ((uint32_t)LOGN - 1u) * (Q - 1u), "and above zero: the first stage leaves >= BIAS_INV - q - INV_IN_MIN, every later one takes at most q - 1");
  static_assert(((65535ull + 2ull * BIAS) * (16383ull + 2ull * BIAS) >> 32) < INV_IN_MIN, "high word of a product of two forward outputs");
};

// offset of the first phase-C table entry of the stage on register bit lb (stages run lb = PB-1 .. 0,
// the stage on bit lb uses 2^(LB-1-lb) consecutive entries)
template <int LOGN>
__host__ __device__ constexpr int twc_offset(int lb)
{
  int o = 0;
  for (int l = Geo<LOGN>::PB - 1; l > lb; l--) o += 1 << (Geo<LOGN>::LB - 1 - l);
  return o;
}

struct NttConsts {
  Tw twA[16];  // forward twiddles 0..15 of the reference's table (phase A, lane-uniform)
  Tw itwA[8];  // inverse twiddles 0..7 of the cyclic table: the stages on index bits 0..2 (phase C, lane-uniform)
};

enum { TW_UNIFORM = 0, TW_PHASE_B = 1, TW_PHASE_C = 2, TW_UNIFORM_LDS = 3, TW_PHASE_A = 4 };

// element index held by (lane, reg m) in each phase
template <int LOGN> __device__ __forceinline__ uint32_t idxA(uint32_t lane, int m) { return ((uint32_t)m << 6) | lane; }
template <int LOGN> __device__ __forceinline__ uint32_t idxB(uint32_t lane, int m)
{
  constexpr int PB = Geo<LOGN>::PB;
  return ((lane >> PB) << 6) | ((uint32_t)m << PB) | (lane & ((1u << PB) - 1u));
}
template <int LOGN> __device__ __forceinline__ uint32_t idxC(uint32_t lane, int m) { return (lane << Geo<LOGN>::LB) | (uint32_t)m; }

// padded LDS addresses (dword units) of element i for the two transposes
template <int LOGN> __device__ __forceinline__ uint32_t padAB(uint32_t i) { return i + (i >> 6) * (64u / Geo<LOGN>::R); }
template <int LOGN> __device__ __forceinline__ uint32_t padBC(uint32_t i) { return i + (i >> Geo<LOGN>::LB); }

// Sum of v over the 64 lanes (v < 2^26 per lane), returned wave-uniform. Six DPP adds: within rows of
// 16 lanes by row_shr, then across rows by row_bcast; lane 63 ends up holding the total.
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Region marks in the device assembly (comments only), for counting instructions per region: -DFA_MARKS
#ifdef FA_MARKS
#define FA_MARK(name) asm volatile("; FA_MARK " name)
#else
#define FA_MARK(name)
#endif

// The wave's number within its workgroup, as a scalar: addresses derived from it (the wave's LDS buffers) are then
// scalar too and fold into instructions as operands instead of occupying vector registers and adds.
__device__ __forceinline__ uint32_t wave_index() { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ void wave_lds_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Forward twiddle of the butterfly (m, m | 1<<lb) at global bit b. Reference index (ntt.hpp:133):
//   (N >> (b+1)) + (i >> (b+1)),  i >> (b+1) = (lanepart << (LB-1-lb)) | (m >> (lb+1)).
// KIND selects where it lives; `tw` is the kernel-argument array, the phase-B LDS table, or this
// lane's row of the phase-C LDS table. TW_UNIFORM_LDS: lane-uniform like TW_UNIFORM but read (as a broadcast)
// from the LDS table that holds the reference-order twiddles.
template <int LOGN, int KIND>
__device__ __forceinline__ Tw twiddle(const Tw* __restrict__ tw, uint32_t lanepart, int b, int lb, int m)
{
  constexpr int LB = Geo<LOGN>::LB, N = Geo<LOGN>::N;
  if constexpr (KIND == TW_UNIFORM || KIND == TW_UNIFORM_LDS) return tw[(N >> (b + 1)) + (m >> (lb + 1))];
  if constexpr (KIND == TW_PHASE_B) return tw[(N >> (b + 1)) + (lanepart << (LB - 1 - lb)) + (m >> (lb + 1))];
  return tw[twc_offset<LOGN>(lb) + (m >> (lb + 1))];
}

// Inverse (decimation-in-time) twiddle of the butterfly (m, m | 1<<lb) at global bit b: entry 2^b + (p mod 2^b) of
// the cyclic table, p the element's index. The register part of p mod 2^b is m's bits below lb; the lane's part is
// already in the pointer: phase C none (tw = kernel arguments), phase B tw = table + (lane & (2^PB - 1)) with the
// register bits PB places up, phase A tw = table + lane with the register bits six places up.
template <int LOGN, int KIND>
__device__ __forceinline__ Tw itwiddle(const Tw* __restrict__ tw, int b, int lb, int m)
{
  const int jreg = m & ((1 << lb) - 1);
  if constexpr (KIND == TW_UNIFORM) return tw[(1 << b) + jreg];
  if constexpr (KIND == TW_PHASE_B) return tw[(1 << b) + (jreg << Geo<LOGN>::PB)];
  return tw[(1 << b) + (jreg << 6)];
}

// Instruction order inside a stage. Measured on the box (tools/valu_mix.hip, tools/valu_r03.hip): hipcc emits a
// stage butterfly after butterfly, a dependent chain; grouped by kind (all low products of the stage, all high
// products, then all subs and adds) neighbours are independent. In the kernels this is worth 4 % for Falcon-1024
// (8 butterflies per stage and lane, four waves per SIMD) and nothing measurable for Falcon-512 (4 per stage, five
// waves per SIMD: the other waves already fill the slots). The fences keep hipcc's scheduler from undoing the
// grouping: vector ALU instructions may not cross them, LDS, memory and scalar ones may.
#ifndef FA_GROUPED
#define FA_GROUPED 1
#endif
__device__ __forceinline__ void valu_fence()
{
#if FA_GROUPED
  __builtin_amdgcn_sched_barrier(0x4 | 0x10 | 0x80);  // SALU, VMEM and DS instructions may be scheduled across
#endif
}

// Butterfly stages on register bits of x[]: t = w * x[hi] (Plantard, canonical), x[hi] = x[lo] - t, x[lo] += t.
// FORWARD: bits LBTOP .. LBTOP-NST+1 (global bits BTOP ..), the reference's twiddles; else bits LBTOP .. LBTOP+NST-1
// upwards (global bits BTOP ..), the cyclic twiddles of the inverse. FIRST: the transform's first stage, which adds
// BIAS to the x[lo] of its butterflies (every later value descends from one of them); the inverse's first stage
// (index bit 0) has the twiddle 1 throughout and multiplies nothing.
template <int LOGN, bool FORWARD, int NST, int BTOP, int LBTOP, int KIND, bool FIRST>
__device__ __forceinline__ void bfly_stages(uint32_t (&x)[Geo<LOGN>::R], const Tw* __restrict__ tw, uint32_t lanepart)
{
  constexpr int R = Geo<LOGN>::R;
  constexpr bool UNI = KIND == TW_UNIFORM;
#pragma unroll
  for (int s = 0; s < NST; s++) {
    const int b = FORWARD ? BTOP - s : BTOP + s, lb = FORWARD ? LBTOP - s : LBTOP + s;
    const bool first = FIRST && s == 0;
    const bool nomul = first && !FORWARD;
    uint32_t v[R / 2];
    valu_fence();
#pragma unroll
    for (int m = 0, k = 0; m < R; m++) {
      if (m & (1 << lb)) continue;
      const uint32_t hi = x[m | (1 << lb)];
      if (nomul)
        v[k] = hi;
      else
        v[k] = plant_lo<UNI>(hi, FORWARD ? twiddle<LOGN, KIND>(tw, lanepart, b, lb, m) : itwiddle<LOGN, KIND>(tw, b, lb, m));
      k++;
    }
    if (!nomul) {
      valu_fence();
#pragma unroll
      for (int k = 0; k < R / 2; k++) v[k] = plant_hi(v[k]);
    }
    valu_fence();
    if (first) {
#pragma unroll
      for (int m = 0; m < R; m++)
        if (!(m & (1 << lb))) x[m] += FORWARD ? Geo<LOGN>::BIAS : Geo<LOGN>::BIAS_INV;
    }
#pragma unroll
    for (int m = 0, k = 0; m < R; m++) {
      if (m & (1 << lb)) continue;
      x[m | (1 << lb)] = x[m] - v[k];
      k++;
    }
#pragma unroll
    for (int m = 0, k = 0; m < R; m++) {
      if (m & (1 << lb)) continue;
      x[m] = x[m] + v[k];
      k++;
    }
    valu_fence();
  }
}

template <int LOGN, int PHASE_FROM>  // 0: A->B, 1: B->C, 2: C->B, 3: B->A
__device__ __forceinline__ void transpose_out(const uint32_t (&x)[Geo<LOGN>::R], uint32_t* __restrict__ buf, uint32_t lane)
{
  constexpr int R = Geo<LOGN>::R;
#pragma unroll
  for (int m = 0; m < R; m++) {
    if constexpr (PHASE_FROM == 0) buf[padAB<LOGN>(idxA<LOGN>(lane, m))] = x[m];
    if constexpr (PHASE_FROM == 1) buf[padBC<LOGN>(idxB<LOGN>(lane, m))] = x[m];
    if constexpr (PHASE_FROM == 2) buf[padBC<LOGN>(idxC<LOGN>(lane, m))] = x[m];
    if constexpr (PHASE_FROM == 3) buf[padAB<LOGN>(idxB<LOGN>(lane, m))] = x[m];
  }
}
template <int LOGN, int PHASE_FROM>
__device__ __forceinline__ void transpose_in(uint32_t (&x)[Geo<LOGN>::R], const uint32_t* __restrict__ buf, uint32_t lane)
{
  constexpr int R = Geo<LOGN>::R;
#pragma unroll
  for (int m = 0; m < R; m++) {
    if constexpr (PHASE_FROM == 0) x[m] = buf[padAB<LOGN>(idxB<LOGN>(lane, m))];
    if constexpr (PHASE_FROM == 1) x[m] = buf[padBC<LOGN>(idxC<LOGN>(lane, m))];
    if constexpr (PHASE_FROM == 2) x[m] = buf[padBC<LOGN>(idxB<LOGN>(lane, m))];
    if constexpr (PHASE_FROM == 3) x[m] = buf[padAB<LOGN>(idxA<LOGN>(lane, m))];
  }
}
template <int LOGN, int PHASE_FROM>
__device__ __forceinline__ void transpose(uint32_t (&x)[Geo<LOGN>::R], uint32_t* __restrict__ buf, uint32_t lane)
{
  transpose_out<LOGN, PHASE_FROM>(x, buf, lane);
  wave_lds_sync();
  transpose_in<LOGN, PHASE_FROM>(x, buf, lane);
  wave_lds_sync();
}

// Per-wave view of the twiddle tables.
struct TwTabs {
  const Tw* fA;  // kernel arguments
  const Tw* fB;  // LDS, phase B
  const Tw* fC;  // LDS, this lane's phase-C row
  const Tw* iC;  // kernel arguments
  const Tw* iB;  // LDS, the cyclic table + (lane & (2^PB - 1))
  const Tw* iA;  // LDS, the cyclic table + lane
  const Tw* sc;  // LDS, the final scaling zeta^-p / N, + lane
};

// Phase A of the forward transform: the stages on the register-held index bits [6, logn), lane-uniform twiddles.
// Falcon-1024's widest stage (8 twiddles) reads them from LDS as a broadcast: as kernel arguments they pushed the
// kernel past its SGPR budget (the spills went to VGPR lanes).
template <int LOGN>
__device__ __forceinline__ void phase_a_forward(uint32_t (&x)[Geo<LOGN>::R], const TwTabs& T)
{
  using G = Geo<LOGN>;
  if constexpr (LOGN == 10) {
    bfly_stages<LOGN, true, G::LB - 1, LOGN - 1, G::LB - 1, TW_UNIFORM, true>(x, T.fA, 0u);
    bfly_stages<LOGN, true, 1, 6, 0, TW_UNIFORM_LDS, false>(x, T.fB, 0u);
  } else {
    bfly_stages<LOGN, true, G::LB, LOGN - 1, G::LB - 1, TW_UNIFORM, true>(x, T.fA, 0u);
  }
}

// x: < 2^16, phase-A layout (natural order)  ->  NTT in phase-C layout (bit-reversed order), < 2^16 + 2 logn q
template <int LOGN>
__device__ __forceinline__ void ntt_forward(uint32_t (&x)[Geo<LOGN>::R], const TwTabs& T, uint32_t* __restrict__ buf,
                                            uint32_t lane)
{
  using G = Geo<LOGN>;
  phase_a_forward<LOGN>(x, T);
  transpose<LOGN, 0>(x, buf, lane);
  bfly_stages<LOGN, true, G::LB, 5, G::LB - 1, TW_PHASE_B, false>(x, T.fB, lane >> G::PB);
  transpose<LOGN, 1>(x, buf, lane);
  bfly_stages<LOGN, true, G::PB, G::PB - 1, G::PB - 1, TW_PHASE_C, false>(x, T.fC, lane);
}

// Two independent forward transforms side by side (a and b of one product, f and g of one key):
// stage for stage on both, each with its own transpose buffer, so one transform's LDS round trips and
// dependent product chains hide behind the other's arithmetic.
template <int LOGN>
__device__ __forceinline__ void ntt_forward2(uint32_t (&x)[Geo<LOGN>::R], uint32_t (&y)[Geo<LOGN>::R], const TwTabs& T,
                                             uint32_t* __restrict__ bufx, uint32_t* __restrict__ bufy, uint32_t lane)
{
  using G = Geo<LOGN>;
  phase_a_forward<LOGN>(x, T);
  transpose_out<LOGN, 0>(x, bufx, lane);
  phase_a_forward<LOGN>(y, T);
  transpose_out<LOGN, 0>(y, bufy, lane);
  wave_lds_sync();
  transpose_in<LOGN, 0>(x, bufx, lane);
  transpose_in<LOGN, 0>(y, bufy, lane);
  wave_lds_sync();
  bfly_stages<LOGN, true, G::LB, 5, G::LB - 1, TW_PHASE_B, false>(x, T.fB, lane >> G::PB);
  transpose_out<LOGN, 1>(x, bufx, lane);
  bfly_stages<LOGN, true, G::LB, 5, G::LB - 1, TW_PHASE_B, false>(y, T.fB, lane >> G::PB);
  transpose_out<LOGN, 1>(y, bufy, lane);
  wave_lds_sync();
  transpose_in<LOGN, 1>(x, bufx, lane);
  transpose_in<LOGN, 1>(y, bufy, lane);
  wave_lds_sync();
  bfly_stages<LOGN, true, G::PB, G::PB - 1, G::PB - 1, TW_PHASE_C, false>(x, T.fC, lane);
  bfly_stages<LOGN, true, G::PB, G::PB - 1, G::PB - 1, TW_PHASE_C, false>(y, T.fC, lane);
}

// x: in [-64, q) (as u32, wrapping), phase-C layout (bit-reversed order)  ->  coefficients in phase-A layout (natural order), canonical,
// the 1/N of ntt.hpp:193-199 included in the final scaling. SCALE = false leaves the final scaling to the caller:
// x[m] is then lazy (below (2 logn + 1) q) and still to be multiplied by T.sc[m << 6] (the verify kernel's epilogue
// does that inside the reduction it needs anyway).
template <int LOGN, bool SCALE = true>
__device__ __forceinline__ void ntt_inverse(uint32_t (&x)[Geo<LOGN>::R], const TwTabs& T, uint32_t* __restrict__ buf,
                                            uint32_t lane)
{
  using G = Geo<LOGN>;
  constexpr int R = G::R;
  static_assert(G::PB <= 3, "the lane-uniform inverse twiddles are entries 1..7 of the cyclic table");
  bfly_stages<LOGN, false, G::PB, 0, 0, TW_UNIFORM, true>(x, T.iC, 0u);
  transpose<LOGN, 2>(x, buf, lane);
  bfly_stages<LOGN, false, G::LB, G::PB, 0, TW_PHASE_B, false>(x, T.iB, 0u);
  transpose<LOGN, 3>(x, buf, lane);
  bfly_stages<LOGN, false, G::LB, 6, 0, TW_PHASE_A, false>(x, T.iA, 0u);
  if constexpr (!SCALE) return;
  valu_fence();
#pragma unroll
  for (int m = 0; m < R; m++) x[m] = plant_lo<false>(x[m], T.sc[m << 6]);
  valu_fence();
#pragma unroll
  for (int m = 0; m < R; m++) x[m] = plant_hi(x[m]);
  valu_fence();
}

// Workgroup-shared twiddle tables in LDS and their one-time fill.
template <int LOGN>
struct TwLds {
  Tw fB[Geo<LOGN>::TWB];      // forward, reference order, entries below N >> PB
  Tw fC[64 * Geo<LOGN>::TWC]; // forward, per lane
  Tw inv[Geo<LOGN>::N];       // inverse: entry 2^b + j = omega^-(N/2^(b+1) * j), j < 2^b
  Tw sc[Geo<LOGN>::N];        // inverse: entry p = zeta^-p / N
};
template <int LOGN>
__device__ __forceinline__ void fill_tw_lds(TwLds<LOGN>& L, const Tw* __restrict__ tw, const Tw* __restrict__ itw,
                                            const Tw* __restrict__ twc, const Tw* __restrict__ isc)
{
  using G = Geo<LOGN>;
  for (int i = threadIdx.x; i < G::TWB; i += blockDim.x) L.fB[i] = tw[i];
  for (int i = threadIdx.x; i < 64 * G::TWC; i += blockDim.x) L.fC[i] = twc[i];
  for (int i = threadIdx.x; i < G::N; i += blockDim.x) {
    L.inv[i] = itw[i];
    L.sc[i] = isc[i];
  }
  __syncthreads();
}
template <int LOGN>
__device__ __forceinline__ TwTabs make_tabs(const TwLds<LOGN>& L, const NttConsts& nc, uint32_t lane)
{
  using G = Geo<LOGN>;
  return TwTabs{ nc.twA, L.fB, L.fC + lane * G::TWC, nc.itwA, L.inv + (lane & ((1u << G::PB) - 1u)), L.inv + lane, L.sc + lane };
}

struct DebugOut {  // device pointers, any may be null
  uint16_t* c;
  int16_t* s2;
  int16_t* s1;
  unsigned long long* norm_exact;
  uint32_t* norm_wrap32;
  uint8_t* status;
};

template <int LOGN>
struct VerifyArgs {
  Rows pk;                  // packed public keys, one record per item (packed-key kernels)
  const uint16_t* prepared; // [keys][N] prepared keys, see prepare_keys_kernel (prepared-key kernels)
  const uint32_t* key_idx;  // item -> key, nullable (then key = item_base + item)
  const uint64_t* hraw;     // [slots][nblk*17]
  const uint16_t* s2;       // lifted s2 in the tiled layout of s2_tile_offset (decode kernels)
  const uint32_t* sqn;      // [slots]
  const uint8_t* st2;       // [slots]
  const uint32_t* idx;      // slot -> item, nullable
  const Tw* tw;             // forward twiddles [N], reference order (ntt.hpp:62-83), Plantard form
  const Tw* itw;            // inverse twiddles [N], the cyclic table (TwLds::inv)
  const Tw* twc;            // forward phase-C table [64][TWC]
  const Tw* isc;            // inverse scaling [N] (TwLds::sc)
  NttConsts nc;
  uint32_t count;           // slots
  const uint32_t* dcount;   // nullable: device-side number of slots, count = min(count, *dcount) (retry pass, mixed batches)
  uint32_t nblk;
  uint32_t wrap32;          // norm policy
  uint32_t item_base;       // added to item in retry_list entries (chunked main pass)
  uint32_t half;            // s1 values >= half are negative: 6144 as the reference centres (verification.hpp:59-68), 6145 as the specification does
  uint8_t* accept;          // [items]
  uint32_t* retry_count;    // items whose hash needs more blocks
  uint32_t* retry_list;
  DebugOut dbg;
};

constexpr uint32_t PREPARED_BAD_KEY = 0xffffu;  // first coefficient of a prepared key whose header was wrong

// decode_pkey body (decoding.hpp:34-45): the key row, held as aligned dwords t*64+lane with the header
// at byte `pal` of dword 0, goes through LDS so that each lane can pick its 14-bit little-endian fields
// (phase-A layout, raw values <= 16383).
template <int LOGN>
__device__ __forceinline__ void unpack_pkey(const uint32_t (&pkw)[Geo<LOGN>::PKDW], uint32_t pal, uint32_t* __restrict__ pkb,
                                            uint32_t lane, uint32_t (&y)[Geo<LOGN>::R])
{
  using G = Geo<LOGN>;
#pragma unroll
  for (int t = 0; t < G::PKDW; t++) pkb[t * 64 + lane] = pkw[t];
  wave_lds_sync();
  // Coefficient m*64 + lane starts at bit (pal + 1)*8 + 14*lane + 896*m: 896 bits are 28 dwords, so every m has the
  // same shift and a dword index 28 apart -- one address and one shift per item, immediate offsets after that.
  const uint32_t bit0 = (pal + 1u) * 8u + 14u * lane;
  const uint32_t* src = pkb + (bit0 >> 5);
  const uint32_t sh = bit0 & 31u;
  static_assert(14 * 64 == 28 * 32 && ((3 + 1) * 8 + 14 * (Geo<LOGN>::N - 1)) / 32 + 1 < Geo<LOGN>::PKDW * 64, "pk row layout");
#pragma unroll
  for (int m = 0; m < G::R; m++)
    y[m] = __builtin_amdgcn_alignbit(src[28 * m + 1], src[28 * m], sh) & 0x3fffu;  // <= 16383, reduced by the transform itself
}

// Prepared keys: NTT(h) of a public key times +2^32, canonical u16, in the order the verify kernel consumes it (lane L
// holds positions L*R .. L*R+R-1 of the transform's bit-reversed output): an opaque blob. One wave per key. The factor:
// the verify kernel's two-variable Plantard product (zq.hpp plant2) yields -a*b*2^-32, so with it the product comes out
// as MINUS NTT(s2)*NTT(h) -- which is what the epilogue wants (s1 = c + (-s2*h) is a sum of non-negative values).
// PACKED: input rows are encoded keys (decode_pkey, decoding.hpp:18-48; a wrong header yields the
// PREPARED_BAD_KEY mark, the counterpart of falcon.hpp:240-243); else rows are decoded polynomials h as
// verification::verify takes them (N u16, any value: the ff_t arithmetic reduces, ff.hpp:69-76).
template <int LOGN, bool PACKED>
__global__ __launch_bounds__(VERIFY_WG) void prepare_keys_kernel(const uint8_t* __restrict__ pk, const uint16_t* __restrict__ h,
                                                                 uint32_t nkeys, uint16_t* __restrict__ prepared,
                                                                 const Tw* __restrict__ tw, const Tw* __restrict__ itw,
                                                                 const Tw* __restrict__ twc, const Tw* __restrict__ isc,
                                                                 const NttConsts nc)
{
  using G = Geo<LOGN>;
  constexpr int N = G::N, R = G::R;
  __shared__ TwLds<LOGN> s_tw;
  __shared__ uint32_t s_buf[VERIFY_WAVES][G::XBUF];
  __shared__ uint32_t s_pk[VERIFY_WAVES][G::PKDW * 64];
  fill_tw_lds<LOGN>(s_tw, tw, itw, twc, isc);
  const uint32_t lane = threadIdx.x & 63u, wave = wave_index();
  const TwTabs T = make_tabs<LOGN>(s_tw, nc, lane);
  for (uint32_t key = blockIdx.x * VERIFY_WAVES + wave; key < nkeys; key += gridDim.x * VERIFY_WAVES) {
    uint32_t y[R];
    bool ok = true;
    if constexpr (PACKED) {
      const uint8_t* pkrow = pk + (size_t)key * G::PKLEN;
      const uint32_t pal = (uint32_t)((uintptr_t)pkrow & 3u);
      const uint32_t* pbase = (const uint32_t*)(pkrow - pal);
      const uint32_t last_dw = (pal + (uint32_t)G::PKLEN - 1u) >> 2;
      uint32_t pkw[G::PKDW];
#pragma unroll
      for (int t = 0; t < G::PKDW; t++) pkw[t] = pbase[min(t * 64u + lane, last_dw)];
      ok = ((__builtin_amdgcn_readfirstlane(pkw[0]) >> (8u * __builtin_amdgcn_readfirstlane(pal))) & 0xffu) == (uint32_t)LOGN;
      unpack_pkey<LOGN>(pkw, pal, s_pk[wave], lane, y);
    } else {
#pragma unroll
      for (int m = 0; m < R; m++) y[m] = h[(size_t)key * N + idxA<LOGN>(lane, m)];
    }
    ntt_forward<LOGN>(y, T, s_buf[wave], lane);
    uint32_t* out = (uint32_t*)(prepared + (size_t)key * N) + lane * (R / 2);
#pragma unroll
    for (int j = 0; j < R / 2; j++) {
      uint32_t lo = plant(y[2 * j], W_POS_R), hi = plant(y[2 * j + 1], W_POS_R);
      if (!ok) lo = hi = 0u;
      if (!ok && j == 0 && lane == 0) lo = PREPARED_BAD_KEY;
      out[j] = lo | (hi << 16);
    }
    wave_lds_sync();
  }
}

// Everything one item needs from HBM, loaded one item ahead of its use so that the latency of
// the loads hides behind the previous item's transforms.
template <int LOGN>
struct ItemRegs {
  uint32_t hw[Geo<LOGN>::MAXHW];   // squeezed blocks, dword t*64+lane
  uint32_t s2v[Geo<LOGN>::R];      // s2[m*64 + lane]
  uint32_t pkw[Geo<LOGN>::PKDW];   // key row, aligned dword t*64+lane          (packed keys)
  uint32_t hk[Geo<LOGN>::R / 2];   // prepared key, coefficients lane*R .. lane*R+R-1, two per dword
  uint32_t sqn, st, item, pal;
  uint32_t key;                    // prepared keys: the item's key index (wave-uniform)
};

template <int LOGN, bool PREPARED, bool GENERAL>
__device__ __forceinline__ void prefetch_item(const VerifyArgs<LOGN>& A, uint32_t slot, uint32_t lane, ItemRegs<LOGN>& r)
{
  using G = Geo<LOGN>;
  slot = __builtin_amdgcn_readfirstlane(slot);  // wave-uniform: row bases become scalar, lanes add 32-bit offsets
  r.item = A.idx ? A.idx[slot] : slot;
  r.item = __builtin_amdgcn_readfirstlane(r.item);
  if constexpr (PREPARED) {  // the key's index, before the bulk loads: a load that must be waited for in the middle of
    const uint32_t key = A.key_idx ? A.key_idx[r.item] : A.item_base + r.item;  // them makes every load before it a load waited for
    r.key = __builtin_amdgcn_readfirstlane(key);  // (the key row itself is requested later, with s2: prefetch_key)
  }
  const uint32_t* hrow = (const uint32_t*)(A.hraw + (size_t)slot * A.nblk * 17u);
  const uint32_t ndw = A.nblk * 34u;
#pragma unroll
  for (int t = 0; t < G::MAXHW; t++) {
    const uint32_t d = t * 64u + lane;
    r.hw[t] = d < ndw ? hrow[d] : 0xffffffffu;  // 0xffff draws are always rejected
  }
  if constexpr (PREPARED) {
    r.pal = 0u;
  } else {
    const uint8_t* pkrow = GENERAL ? A.pk.at(r.item) : A.pk.base + (size_t)r.item * G::PKLEN;
    r.pal = (uint32_t)((uintptr_t)pkrow & 3u);
    const uint32_t* pbase = (const uint32_t*)(pkrow - r.pal);
    const uint32_t last_dw = (r.pal + (uint32_t)G::PKLEN - 1u) >> 2;
#pragma unroll
    for (int t = 0; t < G::PKDW; t++) {
      const uint32_t d = t * 64u + lane;
      r.pkw[t] = pbase[min(d, last_dw)];
    }
  }
  r.sqn = A.sqn[slot];
  r.st = A.st2[slot];
}

// The prepared key of an item: like s2 it is requested when its registers are free, after the pointwise product that is
// their last use -- the product reads cur.hk itself, no copy -- and arrives behind the inverse transform.
template <int LOGN>
__device__ __forceinline__ void prefetch_key(const uint16_t* __restrict__ prepared, uint32_t key, uint32_t lane,
                                             uint32_t (&hk)[Geo<LOGN>::R / 2])
{
  using G = Geo<LOGN>;
  const uint32_t* krow = (const uint32_t*)(prepared + (size_t)__builtin_amdgcn_readfirstlane(key) * G::N);
#pragma unroll
  for (int j = 0; j < G::R / 2; j++) hk[j] = krow[lane * (G::R / 2) + j];
}

// The s2 part of an item's loads, apart from the rest: its registers are the ones the item's first transform works in,
// so they are free (and the next item's s2 can be requested) only once the pointwise product has been formed.
//
// s2 is tiled (chunk stride 64*32, see decode_kernel.hpp): coefficient i = 64 m + lane sits at row base + m * 4096 +
// (lane >> 5) * 2048 + (lane & 31). The per-m bases are wave-uniform: each load is SGPR base + one VGPR byte offset
// shared by all m.
//
// The loads are written as instructions, not as C++ loads, because of what hipcc does with the C++ form: the values
// are carried around the item loop as 16-bit quantities and zero-extended where the loop closes -- a v_and per
// coefficient and item for something global_load_ushort has already done -- and Falcon-1024's sixteen addresses were
// built with 64-bit vector adds (31 instructions per item). The price: the compiler does not know these registers are
// the targets of loads in flight. s2_arrived() is the only thing that may touch them after prefetch_s2(): it waits
// for all vector memory operations and then hands the registers back to the compiler as ordinary values. The two
// calls bracket straight-line code in which the values have no other use; tools/check_asm_loads.py (run by build.sh on
// the device assembly) checks that nothing reads or writes the target registers between a load and the wait.
template <int LOGN>
__device__ __forceinline__ void prefetch_s2(const uint16_t* __restrict__ s2, uint32_t slot, uint32_t lane, uint32_t (&s2v)[Geo<LOGN>::R])
{
  using G = Geo<LOGN>;
  slot = __builtin_amdgcn_readfirstlane(slot);
  const uint16_t* s2row = s2 + s2_tile_offset<LOGN>(slot, 0u);
  const uint32_t lane_off = ((lane >> 5) * 2048u + (lane & 31u)) * 2u;  // in bytes
#pragma unroll
  for (int m = 0; m < G::R; m++) {
    const uint16_t* base_m = s2row + (size_t)m * 4096u;
    asm volatile("global_load_ushort %0, %1, %2 ; FA_ASM_LOAD" : "=v"(s2v[m]) : "v"(lane_off), "s"(base_m));
  }
}
template <int LOGN>
__device__ __forceinline__ void s2_arrived(uint32_t (&s2v)[Geo<LOGN>::R])
{
  asm volatile("s_waitcnt vmcnt(0) ; FA_ASM_LOAD_WAIT" ::: "memory");
#pragma unroll
  for (int m = 0; m < Geo<LOGN>::R; m++) asm volatile("" : "+v"(s2v[m]));
}

// PREPARED: the public key arrives as NTT(h) (prepare_keys_kernel) instead of packed bytes -- the
// decoded-key form of verification::verify<N, beta2>(h, ...) (verification.hpp:20-26, falcon.hpp:226-229) for
// callers that verify many signatures per key: no unpacking and one forward transform less per item.
// GENERAL: key records through A.pk's offset array and a device-side slot count (mixed batches, the device-side retry
// pass). A template parameter rather than a run-time test because the two extra pointers cost the main instantiation its
// fifth wave per SIMD (82 -> 101 VGPRs, scalar registers spilling to vector lanes: +11 % cycles, measured).
// The work of one wave: slots slot, slot + nwaves, ... below count. buf: this wave's transpose buffer (G::XBUF dwords; the
// key row is staged in it before the first transpose), cbuf: its N hash-to-point coefficients, T: the twiddle tables.
// A function of its own so that retry_item_kernel (capi.hip) can run it behind a decoder wave and a sponge wave in the
// same launch.
template <int LOGN, bool DEBUG, bool PREPARED, bool GENERAL>
__device__ __forceinline__ void verify_items(const VerifyArgs<LOGN>& A, const TwTabs& T, uint32_t* __restrict__ buf,
                                             uint16_t* __restrict__ cbuf, uint32_t lane, uint32_t slot, uint32_t nwaves, uint32_t count)
{
  using G = Geo<LOGN>;
  constexpr int N = G::N, R = G::R;
  static_assert(G::PKDW * 64 <= G::XBUF, "the key row fits the transpose buffer");
  uint32_t* pkb = buf;  // same wave, LDS operations of a wave complete in order: the fields are read before buf is rewritten
  if (slot >= count) return;
  ItemRegs<LOGN> cur;
  prefetch_item<LOGN, PREPARED, GENERAL>(A, slot, lane, cur);
  if constexpr (PREPARED) prefetch_key<LOGN>(A.prepared, cur.key, lane, cur.hk);
  prefetch_s2<LOGN>(A.s2, slot, lane, cur.s2v);
  s2_arrived<LOGN>(cur.s2v);  // once per wave: the first item's s2 is waited for here

  // One register set: an item's loaded words are consumed at the top of its iteration (sampling, key fields into y[]),
  // and the NEXT item's loads are then issued straight into the same registers, to arrive behind this item's
  // transforms. (Round 1 kept two sets and copied nxt into cur at the end of every iteration: 21 v_mov per item and 21
  // more live registers.) The loads are waited for once, at the top of the iteration that uses them.
  // s2 is the exception: its registers ARE x[], the array the first transform works in (no copy), so the next item's
  // s2 is requested when x[] is dead -- after the pointwise product, whose result goes to y[] -- and arrives behind the
  // inverse transform and the epilogue.
  while (true) {
    FA_MARK("top");
    vmem_wait_all();
    const uint32_t item = __builtin_amdgcn_readfirstlane(cur.item);
    const uint32_t sqn = __builtin_amdgcn_readfirstlane(cur.sqn);
    // header byte of the key row = byte `pal` of aligned dword 0 (lane 0 holds it); falcon.hpp:240-243
    uint32_t status = __builtin_amdgcn_readfirstlane(cur.st);
    if constexpr (PREPARED) {  // a key whose header was wrong is marked by 0xffff in its first coefficient
      if ((__builtin_amdgcn_readfirstlane(cur.hk[0]) & 0xffffu) == PREPARED_BAD_KEY) status = ST_BAD_PK_HEADER;
    } else {
      const uint32_t hdr = (__builtin_amdgcn_readfirstlane(cur.pkw[0]) >> (8u * __builtin_amdgcn_readfirstlane(cur.pal))) & 0xffu;
      if (hdr != (uint32_t)LOGN) status = ST_BAD_PK_HEADER;  // checked before the signature, as the reference does
    }

    FA_MARK("sampling");
    bool done = false;
    uint32_t have = 0;  // accepted draws so far (wave-uniform)
    if (status == ST_OK) {
      // ---- rejection sampling (hashing.hpp:38-49): 16-bit big-endian draws in stream order ----
      const uint32_t ndw = A.nblk * 34u;
      // `bounded`: the draws so far plus this round's 128 cannot exceed N, so no position needs checking
      auto sample = [&](uint32_t w, bool bounded = false) {
        // both draws of the dword byte-swapped in place by one v_perm: t0 = low half, t1 = high half. t1 < 61445 is
        // p < 61445 << 16 on the whole register, the stores take the halves as they lie (ds_write_b16 / _d16_hi), and the
        // second draw's position is the first's plus one exactly in the lanes that accepted the first: two stores at a
        // fixed distance under complementary masks instead of a per-lane address.
        const uint32_t p = __builtin_amdgcn_perm(0u, w, 0x02030001u);
        const bool a0 = (p & 0xffffu) < 61445u, a1 = p < (61445u << 16);
        const unsigned long long m0 = __ballot(a0), m1 = __ballot(a1);
        // both draws of a lane come after all draws of the lanes below it: one prefix count over each mask, chained
        // (the count so far, `have`, is a scalar: it goes into the address, not into the prefix count's vector operand)
        const uint32_t p0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1,
                            __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))));
        uint16_t* dst = (cbuf + have) + p0;
        const uint32_t room = (uint32_t)N - have;  // > 0
        if (a0 && (bounded || p0 < room)) dst[0] = (uint16_t)p;
        if (a1 && a0 && (bounded || p0 + 1u < room)) dst[1] = (uint16_t)(p >> 16);
        if (a1 && !a0 && (bounded || p0 < room)) dst[0] = (uint16_t)(p >> 16);
        have += __popcll(m0) + __popcll(m1);
      };
#pragma unroll
      for (int t = 0; t < G::MAXHW; t++) {
        if (have < (uint32_t)N) sample(cur.hw[t], (t + 1) * 128 <= N);
      }
      if (have < (uint32_t)N && ndw > (uint32_t)G::MAXHW * 64u) {  // re-hash pass with more blocks: not prefetched
        const uint32_t* hrow = (const uint32_t*)(A.hraw + (size_t)slot * A.nblk * 17u);
        for (uint32_t base = G::MAXHW * 64u; base < ndw && have < (uint32_t)N; base += 64u) {
          const uint32_t d = base + lane;
          sample(d < ndw ? hrow[d] : 0xffffffffu);
        }
      }
      if (have < (uint32_t)N) {  // vanishingly rare: this item is hashed again with more blocks (host loop or retry pass)
        if (lane == 0) {
          // system scope: the counter of a synchronising call lives in host memory the GPU maps (capi.hip, Workspace)
          A.retry_list[__hip_atomic_fetch_add(A.retry_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)] = A.item_base + item;
          A.accept[item] = 0;  // defined until then
        }
        done = true;
      }
    }

    // ---- what the transforms need, out of the loaded words: s2 (already lifted to [0,q), verification.hpp:36-39) and
    //      the key ----
    FA_MARK("unpack");
    const bool run = status == ST_OK && !done;
    uint32_t (&x)[R] = cur.s2v;  // the first transform works in the registers the loads filled
    uint32_t y[R];
    uint32_t s2keep[DEBUG ? R : 1];
    if (run) {
      if constexpr (DEBUG) {
#pragma unroll
        for (int m = 0; m < R; m++) s2keep[m] = cur.s2v[m];
      }
      if constexpr (!PREPARED) {
        // ---- public key: 14-bit little-endian fields after the header byte (decoding.hpp:34-45), taken
        //      lazily (equivalent to the reference's raw values). The row was loaded
        //      as contiguous dwords; it goes through LDS so each lane can pick its own fields. ----
        unpack_pkey<LOGN>(cur.pkw, cur.pal, pkb, lane, y);
      }
    }
    FA_MARK("prefetch");
    // cur's registers are free: the next item's loads
    const uint32_t nslot = slot + nwaves;
    const bool more = nslot < count;  // wave-uniform
    if (more) prefetch_item<LOGN, PREPARED, GENERAL>(A, nslot, lane, cur);

    FA_MARK("branch");
    if (status != ST_OK) {
      if (lane == 0) {
        A.accept[item] = 0;
        if (DEBUG) {
          if (A.dbg.status) A.dbg.status[item] = (uint8_t)status;
          if (A.dbg.norm_exact) A.dbg.norm_exact[item] = 0ull;
          if (A.dbg.norm_wrap32) A.dbg.norm_wrap32[item] = 0u;
        }
      }
      if (DEBUG) {
        for (int m = 0; m < R; m++) {
          const uint32_t i = idxA<LOGN>(lane, m);
          if (A.dbg.c) A.dbg.c[(size_t)item * N + i] = 0;
          if (A.dbg.s2) A.dbg.s2[(size_t)item * N + i] = 0;
          if (A.dbg.s1) A.dbg.s1[(size_t)item * N + i] = 0;
        }
      }
    } else if (run) {
      if constexpr (PREPARED) {
        ntt_forward<LOGN>(x, T, buf, lane);
#pragma unroll
        for (int m = 0; m < R; m++) {  // 2^32 NTT(h), canonical, already in this lane's phase-C positions
          const uint32_t hv = (cur.hk[m >> 1] >> (16 * (m & 1))) & 0xffffu;
          y[m] = plant2(x[m], hv);  // MINUS the product of polynomial.hpp:99-110; x < 2^16 + 2 logn q, hv < q: below 2^32
        }
      } else {
        // One transform after the other: interleaving them (ntt_forward2) costs Falcon-1024 its third wave per SIMD
        // (31 more VGPRs, 17 KB more LDS) and, with the stages grouped by instruction kind, is 13 % slower.
        FA_MARK("ntt_y");
        ntt_forward<LOGN>(y, T, buf, lane);
        FA_MARK("ntt_x");
        ntt_forward<LOGN>(x, T, buf, lane);
        FA_MARK("pointwise");
        // polynomial.hpp:99-110, as -x*y*2^-32 from BOTH lazy factors in four instructions (zq.hpp plant2_wide: one
        // v_mad_u64_u32 for the whole product, the Plantard reduction of its low word, minus its high word). The 2^32
        // that is missing sits in this path's scaling table (A.isc = zeta^-p / n * 2^32, capi.hip make_tables).
#pragma unroll
        for (int m = 0; m < R; m++) y[m] = plant2_wide(x[m], y[m]);
      }
    }
    // x[] is dead: the next item's s2, into the same registers. One place for every path (an item that ran no transform
    // passes here too). Not under `if (more)`: a conditional load would keep the old values alive beside x[] (they
    // would be what the registers hold on the other path) and bring a copy back; the last iteration loads its own
    // row once more instead. The loads arrive behind the inverse transform and the epilogue; s2_arrived() at the end
    // of the iteration waits for them.
    if constexpr (PREPARED) prefetch_key<LOGN>(A.prepared, cur.key, lane, cur.hk);  // cur.key: the next item's if there is one
    prefetch_s2<LOGN>(A.s2, more ? nslot : slot, lane, cur.s2v);
    if (run) {
      FA_MARK("intt");
      ntt_inverse<LOGN, false>(y, T, buf, lane);  // sc[i] * y = -s2*h mod q, natural order, phase-A layout; y lazy

      // ---- s1 = c - s2*h, centred; squared norm (verification.hpp:53-79) ----
      // s1 = craw + sc*y (mod q), a sum of non-negative values. The centred value v lies in [half - q, half - 1]
      // (verification.hpp:59-68: values >= half = 6144 go negative; the specification's centring has half = 6145), so
      // v + (q - half) is the canonical residue of u = craw + (q - half) + sc*y -- and that is ONE Plantard reduction:
      // the low products are linear, low(y * W_sc) + low(u0 * W_one) is the low product of the sum, and the high product
      // by q reduces it (zq.hpp; exact because y * q + u0 * 1337 < 270,486 q + 79,878 * 1337 < 2^32). Both products are
      // v_mad_u64_u32 (zq.hpp mad64), whose 64-bit addend takes the constants along for nothing: the first adds
      // (q - half) * W_one to the low word, the second subtracts (q - half) from the high word. Per coefficient: product
      // (+ constant), multiply-add of c, product (- constant), multiply-add of the square -- four instructions, no
      // comparison, no select, and the inverse transform's final scaling costs nothing of its own.
      FA_MARK("epilogue");
      uint32_t acc = 0;
      const uint32_t shift = Q - A.half;
      static_assert(((2ull * LOGN + 2ull) * Q + 2ull * G::INV_IN_MIN) * (Q - 1ull) + (61444ull + Q) * NEG_R < (1ull << 32), "fused scaling and reduction");
      const uint64_t add_lo = (uint64_t)(shift * W_ONE);         // low word: + (q - half) * W_one (mod 2^32)
      const uint64_t add_hi = (uint64_t)(0u - shift) << 32;      // high word: - (q - half)      (both wave-uniform: scalar pairs)
      uint32_t qv = Q;                                           // q as a vector operand: the scalar one is the addend
      asm volatile("" : "+v"(qv));
      constexpr int EG = 4;
#pragma unroll
      for (int m0 = 0; m0 < R; m0 += EG) {  // a few at a time: the 64-bit results are register pairs, and six waves per SIMD end at 80 registers
        uint32_t craw[EG], w[EG];
#pragma unroll
        for (int k = 0; k < EG; k++) craw[k] = cbuf[idxA<LOGN>(lane, m0 + k)];  // < 61445
        valu_fence();
#pragma unroll
        for (int k = 0; k < EG; k++) w[k] = (uint32_t)mad64(y[m0 + k], T.sc[(m0 + k) << 6], add_lo);
        valu_fence();
#pragma unroll
        for (int k = 0; k < EG; k++) w[k] += __umul24(craw[k], W_ONE);
        valu_fence();
#pragma unroll
        for (int k = 0; k < EG; k++) w[k] = (uint32_t)(mad64(w[k], qv, add_hi) >> 32);  // v = s1 centred, as i32
        valu_fence();
#pragma unroll
        for (int k = 0; k < EG; k++) {
          const int m = m0 + k;
          const uint32_t i = idxA<LOGN>(lane, m);
          const int32_t v = (int32_t)w[k];
          // acc += v * v (|v| <= 6145), stated as the one instruction it is: left to itself hipcc multiplies the eight
          // squares apart and adds them up in threes, eleven instructions instead of eight
          asm("v_mad_i32_i24 %0, %1, %1, %0" : "+v"(acc) : "v"(v));
          if (DEBUG) {
            if (A.dbg.c) A.dbg.c[(size_t)item * N + i] = (uint16_t)mod_q_17(craw[k]);
            if (A.dbg.s1) A.dbg.s1[(size_t)item * N + i] = (int16_t)v;
            if (A.dbg.s2) {
              const uint32_t sv = s2keep[m];
              A.dbg.s2[(size_t)item * N + i] = (int16_t)(sv > Q / 2u ? (int32_t)sv - (int32_t)Q : (int32_t)sv);
            }
          }
        }
      }
      FA_MARK("norm");
      // per-lane sums are < 2^30 but the wave total needs 36 bits: reduce the low 24 bits and the rest apart.
      // Falcon-512 under the exact policy, no debug outputs: ONE reduction of min(acc, 2^26 - 1). beta^2 < 2^26 - 1, so a
      // lane that is clamped decides the item by itself (the clamped total is still above beta^2, as the true one is),
      // sixty-four clamped values fit 32 bits, and with no lane clamped the total is exact.
      unsigned long long total;
      constexpr uint32_t CLAMP = (1u << 26) - 1u;
      if (!DEBUG && LOGN == 9 && !A.wrap32) {
        static_assert(LOGN != 9 || G::BETA2 < CLAMP, "a clamped lane alone exceeds beta^2");
        total = (unsigned long long)wave_sum_u32(min(acc, CLAMP));
      } else {
        total = (unsigned long long)wave_sum_u32(acc & 0xffffffu) + ((unsigned long long)wave_sum_u32(acc >> 24) << 24);
      }
      total += sqn;
      const uint32_t wrapped = (uint32_t)total;
      const bool ok = A.wrap32 ? ((int32_t)wrapped <= (int32_t)G::BETA2) : (total <= (unsigned long long)G::BETA2);
      if (lane == 0) {
        A.accept[item] = ok ? 1 : 0;
        if (DEBUG) {
          if (A.dbg.status) A.dbg.status[item] = ok ? ST_OK : ST_NORM_TOO_LARGE;
          if (A.dbg.norm_exact) A.dbg.norm_exact[item] = total;
          if (A.dbg.norm_wrap32) A.dbg.norm_wrap32[item] = wrapped;
        }
      }
    }
    s2_arrived<LOGN>(cur.s2v);
    wave_lds_sync();  // cbuf / pkb / buf are reused by the next item
    FA_MARK("end");
    if (!more) break;
    slot = nslot;
  }
}

template <int LOGN, bool DEBUG, bool PREPARED, bool GENERAL>
__global__ __launch_bounds__(64 * verify_waves<LOGN>()) void verify_kernel(const VerifyArgs<LOGN> A)
{
  using G = Geo<LOGN>;
  constexpr int VW = verify_waves<LOGN>();
  __shared__ TwLds<LOGN> s_tw;
  __shared__ uint32_t s_buf[VW][G::XBUF];  // transposes
  __shared__ uint16_t s_c[VW][G::N];
  fill_tw_lds<LOGN>(s_tw, A.tw, A.itw, A.twc, A.isc);
  const uint32_t lane = threadIdx.x & 63u, wave = wave_index();
  const TwTabs T = make_tabs<LOGN>(s_tw, A.nc, lane);
  const uint32_t count = GENERAL && A.dcount ? min(A.count, *A.dcount) : A.count;
  verify_items<LOGN, DEBUG, PREPARED, GENERAL>(A, T, s_buf[wave], s_c[wave], lane, blockIdx.x * VW + wave, gridDim.x * VW, count);
}

// ---- standalone hash_to_point (hashing.hpp:16-50): the rejection-sampling half on its own ----
// One wave per item: the squeezed SHAKE256 blocks of hash_kernel -> c[item][N] (u16, canonical), in draw
// order. Items that run out of blocks are queued in retry_list, as in the verify kernel.
template <int LOGN>
__global__ __launch_bounds__(VERIFY_WG) void sample_kernel(const uint64_t* __restrict__ hraw, const uint32_t* __restrict__ idx,
                                                           uint32_t count, uint32_t nblk, uint32_t item_base,
                                                           uint16_t* __restrict__ c, uint32_t* __restrict__ retry_count,
                                                           uint32_t* __restrict__ retry_list)
{
  constexpr uint32_t N = 1u << LOGN;
  __shared__ uint16_t s_c[VERIFY_WAVES][N];
  const uint32_t lane = threadIdx.x & 63u, wave = wave_index();
  uint16_t* cbuf = s_c[wave];
  const uint32_t ndw = nblk * 34u;
  for (uint32_t slot = blockIdx.x * VERIFY_WAVES + wave; slot < count; slot += gridDim.x * VERIFY_WAVES) {
    const uint32_t item = idx ? idx[slot] : slot;
    const uint32_t* hrow = (const uint32_t*)(hraw + (size_t)slot * nblk * 17u);
    uint32_t have = 0;  // accepted draws so far (wave-uniform)
    for (uint32_t base = 0; base < ndw && have < N; base += 64u) {
      const uint32_t d = base + lane;
      const uint32_t w = d < ndw ? hrow[d] : 0xffffffffu;  // 0xffff draws are always rejected
      const uint32_t t0 = ((w & 0xffu) << 8) | ((w >> 8) & 0xffu);  // big-endian 16-bit draws, hashing.hpp:42-43
      const uint32_t t1 = ((w >> 8) & 0xff00u) | (w >> 24);
      const bool a0 = t0 < 61445u, a1 = t1 < 61445u;                // hashing.hpp:25-28,44
      const unsigned long long m0 = __ballot(a0), m1 = __ballot(a1);
      const uint32_t p0 = have + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u)) +
                          __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
      const uint32_t p1 = p0 + (a0 ? 1u : 0u);
      if (a0 && p0 < N) cbuf[p0] = (uint16_t)mod_q_17(t0);          // ff_t{t}, hashing.hpp:45
      if (a1 && p1 < N) cbuf[p1] = (uint16_t)mod_q_17(t1);
      have += __popcll(m0) + __popcll(m1);
    }
    wave_lds_sync();
    if (have < N) {
      if (lane == 0) retry_list[atomicAdd(retry_count, 1u)] = item_base + item;
    } else {
      uint32_t* out = (uint32_t*)(c + (size_t)item * N);
      const uint32_t* src = (const uint32_t*)cbuf;
      for (uint32_t j = lane; j < N / 2; j += 64u) out[j] = src[j];
    }
    wave_lds_sync();
  }
}

// ---- public-key derivation h = g / f in Z_q[x]/(x^n+1) (keygen::compute_public_key, keygen.hpp:67-90) ----
// One wave per key: lift f, g from [-6145, 6143] to [0, q) (keygen.hpp:80-83), forward transforms, pointwise
// division (polynomial.hpp:119-130; ff_t::inv of 0 is 0, ff.hpp:181-196, as is 0^(q-2)), inverse transform.
template <int LOGN>
__global__ __launch_bounds__(VERIFY_WG) void pubkey_kernel(const int32_t* __restrict__ f, const int32_t* __restrict__ g,
                                                           uint16_t* __restrict__ h, uint32_t count,
                                                           const Tw* __restrict__ tw, const Tw* __restrict__ itw,
                                                           const Tw* __restrict__ twc, const Tw* __restrict__ isc,
                                                           const NttConsts nc)
{
  using G = Geo<LOGN>;
  constexpr int N = G::N, R = G::R;
  __shared__ TwLds<LOGN> s_tw;
  __shared__ uint32_t s_buf[VERIFY_WAVES][G::XBUF];
  __shared__ uint32_t s_buf2[VERIFY_WAVES][G::XBUF];
  fill_tw_lds<LOGN>(s_tw, tw, itw, twc, isc);
  const uint32_t lane = threadIdx.x & 63u, wave = wave_index();
  const TwTabs T = make_tabs<LOGN>(s_tw, nc, lane);
  for (uint32_t key = blockIdx.x * VERIFY_WAVES + wave; key < count; key += gridDim.x * VERIFY_WAVES) {
    uint32_t x[R], y[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
      const int32_t fv = f[(size_t)key * N + idxA<LOGN>(lane, m)], gv = g[(size_t)key * N + idxA<LOGN>(lane, m)];
      x[m] = (uint32_t)reduce(fv);  // [0, q], keygen.hpp:80-83 (any |v| < 2^22 is taken mod q)
      y[m] = (uint32_t)reduce(gv);
    }
    ntt_forward2<LOGN>(x, y, T, s_buf[wave], s_buf2[wave], lane);
#pragma unroll
    for (int m = 0; m < R; m++) {
      const int32_t b = (int32_t)plant(x[m], W_ONE);  // canonical NTT(f) element
      int32_t inv = b;                                            // b^(q-2), q - 2 = 0b10111111111111
#pragma unroll
      for (int bit = 12; bit >= 0; bit--) {
        inv = mulmod(inv, inv);
        if ((12287 >> bit) & 1) inv = mulmod(inv, b);
      }
      x[m] = (uint32_t)mulmod((int32_t)plant(y[m], W_ONE), inv);
    }
    ntt_inverse<LOGN>(x, T, s_buf[wave], lane);
#pragma unroll
    for (int m = 0; m < R; m++) h[(size_t)key * N + idxA<LOGN>(lane, m)] = (uint16_t)x[m];
  }
}

// ---- ntt::ntt<LOG2N> / ntt::intt<LOG2N> (ntt.hpp:120-150, 161-200) as batched operators, out of place ----
// One wave per polynomial. Forward: natural order in, bit-reversed order out -- position k of the output row is
// position k of the reference's array after its in-place transform (the stages here act on the same index
// bits with the same twiddle indices, only distributed over lanes), canonical residues. Inverse: the converse,
// including the 1/N scaling (ntt.hpp:193-199). Inputs may be any u16; they are taken mod q as the ff_t
// constructor does (ff.hpp:69-76).
template <int LOGN, bool INVERSE>
__global__ __launch_bounds__(VERIFY_WG) void ntt_kernel(const uint16_t* __restrict__ a, uint16_t* __restrict__ out, uint32_t count,
                                                        const Tw* __restrict__ tw, const Tw* __restrict__ itw,
                                                        const Tw* __restrict__ twc, const Tw* __restrict__ isc,
                                                        const NttConsts nc)
{
  using G = Geo<LOGN>;
  constexpr int N = G::N, R = G::R;
  __shared__ TwLds<LOGN> s_tw;
  __shared__ uint32_t s_buf[VERIFY_WAVES][G::XBUF];
  fill_tw_lds<LOGN>(s_tw, tw, itw, twc, isc);
  const uint32_t lane = threadIdx.x & 63u, wave = wave_index();
  const TwTabs T = make_tabs<LOGN>(s_tw, nc, lane);
  for (uint32_t p = blockIdx.x * VERIFY_WAVES + wave; p < count; p += gridDim.x * VERIFY_WAVES) {
    uint32_t x[R];
    if constexpr (!INVERSE) {
#pragma unroll
      for (int m = 0; m < R; m++) x[m] = a[(size_t)p * N + idxA<LOGN>(lane, m)];
      ntt_forward<LOGN>(x, T, s_buf[wave], lane);
#pragma unroll
      for (int m = 0; m < R; m++) out[(size_t)p * N + idxC<LOGN>(lane, m)] = (uint16_t)plant(x[m], W_ONE);
    } else {
#pragma unroll
      for (int m = 0; m < R; m++) x[m] = plant(a[(size_t)p * N + idxC<LOGN>(lane, m)], W_ONE);
      ntt_inverse<LOGN>(x, T, s_buf[wave], lane);
#pragma unroll
      for (int m = 0; m < R; m++) out[(size_t)p * N + idxA<LOGN>(lane, m)] = (uint16_t)x[m];
    }
    wave_lds_sync();
  }
}

// ---- decoding::decode_pkey<N> (decoding.hpp:18-48) as a batched operator ----
// One wave per key: h[key][N] = the raw 14-bit little-endian fields (values up to 16383 are stored as they are,
// the reference does not range-check, decoding.hpp:34-45), ok[key] = 1; a wrong header byte gives a zero row
// and ok[key] = 0 (decoding.hpp:29-32).
template <int LOGN>
__global__ __launch_bounds__(VERIFY_WG) void decode_pkeys_kernel(const uint8_t* __restrict__ pk, uint32_t nkeys,
                                                                 uint16_t* __restrict__ h, uint8_t* __restrict__ ok)
{
  using G = Geo<LOGN>;
  constexpr int N = G::N, R = G::R;
  __shared__ uint32_t s_pk[VERIFY_WAVES][G::PKDW * 64];
  const uint32_t lane = threadIdx.x & 63u, wave = wave_index();
  for (uint32_t key = blockIdx.x * VERIFY_WAVES + wave; key < nkeys; key += gridDim.x * VERIFY_WAVES) {
    const uint8_t* pkrow = pk + (size_t)key * G::PKLEN;
    const uint32_t pal = (uint32_t)((uintptr_t)pkrow & 3u);
    const uint32_t* pbase = (const uint32_t*)(pkrow - pal);
    const uint32_t last_dw = (pal + (uint32_t)G::PKLEN - 1u) >> 2;
    uint32_t pkw[G::PKDW];
#pragma unroll
    for (int t = 0; t < G::PKDW; t++) pkw[t] = pbase[min(t * 64u + lane, last_dw)];
    const bool good = ((__builtin_amdgcn_readfirstlane(pkw[0]) >> (8u * __builtin_amdgcn_readfirstlane(pal))) & 0xffu) == (uint32_t)LOGN;
    uint32_t y[R];
    unpack_pkey<LOGN>(pkw, pal, s_pk[wave], lane, y);
#pragma unroll
    for (int m = 0; m < R; m++) h[(size_t)key * N + idxA<LOGN>(lane, m)] = good ? (uint16_t)y[m] : (uint16_t)0;
    if (lane == 0) ok[key] = good ? 1 : 0;
    wave_lds_sync();
  }
}

// ---- specification wire format, public keys (FA_WIRE_SPEC; no counterpart in the reference, README.md:74-75) ----
// Falcon specification section 3.11.4: header 0x00 | logn, then n 14-bit fields packed most significant bit first,
// each < q. Four fields are 7 bytes in either packing, so a key is repacked group by group into the reference's
// little-endian form (decoding.hpp:34-45) and the pipeline runs unchanged on the result. A key with a field >= q
// (invalid in the specification) gets header 0xff, which the pipeline rejects like any wrong header. One wave per
// key, lane L takes groups L, L + 64, ...
template <int LOGN>
__global__ __launch_bounds__(256) void pk_spec_to_ref_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t nkeys)
{
  constexpr uint32_t N = 1u << LOGN, PKLEN = 1u + 14u * N / 8u, GROUPS = N / 4u;
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t key = blockIdx.x * 4u + (threadIdx.x >> 6); key < nkeys; key += gridDim.x * 4u) {
    const uint8_t* src = in + (size_t)key * PKLEN;
    uint8_t* dst = out + (size_t)key * PKLEN;
    bool ok = true;
    for (uint32_t g = lane; g < GROUPS; g += 64u) {
      unsigned long long v = 0;
#pragma unroll
      for (int b = 0; b < 7; b++) v = (v << 8) | src[1u + 7u * g + b];
      unsigned long long w = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t c = (uint32_t)(v >> (14 * (3 - j))) & 0x3fffu;
        ok &= c < Q;
        w |= (unsigned long long)c << (14 * j);
      }
#pragma unroll
      for (int b = 0; b < 7; b++) dst[1u + 7u * g + b] = (uint8_t)(w >> (8 * b));
    }
    const bool all_ok = __ballot(!ok) == 0ull;
    if (lane == 0) dst[0] = all_ok ? src[0] : (uint8_t)0xff;
  }
}

// ---- batched negacyclic product c = a*b in Z_q[x]/(x^n+1) (BASELINE.json config 2) ----
// Same transforms as the verify path: ntt, polynomial::mul, intt (tests/test_ntt.cpp:48-57).
template <int LOGN>
__global__ __launch_bounds__(VERIFY_WG) void polymul_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                            uint16_t* __restrict__ c, uint32_t count,
                                                            const Tw* __restrict__ tw, const Tw* __restrict__ itw,
                                                            const Tw* __restrict__ twc, const Tw* __restrict__ isc,
                                                            const NttConsts nc)  // isc: zeta^-p / n * (-2^32), see below
{
  using G = Geo<LOGN>;
  constexpr int N = G::N, R = G::R;
  __shared__ TwLds<LOGN> s_tw;
  __shared__ uint32_t s_buf[VERIFY_WAVES][G::XBUF];
  __shared__ uint32_t s_buf2[VERIFY_WAVES][G::XBUF];
  fill_tw_lds<LOGN>(s_tw, tw, itw, twc, isc);
  const uint32_t lane = threadIdx.x & 63u, wave = wave_index();
  uint32_t* buf = s_buf[wave];
  const TwTabs T = make_tabs<LOGN>(s_tw, nc, lane);
  const uint32_t nwaves = gridDim.x * VERIFY_WAVES;
  uint32_t p = blockIdx.x * VERIFY_WAVES + wave;
  if (p >= count) return;
  // One row of u16 coefficients into the phase-A layout (element 64 m + lane): scalar row base, one vector byte offset,
  // the register index in the immediate offset. Written as instructions for the reasons prefetch_s2 gives; rows_arrived()
  // is what makes the values usable (tools/check_asm_loads.py checks that nothing touches them before it).
  auto load_row = [&](const uint16_t* base, uint32_t row, uint32_t (&v)[R]) {
    const uint16_t* rb = base + (size_t)__builtin_amdgcn_readfirstlane(row) * N;  // wave-uniform, and known to be
    const uint32_t off = lane * 2u;
#pragma unroll
    for (int m = 0; m < R; m++)
      asm volatile("global_load_ushort %0, %1, %2 offset:%3 ; FA_ASM_LOAD" : "=v"(v[m]) : "v"(off), "s"(rb), "n"(m * 128));
  };
  auto rows_arrived = [&](uint32_t (&u)[R], uint32_t (&v)[R]) {
    asm volatile("s_waitcnt vmcnt(0) ; FA_ASM_LOAD_WAIT" ::: "memory");
#pragma unroll
    for (int m = 0; m < R; m++) asm volatile("" : "+v"(u[m]), "+v"(v[m]));
  };
  // Two arrays and no copies: a's transform and the product live in x[], b's transform in y[]. y[] is dead after the
  // pointwise product and takes the next b at once; x[] is dead after the stores of c and takes the next a then. That
  // last request is waited for straight away -- the other five waves of the SIMD run meanwhile.
  uint32_t x[R], y[R];  // inputs may be any u16 (the reference's ff_t ctor reduces, ff.hpp:69-76): lazy residues
  load_row(a, p, x);
  load_row(b, p, y);
  rows_arrived(x, y);
  while (true) {
    const uint32_t np = p + nwaves;
    const bool more = np < count;
    const uint32_t pn = more ? np : p;  // the last trip loads its own rows once more: no conditional load, no second path
    ntt_forward2<LOGN>(x, y, T, buf, s_buf2[wave], lane);
    static_assert(((65535ull + 2ull * G::BIAS) * (65535ull + 2ull * G::BIAS) >> 32) < G::INV_IN_MIN, "high word of the product");
#pragma unroll
    for (int m = 0; m < R; m++) x[m] = plant2_wide(x[m], y[m]);  // polynomial.hpp:99-110 as -x*y*2^-32; `isc` carries the -2^32
    load_row(b, pn, y);
    ntt_inverse<LOGN>(x, T, buf, lane);
#pragma unroll
    for (int m = 0; m < R; m++) c[(size_t)p * N + idxA<LOGN>(lane, m)] = (uint16_t)x[m];
    load_row(a, pn, x);
    rows_arrived(x, y);
    if (!more) break;
    p = np;
  }
}

}  // namespace fa
