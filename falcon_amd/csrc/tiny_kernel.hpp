// tiny_kernel.hpp -- the front of the pipeline for passes of a handful of items: one WAVE per item.
//
// The lane-per-item decoder and sponge (decode_kernel.hpp, hash_kernel.hpp) take as long for one item as for
// 64: 42 us to decode one Falcon-512 signature, 71-86 us to squeeze its nine SHAKE256 blocks -- a lone wave issues an
// instruction every 5-6 cycles whatever its lanes hold. A caller with one signature (falcon::verify<N>,
// falcon.hpp:230-249) waits for exactly that. Here a wave spreads ONE item over its lanes:
//   * hash_wave_body: the 25 x 64-bit Keccak state lives in 50 lanes (one half-word per lane), a round is 20 vector
//     instructions, nine of them ds_bpermute, instead of 180 (hashing.hpp:30-39, sha3 submodule). Measured 148 ns a
//     round, bound by the two LDS round trips in it (unrolling and trimming the instruction stream changed nothing):
//     nine blocks in 31 us instead of 71-86;
//   * decode_wave_body: the code boundaries of the compressed signature are found by pointer doubling over all bit
//     positions (decoding.hpp:267-361): next(p) = p + 9 + k(p) for every p, then next^2, next^4, ... in LDS; the
//     position of coefficient i is the composition of the tables named by the bits of i.
// Both leave exactly what their lane-per-item counterparts leave in the workspace (s2 tiles, sqn, status, hraw rows),
// so the verify kernel that follows is the same. front_tiny_kernel (capi.hip) runs one decoder wave and one hash wave
// per item, as the two waves of one workgroup; the decoder wave (about 10 / 20 us) finishes well inside the hash wave's time. Device-resident call, one
// item: 48 us (Falcon-512), 82 us (Falcon-1024) against 102 / 162 for the lane-per-item front; ahead of it up to about
// 1,024 (512) items a pass, where the decoder's LDS tables (21 / 41 KB a wave) limit how many waves a CU holds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "decode_kernel.hpp"
#include "hash_kernel.hpp"
#include "keccak_lanes.hpp"
#include "verify_kernel.hpp"  // wave_lds_sync

namespace fa {

// ------------------------------------------------------------------------------------------------
// Keccak-f[1600] across lanes. Word (x, y) of the state, half h (0: bits 0-31, 1: bits 32-63), lives in lane
//   32 h + 16 (x / 3) + 5 (x % 3) + y
// i.e. a column (fixed x, y = 0..4) is five consecutive lanes inside one 16-lane DPP row, three columns to a row.
//   theta: the parity of a column gathers in its y = 4 lane by three v_xor_dpp row_shr (1, 2, 4); every lane then
//          fetches C[x-1], C[x+1] and the other half of C[x+1] (ds_bpermute), one v_alignbit makes its half of
//          rot(C[x+1], 1), one v_bitop3 applies D;
//   rho, pi, chi: lane (x, y) needs B[x][y], B[x+1][y], B[x+2][y] with B[X][Y] = rot(A[(X + 3Y) % 5][X], rho): for each
//          of the three, one ds_bpermute per operand of the v_alignbit that forms this lane's half of the rotated
//          word (own half and other half of the source word, in the order the rotation amount dictates), then one
//          v_bitop3 (b0 ^ (~b1 & b2));
//   iota:  lanes 0 and 32.
// The per-lane fetch addresses and shift amounts are constants of the lane: KeccakLane, built at compile time.
// Tried and dropped: ONE round trip a round (chi operands fetched before theta, theta applied to the fetched halves from
// four parity fetches per operand: 18 ds_bpermute, 38 instructions) -- 60 us instead of 48 for the one-item call; a
// lone wave pays about 14 cycles to issue a ds_bpermute and only about 50 for the trip, so the count matters more.
// ------------------------------------------------------------------------------------------------
__constant__ KeccakLaneTable KECCAK_LANES = make_keccak_lanes();

struct KeccakWave {  // this lane's constants, byte addresses for ds_bpermute
  uint32_t cm, cp, cpp, s0[3], s1[3], sh[3];
  uint32_t rc[24];  // this lane's part of the round constants: lane 0 the low halves, lane 32 the high halves, else 0
  uint32_t word;

  __device__ __forceinline__ void init(uint32_t lane)
  {
    const KeccakLane k = KECCAK_LANES.lane[lane];
    cm = 4u * k.cm, cp = 4u * k.cp, cpp = 4u * k.cpp;
#pragma unroll
    for (int j = 0; j < 3; j++) s0[j] = 4u * k.s0[j], s1[j] = 4u * k.s1[j], sh[j] = k.sh[j];
    word = k.word;
#pragma unroll
    for (int r = 0; r < 24; r++) rc[r] = lane == 0u ? KECCAK_RC_LO[r] : lane == 32u ? KECCAK_RC_HI[r] : 0u;
  }
};

__device__ __forceinline__ uint32_t lane_fetch(uint32_t addr, uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)v);
}
// b ^ (lane l - SHR of a's 16-lane row, 0 outside the row): one v_xor_b32_dpp. Written out because hipcc folds only
// some of the equivalent update_dpp + xor pairs and materialises a zero for the others; the s_nop covers the two wait
// states a DPP read of a just-written register needs, which the compiler cannot see through inline assembly.
template <int SHR>
__device__ __forceinline__ uint32_t xor_row_shr(uint32_t a, uint32_t b)
{
  uint32_t d;
  asm volatile("s_nop 1\n\tv_xor_b32_dpp %0, %1, %2 row_shr:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(a), "v"(b), "n"(SHR));
  return d;
}

__device__ __forceinline__ uint32_t keccak_round_wave(uint32_t a, const KeccakWave& K, uint32_t rc)
{
  const uint32_t u = xor_row_shr<1>(a, a);
  const uint32_t v = xor_row_shr<2>(u, u);
  const uint32_t s = xor_row_shr<4>(a, v);  // in a column's y = 4 lane: the column's parity
  const uint32_t cm = lane_fetch(K.cm, s), cp = lane_fetch(K.cp, s), cpp = lane_fetch(K.cpp, s);
  a = (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)cm, (int)__builtin_amdgcn_alignbit(cp, cpp, 31), 0x96);
  uint32_t b[3];
#pragma unroll
  for (int j = 0; j < 3; j++) b[j] = __builtin_amdgcn_alignbit(lane_fetch(K.s0[j], a), lane_fetch(K.s1[j], a), K.sh[j]);
  return (uint32_t)__builtin_amdgcn_bitop3_b32((int)b[0], (int)b[1], (int)b[2], 0xD2) ^ rc;  // b0 ^ (~b1 & b2), iota
}

__device__ __forceinline__ uint32_t keccak_f1600_wave(uint32_t a, const KeccakWave& K)
{
#pragma unroll
  for (int round = 0; round < 24; round++) a = keccak_round_wave(a, K, K.rc[round]);
  return a;
}

// hash_body for ONE item: workgroup (= wave) `slot`. Same inputs and the same hraw row as hash_body.
__device__ __forceinline__ void hash_wave_body(uint32_t slot, const Rows& sig, const uint8_t* __restrict__ msg,
                                               const uint64_t* __restrict__ msg_off, uint32_t mlen,
                                               const uint32_t* __restrict__ idx, uint32_t nblk, uint64_t* __restrict__ hraw)
{
  const uint32_t lane = threadIdx.x & 63u, half = lane >> 5;
  const uint32_t item = idx ? idx[slot] : slot;
  KeccakWave K;
  K.init(lane);
  const bool rate_lane = K.word < 17u;  // holds a word of the 136-byte rate

  // --- absorb salt || msg || pad (hashing.hpp:30-33): a lane of the rate takes its dword of each block ---
  const uint8_t* sp = sig.at(item) + 1;
  const uint32_t sal = (uint32_t)((uintptr_t)sp & 3u);
  const uint32_t* sbase = (const uint32_t*)(sp - sal);
  const uint8_t* mp;
  int32_t ml;
  if (msg_off) {
    mp = msg + msg_off[item];
    ml = (int32_t)(msg_off[item + 1] - msg_off[item]);
  } else {
    mp = msg + (size_t)item * mlen;
    ml = (int32_t)mlen;
  }
  const uint32_t mal = (uint32_t)((uintptr_t)mp & 3u);
  const uint32_t* mbase = (const uint32_t*)(mp - mal);
  const int32_t total = 40 + ml, nabs = total / 136 + 1;  // wave-uniform: one item

  uint32_t a = 0u;
  for (int32_t blk = 0; blk < nabs; blk++) {
    if (rate_lane) {
      const int32_t off = blk * 136 + (int32_t)K.word * 8 + 4 * (int32_t)half;  // stream offset of this lane's dword
      uint32_t v = off < 40 ? load_bytes4(sbase, sal, off, 40) : load_bytes4(mbase, mal, off - 40, ml);
      if (total >= off && total < off + 4) v |= 0x1Fu << (8 * (total - off));  // SHAKE domain byte right after the data
      if (K.word == 16u && half && blk == nabs - 1) v ^= 0x80000000u;           // 0x80 on the last byte of the block
      a ^= v;
    }
    if (blk != nabs - 1) a = keccak_f1600_wave(a, K);
  }

  // --- squeeze nblk blocks (hashing.hpp:36-39): the 34 rate lanes write the block's 136 bytes ---
  uint32_t* out = (uint32_t*)(hraw + (size_t)slot * nblk * 17u) + 2u * K.word + half;
  for (uint32_t b = 0; b < nblk; b++) {
    a = keccak_f1600_wave(a, K);
    if (rate_lane) out[(size_t)b * 34u] = a;
  }
}

// ------------------------------------------------------------------------------------------------
// Compressed-signature decoder, one wave per signature (decoding.hpp:267-361; the rules are decode_kernel.hpp's).
//   1. The body (bytes 41..) goes to LDS as big-endian dwords, zero beyond its end.
//   2. For EVERY bit position p: J[p] = the position after a code that starts at p: p + 9 + k(p) (k = zeros in the 8-bit
//      window behind the 8 head bits) or, when the window holds no 1, where the reference's window rule lands; clamped
//      to a last position that maps to itself.
//   3. LOGN - 1 times: J <- J o J (two tables, ping-pong). Before each doubling, every lane moves the cursors of its
//      coefficients i = lane + 64 m whose bit j is set through the table of level j: jumps by powers of the same
//      function commute, so after LOGN levels cursor i is next^i(0).
//   4. Every lane decodes its coefficients at their cursors. "-0" or a last code ending at or beyond the last bit fail
//      (decoding.hpp:327-330, 340), and the bits behind the last code must be zero (decoding.hpp:342-357). A code whose
//      run of empty windows is longer than eight, and in specification mode any k = 8 code, sends the item to the
//      generic decoder, run by lane 0 on the spot (about 90 us).
// ------------------------------------------------------------------------------------------------
template <int LOGN>
struct DecodeWaveGeo {
  static constexpr uint32_t N = 1u << LOGN, SIGLEN = LOGN == 9 ? 666u : 1280u, BODY = SIGLEN - 41u;
  static constexpr uint32_t BITS = 8u * BODY;
  static constexpr uint32_t NPOS = (BITS + 17u + 127u) / 128u * 128u;  // positions with a table entry; multiple of 128
  static constexpr uint32_t NDW = NPOS / 32u + 2u;                     // dwords of the bit image (a window reads two)
};

template <int LOGN>
__device__ __forceinline__ void decode_wave_body(uint32_t slot, const Rows& sig, const uint32_t* __restrict__ idx,
                                                 uint16_t* __restrict__ s2, uint32_t* __restrict__ sqn,
                                                 uint8_t* __restrict__ st, uint32_t spec)
{
  using G = DecodeWaveGeo<LOGN>;
  constexpr uint32_t N = G::N, BODY = G::BODY, BITS = G::BITS, NPOS = G::NPOS, NDW = G::NDW, LAST = NPOS - 1u;
  __shared__ uint32_t image[NDW];
  __shared__ uint32_t table[2][NPOS / 2u];  // u16 pairs: entry p in the low (p even) or high half of dword p / 2
  const uint32_t lane = threadIdx.x & 63u;
  const uint8_t* row = sig.at(idx ? idx[slot] : slot);

  if (row[0] != (uint8_t)(0x30 | LOGN)) {  // decoding.hpp:376-382
    if (lane == 0) {
      st[slot] = ST_BAD_SIG_HEADER;
      sqn[slot] = 0u;
    }
    return;
  }

  // 1. bit image
  {
    const uint8_t* body = row + 41;
    const uint32_t al = (uint32_t)((uintptr_t)body & 3u);
    const uint32_t* base = (const uint32_t*)(body - al);
    for (uint32_t d = lane; d < NDW; d += 64u) image[d] = __builtin_bswap32(load_bytes4(base, al, (int32_t)(4u * d), (int32_t)BODY));
  }
  wave_lds_sync();
  auto window = [&](uint32_t p) -> uint32_t {  // the 32 bits from position p on
    const uint32_t d = p >> 5;
    return (uint32_t)(((((uint64_t)image[d] << 32) | image[d + 1u]) << (p & 31u)) >> 32);
  };

  // The position behind a code that starts at p and has no 1 in its window (k = 8), by the reference's rule
  // (decoding.hpp:300-320, 335): whole 8-bit windows are skipped until one holds a 1 or the stream ends, then one more
  // bit. Up to eight windows are followed here; a longer run, and the specification's different rule, answer SLOW --
  // a second position that maps to itself, "a code here needs the generic decoder".
  constexpr uint32_t SLOW = NPOS - 2u;
  int32_t last_one = -1;  // position of the last 1 of the body: behind it every window is empty and a run ends at the end of the stream
  for (uint32_t d = lane; d < NPOS / 32u; d += 64u)
    if (image[d]) last_one = max(last_one, (int32_t)(32u * d + 31u - (uint32_t)__builtin_ctz(image[d])));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) last_one = max(last_one, __shfl_xor(last_one, o));
  auto jump8 = [&](uint32_t p) -> uint32_t {
    if (spec) return SLOW;
    uint32_t q = p + 16u;
    if ((int32_t)q > last_one) return BITS + 1u;  // nothing but zeros from q on: the scan stops at the end, then the extra bit
    bool found = false;
#pragma unroll 1
    for (int it = 0; it < 8 && q < BITS && !found; it++) {
      const uint32_t e = min(8u, BITS - q);
      found = (window(q) >> (32u - e)) != 0u;
      q += e;
    }
    return !found && q < BITS ? SLOW : min(q + 1u, LAST);
  };

  // 2. level 0: two neighbouring positions per lane and step, so that a table dword is written whole
  for (uint32_t q = lane; q < NPOS / 2u; q += 64u) {
    const uint32_t d = q >> 4;
    const uint64_t bits = (((uint64_t)image[d] << 32) | image[d + 1u]) << ((2u * q) & 31u);  // >= 33 valid bits
    const uint32_t w0 = (uint32_t)(bits >> 32), w1 = (uint32_t)(bits >> 31);
    const uint32_t k0 = (uint32_t)__builtin_clz((w0 << 8) | 0x00800000u), k1 = (uint32_t)__builtin_clz((w1 << 8) | 0x00800000u);
    uint32_t j0 = min(2u * q + 9u + k0, LAST), j1 = min(2u * q + 10u + k1, LAST);
    if (k0 == 8u) j0 = jump8(2u * q);
    if (k1 == 8u) j1 = jump8(2u * q + 1u);
    if (2u * q >= BITS) j0 = LAST;       // a code that would start at or beyond the end: decoding stops (decoding.hpp:281) --
    if (2u * q + 1u >= BITS) j1 = LAST;  // straight to the last position, so that SLOW is reached by a real long run only
    if (2u * q >= SLOW) j0 = SLOW;       // SLOW and LAST map to themselves
    table[0][q] = j0 | (j1 << 16);
  }
  wave_lds_sync();

  // 3. cursors and doubling
  constexpr uint32_t PER = N / 64u;
  uint32_t cur[PER];
#pragma unroll
  for (uint32_t m = 0; m < PER; m++) cur[m] = 0u;
  auto entry = [&](const uint32_t* t, uint32_t p) -> uint32_t { return ((const uint16_t*)t)[p]; };
#pragma unroll 1
  for (uint32_t level = 0; level < (uint32_t)LOGN; level++) {
    const uint32_t* t = table[level & 1u];
#pragma unroll
    for (uint32_t m = 0; m < PER; m++) {
      const uint32_t i = lane + 64u * m;
      if ((i >> level) & 1u) cur[m] = entry(t, cur[m]);
    }
    if (level + 1u < (uint32_t)LOGN) {
      uint32_t* nt = table[(level + 1u) & 1u];
#pragma unroll 4
      for (uint32_t q = lane; q < NPOS / 2u; q += 64u) {
        const uint32_t pair = t[q];
        nt[q] = entry(t, pair & 0xffffu) | (entry(t, pair >> 16) << 16);
      }
      wave_lds_sync();
    }
  }

  // 4. coefficients
  uint32_t norm = 0u, maxv = 0u, endpos = 0u;
  bool slow = false;
  uint32_t val[PER];
#pragma unroll
  for (uint32_t m = 0; m < PER; m++) {
    const uint32_t top = window(cur[m]);
    const uint32_t k = (uint32_t)__builtin_clz((top << 8) | 0x00800000u);
    const uint32_t mag = ((top >> 24) & 0x7fu) + (k < 8u ? k << 7 : 0u);  // a k = 8 code keeps its low 7 bits only (decoding.hpp:300-320)
    const uint32_t neg = (uint32_t)((int32_t)top >> 31);
    val[m] = (mag ^ neg) + (neg & (Q + 1u));  // sign ? q - mag : mag (verification.hpp:36-39)
    norm += mag * mag;
    maxv = max(maxv, val[m]);
    slow |= cur[m] == SLOW || (k == 8u && spec != 0u);
    if (m == PER - 1u) endpos = k < 8u ? cur[m] + 9u + k : jump8(cur[m]);  // lane 63: the position behind the last code
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    norm += __shfl_xor(norm, o);
    maxv = max(maxv, __shfl_xor(maxv, o));
  }
  endpos = __shfl(endpos, 63);

  if (__ballot(slow) != 0ull || endpos == SLOW) {  // wave-uniform: the rare shapes go through the generic decoder, every rule in one place
    // The row pointer is wave-uniform here, and left to itself hipcc turns the generic decoder's 16-byte loads into
    // scalar loads whose immediate offset (41) and base are each unaligned with an aligned sum -- the scalar unit drops
    // the low two bits of every address component on its own and reads the wrong dwords (seen on gfx950: every item
    // failed). The always-zero mbcnt makes the address a vector value, the loads vector loads.
    if (lane == 0) decode_generic_item<LOGN>(row + __builtin_amdgcn_mbcnt_lo(0u, 0u), slot, s2, sqn, st, spec);
    return;
  }
  // a cursor at or beyond the end of the body leaves endpos there too (positions only grow): decoding.hpp:281, 340
  bool failed = (spec ? endpos > BITS : endpos >= BITS) || maxv == Q;
  uint32_t tail = 0u;  // bits behind the last code
  for (uint32_t d = lane; d < NPOS / 32u; d += 64u) {
    const uint32_t first = 32u * d;
    if (first + 32u > endpos) tail |= image[d] & (first >= endpos ? ~0u : ~0u >> (endpos - first));
  }
  failed |= __ballot(tail != 0u) != 0ull;
#pragma unroll
  for (uint32_t m = 0; m < PER; m++) s2[s2_tile_offset<LOGN>(slot, lane + 64u * m)] = (uint16_t)val[m];
  if (lane == 0) {
    st[slot] = failed ? ST_BAD_ENCODING : ST_OK;
    sqn[slot] = norm;
  }
}

}  // namespace fa
