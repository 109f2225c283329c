// hash_kernel.hpp -- SHAKE256 squeeze blocks for hash_to_point, one Keccak sponge per lane.
//
// Replaces the sponge half of hashing::hash_to_point (hashing.hpp:30-39, sha3 submodule):
// absorb salt || msg, pad, then squeeze `nblk` 136-byte blocks per item. The rejection
// sampling half (hashing.hpp:41-48) runs in the verify kernel, which reads the blocks this
// kernel leaves in HBM. One lane owns one sponge: 25 x 64-bit lanes = 50 VGPRs, rotations by
// v_alignbit_b32, theta/chi by the gfx950 three-input v_bitop3_b32 (emitted by hipcc from the
// boolean expressions below), so a round is about 180 VALU instructions.
//
// HBM layout: hraw[slot][nblk][17] little-endian u64 words (row stride nblk*136 bytes). Lanes
// stage each block in LDS and the wave writes it out as contiguous 136-byte runs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "decode_kernel.hpp"  // Rows

namespace fa {

__constant__ uint32_t KECCAK_RC_LO[24] = {
  0x00000001u, 0x00008082u, 0x0000808au, 0x80008000u, 0x0000808bu, 0x80000001u, 0x80008081u, 0x00008009u,
  0x0000008au, 0x00000088u, 0x80008009u, 0x8000000au, 0x8000808bu, 0x0000008bu, 0x00008089u, 0x00008003u,
  0x00008002u, 0x00000080u, 0x0000800au, 0x8000000au, 0x80008081u, 0x00008080u, 0x80000001u, 0x80008008u };
__constant__ uint32_t KECCAK_RC_HI[24] = {
  0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u,
  0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u,
  0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u };

struct u64s {  // a 64-bit Keccak lane as two VGPRs
  uint32_t lo, hi;
};

template <int R>
__device__ __forceinline__ u64s rotl64(u64s v)
{
  if constexpr (R == 0) {
    return v;
  } else if constexpr (R == 32) {
    return u64s{ v.hi, v.lo };
  } else if constexpr (R < 32) {
    return u64s{ __builtin_amdgcn_alignbit(v.lo, v.hi, 32 - R), __builtin_amdgcn_alignbit(v.hi, v.lo, 32 - R) };
  } else {
    return u64s{ __builtin_amdgcn_alignbit(v.hi, v.lo, 64 - R), __builtin_amdgcn_alignbit(v.lo, v.hi, 64 - R) };
  }
}

// three-input XOR in one v_bitop3_b32 (truth table 0x96); hipcc does not form it from a ^ b ^ c by itself
__device__ __forceinline__ u64s xor3(u64s a, u64s b, u64s c)
{
  return u64s{ (uint32_t)__builtin_amdgcn_bitop3_b32((int)a.lo, (int)b.lo, (int)c.lo, 0x96),
               (uint32_t)__builtin_amdgcn_bitop3_b32((int)a.hi, (int)b.hi, (int)c.hi, 0x96) };
}
__device__ __forceinline__ u64s chi(u64s a, u64s b, u64s c) { return u64s{ a.lo ^ (~b.lo & c.lo), a.hi ^ (~b.hi & c.hi) }; }

// One Keccak-f[1600] round on a[x + 5y].
__device__ __forceinline__ void keccak_round(u64s (&a)[25], uint32_t rc_lo, uint32_t rc_hi)
{
  u64s c[5], r[5], b[25];
#pragma unroll
  for (int x = 0; x < 5; x++) c[x] = xor3(xor3(a[x], a[x + 5], a[x + 10]), a[x + 15], a[x + 20]);
#pragma unroll
  for (int x = 0; x < 5; x++) r[x] = rotl64<1>(c[x]);
  // theta: a[x,y] ^= c[x-1] ^ rot(c[x+1], 1)
#pragma unroll
  for (int i = 0; i < 25; i++) a[i] = xor3(a[i], c[(i % 5 + 4) % 5], r[(i % 5 + 1) % 5]);
  // rho + pi: b[y, 2x+3y] = rot(a[x,y], rho[x,y])
  b[0] = a[0];
  b[10] = rotl64<1>(a[1]);
  b[20] = rotl64<62>(a[2]);
  b[5] = rotl64<28>(a[3]);
  b[15] = rotl64<27>(a[4]);
  b[16] = rotl64<36>(a[5]);
  b[1] = rotl64<44>(a[6]);
  b[11] = rotl64<6>(a[7]);
  b[21] = rotl64<55>(a[8]);
  b[6] = rotl64<20>(a[9]);
  b[7] = rotl64<3>(a[10]);
  b[17] = rotl64<10>(a[11]);
  b[2] = rotl64<43>(a[12]);
  b[12] = rotl64<25>(a[13]);
  b[22] = rotl64<39>(a[14]);
  b[23] = rotl64<41>(a[15]);
  b[8] = rotl64<45>(a[16]);
  b[18] = rotl64<15>(a[17]);
  b[3] = rotl64<21>(a[18]);
  b[13] = rotl64<8>(a[19]);
  b[14] = rotl64<18>(a[20]);
  b[24] = rotl64<2>(a[21]);
  b[9] = rotl64<61>(a[22]);
  b[19] = rotl64<56>(a[23]);
  b[4] = rotl64<14>(a[24]);
#pragma unroll
  for (int y = 0; y < 25; y += 5)
#pragma unroll
    for (int x = 0; x < 5; x++) a[y + x] = chi(b[y + x], b[y + (x + 1) % 5], b[y + (x + 2) % 5]);
  a[0].lo ^= rc_lo;
  a[0].hi ^= rc_hi;
}

__device__ __forceinline__ void keccak_f1600(u64s (&a)[25])
{
#pragma unroll 1
  for (int round = 0; round < 24; round++) keccak_round(a, KECCAK_RC_LO[round], KECCAK_RC_HI[round]);
}

// 4 bytes of a byte string at offset `off` (multiple of 4) from p, where p is at byte `al` of
// the aligned dword array `base`; bytes at or beyond `len` read as 0. Loads only dwords that
// hold at least one valid byte, so nothing outside the pages of the string is touched.
__device__ __forceinline__ uint32_t load_bytes4(const uint32_t* base, uint32_t al, int32_t off, int32_t len)
{
  const int32_t rem = len - off;  // valid bytes from off
  if (rem <= 0) return 0u;
  const uint32_t* d = base + (off >> 2);
  uint32_t lo = d[0], hi = 0u;
  if (al != 0 && rem > (int32_t)(4 - al)) hi = d[1];
  uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, al);
  if (rem < 4) v &= (1u << (8 * rem)) - 1u;
  return v;
}

constexpr int HASH_WG = 256;

// One lane per item. sig rows (a Rows map, decode_kernel.hpp): salt = bytes 1..40 (decoding.hpp:389). Messages: fixed length
// mlen at msg + item*mlen when msg_off == nullptr, else msg + msg_off[item] with length
// msg_off[item+1] - msg_off[item] (may differ per lane: the absorb loop diverges, the squeeze
// loop does not). idx (nullable) maps output slot -> item.
// Register budget: the sponge (50 VGPRs), the rho/pi copy and the theta temporaries want about 150 VGPRs =
// three waves per SIMD. Asking for four (128 VGPRs) makes hipcc spill 20 registers and costs 9 % (measured);
// the kernel is VALU-bound at three.
#ifndef FA_HASH_WAVES
#define FA_HASH_WAVES 3
#endif
// The kernel's body for workgroup `block` of WG threads (a template so that front_small_kernel, decode_kernel.hpp's
// companion for small batches, can run it in 64-thread workgroups beside the decoder).
template <int WG>
__device__ __forceinline__ void hash_body(uint32_t block, const Rows& sig,
                                          const uint8_t* __restrict__ msg, const uint64_t* __restrict__ msg_off, uint32_t mlen,
                                          const uint32_t* __restrict__ idx, uint32_t count, const uint32_t* __restrict__ dcount,
                                          uint32_t nblk, uint64_t* __restrict__ hraw)
{
  __shared__ uint64_t stage[WG / 64][64 * 9];  // one pass of a block: 9 words per lane (the second pass uses 8 of them)
  const uint32_t slot = block * WG + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (dcount) count = min(count, *dcount);  // device-side number of slots (retry pass, mixed batches)
  if (block * WG + wave * 64u >= count) return;  // wave-uniform: nothing for this wave (no workgroup barrier below)
  const bool live = slot < count;
  const uint32_t item = live ? (idx ? idx[slot] : slot) : 0u;

  u64s a[25];
#pragma unroll
  for (int i = 0; i < 25; i++) a[i] = u64s{ 0u, 0u };

  if (live) {
    // --- absorb salt || msg || pad, 136 bytes at a time (hashing.hpp:30-33) ---
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
    const int32_t total = 40 + ml;           // bytes before padding
    const int32_t nabs = total / 136 + 1;    // absorb blocks, the last one holds the padding
    for (int32_t blk = 0; blk < nabs; blk++) {
#pragma unroll
      for (int w = 0; w < 17; w++) {
        const int32_t off = blk * 136 + w * 8;  // stream offset of this 64-bit word
        uint32_t lo, hi;
        if (off < 40) {  // salt: 40 bytes = words 0..4 of block 0
          lo = load_bytes4(sbase, sal, off, 40);
          hi = load_bytes4(sbase, sal, off + 4, 40);
        } else {
          lo = load_bytes4(mbase, mal, off - 40, ml);
          hi = load_bytes4(mbase, mal, off - 36, ml);
        }
        // SHAKE domain byte 0x1F right after the data, 0x80 on the last byte of the block
        if (total >= off && total < off + 4) lo |= 0x1Fu << (8 * (total - off));
        if (total >= off + 4 && total < off + 8) hi |= 0x1Fu << (8 * (total - off - 4));
        if (w == 16 && blk == nabs - 1) hi ^= 0x80000000u;
        a[w].lo ^= lo;
        a[w].hi ^= hi;
      }
      if (blk != nabs - 1) keccak_f1600(a);
    }
  }

  // --- squeeze nblk blocks (hashing.hpp:36-39); wave-uniform trip count ---
  // A block leaves the wave through LDS, in two passes (words 0..8, then 9..16: half the staging buffer, so that the LDS
  // does not cap the kernel at four waves per SIMD): lane L writes its words of the pass (as dwords, ds_write2_b32 takes the
  // halves from wherever they are: no register pairs to assemble), then the 64 rows go out SEVEN (EIGHT) rows per store --
  // lane L holds word L % nw of row RP t + L / nw, i.e. LDS word (RP t + L / nw) * 9 + L % nw, and the pieces of a store lie
  // one row stride apart. The point of whole rows per store: a lane's part of the address ((L / nw) * row stride + L % nw)
  // does not depend on t or the block, so it is computed once, and a store's address is one vector add to a scalar base.
  // (17 stores of 64 consecutive words need e / 17 and e % 17 per lane and store.)
  uint32_t* st32 = (uint32_t*)stage[wave];
  const uint64_t* st = stage[wave];
  const uint32_t slot0 = block * WG + wave * 64u;  // first slot of this wave
  const uint32_t row_words = nblk * 17u;
  const bool full = slot0 + 64u <= count;           // wave-uniform: every row of this wave is a slot
  uint64_t* wave_rows = hraw + (size_t)slot0 * row_words;
  for (uint32_t b = 0; b < nblk; b++) {
    keccak_f1600(a);
    uint64_t* blk_rows = wave_rows + (size_t)b * 17u;
    auto pass = [&](auto first_tag) {
      constexpr bool FIRST = decltype(first_tag)::value;
      constexpr int W0 = FIRST ? 0 : 9, NW = FIRST ? 9 : 8, RP = 64 / NW, NST = (64 + RP - 1) / RP;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        st32[(lane * 9u + w) * 2u] = a[W0 + w].lo;
        st32[(lane * 9u + w) * 2u + 1u] = a[W0 + w].hi;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t r0 = lane / (uint32_t)NW, w0 = lane - r0 * (uint32_t)NW;
      const uint32_t lane_words = r0 * row_words + w0 + (uint32_t)W0;
      auto store_rows = [&](auto checked) {  // a full wave (all but the batch's last) tests no row against the count
#pragma unroll
        for (int t = 0; t < NST; t++) {
          const uint32_t row = (uint32_t)(RP * t) + r0;
          bool mine = r0 < (uint32_t)RP && (RP * (t + 1) <= 64 || row < 64u);
          if (decltype(checked)::value) mine = mine && slot0 + row < count;
          uint64_t* rows_t = blk_rows + (size_t)(RP * t) * row_words;  // wave-uniform
          if (mine) rows_t[lane_words] = st[(uint32_t)(RP * t * 9) + r0 * 9u + w0];
        }
      };
      if (full) store_rows(std::false_type{});
      else store_rows(std::true_type{});
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    pass(std::true_type{});
    pass(std::false_type{});
  }
}

// ------------------------------------------------------------------------------------------------
// Small batches: one sponge per LANE PAIR. A pass of a few items is bound by the latency of one lane running
// 9 (17) permutations of 24 x 180 instructions -- a lone wave issues an instruction every 5-6 cycles whatever its
// lanes hold. Here the even lane of a pair keeps the low halves of the 25 Keccak lanes and the odd lane the high
// halves: theta and chi work on halves independently, and a 64-bit rotation needs the partner's half of the same
// word, one DPP quad_perm move away; with it both lanes compute their half of the rotated word by the SAME
// v_alignbit (own, partner, 32 - r for r < 32; partner, own, 64 - r for r > 32). A round is 122 VALU instructions
// instead of 180 (60 bitop3 on the halves, 29 exchanges, 29 alignbits); a wave serves 32 items. Measured, one
// Falcon-512 item: 71 us instead of 86 (a lone wave issues the exchange-alignbit chains more slowly than the plain
// stream, 5.9 against 4.9 cycles an instruction), Falcon-1024 126 instead of 151; it stays ahead up to 2^14 items a
// pass and loses beyond (twice the hash waves). Used by front_small_kernel for passes of at most
// FA_OPT_PAIR_HASH_ITEMS items.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t partner(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);  // quad_perm:[1,0,3,2]
}
template <int R>
__device__ __forceinline__ uint32_t rotl64_half(uint32_t own)  // this lane's half of rotl64(word, R)
{
  if constexpr (R == 0) {
    return own;
  } else if constexpr (R == 32) {
    return partner(own);
  } else if constexpr (R < 32) {
    return __builtin_amdgcn_alignbit(own, partner(own), 32 - R);
  } else {
    return __builtin_amdgcn_alignbit(partner(own), own, 64 - R);
  }
}
__device__ __forceinline__ uint32_t xor3h(uint32_t a, uint32_t b, uint32_t c)
{
  return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, 0x96);
}

__device__ __forceinline__ void keccak_round_pair(uint32_t (&a)[25], uint32_t rc)
{
  uint32_t c[5], r[5], b[25];
#pragma unroll
  for (int x = 0; x < 5; x++) c[x] = xor3h(xor3h(a[x], a[x + 5], a[x + 10]), a[x + 15], a[x + 20]);
#pragma unroll
  for (int x = 0; x < 5; x++) r[x] = rotl64_half<1>(c[x]);
#pragma unroll
  for (int i = 0; i < 25; i++) a[i] = xor3h(a[i], c[(i % 5 + 4) % 5], r[(i % 5 + 1) % 5]);
  b[0] = a[0];
  b[10] = rotl64_half<1>(a[1]);
  b[20] = rotl64_half<62>(a[2]);
  b[5] = rotl64_half<28>(a[3]);
  b[15] = rotl64_half<27>(a[4]);
  b[16] = rotl64_half<36>(a[5]);
  b[1] = rotl64_half<44>(a[6]);
  b[11] = rotl64_half<6>(a[7]);
  b[21] = rotl64_half<55>(a[8]);
  b[6] = rotl64_half<20>(a[9]);
  b[7] = rotl64_half<3>(a[10]);
  b[17] = rotl64_half<10>(a[11]);
  b[2] = rotl64_half<43>(a[12]);
  b[12] = rotl64_half<25>(a[13]);
  b[22] = rotl64_half<39>(a[14]);
  b[23] = rotl64_half<41>(a[15]);
  b[8] = rotl64_half<45>(a[16]);
  b[18] = rotl64_half<15>(a[17]);
  b[3] = rotl64_half<21>(a[18]);
  b[13] = rotl64_half<8>(a[19]);
  b[14] = rotl64_half<18>(a[20]);
  b[24] = rotl64_half<2>(a[21]);
  b[9] = rotl64_half<61>(a[22]);
  b[19] = rotl64_half<56>(a[23]);
  b[4] = rotl64_half<14>(a[24]);
#pragma unroll
  for (int y = 0; y < 25; y += 5)
#pragma unroll
    for (int x = 0; x < 5; x++) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
  a[0] ^= rc;
}

__device__ __forceinline__ void keccak_f1600_pair(uint32_t (&a)[25], uint32_t half)
{
  const uint32_t hi_lane = 0u - half;
#pragma unroll 1
  for (int round = 0; round < 24; round++) {
    const uint32_t lo = KECCAK_RC_LO[round], hi = KECCAK_RC_HI[round];  // scalar loads
    keccak_round_pair(a, lo ^ ((lo ^ hi) & hi_lane));
  }
}

// hash_body for one wave of 32 items (workgroup = one wave). Same inputs, same hraw layout.
__device__ __forceinline__ void hash_pair_body(uint32_t block, const Rows& sig, const uint8_t* __restrict__ msg,
                                               const uint64_t* __restrict__ msg_off, uint32_t mlen,
                                               const uint32_t* __restrict__ idx, uint32_t count, const uint32_t* __restrict__ dcount,
                                               uint32_t nblk, uint64_t* __restrict__ hraw)
{
  __shared__ uint32_t stage[32 * 34];
  const uint32_t lane = threadIdx.x & 63u, half = lane & 1u;
  const uint32_t slot0 = block * 32u, slot = slot0 + (lane >> 1);
  if (dcount) count = min(count, *dcount);
  if (slot0 >= count) return;  // wave-uniform
  const bool live = slot < count;
  const uint32_t item = live ? (idx ? idx[slot] : slot) : 0u;

  uint32_t a[25];
#pragma unroll
  for (int i = 0; i < 25; i++) a[i] = 0u;

  // --- absorb salt || msg || pad (hashing.hpp:30-33): every lane takes its half of each 64-bit word. Both lanes of a
  //     pair run the same number of blocks; idle pairs (beyond count) absorb nothing and are never stored. ---
  int32_t nabs = 0, total = 0, ml = 0;
  const uint32_t *sbase = nullptr, *mbase = nullptr;
  uint32_t sal = 0, mal = 0;
  if (live) {
    const uint8_t* sp = sig.at(item) + 1;
    sal = (uint32_t)((uintptr_t)sp & 3u);
    sbase = (const uint32_t*)(sp - sal);
    const uint8_t* mp;
    if (msg_off) {
      mp = msg + msg_off[item];
      ml = (int32_t)(msg_off[item + 1] - msg_off[item]);
    } else {
      mp = msg + (size_t)item * mlen;
      ml = (int32_t)mlen;
    }
    mal = (uint32_t)((uintptr_t)mp & 3u);
    mbase = (const uint32_t*)(mp - mal);
    total = 40 + ml;
    nabs = total / 136 + 1;
  }
  // The permutation exchanges halves inside a pair only, and both lanes of a pair take the same trips (nabs is the
  // pair's): the loop may diverge between pairs like hash_body's does between lanes.
  for (int32_t blk = 0; blk < nabs; blk++) {
#pragma unroll
    for (int w = 0; w < 17; w++) {
      const int32_t off = blk * 136 + w * 8 + 4 * (int32_t)half;  // stream offset of this lane's dword of the word
      uint32_t v = off < 40 ? load_bytes4(sbase, sal, off, 40) : load_bytes4(mbase, mal, off - 40, ml);
      if (total >= off && total < off + 4) v |= 0x1Fu << (8 * (total - off));  // SHAKE domain byte right after the data
      if (w == 16 && half && blk == nabs - 1) v ^= 0x80000000u;                 // 0x80 on the last byte of the block
      a[w] ^= v;
    }
    if (blk != nabs - 1) keccak_f1600_pair(a, half);
  }

  // --- squeeze nblk blocks (hashing.hpp:36-39) ---
  const size_t row_words = (size_t)nblk * 17u;
  const uint32_t rows = min(32u, count - slot0);
  for (uint32_t b = 0; b < nblk; b++) {
    keccak_f1600_pair(a, half);
#pragma unroll
    for (int w = 0; w < 17; w++) stage[(lane >> 1) * 34u + 2u * w + half] = a[w];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // 32 rows x 34 dwords, written as 17 coalesced wave stores
    uint32_t* out = (uint32_t*)hraw;
#pragma unroll
    for (int t = 0; t < 17; t++) {
      const uint32_t e = t * 64u + lane;
      const uint32_t row = e / 34u, d = e - row * 34u;
      if (row < rows) out[((size_t)(slot0 + row) * row_words + (size_t)b * 17u) * 2u + d] = stage[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

__global__ __launch_bounds__(HASH_WG, FA_HASH_WAVES) void hash_kernel(const Rows sig, const uint8_t* __restrict__ msg,
                                                       const uint64_t* __restrict__ msg_off, uint32_t mlen,
                                                       const uint32_t* __restrict__ idx, uint32_t count,
                                                       const uint32_t* __restrict__ dcount, uint32_t nblk,
                                                       uint64_t* __restrict__ hraw)
{
  hash_body<HASH_WG>(blockIdx.x, sig, msg, msg_off, mlen, idx, count, dcount, nblk, hraw);
}

}  // namespace fa
