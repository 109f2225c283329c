// decode_kernel.hpp -- compressed-signature decoder, one signature per lane.
//
// Replaces decoding::decode_sig / decompress_sig (decoding.hpp:267-391) including every
// non-spec behaviour of the reference that SURVEY.md section 8(a5) lists:
//   * header must be 0x30 | log2 n;
//   * per coefficient: 8 bits (sign, low 7), then the unary part read through an 8-bit window;
//     k < 8 zeros: value += 128k; k == 8: whole 8-bit windows are skipped until one holds a 1
//     (or the stream ends), the high part is dropped and the cursor lands at the END of that
//     window, then one more bit is skipped as in the normal path;
//   * "-0" is rejected; a stream whose last coefficient ends on or after the last bit is
//     rejected; bits after the last coefficient must all be 0.
// Bits at or beyond the end of the signature read as 0 (the reference over-reads up to two
// bytes there; no outcome depends on them, DESIGN.md "decoder over-read").
//
// Output per item: s2 lifted to [0, q) as u16 in the tiled layout of s2_tile_offset below (the lift of
// verification.hpp:36-39), the partial squared norm sum s2[i]^2, and a status byte.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zq.hpp"

namespace fa {

enum : uint8_t { ST_OK = 0, ST_BAD_PK_HEADER = 1, ST_BAD_SIG_HEADER = 2, ST_BAD_ENCODING = 3, ST_NORM_TOO_LARGE = 4,
                 ST_HASH_UNDERFLOW = 5 /* internal: needs more squeeze blocks */ };

constexpr int DECODE_WG = 256;

// Layout of the s2 workspace (internal to the pipeline, written by the decoders and read by the
// verify kernel): tiles of 64 slots x 32 coefficients, s2t[slot / 64][i / 32][slot % 64][i % 32] (u16).
// A wave of the fast decoder (64 consecutive slots) thus emits one contiguous 4 KiB tile per chunk of 32
// coefficients -- full cache lines -- instead of 64-byte pieces scattered at row stride.
template <int LOGN>
__host__ __device__ __forceinline__ size_t s2_tile_offset(uint32_t slot, uint32_t i)
{
  return (((size_t)(slot >> 6) * ((1u << LOGN) / 32u) + (i >> 5)) * 64u + (slot & 63u)) * 32u + (i & 31u);
}
template <int LOGN>
__host__ __device__ __forceinline__ size_t s2_tile_elems(size_t slots) { return ((slots + 63u) / 64u * 64u) << LOGN; }

// Where the records (signatures, public keys, salts) of a pass live: the record of item i starts at base + off[i]
// when `off` is given (mixed Falcon-512/1024 batches, fa_verify_mixed_dev), else at base + i * stride.
struct Rows {
  const uint8_t* base;
  const uint64_t* off;  // nullable
  uint32_t stride;
  __device__ __forceinline__ const uint8_t* at(uint32_t item) const
  {
    return base + (off ? (size_t)off[item] : (size_t)item * stride);
  }
};

// MSB-first bit reader over the signature body (bytes 41..siglen-1), fed by aligned 16-byte loads.
struct BitReader {
  const uint4* base;     // 16-byte aligned address at or below body byte 0
  uint32_t al;           // body byte 0 sits at byte `al` (0..15) of base[0]
  int32_t body_bytes;    // siglen - 41
  int32_t next;          // next dword index (from base) to hand out
  uint4 cache;           // base[next >> 2] while (next & 3) != 0
  uint64_t buf;          // MSB-aligned bit buffer
  int32_t nbits;         // valid bits in buf

  __device__ __forceinline__ uint32_t fetch()
  {
    // dword `next` covers body bytes [4*next - al, 4*next - al + 4); bytes outside the body -> 0
    const int32_t first = 4 * next - (int32_t)al;
    uint32_t w = 0u;
    if (first < body_bytes) {
      if ((next & 3) == 0 || first <= 0) cache = base[next >> 2];  // never a piece that lies wholly past the row
      const uint32_t sel = (uint32_t)next & 3u;
      const uint32_t raw = sel == 0u ? cache.x : sel == 1u ? cache.y : sel == 2u ? cache.z : cache.w;
      w = __builtin_bswap32(raw);  // byte k of the dword -> bits 31-8k .. 24-8k
      const int32_t valid = body_bytes - first;  // > 0
      if (valid < 4) w &= ~0u << (8 * (4 - valid));
      if (first < 0) w &= ~0u >> (8 * (-first));  // bytes before the body (only for the first dword handed out)
    }
    next++;
    return w;
  }

  __device__ __forceinline__ void init(const uint8_t* body, int32_t nbytes)
  {
    al = (uint32_t)((uintptr_t)body & 15u);
    base = (const uint4*)(body - al);
    body_bytes = nbytes;
    next = (int32_t)(al >> 2);  // the dword that holds body byte 0
    const uint32_t w0 = fetch();
    buf = (uint64_t)w0 << (32 + 8 * (al & 3u));
    nbits = 32 - 8 * (int32_t)(al & 3u);
    refill();
  }

  // keep at least 33 valid bits in the buffer
  __device__ __forceinline__ void refill()
  {
    if (nbits <= 32) {
      buf |= (uint64_t)fetch() << (32 - nbits);
      nbits += 32;
    }
  }

  __device__ __forceinline__ uint32_t peek32() const { return (uint32_t)(buf >> 32); }

  __device__ __forceinline__ void skip(int32_t n)  // n <= 32
  {
    buf <<= n;
    nbits -= n;
  }
};

// Generic decoder (every rule, including the k >= 8 windows), one signature per lane. sig rows -> s2 (u16,
// lifted, tiled layout, written 8 coefficients = 16 bytes at a time), sqn[slot] = sum s2^2, st[slot]. Thread t
// works on workspace slot list[t] (list nullable: slot t; the list is the fix-up queue of decode_fast_kernel),
// whose signature is item idx[slot] (idx nullable: item = slot). The number of threads that work is
// min(count, *dcount) when dcount is given (a device-side list length). For an item that fails, the s2 row
// is left unspecified (nothing reads it).
template <int LOGN>
__device__ __forceinline__ void decode_generic_item(const uint8_t* __restrict__ row, uint32_t slot, uint16_t* __restrict__ s2,
                                                    uint32_t* __restrict__ sqn, uint8_t* __restrict__ st, uint32_t spec)
{
  constexpr int N = 1 << LOGN;
  constexpr int SIGLEN = LOGN == 9 ? 666 : 1280;
  constexpr int32_t SLEN = 8 * SIGLEN;  // in bits, decoding.hpp:274

  if (row[0] != (uint8_t)(0x30 | LOGN)) {  // decoding.hpp:376-382
    st[slot] = ST_BAD_SIG_HEADER;
    sqn[slot] = 0u;
    return;
  }

  BitReader br;
  br.init(row + 41, SIGLEN - 41);
  int32_t pos = 8 + 320;  // absolute bit index, decoding.hpp:276-277
  int32_t i = 0;
  bool failed = false, stop = false;  // stop: the loop of decoding.hpp:281 has ended
  uint32_t norm = 0u;

  for (int g = 0; g < N / 8 && !stop; g++) {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      v[j] = 0u;
      stop |= pos >= SLEN;  // decoding.hpp:281
      if (stop) continue;
      br.refill();
      const uint32_t top = br.peek32();
      const uint32_t sign = top >> 31;
      uint32_t mag = (top >> 24) & 0x7fu;
      const uint32_t k = min((uint32_t)__builtin_clz((top << 8) | 0x00800000u), 8u);  // zeros in the 8-bit window
      if (k < 8) {  // decoding.hpp:297-299
        mag += k << 7;
        br.skip(9 + (int32_t)k);
        pos += 9 + (int32_t)k;
      } else if (spec) {  // specification (Algorithm 18): the unary run simply goes on, at most 15 zeros (|s| <= 2047)
        br.skip(16);
        pos += 16;
        uint32_t kk = 8u;
        bool term = false;
        while (!term && pos < SLEN && kk <= 15u) {
          br.refill();
          term = (br.peek32() >> 31) != 0u;
          br.skip(1);
          pos += 1;
          if (!term) kk++;
        }
        if (!term) {  // ran out of bits or more than 15 zeros
          failed = true;
          stop = true;
          continue;
        }
        mag += kk << 7;
      } else {  // decoding.hpp:300-320: skip whole windows until one holds a 1 or the stream ends
        br.skip(16);
        pos += 16;
        while (pos < SLEN) {
          br.refill();
          const int32_t ebits = min(8, SLEN - pos);
          const uint32_t t = br.peek32() >> (32 - ebits);
          br.skip(ebits);
          pos += ebits;
          if (t != 0u) break;
        }
        br.refill();
        br.skip(1);  // decoding.hpp:335
        pos += 1;
      }
      if (sign && mag == 0u) {  // decoding.hpp:327-330
        failed = true;
        stop = true;
        continue;
      }
      v[j] = sign ? Q - mag : mag;
      norm += mag * mag;
      i++;
    }
    *(uint4*)(s2 + s2_tile_offset<LOGN>(slot, 8u * (uint32_t)g)) =
      make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
  }

  failed |= (spec ? pos > SLEN : pos >= SLEN) | (i < N);  // decoding.hpp:340; the specification lets the last code end on the last bit
  if (!failed) {  // decoding.hpp:342-357: the rest of the stream must be zero
    while (pos < SLEN) {
      br.refill();
      const int32_t e = min(32, SLEN - pos);
      const uint32_t t = br.peek32() >> (32 - e);
      failed |= t != 0u;
      br.skip(e);
      pos += e;
    }
  }
  st[slot] = failed ? ST_BAD_ENCODING : ST_OK;
  sqn[slot] = norm;
}

template <int LOGN>
__global__ __launch_bounds__(DECODE_WG) void decode_kernel(const Rows sig, const uint32_t* __restrict__ list,
                                                           const uint32_t* __restrict__ idx, uint32_t count,
                                                           const uint32_t* __restrict__ dcount,
                                                           uint16_t* __restrict__ s2, uint32_t* __restrict__ sqn,
                                                           uint8_t* __restrict__ st, uint32_t spec)
{
  const uint32_t t = blockIdx.x * DECODE_WG + threadIdx.x;
  if (dcount) count = min(count, *dcount);
  if (t >= count) return;
  const uint32_t slot = list ? list[t] : t;        // workspace slot (list: the fix-up pass behind decode_fast_kernel)
  const uint32_t item = idx ? idx[slot] : slot;    // slot -> item of the caller's arrays
  decode_generic_item<LOGN>(sig.at(item), slot, s2, sqn, st, spec);
}

// ------------------------------------------------------------------------------------------------
// Fast path: one signature per lane, no bit buffer and no per-coefficient control flow.
//   * Each lane owns a 128-byte ring in LDS holding the next 32 big-endian dwords of its row. A
//     coefficient is decoded from ONE aligned ds_read2_b32 (the dword its code starts in and the next)
//     and a 64-bit shift; an unaligned ds_read_b32 would save an instruction but stalls the LDS for the
//     whole wave (SQ_LDS_UNALIGNED_STALL). The cursor is a bit position, so there is no refill branch;
//     codes are 9..16 bits, hence a chunk of 32 coefficients consumes at most 64 bytes. Stream dword n
//     sits at ring index 32 - (n & 31) (descending, so the pair comes back in {low, high} register
//     order); index 0 mirrors index 32 for the pair that straddles the wrap.
//   * The ring is refilled in 16-byte pieces: the pieces the chunk after next can need (at most 4 per
//     row) are requested at the end of a chunk, four lanes per row, into registers, and stored to LDS one
//     chunk later, so the HBM latency hides behind a chunk's arithmetic and every piece is fetched once.
//   * Nothing is predicated per coefficient: a lane keeps decoding whatever bits it sees and the
//     verdict is derived afterwards from three running values -- the largest unary count (8 = the
//     k >= 8 window rule, which only decode_kernel implements: the item is queued for it), the largest
//     output value (q = "-0"), and the final cursor (at or past the last bit = rejected,
//     decoding.hpp:340). Bits past the end of a row are whatever follows in memory; every path that
//     looks at them ends in one of those three outcomes, so they never influence a result.
//   * Output: the 64 x 32 coefficients of a chunk form one contiguous 4 KiB tile (s2_tile_offset); it is
//     turned through LDS and leaves as four fully coalesced 1 KiB stores.
// ------------------------------------------------------------------------------------------------
// s_waitcnt vmcnt(0), stated once on every path. The loads of a refill are consumed under per-lane conditions; left
// to itself hipcc puts its wait inside each conditional block, so on the path around the block the load still counts
// as outstanding, and the next refill -- which reuses those registers -- then waits for ALL memory operations before
// EACH of its loads: the four loads of a refill (and the eight of the prologue) went out one HBM round trip after
// the other. With the wait in front of the blocks the compiler knows the registers are free and the loads issue
// back to back.
__device__ __forceinline__ void vmem_wait_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0), expcnt/lgkmcnt untouched

constexpr int DF_CH = 32;        // coefficients per chunk
constexpr int DF_RING_DW = 33;   // dwords per lane: mirror + 32 ring dwords (odd stride: conflict free)
constexpr int DF_OUT = 20;       // dwords per lane of the output stage: 16 + 4 (keeps 16-byte alignment, spreads banks)
constexpr uint32_t DF_AHEAD = 72;  // bytes beyond the cursor a chunk may touch: 64 consumed + one dword pair

// The kernel's body for workgroup (= wave) `block`; also run by front_small_kernel (capi.hip).
template <int LOGN>
__device__ __forceinline__ void decode_fast_body(uint32_t block, const Rows& sig, const uint32_t* __restrict__ idx,
                                                 uint32_t count, const uint32_t* __restrict__ dcount,
                                                 uint16_t* __restrict__ s2, uint32_t* __restrict__ sqn,
                                                 uint8_t* __restrict__ st, uint32_t* __restrict__ slow_count,
                                                 uint32_t* __restrict__ slow_list, uint32_t spec)
{
  constexpr int N = 1 << LOGN;
  constexpr uint32_t SIGLEN = LOGN == 9 ? 666 : 1280;
  constexpr uint32_t BODY = SIGLEN - 41;
  __shared__ uint32_t ring[64 * DF_RING_DW];
  __shared__ __attribute__((aligned(16))) uint32_t outb[64 * DF_OUT];
  __shared__ uint32_t s_ld[64], s_want[64];  // per row: pieces in the ring, pieces wanted

  const uint32_t lane = threadIdx.x;
  const uint32_t slot0 = block * 64u;
  const uint32_t slot = slot0 + lane;
  if (dcount) count = min(count, *dcount);  // device-side number of slots (mixed batches); wave-uniform
  if (slot0 >= count) return;
  const bool live = slot < count;
  const uint32_t row = live ? slot : count - 1u;  // idle lanes shadow the last row and write nothing
  // The signature bodies this lane deals with (slot -> item through idx, item -> address through sig): its own row's, the
  // eight rows' it helps to load in the prologue, the four rows' it helps to refill. All thirteen addresses are formed
  // here, the optional index and offset loads of all of them under ONE test each: written per row (test, load, use) every
  // row's address waited for the previous row's loads -- the optional loads meet the unconditional ones at a join, and
  // hipcc waits for everything outstanding there even on the path that loaded nothing -- so the eight piece loads of the
  // prologue went out one HBM round trip after the other at the start of every wave.
  const uint8_t* bodies[13];
  {
    uint32_t rws[13], items[13];
    rws[0] = row;
#pragma unroll
    for (int t = 0; t < 8; t++) rws[1 + t] = min(slot0 + 8u * t + (lane >> 3), count - 1u);
#pragma unroll
    for (int t = 0; t < 4; t++) rws[9 + t] = min(slot0 + 16u * t + (lane >> 2), count - 1u);
    if (idx) {
#pragma unroll
      for (int i = 0; i < 13; i++) items[i] = idx[rws[i]];
    } else {
#pragma unroll
      for (int i = 0; i < 13; i++) items[i] = rws[i];
    }
    if (sig.off) {
#pragma unroll
      for (int i = 0; i < 13; i++) bodies[i] = sig.base + (size_t)sig.off[items[i]] + 41;
    } else {
#pragma unroll
      for (int i = 0; i < 13; i++) bodies[i] = sig.base + (size_t)items[i] * sig.stride + 41;
    }
  }

  // geometry of this lane's row, relative to the 16-byte aligned address at or below the body (`lbase` below)
  const uint8_t* body = bodies[0];
  const uint32_t boff = (uint32_t)((uintptr_t)body & 15u);
  const uint32_t end_bits = 8u * (boff + BODY);        // first bit past the signature
  const uint32_t plast = (boff + BODY - 1u) >> 4;      // last piece holding a byte of the row (never read past it)
  uint32_t* myring = ring + lane * DF_RING_DW;

  auto put_piece = [&](uint32_t* rowring, uint32_t piece, const uint4& v) {  // stream dwords 4*piece .. 4*piece+3
    uint32_t* dst = rowring + 32u - (piece & 7u) * 4u;
    dst[0] = __builtin_bswap32(v.x);
    dst[-1] = __builtin_bswap32(v.y);
    dst[-2] = __builtin_bswap32(v.z);
    dst[-3] = __builtin_bswap32(v.w);
    if ((piece & 7u) == 0u) rowring[0] = __builtin_bswap32(v.x);  // mirror of index 32
  };
  // the stream from bit p (relative to the aligned base) on, MSB first, in the top 64 - (p & 31) bits
  auto window64 = [&](uint32_t p) -> uint64_t {
    const uint32_t* src = myring + 31u - ((p >> 5) & 31u);  // {next dword, this dword}
    return (((uint64_t)src[1] << 32) | src[0]) << (p & 31u);
  };
  auto window = [&](uint32_t p) -> uint32_t { return (uint32_t)(window64(p) >> 32); };

  uint32_t pos = 8u * boff;  // bit cursor relative to the aligned base; body bit 0 == absolute bit 328 (decoding.hpp:276-277)
  uint32_t ld = 0;           // pieces [0, ld) have been put into the ring
  {  // prologue: pieces 0..5 of every row (enough for the first chunk), eight lanes per row
    uint4 v[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const uint8_t* b = bodies[1 + t];
      const uint32_t last = ((uint32_t)((uintptr_t)b & 15u) + BODY - 1u) >> 4;
      v[t] = ((const uint4*)(b - ((uintptr_t)b & 15u)))[min(lane & 7u, last)];  // lanes 6, 7 of a group load too (and drop it):
    }                                                                          // unconditional loads issue back to back
    vmem_wait_all();
#pragma unroll
    for (int t = 0; t < 8; t++)
      if ((lane & 7u) < 6u) put_piece(ring + (8u * t + (lane >> 3)) * DF_RING_DW, lane & 7u, v[t]);
    ld = 6;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  uint32_t norm = 0u, maxk = 0u, maxv = 0u;

  // Refill. A row needs the pieces below byte (cursor + 64) + DF_AHEAD for the chunk AFTER the one that starts
  // at the current cursor: at most 4 new ones. Rows publish [ld, want) in LDS and four neighbouring lanes
  // fetch one row's pieces together (row 16 t + lane / 4, piece ld + lane % 4), so the up to 64 contiguous
  // bytes of a row are one L2 request instead of four -- the kernel is bound by the number of outstanding
  // requests, not by bytes. The data waits in registers and is put into the rings one chunk later.
  const uint4* lbase[4];  // aligned base of the row this lane helps to load in round t
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const uint8_t* b = bodies[9 + t];
    lbase[t] = (const uint4*)(b - ((uintptr_t)b & 15u));
  }
  uint32_t want = 0;
  uint32_t lp[4];  // piece held in nv[t], ~0 = none
  uint4 nv[4];
  auto request = [&]() {
    want = min((((pos >> 3) + 64u + DF_AHEAD + 15u) >> 4), plast + 1u);
    s_ld[lane] = ld;
    s_want[lane] = want;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const uint32_t r = 16u * t + (lane >> 2);
      const uint32_t p = s_ld[r] + (lane & 3u);
      lp[t] = p < s_want[r] ? p : ~0u;
      if (lp[t] != ~0u) nv[t] = lbase[t][p];
    }
  };
  request();

#pragma unroll 1
  for (int ch = 0; ch < N / DF_CH; ch++) {
    // Two coefficients per LDS read: after the alignment shift at least 33 bits of the pair are valid, a
    // code is at most 16 bits long and its decoding looks at 16 bits, so the second code still sees only
    // valid bits. This halves the LDS round trips on the serial cursor chain.
    uint32_t packed[DF_CH / 2];
#pragma unroll
    for (int j = 0; j < DF_CH; j += 2) {
      uint64_t win = window64(pos);
      uint32_t val[2], len[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t top = (uint32_t)(win >> 32);  // code starts at bit 31
        const uint32_t k = (uint32_t)__builtin_clz((top << 8) | 0x00800000u);  // zeros in the 8-bit unary window
        const uint32_t mag = ((top >> 24) & 0x7fu) + (k << 7);
        const uint32_t neg = (uint32_t)((int32_t)top >> 31);
        val[h] = (mag ^ neg) + (neg & (Q + 1u));     // sign ? q - mag : mag (verification.hpp:36-39)
        len[h] = 9u + k;
        norm += mul24(mag, mag);
        maxk = max(maxk, k);
        maxv = max(maxv, val[h]);
        if (h == 0) win <<= len[0];
      }
      pos += len[0] + len[1];
      packed[j >> 1] = val[0] | (val[1] << 16);
    }
    // The pieces requested one chunk ago have arrived behind the arithmetic; by now the cursor has moved
    // on by at least 36 bytes, so the at most 8 live pieces [cursor >> 7, want) fit the ring.
    vmem_wait_all();
#pragma unroll
    for (int t = 0; t < 4; t++)
      if (lp[t] != ~0u) put_piece(ring + (16u * t + (lane >> 2)) * DF_RING_DW, lp[t], nv[t]);
    ld = max(ld, want);
    request();  // for the chunk after next; issued before the stores below so that it waits for nothing
    // The wave's 4 KiB tile (s2_tile_offset: [block][ch][row][32]) goes out as four 1 KiB stores, 16
    // contiguous bytes per lane: lane L holds row L's 64 bytes, so the tile is turned through LDS first
    // (a store of 16 bytes per lane at a 64-byte lane stride is 64 separate L2 requests per instruction:
    // measured in round 2 with every lane storing its own pieces, no LDS stage, 82 VGPRs and 18 waves per CU
    // instead of 168 and 11 -- the kernel took 1.82 ms instead of 0.455 ms per 2^20 Falcon-512 items).
    {
      uint4* mine = (uint4*)(outb + lane * DF_OUT);
#pragma unroll
      for (int t = 0; t < 4; t++) mine[t] = make_uint4(packed[4 * t], packed[4 * t + 1], packed[4 * t + 2], packed[4 * t + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      uint4* tile = (uint4*)(s2 + s2_tile_offset<LOGN>(slot0, (uint32_t)ch * DF_CH));
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const uint32_t r = 16u * t + (lane >> 2);  // row, piece lane & 3
        // Non-temporal: the tile is read once, by the verify kernel, after 1 GB of other tiles has gone by -- no point in
        // keeping it in L2 (decoder -7 %; the same hint on the hash kernel's stores gains nothing, and on the
        // decoder's loads it loses a third: the 64-byte pieces of a row share 128-byte lines across refills).
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        if (slot0 + r < count) __builtin_nontemporal_store(*(const u32x4*)(outb + r * DF_OUT + (lane & 3u) * 4u), (u32x4*)&tile[64u * t + lane]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  if (!live) return;
  if (body[-41] != (uint8_t)(0x30 | LOGN)) {  // decoding.hpp:376-382 (live lane: body is this slot's own row)
    st[slot] = ST_BAD_SIG_HEADER;
    sqn[slot] = 0u;
    return;
  }
  // Trailing bits (decoding.hpp:342-357) are checked from the ring, which holds the row up to piece ld.
  const bool past_end = spec ? pos > end_bits : pos >= end_bits;      // decoding.hpp:340 (the specification allows ==)
  const bool tail_missing = !past_end && end_bits > 128u * ld;        // tail longer than the ring holds: rare
  if (maxk == 8u || tail_missing) {                                   // needs the generic decoder
    st[slot] = ST_BAD_ENCODING;  // overwritten by the fix-up pass
    sqn[slot] = 0u;
    slow_list[atomicAdd(slow_count, 1u)] = slot;
    return;
  }
  bool failed = past_end || maxv == Q;                                // "-0": decoding.hpp:327-330
  if (!failed) {
    uint32_t acc = 0u;
    uint32_t p = pos;
    while (p < end_bits) {
      const uint32_t e = min(32u, end_bits - p);
      acc |= window(p) >> (32u - e);
      p += e;
    }
    failed = acc != 0u;
  }
  st[slot] = failed ? ST_BAD_ENCODING : ST_OK;
  sqn[slot] = norm;
}

template <int LOGN>
__global__ __launch_bounds__(64) void decode_fast_kernel(const Rows sig, const uint32_t* __restrict__ idx, uint32_t count,
                                                         const uint32_t* __restrict__ dcount,
                                                         uint16_t* __restrict__ s2, uint32_t* __restrict__ sqn,
                                                         uint8_t* __restrict__ st, uint32_t* __restrict__ slow_count,
                                                         uint32_t* __restrict__ slow_list, uint32_t spec)
{
  decode_fast_body<LOGN>(blockIdx.x, sig, idx, count, dcount, s2, sqn, st, slow_count, slow_list, spec);
}

// ---- decoding::decode_sig<N> (decoding.hpp:369-391) as a batched operator: the last step ----
// The decoders above leave s2 lifted to [0, q) in the tiled workspace; this kernel (one wave per item) turns a
// chunk of it into what the reference hands back: salt = bytes 1..40 and s2 as signed coefficients, natural
// order. As in the reference, outputs are written only as far as it writes them: a wrong header leaves salt and
// s2 alone (decoding.hpp:380-382), a body that does not decode zeroes s2 and leaves salt alone (:359, :385-387).
template <int LOGN>
__global__ __launch_bounds__(256) void untile_sig_kernel(const uint8_t* __restrict__ sig, const uint16_t* __restrict__ s2t,
                                                         const uint8_t* __restrict__ st, uint32_t count,
                                                         uint8_t* __restrict__ salt, int16_t* __restrict__ s2,
                                                         uint8_t* __restrict__ ok)
{
  constexpr uint32_t N = 1u << LOGN, SIGLEN = LOGN == 9 ? 666 : 1280;
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t slot = blockIdx.x * 4u + (threadIdx.x >> 6); slot < count; slot += gridDim.x * 4u) {
    const uint32_t status = st[slot];
    if (lane == 0) ok[slot] = status == ST_OK ? 1 : 0;
    if (status == ST_BAD_SIG_HEADER) continue;
    const uint16_t* row = s2t + s2_tile_offset<LOGN>(slot, 0u);
    for (uint32_t m = 0; m < N / 64u; m++) {  // i = 64 m + lane
      const uint32_t v = row[(2u * m + (lane >> 5)) * 2048u + (lane & 31u)];
      const int32_t c = v > Q / 2u ? (int32_t)v - QI : (int32_t)v;
      s2[(size_t)slot * N + 64u * m + lane] = status == ST_OK ? (int16_t)c : (int16_t)0;
    }
    if (status == ST_OK && lane < 40u) salt[(size_t)slot * 40u + lane] = sig[(size_t)slot * SIGLEN + 1u + lane];
  }
}

}  // namespace fa
