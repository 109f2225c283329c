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
// Output per item: s2 lifted to [0, q) as u16 in natural order (the lift of
// verification.hpp:36-39), the partial squared norm sum s2[i]^2, and a status byte.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zq.hpp"

namespace fa {

enum : uint8_t { ST_OK = 0, ST_BAD_PK_HEADER = 1, ST_BAD_SIG_HEADER = 2, ST_BAD_ENCODING = 3, ST_NORM_TOO_LARGE = 4,
                 ST_HASH_UNDERFLOW = 5 /* internal: needs more squeeze blocks */ };

constexpr int DECODE_WG = 256;

// MSB-first bit reader over the signature body (bytes 41..siglen-1), fed by aligned dwords.
struct BitReader {
  const uint32_t* base;  // aligned dword holding body byte 0
  uint32_t al;           // body byte 0 sits at byte `al` of base[0]
  int32_t body_bytes;    // siglen - 41
  int32_t next;          // next dword index to fetch
  uint64_t buf;          // MSB-aligned bit buffer
  int32_t nbits;         // valid bits in buf

  __device__ __forceinline__ uint32_t fetch()
  {
    // dword `next` covers body bytes [4*next - al, 4*next - al + 4); bytes outside the body -> 0
    const int32_t first = 4 * next - (int32_t)al;
    uint32_t w = 0u;
    if (first < body_bytes) {
      w = __builtin_bswap32(base[next]);  // byte k of the dword -> bits 31-8k .. 24-8k
      const int32_t valid = body_bytes - first;  // > 0
      if (valid < 4) w &= ~0u << (8 * (4 - valid));
      if (first < 0) w &= ~0u >> (8 * (-first));  // bytes before the body (only for next == 0)
    }
    next++;
    return w;
  }

  __device__ __forceinline__ void init(const uint8_t* body, int32_t nbytes)
  {
    al = (uint32_t)((uintptr_t)body & 3u);
    base = (const uint32_t*)(body - al);
    body_bytes = nbytes;
    next = 0;
    const uint32_t w0 = fetch();
    buf = (uint64_t)w0 << (32 + 8 * al);
    nbits = 32 - 8 * (int32_t)al;
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

// Generic decoder (every rule, including the k >= 8 windows). sig rows -> s2 (u16, lifted, natural
// order, row stride n), sqn[o] = sum s2^2, st[o]. idx (nullable) maps slot -> item; outputs go to
// index o = slot, or o = item when `scatter` (the fix-up pass behind decode_fast_kernel). The number
// of slots is min(count, *dcount) when dcount is given (a device-side list length).
template <int LOGN>
__global__ __launch_bounds__(DECODE_WG) void decode_kernel(const uint8_t* __restrict__ sig,
                                                           const uint32_t* __restrict__ idx, uint32_t count,
                                                           const uint32_t* __restrict__ dcount, bool scatter,
                                                           uint16_t* __restrict__ s2, uint32_t* __restrict__ sqn,
                                                           uint8_t* __restrict__ st)
{
  constexpr int N = 1 << LOGN;
  constexpr int SIGLEN = LOGN == 9 ? 666 : 1280;
  constexpr int32_t SLEN = 8 * SIGLEN;  // in bits, decoding.hpp:274
  uint32_t slot = blockIdx.x * DECODE_WG + threadIdx.x;
  if (dcount) count = min(count, *dcount);
  if (slot >= count) return;
  const uint32_t item = idx ? idx[slot] : slot;
  if (scatter) slot = item;
  const uint8_t* row = sig + (size_t)item * SIGLEN;
  uint16_t* out = s2 + (size_t)slot * N;

  if (row[0] != (uint8_t)(0x30 | LOGN)) {  // decoding.hpp:376-382
    st[slot] = ST_BAD_SIG_HEADER;
    sqn[slot] = 0u;
    return;
  }

  BitReader br;
  br.init(row + 41, SIGLEN - 41);
  int32_t pos = 8 + 320;  // absolute bit index, decoding.hpp:276-277
  int32_t i = 0;
  bool failed = false;
  uint32_t norm = 0u;

  while (i < N && pos < SLEN) {  // decoding.hpp:281
    br.refill();
    const uint32_t top = br.peek32();
    const uint32_t sign = top >> 31;
    uint32_t mag = (top >> 24) & 0x7fu;
    const uint32_t k = min((uint32_t)__builtin_clz((top << 8) | 0x00800000u), 8u);  // zeros in the 8-bit window
    if (k < 8) {  // decoding.hpp:297-299
      mag += k << 7;
      br.skip(9 + (int32_t)k);
      pos += 9 + (int32_t)k;
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
      break;
    }
    out[i] = (uint16_t)(sign ? Q - mag : mag);
    norm += mag * mag;
    i++;
  }

  failed |= (pos >= SLEN) | (i < N);  // decoding.hpp:340
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

// ------------------------------------------------------------------------------------------------
// Fast path: one signature per lane, but all memory traffic is wave-cooperative.
//   * input: each lane owns a 128-byte ring in LDS that the whole wave refills with aligned 16-byte
//     global loads (8 lanes per row), byte-swapped on the way in so a lane pulls MSB-first dwords;
//   * output: 32 coefficients per lane are staged in LDS and written out as 64-byte row segments.
// Only the common rule (k < 8 zeros in the unary part, 9..16 bits per coefficient) is handled here:
// a lane that meets k == 8 stops and queues its item for decode_kernel (scatter mode), which
// implements the window-skipping rule. Everything else (header, "-0", end-of-stream and trailing-zero
// rules) is decided here exactly as in decode_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int DF_CH = 32;       // coefficients per chunk
constexpr int DF_RING = 33;     // dwords per lane: 32 of ring + 1 of padding (bank-conflict free)
constexpr int DF_OUT = 17;      // dwords per lane: 32 u16 + padding

template <int LOGN>
__global__ __launch_bounds__(64) void decode_fast_kernel(const uint8_t* __restrict__ sig, uint32_t count,
                                                         uint16_t* __restrict__ s2, uint32_t* __restrict__ sqn,
                                                         uint8_t* __restrict__ st, uint32_t* __restrict__ slow_count,
                                                         uint32_t* __restrict__ slow_list)
{
  constexpr int N = 1 << LOGN;
  constexpr uint32_t SIGLEN = LOGN == 9 ? 666 : 1280;
  constexpr uint32_t BODY = SIGLEN - 41;
  __shared__ uint32_t ring[64 * DF_RING];
  __shared__ uint32_t outb[64 * DF_OUT];
  __shared__ uint32_t s_wq[64], s_tq[64];  // per row: next piece to load, piece to load up to

  const uint32_t lane = threadIdx.x;
  const uint32_t slot0 = blockIdx.x * 64u;
  const uint32_t slot = slot0 + lane;
  const bool live = slot < count;

  // geometry of this lane's row, relative to the 16-byte aligned address `abase` below the body
  const uintptr_t body = (uintptr_t)(sig + (size_t)(live ? slot : 0u) * SIGLEN + 41);
  const uint32_t boff = (uint32_t)(body & 15u);
  const uint32_t end_bits = 8u * (boff + BODY);  // first bit past the signature

  // the same for the row this lane helps to load: row (lane>>3) + 8t, piece lane&7
  // Row geometry of the 8 rows this lane helps to load is loop-invariant: keep it in registers.
  uintptr_t rbase[8];
  uint32_t rlast[8];  // last piece that holds a byte of the row
#pragma unroll
  for (int t = 0; t < 8; t++) {
    const uint32_t rs = min(slot0 + (lane >> 3) + 8u * t, count - 1u);
    const uintptr_t rbody = (uintptr_t)(sig + (size_t)rs * SIGLEN + 41);
    rbase[t] = rbody & ~(uintptr_t)15;
    rlast[t] = ((uint32_t)(rbody & 15u) + BODY - 1u) >> 4;
  }
  auto load_pieces = [&]() {
    uint4 v[8];
    uint32_t q[8];
    bool want[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {  // all eight loads in flight before the first LDS write
      const uint32_t r = (lane >> 3) + 8u * t;
      q[t] = s_wq[r] + (lane & 7u);
      want[t] = q[t] < s_tq[r] && q[t] <= rlast[t];
      v[t] = *(const uint4*)(rbase[t] + 16u * (uintptr_t)min(q[t], rlast[t]));
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
      if (want[t]) {
        uint32_t* dst = ring + ((lane >> 3) + 8u * t) * DF_RING + (q[t] & 7u) * 4u;
        dst[0] = __builtin_bswap32(v[t].x);
        dst[1] = __builtin_bswap32(v[t].y);
        dst[2] = __builtin_bswap32(v[t].z);
        dst[3] = __builtin_bswap32(v[t].w);
      }
    }
  };
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  uint32_t rd = boff & ~3u;  // byte offset (from abase) of the next dword to pull
  s_wq[lane] = 0u;
  s_tq[lane] = 8u;
  wave_sync();
  load_pieces();
  s_wq[lane] = 8u;
  wave_sync();

  const uint32_t* myring = ring + lane * DF_RING;
  auto pull = [&]() -> uint32_t {
    const uint32_t w = myring[(rd >> 2) & 31u];
    rd += 4u;
    return w;
  };
  uint64_t buf = (uint64_t)pull() << (32u + 8u * (boff & 3u));
  int32_t nbits = 32 - 8 * (int32_t)(boff & 3u);
  uint32_t pos = 8u * boff;  // bit position relative to abase; body bit 0 == absolute bit 328
  int32_t i = 0;
  uint32_t norm = 0u;
  bool failed = false, quirk = false;
  const bool hdr_ok = live && sig[(size_t)slot * SIGLEN] == (uint8_t)(0x30 | LOGN);
  bool alive = hdr_ok;
  uint16_t* orow = s2 + (size_t)slot0 * N;

  for (int ch = 0; ch < N / DF_CH; ch++) {
    uint32_t prev = 0u;
#pragma unroll 2
    for (int j = 0; j < DF_CH; j++) {
      if (alive && nbits <= 32) {
        buf |= (uint64_t)pull() << (32 - nbits);
        nbits += 32;
      }
      alive = alive && pos < end_bits;  // decoding.hpp:281
      const uint32_t top = (uint32_t)(buf >> 32);
      const uint32_t sign = top >> 31;
      const uint32_t k = (uint32_t)__builtin_clz((top << 8) | 0x00800000u);
      const uint32_t mag = ((top >> 24) & 0x7fu) + (k << 7);
      if (alive) {
        if (k == 8u) {
          quirk = true;
          alive = false;
        } else if (sign && mag == 0u) {  // decoding.hpp:327-330
          failed = true;
          alive = false;
        } else {
          buf <<= (9u + k);
          nbits -= (int32_t)(9u + k);
          pos += 9u + k;
          norm += mag * mag;
          i++;
        }
      }
      const uint32_t val = sign ? Q - mag : mag;
      if (j & 1) outb[lane * DF_OUT + (j >> 1)] = prev | (val << 16);
      prev = val;
    }
    wave_sync();
    // flush 64 rows x 64 bytes; refill every ring up to 8 pieces beyond the reader
    if (slot0 + 64u <= count) {  // wave-uniform: full tile, no per-row guard
#pragma unroll
      for (int t = 0; t < 16; t++) {
        const uint32_t e = t * 64u + lane, r = e >> 4, dw = e & 15u;
        ((uint32_t*)(orow + (size_t)r * N + ch * DF_CH))[dw] = outb[r * DF_OUT + dw];
      }
    } else {
      for (int t = 0; t < 16; t++) {
        const uint32_t e = t * 64u + lane, r = e >> 4, dw = e & 15u;
        if (slot0 + r < count) ((uint32_t*)(orow + (size_t)r * N + ch * DF_CH))[dw] = outb[r * DF_OUT + dw];
      }
    }
    s_tq[lane] = (rd >> 4) + 8u;
    wave_sync();
    load_pieces();
    s_wq[lane] = s_tq[lane];
    wave_sync();
  }

  if (!live) return;
  if (!hdr_ok) {  // decoding.hpp:376-382
    st[slot] = ST_BAD_SIG_HEADER;
    sqn[slot] = 0u;
    return;
  }
  if (quirk) {
    st[slot] = ST_BAD_ENCODING;  // overwritten by the fix-up pass
    sqn[slot] = 0u;
    slow_list[atomicAdd(slow_count, 1u)] = slot;
    return;
  }
  failed |= (pos >= end_bits) | (i < N);  // decoding.hpp:340
  if (!failed) {  // decoding.hpp:342-357
    while (pos < end_bits) {
      if (nbits <= 32) {
        buf |= (uint64_t)pull() << (32 - nbits);
        nbits += 32;
      }
      const uint32_t e = min(32u, end_bits - pos);
      failed |= ((uint32_t)(buf >> 32) >> (32u - e)) != 0u;
      buf <<= e;
      nbits -= (int32_t)e;
      pos += e;
    }
  }
  st[slot] = failed ? ST_BAD_ENCODING : ST_OK;
  sqn[slot] = norm;
}

}  // namespace fa
