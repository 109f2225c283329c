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

// sig rows -> s2 (u16, lifted, natural order, row stride n), sqn[slot] = sum s2^2, st[slot].
// idx (nullable) maps slot -> item; outputs are indexed by slot.
template <int LOGN>
__global__ __launch_bounds__(DECODE_WG) void decode_kernel(const uint8_t* __restrict__ sig,
                                                           const uint32_t* __restrict__ idx, uint32_t count,
                                                           uint16_t* __restrict__ s2, uint32_t* __restrict__ sqn,
                                                           uint8_t* __restrict__ st)
{
  constexpr int N = 1 << LOGN;
  constexpr int SIGLEN = LOGN == 9 ? 666 : 1280;
  constexpr int32_t SLEN = 8 * SIGLEN;  // in bits, decoding.hpp:274
  const uint32_t slot = blockIdx.x * DECODE_WG + threadIdx.x;
  if (slot >= count) return;
  const uint32_t item = idx ? idx[slot] : slot;
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

}  // namespace fa
