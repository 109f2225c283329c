// encode_kernel.hpp -- the two encoders next to the verify path (wire-format adjacency).
//
// Replaces, batched:
//   encoding::encode_pkey<N>             encoding.hpp:12-35     N coefficients -> 1 + 14 N / 8 bytes (14-bit little-endian fields)
//   encoding::compress_sig<N, sbytelen>  encoding.hpp:169-230   N coefficients -> bytes 41.. of a signature (sign, low 7 bits,
//                                                               unary high part; fails when the code does not end before the last bit)
// They are the inverses of decode_pkey / decompress_sig (tests/test_encoding.cpp:14-48, 104-237) and are what a signer or
// a corpus generator runs; they are not on the verify hot path, so they are written for coalesced memory traffic rather
// than tuned to an issue limit: records are assembled in LDS and leave as aligned dword stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "verify_kernel.hpp"

namespace fa {

// Writes the `len` bytes of a record to global memory at `dst` from an LDS image in which record byte i sits at image byte
// a + i, a = dst & 3 -- so aligned dwords of the image are aligned dwords of memory and the record leaves as whole-dword
// stores except for its first and last partial dword. zero: the record is all zeros. One wave, lanes take dwords in turn.
__device__ __forceinline__ void store_record(const uint32_t* __restrict__ img, uint8_t* __restrict__ dst, uint32_t len, bool zero,
                                             uint32_t lane)
{
  const uint32_t a = (uint32_t)((uintptr_t)dst & 3u);
  uint8_t* base = dst - a;
  const uint32_t end = a + len, ndw = (end + 3u) >> 2;
  for (uint32_t k = lane; k < ndw; k += 64u) {
    const uint32_t w = zero ? 0u : img[k];
    const uint32_t lo = 4u * k;
    if (lo >= a && lo + 4u <= end) {
      *(uint32_t*)(base + lo) = w;
    } else {
#pragma unroll
      for (uint32_t b = 0; b < 4u; b++)
        if (lo + b >= a && lo + b < end) base[lo + b] = (uint8_t)(w >> (8u * b));
    }
  }
}

// One wave per key. h: [count][N] u16, taken as they are (encoding.hpp:25-33 uses h[i].v unmasked: a value >= 2^14 spills
// into the neighbouring field exactly as the reference's casts and ORs make it). pk: [count][PKLEN].
template <int LOGN>
__global__ __launch_bounds__(VERIFY_WG) void encode_pkeys_kernel(const uint16_t* __restrict__ h, uint8_t* __restrict__ pk, uint32_t count)
{
  using G = Geo<LOGN>;
  constexpr uint32_t N = G::N, PKLEN = G::PKLEN, GROUPS = N / 4u, IMG_DW = (PKLEN + 3u + 3u) / 4u;
  __shared__ uint32_t s_img[VERIFY_WAVES][IMG_DW];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint8_t* img = (uint8_t*)s_img[wave];
  for (uint32_t key = blockIdx.x * VERIFY_WAVES + wave; key < count; key += gridDim.x * VERIFY_WAVES) {
    const uint16_t* row = h + (size_t)key * N;
    uint8_t* dst = pk + (size_t)key * PKLEN;
    const uint32_t a = (uint32_t)((uintptr_t)dst & 3u);
    if (lane == 0) img[a] = (uint8_t)LOGN;  // encoding.hpp:17,22
    for (uint32_t g = lane; g < GROUPS; g += 64u) {
      const uint2 v = *(const uint2*)(row + 4u * g);  // rows are 2 N bytes: 8-byte aligned when h is
      const uint32_t h0 = v.x & 0xffffu, h1 = v.x >> 16, h2 = v.y & 0xffffu, h3 = v.y >> 16;
      uint8_t* o = img + a + 1u + 7u * g;
      o[0] = (uint8_t)h0;                                                  // encoding.hpp:25-33, expression by expression
      o[1] = (uint8_t)(((h1 & 0x03u) << 6) | ((h0 >> 8) & 0xffu));
      o[2] = (uint8_t)(h1 >> 2);
      o[3] = (uint8_t)(((h2 & 0x0fu) << 4) | ((h1 >> 10) & 0xffu));
      o[4] = (uint8_t)(h2 >> 4);
      o[5] = (uint8_t)(((h3 & 0x3fu) << 2) | ((h2 >> 12) & 0xffu));
      o[6] = (uint8_t)(h3 >> 6);
    }
    wave_lds_sync();
    store_record(s_img[wave], dst, PKLEN, false, lane);
    wave_lds_sync();
  }
}

// One signature per lane, 64 per workgroup (= wave). s2: [count][N] int16 (16-byte aligned rows); sig: [count][SIGLEN], of
// which only bytes 41.. are written (encoding.hpp:164-165, 227); ok[i] = the reference's bool. Each lane writes its
// row's code as big-endian dwords into its own LDS image, which starts with `a` filler bytes (a = alignment of the body
// in memory) so that the wave can afterwards copy every row out with aligned dword stores (store_record).
// A code longer than the body fails (the reference zeroes the body then, encoding.hpp:226); bits beyond the image are
// counted, not stored. Codes longer than the reference's own 14 N / 8-byte buffer are undefined behaviour there and
// simply fail here.
template <int LOGN>
__global__ __launch_bounds__(64) void compress_sigs_kernel(const int16_t* __restrict__ s2, uint32_t count, uint8_t* __restrict__ sig,
                                                           uint8_t* __restrict__ ok)
{
  constexpr uint32_t N = 1u << LOGN, SIGLEN = LOGN == 9 ? 666u : 1280u, BODY = SIGLEN - 41u;
  constexpr uint32_t SLEN = 8u * SIGLEN - 328u;                  // encoding.hpp:176
  constexpr uint32_t ROW_DW = ((BODY + 3u + 3u) / 4u) | 1u;      // image dwords per row, odd: lanes fall on different banks
  __shared__ uint32_t s_img[64 * ROW_DW];
  __shared__ uint8_t s_fail[64];
  const uint32_t lane = threadIdx.x;
  const uint32_t slot0 = blockIdx.x * 64u, slot = slot0 + lane;
  const bool live = slot < count;
  if (live) {
    uint32_t* img = s_img + lane * ROW_DW;
    const uint32_t a = (uint32_t)((uintptr_t)(sig + (size_t)slot * SIGLEN + 41u) & 3u);
    const uint32_t img_dw = (a + BODY + 3u) >> 2;
    uint64_t acc = 0;          // MSB-first bit accumulator
    uint32_t nacc = 8u * a;    // bits in it: the filler bytes come first
    uint32_t wpos = 0;         // next image dword
    uint32_t bits = 0;         // bit_idx of the reference
    auto append = [&](uint32_t value, uint32_t len) {  // len <= 32, value < 2^len
      acc |= (uint64_t)value << (64u - nacc - len);
      nacc += len;
      if (nacc >= 32u) {
        if (wpos < img_dw) img[wpos] = __builtin_bswap32((uint32_t)(acc >> 32));
        wpos++;
        acc <<= 32;
        nacc -= 32u;
      }
    };
    const uint4* row = (const uint4*)(s2 + (size_t)slot * N);
    for (uint32_t g = 0; g < N / 8u; g++) {
      const uint4 v = row[g];
      const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int32_t c = (int16_t)(w[j >> 1] >> (16 * (j & 1)));
        const uint32_t mag = (uint32_t)(c < 0 ? -c : c);
        const uint32_t head = ((c < 0 ? 1u : 0u) << 7) | (mag & 0x7fu);   // encoding.hpp:186-206
        uint32_t k = mag >> 7;                                             // :210
        bits += 9u + k;
        if (k <= 23u) {
          append((head << (k + 1u)) | 1u, 9u + k);                         // head, k zeros, a one (:211-220)
        } else {
          append(head, 8u);
          for (; k >= 32u; k -= 32u) append(0u, 32u);
          append(1u, k + 1u);
        }
      }
    }
    // the rest of the body is zero (the reference's buffer is zero-initialised, encoding.hpp:179)
    append(0u, 32u - nacc);          // completes the dword in progress (a no-op dword of zeros when nacc == 0)
    while (wpos < img_dw) append(0u, 32u);
    s_fail[lane] = bits >= SLEN;     // encoding.hpp:226-229
    ok[slot] = bits >= SLEN ? 0 : 1;
  }
  wave_lds_sync();
  const uint32_t rows = min(64u, count - slot0);
  for (uint32_t r = 0; r < rows; r++)
    store_record(s_img + r * ROW_DW, sig + (size_t)(slot0 + r) * SIGLEN + 41u, BODY, s_fail[r] != 0, lane);
}

}  // namespace fa
