// keccak_lanes_check.cpp -- CPU check of the lane table behind hash_wave_body (falcon_amd/csrc/tiny_kernel.hpp): the round
// as the GPU runs it on one state spread over 64 lanes -- column parities by row-wise shifted XORs inside 16-lane rows,
// every other word movement a fetch from a lane named by falcon_amd/csrc/keccak_lanes.hpp, rotations as 32-bit funnel
// shifts -- emulated lane by lane and compared with the textbook Keccak-f[1600] on 25 64-bit words (FIPS 202, 3.2-3.3; the
// permutation under hashing::hash_to_point, hashing.hpp:30-39). Prints "ok" and exits 0 when 24 rounds agree on random states.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../falcon_amd/csrc/keccak_lanes.hpp"

static const uint64_t RC[24] = {
  0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
  0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
  0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
  0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull };

static uint64_t rotl(uint64_t v, int r) { return r ? (v << r) | (v >> (64 - r)) : v; }

static void textbook_round(uint64_t a[25], uint64_t rc)
{
  static const int RHO[25] = { 0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14 };  // [x + 5 y]
  uint64_t c[5], b[25];
  for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
  for (int i = 0; i < 25; i++) a[i] ^= c[(i % 5 + 4) % 5] ^ rotl(c[(i % 5 + 1) % 5], 1);
  for (int x = 0; x < 5; x++)
    for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], RHO[x + 5 * y]);
  for (int y = 0; y < 5; y++)
    for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
  a[0] ^= rc;
}

static uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (s & 31u)); }

// v_xor_b32_dpp d, a, b row_shr:n bound_ctrl:1 -- lane l gets b ^ (a of lane l - n of its 16-lane row, 0 outside the row)
static void xor_row_shr(const uint32_t a[64], const uint32_t b[64], int n, uint32_t d[64])
{
  for (int l = 0; l < 64; l++) d[l] = b[l] ^ ((l & 15) >= n ? a[l - n] : 0u);
}

static void wave_round(uint32_t a[64], const fa::KeccakLaneTable& T, uint64_t rc)
{
  uint32_t u[64], v[64], s[64], t[64];
  xor_row_shr(a, a, 1, u);
  xor_row_shr(u, u, 2, v);
  xor_row_shr(a, v, 4, s);
  for (int l = 0; l < 64; l++) {
    const fa::KeccakLane& k = T.lane[l];
    t[l] = a[l] ^ s[k.cm] ^ alignbit(s[k.cp], s[k.cpp], 31);
  }
  for (int l = 0; l < 64; l++) {
    const fa::KeccakLane& k = T.lane[l];
    uint32_t b[3];
    for (int j = 0; j < 3; j++) b[j] = alignbit(t[k.s0[j]], t[k.s1[j]], k.sh[j]);
    const uint32_t iota = l == 0 ? (uint32_t)rc : l == 32 ? (uint32_t)(rc >> 32) : 0u;
    a[l] = (b[0] ^ (~b[1] & b[2])) ^ iota;
  }
}

int main()
{
  constexpr fa::KeccakLaneTable T = fa::make_keccak_lanes();
  // every word of the state has exactly one lane per half, inside the wave
  int seen[25][2] = {};
  for (int l = 0; l < 64; l++)
    if (T.lane[l].word != 0xff) seen[T.lane[l].word][l >> 5]++;
  for (int w = 0; w < 25; w++)
    if (seen[w][0] != 1 || seen[w][1] != 1) return std::printf("word %d is not held exactly once per half\n", w), 1;
  uint64_t x = 0x9E3779B97F4A7C15ull;
  for (int trial = 0; trial < 200; trial++) {
    uint64_t st[25];
    uint32_t lanes[64];
    for (int i = 0; i < 25; i++) {
      x ^= x << 13, x ^= x >> 7, x ^= x << 17;
      st[i] = trial == 0 ? 0ull : x;
    }
    for (int l = 0; l < 64; l++) {
      x ^= x << 13, x ^= x >> 7, x ^= x << 17;
      const uint8_t w = T.lane[l].word;
      lanes[l] = w == 0xff ? (uint32_t)x /* lanes that hold nothing carry garbage: it must not leak */ : (uint32_t)(st[w] >> (32 * (l >> 5)));
    }
    for (int r = 0; r < 24; r++) {
      textbook_round(st, RC[r]);
      wave_round(lanes, T, RC[r]);
      for (int l = 0; l < 64; l++) {
        const uint8_t w = T.lane[l].word;
        if (w != 0xff && lanes[l] != (uint32_t)(st[w] >> (32 * (l >> 5))))
          return std::printf("trial %d round %d lane %d (word %d half %d) differs\n", trial, r, l, w, l >> 5), 1;
      }
    }
  }
  std::puts("ok");
  return 0;
}
