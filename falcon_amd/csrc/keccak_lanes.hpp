// keccak_lanes.hpp -- where the words of a Keccak-f[1600] state live when ONE state is spread over the 64 lanes of a
// wave, and which lanes each lane reads in a round (tiny_kernel.hpp, hash_wave_body). Plain C++17, no HIP: the table is
// built at compile time for the kernel and is also checked on the CPU against the textbook permutation
// (tests/cpp/keccak_lanes_check.cpp, tests/test_keccak_lanes.py).
#pragma once
#include <stdint.h>

namespace fa {

struct KeccakLane {
  uint8_t cm, cp, cpp;        // lanes holding C[x-1] (own half), C[x+1] (own half), C[x+1] (other half)
  uint8_t s0[3], s1[3], sh[3];  // per chi operand: lanes of the two v_alignbit sources, shift
  uint8_t word;               // x + 5 y, 0xff for a lane that holds nothing
  uint8_t pad[3];
};

constexpr int keccak_lane_of(int x, int y, int h) { return 32 * h + 16 * (x / 3) + 5 * (x % 3) + y; }

struct KeccakLaneTable {
  KeccakLane lane[64];
};

constexpr KeccakLaneTable make_keccak_lanes()
{
  constexpr int RHO[5][5] = { { 0, 36, 3, 41, 18 }, { 1, 44, 10, 45, 2 }, { 62, 6, 43, 15, 61 }, { 28, 55, 25, 21, 56 },
                              { 27, 20, 39, 8, 14 } };  // [x][y]
  KeccakLaneTable t{};
  for (int l = 0; l < 64; l++) {
    KeccakLane& k = t.lane[l];
    k.cm = k.cp = k.cpp = (uint8_t)l;
    for (int j = 0; j < 3; j++) {
      k.s0[j] = k.s1[j] = (uint8_t)l;
      k.sh[j] = 0;
    }
    k.word = 0xff;
    k.pad[0] = k.pad[1] = k.pad[2] = 0;
  }
  for (int h = 0; h < 2; h++)
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) {
        KeccakLane& k = t.lane[keccak_lane_of(x, y, h)];
        k.word = (uint8_t)(x + 5 * y);
        k.cm = (uint8_t)keccak_lane_of((x + 4) % 5, 4, h);
        k.cp = (uint8_t)keccak_lane_of((x + 1) % 5, 4, h);
        k.cpp = (uint8_t)keccak_lane_of((x + 1) % 5, 4, 1 - h);
        for (int j = 0; j < 3; j++) {
          const int X = (x + j) % 5, Y = y;                // the chi operand B[X][Y] ...
          const int sx = (X + 3 * Y) % 5, sy = X;          // ... is the rotated A[sx][sy] (pi: (sx, sy) -> (sy, 2 sx + 3 sy))
          const int r = RHO[sx][sy];
          const int own = keccak_lane_of(sx, sy, h), other = keccak_lane_of(sx, sy, 1 - h);
          if (r == 0) {  // alignbit(., own, 0) == own
            k.s0[j] = (uint8_t)own, k.s1[j] = (uint8_t)own, k.sh[j] = 0;
          } else if (r < 32) {
            k.s0[j] = (uint8_t)own, k.s1[j] = (uint8_t)other, k.sh[j] = (uint8_t)(32 - r);
          } else {
            k.s0[j] = (uint8_t)other, k.s1[j] = (uint8_t)own, k.sh[j] = (uint8_t)(64 - r);
          }
        }
      }
  return t;
}

}  // namespace fa
