// host_api_check.cpp -- exercises the C++20 host API of include/falcon_amd.hpp the way the
// reference's own callers use falcon::verify<N> (tests/test_signing.cpp:28-37,
// examples/sign_one.cpp:34): called by tests/test_cpp_host.py on the GPU box.
//
// usage: host_api_check <logn> <count> <mlen> <pk.bin> <msg.bin> <sig.bin> <ndev>
// prints one line of '0'/'1' accept flags from verify_many<N>, a second line from per-item
// verify<N> calls on the first min(count, 8) items, a third from the prepared-key form, and a fourth from
// verify_many_sharded<N>.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <vector>

#include "falcon_amd.hpp"

static std::vector<uint8_t> slurp(const char* path)
{
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", path);
    std::exit(2);
  }
  return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}

template<size_t N>
static int run(size_t count, size_t mlen, std::vector<uint8_t>& pk, std::vector<uint8_t>& msg, std::vector<uint8_t>& sig, int ndev)
{
  if (pk.size() != count * falcon::pkey_len<N>() || sig.size() != count * falcon::sig_len<N>() || msg.size() != count * mlen) {
    std::fprintf(stderr, "file sizes do not match count\n");
    return 2;
  }
  std::vector<uint8_t> acc(count, 7);
  falcon::verify_many<N>(count, pk.data(), msg.data(), mlen, nullptr, sig.data(), acc.data());
  for (uint8_t a : acc) std::putchar('0' + a);
  std::putchar('\n');
  for (size_t i = 0; i < count && i < 8; i++) {
    const bool ok = falcon::verify<N>(pk.data() + i * falcon::pkey_len<N>(), msg.data() + i * mlen, mlen,
                                      sig.data() + i * falcon::sig_len<N>());
    std::putchar(ok ? '1' : '0');
  }
  std::putchar('\n');
  {  // decoded-key form: keys prepared once, item i against key i
    auto keys = falcon::prepared_keys<N>::from_encoded(count, pk.data());
    std::vector<uint8_t> acc3(count, 7);
    falcon::verify_many<N>(count, keys, nullptr, msg.data(), mlen, nullptr, sig.data(), acc3.data());
    for (uint8_t a : acc3) std::putchar('0' + a);
    std::putchar('\n');
  }
  std::vector<uint8_t> acc2(count, 7);
  falcon::pinned_buffer<uint8_t> ppk(pk.size());  // keys from pinned memory, the rest pageable
  std::copy(pk.begin(), pk.end(), ppk.data());
  falcon::verify_many_sharded<N>(count, ppk.data(), msg.data(), mlen, nullptr, sig.data(), acc2.data(), ndev);
  for (uint8_t a : acc2) std::putchar('0' + a);
  std::putchar('\n');
  return 0;
}

int main(int argc, char** argv)
{
  if (argc != 8) {
    std::fprintf(stderr, "usage: %s logn count mlen pk.bin msg.bin sig.bin ndev\n", argv[0]);
    return 2;
  }
  const int logn = std::atoi(argv[1]);
  const size_t count = std::strtoull(argv[2], nullptr, 10), mlen = std::strtoull(argv[3], nullptr, 10);
  auto pk = slurp(argv[4]), msg = slurp(argv[5]), sig = slurp(argv[6]);
  const int ndev = std::atoi(argv[7]);
  try {
    return logn == 9 ? run<512>(count, mlen, pk, msg, sig, ndev) : run<1024>(count, mlen, pk, msg, sig, ndev);
  } catch (const falcon::device_error& e) {
    std::fprintf(stderr, "device_error: %s\n", e.what());
    return 3;
  }
}
