// both_headers_check.cpp -- one translation unit that uses a CPU library with the reference's names AND this library.
//
// falcon_gpu.hpp keeps the GPU API in namespace falcon_gpu, so it can sit beside the reference's falcon.hpp, which
// defines falcon::verify<N> (falcon.hpp:230-249). The reference's falcon.hpp cannot be compiled in this image (it needs
// the absent sha3 submodule and GMP), so the CPU side here is a declaration with the reference's exact signature in the
// reference's namespace -- enough to prove that the two do not collide -- plus, where /root/reference exists at build
// time, the reference's own headers that do compile (decoding.hpp, ntt.hpp, utils.hpp), included for real.
// Run by tests/test_cpp_host.py: verifies the README triple through falcon_gpu::verify<512> and prints the result.
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>

#if __has_include("/root/reference/include/decoding.hpp")
#include "/root/reference/include/decoding.hpp"
#include "/root/reference/include/ntt.hpp"
#include "/root/reference/include/utils.hpp"
#define HAVE_REFERENCE_HEADERS 1
#endif

namespace falcon {  // the reference's entry point, falcon.hpp:230-236, declared only
template<const size_t N>
static inline bool verify(const uint8_t* __restrict pkey, const uint8_t* __restrict msg, const size_t mlen, uint8_t* __restrict sig)
  requires((N == 512) || (N == 1024));
}

#include "falcon_gpu.hpp"

static std::vector<uint8_t> slurp(const char* path)
{
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}

int main(int argc, char** argv)
{
  if (argc != 4) return 2;
  auto pk = slurp(argv[1]), msg = slurp(argv[2]), sig = slurp(argv[3]);
  if (pk.size() != falcon_gpu::pkey_len<512>() || sig.size() != falcon_gpu::sig_len<512>()) return 2;
  try {
    const bool ok = falcon_gpu::verify<512>(pk.data(), msg.data(), msg.size(), sig.data());
    bool same = true;
#ifdef HAVE_REFERENCE_HEADERS
    // the reference's decode_pkey (CPU) and this library's decode_pkeys (GPU) side by side in one program
    ff::ff_t h_ref[512];
    const bool ok_ref = decoding::decode_pkey<512>(pk.data(), h_ref);
    std::vector<uint16_t> h_gpu(512);
    uint8_t ok_gpu = 0;
    falcon_gpu::decode_pkeys<512>(1, pk.data(), h_gpu.data(), &ok_gpu);
    same = ok_ref == (ok_gpu != 0);
    for (size_t i = 0; i < 512; i++) same = same && h_ref[i].v == h_gpu[i];
    std::printf("reference-headers %d\n", 1);
#else
    std::printf("reference-headers %d\n", 0);
#endif
    std::printf("verified %d decode_pkey_agrees %d\n", ok ? 1 : 0, same ? 1 : 0);
    return ok && same ? 0 : 1;
  } catch (const falcon_gpu::device_error& e) {
    std::fprintf(stderr, "device_error: %s\n", e.what());
    return 3;
  }
}
