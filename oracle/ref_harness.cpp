// ref_harness.cpp -- C entry points over the REFERENCE's own headers, for pinning the oracle.
//
// TEST INFRASTRUCTURE ONLY. Built by `make -C oracle ref` into oracle/_ref/libfalcon_ref.so, in the
// container where /root/reference exists; nothing of the reference is copied into this repo: the
// headers are #included where they lie (-I/root/reference/include). The product never loads this.
//
// What can be built: every header of the verify path EXCEPT hashing.hpp / verification.hpp, which need
// sha3/include/shake256.hpp (an un-vendored submodule, absent here). So this library exposes the
// reference's decode_pkey, decode_sig / decompress_sig, ntt / intt, the Z_q field and polynomial
// operations and the two encoders as they are, and -- in fr_verify_after_hash -- the steps of
// verification::verify after the hash (verification.hpp:36-79) carried out with the reference's own
// functions on a `c` computed elsewhere (the oracle's hash_to_point is pinned by hashlib.shake_256).
#include <cstdint>
#include <cstring>

#include "decoding.hpp"
#include "encoding.hpp"
#include "ff.hpp"
#include "ntt.hpp"
#include "polynomial.hpp"

static_assert(sizeof(ff::ff_t) == sizeof(uint16_t), "ff_t is one uint16_t");

namespace {

template<size_t N>
int decode_pkey_n(const uint8_t* pk, uint16_t* h)
{
  return decoding::decode_pkey<N>(pk, reinterpret_cast<ff::ff_t*>(h)) ? 1 : 0;
}

// decompress_sig reads up to two bytes past the signature (decoding.hpp:217-218); give it a padded copy.
template<size_t N, size_t SLEN>
int decompress_sig_n(const uint8_t* sig, int32_t* s2)
{
  uint8_t buf[SLEN + 8] = {};
  std::memcpy(buf, sig, SLEN);
  return decoding::decompress_sig<N, SLEN>(buf, s2) ? 1 : 0;
}

template<size_t N, size_t SLEN>
int decode_sig_n(const uint8_t* sig, uint8_t* salt, int32_t* s2)
{
  uint8_t buf[SLEN + 8] = {};
  std::memcpy(buf, sig, SLEN);
  return decoding::decode_sig<N>(buf, salt, s2) ? 1 : 0;
}

// verification.hpp:36-79 with the reference's functions; c = hash_to_point output (canonical). The norm is
// summed in 64 bits here and returned, so the caller derives both the exact and the int32-wrapped flag.
template<size_t N>
uint64_t verify_after_hash_n(const uint16_t* h_in, const uint16_t* c_in, const int32_t* s2, int16_t* s1_out)
{
  constexpr size_t LOG2N = N == 512 ? 9 : 10;
  ff::ff_t c[N], h[N], s2_ntt[N], s1[N];
  for (size_t i = 0; i < N; i++) {
    c[i].v = c_in[i];
    h[i].v = h_in[i];                                                           // raw, as decode_pkey leaves them
    s2_ntt[i] = ff::ff_t{ static_cast<uint16_t>(s2[i] < 0 ? s2[i] + ff::Q : s2[i]) };  // :36-39
  }
  ntt::ntt<LOG2N>(c);       // :47-49
  ntt::ntt<LOG2N>(s2_ntt);
  ntt::ntt<LOG2N>(h);
  polynomial::mul<LOG2N>(s2_ntt, h, s1);  // :53-55
  polynomial::neg<LOG2N>(s1);
  polynomial::add_to<LOG2N>(s1, c);
  ntt::intt<LOG2N>(s1);     // :57
  uint64_t norm = 0;
  for (size_t i = 0; i < N; i++) {
    const int32_t v = s1[i].v >= (ff::Q >> 1) ? static_cast<int32_t>(s1[i].v) - static_cast<int32_t>(ff::Q)
                                              : static_cast<int32_t>(s1[i].v);  // :59-68
    s1_out[i] = static_cast<int16_t>(v);
    norm += static_cast<uint64_t>(static_cast<int64_t>(v) * v) + static_cast<uint64_t>(static_cast<int64_t>(s2[i]) * s2[i]);
  }
  return norm;
}

}  // namespace

extern "C" {

int fr_decode_pkey(unsigned logn, const uint8_t* pk, uint16_t* h)
{
  return logn == 9 ? decode_pkey_n<512>(pk, h) : decode_pkey_n<1024>(pk, h);
}
int fr_decompress_sig(unsigned logn, const uint8_t* sig, int32_t* s2)
{
  return logn == 9 ? decompress_sig_n<512, 666>(sig, s2) : decompress_sig_n<1024, 1280>(sig, s2);
}
int fr_decode_sig(unsigned logn, const uint8_t* sig, uint8_t* salt, int32_t* s2)
{
  return logn == 9 ? decode_sig_n<512, 666>(sig, salt, s2) : decode_sig_n<1024, 1280>(sig, salt, s2);
}
void fr_encode_pkey(unsigned logn, const uint16_t* h, uint8_t* pk)
{
  if (logn == 9) encoding::encode_pkey<512>(reinterpret_cast<const ff::ff_t*>(h), pk);
  else encoding::encode_pkey<1024>(reinterpret_cast<const ff::ff_t*>(h), pk);
}
int fr_compress_sig(unsigned logn, const int32_t* s2, uint8_t* sig)
{
  return (logn == 9 ? encoding::compress_sig<512, 666>(s2, sig) : encoding::compress_sig<1024, 1280>(s2, sig)) ? 1 : 0;
}
void fr_ntt(unsigned logn, uint16_t* poly)
{
  if (logn == 9) ntt::ntt<9>(reinterpret_cast<ff::ff_t*>(poly));
  else ntt::ntt<10>(reinterpret_cast<ff::ff_t*>(poly));
}
void fr_intt(unsigned logn, uint16_t* poly)
{
  if (logn == 9) ntt::intt<9>(reinterpret_cast<ff::ff_t*>(poly));
  else ntt::intt<10>(reinterpret_cast<ff::ff_t*>(poly));
}
// element-wise field operations on canonical residues: op 0 add, 1 sub, 2 mul, 3 neg(a)
void fr_ff_op(int op, size_t n, const uint16_t* a, const uint16_t* b, uint16_t* out)
{
  for (size_t i = 0; i < n; i++) {
    const ff::ff_t x{ a[i] }, y{ b[i] };
    const ff::ff_t r = op == 0 ? x + y : op == 1 ? x - y : op == 2 ? x * y : -x;
    out[i] = r.v;
  }
}
uint64_t fr_verify_after_hash(unsigned logn, const uint16_t* h, const uint16_t* c, const int32_t* s2, int16_t* s1)
{
  return logn == 9 ? verify_after_hash_n<512>(h, c, s2, s1) : verify_after_hash_n<1024>(h, c, s2, s1);
}

}  // extern "C"
