// falcon_amd.hpp -- C++20 host side of the MI355X batched Falcon verifier.
//
// Keeps the reference's verify API surface and adds the batch entry point (BASELINE.json
// north_star). Each function names the reference interface it stands in for (file:line under the
// reference tree):
//
//   falcon::verify<N>(pkey, msg, mlen, sig)            falcon.hpp:230-249   same signature, same bool
//   falcon::verify<N>(h, msg, mlen, sig)               verification.hpp:20-26   one call with the key kept decoded
//   falcon::verify_many<N>(count, pkeys, msgs, ...)    no counterpart: `count` independent calls of the above
//   falcon::prepared_keys<N> + verify_many<N>(count, keys, key_idx, ...)
//                                                      verification.hpp:20-26   verify with the key kept decoded
//   falcon::resident_keys<N> + verify_many<N>(count, keys, key_idx, ...)
//                                                      the same with the keys kept in the GPU's memory across calls
//   falcon::verify_many_sharded<N> / verify_many_sharded_device<N>
//                                                      no counterpart: one batch over several GPUs, host or device buffers
//   falcon::hash_to_point_many<N>(count, salts, ...)   hashing.hpp:16-50
//   falcon::compute_public_keys<N>(count, f, g, h)     keygen.hpp:67-90
//   falcon::ntt_many<N> / intt_many<N>(count, a, out)  ntt.hpp:120-150 / 161-200
//   falcon::decode_pkeys<N>(count, pkeys, h, ok)       decoding.hpp:18-48
//   falcon::decode_sigs<N>(count, sigs, salts, s2, ok) decoding.hpp:369-391
//   falcon::encode_pkeys<N>(count, h, pkeys)           encoding.hpp:12-35
//   falcon::compress_sigs<N>(count, s2, sigs, ok)      encoding.hpp:169-230
//
// Everything lives in namespace falcon_gpu. Unless FALCON_AMD_NO_REFERENCE_NAMESPACE is defined, the same names
// are also put into namespace falcon (the reference's own: a drop-in for code that includes this header INSTEAD of
// the reference's falcon.hpp). A translation unit that includes BOTH libraries includes falcon_gpu.hpp, which
// defines that macro, and calls falcon_gpu::verify<N> beside falcon::verify<N>.
//
// All of it runs on the GPU through the C ABI of falcon_amd.h (libfalcon_amd.so); there is no CPU path
// in this header. Conventions are the reference's: the caller owns every
// buffer, record lengths are implied by N (utils.hpp:9-17,39-49), the only per-item result is the
// bool / accept byte. Errors of the device layer (no GPU, out of memory) are not representable in
// the reference's `bool`, so they throw falcon::device_error.
//
// A context (device tables + workspace) is created per (thread, device) on first use, so the
// functions are safe to call concurrently from any number of threads, as the reference's are
// (verify_many_sharded, whose worker threads live for one call, uses one shared context per device, and so do
// concurrent single verify<N> calls, which are combined into batches: detail::call_combiner).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <exception>
#include <stdexcept>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "falcon_amd.h"

namespace falcon_gpu {

struct device_error : std::runtime_error {
  int code;
  device_error(const char* what, int c) : std::runtime_error(std::string(what) + ": " + fa_strerror(c)), code(c) {}
};

// Norm policy, SURVEY.md section 0.1: `exact` is what verification.hpp:70-79 means and what a
// clang++-built reference returns; `wrap32` reproduces a g++-built reference on invalid inputs.
enum class norm_policy : int { exact = FA_NORM_EXACT64, wrap32 = FA_NORM_WRAP32 };

namespace detail {

template<size_t N>
constexpr int logn()
{
  static_assert(N == 512 || N == 1024, "N must be 512 or 1024");  // falcon.hpp:236
  return N == 512 ? 9 : 10;
}

// One context per (thread, device), destroyed at thread exit.
class context_cache
{
public:
  fa_ctx* get(int device)
  {
    for (auto& s : slots_)
      if (s.ctx && s.device == device) return s.ctx;
    for (auto& s : slots_) {
      if (s.ctx) continue;
      const int rc = fa_ctx_create(device, &s.ctx);
      if (rc) throw device_error("fa_ctx_create", rc);
      s.device = device;
      return s.ctx;
    }
    throw device_error("fa_ctx_create (more than 16 devices per thread)", FA_E_ARG);
  }
  ~context_cache()
  {
    for (auto& s : slots_)
      if (s.ctx) fa_ctx_destroy(s.ctx);
  }

private:
  struct slot {
    fa_ctx* ctx = nullptr;
    int device = -1;
  };
  slot slots_[16];
};

// size_t message length -> the C ABI's uint32_t; lengths the library cannot take throw instead of being truncated
inline uint32_t msg_len(size_t mlen)
{
  if (mlen > FA_MAX_MSG_LEN) throw device_error("message length", FA_E_ARG);
  return static_cast<uint32_t>(mlen);
}

inline fa_ctx* context(int device)
{
  thread_local context_cache cache;
  return cache.get(device);
}

// One more context per device, shared by the whole process and used under its mutex: verify_many_sharded's worker
// threads live for one call, and a per-thread context would be created (twiddle tables, workspace, staging buffers:
// gigabytes of hipMalloc) and destroyed again on every call. These contexts are kept until the process ends and are
// deliberately not destroyed from a static destructor, which could run after the HIP runtime has shut down.
struct shared_context {
  std::mutex lock;
  fa_ctx* ctx = nullptr;
};
inline int& thread_wire_flag()
{
  thread_local int flag = FA_WIRE_REFERENCE;
  return flag;
}
inline shared_context& shared_context_for(int device)
{
  constexpr int MAX_DEVICES = 64;
  static shared_context* const slots = new shared_context[MAX_DEVICES];
  if (device < 0 || device >= MAX_DEVICES) throw device_error("verify_many_sharded (device index)", FA_E_ARG);
  return slots[device];
}

}  // namespace detail

// Wire format of keys and signatures for this thread's context on `device` (FA_OPT_WIRE_FORMAT): `reference`, the
// default, is the reference's own format and rules; `spec` reads the Falcon specification's formats instead, which
// the reference does not (README.md:74-75) -- no counterpart there, see falcon_amd.h.
enum class wire_format : int { reference = FA_WIRE_REFERENCE, spec = FA_WIRE_SPEC };
inline void set_wire_format(wire_format w, int device = 0)
{
  const int rc = fa_set_option(detail::context(device), FA_OPT_WIRE_FORMAT, (int)w);
  if (rc) throw device_error("fa_set_option", rc);
  detail::thread_wire_flag() = (int)w;  // single calls of this thread are combined with other threads' only in the default format
}

// Record sizes, utils.hpp:9-17 and :39-49.
template<size_t N> constexpr size_t pkey_len() { return 1 + 14 * N / 8; }
template<size_t N> constexpr size_t sig_len() { return N == 512 ? 666 : 1280; }

// Pinned host memory (fa_host_alloc): batches assembled in it cross the host link without the staging copy
// that pageable memory (std::vector, new[]) needs. No reference counterpart.
template<typename T>
class pinned_buffer
{
public:
  explicit pinned_buffer(size_t count) : size_(count), p_(static_cast<T*>(fa_host_alloc(count > 0 ? count * sizeof(T) : 1)))
  {
    if (!p_) throw device_error("fa_host_alloc", FA_E_INTERNAL);
  }
  pinned_buffer(const pinned_buffer&) = delete;
  pinned_buffer& operator=(const pinned_buffer&) = delete;
  pinned_buffer(pinned_buffer&& o) noexcept : size_(o.size_), p_(o.p_) { o.p_ = nullptr; }
  ~pinned_buffer() { fa_host_free(p_); }
  T* data() { return p_; }
  const T* data() const { return p_; }
  size_t size() const { return size_; }
  T& operator[](size_t i) { return p_[i]; }
  const T& operator[](size_t i) const { return p_[i]; }

private:
  size_t size_;
  T* p_;
};

// Batch of `count` independent falcon::verify<N> calls (falcon.hpp:230-249).
//   pkeys  count * pkey_len<N>() bytes, sigs  count * sig_len<N>() bytes, contiguous records;
//   msgs   msg_off == nullptr: message i is the mlen bytes at msgs + i*mlen,
//          else message i is msgs[msg_off[i] .. msg_off[i+1]) and mlen is ignored;
//   accept one byte per item: 1 iff the reference returns true for that triple.
template<size_t N>
inline void verify_many(size_t count, const uint8_t* pkeys, const uint8_t* msgs, size_t mlen, const uint64_t* msg_off,
                        const uint8_t* sigs, uint8_t* accept, int device = 0, norm_policy policy = norm_policy::exact,
                        const fa_debug* debug = nullptr)
{
  const int rc = fa_verify_many(detail::context(device), detail::logn<N>(), count, pkeys, msgs, msg_off, detail::msg_len(mlen),
                                sigs, (int)policy, accept, debug);
  if (rc) throw device_error("fa_verify_many", rc);
}

// The same on buffers that already live in the HBM of `device` (hipMalloc'ed, 4-byte aligned);
// `stream` is a hipStream_t. The call returns after the stream has been synchronised once.
template<size_t N>
inline void verify_many_device(size_t count, const uint8_t* d_pkeys, const uint8_t* d_msgs, size_t mlen,
                               const uint64_t* d_msg_off, const uint8_t* d_sigs, uint8_t* d_accept, int device = 0,
                               void* stream = nullptr, norm_policy policy = norm_policy::exact)
{
  const int rc = fa_verify_many_dev(detail::context(device), detail::logn<N>(), count, d_pkeys, d_msgs, d_msg_off,
                                    detail::msg_len(mlen), d_sigs, (int)policy, d_accept, nullptr, stream);
  if (rc) throw device_error("fa_verify_many_dev", rc);
}

// Multi-GPU form: the batch is cut into devices.size() contiguous ranges of
// ceil(count / devices.size()) items, range g goes to GPU devices[g] on its own host thread, and every range
// writes its slice of `accept` -- items are independent, so there is no collective and nothing to merge beyond
// that concatenation. Messages must be fixed-stride or offset-indexed as above (offsets are global, so every
// shard reads the same msgs/msg_off arrays). A device may be named more than once (its ranges then run one after
// the other on that GPU's shared context). `wire` is applied to each shared context for the call. If a range
// fails, its slice of `accept` is zeroed and the first error is rethrown after all ranges have finished.
template<size_t N>
inline void verify_many_sharded(size_t count, const uint8_t* pkeys, const uint8_t* msgs, size_t mlen,
                                const uint64_t* msg_off, const uint8_t* sigs, uint8_t* accept, const std::vector<int>& devices,
                                norm_policy policy = norm_policy::exact, wire_format wire = wire_format::reference)
{
  const size_t ndev = devices.size();
  if (ndev < 1) throw device_error("verify_many_sharded", FA_E_ARG);
  const size_t per = (count + ndev - 1) / ndev;
  std::vector<std::exception_ptr> errors(ndev);
  auto run_range = [&](size_t g) noexcept {
    const size_t lo = per * g < count ? per * g : count, hi = lo + per < count ? lo + per : count;
    if (lo >= hi) return;
    try {
      detail::shared_context& sc = detail::shared_context_for(devices[g]);
      std::lock_guard<std::mutex> hold(sc.lock);  // a context serves one host thread at a time
      if (!sc.ctx) {
        const int rc = fa_ctx_create(devices[g], &sc.ctx);
        if (rc) throw device_error("fa_ctx_create", rc);
      }
      int rc = fa_set_option(sc.ctx, FA_OPT_WIRE_FORMAT, (int)wire);
      if (rc) throw device_error("fa_set_option", rc);
      // with msg_off the shard passes the whole message array and its own window of offsets
      rc = fa_verify_many(sc.ctx, detail::logn<N>(), hi - lo, pkeys + lo * pkey_len<N>(), msg_off ? msgs : msgs + lo * mlen,
                          msg_off ? msg_off + lo : nullptr, detail::msg_len(mlen), sigs + lo * sig_len<N>(), (int)policy, accept + lo,
                          nullptr);
      if (rc) throw device_error("fa_verify_many", rc);
    } catch (...) {
      for (size_t i = lo; i < hi; i++) accept[i] = 0;
      errors[g] = std::current_exception();
    }
  };
  std::vector<std::thread> workers;
  size_t started = 1;  // the calling thread takes range 0
  try {
    workers.reserve(ndev);
    for (; started < ndev; started++) workers.emplace_back(run_range, started);
  } catch (...) {  // threads that could not be started (thread limit, bad_alloc): their ranges run inline below
  }
  run_range(0);
  for (auto& w : workers) w.join();
  for (size_t g = started; g < ndev; g++) run_range(g);
  for (auto& e : errors)
    if (e) std::rethrow_exception(e);
}

// The same over GPUs first_device .. first_device + ndev - 1.
template<size_t N>
inline void verify_many_sharded(size_t count, const uint8_t* pkeys, const uint8_t* msgs, size_t mlen,
                                const uint64_t* msg_off, const uint8_t* sigs, uint8_t* accept, int ndev,
                                int first_device = 0, norm_policy policy = norm_policy::exact,
                                wire_format wire = wire_format::reference)
{
  if (ndev < 1) throw device_error("verify_many_sharded", FA_E_ARG);
  std::vector<int> devices((size_t)ndev);
  for (int g = 0; g < ndev; g++) devices[(size_t)g] = first_device + g;
  verify_many_sharded<N>(count, pkeys, msgs, mlen, msg_off, sigs, accept, devices, policy, wire);
}

// Multi-GPU form for batches that already live in the GPUs' memory (one host thread, one
// context and one queue per GPU): every shard names its device and its own device-resident arrays, laid out as
// verify_many_device takes them. One host thread per distinct device enqueues all of that device's shards
// asynchronously (FA_OPT_ASYNC) on the device's default stream and waits once (fa_finish); a shard the device-side
// retry pass could not settle (probability ~1e-50 per item) is verified again synchronously. No host staging, no
// collective: N GPUs run at N times the device-resident rate. The first error is rethrown after all devices finished.
struct device_shard {
  int device = 0;
  size_t count = 0;
  const uint8_t* d_pkeys = nullptr;
  const uint8_t* d_msgs = nullptr;
  const uint64_t* d_msg_off = nullptr;  // nullable: fixed-length messages
  const uint8_t* d_sigs = nullptr;
  uint8_t* d_accept = nullptr;
};
template<size_t N>
inline void verify_many_sharded_device(const std::vector<device_shard>& shards, size_t mlen, norm_policy policy = norm_policy::exact)
{
  std::vector<int> devices;  // distinct devices, in order of first appearance
  for (const device_shard& sh : shards) {
    bool seen = false;
    for (int d : devices) seen |= d == sh.device;
    if (!seen) devices.push_back(sh.device);
  }
  const uint32_t ml = detail::msg_len(mlen);
  std::vector<std::exception_ptr> errors(devices.size());
  auto run_device = [&](size_t g) noexcept {
    try {
      detail::shared_context& sc = detail::shared_context_for(devices[g]);
      std::lock_guard<std::mutex> hold(sc.lock);
      if (!sc.ctx) {
        const int rc = fa_ctx_create(devices[g], &sc.ctx);
        if (rc) throw device_error("fa_ctx_create", rc);
      }
      int rc = fa_set_option(sc.ctx, FA_OPT_WIRE_FORMAT, FA_WIRE_REFERENCE);
      if (!rc) rc = fa_set_option(sc.ctx, FA_OPT_ASYNC, 1);
      if (rc) throw device_error("fa_set_option", rc);
      int first = 0;
      for (const device_shard& sh : shards)
        if (sh.device == devices[g] && !first)
          first = fa_verify_many_dev(sc.ctx, detail::logn<N>(), sh.count, sh.d_pkeys, sh.d_msgs, sh.d_msg_off, ml, sh.d_sigs, (int)policy,
                                     sh.d_accept, nullptr, nullptr);
      uint64_t unresolved = 0;
      const int rf = fa_finish(sc.ctx, nullptr, &unresolved);  // also after a failure: nothing of this call stays in flight
      (void)fa_set_option(sc.ctx, FA_OPT_ASYNC, 0);
      if (first) throw device_error("fa_verify_many_dev", first);
      if (rf == FA_E_UNRESOLVED) {  // settle the stragglers with the synchronising call (its host loop takes any number of blocks)
        for (const device_shard& sh : shards)
          if (sh.device == devices[g]) {
            rc = fa_verify_many_dev(sc.ctx, detail::logn<N>(), sh.count, sh.d_pkeys, sh.d_msgs, sh.d_msg_off, ml, sh.d_sigs, (int)policy,
                                    sh.d_accept, nullptr, nullptr);
            if (rc) throw device_error("fa_verify_many_dev", rc);
          }
      } else if (rf) {
        throw device_error("fa_finish", rf);
      }
    } catch (...) {
      errors[g] = std::current_exception();
    }
  };
  std::vector<std::thread> workers;
  size_t started = 1;  // the calling thread takes the first device
  try {
    workers.reserve(devices.size());
    for (; started < devices.size(); started++) workers.emplace_back(run_device, started);
  } catch (...) {  // threads that could not be started: their devices run inline below
  }
  if (!devices.empty()) run_device(0);
  for (auto& w : workers) w.join();
  for (size_t g = started; g < devices.size(); g++) run_device(g);
  for (auto& e : errors)
    if (e) std::rethrow_exception(e);
}

// ---- decoded-key form: verification::verify<N, beta2>(h, msg, mlen, sig), verification.hpp:20-26 ----
// Keys are prepared once (decoded + forward NTT, kept in HBM-friendly uint16 form on the host side here)
// and reused for any number of signatures; beta2 is implied by N (falcon.hpp:245-246).
template<size_t N>
class prepared_keys
{
public:
  // from encoded public keys (decode_pkey, decoding.hpp:18-48); a key with a wrong header rejects every item using it
  static prepared_keys from_encoded(size_t nkeys, const uint8_t* pkeys, int device = 0)
  {
    prepared_keys k(nkeys);
    const int rc = fa_prepare_keys(detail::context(device), detail::logn<N>(), nkeys, pkeys, nullptr, k.blob_.data());
    if (rc) throw device_error("fa_prepare_keys", rc);
    return k;
  }
  // from decoded polynomials h (N coefficients each, natural order), the type verification::verify takes
  static prepared_keys from_decoded(size_t nkeys, const uint16_t* h, int device = 0)
  {
    prepared_keys k(nkeys);
    const int rc = fa_prepare_keys(detail::context(device), detail::logn<N>(), nkeys, nullptr, h, k.blob_.data());
    if (rc) throw device_error("fa_prepare_keys", rc);
    return k;
  }
  size_t size() const { return nkeys_; }
  const uint16_t* data() const { return blob_.data(); }

private:
  explicit prepared_keys(size_t nkeys) : nkeys_(nkeys), blob_(nkeys * N) {}
  size_t nkeys_;
  std::vector<uint16_t> blob_;
};

// Item i is checked against keys[key_idx[i]] (key_idx == nullptr: against keys[i]).
template<size_t N>
inline void verify_many(size_t count, const prepared_keys<N>& keys, const uint32_t* key_idx, const uint8_t* msgs, size_t mlen,
                        const uint64_t* msg_off, const uint8_t* sigs, uint8_t* accept, int device = 0,
                        norm_policy policy = norm_policy::exact)
{
  const int rc = fa_verify_many_prepared(detail::context(device), detail::logn<N>(), count, keys.data(), keys.size(), key_idx,
                                         msgs, msg_off, detail::msg_len(mlen), sigs, (int)policy, accept, nullptr);
  if (rc) throw device_error("fa_verify_many_prepared", rc);
}

// The decoded-key form for callers that keep their keys ACROSS calls (falcon.hpp:226-229; benchmarks/bench_signing.cpp:105
// keeps `h` for the whole run): the prepared keys stay in the memory of GPU `device` (fa_keyset), and a verify_many call
// sends only signatures, messages and 4-byte key indices over the host link -- 702 instead of 1,595 bytes per Falcon-512
// item with a 32-byte message. prepared_keys<N> above holds the same data in host memory and uploads it with every call.
template<size_t N>
class resident_keys
{
public:
  static resident_keys from_encoded(size_t nkeys, const uint8_t* pkeys, int device = 0)
  {
    return resident_keys(nkeys, pkeys, nullptr, device);
  }
  static resident_keys from_decoded(size_t nkeys, const uint16_t* h, int device = 0)
  {
    return resident_keys(nkeys, nullptr, h, device);
  }
  resident_keys(const resident_keys&) = delete;
  resident_keys& operator=(const resident_keys&) = delete;
  resident_keys(resident_keys&& o) noexcept : ks_(o.ks_), device_(o.device_) { o.ks_ = nullptr; }
  resident_keys& operator=(resident_keys&& o) noexcept
  {
    if (this != &o) {
      fa_keyset_destroy(ks_);
      ks_ = o.ks_, device_ = o.device_;
      o.ks_ = nullptr;
    }
    return *this;
  }
  ~resident_keys() { fa_keyset_destroy(ks_); }
  size_t size() const { return fa_keyset_size(ks_); }
  int device() const { return device_; }
  const fa_keyset* handle() const { return ks_; }
  const uint16_t* device_data() const { return fa_keyset_device_keys(ks_); }  // for verify_many_prepared_dev-style callers

private:
  resident_keys(size_t nkeys, const uint8_t* pkeys, const uint16_t* h, int device) : device_(device)
  {
    const int rc = fa_keyset_create(detail::context(device), detail::logn<N>(), nkeys, pkeys, h, &ks_);
    if (rc) throw device_error("fa_keyset_create", rc);
  }
  fa_keyset* ks_ = nullptr;
  int device_ = 0;
};

// Item i is checked against keys[key_idx[i]] (key_idx == nullptr: against keys[i]); runs on the GPU that holds the keys.
template<size_t N>
inline void verify_many(size_t count, const resident_keys<N>& keys, const uint32_t* key_idx, const uint8_t* msgs, size_t mlen,
                        const uint64_t* msg_off, const uint8_t* sigs, uint8_t* accept, norm_policy policy = norm_policy::exact)
{
  const int rc = fa_verify_many_keyset(detail::context(keys.device()), keys.handle(), count, key_idx, msgs, msg_off,
                                       detail::msg_len(mlen), sigs, (int)policy, accept, nullptr);
  if (rc) throw device_error("fa_verify_many_keyset", rc);
}

// hashing::hash_to_point<N>(salt, 40, msg, mlen, poly) (hashing.hpp:16-50) for `count` (salt, message) pairs.
template<size_t N>
inline void hash_to_point_many(size_t count, const uint8_t* salts, const uint8_t* msgs, size_t mlen, const uint64_t* msg_off,
                               uint16_t* c, int device = 0)
{
  const int rc = fa_hash_to_point_many(detail::context(device), detail::logn<N>(), count, salts, msgs, msg_off, detail::msg_len(mlen), c);
  if (rc) throw device_error("fa_hash_to_point_many", rc);
}

// keygen::compute_public_key<N>(f, g, h) (keygen.hpp:67-90) for `count` (f, g) pairs: h = g / f mod (x^N + 1, q).
template<size_t N>
inline void compute_public_keys(size_t count, const int32_t* f, const int32_t* g, uint16_t* h, int device = 0)
{
  const int rc = fa_compute_public_keys(detail::context(device), detail::logn<N>(), count, f, g, h);
  if (rc) throw device_error("fa_compute_public_keys", rc);
}

// ntt::ntt<log2 N>(poly) / ntt::intt<log2 N>(poly) (ntt.hpp:120-150, 161-200) on `count` polynomials, out of place:
// natural order in and bit-reversed order out for ntt_many, the converse (with the 1/N scaling) for intt_many.
template<size_t N>
inline void ntt_many(size_t count, const uint16_t* a, uint16_t* out, int device = 0)
{
  const int rc = fa_ntt_many(detail::context(device), detail::logn<N>(), count, a, out, 0);
  if (rc) throw device_error("fa_ntt_many", rc);
}
template<size_t N>
inline void intt_many(size_t count, const uint16_t* a, uint16_t* out, int device = 0)
{
  const int rc = fa_ntt_many(detail::context(device), detail::logn<N>(), count, a, out, 1);
  if (rc) throw device_error("fa_ntt_many", rc);
}

// decoding::decode_pkey<N>(pkey, h) (decoding.hpp:18-48) for `count` keys; ok[i] is the reference's bool.
template<size_t N>
inline void decode_pkeys(size_t count, const uint8_t* pkeys, uint16_t* h, uint8_t* ok, int device = 0)
{
  const int rc = fa_decode_pkeys(detail::context(device), detail::logn<N>(), count, pkeys, h, ok);
  if (rc) throw device_error("fa_decode_pkeys", rc);
}

// decoding::decode_sig<N>(sig, salt, s2) (decoding.hpp:369-391) for `count` signatures; s2 as int16 (the reference's
// int32 values lie in [-1023, 1023]); rows of failing items are written exactly as far as the reference writes them.
template<size_t N>
inline void decode_sigs(size_t count, const uint8_t* sigs, uint8_t* salts, int16_t* s2, uint8_t* ok, int device = 0)
{
  const int rc = fa_decode_sigs(detail::context(device), detail::logn<N>(), count, sigs, salts, s2, ok);
  if (rc) throw device_error("fa_decode_sigs", rc);
}

// encoding::encode_pkey<N>(h, pkey) (encoding.hpp:12-35) for `count` polynomials: the inverse of decode_pkeys.
template<size_t N>
inline void encode_pkeys(size_t count, const uint16_t* h, uint8_t* pkeys, int device = 0)
{
  const int rc = fa_encode_pkeys(detail::context(device), detail::logn<N>(), count, h, pkeys);
  if (rc) throw device_error("fa_encode_pkeys", rc);
}

// encoding::compress_sig<N, sig_len<N>()>(poly_s, sig) (encoding.hpp:169-230) for `count` polynomials (int16 coefficients):
// writes bytes 41.. of each signature only, ok[i] is the reference's bool (0: the code did not fit, body zeroed).
template<size_t N>
inline void compress_sigs(size_t count, const int16_t* s2, uint8_t* sigs, uint8_t* ok, int device = 0)
{
  const int rc = fa_compress_sigs(detail::context(device), detail::logn<N>(), count, s2, sigs, ok);
  if (rc) throw device_error("fa_compress_sigs", rc);
}

namespace detail {

// Single calls from many host threads at once. One falcon::verify<N> costs the GPU path about 50 us whatever else is
// going on, and calls issued concurrently through contexts of their own queue up behind each other in the runtime
// (measured: 8 threads, 33,000 calls a second in all, 240 us each). A call of up to a thousand items costs about the
// same 50-60 us as a call of one, so concurrent callers are COMBINED: a caller appends its request to a queue; if no
// batch is in flight it becomes the leader, takes what is queued (its own request included), runs it as ONE
// fa_verify_many call on the process-wide context of device 0 and hands the results out; callers that arrive meanwhile
// wait, and the first of them to wake leads the next batch. Alone, a thread pays a mutex and a 2 KB copy on top of the
// call it would have made anyway. No counterpart in the reference (its verify is a pure function on the calling core).
template<size_t N>
class call_combiner
{
public:
  static call_combiner& instance()
  {
    static call_combiner* const c = new call_combiner;  // kept until the process ends, like the shared contexts
    return *c;
  }

  bool verify(const uint8_t* pkey, const uint8_t* msg, uint32_t mlen, const uint8_t* sig)
  {
    request r{ pkey, msg, mlen, sig, 0, 0, { false } };
    bool lead;
    {
      queue_guard hold(*this);  // released on every way out, an exception from the push included
      pending_.push_back(&r);
      // acquire: the thread that takes the lead must see everything the previous leader wrote (batch_, the gather
      // buffers, last_batch_) before it released busy_
      lead = !busy_.load(std::memory_order_acquire);
      if (lead) busy_.store(true, std::memory_order_relaxed);
    }
    while (!lead) {
      // Wait for the result or for the leader's place. A batch takes 50-100 us, a sleeping thread about as long to be
      // woken: watch the two flags for a while, then poll at intervals.
      const auto t0 = std::chrono::steady_clock::now();
      while (!r.done.load(std::memory_order_acquire) && busy_.load(std::memory_order_acquire)) {
        if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(400))
          relax();
        else
          std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
      if (r.done.load(std::memory_order_acquire)) break;  // served: leaves without touching the queue again
      queue_guard hold(*this);
      lead = !r.done.load(std::memory_order_acquire) && !busy_.load(std::memory_order_acquire);
      if (lead) busy_.store(true, std::memory_order_relaxed);
    }
    if (lead) {  // batches until this thread's own request has been served (normally the first one taken)
      while (!r.done.load(std::memory_order_relaxed)) {
        if (last_batch_ > 1) {  // callers served by the previous batch are about to come back: give them a few microseconds
          const auto t0 = std::chrono::steady_clock::now();
          while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(5)) relax();
        }
        batch_.clear();
        try {
          batch_.reserve(MAX_BATCH);  // outside the queue's lock: the pushes below cannot throw
        } catch (...) {
          fail_own(r);  // out of memory: this thread's request fails, the lead passes on
          break;
        }
        size_t bytes = 0;
        {
          queue_guard hold(*this);
          while (!pending_.empty() && batch_.size() < MAX_BATCH && bytes <= MAX_MSG_BYTES) {
            bytes += pending_.front()->mlen;
            batch_.push_back(pending_.front());
            pending_.pop_front();
          }
        }
        run();
        last_batch_ = batch_.size();
        for (request* q : batch_) q->done.store(true, std::memory_order_release);  // *q may be gone right after this store
      }
      busy_.store(false, std::memory_order_release);  // a waiter whose request is still queued takes over
    }
    if (r.rc) throw device_error("fa_verify_many", r.rc);
    return r.ok != 0;
  }

  // batches run and requests served so far (diagnostics; may be read while other threads are calling)
  void stats(uint64_t* batches, uint64_t* requests, double* run_us = nullptr) const
  {
    *batches = batches_.load(std::memory_order_relaxed), *requests = requests_.load(std::memory_order_relaxed);
    if (run_us) *run_us = (double)run_ns_.load(std::memory_order_relaxed) * 1e-3;
  }

private:
  struct request {
    const uint8_t* pkey;
    const uint8_t* msg;
    uint32_t mlen;
    const uint8_t* sig;
    uint8_t ok;
    int rc;
    std::atomic<bool> done;
  };
  static constexpr size_t MAX_BATCH = 256, MAX_MSG_BYTES = 1u << 19;
  static void relax()
  {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }

  void run() noexcept  // leader only, queue lock released: batch_ and the gather buffers belong to the leader
  {
    int rc = 0;
    const size_t n = batch_.size();
    batches_.fetch_add(1, std::memory_order_relaxed), requests_.fetch_add(n, std::memory_order_relaxed);
    const auto t_run = std::chrono::steady_clock::now();
    try {
      pk_.resize(n * pkey_len_), sig_.resize(n * sig_len_), off_.resize(n + 1), acc_.assign(n, 0);
      size_t total = 0;
      for (size_t i = 0; i < n; i++) off_[i] = total, total += batch_[i]->mlen;
      off_[n] = total;
      msg_.resize(total + 8);
      for (size_t i = 0; i < n; i++) {
        std::memcpy(pk_.data() + i * pkey_len_, batch_[i]->pkey, pkey_len_);
        std::memcpy(sig_.data() + i * sig_len_, batch_[i]->sig, sig_len_);
        if (batch_[i]->mlen) std::memcpy(msg_.data() + off_[i], batch_[i]->msg, batch_[i]->mlen);
      }
      shared_context& sc = shared_context_for(0);
      std::lock_guard<std::mutex> hold(sc.lock);
      if (!sc.ctx) rc = fa_ctx_create(0, &sc.ctx);
      if (!rc) rc = fa_set_option(sc.ctx, FA_OPT_WIRE_FORMAT, FA_WIRE_REFERENCE);
      if (!rc)
        rc = fa_verify_many(sc.ctx, logn<N>(), n, pk_.data(), msg_.data(), off_.data(), 0, sig_.data(), FA_NORM_EXACT64, acc_.data(), nullptr);
    } catch (...) {
      rc = FA_E_INTERNAL;
    }
    for (size_t i = 0; i < n; i++) batch_[i]->ok = rc ? 0 : acc_[i], batch_[i]->rc = rc;
    run_ns_.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_run).count(),
                      std::memory_order_relaxed);
  }

  // The leader could not even reserve its batch: take its own request out of the queue (if it is still there) and fail it.
  void fail_own(request& r)
  {
    queue_guard hold(*this);
    for (auto it = pending_.begin(); it != pending_.end(); ++it)
      if (*it == &r) {
        pending_.erase(it);
        r.rc = FA_E_INTERNAL;
        r.done.store(true, std::memory_order_release);
        break;
      }
  }

  static constexpr size_t pkey_len_ = 1 + 14 * N / 8, sig_len_ = N == 512 ? 666 : 1280;
  // the queue's lock is held for a push or a drain, a fraction of a microsecond: a spin lock, because eight threads
  // meeting at a std::mutex put each other to sleep for longer than a whole batch takes
  void lock()
  {
    while (qlock_.test_and_set(std::memory_order_acquire)) relax();
  }
  void unlock() { qlock_.clear(std::memory_order_release); }
  struct queue_guard {  // the spin lock as a scope
    call_combiner& c;
    explicit queue_guard(call_combiner& cc) : c(cc) { c.lock(); }
    ~queue_guard() { c.unlock(); }
    queue_guard(const queue_guard&) = delete;
    queue_guard& operator=(const queue_guard&) = delete;
  };
  std::atomic_flag qlock_ = ATOMIC_FLAG_INIT;
  std::deque<request*> pending_;
  std::atomic<bool> busy_{ false };
  std::vector<request*> batch_;
  std::vector<uint8_t> pk_, sig_, msg_, acc_;
  std::vector<uint64_t> off_;
  std::atomic<uint64_t> batches_{ 0 }, requests_{ 0 };
  std::atomic<uint64_t> run_ns_{ 0 };  // time spent inside run()
  size_t last_batch_ = 0;  // size of the previous batch: more than one means other threads are calling too
};

}  // namespace detail

// falcon.hpp:230-249. `sig` is non-const there too (never written). Calls made by several threads at the same time
// are combined into one batch (detail::call_combiner) unless FALCON_AMD_NO_CALL_COMBINING is defined or the calling
// thread has chosen the specification wire format.
template<size_t N>
static inline bool verify(const uint8_t* __restrict pkey, const uint8_t* __restrict msg, const size_t mlen,
                          uint8_t* __restrict sig)
  requires((N == 512) || (N == 1024))
{
#ifndef FALCON_AMD_NO_CALL_COMBINING
  if (detail::thread_wire_flag() == FA_WIRE_REFERENCE)
    return detail::call_combiner<N>::instance().verify(pkey, msg, detail::msg_len(mlen), sig);
#endif
  uint8_t ok = 0;
  verify_many<N>(1, pkey, msg, mlen, nullptr, sig, &ok);
  return ok != 0;
}

// verification::verify<N, beta2>(h, msg, mlen, sig) (verification.hpp:20-26; caller benchmarks/bench_signing.cpp:105):
// one call with the public key kept decoded -- h is N coefficients in natural order, as decode_pkey leaves them
// (the reference's ff_t is a uint16_t wrapper; any value is taken mod q as its arithmetic does). beta2 is implied
// by N (falcon.hpp:245-246). A caller with many signatures per key prepares the key once (prepared_keys) instead.
template<size_t N>
static inline bool verify(const uint16_t* __restrict h, const uint8_t* __restrict msg, const size_t mlen,
                          const uint8_t* __restrict sig, int device = 0)
  requires((N == 512) || (N == 1024))
{
  // A polynomial whose coefficients fit 14 bits is exactly what decode_pkey (decoding.hpp:34-45) yields from the key
  // packed the obvious way, and the packed path takes raw fields as they are: pack it on the stack (a microsecond) and
  // make the ordinary single call -- one round trip, combinable with other threads' calls -- instead of preparing a key
  // on the GPU, copying it back and verifying against it (three round trips).
  bool fits = device == 0;
  for (size_t i = 0; i < N && fits; i++) fits = h[i] < 16384u;
  if (fits) {
    uint8_t pk[pkey_len<N>()] = {};
    pk[0] = (uint8_t)detail::logn<N>();
    for (size_t i = 0; i < N; i++) {  // coefficient i at bit 14 i behind the header byte, least significant bit first
      const size_t bit = 14 * i;
      const uint32_t v = (uint32_t)h[i] << (bit & 7);
      pk[1 + (bit >> 3)] |= (uint8_t)v;
      pk[2 + (bit >> 3)] |= (uint8_t)(v >> 8);
      if ((bit & 7) > 2) pk[3 + (bit >> 3)] |= (uint8_t)(v >> 16);
    }
    return verify<N>(pk, msg, mlen, const_cast<uint8_t*>(sig));
  }
  const prepared_keys<N> key = prepared_keys<N>::from_decoded(1, h, device);
  uint8_t ok = 0;
  verify_many<N>(1, key, nullptr, msg, mlen, nullptr, sig, &ok, device);
  return ok != 0;
}

}  // namespace falcon_gpu

#ifndef FALCON_AMD_NO_REFERENCE_NAMESPACE
// Drop-in spelling: the reference's own namespace, for code that includes this header instead of falcon.hpp.
namespace falcon {
using namespace falcon_gpu;
}
#endif
