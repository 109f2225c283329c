// host_api_check.cpp -- exercises the C++20 host API of include/falcon_amd.hpp the way the
// reference's own callers use falcon::verify<N> (tests/test_signing.cpp:28-37,
// examples/sign_one.cpp:34): called by tests/test_cpp_host.py on the GPU box.
//
// usage: host_api_check <logn> <count> <mlen> <pk.bin> <msg.bin> <sig.bin> <ndev> [<outdir>]
// prints one line of '0'/'1' accept flags from verify_many<N>, a second line from per-item
// verify<N> calls on the first min(count, 8) items, a third from the prepared-key form, and a fourth from
// verify_many_sharded<N>. With <outdir>: the operators of the path through their C++ names -- decode_pkeys<N>,
// decode_sigs<N>, hash_to_point_many<N> (on the decoded salts), ntt_many<N> / intt_many<N>, prepared_keys<N>::from_decoded
// -- whose raw outputs go to <outdir>/{h,hok,salt,s2,sok,c,ntt,intt}.bin for the caller to compare; a fifth line holds
// the accept flags of verify_many<N> over the keys prepared from the decoded polynomials, a sixth those of the single-call
// decoded-key verify<N>(h, msg, mlen, sig) on the first min(count, 8) items.
#include <algorithm>
#include <atomic>
#include <barrier>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>  // device buffers for verify_many_sharded_device (the library's own headers need no HIP)

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

template<typename T>
static void dump(const std::string& dir, const char* name, const std::vector<T>& v)
{
  std::ofstream f(dir + "/" + name, std::ios::binary);
  f.write((const char*)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}

static void stage(const char* what) { std::fprintf(stderr, "[host_api_check] %s\n", what); }

template<size_t N>
static int run(size_t count, size_t mlen, std::vector<uint8_t>& pk, std::vector<uint8_t>& msg, std::vector<uint8_t>& sig, int ndev,
               const char* outdir)
{
  if (pk.size() != count * falcon::pkey_len<N>() || sig.size() != count * falcon::sig_len<N>() || msg.size() != count * mlen) {
    std::fprintf(stderr, "file sizes do not match count\n");
    return 2;
  }
  stage("verify_many");
  std::vector<uint8_t> acc(count, 7);
  falcon::verify_many<N>(count, pk.data(), msg.data(), mlen, nullptr, sig.data(), acc.data());
  for (uint8_t a : acc) std::putchar('0' + a);
  std::putchar('\n');
  stage("verify, one call per item");
  for (size_t i = 0; i < count && i < 8; i++) {
    const bool ok = falcon::verify<N>(pk.data() + i * falcon::pkey_len<N>(), msg.data() + i * mlen, mlen,
                                      sig.data() + i * falcon::sig_len<N>());
    std::putchar(ok ? '1' : '0');
  }
  std::putchar('\n');
  {  // what one call costs from C++ (stderr: information for the log, not compared): the reference's own benchmark
     // shape, one falcon::verify<N> per signature (benchmarks/bench_verify.cpp)
    const auto t0 = std::chrono::steady_clock::now();
    constexpr int REPS = 200;
    unsigned good = 0;
    for (int r = 0; r < REPS; r++) good += falcon::verify<N>(pk.data(), msg.data(), mlen, sig.data()) ? 1u : 0u;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REPS;
    std::fprintf(stderr, "[host_api_check] falcon::verify<%zu>, one signature per call: %.1f us per call (%u of %d true)\n", N, us, good, REPS);
  }
  {  // the same from several host threads at once, each with its own context (falcon_amd.hpp keeps one per thread and device)
    constexpr int THREADS = 8, CALLS = 100;
    std::vector<std::thread> pool;
    std::vector<unsigned> good(THREADS, 0);
    std::vector<size_t> pick;  // eight accepted items (a malformed one may take the slow decoder: not what is being timed)
    for (size_t i = 0; i < count && pick.size() < 8; i++)
      if (acc[i]) pick.push_back(i);
    if (pick.empty()) pick.push_back(0);
    const size_t span = pick.size();
    uint64_t b0 = 0, q0 = 0, b1 = 0, q1 = 0;
    double u0 = 0, u1 = 0;
    std::barrier ready(THREADS + 1);
    for (int t = 0; t < THREADS; t++)
      pool.emplace_back([&, t] {
        (void)falcon::verify<N>(pk.data(), msg.data(), mlen, sig.data());  // creates this thread's context
        ready.arrive_and_wait();
        for (int r = 0; r < CALLS; r++) {
          const size_t i = pick[(size_t)(t + r) % span];
          const bool ok = falcon::verify<N>(pk.data() + i * falcon::pkey_len<N>(), msg.data() + i * mlen, mlen, sig.data() + i * falcon::sig_len<N>());
          good[(size_t)t] += ok == (acc[i] != 0) ? 1u : 0u;
        }
      });
    ready.arrive_and_wait();
    falcon::detail::call_combiner<N>::instance().stats(&b0, &q0, &u0);
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& th : pool) th.join();
    falcon::detail::call_combiner<N>::instance().stats(&b1, &q1, &u1);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned agree = 0;
    for (unsigned g : good) agree += g;
    std::fprintf(stderr, "[host_api_check] falcon::verify<%zu> from %d threads: %.0f calls per second in all, per call %.1f us, %.1f calls per batch, %.1f us per batch of which %.1f in the library call\n", N,
                 THREADS, THREADS * CALLS / sec, sec / CALLS * 1e6, (double)(q1 - q0) / (double)(b1 - b0), sec * 1e6 / (double)(b1 - b0), (u1 - u0) / (double)(b1 - b0));
    if (agree != THREADS * CALLS) {
      std::fprintf(stderr, "concurrent single calls disagree with the batch: %u of %d\n", agree, THREADS * CALLS);
      return 3;
    }
    // and 16 threads over every kind of item the corpus holds (rejected ones included), different messages lengths apart
    std::vector<std::thread> more;
    std::atomic<unsigned> wrong{ 0 };
    const size_t items = count < 64 ? count : 64;
    for (int t = 0; t < 16; t++)
      more.emplace_back([&, t] {
        for (int r = 0; r < 60; r++) {
          const size_t i = (size_t)(7 * t + 13 * r) % items;
          const bool ok = falcon::verify<N>(pk.data() + i * falcon::pkey_len<N>(), msg.data() + i * mlen, mlen, sig.data() + i * falcon::sig_len<N>());
          if (ok != (acc[i] != 0)) wrong++;
        }
      });
    for (auto& th : more) th.join();
    if (wrong.load()) {
      std::fprintf(stderr, "16 threads, mixed items: %u single calls disagree with the batch\n", wrong.load());
      return 3;
    }
  }
  stage("prepared keys");
  {  // decoded-key form: keys prepared once, item i against key i
    auto keys = falcon::prepared_keys<N>::from_encoded(count, pk.data());
    std::vector<uint8_t> acc3(count, 7);
    falcon::verify_many<N>(count, keys, nullptr, msg.data(), mlen, nullptr, sig.data(), acc3.data());
    for (uint8_t a : acc3) std::putchar('0' + a);
    std::putchar('\n');
  }
  stage("resident keys");
  {  // the same keys kept on the GPU (fa_keyset): item i against key i, then five keys shared by all items through an index
    auto rk = falcon::resident_keys<N>::from_encoded(count, pk.data());
    std::vector<uint8_t> a(count, 7);
    falcon::verify_many<N>(count, rk, nullptr, msg.data(), mlen, nullptr, sig.data(), a.data());
    if (a != acc) {
      std::fprintf(stderr, "resident_keys: differs from verify_many\n");
      return 8;
    }
    const size_t nk = count < 5 ? count : 5;
    std::vector<uint32_t> idx(count);
    for (size_t i = 0; i < count; i++) idx[i] = (uint32_t)(i % nk);
    auto hk = falcon::prepared_keys<N>::from_encoded(nk, pk.data());
    auto rk5 = falcon::resident_keys<N>::from_encoded(nk, pk.data());
    std::vector<uint8_t> want(count, 7), got(count, 9);
    falcon::verify_many<N>(count, hk, idx.data(), msg.data(), mlen, nullptr, sig.data(), want.data());
    falcon::verify_many<N>(count, rk5, idx.data(), msg.data(), mlen, nullptr, sig.data(), got.data());
    if (want != got || !std::equal(got.begin(), got.begin() + (std::ptrdiff_t)nk, acc.begin())) {
      std::fprintf(stderr, "resident_keys with an index: differs from prepared_keys\n");
      return 8;
    }
    bool threw = false;  // an index past the set is refused
    idx[count / 2] = (uint32_t)nk;
    try {
      falcon::verify_many<N>(count, rk5, idx.data(), msg.data(), mlen, nullptr, sig.data(), got.data());
    } catch (const falcon::device_error&) {
      threw = true;
    }
    if (!threw) {
      std::fprintf(stderr, "resident_keys: key index out of range accepted\n");
      return 8;
    }
  }
  stage("verify_many_sharded_device");
  {  // device-resident shards: the batch cut in two (and in three) ranges, all on GPU 0 -- the multi-GPU path's host logic
    uint8_t *d_pk = nullptr, *d_msg = nullptr, *d_sig = nullptr, *d_acc = nullptr;
    auto hip_ok = [](hipError_t e) { return e == hipSuccess; };
    if (!hip_ok(hipMalloc((void**)&d_pk, pk.size() + 8)) || !hip_ok(hipMalloc((void**)&d_msg, msg.size() + 8)) ||
        !hip_ok(hipMalloc((void**)&d_sig, sig.size() + 8)) || !hip_ok(hipMalloc((void**)&d_acc, count + 8)) ||
        !hip_ok(hipMemcpy(d_pk, pk.data(), pk.size(), hipMemcpyHostToDevice)) ||
        !hip_ok(hipMemcpy(d_msg, msg.data(), msg.size(), hipMemcpyHostToDevice)) ||
        !hip_ok(hipMemcpy(d_sig, sig.data(), sig.size(), hipMemcpyHostToDevice))) {
      std::fprintf(stderr, "device buffers: HIP error\n");
      return 9;
    }
    for (size_t parts : { (size_t)1, (size_t)2, (size_t)3 }) {
      // ranges start at multiples of 4 items so that every shard's arrays stay 4-byte aligned (fa_verify_many_dev)
      const size_t per = ((count + parts - 1) / parts + 3) & ~(size_t)3;
      std::vector<falcon::device_shard> shards;
      for (size_t lo = 0; lo < count; lo += per) {
        const size_t n = std::min(per, count - lo);
        shards.push_back(falcon::device_shard{ 0, n, d_pk + lo * falcon::pkey_len<N>(), d_msg + lo * mlen, nullptr,
                                               d_sig + lo * falcon::sig_len<N>(), d_acc + lo });
      }
      if (!hip_ok(hipMemset(d_acc, 7, count))) return 9;
      falcon::verify_many_sharded_device<N>(shards, mlen);
      std::vector<uint8_t> got(count, 9);
      if (!hip_ok(hipMemcpy(got.data(), d_acc, count, hipMemcpyDeviceToHost))) return 9;
      if (got != acc) {
        std::fprintf(stderr, "verify_many_sharded_device over %zu shards differs from verify_many\n", shards.size());
        return 9;
      }
    }
    bool threw = false;  // a device that does not exist: the error surfaces after the good device's shard has run
    try {
      std::vector<falcon::device_shard> shards{ { 0, count, d_pk, d_msg, nullptr, d_sig, d_acc }, { 63, count, d_pk, d_msg, nullptr, d_sig, d_acc } };
      falcon::verify_many_sharded_device<N>(shards, mlen);
    } catch (const falcon::device_error&) {
      threw = true;
    }
    (void)hipFree(d_pk), (void)hipFree(d_msg), (void)hipFree(d_sig), (void)hipFree(d_acc);
    if (!threw) {
      std::fprintf(stderr, "verify_many_sharded_device: missing device not reported\n");
      return 9;
    }
  }
  stage("verify_many_sharded");
  std::vector<uint8_t> acc2(count, 7);
  falcon::pinned_buffer<uint8_t> ppk(pk.size());  // keys from pinned memory, the rest pageable
  std::copy(pk.begin(), pk.end(), ppk.data());
  falcon::verify_many_sharded<N>(count, ppk.data(), msg.data(), mlen, nullptr, sig.data(), acc2.data(), ndev);
  {  // a second call runs on the per-device contexts the first one left behind
    std::vector<uint8_t> again(count, 7);
    falcon::verify_many_sharded<N>(count, ppk.data(), msg.data(), mlen, nullptr, sig.data(), again.data(), ndev);
    if (again != acc2) {
      std::fprintf(stderr, "verify_many_sharded: second call differs from the first\n");
      return 4;
    }
  }
  for (uint8_t a : acc2) std::putchar('0' + a);
  std::putchar('\n');
  stage("verify_many_sharded, several ranges on one GPU");
  {  // two ranges mapped to GPU 0 twice (and three, with an empty-ish tail): both workers go through the one shared
     // context of the device, one after the other under its mutex -- the multi-GPU path's host logic on a one-GPU box
    for (const std::vector<int>& devs : { std::vector<int>{ 0, 0 }, std::vector<int>{ 0, 0, 0 } }) {
      std::vector<uint8_t> two(count, 7);
      falcon::verify_many_sharded<N>(count, ppk.data(), msg.data(), mlen, nullptr, sig.data(), two.data(), devs);
      if (two != acc2) {
        std::fprintf(stderr, "verify_many_sharded over %zu ranges on one GPU differs\n", devs.size());
        return 5;
      }
    }
    stage("verify_many_sharded, a device that does not exist");
    bool threw = false;  // a device that does not exist: the error surfaces, the range's accept bytes are zeroed
    std::vector<uint8_t> bad(count, 7);
    try {
      falcon::verify_many_sharded<N>(count, ppk.data(), msg.data(), mlen, nullptr, sig.data(), bad.data(), std::vector<int>{ 0, 63 });
    } catch (const falcon::device_error&) {
      threw = true;
    }
    const size_t per = (count + 1) / 2;
    if (!threw || !std::equal(bad.begin(), bad.begin() + (std::ptrdiff_t)per, acc2.begin()) ||
        std::any_of(bad.begin() + (std::ptrdiff_t)per, bad.end(), [](uint8_t a) { return a != 0; })) {
      std::fprintf(stderr, "verify_many_sharded: failing range not reported / not zeroed\n");
      return 6;
    }
  }
  if (outdir) {
    stage("operators");
    std::vector<uint16_t> h(count * N, 0xAAAA), c(count * N), t(count * N), back(count * N);
    std::vector<int16_t> s2(count * N, 0x5555);
    std::vector<uint8_t> hok(count, 7), sok(count, 7), salt(count * 40, 0xEE);
    falcon::decode_pkeys<N>(count, pk.data(), h.data(), hok.data());
    falcon::decode_sigs<N>(count, sig.data(), salt.data(), s2.data(), sok.data());
    falcon::hash_to_point_many<N>(count, salt.data(), msg.data(), mlen, nullptr, c.data());
    falcon::ntt_many<N>(count, h.data(), t.data());
    falcon::intt_many<N>(count, t.data(), back.data());
    dump(outdir, "h.bin", h); dump(outdir, "hok.bin", hok); dump(outdir, "salt.bin", salt); dump(outdir, "s2.bin", s2);
    dump(outdir, "sok.bin", sok); dump(outdir, "c.bin", c); dump(outdir, "ntt.bin", t); dump(outdir, "intt.bin", back);
    // verification::verify<N, beta2>(h, ...) as its callers use it: keys kept decoded (falcon.hpp:226-229). A key whose
    // header was wrong decodes to zeros here and is then an ordinary (wrong) key.
    auto keys = falcon::prepared_keys<N>::from_decoded(count, h.data());
    std::vector<uint8_t> acc5(count, 7);
    falcon::verify_many<N>(count, keys, nullptr, msg.data(), mlen, nullptr, sig.data(), acc5.data());
    for (uint8_t a : acc5) std::putchar('0' + a);
    std::putchar('\n');
    stage("verify(h, ...), one call per item");
    // the single-call form of the same: verification::verify<N, beta2>(h, msg, mlen, sig) (verification.hpp:20-26)
    for (size_t i = 0; i < count && i < 8; i++)
      std::putchar(falcon::verify<N>(h.data() + i * N, msg.data() + i * mlen, mlen, sig.data() + i * falcon::sig_len<N>()) ? '1' : '0');
    std::putchar('\n');
    {  // its cost (stderr, information)
      const auto t0 = std::chrono::steady_clock::now();
      unsigned good = 0;
      for (int r = 0; r < 200; r++) good += falcon::verify<N>(h.data(), msg.data(), mlen, sig.data()) ? 1u : 0u;
      std::fprintf(stderr, "[host_api_check] falcon::verify<%zu>(h, ...), decoded key, one signature per call: %.1f us per call (%u of 200 true)\n", N,
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200, good);
    }
    {  // a coefficient that does not fit 14 bits takes the prepared-key route: same answer as the batch form gives
      std::vector<uint16_t> big(h.begin(), h.begin() + N);
      big[3] = (uint16_t)(big[3] % 12289u + 2u * 12289u);  // same residue mod q, 24,578 or more
      auto k1 = falcon::prepared_keys<N>::from_decoded(1, big.data());
      uint8_t want = 7;
      falcon::verify_many<N>(1, k1, nullptr, msg.data(), mlen, nullptr, sig.data(), &want);
      const bool got = falcon::verify<N>(big.data(), msg.data(), mlen, sig.data());
      if (got != (want != 0) || got != (acc5[0] != 0)) {
        std::fprintf(stderr, "verify(h, ...) with a coefficient >= 2^14 differs\n");
        return 7;
      }
    }
  }
  stage("done");
  return 0;
}

int main(int argc, char** argv)
{
  if (argc != 8 && argc != 9) {
    std::fprintf(stderr, "usage: %s logn count mlen pk.bin msg.bin sig.bin ndev [outdir]\n", argv[0]);
    return 2;
  }
  const int logn = std::atoi(argv[1]);
  const size_t count = std::strtoull(argv[2], nullptr, 10), mlen = std::strtoull(argv[3], nullptr, 10);
  auto pk = slurp(argv[4]), msg = slurp(argv[5]), sig = slurp(argv[6]);
  const int ndev = std::atoi(argv[7]);
  const char* outdir = argc == 9 ? argv[8] : nullptr;
  try {
    return logn == 9 ? run<512>(count, mlen, pk, msg, sig, ndev, outdir) : run<1024>(count, mlen, pk, msg, sig, ndev, outdir);
  } catch (const falcon::device_error& e) {
    std::fprintf(stderr, "device_error: %s\n", e.what());
    return 3;
  }
}
