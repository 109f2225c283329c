// capi.hip -- the C ABI declared in include/falcon_amd.h: contexts, workspace, the chunked
// decode -> hash -> verify pipeline and the host-buffer wrappers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/falcon_amd.h"
#include "decode_kernel.hpp"
#include "hash_kernel.hpp"
#include "verify_kernel.hpp"
#include "zq.hpp"

namespace fa {

// Small batches (a pass that does not fill the GPU) are bound by the latency of the two lane-per-item kernels --
// one item: fast decoder 42 us, hash kernel 98 us (Falcon-512) -- which do not depend on each other. This kernel
// runs both at once, in 64-thread workgroups: the first `nd` workgroups are decoder waves, the rest hash waves.
// (Two streams do the same but each cross-stream dependency costs about as much as it saves; on a full GPU the
// pair gains nothing)
template <int LOGN>
__global__ __launch_bounds__(64) void front_small_kernel(uint32_t nd, const uint8_t* __restrict__ sig, uint32_t count,
                                                         uint16_t* __restrict__ s2, uint32_t* __restrict__ sqn,
                                                         uint8_t* __restrict__ st, uint32_t* __restrict__ slow_count,
                                                         uint32_t* __restrict__ slow_list, uint32_t spec,
                                                         const uint8_t* __restrict__ msg, const uint64_t* __restrict__ msg_off,
                                                         uint32_t mlen, uint32_t nblk, uint64_t* __restrict__ hraw)
{
  constexpr uint32_t SIGLEN = LOGN == 9 ? 666 : 1280;
  if (blockIdx.x < nd)  // workgroup-uniform
    decode_fast_body<LOGN>(blockIdx.x, sig, count, s2, sqn, st, slow_count, slow_list, spec);
  else
    hash_body<64>(blockIdx.x - nd, sig, SIGLEN, msg, msg_off, mlen, nullptr, count, nblk, hraw);
}

}  // namespace fa

using namespace fa;

#define HIP_TRY(expr)                      \
  do {                                     \
    const hipError_t e_ = (expr);          \
    if (e_ != hipSuccess) return (int)e_;  \
  } while (0)

namespace {

// ---- twiddle tables (host, plain % arithmetic), indexed like ntt.hpp:62-83 ----
uint32_t powmod(uint32_t b, uint32_t e)
{
  uint32_t r = 1;
  for (b %= Q; e; e >>= 1, b = b * b % Q)
    if (e & 1) r = r * b % Q;
  return r;
}
uint32_t bitrev(uint32_t v, int bits)
{
  uint32_t r = 0;
  for (int i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}
// centred twiddle and its Shoup companion round(w * 2^32 / q), zq.hpp
Tw make_tw(uint32_t w)
{
  const int64_t wc = w > Q / 2 ? (int64_t)w - (int64_t)Q : (int64_t)w;
  const int64_t num = 2 * (wc * ((int64_t)1 << 32)) + (int64_t)Q, den = 2 * (int64_t)Q;
  int64_t wp = num / den;
  if (num % den < 0) wp--;  // floor division
  return Tw{ (int32_t)wc, (int32_t)wp };
}

struct Tables {
  std::vector<Tw> fwd, inv;    // reference order, ntt.hpp:62-83
  std::vector<Tw> fwdC, invC;  // phase-C layout [lane][TWC], verify_kernel.hpp
  NttConsts nc;
};
Tables make_tables(int logn)
{
  const uint32_t n = 1u << logn, zeta = logn == 9 ? 49u : 7u;  // ntt.hpp:18,23
  Tables t;
  t.fwd.resize(n);
  t.inv.resize(n);
  uint32_t inv1 = 1;  // canonical inverse twiddle of index 1
  for (uint32_t k = 0; k < n; k++) {
    const uint32_t w = powmod(zeta, bitrev(k, logn)), wi = powmod(w, Q - 2);
    t.fwd[k] = make_tw(w);
    t.inv[k] = make_tw(wi);
    if (k == 1) inv1 = wi;
  }
  const uint32_t ninv = powmod(n, Q - 2);  // ntt.hpp:25-29
  t.nc.ninv = make_tw(ninv);
  t.nc.inv_last = make_tw(inv1 * ninv % Q);
  for (int k = 0; k < 16; k++) {
    t.nc.twA[k] = t.fwd[k];
    t.nc.itwA[k] = t.inv[k];
  }
  // phase C: registers hold index bits [0, LB); the stage on bit lb uses, for lane L, the entries
  // (N >> (lb+1)) + (L << (LB-1-lb)) + j, j < 2^(LB-1-lb), stored consecutively in the lane's row.
  const int LB = logn - 6, PB = 6 - LB, TWC = logn == 9 ? 7 : 13;
  t.fwdC.assign(64 * TWC, Tw{ 0, 0 });
  t.invC.assign(64 * TWC, Tw{ 0, 0 });
  for (uint32_t lane = 0; lane < 64; lane++) {
    int e = 0;
    for (int lb = PB - 1; lb >= 0; lb--)
      for (uint32_t j = 0; j < (1u << (LB - 1 - lb)); j++, e++) {
        const uint32_t k = (n >> (lb + 1)) + (lane << (LB - 1 - lb)) + j;
        t.fwdC[lane * TWC + e] = t.fwd[k];
        t.invC[lane * TWC + e] = t.inv[k];
      }
  }
  return t;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes)
  {
    if (bytes <= cap) return 0;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(&p, bytes));
    cap = bytes;
    return 0;
  }
  void release()
  {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct PinBuf {  // pinned host staging memory
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes)
  {
    if (bytes <= cap) return 0;
    if (p) HIP_TRY(hipHostFree(p));
    p = nullptr;
    cap = 0;
    HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    cap = bytes;
    return 0;
  }
  void release()
  {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

bool is_pinned(const void* p)
{
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // plain pageable memory is unknown to the runtime: not an error here
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// Host-to-device copy of `bytes` on stream `xs`. Pinned caller memory goes out as one asynchronous copy.
// Pageable memory is staged through `pin` by `threads` host threads, 4 MiB pieces round-robin: a thread
// copies a piece into the pinned buffer and enqueues its DMA itself, so the host-side copying of one piece
// overlaps the DMA of the others (the runtime's own pageable path does this on a single thread and reaches
// about half the link's rate). Returns once everything is enqueued; `pin` is busy until `xs` has run.
int h2d(int device, void* dst, const void* src, size_t bytes, hipStream_t xs, PinBuf& pin, unsigned threads)
{
  constexpr size_t PIECE = (size_t)4 << 20;
  if (bytes == 0) return 0;
  if (bytes < PIECE || threads == 0 || is_pinned(src)) return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, xs);
  int rc = pin.reserve(bytes);
  if (rc) return rc;
  const size_t pieces = (bytes + PIECE - 1) / PIECE;
  const unsigned nt = (unsigned)std::min<size_t>(threads, pieces);
  std::vector<int> err(nt, 0);
  auto work = [&](unsigned t) {
    if (hipSetDevice(device) != hipSuccess) {
      err[t] = (int)hipErrorInvalidDevice;
      return;
    }
    for (size_t i = t; i < pieces && !err[t]; i += nt) {
      const size_t o = i * PIECE, len = std::min(PIECE, bytes - o);
      std::memcpy((uint8_t*)pin.p + o, (const uint8_t*)src + o, len);
      err[t] = (int)hipMemcpyAsync((uint8_t*)dst + o, (const uint8_t*)pin.p + o, len, hipMemcpyHostToDevice, xs);
    }
  };
  std::vector<std::thread> pool;
  unsigned started = 1;  // the calling thread is worker 0
  try {
    for (; started < nt; started++) pool.emplace_back(work, started);
  } catch (...) {  // no exception crosses the C ABI: the workers that did start still cover every piece below
  }
  work(0);
  for (auto& th : pool) th.join();
  for (int e : err)
    if (e) return e;
  // pieces of workers that could not be started
  for (unsigned t = started; t < nt; t++) {
    work(t);
    if (err[t]) return err[t];
  }
  return 0;
}

struct KeySrc {  // where a pass takes its public keys from
  const uint8_t* pk = nullptr;         // packed keys, one per item
  const uint16_t* prepared = nullptr;  // or prepared keys (prepare_keys_kernel) ...
  const uint32_t* key_idx = nullptr;   // ... indexed per item (nullable: key = item)
};

struct EventPair {
  hipEvent_t a, b;
  int kind;  // 0 decode, 1 hash, 2 verify, 3 polymul
};

}  // namespace

struct fa_ctx {
  int device = 0;
  int num_cu = 256;
  hipStream_t stream = nullptr;  // private stream for the host-buffer entry points
  hipStream_t aux = nullptr;     // decode kernels run here, beside the hash kernel on the caller's stream
  hipStream_t copy = nullptr;    // host-buffer entry points: H2D staging of the next sub-batch
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_ready[2] = { nullptr, nullptr };  // staging set k is filled
  uint32_t host_chunk = 1u << 17;  // items per sub-batch of the host-buffer pipeline
  int wire = FA_WIRE_REFERENCE;    // wire format of keys and signatures (FA_OPT_WIRE_FORMAT)
  DevBuf spec_pk;                  // FA_WIRE_SPEC: the batch's keys repacked into the reference's form
  bool overlap = true;
  Tw* d_tw[2] = { nullptr, nullptr };
  Tw* d_itw[2] = { nullptr, nullptr };
  Tw* d_twc[2] = { nullptr, nullptr };
  Tw* d_itwc[2] = { nullptr, nullptr };
  NttConsts nc[2];
  uint32_t chunk = 1u << 20;
  uint32_t hash_blocks[2] = { 9, 17 };
  bool profile = false;
  // pipeline workspace (per chunk)
  DevBuf hraw, s2, sqn, st2, retry_count, retry_list, slow_count, slow_list;
  // staging for the host-buffer entry points; the verify pipeline double-buffers its per-item inputs
  struct HostStage {
    DevBuf pk, sig, msg, off, kidx, accept, dbg[6];
    PinBuf pin_pk, pin_sig, pin_msg;  // pinned staging for pageable caller memory
  } hs[2];
  uint32_t host_threads = 6;  // host threads staging pageable memory (0: leave it to the runtime)
  DevBuf h_pk, h_sig, h_msg, h_off, h_a, h_b, h_c;
  std::vector<EventPair> events;
  uint32_t retried = 0;
  uint32_t small_batch = 1u << 15;  // passes of at most this many items run decoder and hash waves in one kernel (front_small_kernel)
  uint32_t grid_per_cu = 0;  // workgroups per CU of the grid-stride kernels (FA_OPT_GRID_PER_CU); 0: GRID_ROUNDS x what the GPU holds at once
  std::vector<std::pair<const void*, uint32_t>> resident;  // kernel -> workgroups of VERIFY_WG threads resident per CU
};

namespace {

struct Guard {  // set the context's device for the duration of a call
  int prev = -1;
  int err = 0;
  explicit Guard(int dev)
  {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) err = (int)hipSetDevice(dev);
  }
  ~Guard()
  {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

template <typename F>
int timed(fa_ctx* ctx, hipStream_t s, int kind, F&& launch)
{
  if (!ctx->profile) {
    launch();
    return (int)hipGetLastError();
  }
  EventPair ev{ nullptr, nullptr, kind };
  HIP_TRY(hipEventCreate(&ev.a));
  HIP_TRY(hipEventCreate(&ev.b));
  HIP_TRY(hipEventRecord(ev.a, s));
  launch();
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ev.b, s));
  ctx->events.push_back(ev);
  return 0;
}

bool aligned4(const void* p) { return ((uintptr_t)p & 3u) == 0; }

// Grid of a grid-stride kernel (one item per wave, VERIFY_WG threads per workgroup) that has `want` workgroups of
// work. Default: GRID_ROUNDS times what the GPU holds at once. Measured (tools/grid_sweep.sh, 2^20 items): waves do
// not advance at one rate, so a grid of exactly the resident workgroups ends with most of the GPU waiting for the
// slowest (Falcon-512 verify kernel 1.74 ms), and so does a small multiple of it (8 per CU: 1.64 ms, the last
// 3/8 of the items at three waves per SIMD instead of five); with 60-80 per CU (12-16 rounds, 13-17 items per
// wave) the dispatcher evens that out (1.47 ms), and beyond that the per-workgroup start (twiddle tables into LDS,
// one item's loads not hidden) costs more than it gains (480 per CU: 1.54 ms; one item per wave: 1.65 ms).
// Falcon-1024: 3.78 ms at 8 per CU, 3.48-3.52 ms at 24-48 (8-16 rounds).
constexpr uint32_t GRID_ROUNDS = 12;
template <typename K>
uint32_t stride_grid(fa_ctx* ctx, K kernel, uint32_t want)
{
  uint32_t per_cu = ctx->grid_per_cu;
  if (per_cu == 0) {  // resident workgroups per CU of this kernel, asked once
    const void* key = (const void*)kernel;
    for (const auto& e : ctx->resident)
      if (e.first == key) per_cu = e.second;
    if (per_cu == 0) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, VERIFY_WG, 0) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 8;
      }
      per_cu = (uint32_t)nb;
      try {
        ctx->resident.emplace_back(key, per_cu);
      } catch (...) {  // not cached: asked again next time
      }
    }
    per_cu *= GRID_ROUNDS;
  }
  return std::max<uint32_t>(1u, std::min<uint32_t>(want, (uint32_t)ctx->num_cu * per_cu));
}

// FA_WIRE_SPEC: repack `count` specification-format public keys into the reference's form in the context's
// buffer and return that; FA_WIRE_REFERENCE: the keys as they are.
int keys_in_reference_form(fa_ctx* ctx, int logn, hipStream_t s, const uint8_t* pk, size_t count, const uint8_t** out)
{
  *out = pk;
  if (ctx->wire != FA_WIRE_SPEC || !pk || count == 0) return 0;
  const size_t pklen = logn == 9 ? 897 : 1793;
  int rc = ctx->spec_pk.reserve(count * pklen + 8);
  if (rc) return rc;
  const uint32_t n = (uint32_t)count;
  const uint32_t grid = std::min<uint32_t>((n + 3) / 4, (uint32_t)ctx->num_cu * 16u);
  if (logn == 9)
    hipLaunchKernelGGL(pk_spec_to_ref_kernel<9>, dim3(grid), dim3(256), 0, s, pk, (uint8_t*)ctx->spec_pk.p, n);
  else
    hipLaunchKernelGGL(pk_spec_to_ref_kernel<10>, dim3(grid), dim3(256), 0, s, pk, (uint8_t*)ctx->spec_pk.p, n);
  HIP_TRY(hipGetLastError());
  *out = (const uint8_t*)ctx->spec_pk.p;
  return 0;
}

// The signature decoders over `slots` consecutive rows of `sig`, into the context's workspace (s2 tiled,
// sqn, st2): the fast decoder, then the generic one on the items it queued (those that met the k >= 8 rule).
// The fix-up pass is one lane per queued item, a few hundred latency-bound waves at most; with `fork` it runs
// on the context's second stream and the caller joins (hipStreamWaitEvent(s, ctx->ev_join)) before reading.
template <int LOGN>
int decode_stage(fa_ctx* ctx, hipStream_t s, uint32_t slots, const uint8_t* sig, bool fork)
{
  int rc;
  if ((rc = ctx->s2.reserve(s2_tile_elems<LOGN>(slots) * 2))) return rc;
  if ((rc = ctx->sqn.reserve((size_t)slots * 4))) return rc;
  if ((rc = ctx->st2.reserve((size_t)slots))) return rc;
  if ((rc = ctx->slow_count.reserve(4))) return rc;
  if ((rc = ctx->slow_list.reserve((size_t)slots * 4))) return rc;
  uint16_t* d_s2 = (uint16_t*)ctx->s2.p;
  uint32_t* d_sqn = (uint32_t*)ctx->sqn.p;
  uint8_t* d_st2 = (uint8_t*)ctx->st2.p;
  uint32_t* d_cnt = (uint32_t*)ctx->slow_count.p;
  uint32_t* d_list = (uint32_t*)ctx->slow_list.p;
  HIP_TRY(hipMemsetAsync(d_cnt, 0, 4, s));
  rc = timed(ctx, s, 0, [&] {
    hipLaunchKernelGGL(decode_fast_kernel<LOGN>, dim3((slots + 63) / 64), dim3(64), 0, s, sig, slots, d_s2, d_sqn,
                       d_st2, d_cnt, d_list, (uint32_t)(ctx->wire == FA_WIRE_SPEC));
  });
  if (rc) return rc;
  hipStream_t ds = fork ? ctx->aux : s;
  if (fork) {
    HIP_TRY(hipEventRecord(ctx->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(ds, ctx->ev_fork, 0));
  }
  rc = timed(ctx, ds, 0, [&] {
    hipLaunchKernelGGL(decode_kernel<LOGN>, dim3((slots + DECODE_WG - 1) / DECODE_WG), dim3(DECODE_WG), 0, ds, sig,
                       (const uint32_t*)d_list, slots, (const uint32_t*)d_cnt, true, d_s2, d_sqn, d_st2,
                       (uint32_t)(ctx->wire == FA_WIRE_SPEC));
  });
  if (!rc && fork) rc = (int)hipEventRecord(ctx->ev_join, ds);
  return rc;
}

// decode_stage and the hash kernel for a small pass: front_small_kernel, then the generic decoder's fix-up pass,
// all on `s` (no second stream: a cross-stream dependency costs more than the fix-up pass takes).
template <int LOGN>
int front_small(fa_ctx* ctx, hipStream_t s, uint32_t slots, const uint8_t* sig, const uint8_t* msg, const uint64_t* msg_off,
                uint32_t mlen, uint32_t nblk)
{
  int rc;
  if ((rc = ctx->slow_count.reserve(4))) return rc;
  if ((rc = ctx->slow_list.reserve((size_t)slots * 4))) return rc;
  uint16_t* d_s2 = (uint16_t*)ctx->s2.p;
  uint32_t* d_sqn = (uint32_t*)ctx->sqn.p;
  uint8_t* d_st2 = (uint8_t*)ctx->st2.p;
  uint32_t* d_cnt = (uint32_t*)ctx->slow_count.p;
  uint32_t* d_list = (uint32_t*)ctx->slow_list.p;
  const uint32_t spec = (uint32_t)(ctx->wire == FA_WIRE_SPEC), nd = (slots + 63) / 64;
  HIP_TRY(hipMemsetAsync(d_cnt, 0, 4, s));
  rc = timed(ctx, s, 1, [&] {  // booked as hash time: the hash waves are the longer half
    hipLaunchKernelGGL(front_small_kernel<LOGN>, dim3(2 * nd), dim3(64), 0, s, nd, sig, slots, d_s2, d_sqn, d_st2, d_cnt, d_list,
                       spec, msg, msg_off, mlen, nblk, (uint64_t*)ctx->hraw.p);
  });
  if (rc) return rc;
  return timed(ctx, s, 0, [&] {
    hipLaunchKernelGGL(decode_kernel<LOGN>, dim3((slots + DECODE_WG - 1) / DECODE_WG), dim3(DECODE_WG), 0, s, sig,
                       (const uint32_t*)d_list, slots, (const uint32_t*)d_cnt, true, d_s2, d_sqn, d_st2, spec);
  });
}

// One pass of the three kernels over `slots` items. idx == nullptr: slot s is item s of the arrays
// given (already offset to the chunk); else item idx[s] of the whole arrays.
template <int LOGN>
int run_pass(fa_ctx* ctx, hipStream_t s, uint32_t slots, const uint32_t* idx, const KeySrc& keys, const uint8_t* msg,
             const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig, uint32_t nblk, int policy, uint8_t* accept,
             const DebugOut& dbg, bool debug, uint32_t item_base)
{
  constexpr int T = LOGN - 9;
  constexpr uint32_t SIGLEN = LOGN == 9 ? 666 : 1280;
  if (slots == 0) return 0;
  int rc = ctx->hraw.reserve((size_t)slots * nblk * 136);
  if (rc) return rc;
  if ((rc = ctx->s2.reserve(s2_tile_elems<LOGN>(slots) * 2))) return rc;
  if ((rc = ctx->sqn.reserve((size_t)slots * 4))) return rc;
  if ((rc = ctx->st2.reserve((size_t)slots))) return rc;

  uint16_t* d_s2 = (uint16_t*)ctx->s2.p;
  uint32_t* d_sqn = (uint32_t*)ctx->sqn.p;
  uint8_t* d_st2 = (uint8_t*)ctx->st2.p;
  uint64_t* d_hraw = (uint64_t*)ctx->hraw.p;
  const bool small = !idx && slots <= ctx->small_batch;  // does not fill the GPU: latency-bound

  if (idx) {  // re-hash pass: a handful of items, the generic decoder is enough
    rc = timed(ctx, s, 0, [&] {
      hipLaunchKernelGGL(decode_kernel<LOGN>, dim3((slots + DECODE_WG - 1) / DECODE_WG), dim3(DECODE_WG), 0, s, sig, idx,
                         slots, (const uint32_t*)nullptr, false, d_s2, d_sqn, d_st2, (uint32_t)(ctx->wire == FA_WIRE_SPEC));
    });
  } else if (small) {
    rc = front_small<LOGN>(ctx, s, slots, sig, msg, msg_off, mlen, nblk);
  } else {
    rc = decode_stage<LOGN>(ctx, s, slots, sig, ctx->overlap);
  }
  if (rc) return rc;
  if (!small) {
    rc = timed(ctx, s, 1, [&] {
      hipLaunchKernelGGL(hash_kernel, dim3((slots + HASH_WG - 1) / HASH_WG), dim3(HASH_WG), 0, s, sig, SIGLEN, msg,
                         msg_off, mlen, idx, slots, nblk, d_hraw);
    });
    if (rc) return rc;
    if (!idx && ctx->overlap) HIP_TRY(hipStreamWaitEvent(s, ctx->ev_join, 0));
  }

  VerifyArgs<LOGN> A;
  A.pk = keys.pk;
  A.prepared = keys.prepared;
  A.key_idx = keys.key_idx;
  A.hraw = d_hraw;
  A.s2 = d_s2;
  A.sqn = d_sqn;
  A.st2 = d_st2;
  A.idx = idx;
  A.tw = ctx->d_tw[T];
  A.itw = ctx->d_itw[T];
  A.twc = ctx->d_twc[T];
  A.itwc = ctx->d_itwc[T];
  A.nc = ctx->nc[T];
  A.count = slots;
  A.nblk = nblk;
  A.wrap32 = policy == FA_NORM_WRAP32;
  A.accept = accept;
  A.retry_count = (uint32_t*)ctx->retry_count.p;
  A.retry_list = (uint32_t*)ctx->retry_list.p;
  A.item_base = item_base;
  A.half = ctx->wire == FA_WIRE_SPEC ? Q / 2u + 1u : Q / 2u;
  A.dbg = dbg;
  const uint32_t want = (slots + VERIFY_WAVES - 1) / VERIFY_WAVES;
  auto launch = [&](auto kernel) {
    const uint32_t grid = stride_grid(ctx, kernel, want);
    return timed(ctx, s, 2, [&] { hipLaunchKernelGGL(kernel, dim3(grid), dim3(VERIFY_WG), 0, s, A); });
  };
  if (keys.prepared) return debug ? launch(verify_kernel<LOGN, true, true>) : launch(verify_kernel<LOGN, false, true>);
  return debug ? launch(verify_kernel<LOGN, true, false>) : launch(verify_kernel<LOGN, false, false>);
}

DebugOut debug_out(const fa_debug* dbgp, bool& debug)
{
  DebugOut dbg{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  debug = false;
  if (dbgp) {
    dbg = DebugOut{ dbgp->c, dbgp->s2, dbgp->s1, (unsigned long long*)dbgp->norm_exact, dbgp->norm_wrap32, dbgp->status };
    debug = dbg.c || dbg.s2 || dbg.s1 || dbg.norm_exact || dbg.norm_wrap32 || dbg.status;
  }
  return dbg;
}

// Main passes of a batch, chunk by chunk, enqueued on `s` without synchronising. Items whose rejection
// sampling ran out of squeezed blocks are left in ctx->retry_list for verify_finish.
template <int LOGN>
int verify_enqueue(fa_ctx* ctx, size_t count, const KeySrc& keys, const uint8_t* msg, const uint64_t* msg_off, uint32_t mlen,
                   const uint8_t* sig, int policy, uint8_t* accept, const fa_debug* dbgp, hipStream_t s)
{
  using G = Geo<LOGN>;
  constexpr int T = LOGN - 9;
  constexpr size_t SIGLEN = LOGN == 9 ? 666 : 1280;
  bool debug;
  const DebugOut dbg = debug_out(dbgp, debug);
  int rc;
  if ((rc = ctx->retry_count.reserve(4))) return rc;
  if ((rc = ctx->retry_list.reserve(std::max<size_t>(count, 1) * 4))) return rc;
  HIP_TRY(hipMemsetAsync(ctx->retry_count.p, 0, 4, s));

  for (size_t c0 = 0; c0 < count; c0 += ctx->chunk) {
    const uint32_t slots = (uint32_t)std::min<size_t>(ctx->chunk, count - c0);
    DebugOut d = dbg;
    if (d.c) d.c += c0 * G::N;
    if (d.s2) d.s2 += c0 * G::N;
    if (d.s1) d.s1 += c0 * G::N;
    if (d.norm_exact) d.norm_exact += c0;
    if (d.norm_wrap32) d.norm_wrap32 += c0;
    if (d.status) d.status += c0;
    KeySrc k = keys;  // this chunk's window of the per-item arrays
    if (k.pk) k.pk += c0 * G::PKLEN;
    if (k.key_idx) k.key_idx += c0;
    rc = run_pass<LOGN>(ctx, s, slots, nullptr, k, msg_off ? msg : msg + c0 * mlen,
                        msg_off ? msg_off + c0 : nullptr, mlen, sig + c0 * SIGLEN, ctx->hash_blocks[T], policy,
                        accept + c0, d, debug, (uint32_t)c0);
    if (rc) return rc;
  }
  return 0;
}

// Synchronises `s` and re-hashes the items verify_enqueue left over (probability ~1e-25 / ~1e-13 per item
// at the default block counts) with twice as many blocks until none is left. Same arguments as the
// verify_enqueue call it completes.
template <int LOGN>
int verify_finish(fa_ctx* ctx, const KeySrc& keys, const uint8_t* msg, const uint64_t* msg_off, uint32_t mlen,
                  const uint8_t* sig, int policy, uint8_t* accept, const fa_debug* dbgp, hipStream_t s)
{
  constexpr int T = LOGN - 9;
  bool debug;
  const DebugOut dbg = debug_out(dbgp, debug);
  int rc;
  uint32_t nblk = ctx->hash_blocks[T];
  for (;;) {
    uint32_t pending = 0;
    HIP_TRY(hipMemcpyAsync(&pending, ctx->retry_count.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (pending == 0) break;
    ctx->retried += pending;
    nblk *= 2;
    DevBuf list;  // this round's list; the kernels refill ctx->retry_list
    if ((rc = list.reserve((size_t)pending * 4))) return rc;
    rc = (int)hipMemcpyAsync(list.p, ctx->retry_list.p, (size_t)pending * 4, hipMemcpyDeviceToDevice, s);
    if (!rc) rc = (int)hipMemsetAsync(ctx->retry_count.p, 0, 4, s);
    for (uint32_t r0 = 0; !rc && r0 < pending; r0 += ctx->chunk) {
      const uint32_t slots = std::min<uint32_t>(ctx->chunk, pending - r0);
      rc = run_pass<LOGN>(ctx, s, slots, (const uint32_t*)list.p + r0, keys, msg, msg_off, mlen, sig, nblk, policy, accept,
                          dbg, debug, 0u);
    }
    if (!rc) rc = (int)hipStreamSynchronize(s);
    list.release();
    if (rc) return rc;
  }
  return 0;
}

template <int LOGN>
int verify_dev(fa_ctx* ctx, size_t count, const KeySrc& keys, const uint8_t* msg, const uint64_t* msg_off, uint32_t mlen,
               const uint8_t* sig, int policy, uint8_t* accept, const fa_debug* dbgp, hipStream_t s)
{
  const int rc = verify_enqueue<LOGN>(ctx, count, keys, msg, msg_off, mlen, sig, policy, accept, dbgp, s);
  return rc ? rc : verify_finish<LOGN>(ctx, keys, msg, msg_off, mlen, sig, policy, accept, dbgp, s);
}

// hash_to_point for `count` (salt, message) pairs: SHAKE256 blocks, then rejection sampling; items that need
// more blocks than squeezed up front are redone with twice as many (same scheme as verify_dev).
template <int LOGN>
int hash_to_point_dev(fa_ctx* ctx, size_t count, const uint8_t* salt, const uint8_t* msg, const uint64_t* msg_off,
                      uint32_t mlen, uint16_t* c, hipStream_t s)
{
  constexpr int T = LOGN - 9;
  int rc;
  if ((rc = ctx->retry_count.reserve(4))) return rc;
  if ((rc = ctx->retry_list.reserve(count * 4))) return rc;
  HIP_TRY(hipMemsetAsync(ctx->retry_count.p, 0, 4, s));
  const uint8_t* rows = salt - 1;  // hash_kernel reads the salt at bytes 1..40 of a row: rows of 40 bytes starting one byte early
  auto pass = [&](uint32_t slots, const uint32_t* idx, const uint8_t* rows_, const uint8_t* msg_, const uint64_t* off_,
                  uint32_t nblk, uint16_t* c_, uint32_t item_base) -> int {
    int r = ctx->hraw.reserve((size_t)slots * nblk * 136);
    if (r) return r;
    r = timed(ctx, s, 1, [&] {
      hipLaunchKernelGGL(hash_kernel, dim3((slots + HASH_WG - 1) / HASH_WG), dim3(HASH_WG), 0, s, rows_, 40u, msg_, off_, mlen,
                         idx, slots, nblk, (uint64_t*)ctx->hraw.p);
    });
    if (r) return r;
    const uint32_t grid = stride_grid(ctx, sample_kernel<LOGN>, (slots + VERIFY_WAVES - 1) / VERIFY_WAVES);
    hipLaunchKernelGGL(sample_kernel<LOGN>, dim3(grid), dim3(VERIFY_WG), 0, s, (const uint64_t*)ctx->hraw.p, idx, slots, nblk,
                       item_base, c_, (uint32_t*)ctx->retry_count.p, (uint32_t*)ctx->retry_list.p);
    return (int)hipGetLastError();
  };
  for (size_t c0 = 0; c0 < count; c0 += ctx->chunk) {
    const uint32_t slots = (uint32_t)std::min<size_t>(ctx->chunk, count - c0);
    rc = pass(slots, nullptr, rows + c0 * 40, msg_off ? msg : msg + c0 * mlen, msg_off ? msg_off + c0 : nullptr,
              ctx->hash_blocks[T], c + (c0 << LOGN), (uint32_t)c0);
    if (rc) return rc;
  }
  uint32_t nblk = ctx->hash_blocks[T];
  for (;;) {
    uint32_t pending = 0;
    HIP_TRY(hipMemcpyAsync(&pending, ctx->retry_count.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (pending == 0) break;
    ctx->retried += pending;
    nblk *= 2;
    DevBuf list;
    if ((rc = list.reserve((size_t)pending * 4))) return rc;
    rc = (int)hipMemcpyAsync(list.p, ctx->retry_list.p, (size_t)pending * 4, hipMemcpyDeviceToDevice, s);
    if (!rc) rc = (int)hipMemsetAsync(ctx->retry_count.p, 0, 4, s);
    for (uint32_t r0 = 0; !rc && r0 < pending; r0 += ctx->chunk)
      rc = pass(std::min<uint32_t>(ctx->chunk, pending - r0), (const uint32_t*)list.p + r0, rows, msg, msg_off, nblk, c, 0u);
    if (!rc) rc = (int)hipStreamSynchronize(s);
    list.release();
    if (rc) return rc;
  }
  return 0;
}

template <int LOGN>
int polymul_dev(fa_ctx* ctx, size_t count, const uint16_t* a, const uint16_t* b, uint16_t* c, hipStream_t s)
{
  constexpr int T = LOGN - 9;
  constexpr size_t MAXP = 1u << 30;  // the kernel indexes polynomials with 32 bits
  for (size_t p0 = 0; p0 < count; p0 += MAXP) {
    const uint32_t n = (uint32_t)std::min(MAXP, count - p0);
    const uint32_t grid = stride_grid(ctx, polymul_kernel<LOGN>, (n + VERIFY_WAVES - 1) / VERIFY_WAVES);
    const size_t o = p0 << LOGN;
    const int rc = timed(ctx, s, 3, [&] {
      hipLaunchKernelGGL(polymul_kernel<LOGN>, dim3(grid), dim3(VERIFY_WG), 0, s, a + o, b + o, c + o, n, ctx->d_tw[T],
                         ctx->d_itw[T], ctx->d_twc[T], ctx->d_itwc[T], ctx->nc[T]);
    });
    if (rc) return rc;
  }
  return 0;
}

// decode_sig for a batch: the decoders chunk by chunk, then the untiling kernel.
template <int LOGN>
int decode_sigs_dev(fa_ctx* ctx, size_t count, const uint8_t* sig, uint8_t* salt, int16_t* s2, uint8_t* ok, hipStream_t s)
{
  constexpr size_t SIGLEN = LOGN == 9 ? 666 : 1280;
  for (size_t c0 = 0; c0 < count; c0 += ctx->chunk) {
    const uint32_t slots = (uint32_t)std::min<size_t>(ctx->chunk, count - c0);
    int rc = decode_stage<LOGN>(ctx, s, slots, sig + c0 * SIGLEN, false);
    if (rc) return rc;
    const uint32_t grid = std::min<uint32_t>((slots + 3) / 4, (uint32_t)ctx->num_cu * 16u);
    hipLaunchKernelGGL(untile_sig_kernel<LOGN>, dim3(grid), dim3(256), 0, s, sig + c0 * SIGLEN, (const uint16_t*)ctx->s2.p,
                       (const uint8_t*)ctx->st2.p, slots, salt + c0 * 40, s2 + (c0 << LOGN), ok + c0);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

}  // namespace

extern "C" {

size_t fa_pkey_len(int logn) { return logn == 9 ? 897 : logn == 10 ? 1793 : 0; }
size_t fa_sig_len(int logn) { return logn == 9 ? 666 : logn == 10 ? 1280 : 0; }

const char* fa_strerror(int code)
{
  switch (code) {
    case 0: return "ok";
    case FA_E_ARG: return "invalid argument";
    case FA_E_LOGN: return "logn must be 9 (Falcon-512) or 10 (Falcon-1024)";
    case FA_E_ALIGN: return "device arrays must be 4-byte aligned";
    case FA_E_INTERNAL: return "internal error";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

int fa_ctx_create(int device, fa_ctx** out)
{
  if (!out) return FA_E_ARG;
  *out = nullptr;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return FA_E_ARG;
  Guard g(device);
  if (g.err) return g.err;
  fa_ctx* ctx = new (std::nothrow) fa_ctx;
  if (!ctx) return FA_E_INTERNAL;
  ctx->device = device;
  hipDeviceProp_t prop;
  int rc = (int)hipGetDeviceProperties(&prop, device);
  if (!rc) {
    ctx->num_cu = prop.multiProcessorCount;
    rc = (int)hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (!rc) rc = (int)hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking);
    if (!rc) rc = (int)hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking);
    for (int k = 0; k < 2 && !rc; k++) rc = (int)hipEventCreateWithFlags(&ctx->ev_ready[k], hipEventDisableTiming);
    if (!rc) rc = (int)hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming);
    if (!rc) rc = (int)hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming);
  }
  for (int t = 0; t < 2 && !rc; t++) try {
    const Tables tb = make_tables(9 + t);
    const size_t bytes = tb.fwd.size() * sizeof(Tw);
    ctx->nc[t] = tb.nc;
    rc = (int)hipMalloc((void**)&ctx->d_tw[t], bytes);
    if (!rc) rc = (int)hipMalloc((void**)&ctx->d_itw[t], bytes);
    if (!rc) rc = (int)hipMemcpy(ctx->d_tw[t], tb.fwd.data(), bytes, hipMemcpyHostToDevice);
    if (!rc) rc = (int)hipMemcpy(ctx->d_itw[t], tb.inv.data(), bytes, hipMemcpyHostToDevice);
    const size_t cbytes = tb.fwdC.size() * sizeof(Tw);
    if (!rc) rc = (int)hipMalloc((void**)&ctx->d_twc[t], cbytes);
    if (!rc) rc = (int)hipMalloc((void**)&ctx->d_itwc[t], cbytes);
    if (!rc) rc = (int)hipMemcpy(ctx->d_twc[t], tb.fwdC.data(), cbytes, hipMemcpyHostToDevice);
    if (!rc) rc = (int)hipMemcpy(ctx->d_itwc[t], tb.invC.data(), cbytes, hipMemcpyHostToDevice);
  } catch (...) {  // std::bad_alloc while building the tables
    rc = FA_E_INTERNAL;
  }
  if (rc) {
    fa_ctx_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return 0;
}

void fa_ctx_destroy(fa_ctx* ctx)
{
  if (!ctx) return;
  Guard g(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
  if (ctx->copy) (void)hipStreamSynchronize(ctx->copy);
  for (auto& e : ctx->events) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (int t = 0; t < 2; t++) {
    if (ctx->d_tw[t]) (void)hipFree(ctx->d_tw[t]);
    if (ctx->d_itw[t]) (void)hipFree(ctx->d_itw[t]);
    if (ctx->d_twc[t]) (void)hipFree(ctx->d_twc[t]);
    if (ctx->d_itwc[t]) (void)hipFree(ctx->d_itwc[t]);
  }
  DevBuf* bufs[] = { &ctx->hraw, &ctx->s2, &ctx->sqn, &ctx->st2, &ctx->retry_count, &ctx->retry_list, &ctx->slow_count, &ctx->slow_list, &ctx->h_pk,
                     &ctx->h_sig, &ctx->h_msg, &ctx->h_off, &ctx->h_a, &ctx->h_b, &ctx->h_c, &ctx->spec_pk };
  for (DevBuf* b : bufs) b->release();
  for (auto& st : ctx->hs) {
    DevBuf* sb[] = { &st.pk, &st.sig, &st.msg, &st.off, &st.kidx, &st.accept };
    for (DevBuf* b : sb) b->release();
    for (DevBuf& b : st.dbg) b.release();
    st.pin_pk.release();
    st.pin_sig.release();
    st.pin_msg.release();
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
  if (ctx->copy) (void)hipStreamDestroy(ctx->copy);
  for (hipEvent_t e : ctx->ev_ready)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  delete ctx;
}

int fa_set_option(fa_ctx* ctx, int option, int64_t value)
{
  if (!ctx) return FA_E_ARG;
  switch (option) {
    case FA_OPT_CHUNK_ITEMS:
      if (value < 64 || value > (1 << 24)) return FA_E_ARG;
      ctx->chunk = (uint32_t)value;
      return 0;
    case FA_OPT_HASH_BLOCKS_512:
    case FA_OPT_HASH_BLOCKS_1024:
      if (value < 1 || value > 64) return FA_E_ARG;
      ctx->hash_blocks[option - FA_OPT_HASH_BLOCKS_512] = (uint32_t)value;
      return 0;
    case FA_OPT_PROFILE:
      ctx->profile = value != 0;
      return 0;
    case FA_OPT_OVERLAP:
      ctx->overlap = value != 0;
      return 0;
    case FA_OPT_HOST_CHUNK_ITEMS:
      if (value < 64 || value > (1 << 24)) return FA_E_ARG;
      ctx->host_chunk = (uint32_t)value;
      return 0;
    case FA_OPT_WIRE_FORMAT:
      if (value != FA_WIRE_REFERENCE && value != FA_WIRE_SPEC) return FA_E_ARG;
      ctx->wire = (int)value;
      return 0;
    case FA_OPT_HOST_THREADS:
      if (value < 0 || value > 64) return FA_E_ARG;
      ctx->host_threads = (uint32_t)value;
      return 0;
    case FA_OPT_SMALL_BATCH_ITEMS:
      if (value < 0 || value > (1 << 20)) return FA_E_ARG;
      ctx->small_batch = (uint32_t)value;
      return 0;
    case FA_OPT_GRID_PER_CU:
      if (value < 0 || value > 65536) return FA_E_ARG;
      ctx->grid_per_cu = (uint32_t)value;
      return 0;
    default:
      return FA_E_ARG;
  }
}

int fa_get_profile(fa_ctx* ctx, fa_profile* out)
{
  if (!ctx || !out) return FA_E_ARG;
  Guard g(ctx->device);
  std::memset(out, 0, sizeof(*out));
  for (auto& e : ctx->events) {
    HIP_TRY(hipEventSynchronize(e.b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
    double* acc[] = { &out->ms_decode, &out->ms_hash, &out->ms_verify, &out->ms_polymul };
    uint32_t* cnt[] = { &out->launches_decode, &out->launches_hash, &out->launches_verify, &out->launches_polymul };
    *acc[e.kind] += ms;
    *cnt[e.kind] += 1;
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  ctx->events.clear();
  out->retried_items = ctx->retried;
  ctx->retried = 0;
  return 0;
}

int fa_verify_many_dev(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, const uint8_t* msg,
                       const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig, int norm_policy, uint8_t* accept,
                       const fa_debug* dbg, void* stream)
{
  if (!ctx || (norm_policy != FA_NORM_EXACT64 && norm_policy != FA_NORM_WRAP32)) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!pk || !sig || !accept || (!msg && (msg_off || mlen))) return FA_E_ARG;
  if (count > 0xFFFFFFFFull) return FA_E_ARG;
  if (!aligned4(pk) || !aligned4(sig) || !aligned4(msg) || ((uintptr_t)msg_off & 7u)) return FA_E_ALIGN;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  KeySrc keys;
  const int rc = keys_in_reference_form(ctx, logn, s, pk, count, &keys.pk);
  if (rc) return rc;
  return logn == 9 ? verify_dev<9>(ctx, count, keys, msg, msg_off, mlen, sig, norm_policy, accept, dbg, s)
                   : verify_dev<10>(ctx, count, keys, msg, msg_off, mlen, sig, norm_policy, accept, dbg, s);
}

int fa_verify_many_prepared_dev(fa_ctx* ctx, int logn, size_t count, const uint16_t* prepared, const uint32_t* key_idx,
                                const uint8_t* msg, const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig,
                                int norm_policy, uint8_t* accept, const fa_debug* dbg, void* stream)
{
  if (!ctx || (norm_policy != FA_NORM_EXACT64 && norm_policy != FA_NORM_WRAP32)) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!prepared || !sig || !accept || (!msg && (msg_off || mlen))) return FA_E_ARG;
  if (count > 0xFFFFFFFFull) return FA_E_ARG;
  if (!aligned4(prepared) || !aligned4(key_idx) || !aligned4(sig) || !aligned4(msg) || ((uintptr_t)msg_off & 7u)) return FA_E_ALIGN;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  KeySrc keys;
  keys.prepared = prepared;
  keys.key_idx = key_idx;
  return logn == 9 ? verify_dev<9>(ctx, count, keys, msg, msg_off, mlen, sig, norm_policy, accept, dbg, s)
                   : verify_dev<10>(ctx, count, keys, msg, msg_off, mlen, sig, norm_policy, accept, dbg, s);
}

size_t fa_prepared_key_len(int logn) { return logn == 9 || logn == 10 ? (size_t)2 << logn : 0; }

int fa_prepare_keys_dev(fa_ctx* ctx, int logn, size_t nkeys, const uint8_t* pk, const uint16_t* h, uint16_t* prepared,
                        void* stream)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (nkeys == 0) return 0;
  if (!prepared || (!pk) == (!h) || nkeys > 0xFFFFFFFFull) return FA_E_ARG;  // exactly one of pk, h
  if (!aligned4(prepared) || ((uintptr_t)h & 1u)) return FA_E_ALIGN;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  const int T = logn - 9;
  const uint32_t n = (uint32_t)nkeys, want = (n + VERIFY_WAVES - 1) / VERIFY_WAVES;
  {
    const int rc = keys_in_reference_form(ctx, logn, s, pk, nkeys, &pk);
    if (rc) return rc;
  }
#define FA_LAUNCH_PREP(LN, PACKED)                                                                                      \
  hipLaunchKernelGGL((prepare_keys_kernel<LN, PACKED>), dim3(stride_grid(ctx, prepare_keys_kernel<LN, PACKED>, want)),   \
                     dim3(VERIFY_WG), 0, s, pk, h, n, prepared, ctx->d_tw[T], ctx->d_itw[T], ctx->d_twc[T],              \
                     ctx->d_itwc[T], ctx->nc[T])
  if (logn == 9) {
    if (pk) FA_LAUNCH_PREP(9, true); else FA_LAUNCH_PREP(9, false);
  } else {
    if (pk) FA_LAUNCH_PREP(10, true); else FA_LAUNCH_PREP(10, false);
  }
#undef FA_LAUNCH_PREP
  return (int)hipGetLastError();
}

int fa_prepare_keys(fa_ctx* ctx, int logn, size_t nkeys, const uint8_t* pk, const uint16_t* h, uint16_t* prepared)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (nkeys == 0) return 0;
  if (!prepared || (!pk) == (!h)) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  const size_t in_bytes = nkeys * (pk ? fa_pkey_len(logn) : fa_prepared_key_len(logn));
  const size_t out_bytes = nkeys * fa_prepared_key_len(logn);
  hipStream_t s = ctx->stream;
  int rc;
  if ((rc = ctx->h_pk.reserve(in_bytes + 8))) return rc;
  if ((rc = ctx->h_c.reserve(out_bytes))) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->h_pk.p, pk ? (const void*)pk : (const void*)h, in_bytes, hipMemcpyHostToDevice, s));
  rc = fa_prepare_keys_dev(ctx, logn, nkeys, pk ? (const uint8_t*)ctx->h_pk.p : nullptr, pk ? nullptr : (const uint16_t*)ctx->h_pk.p,
                           (uint16_t*)ctx->h_c.p, s);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(prepared, ctx->h_c.p, out_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

// Host-buffer front end shared by fa_verify_many and fa_verify_many_prepared. The batch is cut into
// sub-batches of ctx->host_chunk items with two sets of device staging buffers: while the kernels of
// sub-batch k run on ctx->stream, the inputs of sub-batch k+1 cross the host link on ctx->copy. With
// pageable caller memory the copy calls block the host thread (the runtime stages them) but still overlap
// the kernels already enqueued; with pinned memory (fa_host_alloc / fa_host_register) they are asynchronous
// and the whole call runs at the rate of the host link.
static int verify_host(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, const uint16_t* prepared, size_t nkeys,
                       const uint32_t* key_idx, const uint8_t* msg, const uint64_t* msg_off, uint32_t mlen,
                       const uint8_t* sig, int norm_policy, uint8_t* accept, const fa_debug* dbg)
{
  if (!ctx || (norm_policy != FA_NORM_EXACT64 && norm_policy != FA_NORM_WRAP32)) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if ((!pk && !prepared) || !sig || !accept || (!msg && (msg_off || mlen))) return FA_E_ARG;
  if (count > 0xFFFFFFFFull) return FA_E_ARG;
  if (prepared) {  // every item must name an existing key
    if (nkeys == 0 || nkeys > 0xFFFFFFFFull || (!key_idx && nkeys < count)) return FA_E_ARG;
    if (key_idx)
      for (size_t i = 0; i < count; i++)
        if (key_idx[i] >= nkeys) return FA_E_ARG;
  }
  if (msg_off)
    for (size_t i = 0; i < count; i++)
      if (msg_off[i + 1] < msg_off[i]) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  const size_t n = (size_t)1 << logn, pklen = fa_pkey_len(logn), siglen = fa_sig_len(logn);
  hipStream_t cs = ctx->stream, xs = ctx->copy;
  int rc;

  const uint16_t* d_prepared = nullptr;
  if (prepared) {  // the key table is shared by all sub-batches: staged once, ahead of the first kernel
    if ((rc = ctx->h_pk.reserve(nkeys * n * 2 + 8))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->h_pk.p, prepared, nkeys * n * 2, hipMemcpyHostToDevice, cs));
    d_prepared = (const uint16_t*)ctx->h_pk.p;
  }

  struct Sub {  // one sub-batch: its window of the caller's arrays and where it sits on the device
    size_t c0 = 0, cnt = 0;
    KeySrc keys;
    const uint8_t* d_msg = nullptr;
    const uint64_t* d_off = nullptr;
    const uint8_t* d_sig = nullptr;
    uint8_t* d_accept = nullptr;
    fa_debug d{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    size_t dsz[6] = { 0, 0, 0, 0, 0, 0 };
  };

  // H2D staging of sub-batch k on the copy stream. +8 / +12 bytes of slack: the kernels read whole aligned
  // dwords around the last record.
  auto stage_in = [&](size_t k, Sub& S) -> int {
    fa_ctx::HostStage& st = ctx->hs[k & 1];
    S = Sub{};
    S.c0 = k * (size_t)ctx->host_chunk;
    S.cnt = std::min<size_t>(ctx->host_chunk, count - S.c0);
    int r;
    if ((r = st.sig.reserve(S.cnt * siglen + 8))) return r;
    if ((r = st.accept.reserve(S.cnt))) return r;
    if ((r = h2d(ctx->device, st.sig.p, sig + S.c0 * siglen, S.cnt * siglen, xs, st.pin_sig, ctx->host_threads))) return r;
    S.d_sig = (const uint8_t*)st.sig.p;
    S.d_accept = (uint8_t*)st.accept.p;
    if (prepared) {
      S.keys.prepared = key_idx ? d_prepared : d_prepared + S.c0 * n;  // no index: item i uses key i
      if (key_idx) {
        if ((r = st.kidx.reserve(S.cnt * 4))) return r;
        HIP_TRY(hipMemcpyAsync(st.kidx.p, key_idx + S.c0, S.cnt * 4, hipMemcpyHostToDevice, xs));
        S.keys.key_idx = (const uint32_t*)st.kidx.p;
      }
    } else {
      if ((r = st.pk.reserve(S.cnt * pklen + 8))) return r;
      if ((r = h2d(ctx->device, st.pk.p, pk + S.c0 * pklen, S.cnt * pklen, xs, st.pin_pk, ctx->host_threads))) return r;
      S.keys.pk = (const uint8_t*)st.pk.p;
    }
    // With offsets only the window [msg_off[c0], msg_off[c0 + cnt]) is staged; it lands at byte (first & 3)
    // of the staging buffer so that the device-side base pointer `stage - (first & ~3)` stays 4-byte
    // aligned, and message i is found at d_msg + msg_off[i] with the caller's own offsets.
    const size_t first = msg_off ? (size_t)msg_off[S.c0] : 0;
    const size_t msg_bytes = msg_off ? (size_t)msg_off[S.c0 + S.cnt] - first : S.cnt * (size_t)mlen;
    if ((r = st.msg.reserve(msg_bytes + 12))) return r;
    uint8_t* stage = (uint8_t*)st.msg.p + (first & 3u);
    if ((r = h2d(ctx->device, stage, msg_off ? msg + first : msg + S.c0 * (size_t)mlen, msg_bytes, xs, st.pin_msg, ctx->host_threads)))
      return r;
    S.d_msg = stage - first;
    if (msg_off) {
      if ((r = st.off.reserve((S.cnt + 1) * 8))) return r;
      HIP_TRY(hipMemcpyAsync(st.off.p, msg_off + S.c0, (S.cnt + 1) * 8, hipMemcpyHostToDevice, xs));
      S.d_off = (const uint64_t*)st.off.p;
    }
    if (dbg) {
      void* const host[6] = { dbg->c, dbg->s2, dbg->s1, dbg->norm_exact, dbg->norm_wrap32, dbg->status };
      void** dev[6] = { (void**)&S.d.c, (void**)&S.d.s2, (void**)&S.d.s1, (void**)&S.d.norm_exact, (void**)&S.d.norm_wrap32,
                        (void**)&S.d.status };
      const size_t sz[6] = { S.cnt * n * 2, S.cnt * n * 2, S.cnt * n * 2, S.cnt * 8, S.cnt * 4, S.cnt };
      for (int i = 0; i < 6; i++) {
        if (!host[i]) continue;
        if ((r = st.dbg[i].reserve(sz[i]))) return r;
        *dev[i] = st.dbg[i].p;
        S.dsz[i] = sz[i];
      }
    }
    HIP_TRY(hipEventRecord(ctx->ev_ready[k & 1], xs));
    return 0;
  };
  auto launch = [&](size_t k, Sub& S) -> int {
    HIP_TRY(hipStreamWaitEvent(cs, ctx->ev_ready[k & 1], 0));
    if (S.keys.pk) {  // FA_WIRE_SPEC: this sub-batch's keys, repacked on the compute stream (one buffer: sub-batches run in turn)
      const int r = keys_in_reference_form(ctx, logn, cs, S.keys.pk, S.cnt, &S.keys.pk);
      if (r) return r;
    }
    const fa_debug* dp = dbg ? &S.d : nullptr;
    return logn == 9 ? verify_enqueue<9>(ctx, S.cnt, S.keys, S.d_msg, S.d_off, mlen, S.d_sig, norm_policy, S.d_accept, dp, cs)
                     : verify_enqueue<10>(ctx, S.cnt, S.keys, S.d_msg, S.d_off, mlen, S.d_sig, norm_policy, S.d_accept, dp, cs);
  };
  auto finish = [&](const Sub& S) -> int {
    const fa_debug* dp = dbg ? &S.d : nullptr;
    int r = logn == 9 ? verify_finish<9>(ctx, S.keys, S.d_msg, S.d_off, mlen, S.d_sig, norm_policy, S.d_accept, dp, cs)
                      : verify_finish<10>(ctx, S.keys, S.d_msg, S.d_off, mlen, S.d_sig, norm_policy, S.d_accept, dp, cs);
    if (r) return r;
    HIP_TRY(hipMemcpyAsync(accept + S.c0, S.d_accept, S.cnt, hipMemcpyDeviceToHost, cs));
    if (dbg) {
      uint8_t* const host[6] = { (uint8_t*)dbg->c, (uint8_t*)dbg->s2, (uint8_t*)dbg->s1, (uint8_t*)dbg->norm_exact,
                                 (uint8_t*)dbg->norm_wrap32, (uint8_t*)dbg->status };
      void* const dev[6] = { S.d.c, S.d.s2, S.d.s1, S.d.norm_exact, S.d.norm_wrap32, S.d.status };
      const size_t per_item[6] = { n * 2, n * 2, n * 2, 8, 4, 1 };
      for (int i = 0; i < 6; i++)
        if (host[i]) HIP_TRY(hipMemcpyAsync(host[i] + S.c0 * per_item[i], dev[i], S.dsz[i], hipMemcpyDeviceToHost, cs));
    }
    return 0;
  };

  // Order per sub-batch k: kernels of k, then the copies of k+1 (they overlap those kernels), then the
  // synchronising tail of k. finish(k) has synchronised ctx->stream before staging set k & 1 is refilled for
  // sub-batch k+2, so the sets need no further events.
  const size_t nsub = (count + ctx->host_chunk - 1) / ctx->host_chunk;
  Sub cur, nxt;
  rc = stage_in(0, cur);
  for (size_t k = 0; !rc && k < nsub; k++) {
    rc = launch(k, cur);
    if (!rc && k + 1 < nsub) rc = stage_in(k + 1, nxt);
    if (!rc) rc = finish(cur);
    cur = nxt;
  }
  const int rs = (int)hipStreamSynchronize(xs), rt = (int)hipStreamSynchronize(cs);  // nothing of the caller's is in flight on return
  return rc ? rc : rs ? rs : rt;
}

void* fa_host_alloc(size_t bytes)
{
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void fa_host_free(void* p)
{
  if (p) (void)hipHostFree(p);
}
int fa_host_register(void* p, size_t bytes)
{
  if (!p || bytes == 0) return FA_E_ARG;
  const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) (void)hipGetLastError();  // reported here; must not linger as the "last error" of a later launch
  return (int)e;
}
int fa_host_unregister(void* p)
{
  if (!p) return FA_E_ARG;
  const hipError_t e = hipHostUnregister(p);
  if (e != hipSuccess) (void)hipGetLastError();
  return (int)e;
}

int fa_verify_many(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, const uint8_t* msg, const uint64_t* msg_off,
                   uint32_t mlen, const uint8_t* sig, int norm_policy, uint8_t* accept, const fa_debug* dbg)
{
  if (!pk && count) return FA_E_ARG;
  return verify_host(ctx, logn, count, pk, nullptr, 0, nullptr, msg, msg_off, mlen, sig, norm_policy, accept, dbg);
}

int fa_verify_many_prepared(fa_ctx* ctx, int logn, size_t count, const uint16_t* prepared, size_t nkeys,
                            const uint32_t* key_idx, const uint8_t* msg, const uint64_t* msg_off, uint32_t mlen,
                            const uint8_t* sig, int norm_policy, uint8_t* accept, const fa_debug* dbg)
{
  if (!prepared && count) return FA_E_ARG;
  return verify_host(ctx, logn, count, nullptr, prepared, nkeys, key_idx, msg, msg_off, mlen, sig, norm_policy, accept, dbg);
}

int fa_verify_mixed(fa_ctx* ctx, size_t count, const uint8_t* logn, const uint8_t* pk, const uint64_t* pk_off,
                    const uint8_t* msg, const uint64_t* msg_off, const uint8_t* sig, const uint64_t* sig_off,
                    int norm_policy, uint8_t* accept)
{
  if (!ctx) return FA_E_ARG;
  if (count == 0) return 0;
  if (!logn || !pk || !pk_off || !msg_off || !sig || !sig_off || !accept) return FA_E_ARG;
  for (size_t i = 0; i < count; i++)
    if (logn[i] != 9 && logn[i] != 10) return FA_E_LOGN;
  try {
  for (int ln = 9; ln <= 10; ln++) {  // partition by N on the host, one pass per class
    const size_t pklen = fa_pkey_len(ln), siglen = fa_sig_len(ln);
    std::vector<size_t> who;
    for (size_t i = 0; i < count; i++)
      if (logn[i] == ln) who.push_back(i);
    if (who.empty()) continue;
    const size_t nc = who.size();
    std::vector<uint64_t> goff(nc + 1, 0);
    for (size_t k = 0; k < nc; k++) {
      if (msg_off[who[k] + 1] < msg_off[who[k]]) return FA_E_ARG;
      goff[k + 1] = goff[k] + (msg_off[who[k] + 1] - msg_off[who[k]]);
    }
    if (goff[nc] && !msg) return FA_E_ARG;
    std::vector<uint8_t> gpk(nc * pklen), gsig(nc * siglen), gmsg(std::max<uint64_t>(goff[nc], 1)), gacc(nc);
    // the class's records, gathered into contiguous arrays by the context's host threads (contiguous ranges of items)
    auto gather = [&](size_t k0, size_t k1) {
      for (size_t k = k0; k < k1; k++) {
        const size_t i = who[k];
        std::memcpy(&gpk[k * pklen], pk + pk_off[i], pklen);
        std::memcpy(&gsig[k * siglen], sig + sig_off[i], siglen);
        const size_t ml = (size_t)(goff[k + 1] - goff[k]);
        if (ml) std::memcpy(&gmsg[goff[k]], msg + msg_off[i], ml);
      }
    };
    const size_t nt = std::max<size_t>(1, std::min<size_t>(ctx->host_threads, nc / 4096));
    std::vector<std::thread> pool;
    size_t started = 1;  // the calling thread takes range 0
    try {
      for (; started < nt; started++) pool.emplace_back(gather, nc * started / nt, nc * (started + 1) / nt);
    } catch (...) {  // threads that could not be started: their ranges are done below
    }
    gather(0, nc / nt);
    for (auto& th : pool) th.join();
    for (size_t t = started; t < nt; t++) gather(nc * t / nt, nc * (t + 1) / nt);
    const int rc = fa_verify_many(ctx, ln, nc, gpk.data(), gmsg.data(), goff.data(), 0, gsig.data(), norm_policy,
                                  gacc.data(), nullptr);
    if (rc) return rc;
    for (size_t k = 0; k < nc; k++) accept[who[k]] = gacc[k];
  }
  } catch (...) {  // std::bad_alloc of the host-side partition: no exception crosses the C ABI
    return FA_E_INTERNAL;
  }
  return 0;
}

int fa_hash_to_point_many_dev(fa_ctx* ctx, int logn, size_t count, const uint8_t* salt, const uint8_t* msg,
                              const uint64_t* msg_off, uint32_t mlen, uint16_t* c, void* stream)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!salt || !c || (!msg && (msg_off || mlen)) || count > 0xFFFFFFFFull) return FA_E_ARG;
  if (!aligned4(salt) || !aligned4(msg) || !aligned4(c) || ((uintptr_t)msg_off & 7u)) return FA_E_ALIGN;
  Guard g(ctx->device);
  if (g.err) return g.err;
  return logn == 9 ? hash_to_point_dev<9>(ctx, count, salt, msg, msg_off, mlen, c, (hipStream_t)stream)
                   : hash_to_point_dev<10>(ctx, count, salt, msg, msg_off, mlen, c, (hipStream_t)stream);
}

int fa_hash_to_point_many(fa_ctx* ctx, int logn, size_t count, const uint8_t* salt, const uint8_t* msg,
                          const uint64_t* msg_off, uint32_t mlen, uint16_t* c)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!salt || !c || (!msg && (msg_off || mlen)) || count > 0xFFFFFFFFull) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  const size_t first = msg_off ? (size_t)msg_off[0] : 0;
  if (msg_off && (size_t)msg_off[count] < first) return FA_E_ARG;
  const size_t msg_bytes = msg_off ? (size_t)msg_off[count] - first : count * (size_t)mlen;
  const size_t out_bytes = (count << logn) * 2;
  hipStream_t s = ctx->stream;
  int rc;
  if ((rc = ctx->h_sig.reserve(count * 40 + 16))) return rc;
  if ((rc = ctx->h_msg.reserve(msg_bytes + 12))) return rc;
  if ((rc = ctx->h_c.reserve(out_bytes))) return rc;
  uint8_t* d_salt = (uint8_t*)ctx->h_sig.p + 4;  // the kernel's row pointer is salt - 1
  HIP_TRY(hipMemcpyAsync(d_salt, salt, count * 40, hipMemcpyHostToDevice, s));
  uint8_t* d_msg_stage = (uint8_t*)ctx->h_msg.p + (first & 3u);
  if (msg_bytes) HIP_TRY(hipMemcpyAsync(d_msg_stage, msg + first, msg_bytes, hipMemcpyHostToDevice, s));
  const uint64_t* d_off = nullptr;
  if (msg_off) {
    if ((rc = ctx->h_off.reserve((count + 1) * 8))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->h_off.p, msg_off, (count + 1) * 8, hipMemcpyHostToDevice, s));
    d_off = (const uint64_t*)ctx->h_off.p;
  }
  rc = fa_hash_to_point_many_dev(ctx, logn, count, d_salt, d_msg_stage - first, d_off, mlen, (uint16_t*)ctx->h_c.p, s);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c, ctx->h_c.p, out_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int fa_compute_public_keys_dev(fa_ctx* ctx, int logn, size_t count, const int32_t* f, const int32_t* g, uint16_t* h,
                               void* stream)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!f || !g || !h || count > 0xFFFFFFFFull) return FA_E_ARG;
  if (!aligned4(f) || !aligned4(g) || ((uintptr_t)h & 1u)) return FA_E_ALIGN;
  Guard gd(ctx->device);
  if (gd.err) return gd.err;
  hipStream_t s = (hipStream_t)stream;
  const int T = logn - 9;
  const uint32_t n = (uint32_t)count, want = (n + VERIFY_WAVES - 1) / VERIFY_WAVES;
  if (logn == 9)
    hipLaunchKernelGGL(pubkey_kernel<9>, dim3(stride_grid(ctx, pubkey_kernel<9>, want)), dim3(VERIFY_WG), 0, s, f, g, h, n,
                       ctx->d_tw[T], ctx->d_itw[T], ctx->d_twc[T], ctx->d_itwc[T], ctx->nc[T]);
  else
    hipLaunchKernelGGL(pubkey_kernel<10>, dim3(stride_grid(ctx, pubkey_kernel<10>, want)), dim3(VERIFY_WG), 0, s, f, g, h, n,
                       ctx->d_tw[T], ctx->d_itw[T], ctx->d_twc[T], ctx->d_itwc[T], ctx->nc[T]);
  return (int)hipGetLastError();
}

int fa_compute_public_keys(fa_ctx* ctx, int logn, size_t count, const int32_t* f, const int32_t* g, uint16_t* h)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!f || !g || !h) return FA_E_ARG;
  Guard gd(ctx->device);
  if (gd.err) return gd.err;
  const size_t in_bytes = (count << logn) * 4, out_bytes = (count << logn) * 2;
  int rc;
  if ((rc = ctx->h_a.reserve(in_bytes))) return rc;
  if ((rc = ctx->h_b.reserve(in_bytes))) return rc;
  if ((rc = ctx->h_c.reserve(out_bytes))) return rc;
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->h_a.p, f, in_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->h_b.p, g, in_bytes, hipMemcpyHostToDevice, s));
  rc = fa_compute_public_keys_dev(ctx, logn, count, (const int32_t*)ctx->h_a.p, (const int32_t*)ctx->h_b.p,
                                  (uint16_t*)ctx->h_c.p, s);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h, ctx->h_c.p, out_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int fa_ntt_many_dev(fa_ctx* ctx, int logn, size_t count, const uint16_t* a, uint16_t* out, int inverse, void* stream)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!a || !out || count > 0xFFFFFFFFull) return FA_E_ARG;
  if (((uintptr_t)a & 1u) || ((uintptr_t)out & 1u)) return FA_E_ALIGN;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  const int T = logn - 9;
  const uint32_t n = (uint32_t)count, want = (n + VERIFY_WAVES - 1) / VERIFY_WAVES;
#define FA_LAUNCH_NTT(LN, INV)                                                                                          \
  hipLaunchKernelGGL((ntt_kernel<LN, INV>), dim3(stride_grid(ctx, ntt_kernel<LN, INV>, want)), dim3(VERIFY_WG), 0, s,   \
                     a, out, n, ctx->d_tw[T], ctx->d_itw[T], ctx->d_twc[T], ctx->d_itwc[T], ctx->nc[T])
  if (logn == 9) {
    if (inverse) FA_LAUNCH_NTT(9, true); else FA_LAUNCH_NTT(9, false);
  } else {
    if (inverse) FA_LAUNCH_NTT(10, true); else FA_LAUNCH_NTT(10, false);
  }
#undef FA_LAUNCH_NTT
  return (int)hipGetLastError();
}

int fa_ntt_many(fa_ctx* ctx, int logn, size_t count, const uint16_t* a, uint16_t* out, int inverse)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!a || !out) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  const size_t bytes = (count << logn) * 2;
  int rc;
  if ((rc = ctx->h_a.reserve(bytes))) return rc;
  if ((rc = ctx->h_c.reserve(bytes))) return rc;
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->h_a.p, a, bytes, hipMemcpyHostToDevice, s));
  rc = fa_ntt_many_dev(ctx, logn, count, (const uint16_t*)ctx->h_a.p, (uint16_t*)ctx->h_c.p, inverse, s);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, ctx->h_c.p, bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int fa_decode_pkeys_dev(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, uint16_t* h, uint8_t* ok, void* stream)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!pk || !h || !ok || count > 0xFFFFFFFFull) return FA_E_ARG;
  if ((uintptr_t)h & 1u) return FA_E_ALIGN;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n = (uint32_t)count, want = (n + VERIFY_WAVES - 1) / VERIFY_WAVES;
  {
    const int rc = keys_in_reference_form(ctx, logn, s, pk, count, &pk);
    if (rc) return rc;
  }
  if (logn == 9)
    hipLaunchKernelGGL(decode_pkeys_kernel<9>, dim3(stride_grid(ctx, decode_pkeys_kernel<9>, want)), dim3(VERIFY_WG), 0, s, pk, n, h, ok);
  else
    hipLaunchKernelGGL(decode_pkeys_kernel<10>, dim3(stride_grid(ctx, decode_pkeys_kernel<10>, want)), dim3(VERIFY_WG), 0, s, pk, n, h, ok);
  return (int)hipGetLastError();
}

int fa_decode_pkeys(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, uint16_t* h, uint8_t* ok)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!pk || !h || !ok) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  const size_t in_bytes = count * fa_pkey_len(logn), out_bytes = (count << logn) * 2;
  int rc;
  if ((rc = ctx->h_pk.reserve(in_bytes + 8))) return rc;
  if ((rc = ctx->h_c.reserve(out_bytes))) return rc;
  if ((rc = ctx->h_off.reserve(count))) return rc;
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->h_pk.p, pk, in_bytes, hipMemcpyHostToDevice, s));
  rc = fa_decode_pkeys_dev(ctx, logn, count, (const uint8_t*)ctx->h_pk.p, (uint16_t*)ctx->h_c.p, (uint8_t*)ctx->h_off.p, s);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h, ctx->h_c.p, out_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(ok, ctx->h_off.p, count, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int fa_decode_sigs_dev(fa_ctx* ctx, int logn, size_t count, const uint8_t* sig, uint8_t* salt, int16_t* s2, uint8_t* ok,
                       void* stream)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!sig || !salt || !s2 || !ok || count > 0xFFFFFFFFull) return FA_E_ARG;
  if (!aligned4(sig) || ((uintptr_t)s2 & 1u)) return FA_E_ALIGN;
  Guard g(ctx->device);
  if (g.err) return g.err;
  return logn == 9 ? decode_sigs_dev<9>(ctx, count, sig, salt, s2, ok, (hipStream_t)stream)
                   : decode_sigs_dev<10>(ctx, count, sig, salt, s2, ok, (hipStream_t)stream);
}

int fa_decode_sigs(fa_ctx* ctx, int logn, size_t count, const uint8_t* sig, uint8_t* salt, int16_t* s2, uint8_t* ok)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!sig || !salt || !s2 || !ok) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  const size_t in_bytes = count * fa_sig_len(logn), s2_bytes = (count << logn) * 2;
  int rc;
  if ((rc = ctx->h_sig.reserve(in_bytes + 16))) return rc;
  if ((rc = ctx->h_c.reserve(s2_bytes))) return rc;
  if ((rc = ctx->h_msg.reserve(count * 40))) return rc;
  if ((rc = ctx->h_off.reserve(count))) return rc;
  hipStream_t s = ctx->stream;
  // the caller's salt / s2 rows come along, so that rows the reference leaves untouched keep their contents
  HIP_TRY(hipMemcpyAsync(ctx->h_sig.p, sig, in_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->h_c.p, s2, s2_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->h_msg.p, salt, count * 40, hipMemcpyHostToDevice, s));
  rc = fa_decode_sigs_dev(ctx, logn, count, (const uint8_t*)ctx->h_sig.p, (uint8_t*)ctx->h_msg.p, (int16_t*)ctx->h_c.p,
                          (uint8_t*)ctx->h_off.p, s);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(s2, ctx->h_c.p, s2_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(salt, ctx->h_msg.p, count * 40, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(ok, ctx->h_off.p, count, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int fa_polymul_many_dev(fa_ctx* ctx, int logn, size_t count, const uint16_t* a, const uint16_t* b, uint16_t* c,
                        void* stream)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!a || !b || !c) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  return logn == 9 ? polymul_dev<9>(ctx, count, a, b, c, (hipStream_t)stream)
                   : polymul_dev<10>(ctx, count, a, b, c, (hipStream_t)stream);
}

int fa_polymul_many(fa_ctx* ctx, int logn, size_t count, const uint16_t* a, const uint16_t* b, uint16_t* c)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!a || !b || !c) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  const size_t bytes = (count << logn) * 2;
  int rc;
  if ((rc = ctx->h_a.reserve(bytes))) return rc;
  if ((rc = ctx->h_b.reserve(bytes))) return rc;
  if ((rc = ctx->h_c.reserve(bytes))) return rc;
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->h_a.p, a, bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->h_b.p, b, bytes, hipMemcpyHostToDevice, s));
  rc = fa_polymul_many_dev(ctx, logn, count, (const uint16_t*)ctx->h_a.p, (const uint16_t*)ctx->h_b.p,
                           (uint16_t*)ctx->h_c.p, s);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c, ctx->h_c.p, bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

}  // extern "C"
