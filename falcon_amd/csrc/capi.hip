// capi.hip -- the C ABI declared in include/falcon_amd.h: contexts, workspace, the chunked
// decode -> hash -> verify pipeline and the host-buffer wrappers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/falcon_amd.h"
#include "decode_kernel.hpp"
#include "encode_kernel.hpp"
#include "hash_kernel.hpp"
#include "tiny_kernel.hpp"
#include "verify_kernel.hpp"
#include "zq.hpp"

namespace fa {

// Small batches (a pass that does not fill the GPU) are bound by the latency of the two lane-per-item kernels --
// one item: fast decoder 42 us, hash kernel 98 us (Falcon-512) -- which do not depend on each other. This kernel
// runs both at once, in 64-thread workgroups: the first `nd` workgroups are decoder waves, the rest hash waves.
// (Two streams do the same but each cross-stream dependency costs about as much as it saves; on a full GPU the
// pair gains nothing)
template <int LOGN>
__global__ __launch_bounds__(64) void front_small_kernel(uint32_t nd, const Rows sig, const uint32_t* __restrict__ idx, uint32_t count,
                                                         const uint32_t* __restrict__ dcount,
                                                         uint16_t* __restrict__ s2, uint32_t* __restrict__ sqn,
                                                         uint8_t* __restrict__ st, uint32_t* __restrict__ slow_count,
                                                         uint32_t* __restrict__ slow_list, uint32_t spec,
                                                         const uint8_t* __restrict__ msg, const uint64_t* __restrict__ msg_off,
                                                         uint32_t mlen, uint32_t nblk, uint64_t* __restrict__ hraw, uint32_t pair)
{
  if (blockIdx.x < nd)  // workgroup-uniform
    decode_fast_body<LOGN>(blockIdx.x, sig, idx, count, dcount, s2, sqn, st, slow_count, slow_list, spec);
  else if (pair)  // kernel argument: uniform. 32 items per hash wave, one sponge per lane pair (hash_kernel.hpp)
    hash_pair_body(blockIdx.x - nd, sig, msg, msg_off, mlen, idx, count, dcount, nblk, hraw);
  else
    hash_body<64>(blockIdx.x - nd, sig, msg, msg_off, mlen, idx, count, dcount, nblk, hraw);
}

// A handful of items (FA_OPT_TINY_BATCH_ITEMS): one decoder WAVE and one hash WAVE per item (tiny_kernel.hpp), the two
// waves of a 128-thread workgroup -- so that the decoder's LDS tables (21 / 41 KB) are counted once per item, not also
// for a separate hash workgroup, and twice as many items fit the GPU at once. No fix-up queue: a signature the
// wave-parallel decoder cannot take is decoded by the generic decoder inside the same wave.
// `split` (passes of at most 128 items, which leave most CUs idle anyway): decoder and sponge in workgroups of their
// own, 64 threads each, the first `split` workgroups decoding -- on different CUs the two do not share an LDS, which a
// lone Falcon-1024 item notices (82 against 87 us).
template <int LOGN>
__global__ __launch_bounds__(128) void front_tiny_kernel(const Rows sig, const uint32_t* __restrict__ idx, uint32_t split,
                                                         uint16_t* __restrict__ s2, uint32_t* __restrict__ sqn, uint8_t* __restrict__ st,
                                                         uint32_t spec, const uint8_t* __restrict__ msg,
                                                         const uint64_t* __restrict__ msg_off, uint32_t mlen, uint32_t nblk,
                                                         uint64_t* __restrict__ hraw)
{
  const bool decoder = split ? blockIdx.x < split : threadIdx.x < 64u;  // wave-uniform
  const uint32_t slot = split && !decoder ? blockIdx.x - split : blockIdx.x;
  if (decoder)
    decode_wave_body<LOGN>(slot, sig, idx, s2, sqn, st, spec);
  else
    hash_wave_body(slot, sig, msg, msg_off, mlen, idx, nblk, hraw);
}

// The device-side retry pass (retry_tail) in ONE launch: a workgroup of three waves per queued item -- wave 0 decodes
// the signature, wave 1 squeezes the (doubled) SHAKE256 blocks, both with an item spread over the lanes of a wave
// (tiny_kernel.hpp), and behind a barrier wave 2 runs the verify kernel's own item code on what they left in the
// workspace. The pass is launched behind every asynchronous and allocation-free call and almost always finds its
// queue empty (*A.dcount == 0): one launch that exits at once instead of three. (As the front of ordinary small calls
// the same arrangement gains nothing over front_tiny_kernel + verify_kernel)
struct TinyFront {  // what waves 0 and 1 need beyond VerifyArgs: the inputs, and the workspace as writable pointers
  Rows sig;
  const uint8_t* msg;
  const uint64_t* msg_off;
  uint32_t mlen, spec;
  uint16_t* s2;
  uint32_t* sqn;
  uint8_t* st;
  uint64_t* hraw;
};
template <int LOGN, bool DEBUG, bool PREPARED>
__global__ __launch_bounds__(192) void retry_item_kernel(const VerifyArgs<LOGN> A, const TinyFront F)
{
  using G = Geo<LOGN>;
  __shared__ TwLds<LOGN> s_tw;
  __shared__ uint32_t s_buf[G::XBUF];
  __shared__ uint16_t s_c[G::N];
  const uint32_t slot = blockIdx.x;
  const uint32_t count = A.dcount ? min(A.count, *A.dcount) : A.count;
  if (slot >= count) return;  // workgroup-uniform, ahead of every barrier
  fill_tw_lds<LOGN>(s_tw, A.tw, A.itw, A.twc, A.isc);  // all three waves; ends in a workgroup barrier
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (wave == 0)  // wave-uniform
    decode_wave_body<LOGN>(slot, F.sig, A.idx, F.s2, F.sqn, F.st, F.spec);
  else if (wave == 1)
    hash_wave_body(slot, F.sig, F.msg, F.msg_off, F.mlen, A.idx, A.nblk, F.hraw);
  // Rows of neighbouring slots share cache lines (s2 tiles, hraw rows), and another workgroup on this CU may have pulled
  // such a line into the vector cache before this one's writes: the writers release to L2, the reader invalidates.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (wave == 2) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    verify_items<LOGN, DEBUG, PREPARED, true>(A, make_tabs<LOGN>(s_tw, A.nc, lane), s_buf, s_c, lane, slot, gridDim.x, count);
  }
}

// The two halves of front_tiny_kernel on their own, for the batched operators (fa_hash_to_point_many, fa_decode_sigs)
// when they are called with a handful of items.
__global__ __launch_bounds__(64) void hash_tiny_kernel(const Rows sig, const uint32_t* __restrict__ idx, const uint8_t* __restrict__ msg,
                                                       const uint64_t* __restrict__ msg_off, uint32_t mlen, uint32_t nblk,
                                                       uint64_t* __restrict__ hraw)
{
  hash_wave_body(blockIdx.x, sig, msg, msg_off, mlen, idx, nblk, hraw);
}
template <int LOGN>
__global__ __launch_bounds__(64) void decode_tiny_kernel(const Rows sig, uint16_t* __restrict__ s2, uint32_t* __restrict__ sqn,
                                                         uint8_t* __restrict__ st, uint32_t spec)
{
  decode_wave_body<LOGN>(blockIdx.x, sig, nullptr, s2, sqn, st, spec);
}

// ---- mixed batches (fa_verify_mixed_dev): order-preserving split of the item indices by N ----
// logn[i] in {9, 10}. Pass 1 counts the Falcon-512 items of every run of PART_ITEMS items, pass 2 (one
// workgroup) turns the counts into exclusive prefix sums and leaves the totals, pass 3 writes the item
// indices: class 9 to list[0, n9), class 10 to list[n9, count), each in ascending order (so a pass over a
// class list walks the records in memory order).
constexpr uint32_t PART_WG = 256, PART_ITEMS = 1024;

__global__ __launch_bounds__(PART_WG) void partition_count_kernel(const uint8_t* __restrict__ logn, uint32_t count,
                                                                  uint32_t* __restrict__ run_n9, uint32_t* __restrict__ totals)
{
  __shared__ uint32_t s_n9, s_bad;
  if (threadIdx.x == 0) s_n9 = s_bad = 0u;
  __syncthreads();
  uint32_t n9 = 0, bad = 0;
  for (uint32_t r = 0; r < PART_ITEMS / PART_WG; r++) {
    const uint32_t i = blockIdx.x * PART_ITEMS + r * PART_WG + threadIdx.x;
    if (i < count) {
      const uint32_t t = logn[i];
      n9 += t == 9u;
      bad += t != 9u && t != 10u;
    }
  }
  if (n9) atomicAdd(&s_n9, n9);
  if (bad) atomicAdd(&s_bad, bad);
  __syncthreads();
  if (threadIdx.x == 0) {
    run_n9[blockIdx.x] = s_n9;
    if (s_bad) atomicAdd(&totals[1], s_bad);
  }
}

__global__ __launch_bounds__(PART_WG) void partition_scan_kernel(uint32_t* __restrict__ run_n9, uint32_t nruns,
                                                                 uint32_t* __restrict__ totals)
{
  __shared__ uint32_t s_part[PART_WG];
  // thread t owns the contiguous range of runs [t * per, (t + 1) * per)
  const uint32_t per = (nruns + PART_WG - 1) / PART_WG;
  const uint32_t lo = min(threadIdx.x * per, nruns), hi = min(lo + per, nruns);
  uint32_t sum = 0;
  for (uint32_t r = lo; r < hi; r++) sum += run_n9[r];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t t = 0; t < PART_WG; t++) {
      const uint32_t v = s_part[t];
      s_part[t] = acc;
      acc += v;
    }
    totals[0] = acc;  // n9
  }
  __syncthreads();
  uint32_t acc = s_part[threadIdx.x];
  for (uint32_t r = lo; r < hi; r++) {
    const uint32_t v = run_n9[r];
    run_n9[r] = acc;
    acc += v;
  }
}

__global__ __launch_bounds__(PART_WG) void partition_write_kernel(const uint8_t* __restrict__ logn, uint32_t count,
                                                                  const uint32_t* __restrict__ run_n9,
                                                                  const uint32_t* __restrict__ totals,
                                                                  uint32_t* __restrict__ list)
{
  __shared__ uint32_t s_wave9[PART_WG / 64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t base9 = run_n9[blockIdx.x];                                   // class-9 items before this run
  uint32_t base10 = totals[0] + (blockIdx.x * PART_ITEMS - base9);       // class-10 items go behind all class-9 ones
  for (uint32_t r = 0; r < PART_ITEMS / PART_WG; r++) {
    const uint32_t i = blockIdx.x * PART_ITEMS + r * PART_WG + threadIdx.x;
    const bool in = i < count;
    const bool is9 = in && logn[i] == 9u;
    const unsigned long long m9 = __ballot(is9), min_ = __ballot(in);
    if (lane == 0) s_wave9[wave] = (uint32_t)__popcll(m9);
    __syncthreads();
    uint32_t before9 = 0, round9 = 0;
    for (uint32_t w = 0; w < PART_WG / 64; w++) {
      if (w < wave) before9 += s_wave9[w];
      round9 += s_wave9[w];
    }
    const uint32_t r9 = before9 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m9 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m9, 0u));
    const uint32_t rin = wave * 64u + __builtin_amdgcn_mbcnt_hi((uint32_t)(min_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)min_, 0u));
    if (in) {
      if (is9) list[base9 + r9] = i;
      else list[base10 + (rin - r9)] = i;
    }
    const uint32_t live = min(PART_WG, count > blockIdx.x * PART_ITEMS + r * PART_WG ? count - (blockIdx.x * PART_ITEMS + r * PART_WG) : 0u);
    base9 += round9;
    base10 += live - round9;
    __syncthreads();
  }
}

// End of the device-side retry pass (asynchronous and allocation-free entry points): this call's number of items
// that could not be settled on the device -- those beyond the retry pass's capacity plus those that ran out of
// blocks a second time -- is left in ctl[CTL_UNRESOLVED] and added to the context's running total.
__global__ void retry_account_kernel(uint32_t* __restrict__ ctl, uint32_t tail_cap, uint32_t* __restrict__ total)
{
  uint32_t u = ctl[2];
  const uint32_t r = ctl[0];
  if (r > tail_cap) u += r - tail_cap;
  ctl[2] = u;
  if (total && u) __hip_atomic_fetch_add(total, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // host memory (fa_ctx::h_ctl)
}

}  // namespace fa

using namespace fa;

// A failing runtime call is reported through the return value; it must not linger as the runtime's "last error",
// where the hipGetLastError() behind a later, good launch would find it.
#define HIP_TRY(expr)                      \
  do {                                     \
    const hipError_t e_ = (expr);          \
    if (e_ != hipSuccess) {                \
      (void)hipGetLastError();             \
      return (int)e_;                      \
    }                                      \
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
struct Tables {
  std::vector<Tw> fwd;   // forward twiddles zeta^bitrev(k), reference order (ntt.hpp:62-83), Plantard form (zq.hpp)
  std::vector<Tw> fwdC;  // the same in the phase-C layout [lane][TWC], verify_kernel.hpp
  std::vector<Tw> inv;   // inverse (decimation in time): entry 2^b + j = omega^-(N/2^(b+1) * j), omega = zeta^2
  std::vector<Tw> isc;   // inverse scaling: entry p = zeta^-p / N (ntt.hpp:25-29 folded in)
  std::vector<Tw> isc_pos;  // the same times 2^32: the packed-key verify kernel, whose pointwise product is -x*y*2^-32 (zq.hpp plant2_wide)
  std::vector<Tw> isc_neg;  // the same times -2^32: polymul_kernel, same product, wanted with the other sign
  NttConsts nc;
};
Tables make_tables(int logn)
{
  const uint32_t n = 1u << logn, zeta = logn == 9 ? 49u : 7u;  // ntt.hpp:18,23
  Tables t;
  t.fwd.resize(n);
  for (uint32_t k = 0; k < n; k++) t.fwd[k] = plant_const(powmod(zeta, bitrev(k, logn)));
  const uint32_t zinv = powmod(zeta, Q - 2), oinv = zinv * zinv % Q, ninv = powmod(n, Q - 2);
  t.inv.assign(n, 0u);
  for (int b = 0; b < logn; b++)
    for (uint32_t j = 0; j < (1u << b); j++) t.inv[(1u << b) + j] = plant_const(powmod(oinv, (n >> (b + 1)) * j));
  t.isc.resize(n);
  t.isc_pos.resize(n);
  t.isc_neg.resize(n);
  for (uint32_t p = 0; p < n; p++) {
    const uint32_t sc = powmod(zinv, p) * ninv % Q;
    t.isc[p] = plant_const(sc);
    t.isc_pos[p] = plant_const((uint32_t)((uint64_t)sc * POS_R % Q));
    t.isc_neg[p] = plant_const((uint32_t)((uint64_t)sc * NEG_R % Q));
  }
  for (int k = 0; k < 16; k++) t.nc.twA[k] = t.fwd[k];
  for (int k = 0; k < 8; k++) t.nc.itwA[k] = t.inv[k];
  // phase C: registers hold index bits [0, LB); the stage on bit lb uses, for lane L, the entries
  // (N >> (lb+1)) + (L << (LB-1-lb)) + j, j < 2^(LB-1-lb), stored consecutively in the lane's row.
  const int LB = logn - 6, PB = 6 - LB, TWC = logn == 9 ? 7 : 13;
  t.fwdC.assign(64 * TWC, 0u);
  for (uint32_t lane = 0; lane < 64; lane++) {
    int e = 0;
    for (int lb = PB - 1; lb >= 0; lb--)
      for (uint32_t j = 0; j < (1u << (LB - 1 - lb)); j++, e++)
        t.fwdC[lane * TWC + e] = t.fwd[(n >> (lb + 1)) + (lane << (LB - 1 - lb)) + j];
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
  const unsigned nt = (unsigned)std::min<size_t>(std::min(threads, 64u), pieces);
  int err[64] = {};  // per worker; FA_OPT_HOST_THREADS is at most 64
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
    pool.reserve(nt);
    for (; started < nt; started++) pool.emplace_back(work, started);
  } catch (...) {  // no exception crosses the C ABI: the workers that did start still cover every piece below
  }
  work(0);
  for (auto& th : pool) th.join();
  for (unsigned t = 0; t < started; t++)
    if (err[t]) return err[t];
  // pieces of workers that could not be started
  for (unsigned t = started; t < nt; t++) {
    work(t);
    if (err[t]) return err[t];
  }
  return 0;
}

struct KeySrc {  // where a pass takes its public keys from
  Rows pk{ nullptr, nullptr, 0 };      // packed keys, one per item
  const uint16_t* prepared = nullptr;  // or prepared keys (prepare_keys_kernel) ...
  const uint32_t* key_idx = nullptr;   // ... indexed per item (nullable: key = item)
};

// The caller's arrays of one verify call (device pointers).
struct Batch {
  KeySrc keys;
  Rows sig{ nullptr, nullptr, 0 };
  const uint8_t* msg = nullptr;
  const uint64_t* msg_off = nullptr;
  uint32_t mlen = 0;
  int policy = FA_NORM_EXACT64;
  uint8_t* accept = nullptr;
  DebugOut dbg{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  bool debug = false;
};

struct EventPair {
  hipEvent_t a, b;
  int kind;  // 0 decode, 1 hash, 2 verify, 3 polymul
};

// Device memory of the pipeline, either the context's own buffers (grown on demand) or a
// carve-up of the library's static scratch array (fa_verify_many_dev_noalloc).
enum { CTL_RETRY = 0, CTL_SLOW_UNUSED = 1, CTL_UNRESOLVED = 2, CTL_WORDS = 64 };
struct Workspace {
  uint64_t* hraw = nullptr;  // squeezed SHAKE256 blocks
  size_t hraw_bytes = 0;
  uint16_t* s2 = nullptr;    // tiled, cap_slots rows
  uint32_t* sqn = nullptr;
  uint8_t* st2 = nullptr;
  uint32_t cap_slots = 0;
  uint32_t* ctl = nullptr;         // CTL_WORDS counters (256 bytes), zeroed once per call: [CTL_RETRY], [CTL_UNRESOLVED]
  // Number of items queued for a re-hash pass. Calls that end in a device-side retry pass count in ctl[CTL_RETRY]
  // (the pass reads it as its size). Calls that synchronise count in a word of pinned host memory the GPU maps
  // (host_counter = the host's address of it): the host zeroes it with a store and reads it after the stream
  // synchronisation it makes anyway -- no memset and no device-to-host copy per call, which a one-item call notices.
  uint32_t* retry_counter = nullptr;
  volatile uint32_t* host_counter = nullptr;
  uint32_t* slow_count = nullptr;  // its own 256-byte block, zeroed before every fast-decoder launch
  uint32_t* retry_list = nullptr;  // items whose hash needs more blocks; as many entries as the call has items
  uint32_t* slow_list = nullptr;   // cap_slots entries
  uint32_t* retry2_list = nullptr; // items that failed the device-side retry pass too; tail_cap() entries
  // slots the device-side retry pass (twice the blocks per item, in the same hraw buffer) can take
  uint32_t tail_cap() const { return std::min<uint32_t>(cap_slots / 2u, 4096u); }
};

}  // namespace

constexpr int FA_MAX_DEVICES = 64;

struct fa_ctx {
  int device = 0;
  int num_cu = 256;
  hipStream_t stream = nullptr;  // private stream for the host-buffer entry points
  hipStream_t aux = nullptr;     // decode kernels run here, beside the hash kernel on the caller's stream
  hipStream_t copy = nullptr;    // host-buffer entry points: H2D staging of the next sub-batch
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_ready[2] = { nullptr, nullptr };  // staging set k is filled
  hipEvent_t ev_ws = nullptr;    // recorded behind the last kernel that touches the workspace; every entry point waits for it first
  bool ws_busy = false;          // ev_ws has been recorded at least once
  uint32_t host_chunk = 1u << 17;  // items per sub-batch of the host-buffer pipeline
  int wire = FA_WIRE_REFERENCE;    // wire format of keys and signatures (FA_OPT_WIRE_FORMAT)
  DevBuf spec_pk;                  // FA_WIRE_SPEC: the batch's keys repacked into the reference's form
  bool overlap = true;
  bool async = false;              // FA_OPT_ASYNC: device-resident verify calls return without synchronising
  Tw* d_tw[2] = { nullptr, nullptr };   // the four tables of make_tables, per N
  Tw* d_itw[2] = { nullptr, nullptr };
  Tw* d_twc[2] = { nullptr, nullptr };
  Tw* d_isc[2] = { nullptr, nullptr };
  Tw* d_isc_pos[2] = { nullptr, nullptr };
  Tw* d_isc_neg[2] = { nullptr, nullptr };
  NttConsts nc[2];
  uint32_t chunk = 1u << 20;
  uint32_t hash_blocks[2] = { 9, 17 };
  bool profile = false;
  bool quiet = false;  // inside an allocation-free call: no profiling events, nothing that allocates
  // pipeline workspace (per chunk)
  DevBuf hraw, s2, sqn, st2, ctl, retry_list, slow_list, retry2_list;
  DevBuf part_list, part_runs;  // mixed batches: the item indices split by N, per-run counts + totals
  uint32_t* d_unresolved = nullptr;  // running total of unresolved items of asynchronous calls: word 16 of the GPU-mapped host block h_ctl, as the GPU sees it
  uint32_t* h_word = nullptr;        // pinned host words for small device-to-host reads
  uint32_t* h_ctl = nullptr;         // pinned, GPU-mapped host word: the re-hash counter of synchronising calls (Workspace::host_counter)
  uint32_t* d_hctl = nullptr;        // the GPU's address of it
  unsigned char* scratch = nullptr;  // this device's copy of the library's static scratch array (hipGetSymbolAddress)
  // staging for the host-buffer entry points; the verify pipeline double-buffers its per-item inputs
  struct HostStage {
    DevBuf pk, sig, msg, off, kidx, accept, dbg[6];
    PinBuf pin_pk, pin_sig, pin_msg;  // pinned staging for pageable caller memory
  } hs[2];
  PinBuf zc;  // tiny host-buffer calls: inputs and accept bytes in pinned host memory the kernels read and write in place
  uint32_t host_threads = 6;  // host threads staging pageable memory (0: leave it to the runtime)
  DevBuf h_pk, h_sig, h_msg, h_off, h_a, h_b, h_c, h_off2, h_off3, h_tag;
  std::vector<EventPair> events;
  std::vector<hipEvent_t> ev_free;  // events ready for reuse (no hipEventCreate inside a timed loop)
  uint32_t retried = 0;
  uint32_t small_batch = 1u << 15;  // passes of at most this many items run decoder and hash waves in one kernel (front_small_kernel)
  uint32_t tiny_batch = 1024;       // of those, passes of at most this many Falcon-512 items (half as many Falcon-1024 items: twice the LDS) run one wave per item (FA_OPT_TINY_BATCH_ITEMS)
  uint32_t tiny_limit(int logn) const { return logn == 9 ? tiny_batch : tiny_batch / 2u; }
  uint32_t pair_hash = 1u << 14;    // of those, passes of at most this many items hash with one sponge per lane pair (FA_OPT_PAIR_HASH_ITEMS)
  uint32_t grid_per_cu = 0;  // workgroups per CU of the grid-stride kernels (FA_OPT_GRID_PER_CU); 0: GRID_ROUNDS x what the GPU holds at once
  struct Resident {
    const void* kernel;
    uint32_t per_cu;
  } resident[32] = {};  // kernel -> workgroups of VERIFY_WG threads resident per CU (asked once per kernel)
};

// A set of public keys kept on one GPU in prepared form (fa_keyset_create): what fa_verify_many_keyset and the
// device-resident prepared-key calls read their keys from.
struct fa_keyset {
  int device = 0;
  int logn = 0;
  size_t nkeys = 0;
  uint16_t* d_prepared = nullptr;  // nkeys * N uint16 (prepare_keys_kernel), hipMalloc'ed on `device`
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

// Launches `launch` on `s`; with FA_OPT_PROFILE it is bracketed by two HIP events (taken from the context's pool,
// so a timed loop creates none after its first pass) whose elapsed time fa_get_profile adds up per kind.
template <typename F>
int timed(fa_ctx* ctx, hipStream_t s, int kind, F&& launch)
{
  if (!ctx->profile || ctx->quiet) {
    launch();
    return (int)hipGetLastError();
  }
  EventPair ev{ nullptr, nullptr, kind };
  auto take = [&](hipEvent_t* e) -> int {
    if (!ctx->ev_free.empty()) {
      *e = ctx->ev_free.back();
      ctx->ev_free.pop_back();
      return 0;
    }
    return (int)hipEventCreate(e);
  };
  auto drop = [&]() {
    if (ev.a) (void)hipEventDestroy(ev.a);
    if (ev.b) (void)hipEventDestroy(ev.b);
  };
  int rc = take(&ev.a);
  if (!rc) rc = take(&ev.b);
  if (!rc) rc = (int)hipEventRecord(ev.a, s);
  if (rc) {
    drop();
    return rc;
  }
  launch();
  rc = (int)hipGetLastError();
  if (!rc) rc = (int)hipEventRecord(ev.b, s);
  if (!rc) try {
      ctx->events.push_back(ev);
      return 0;
    } catch (...) {
      rc = FA_E_INTERNAL;
    }
  drop();
  return rc;
}

bool aligned4(const void* p) { return ((uintptr_t)p & 3u) == 0; }

// Every device-side entry point that uses the context's workspace (or the static scratch array) brackets its work
// with these: the workspace is one set of buffers, so a call enqueued on stream B must not start overwriting it
// while the kernels of an earlier call on stream A still read it.
int ws_begin(fa_ctx* ctx, hipStream_t s)
{
  if (ctx->ws_busy) HIP_TRY(hipStreamWaitEvent(s, ctx->ev_ws, 0));
  return 0;
}
int ws_end(fa_ctx* ctx, hipStream_t s)
{
  HIP_TRY(hipEventRecord(ctx->ev_ws, s));
  ctx->ws_busy = true;
  return 0;
}
// The pair as a scope: whatever way a call leaves after begin() -- an allocation that fails, a bad tag array, a launch
// error -- the event is recorded behind what the call did enqueue, so that the next call on another stream waits for
// it (an early return between ws_begin and ws_end used to leave the fence at the previous call's last kernel).
struct WsScope {
  fa_ctx* ctx;
  hipStream_t s;
  bool open = false;
  WsScope(fa_ctx* c, hipStream_t st) : ctx(c), s(st) {}
  WsScope(const WsScope&) = delete;
  WsScope& operator=(const WsScope&) = delete;
  int begin()
  {
    const int rc = ws_begin(ctx, s);
    open = rc == 0;
    return rc;
  }
  int end()
  {
    if (!open) return 0;
    open = false;
    return ws_end(ctx, s);
  }
  ~WsScope()
  {
    if (open) (void)ws_end(ctx, s);
  }
};

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
uint32_t stride_grid(fa_ctx* ctx, K kernel, uint32_t want, int block = VERIFY_WG)
{
  uint32_t per_cu = ctx->grid_per_cu;
  if (per_cu == 0) {  // resident workgroups per CU of this kernel, asked once
    const void* key = (const void*)kernel;
    fa_ctx::Resident* free_slot = nullptr;
    for (auto& e : ctx->resident) {
      if (e.kernel == key) per_cu = e.per_cu;
      if (!e.kernel && !free_slot) free_slot = &e;
    }
    if (per_cu == 0) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 8;
      }
      per_cu = (uint32_t)nb;
      if (free_slot) *free_slot = fa_ctx::Resident{ key, per_cu };  // table full: asked again next time
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

constexpr size_t align256(size_t v) { return (v + 255u) & ~(size_t)255u; }

// The context's own workspace for passes of up to `slots` items with `nblk` blocks per item, in a call of `items`
// items. Buffers only ever grow.
template <int LOGN>
int ws_from_ctx(fa_ctx* ctx, uint32_t slots, uint32_t nblk, size_t items, Workspace& ws)
{
  const uint32_t cap = std::max<uint32_t>(slots, 128u);  // room for a retry pass of at least 64 items
  int rc;
  if ((rc = ctx->hraw.reserve((size_t)cap * nblk * 136))) return rc;
  if ((rc = ctx->s2.reserve(s2_tile_elems<LOGN>(cap) * 2))) return rc;
  if ((rc = ctx->sqn.reserve((size_t)cap * 4))) return rc;
  if ((rc = ctx->st2.reserve((size_t)cap))) return rc;
  if ((rc = ctx->ctl.reserve(512))) return rc;
  if ((rc = ctx->retry_list.reserve(std::max<size_t>(items, 1) * 4))) return rc;
  if ((rc = ctx->slow_list.reserve((size_t)cap * 4))) return rc;
  if ((rc = ctx->retry2_list.reserve(4096 * 4))) return rc;
  ws.hraw = (uint64_t*)ctx->hraw.p;
  ws.hraw_bytes = ctx->hraw.cap;
  ws.s2 = (uint16_t*)ctx->s2.p;
  ws.sqn = (uint32_t*)ctx->sqn.p;
  ws.st2 = (uint8_t*)ctx->st2.p;
  // the buffers may be larger than asked (an earlier, larger call): the capacity is what all of them hold
  ws.cap_slots = (uint32_t)std::min<size_t>({ ctx->hraw.cap / ((size_t)nblk * 136), ctx->s2.cap / ((size_t)2 << LOGN) / 64 * 64,
                                               ctx->sqn.cap / 4, ctx->st2.cap, ctx->slow_list.cap / 4, (size_t)1 << 30 });
  ws.ctl = (uint32_t*)ctx->ctl.p;
  ws.slow_count = ws.ctl + CTL_WORDS;
  if (ctx->async || ctx->quiet) {
    ws.retry_counter = ws.ctl + CTL_RETRY;
    ws.host_counter = nullptr;
  } else {
    ws.retry_counter = ctx->d_hctl;
    ws.host_counter = ctx->h_ctl;
  }
  ws.retry_list = (uint32_t*)ctx->retry_list.p;
  ws.slow_list = (uint32_t*)ctx->slow_list.p;
  ws.retry2_list = (uint32_t*)ctx->retry2_list.p;
  return 0;
}

// The signature decoders over `slots` workspace slots, into the workspace (s2 tiled, sqn, st2): the fast
// decoder, then the generic one on the items it queued (those that met the k >= 8 rule).
// The fix-up pass is one lane per queued item, a few hundred latency-bound waves at most; with `fork` it runs
// on the context's second stream and the caller joins (hipStreamWaitEvent(s, ctx->ev_join)) before reading.
template <int LOGN>
int decode_stage(fa_ctx* ctx, const Workspace& ws, hipStream_t s, uint32_t slots, const Rows& sig, const uint32_t* idx, bool fork)
{
  const uint32_t spec = (uint32_t)(ctx->wire == FA_WIRE_SPEC);
  HIP_TRY(hipMemsetAsync(ws.slow_count, 0, 256, s));
  int rc = timed(ctx, s, 0, [&] {
    hipLaunchKernelGGL(decode_fast_kernel<LOGN>, dim3((slots + 63) / 64), dim3(64), 0, s, sig, idx, slots, (const uint32_t*)nullptr,
                       ws.s2, ws.sqn, ws.st2, ws.slow_count, ws.slow_list, spec);
  });
  if (rc) return rc;
  hipStream_t ds = fork ? ctx->aux : s;
  if (fork) {
    HIP_TRY(hipEventRecord(ctx->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(ds, ctx->ev_fork, 0));
  }
  rc = timed(ctx, ds, 0, [&] {
    hipLaunchKernelGGL(decode_kernel<LOGN>, dim3((slots + DECODE_WG - 1) / DECODE_WG), dim3(DECODE_WG), 0, ds, sig,
                       (const uint32_t*)ws.slow_list, idx, slots, (const uint32_t*)ws.slow_count, ws.s2, ws.sqn, ws.st2, spec);
  });
  if (!rc && fork) rc = (int)hipEventRecord(ctx->ev_join, ds);
  return rc;
}

// decode_stage and the hash kernel for a small pass: front_small_kernel, then the generic decoder's fix-up pass,
// all on `s` (no second stream: a cross-stream dependency costs more than the fix-up pass takes).
template <int LOGN>
int front_small(fa_ctx* ctx, const Workspace& ws, hipStream_t s, uint32_t slots, const uint32_t* idx, const Batch& B, uint32_t nblk)
{
  const uint32_t spec = (uint32_t)(ctx->wire == FA_WIRE_SPEC), nd = (slots + 63) / 64;
  if (slots <= ctx->tiny_limit(LOGN))  // one wave per item and stage, no fix-up pass
    return timed(ctx, s, 1, [&] {
      const uint32_t split = slots <= 128u ? slots : 0u;
      hipLaunchKernelGGL(front_tiny_kernel<LOGN>, dim3(split ? 2u * slots : slots), dim3(split ? 64 : 128), 0, s, B.sig, idx, split, ws.s2,
                         ws.sqn, ws.st2, spec, B.msg, B.msg_off, B.mlen, nblk, ws.hraw);
    });
  const uint32_t pair = slots <= ctx->pair_hash ? 1u : 0u, nh = pair ? (slots + 31) / 32 : nd;
  HIP_TRY(hipMemsetAsync(ws.slow_count, 0, 256, s));
  int rc = timed(ctx, s, 1, [&] {  // booked as hash time: the hash waves are the longer half
    hipLaunchKernelGGL(front_small_kernel<LOGN>, dim3(nd + nh), dim3(64), 0, s, nd, B.sig, idx, slots, (const uint32_t*)nullptr, ws.s2,
                       ws.sqn, ws.st2, ws.slow_count, ws.slow_list, spec, B.msg, B.msg_off, B.mlen, nblk, ws.hraw, pair);
  });
  if (rc) return rc;
  return timed(ctx, s, 0, [&] {
    hipLaunchKernelGGL(decode_kernel<LOGN>, dim3((slots + DECODE_WG - 1) / DECODE_WG), dim3(DECODE_WG), 0, s, B.sig,
                       (const uint32_t*)ws.slow_list, idx, slots, (const uint32_t*)ws.slow_count, ws.s2, ws.sqn, ws.st2, spec);
  });
}

// One pass of the three kernels over `slots` workspace slots. idx == nullptr: slot s is item s of the arrays in
// B (already offset to the chunk); else item idx[s] of B's arrays. dcount (nullable): device-side number of
// slots, at most `slots`. retry: a pass over queued items (host loop or device-side retry pass): the generic
// decoder is enough. Items that run out of blocks are appended to retry_list / *retry_count.
template <int LOGN>
int run_pass(fa_ctx* ctx, const Workspace& ws, hipStream_t s, uint32_t slots, const uint32_t* idx, const uint32_t* dcount,
             bool retry, const Batch& B, uint32_t nblk, uint32_t item_base, uint32_t* retry_count, uint32_t* retry_list)
{
  constexpr int T = LOGN - 9;
  if (slots == 0) return 0;
  if (slots > ws.cap_slots || (size_t)slots * nblk * 136 > ws.hraw_bytes) return FA_E_INTERNAL;
  const uint32_t spec = (uint32_t)(ctx->wire == FA_WIRE_SPEC);
  const bool small = !retry && slots <= ctx->small_batch;  // does not fill the GPU: latency-bound
  const bool one_launch = retry && dcount;  // the device-side retry pass: retry_item_kernel does all three stages
  const bool fork = ctx->overlap && !ctx->quiet;
  int rc = 0;
  if (one_launch) {
    // nothing up front
  } else if (retry) {
    rc = timed(ctx, s, 0, [&] {
      hipLaunchKernelGGL(decode_kernel<LOGN>, dim3((slots + DECODE_WG - 1) / DECODE_WG), dim3(DECODE_WG), 0, s, B.sig,
                         (const uint32_t*)nullptr, idx, slots, dcount, ws.s2, ws.sqn, ws.st2, spec);
    });
  } else if (small) {
    rc = front_small<LOGN>(ctx, ws, s, slots, idx, B, nblk);
  } else {
    rc = decode_stage<LOGN>(ctx, ws, s, slots, B.sig, idx, fork);
  }
  if (rc) return rc;
  if (!small && !one_launch) {
    rc = timed(ctx, s, 1, [&] {
      hipLaunchKernelGGL(hash_kernel, dim3((slots + HASH_WG - 1) / HASH_WG), dim3(HASH_WG), 0, s, B.sig, B.msg, B.msg_off, B.mlen,
                         idx, slots, dcount, nblk, ws.hraw);
    });
    if (rc) return rc;
    if (!retry && fork) HIP_TRY(hipStreamWaitEvent(s, ctx->ev_join, 0));
  }

  VerifyArgs<LOGN> A;
  A.pk = B.keys.pk;
  A.prepared = B.keys.prepared;
  A.key_idx = B.keys.key_idx;
  A.hraw = ws.hraw;
  A.s2 = ws.s2;
  A.sqn = ws.sqn;
  A.st2 = ws.st2;
  A.idx = idx;
  A.tw = ctx->d_tw[T];
  A.itw = ctx->d_itw[T];
  A.twc = ctx->d_twc[T];
  A.isc = B.keys.prepared ? ctx->d_isc[T] : ctx->d_isc_pos[T];  // the packed-key product lacks a factor 2^32 (verify_items)
  A.nc = ctx->nc[T];
  A.count = slots;
  A.dcount = dcount;
  A.nblk = nblk;
  A.wrap32 = B.policy == FA_NORM_WRAP32;
  A.accept = B.accept;
  A.retry_count = retry_count;
  A.retry_list = retry_list;
  A.item_base = item_base;
  A.half = ctx->wire == FA_WIRE_SPEC ? Q / 2u + 1u : Q / 2u;
  A.dbg = B.dbg;
  if (one_launch) {
    const TinyFront F{ B.sig, B.msg, B.msg_off, B.mlen, spec, ws.s2, ws.sqn, ws.st2, ws.hraw };
    auto fused = [&](auto kernel) {
      return timed(ctx, s, 2, [&] { hipLaunchKernelGGL(kernel, dim3(slots), dim3(192), 0, s, A, F); });
    };
    if (B.keys.prepared) return B.debug ? fused(retry_item_kernel<LOGN, true, true>) : fused(retry_item_kernel<LOGN, false, true>);
    return B.debug ? fused(retry_item_kernel<LOGN, true, false>) : fused(retry_item_kernel<LOGN, false, false>);
  }
  constexpr uint32_t VW = (uint32_t)verify_waves<LOGN>();
  const uint32_t want = (slots + VW - 1) / VW;
  auto launch = [&](auto kernel) {
    const uint32_t grid = stride_grid(ctx, kernel, want, 64 * (int)VW);
    return timed(ctx, s, 2, [&] { hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * VW), 0, s, A); });
  };
  if (B.keys.pk.off || dcount) {  // the general instantiations: offset arrays, device-side count
    if (B.keys.prepared) return B.debug ? launch(verify_kernel<LOGN, true, true, true>) : launch(verify_kernel<LOGN, false, true, true>);
    return B.debug ? launch(verify_kernel<LOGN, true, false, true>) : launch(verify_kernel<LOGN, false, false, true>);
  }
  if (B.keys.prepared) return B.debug ? launch(verify_kernel<LOGN, true, true, false>) : launch(verify_kernel<LOGN, false, true, false>);
  return B.debug ? launch(verify_kernel<LOGN, true, false, false>) : launch(verify_kernel<LOGN, false, false, false>);
}

void set_debug(Batch& B, const fa_debug* dbgp)
{
  B.dbg = DebugOut{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  B.debug = false;
  if (dbgp) {
    B.dbg = DebugOut{ dbgp->c, dbgp->s2, dbgp->s1, (unsigned long long*)dbgp->norm_exact, dbgp->norm_wrap32, dbgp->status };
    B.debug = B.dbg.c || B.dbg.s2 || B.dbg.s1 || B.dbg.norm_exact || B.dbg.norm_wrap32 || B.dbg.status;
  }
}

// B's arrays offset to the chunk of items starting at c0 (batches without an index list).
template <int LOGN>
Batch window(const Batch& B, size_t c0)
{
  using G = Geo<LOGN>;
  Batch W = B;
  if (W.keys.pk.base) W.keys.pk.base += c0 * W.keys.pk.stride;
  if (W.keys.key_idx) W.keys.key_idx += c0;
  W.sig.base += c0 * W.sig.stride;
  if (W.msg_off) W.msg_off += c0;
  else if (W.msg) W.msg += c0 * W.mlen;
  W.accept += c0;
  if (W.dbg.c) W.dbg.c += c0 * G::N;
  if (W.dbg.s2) W.dbg.s2 += c0 * G::N;
  if (W.dbg.s1) W.dbg.s1 += c0 * G::N;
  if (W.dbg.norm_exact) W.dbg.norm_exact += c0;
  if (W.dbg.norm_wrap32) W.dbg.norm_wrap32 += c0;
  if (W.dbg.status) W.dbg.status += c0;
  return W;
}

// Main passes of a batch, chunk by chunk, enqueued on `s` without synchronising. list == nullptr: the batch is
// items [0, count) of B's arrays; else it is the `count` items list[0..count) of them (mixed batches: one class).
// Items whose rejection sampling ran out of squeezed blocks are left in ws.retry_list (indices into B's arrays)
// for verify_finish or retry_tail.
template <int LOGN>
int verify_enqueue(fa_ctx* ctx, const Workspace& ws, size_t count, const uint32_t* list, const Batch& B, hipStream_t s)
{
  constexpr int T = LOGN - 9;
  if (ws.host_counter)
    *ws.host_counter = 0u;  // the previous call that used it has synchronised
  else
    HIP_TRY(hipMemsetAsync(ws.ctl, 0, 256, s));
  const uint32_t chunk = std::min<uint32_t>(ctx->chunk, ws.cap_slots);
  for (size_t c0 = 0; c0 < count; c0 += chunk) {
    const uint32_t slots = (uint32_t)std::min<size_t>(chunk, count - c0);
    int rc;
    if (list)
      rc = run_pass<LOGN>(ctx, ws, s, slots, list + c0, nullptr, false, B, ctx->hash_blocks[T], 0u, ws.retry_counter, ws.retry_list);
    else  // the kernels see the chunk's window of the arrays; the retry list gets indices into the whole arrays
      rc = run_pass<LOGN>(ctx, ws, s, slots, nullptr, nullptr, false, window<LOGN>(B, c0), ctx->hash_blocks[T], (uint32_t)c0,
                          ws.retry_counter, ws.retry_list);
    if (rc) return rc;
  }
  return 0;
}

// Synchronises `s` and re-hashes the items verify_enqueue left over (probability ~1e-25 / ~1e-13 per item
// at the default block counts) with twice as many blocks until none is left. Same batch as the
// verify_enqueue call it completes; `max_slots` = the largest pass of that call.
template <int LOGN>
int verify_finish(fa_ctx* ctx, Workspace& ws, uint32_t max_slots, size_t items, const Batch& B, hipStream_t s)
{
  constexpr int T = LOGN - 9;
  int rc;
  uint32_t nblk = ctx->hash_blocks[T];
  for (;;) {
    uint32_t pending = 0;
    if (ws.host_counter) {
      HIP_TRY(hipStreamSynchronize(s));
      pending = *ws.host_counter;
    } else {
      HIP_TRY(hipMemcpyAsync(&pending, ws.ctl + CTL_RETRY, 4, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
    if (pending == 0) break;
    ctx->retried += pending;
    nblk *= 2;
    DevBuf list;  // this round's list; the kernels refill ws.retry_list
    if ((rc = list.reserve((size_t)pending * 4))) return rc;
    rc = (int)hipMemcpyAsync(list.p, ws.retry_list, (size_t)pending * 4, hipMemcpyDeviceToDevice, s);
    if (ws.host_counter)
      *ws.host_counter = 0u;  // the stream is idle
    else if (!rc)
      rc = (int)hipMemsetAsync(ws.ctl, 0, 256, s);
    const uint32_t chunk = std::min<uint32_t>(ctx->chunk, std::max<uint32_t>(max_slots, 128u));
    if (!rc) rc = ws_from_ctx<LOGN>(ctx, std::min<uint32_t>(chunk, pending), nblk, items, ws);  // more blocks per item: a larger hraw
    for (uint32_t r0 = 0; !rc && r0 < pending; r0 += chunk) {
      const uint32_t slots = std::min<uint32_t>(chunk, pending - r0);
      rc = run_pass<LOGN>(ctx, ws, s, slots, (const uint32_t*)list.p + r0, nullptr, true, B, nblk, 0u, ws.retry_counter, ws.retry_list);
    }
    if (!rc) rc = (int)hipStreamSynchronize(s);
    list.release();
    if (rc) return rc;
  }
  return 0;
}

// The device-side counterpart of verify_finish, for callers that must not synchronise or allocate: one more pass,
// over the items verify_enqueue queued (at most ws.tail_cap() of them), with twice the blocks per item, sized and
// driven by the device-side count. What it cannot settle (more queued items than that, or an item that runs out
// of blocks again: ~1e-50 at the default block counts) is counted as unresolved (retry_account_kernel); those
// items keep accept = 0 and the caller learns their number from fa_finish / the call's return value.
template <int LOGN>
int retry_tail(fa_ctx* ctx, const Workspace& ws, const Batch& B, hipStream_t s)
{
  constexpr int T = LOGN - 9;
  const uint32_t cap = ws.tail_cap(), nblk = 2u * ctx->hash_blocks[T];
  if (cap == 0 || (size_t)cap * nblk * 136 > ws.hraw_bytes || ws.retry_counter != ws.ctl + CTL_RETRY) return FA_E_INTERNAL;
  int rc = run_pass<LOGN>(ctx, ws, s, cap, ws.retry_list, ws.ctl + CTL_RETRY, true, B, nblk, 0u, ws.ctl + CTL_UNRESOLVED, ws.retry2_list);
  if (rc) return rc;
  hipLaunchKernelGGL(retry_account_kernel, dim3(1), dim3(1), 0, s, ws.ctl, cap, ctx->quiet ? ctx->d_hctl + 32 : ctx->d_unresolved);
  return (int)hipGetLastError();
}

// A whole device-resident verify call on the context's own workspace: main passes, then either the synchronising
// host loop or (FA_OPT_ASYNC) the device-side retry pass.
template <int LOGN>
int verify_dev(fa_ctx* ctx, size_t count, const uint32_t* list, const Batch& B, hipStream_t s)
{
  constexpr int T = LOGN - 9;
  const uint32_t max_slots = (uint32_t)std::min<size_t>(ctx->chunk, count);
  Workspace ws;
  int rc = ws_from_ctx<LOGN>(ctx, max_slots, ctx->hash_blocks[T], count, ws);
  if (!rc) rc = verify_enqueue<LOGN>(ctx, ws, count, list, B, s);
  if (rc) return rc;
  if (ctx->async) return retry_tail<LOGN>(ctx, ws, B, s);
  return verify_finish<LOGN>(ctx, ws, max_slots, count, B, s);
}

// hash_to_point for `count` (salt, message) pairs: SHAKE256 blocks, then rejection sampling; items that need
// more blocks than squeezed up front are redone with twice as many (same scheme as verify_dev).
template <int LOGN>
int hash_to_point_dev(fa_ctx* ctx, size_t count, const uint8_t* salt, const uint8_t* msg, const uint64_t* msg_off,
                      uint32_t mlen, uint16_t* c, hipStream_t s)
{
  constexpr int T = LOGN - 9;
  int rc;
  if ((rc = ctx->ctl.reserve(512))) return rc;
  if ((rc = ctx->retry_list.reserve(count * 4))) return rc;
  uint32_t* d_retry = (uint32_t*)ctx->ctl.p + CTL_RETRY;
  HIP_TRY(hipMemsetAsync(ctx->ctl.p, 0, 256, s));
  const Rows rows{ salt - 1, nullptr, 40u };  // hash_kernel reads the salt at bytes 1..40 of a row: rows of 40 bytes starting one byte early
  auto pass = [&](uint32_t slots, const uint32_t* idx, const Rows& rows_, const uint8_t* msg_, const uint64_t* off_,
                  uint32_t nblk, uint16_t* c_, uint32_t item_base) -> int {
    int r = ctx->hraw.reserve((size_t)slots * nblk * 136);
    if (r) return r;
    r = timed(ctx, s, 1, [&] {
      if (slots <= ctx->tiny_limit(LOGN))  // one sponge per wave: a third of the latency (tiny_kernel.hpp)
        hipLaunchKernelGGL(hash_tiny_kernel, dim3(slots), dim3(64), 0, s, rows_, idx, msg_, off_, mlen, nblk, (uint64_t*)ctx->hraw.p);
      else
        hipLaunchKernelGGL(hash_kernel, dim3((slots + HASH_WG - 1) / HASH_WG), dim3(HASH_WG), 0, s, rows_, msg_, off_, mlen,
                           idx, slots, (const uint32_t*)nullptr, nblk, (uint64_t*)ctx->hraw.p);
    });
    if (r) return r;
    const uint32_t grid = stride_grid(ctx, sample_kernel<LOGN>, (slots + VERIFY_WAVES - 1) / VERIFY_WAVES);
    hipLaunchKernelGGL(sample_kernel<LOGN>, dim3(grid), dim3(VERIFY_WG), 0, s, (const uint64_t*)ctx->hraw.p, idx, slots, nblk,
                       item_base, c_, d_retry, (uint32_t*)ctx->retry_list.p);
    return (int)hipGetLastError();
  };
  for (size_t c0 = 0; c0 < count; c0 += ctx->chunk) {
    const uint32_t slots = (uint32_t)std::min<size_t>(ctx->chunk, count - c0);
    rc = pass(slots, nullptr, Rows{ rows.base + c0 * 40, nullptr, 40u }, msg_off ? msg : msg + c0 * mlen, msg_off ? msg_off + c0 : nullptr,
              ctx->hash_blocks[T], c + (c0 << LOGN), (uint32_t)c0);
    if (rc) return rc;
  }
  uint32_t nblk = ctx->hash_blocks[T];
  for (;;) {
    uint32_t pending = 0;
    HIP_TRY(hipMemcpyAsync(&pending, d_retry, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (pending == 0) break;
    ctx->retried += pending;
    nblk *= 2;
    DevBuf list;
    if ((rc = list.reserve((size_t)pending * 4))) return rc;
    rc = (int)hipMemcpyAsync(list.p, ctx->retry_list.p, (size_t)pending * 4, hipMemcpyDeviceToDevice, s);
    if (!rc) rc = (int)hipMemsetAsync(ctx->ctl.p, 0, 256, s);
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
                         ctx->d_itw[T], ctx->d_twc[T], ctx->d_isc_neg[T], ctx->nc[T]);  // its product lacks a factor -2^32
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
  constexpr int T = LOGN - 9;
  Workspace ws;
  int rc = ws_from_ctx<LOGN>(ctx, (uint32_t)std::min<size_t>(ctx->chunk, count), ctx->hash_blocks[T], 1, ws);
  if (rc) return rc;
  for (size_t c0 = 0; c0 < count; c0 += ctx->chunk) {
    const uint32_t slots = (uint32_t)std::min<size_t>(ctx->chunk, count - c0);
    const Rows rows{ sig + c0 * SIGLEN, nullptr, (uint32_t)SIGLEN };
    if (slots <= ctx->tiny_limit(LOGN)) {  // one decoder wave per signature (tiny_kernel.hpp)
      rc = timed(ctx, s, 0, [&] {
        hipLaunchKernelGGL(decode_tiny_kernel<LOGN>, dim3(slots), dim3(64), 0, s, rows, ws.s2, ws.sqn, ws.st2,
                           (uint32_t)(ctx->wire == FA_WIRE_SPEC));
      });
    } else {
      rc = decode_stage<LOGN>(ctx, ws, s, slots, rows, nullptr, false);
    }
    if (rc) return rc;
    const uint32_t grid = std::min<uint32_t>((slots + 3) / 4, (uint32_t)ctx->num_cu * 16u);
    hipLaunchKernelGGL(untile_sig_kernel<LOGN>, dim3(grid), dim3(256), 0, s, sig + c0 * SIGLEN, (const uint16_t*)ws.s2,
                       (const uint8_t*)ws.st2, slots, salt + c0 * 40, s2 + (c0 << LOGN), ok + c0);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

}  // namespace

// ---- allocation-free entry point: static device scratch (task item 7; no counterpart in the reference) ----
// The pipeline workspace of a batch of at most FA_NOALLOC_MAX_ITEMS items of either N, carved out of one static
// device array that every GPU the library runs on holds until the process ends. `scratch` is a host-side stand-in
// for the symbol: the device address is taken with hipGetSymbolAddress, per device, at context creation.
namespace {
constexpr size_t NOALLOC_ITEMS = FA_NOALLOC_MAX_ITEMS;
constexpr size_t SC_CTL = 0;                                                        // 512 bytes of counters
constexpr size_t SC_S2 = 512;                                                       // tiled s2, Falcon-1024 rows
constexpr size_t SC_HRAW = SC_S2 + align256(NOALLOC_ITEMS * 1024 * 2);              // 17 blocks per item
constexpr size_t SC_SQN = SC_HRAW + align256(NOALLOC_ITEMS * 17 * 136);
constexpr size_t SC_ST2 = SC_SQN + align256(NOALLOC_ITEMS * 4);
constexpr size_t SC_RETRY = SC_ST2 + align256(NOALLOC_ITEMS);
constexpr size_t SC_SLOW = SC_RETRY + align256(NOALLOC_ITEMS * 4);
constexpr size_t SC_RETRY2 = SC_SLOW + align256(NOALLOC_ITEMS * 4);
constexpr size_t SC_BYTES = SC_RETRY2 + align256(NOALLOC_ITEMS * 4);
static_assert(NOALLOC_ITEMS % 64 == 0 && NOALLOC_ITEMS >= 128 && SC_BYTES < ((size_t)20 << 20), "scratch layout");
}  // namespace
static __device__ __align__(256) unsigned char scratch[SC_BYTES];

namespace {
std::mutex g_scratch_lock;
unsigned char* g_scratch_ptr[FA_MAX_DEVICES] = {};  // per device: each GPU holds its own copy of the array at its own address
std::mutex g_scratch_use[FA_MAX_DEVICES];           // held by the call that is using that GPU's copy (fa_verify_many_dev_noalloc)

// The device address of `scratch` on `device` (which must be current), asked once per device.
int scratch_address(int device, unsigned char** out)
{
  if (device < 0 || device >= FA_MAX_DEVICES) return FA_E_ARG;
  std::lock_guard<std::mutex> hold(g_scratch_lock);
  if (!g_scratch_ptr[device]) {
    unsigned char* p = nullptr;
    HIP_TRY(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
    if (!p) return FA_E_INTERNAL;
    g_scratch_ptr[device] = p;
  }
  *out = g_scratch_ptr[device];
  return 0;
}

// The scratch array as a pipeline workspace (hash_blocks up to 17 per item for the main pass).
Workspace ws_from_scratch(unsigned char* sp)
{
  Workspace ws;
  ws.ctl = (uint32_t*)(sp + SC_CTL);
  ws.slow_count = ws.ctl + CTL_WORDS;
  ws.retry_counter = ws.ctl + CTL_RETRY;
  ws.s2 = (uint16_t*)(sp + SC_S2);
  ws.hraw = (uint64_t*)(sp + SC_HRAW);
  ws.hraw_bytes = NOALLOC_ITEMS * 17 * 136;
  ws.sqn = (uint32_t*)(sp + SC_SQN);
  ws.st2 = (uint8_t*)(sp + SC_ST2);
  ws.retry_list = (uint32_t*)(sp + SC_RETRY);
  ws.slow_list = (uint32_t*)(sp + SC_SLOW);
  ws.retry2_list = (uint32_t*)(sp + SC_RETRY2);
  ws.cap_slots = (uint32_t)NOALLOC_ITEMS;
  return ws;
}

int check_verify_args(const fa_ctx* ctx, int logn, size_t count, const void* keys, const uint8_t* msg, const uint64_t* msg_off,
                      uint32_t mlen, const uint8_t* sig, int norm_policy, const uint8_t* accept)
{
  if (!ctx || (norm_policy != FA_NORM_EXACT64 && norm_policy != FA_NORM_WRAP32)) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!keys || !sig || !accept || (!msg && (msg_off || mlen))) return FA_E_ARG;
  if (count > 0xFFFFFFFFull) return FA_E_ARG;
  if (mlen > FA_MAX_MSG_LEN) return FA_E_ARG;  // the hash kernel counts message bytes in 32-bit signed integers
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
    case FA_E_UNRESOLVED: return "items left unresolved by the device-side retry pass: verify them again with a synchronising call";
    case FA_E_TOO_MANY: return "more items than the allocation-free entry point takes (FA_NOALLOC_MAX_ITEMS)";
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
    if (!rc) rc = (int)hipEventCreateWithFlags(&ctx->ev_ws, hipEventDisableTiming);
    if (!rc) rc = (int)hipHostMalloc((void**)&ctx->h_word, 64, hipHostMallocDefault);
    if (!rc) rc = (int)hipHostMalloc((void**)&ctx->h_ctl, 256, hipHostMallocMapped | hipHostMallocCoherent);
    if (!rc) rc = (int)hipHostGetDevicePointer((void**)&ctx->d_hctl, ctx->h_ctl, 0);
    if (!rc) {
      memset(ctx->h_ctl, 0, 256);
      ctx->d_unresolved = ctx->d_hctl + 16;
    }
    if (!rc) rc = scratch_address(device, &ctx->scratch);  // this device current (Guard above)
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
    if (!rc) rc = (int)hipMalloc((void**)&ctx->d_isc[t], bytes);
    if (!rc) rc = (int)hipMemcpy(ctx->d_twc[t], tb.fwdC.data(), cbytes, hipMemcpyHostToDevice);
    if (!rc) rc = (int)hipMemcpy(ctx->d_isc[t], tb.isc.data(), bytes, hipMemcpyHostToDevice);
    if (!rc) rc = (int)hipMalloc((void**)&ctx->d_isc_pos[t], bytes);
    if (!rc) rc = (int)hipMemcpy(ctx->d_isc_pos[t], tb.isc_pos.data(), bytes, hipMemcpyHostToDevice);
    if (!rc) rc = (int)hipMalloc((void**)&ctx->d_isc_neg[t], bytes);
    if (!rc) rc = (int)hipMemcpy(ctx->d_isc_neg[t], tb.isc_neg.data(), bytes, hipMemcpyHostToDevice);
  } catch (...) {  // std::bad_alloc while building the tables
    rc = FA_E_INTERNAL;
  }
  if (!rc) {  // resident workgroups per CU of the kernels the allocation-free entry point launches, asked now
    (void)stride_grid(ctx, verify_kernel<9, false, false, false>, 1u, 64 * verify_waves<9>());
    (void)stride_grid(ctx, verify_kernel<10, false, false, false>, 1u, 64 * verify_waves<10>());
    (void)stride_grid(ctx, verify_kernel<9, false, false, true>, 1u, 64 * verify_waves<9>());
    (void)stride_grid(ctx, verify_kernel<10, false, false, true>, 1u, 64 * verify_waves<10>());
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
  // Work enqueued on a caller's stream may still use the context's buffers: wait for the last use of the workspace and
  // for the context's own streams. (Not hipDeviceSynchronize: contexts are also destroyed from thread-exit and static
  // destructors, where a device-wide wait on a runtime that is shutting down does not return.)
  if (ctx->ws_busy && ctx->ev_ws) (void)hipEventSynchronize(ctx->ev_ws);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
  if (ctx->copy) (void)hipStreamSynchronize(ctx->copy);
  for (auto& e : ctx->events) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (hipEvent_t e : ctx->ev_free) (void)hipEventDestroy(e);
  for (int t = 0; t < 2; t++) {
    if (ctx->d_tw[t]) (void)hipFree(ctx->d_tw[t]);
    if (ctx->d_itw[t]) (void)hipFree(ctx->d_itw[t]);
    if (ctx->d_twc[t]) (void)hipFree(ctx->d_twc[t]);
    if (ctx->d_isc[t]) (void)hipFree(ctx->d_isc[t]);
    if (ctx->d_isc_pos[t]) (void)hipFree(ctx->d_isc_pos[t]);
    if (ctx->d_isc_neg[t]) (void)hipFree(ctx->d_isc_neg[t]);
  }
  if (ctx->h_word) (void)hipHostFree(ctx->h_word);
  if (ctx->h_ctl) (void)hipHostFree(ctx->h_ctl);
  DevBuf* bufs[] = { &ctx->hraw, &ctx->s2, &ctx->sqn, &ctx->st2, &ctx->ctl, &ctx->retry_list, &ctx->slow_list, &ctx->retry2_list,
                     &ctx->part_list, &ctx->part_runs, &ctx->h_pk, &ctx->h_sig, &ctx->h_msg, &ctx->h_off, &ctx->h_a, &ctx->h_b,
                     &ctx->h_c, &ctx->h_off2, &ctx->h_off3, &ctx->h_tag, &ctx->spec_pk };
  for (DevBuf* b : bufs) b->release();
  for (auto& st : ctx->hs) {
    DevBuf* sb[] = { &st.pk, &st.sig, &st.msg, &st.off, &st.kidx, &st.accept };
    for (DevBuf* b : sb) b->release();
    for (DevBuf& b : st.dbg) b.release();
    st.pin_pk.release();
    st.pin_sig.release();
    st.pin_msg.release();
    ctx->zc.release();
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
  if (ctx->copy) (void)hipStreamDestroy(ctx->copy);
  for (hipEvent_t e : ctx->ev_ready)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_ws) (void)hipEventDestroy(ctx->ev_ws);
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
    case FA_OPT_TINY_BATCH_ITEMS:
      if (value < 0 || value > (1 << 20)) return FA_E_ARG;
      ctx->tiny_batch = (uint32_t)value;
      return 0;
    case FA_OPT_PAIR_HASH_ITEMS:
      if (value < 0 || value > (1 << 20)) return FA_E_ARG;
      ctx->pair_hash = (uint32_t)value;
      return 0;
    case FA_OPT_GRID_PER_CU:
      if (value < 0 || value > 65536) return FA_E_ARG;
      ctx->grid_per_cu = (uint32_t)value;
      return 0;
    case FA_OPT_ASYNC:
      ctx->async = value != 0;
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
  int rc = 0;
  // every recorded pair leaves ctx->events here, whatever happens: back to the pool, or destroyed
  while (!ctx->events.empty()) {
    const EventPair e = ctx->events.back();
    ctx->events.pop_back();
    float ms = 0.f;
    if (!rc) rc = (int)hipEventSynchronize(e.b);
    if (!rc) rc = (int)hipEventElapsedTime(&ms, e.a, e.b);
    if (!rc) {
      double* acc[] = { &out->ms_decode, &out->ms_hash, &out->ms_verify, &out->ms_polymul };
      uint32_t* cnt[] = { &out->launches_decode, &out->launches_hash, &out->launches_verify, &out->launches_polymul };
      *acc[e.kind] += ms;
      *cnt[e.kind] += 1;
    }
    bool kept = false;
    if (ctx->ev_free.size() < 4096) try {
        ctx->ev_free.reserve(ctx->ev_free.size() + 2);
        ctx->ev_free.push_back(e.a);
        ctx->ev_free.push_back(e.b);
        kept = true;
      } catch (...) {
      }
    if (!kept) {
      (void)hipEventDestroy(e.a);
      (void)hipEventDestroy(e.b);
    }
  }
  out->retried_items = ctx->retried;
  ctx->retried = 0;
  return rc;
}

int fa_finish(fa_ctx* ctx, void* stream, uint64_t* unresolved)
{
  if (!ctx) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipStreamSynchronize(s));
  // written by retry_account_kernel with system-scope atomics, in host memory; calls of this context on OTHER streams may
  // still be running, so the word is taken and cleared in one step: an increment lands either in this count or in the next
  const uint32_t n = __atomic_exchange_n(ctx->h_ctl + 16, 0u, __ATOMIC_ACQ_REL);
  if (unresolved) *unresolved = n;
  return n ? FA_E_UNRESOLVED : 0;
}

int fa_verify_many_dev(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, const uint8_t* msg,
                       const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig, int norm_policy, uint8_t* accept,
                       const fa_debug* dbg, void* stream)
{
  int rc = check_verify_args(ctx, logn, count, pk, msg, msg_off, mlen, sig, norm_policy, accept);
  if (rc || count == 0) return rc;
  if (!aligned4(pk) || !aligned4(sig) || !aligned4(msg) || ((uintptr_t)msg_off & 7u)) return FA_E_ALIGN;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  WsScope fence(ctx, s);
  if ((rc = fence.begin())) return rc;
  Batch B;
  const uint8_t* pkr = pk;
  if ((rc = keys_in_reference_form(ctx, logn, s, pk, count, &pkr))) return rc;
  B.keys.pk = Rows{ pkr, nullptr, (uint32_t)fa_pkey_len(logn) };
  B.sig = Rows{ sig, nullptr, (uint32_t)fa_sig_len(logn) };
  B.msg = msg;
  B.msg_off = msg_off;
  B.mlen = mlen;
  B.policy = norm_policy;
  B.accept = accept;
  set_debug(B, dbg);
  rc = logn == 9 ? verify_dev<9>(ctx, count, nullptr, B, s) : verify_dev<10>(ctx, count, nullptr, B, s);
  const int re = fence.end();
  return rc ? rc : re;
}

int fa_verify_many_dev_noalloc(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, const uint8_t* msg,
                               const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig, int norm_policy, uint8_t* accept,
                               void* stream)
{
  int rc = check_verify_args(ctx, logn, count, pk, msg, msg_off, mlen, sig, norm_policy, accept);
  if (rc || count == 0) return rc;
  if (count > NOALLOC_ITEMS) return FA_E_TOO_MANY;
  if (ctx->wire != FA_WIRE_REFERENCE) return FA_E_ARG;  // the specification wire format repacks keys into a buffer of the context
  if (ctx->hash_blocks[logn - 9] > 17u) return FA_E_ARG;  // the scratch array holds 17 blocks per item
  if (!aligned4(pk) || !aligned4(sig) || !aligned4(msg) || ((uintptr_t)msg_off & 7u)) return FA_E_ALIGN;
  if (!ctx->scratch) return FA_E_INTERNAL;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  // Every context on this GPU carves its workspace out of the SAME array: calls through different contexts (the C++
  // layer keeps one per host thread) would overwrite each other's s2 tiles, hash blocks and counters while kernels run.
  // One lock per GPU, held until this call's stream synchronisation below: such calls run one after the other. An
  // uncontended lock costs nothing and allocates nothing.
  std::lock_guard<std::mutex> one_at_a_time(g_scratch_use[ctx->device]);
  WsScope fence(ctx, s);
  if ((rc = fence.begin())) return rc;
  const Workspace ws = ws_from_scratch(ctx->scratch);  // the device address, not the symbol
  Batch B;
  B.keys.pk = Rows{ pk, nullptr, (uint32_t)fa_pkey_len(logn) };
  B.sig = Rows{ sig, nullptr, (uint32_t)fa_sig_len(logn) };
  B.msg = msg;
  B.msg_off = msg_off;
  B.mlen = mlen;
  B.policy = norm_policy;
  B.accept = accept;
  ctx->quiet = true;  // no profiling events, no second stream: nothing below allocates
  volatile uint32_t* unresolved = ctx->h_ctl + 32;  // this call's own count, in the GPU-mapped host block (retry_tail)
  *unresolved = 0u;
  if (logn == 9) {
    rc = verify_enqueue<9>(ctx, ws, count, nullptr, B, s);
    if (!rc) rc = retry_tail<9>(ctx, ws, B, s);
  } else {
    rc = verify_enqueue<10>(ctx, ws, count, nullptr, B, s);
    if (!rc) rc = retry_tail<10>(ctx, ws, B, s);
  }
  ctx->quiet = false;
  const int re = fence.end();
  const int rs = (int)hipStreamSynchronize(s);  // also on failure: nothing of this call may still use the array when the lock goes
  if (!rc) rc = re;
  if (!rc) rc = rs;
  if (rc) return rc;
  return *unresolved ? FA_E_UNRESOLVED : 0;
}

int fa_verify_many_prepared_dev(fa_ctx* ctx, int logn, size_t count, const uint16_t* prepared, const uint32_t* key_idx,
                                const uint8_t* msg, const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig,
                                int norm_policy, uint8_t* accept, const fa_debug* dbg, void* stream)
{
  int rc = check_verify_args(ctx, logn, count, prepared, msg, msg_off, mlen, sig, norm_policy, accept);
  if (rc || count == 0) return rc;
  if (!aligned4(prepared) || !aligned4(key_idx) || !aligned4(sig) || !aligned4(msg) || ((uintptr_t)msg_off & 7u)) return FA_E_ALIGN;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  WsScope fence(ctx, s);
  if ((rc = fence.begin())) return rc;
  Batch B;
  B.keys.prepared = prepared;
  B.keys.key_idx = key_idx;
  B.sig = Rows{ sig, nullptr, (uint32_t)fa_sig_len(logn) };
  B.msg = msg;
  B.msg_off = msg_off;
  B.mlen = mlen;
  B.policy = norm_policy;
  B.accept = accept;
  set_debug(B, dbg);
  rc = logn == 9 ? verify_dev<9>(ctx, count, nullptr, B, s) : verify_dev<10>(ctx, count, nullptr, B, s);
  const int re = fence.end();
  return rc ? rc : re;
}

// Mixed Falcon-512 / Falcon-1024 batch, device-resident: the item indices are split by N on the GPU (three small
// kernels, order-preserving), then each class runs through the ordinary passes with its index list.
int fa_verify_mixed_dev(fa_ctx* ctx, size_t count, const uint8_t* logn, const uint8_t* pk, const uint64_t* pk_off,
                        const uint8_t* msg, const uint64_t* msg_off, const uint8_t* sig, const uint64_t* sig_off,
                        int norm_policy, uint8_t* accept, void* stream)
{
  if (!ctx || (norm_policy != FA_NORM_EXACT64 && norm_policy != FA_NORM_WRAP32)) return FA_E_ARG;
  if (count == 0) return 0;
  if (!logn || !pk || !pk_off || !msg_off || !sig || !sig_off || !accept || count > 0xFFFFFFFFull) return FA_E_ARG;
  if (((uintptr_t)pk_off & 7u) || ((uintptr_t)msg_off & 7u) || ((uintptr_t)sig_off & 7u)) return FA_E_ALIGN;
  if (ctx->wire != FA_WIRE_REFERENCE) return FA_E_ARG;  // the specification wire format is offered for single-N batches only
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  WsScope fence(ctx, s);
  if ((rc = fence.begin())) return rc;
  const uint32_t n = (uint32_t)count, nruns = (n + PART_ITEMS - 1) / PART_ITEMS;
  if ((rc = ctx->part_list.reserve((size_t)n * 4))) return rc;
  if ((rc = ctx->part_runs.reserve(256 + (size_t)nruns * 4))) return rc;
  uint32_t* totals = (uint32_t*)ctx->part_runs.p;  // [0] n9, [1] items whose tag is neither 9 nor 10
  uint32_t* runs = totals + 64;
  uint32_t* list = (uint32_t*)ctx->part_list.p;
  HIP_TRY(hipMemsetAsync(totals, 0, 256, s));
  hipLaunchKernelGGL(partition_count_kernel, dim3(nruns), dim3(PART_WG), 0, s, logn, n, runs, totals);
  hipLaunchKernelGGL(partition_scan_kernel, dim3(1), dim3(PART_WG), 0, s, runs, nruns, totals);
  hipLaunchKernelGGL(partition_write_kernel, dim3(nruns), dim3(PART_WG), 0, s, logn, n, (const uint32_t*)runs, (const uint32_t*)totals, list);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(ctx->h_word, totals, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));  // the class sizes decide the passes
  const uint32_t n9 = ctx->h_word[0], bad = ctx->h_word[1];
  if (bad || n9 > n) return FA_E_LOGN;
  Batch B;
  B.msg = msg;
  B.msg_off = msg_off;
  B.policy = norm_policy;
  B.accept = accept;
  B.keys.pk = Rows{ pk, pk_off, 0u };
  B.sig = Rows{ sig, sig_off, 0u };
  if (n9) rc = verify_dev<9>(ctx, n9, list, B, s);
  if (!rc && n - n9) rc = verify_dev<10>(ctx, n - n9, list + n9, B, s);
  const int re = fence.end();
  return rc ? rc : re;
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
  WsScope fence(ctx, s);  // FA_WIRE_SPEC repacks the keys into a buffer of the context
  if (const int rb = fence.begin()) return rb;
  const int T = logn - 9;
  const uint32_t n = (uint32_t)nkeys, want = (n + VERIFY_WAVES - 1) / VERIFY_WAVES;
  {
    const int rc = keys_in_reference_form(ctx, logn, s, pk, nkeys, &pk);
    if (rc) return rc;
  }
#define FA_LAUNCH_PREP(LN, PACKED)                                                                                      \
  hipLaunchKernelGGL((prepare_keys_kernel<LN, PACKED>), dim3(stride_grid(ctx, prepare_keys_kernel<LN, PACKED>, want)),   \
                     dim3(VERIFY_WG), 0, s, pk, h, n, prepared, ctx->d_tw[T], ctx->d_itw[T], ctx->d_twc[T],              \
                     ctx->d_isc[T], ctx->nc[T])
  if (logn == 9) {
    if (pk) FA_LAUNCH_PREP(9, true); else FA_LAUNCH_PREP(9, false);
  } else {
    if (pk) FA_LAUNCH_PREP(10, true); else FA_LAUNCH_PREP(10, false);
  }
#undef FA_LAUNCH_PREP
  const int rl = (int)hipGetLastError(), re = fence.end();
  return rl ? rl : re;
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
// Keys: packed, one per item (`pk`); or prepared, in caller memory (`prepared`, uploaded with the call); or prepared
// and already resident on the GPU (`resident`, a key set): then only signatures, messages and key indices cross the link.
static int verify_host(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, const uint16_t* prepared, const uint16_t* resident,
                       size_t nkeys, const uint32_t* key_idx, const uint8_t* msg, const uint64_t* msg_off, uint32_t mlen,
                       const uint8_t* sig, int norm_policy, uint8_t* accept, const fa_debug* dbg)
{
  if (!ctx || (norm_policy != FA_NORM_EXACT64 && norm_policy != FA_NORM_WRAP32)) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if ((!pk && !prepared && !resident) || !sig || !accept || (!msg && (msg_off || mlen))) return FA_E_ARG;
  if (count > 0xFFFFFFFFull) return FA_E_ARG;
  const bool keyed = prepared || resident;  // items name their key by index
  if (keyed) {  // every item must name an existing key
    if (nkeys == 0 || nkeys > 0xFFFFFFFFull || (!key_idx && nkeys < count)) return FA_E_ARG;
    if (key_idx)
      for (size_t i = 0; i < count; i++)
        if (key_idx[i] >= nkeys) return FA_E_ARG;
  }
  if (mlen > FA_MAX_MSG_LEN) return FA_E_ARG;
  if (msg_off)
    for (size_t i = 0; i < count; i++)
      if (msg_off[i + 1] < msg_off[i] || msg_off[i + 1] - msg_off[i] > FA_MAX_MSG_LEN) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  const size_t n = (size_t)1 << logn, pklen = fa_pkey_len(logn), siglen = fa_sig_len(logn);
  hipStream_t cs = ctx->stream, xs = ctx->copy;
  int rc;
  WsScope fence(ctx, cs);
  if ((rc = fence.begin())) return rc;

  // A handful of items (the one-wave-per-item front, FA_OPT_TINY_BATCH_ITEMS): three staged copies, two events and
  // the copy back cost more than the kernels take. The inputs are gathered in pinned host memory instead, which the
  // kernels read in place over the host link (a few KB per item, read once), and the accept bytes are written there
  // too: no copy engine, one synchronisation. One Falcon-512 item: 98 -> about 55 us per call.
  constexpr size_t ZC_MAX_BYTES = 2u << 20;
  if (!prepared && !dbg && count <= ctx->tiny_limit(logn)) {
    auto up = [](size_t v) { return (v + 63u) & ~(size_t)63u; };
    const size_t first = msg_off ? (size_t)msg_off[0] : 0;
    const size_t msg_bytes = msg_off ? (size_t)msg_off[count] - first : count * (size_t)mlen;
    // packed keys, or (resident key set) the items' key indices, in the first slot
    const size_t o_pk = 0, o_sig = up(o_pk + (resident ? (key_idx ? count * 4 : 0) : count * pklen + 8)), o_msg = up(o_sig + count * siglen + 8);
    const size_t o_off = up(o_msg + msg_bytes + 16), o_acc = up(o_off + (msg_off ? (count + 1) * 8 : 0)), total = up(o_acc + count);
    if (total <= ZC_MAX_BYTES) {
      if ((rc = ctx->zc.reserve(ZC_MAX_BYTES))) return rc;
      uint8_t* h = (uint8_t*)ctx->zc.p;
      uint8_t* d = nullptr;
      HIP_TRY(hipHostGetDevicePointer((void**)&d, h, 0));
      if (!resident) memcpy(h + o_pk, pk, count * pklen);
      else if (key_idx) memcpy(h + o_pk, key_idx, count * 4);
      memcpy(h + o_sig, sig, count * siglen);
      const size_t m0 = o_msg + (first & 3u);  // message bytes land at their offset modulo 4: the kernels' base stays aligned
      if (msg_bytes) memcpy(h + m0, msg + (msg_off ? first : 0), msg_bytes);
      if (msg_off) memcpy(h + o_off, msg_off, (count + 1) * 8);
      Batch B;
      B.policy = norm_policy;
      B.mlen = mlen;
      if (resident) {
        B.keys.prepared = resident;
        B.keys.key_idx = key_idx ? (const uint32_t*)(d + o_pk) : nullptr;
      } else {
        B.keys.pk = Rows{ d + o_pk, nullptr, (uint32_t)pklen };
      }
      B.sig = Rows{ d + o_sig, nullptr, (uint32_t)siglen };
      B.msg = d + m0 - first;
      B.msg_off = msg_off ? (const uint64_t*)(d + o_off) : nullptr;
      B.accept = d + o_acc;
      if (!resident && ctx->wire == FA_WIRE_SPEC && (rc = keys_in_reference_form(ctx, logn, cs, B.keys.pk.base, count, &B.keys.pk.base))) return rc;
      Workspace ws;
      rc = logn == 9 ? ws_from_ctx<9>(ctx, (uint32_t)count, ctx->hash_blocks[0], count, ws) : ws_from_ctx<10>(ctx, (uint32_t)count, ctx->hash_blocks[1], count, ws);
      if (!rc) rc = logn == 9 ? verify_enqueue<9>(ctx, ws, count, nullptr, B, cs) : verify_enqueue<10>(ctx, ws, count, nullptr, B, cs);
      if (!rc) rc = logn == 9 ? verify_finish<9>(ctx, ws, (uint32_t)count, count, B, cs) : verify_finish<10>(ctx, ws, (uint32_t)count, count, B, cs);
      if (!rc) memcpy(accept, h + o_acc, count);  // verify_finish has synchronised the stream
      return rc;
    }
  }

  const uint16_t* d_prepared = resident;
  if (prepared) {  // the key table is shared by all sub-batches: staged once, ahead of the first kernel
    if ((rc = ctx->h_pk.reserve(nkeys * n * 2 + 8))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->h_pk.p, prepared, nkeys * n * 2, hipMemcpyHostToDevice, cs));
    d_prepared = (const uint16_t*)ctx->h_pk.p;
  }

  struct Sub {  // one sub-batch: its window of the caller's arrays and where it sits on the device
    size_t c0 = 0, cnt = 0;
    Batch B;
    Workspace ws;
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
    S.B.policy = norm_policy;
    S.B.mlen = mlen;
    int r;
    if ((r = st.sig.reserve(S.cnt * siglen + 8))) return r;
    if ((r = st.accept.reserve(S.cnt))) return r;
    if ((r = h2d(ctx->device, st.sig.p, sig + S.c0 * siglen, S.cnt * siglen, xs, st.pin_sig, ctx->host_threads))) return r;
    S.B.sig = Rows{ (const uint8_t*)st.sig.p, nullptr, (uint32_t)siglen };
    S.B.accept = (uint8_t*)st.accept.p;
    if (keyed) {
      S.B.keys.prepared = key_idx ? d_prepared : d_prepared + S.c0 * n;  // no index: item i uses key i
      if (key_idx) {
        if ((r = st.kidx.reserve(S.cnt * 4))) return r;
        HIP_TRY(hipMemcpyAsync(st.kidx.p, key_idx + S.c0, S.cnt * 4, hipMemcpyHostToDevice, xs));
        S.B.keys.key_idx = (const uint32_t*)st.kidx.p;
      }
    } else {
      if ((r = st.pk.reserve(S.cnt * pklen + 8))) return r;
      if ((r = h2d(ctx->device, st.pk.p, pk + S.c0 * pklen, S.cnt * pklen, xs, st.pin_pk, ctx->host_threads))) return r;
      S.B.keys.pk = Rows{ (const uint8_t*)st.pk.p, nullptr, (uint32_t)pklen };
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
    S.B.msg = stage - first;
    if (msg_off) {
      if ((r = st.off.reserve((S.cnt + 1) * 8))) return r;
      HIP_TRY(hipMemcpyAsync(st.off.p, msg_off + S.c0, (S.cnt + 1) * 8, hipMemcpyHostToDevice, xs));
      S.B.msg_off = (const uint64_t*)st.off.p;
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
      set_debug(S.B, &S.d);
    }
    HIP_TRY(hipEventRecord(ctx->ev_ready[k & 1], xs));
    return 0;
  };
  const uint32_t max_slots = (uint32_t)std::min<size_t>({ (size_t)ctx->chunk, (size_t)ctx->host_chunk, count });
  auto launch = [&](size_t k, Sub& S) -> int {
    HIP_TRY(hipStreamWaitEvent(cs, ctx->ev_ready[k & 1], 0));
    if (S.B.keys.pk.base) {  // FA_WIRE_SPEC: this sub-batch's keys, repacked on the compute stream (one buffer: sub-batches run in turn)
      const int r = keys_in_reference_form(ctx, logn, cs, S.B.keys.pk.base, S.cnt, &S.B.keys.pk.base);
      if (r) return r;
    }
    int r = logn == 9 ? ws_from_ctx<9>(ctx, max_slots, ctx->hash_blocks[0], S.cnt, S.ws) : ws_from_ctx<10>(ctx, max_slots, ctx->hash_blocks[1], S.cnt, S.ws);
    if (r) return r;
    return logn == 9 ? verify_enqueue<9>(ctx, S.ws, S.cnt, nullptr, S.B, cs) : verify_enqueue<10>(ctx, S.ws, S.cnt, nullptr, S.B, cs);
  };
  auto finish = [&](Sub& S) -> int {
    int r = logn == 9 ? verify_finish<9>(ctx, S.ws, max_slots, S.cnt, S.B, cs) : verify_finish<10>(ctx, S.ws, max_slots, S.cnt, S.B, cs);
    if (r) return r;
    HIP_TRY(hipMemcpyAsync(accept + S.c0, S.B.accept, S.cnt, hipMemcpyDeviceToHost, cs));
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
  const int re = fence.end();
  const int rs = (int)hipStreamSynchronize(xs), rt = (int)hipStreamSynchronize(cs);  // nothing of the caller's is in flight on return
  return rc ? rc : rs ? rs : rt ? rt : re;
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
  return verify_host(ctx, logn, count, pk, nullptr, nullptr, 0, nullptr, msg, msg_off, mlen, sig, norm_policy, accept, dbg);
}

int fa_verify_many_prepared(fa_ctx* ctx, int logn, size_t count, const uint16_t* prepared, size_t nkeys,
                            const uint32_t* key_idx, const uint8_t* msg, const uint64_t* msg_off, uint32_t mlen,
                            const uint8_t* sig, int norm_policy, uint8_t* accept, const fa_debug* dbg)
{
  if (!prepared && count) return FA_E_ARG;
  return verify_host(ctx, logn, count, nullptr, prepared, nullptr, nkeys, key_idx, msg, msg_off, mlen, sig, norm_policy, accept, dbg);
}

// ---- resident key sets ----
int fa_keyset_create(fa_ctx* ctx, int logn, size_t nkeys, const uint8_t* pk, const uint16_t* h, fa_keyset** out)
{
  if (!out) return FA_E_ARG;
  *out = nullptr;
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (nkeys == 0 || nkeys > 0xFFFFFFFFull || (!pk) == (!h)) return FA_E_ARG;  // exactly one of pk, h
  Guard g(ctx->device);
  if (g.err) return g.err;
  fa_keyset* ks = new (std::nothrow) fa_keyset;
  if (!ks) return FA_E_INTERNAL;
  ks->device = ctx->device;
  ks->logn = logn;
  ks->nkeys = nkeys;
  const size_t n = (size_t)1 << logn, rec = pk ? fa_pkey_len(logn) : n * 2;
  int rc = (int)hipMalloc((void**)&ks->d_prepared, nkeys * n * 2 + 8);
  // the keys go through the context's staging buffer in pieces, each prepared into its place in the set
  const size_t piece = std::max<size_t>(1, std::min<size_t>(nkeys, ((size_t)64 << 20) / rec));
  hipStream_t s = ctx->stream;
  const uint8_t* src = pk ? pk : (const uint8_t*)h;
  for (size_t k0 = 0; !rc && k0 < nkeys; k0 += piece) {
    const size_t cnt = std::min(piece, nkeys - k0);
    {
      WsScope fence(ctx, s);  // the staging buffer belongs to the context
      if (!rc) rc = fence.begin();
      if (!rc) rc = ctx->h_pk.reserve(cnt * rec + 8);
      if (!rc) rc = h2d(ctx->device, ctx->h_pk.p, src + k0 * rec, cnt * rec, s, ctx->hs[0].pin_pk, ctx->host_threads);
    }
    if (!rc)
      rc = fa_prepare_keys_dev(ctx, logn, cnt, pk ? (const uint8_t*)ctx->h_pk.p : nullptr, pk ? nullptr : (const uint16_t*)ctx->h_pk.p,
                               ks->d_prepared + k0 * n, s);
    const int rs = (int)hipStreamSynchronize(s);  // the pinned staging piece is free again
    if (!rc) rc = rs;
  }
  if (rc) {
    fa_keyset_destroy(ks);
    return rc;
  }
  *out = ks;
  return 0;
}

void fa_keyset_destroy(fa_keyset* ks)
{
  if (!ks) return;
  Guard g(ks->device);
  if (ks->d_prepared) (void)hipFree(ks->d_prepared);  // waits for the device's work, as hipFree does
  delete ks;
}

size_t fa_keyset_size(const fa_keyset* ks) { return ks ? ks->nkeys : 0; }
int fa_keyset_logn(const fa_keyset* ks) { return ks ? ks->logn : 0; }
const uint16_t* fa_keyset_device_keys(const fa_keyset* ks) { return ks ? ks->d_prepared : nullptr; }

int fa_verify_many_keyset(fa_ctx* ctx, const fa_keyset* ks, size_t count, const uint32_t* key_idx, const uint8_t* msg,
                          const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig, int norm_policy, uint8_t* accept,
                          const fa_debug* dbg)
{
  if (!ctx || !ks) return FA_E_ARG;
  if (ks->device != ctx->device) return FA_E_ARG;  // the set lives in another GPU's memory
  return verify_host(ctx, ks->logn, count, nullptr, nullptr, ks->d_prepared, ks->nkeys, key_idx, msg, msg_off, mlen, sig, norm_policy,
                     accept, dbg);
}

int fa_verify_mixed(fa_ctx* ctx, size_t count, const uint8_t* logn, const uint8_t* pk, const uint64_t* pk_off,
                    const uint8_t* msg, const uint64_t* msg_off, const uint8_t* sig, const uint64_t* sig_off,
                    int norm_policy, uint8_t* accept)
{
  if (!ctx) return FA_E_ARG;
  if (count == 0) return 0;
  if (!logn || !pk || !pk_off || !msg_off || !sig || !sig_off || !accept || count > 0xFFFFFFFFull) return FA_E_ARG;
  // extent of each array: the records may sit anywhere in it, so the whole span up to the last byte used is staged
  size_t pk_bytes = 0, sig_bytes = 0;
  for (size_t i = 0; i < count; i++) {
    if (logn[i] != 9 && logn[i] != 10) return FA_E_LOGN;
    if (msg_off[i + 1] < msg_off[i] || msg_off[i + 1] - msg_off[i] > FA_MAX_MSG_LEN) return FA_E_ARG;
    pk_bytes = std::max<size_t>(pk_bytes, (size_t)pk_off[i] + fa_pkey_len(logn[i]));
    sig_bytes = std::max<size_t>(sig_bytes, (size_t)sig_off[i] + fa_sig_len(logn[i]));
  }
  const size_t msg_bytes = (size_t)msg_off[count];
  if (msg_bytes && !msg) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = ctx->stream;
  int rc;
  WsScope fence(ctx, s);  // the staging buffers belong to the context (fa_verify_mixed_dev brackets its own work again)
  if ((rc = fence.begin())) return rc;
  if ((rc = ctx->h_pk.reserve(pk_bytes + 16))) return rc;
  if ((rc = ctx->h_sig.reserve(sig_bytes + 16))) return rc;
  if ((rc = ctx->h_msg.reserve(msg_bytes + 16))) return rc;
  if ((rc = ctx->h_off.reserve((count + 1) * 8))) return rc;
  if ((rc = ctx->h_off2.reserve(count * 8))) return rc;
  if ((rc = ctx->h_off3.reserve(count * 8))) return rc;
  if ((rc = ctx->h_tag.reserve(count))) return rc;
  if ((rc = ctx->h_a.reserve(count))) return rc;
  if ((rc = h2d(ctx->device, ctx->h_pk.p, pk, pk_bytes, s, ctx->hs[0].pin_pk, ctx->host_threads))) return rc;
  if ((rc = h2d(ctx->device, ctx->h_sig.p, sig, sig_bytes, s, ctx->hs[0].pin_sig, ctx->host_threads))) return rc;
  if (msg_bytes && (rc = h2d(ctx->device, ctx->h_msg.p, msg, msg_bytes, s, ctx->hs[0].pin_msg, ctx->host_threads))) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->h_off.p, msg_off, (count + 1) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->h_off2.p, pk_off, count * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->h_off3.p, sig_off, count * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->h_tag.p, logn, count, hipMemcpyHostToDevice, s));
  const bool was_async = ctx->async;
  ctx->async = false;  // this call returns results: it settles every item before the copy back
  rc = fa_verify_mixed_dev(ctx, count, (const uint8_t*)ctx->h_tag.p, (const uint8_t*)ctx->h_pk.p, (const uint64_t*)ctx->h_off2.p,
                           (const uint8_t*)ctx->h_msg.p, (const uint64_t*)ctx->h_off.p, (const uint8_t*)ctx->h_sig.p,
                           (const uint64_t*)ctx->h_off3.p, norm_policy, (uint8_t*)ctx->h_a.p, s);
  ctx->async = was_async;
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(accept, ctx->h_a.p, count, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
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
  if (mlen > FA_MAX_MSG_LEN) return FA_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  WsScope fence(ctx, s);
  if (const int rb = fence.begin()) return rb;
  const int rc = logn == 9 ? hash_to_point_dev<9>(ctx, count, salt, msg, msg_off, mlen, c, s) : hash_to_point_dev<10>(ctx, count, salt, msg, msg_off, mlen, c, s);
  const int re = fence.end();
  return rc ? rc : re;
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
                       ctx->d_tw[T], ctx->d_itw[T], ctx->d_twc[T], ctx->d_isc[T], ctx->nc[T]);
  else
    hipLaunchKernelGGL(pubkey_kernel<10>, dim3(stride_grid(ctx, pubkey_kernel<10>, want)), dim3(VERIFY_WG), 0, s, f, g, h, n,
                       ctx->d_tw[T], ctx->d_itw[T], ctx->d_twc[T], ctx->d_isc[T], ctx->nc[T]);
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
                     a, out, n, ctx->d_tw[T], ctx->d_itw[T], ctx->d_twc[T], ctx->d_isc[T], ctx->nc[T])
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
  WsScope fence(ctx, s);  // FA_WIRE_SPEC repacks the keys into a buffer of the context
  if (const int rb = fence.begin()) return rb;
  const uint32_t n = (uint32_t)count, want = (n + VERIFY_WAVES - 1) / VERIFY_WAVES;
  {
    const int rc = keys_in_reference_form(ctx, logn, s, pk, count, &pk);
    if (rc) return rc;
  }
  if (logn == 9)
    hipLaunchKernelGGL(decode_pkeys_kernel<9>, dim3(stride_grid(ctx, decode_pkeys_kernel<9>, want)), dim3(VERIFY_WG), 0, s, pk, n, h, ok);
  else
    hipLaunchKernelGGL(decode_pkeys_kernel<10>, dim3(stride_grid(ctx, decode_pkeys_kernel<10>, want)), dim3(VERIFY_WG), 0, s, pk, n, h, ok);
  const int rl = (int)hipGetLastError(), re = fence.end();
  return rl ? rl : re;
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
  hipStream_t s = (hipStream_t)stream;  // the decoders write the context's workspace, which the untiling kernel reads
  WsScope fence(ctx, s);
  if (const int rb = fence.begin()) return rb;
  const int rc = logn == 9 ? decode_sigs_dev<9>(ctx, count, sig, salt, s2, ok, s) : decode_sigs_dev<10>(ctx, count, sig, salt, s2, ok, s);
  const int re = fence.end();
  return rc ? rc : re;
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

// ---- the encoders ----
int fa_encode_pkeys_dev(fa_ctx* ctx, int logn, size_t count, const uint16_t* h, uint8_t* pk, void* stream)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!h || !pk || count > 0xFFFFFFFFull) return FA_E_ARG;
  if ((uintptr_t)h & 7u) return FA_E_ALIGN;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n = (uint32_t)count, want = (n + VERIFY_WAVES - 1) / VERIFY_WAVES;
  if (logn == 9)
    hipLaunchKernelGGL(encode_pkeys_kernel<9>, dim3(stride_grid(ctx, encode_pkeys_kernel<9>, want)), dim3(VERIFY_WG), 0, s, h, pk, n);
  else
    hipLaunchKernelGGL(encode_pkeys_kernel<10>, dim3(stride_grid(ctx, encode_pkeys_kernel<10>, want)), dim3(VERIFY_WG), 0, s, h, pk, n);
  return (int)hipGetLastError();
}

int fa_encode_pkeys(fa_ctx* ctx, int logn, size_t count, const uint16_t* h, uint8_t* pk)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!h || !pk) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  const size_t in_bytes = (count << logn) * 2, out_bytes = count * fa_pkey_len(logn);
  int rc;
  if ((rc = ctx->h_a.reserve(in_bytes))) return rc;
  if ((rc = ctx->h_pk.reserve(out_bytes + 8))) return rc;
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->h_a.p, h, in_bytes, hipMemcpyHostToDevice, s));
  rc = fa_encode_pkeys_dev(ctx, logn, count, (const uint16_t*)ctx->h_a.p, (uint8_t*)ctx->h_pk.p, s);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(pk, ctx->h_pk.p, out_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int fa_compress_sigs_dev(fa_ctx* ctx, int logn, size_t count, const int16_t* s2, uint8_t* sig, uint8_t* ok, void* stream)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!s2 || !sig || !ok || count > 0xFFFFFFFFull) return FA_E_ARG;
  if ((uintptr_t)s2 & 15u) return FA_E_ALIGN;
  Guard g(ctx->device);
  if (g.err) return g.err;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n = (uint32_t)count;
  if (logn == 9)
    hipLaunchKernelGGL(compress_sigs_kernel<9>, dim3((n + 63) / 64), dim3(64), 0, s, s2, n, sig, ok);
  else
    hipLaunchKernelGGL(compress_sigs_kernel<10>, dim3((n + 63) / 64), dim3(64), 0, s, s2, n, sig, ok);
  return (int)hipGetLastError();
}

int fa_compress_sigs(fa_ctx* ctx, int logn, size_t count, const int16_t* s2, uint8_t* sig, uint8_t* ok)
{
  if (!ctx) return FA_E_ARG;
  if (logn != 9 && logn != 10) return FA_E_LOGN;
  if (count == 0) return 0;
  if (!s2 || !sig || !ok) return FA_E_ARG;
  Guard g(ctx->device);
  if (g.err) return g.err;
  const size_t in_bytes = (count << logn) * 2, sig_bytes = count * fa_sig_len(logn);
  int rc;
  if ((rc = ctx->h_a.reserve(in_bytes))) return rc;
  if ((rc = ctx->h_sig.reserve(sig_bytes + 16))) return rc;
  if ((rc = ctx->h_off.reserve(count))) return rc;
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->h_a.p, s2, in_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->h_sig.p, sig, sig_bytes, hipMemcpyHostToDevice, s));  // header and salt come along and go back untouched
  rc = fa_compress_sigs_dev(ctx, logn, count, (const int16_t*)ctx->h_a.p, (uint8_t*)ctx->h_sig.p, (uint8_t*)ctx->h_off.p, s);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(sig, ctx->h_sig.p, sig_bytes, hipMemcpyDeviceToHost, s));
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
