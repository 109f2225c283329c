/*
 * falcon_amd.h -- C ABI of the MI355X batched Falcon-512/1024 verifier.
 *
 * The reference (itzmeanjan/falcon) is a header-only C++20 template library with no FFI layer;
 * its boundary for this path is the template API itself. Each entry point below names the
 * reference interface it stands in for (file:line under /root/reference). Conventions follow
 * the reference's: the caller owns every buffer, record lengths are implied by N
 * (utils.hpp:9-17,39-49: pk 897/1793 bytes, sig 666/1280 bytes), nothing is written except the
 * outputs, and there is no per-item error channel beyond the accept flag (debug path aside).
 *
 * Return value of every int function: 0 ok; < 0 invalid argument (FA_E_*); > 0 a hipError_t.
 * A context belongs to one GPU and is used by one host thread at a time; contexts on
 * different GPUs are independent (batches shard by item, no collective is involved).
 * Streams: the device-resident (_dev) entry points enqueue on the caller's stream and share the context's one
 * workspace. Calls on different streams are ordered by the library itself (each waits, on the device, for the
 * previous call's last use of the workspace), so they are safe but do not overlap; for concurrent batches on
 * one GPU use one context per stream.
 */
#ifndef FALCON_AMD_H
#define FALCON_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FA_E_ARG (-1)       /* null pointer / bad size */
#define FA_E_LOGN (-2)      /* logn is neither 9 nor 10 */
#define FA_E_ALIGN (-3)     /* a device array is not 4-byte aligned */
#define FA_E_INTERNAL (-4)
#define FA_E_UNRESOLVED (-5) /* asynchronous / allocation-free calls: items the device-side retry pass could not settle (see fa_finish) */
#define FA_E_TOO_MANY (-6)   /* fa_verify_many_dev_noalloc: count > FA_NOALLOC_MAX_ITEMS */

/* Longest message any entry point takes: the hash kernel counts the bytes of salt || message in 32-bit signed
 * integers (the reference takes a size_t mlen, hashing.hpp:16-21). Host entry points return FA_E_ARG beyond it; with
 * device-side offsets (msg_off on the GPU) the limit is the caller's to keep. */
#define FA_MAX_MSG_LEN 0x7fffff00u

/* Norm policy: verification.hpp:70-79 sums the squared norm in int32_t,
 * which overflows for most invalid signatures, so the reference's reject flag depends on its
 * compiler. EXACT64 = what the source means and what a clang++ build returns; WRAP32 = what a
 * g++ build returns ((int32_t)(norm mod 2^32) <= beta^2). Valid signatures agree under both. */
#define FA_NORM_EXACT64 0
#define FA_NORM_WRAP32 1

/* Per-item status on the debug path (precedence as in falcon.hpp:240-243, verification.hpp:31-34). */
#define FA_ST_OK 0
#define FA_ST_BAD_PK_HEADER 1
#define FA_ST_BAD_SIG_HEADER 2
#define FA_ST_BAD_ENCODING 3
#define FA_ST_NORM_TOO_LARGE 4

/* Options for fa_set_option. */
#define FA_OPT_CHUNK_ITEMS 1     /* items per pipeline pass (workspace size), default 2^20 */
#define FA_OPT_HASH_BLOCKS_512 2 /* SHAKE256 blocks squeezed per item up front, default 9 */
#define FA_OPT_HASH_BLOCKS_1024 3 /* default 17; items needing more are re-hashed with twice as many */
#define FA_OPT_PROFILE 4         /* 1: record HIP events around every kernel (see fa_get_profile) */
#define FA_OPT_OVERLAP 5         /* 1 (default): the decoder's fix-up pass runs on a second stream beside the hash kernel */
#define FA_OPT_HOST_CHUNK_ITEMS 6 /* items per sub-batch of the host-buffer pipeline (copy/compute overlap), default 2^17 */
#define FA_OPT_HOST_THREADS 7    /* host threads that stage pageable caller memory through pinned buffers, default 6; 0: the runtime's own path */

#define FA_OPT_WIRE_FORMAT 8     /* FA_WIRE_REFERENCE (default) or FA_WIRE_SPEC, see below */
#define FA_OPT_SMALL_BATCH_ITEMS 10 /* passes of at most this many items run decoder and hash waves in one kernel (latency), default 2^15; 0: never */
#define FA_OPT_GRID_PER_CU 9     /* workgroups per CU of the one-wave-per-item kernels; 0 (default): 12 x what the GPU holds at once */
#define FA_OPT_PAIR_HASH_ITEMS 12 /* of the small passes, those of at most this many items hash with one SHAKE256 sponge per lane pair (lower latency, half the items per wave), default 2^14; 0: never */
#define FA_OPT_TINY_BATCH_ITEMS 13 /* small passes of at most this many Falcon-512 items (half as many Falcon-1024 items) decode and hash with one WAVE per item (lowest latency), default 1024; 0: never */
#define FA_OPT_ASYNC 11          /* 1: the device-resident verify calls return without synchronising their stream, see fa_finish; default 0 */

/* Wire format of public keys and signatures, for every entry point that parses them (verify, prepare_keys from
 * encoded keys, decode_pkeys, decode_sigs). The reference is not wire compatible with the Falcon specification
 * (README.md:74-75); REFERENCE reproduces it bit for bit and is what every parity claim refers to. SPEC reads the
 * specification's formats instead (falcon-sign.info/falcon.pdf: 14-bit big-endian key fields, each < q; Algorithm 18
 * Decompress with |s| <= 2047; a code may end on the last bit) and centres s1 to [-6144, 6144] as the specification
 * does (the reference maps 6144 to -6145, verification.hpp:59-68). SPEC has no counterpart in the reference and no
 * known-answer vector was available to pin it: it is held only to this repository's own restatements. */
#define FA_WIRE_REFERENCE 0
#define FA_WIRE_SPEC 1

typedef struct fa_ctx fa_ctx;

/* Debug outputs: the intermediates of verification.hpp:20-80. Any pointer may be NULL. Arrays are
 * [count][N] (c, s2, s1) or [count]. For items that fail before hashing, c/s2/s1 are zero. */
typedef struct fa_debug {
  uint16_t* c;          /* hash_to_point output, hashing.hpp:16-50 */
  int16_t* s2;          /* decompressed signature polynomial, decoding.hpp:267-361 */
  int16_t* s1;          /* centred c - s2*h, verification.hpp:59-68 */
  uint64_t* norm_exact; /* sum s2^2 + sum s1^2 */
  uint32_t* norm_wrap32;/* the same mod 2^32 */
  uint8_t* status;      /* FA_ST_* */
} fa_debug;

typedef struct fa_profile {
  double ms_decode, ms_hash, ms_verify, ms_polymul; /* summed HIP-event time of each kernel */
  uint32_t launches_decode, launches_hash, launches_verify, launches_polymul;
  uint32_t retried_items;                            /* items re-hashed with more squeeze blocks */
} fa_profile;

/* Context per device: twiddle tables, workspace, a private stream. No reference counterpart (the
 * reference is stateless); the handle only owns device memory. */
int fa_ctx_create(int device, fa_ctx** out);
void fa_ctx_destroy(fa_ctx* ctx);
int fa_set_option(fa_ctx* ctx, int option, int64_t value);
int fa_get_profile(fa_ctx* ctx, fa_profile* out); /* synchronises; totals since the last call */
const char* fa_strerror(int code);

/* Batched falcon::verify<N> (falcon.hpp:230-249), host buffers. pk: count*pklen bytes, sig:
 * count*siglen bytes. Messages: msg_off == NULL -> fixed length mlen at msg + i*mlen; else message i
 * is msg[msg_off[i] .. msg_off[i+1]) (count+1 offsets) and mlen is ignored. accept[i] = 1/0 exactly
 * as the reference's bool. Inputs are copied to device storage; callers need not pad them. The batch is
 * processed in sub-batches of FA_OPT_HOST_CHUNK_ITEMS items, the copies of one overlapping the kernels of
 * the previous one; the call returns when everything is done. Caller memory may be pageable: it is then
 * staged through pinned buffers of the context (two sets of one sub-batch each) by FA_OPT_HOST_THREADS host
 * threads; pinned caller memory (fa_host_alloc / fa_host_register below) is copied from directly. */
int fa_verify_many(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, const uint8_t* msg,
                   const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig, int norm_policy,
                   uint8_t* accept, const fa_debug* dbg);

/* The same with every pointer (including those inside dbg) a device pointer on ctx's GPU; arrays
 * must be 4-byte aligned. Work is enqueued on `stream` (a hipStream_t, NULL = default stream); the
 * call synchronises that stream once before returning (it has to learn whether any item needs the
 * re-hash pass), unless FA_OPT_ASYNC is set (see fa_finish). This is the entry point the throughput figure
 * is measured on. */
int fa_verify_many_dev(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, const uint8_t* msg,
                       const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig, int norm_policy,
                       uint8_t* accept, const fa_debug* dbg, void* stream);

/* Asynchronous use (FA_OPT_ASYNC = 1): fa_verify_many_dev, fa_verify_many_prepared_dev and fa_verify_mixed_dev (the
 * latter after one early synchronisation to learn the class sizes) return as soon as their kernels are enqueued, so
 * a caller can queue batch after batch. The rare item whose rejection sampling needs more SHAKE256 blocks than were
 * squeezed up front (about 1e-25 / 1e-13 of Falcon-512 / Falcon-1024 items at the default block counts) is then not
 * settled by a host-side loop but by one more pass enqueued behind the batch, with twice the blocks, for at most
 * min(4096, half the pass size) such items. fa_finish synchronises `stream` and reports how many items of the calls
 * since the last fa_finish even that pass left unresolved (their accept byte is 0): returns 0 if none, else
 * FA_E_UNRESOLVED with the count in *unresolved (nullable) -- verify those batches again synchronously. The count is
 * per context: with asynchronous calls of one context in flight on several streams, an item is reported by the
 * fa_finish that follows its own pass, whichever stream that names. No reference counterpart: the reference is
 * synchronous by nature. */
int fa_finish(fa_ctx* ctx, void* stream, uint64_t* unresolved);

/* The same as fa_verify_many_dev for callers that cannot allocate device memory (tight serving loops, harnesses that
 * forbid hipMalloc): NO COUNTERPART IN THE REFERENCE. The pipeline's workspace is a static device array of the
 * library (about 18 MB, held by every GPU the library runs on until the process ends) instead of buffers of the
 * context, so the call makes no hipMalloc / hipFree and uses no allocating routine.
 * Limits: count <= FA_NOALLOC_MAX_ITEMS (else FA_E_TOO_MANY); reference wire format; at most 17 hash blocks per
 * item; no debug outputs. Items short of SHAKE256 blocks are settled by the device-side retry pass described at
 * fa_finish; the call synchronises `stream` once and returns FA_E_UNRESOLVED if any item is left (probability
 * ~1e-50 per item). Because of that synchronisation (and a host-side counter it reads) the call is NOT for use under
 * HIP stream capture. All calls on one GPU -- from any context -- share that GPU's array, which keeps the previous
 * call's contents (every call clears the counters it needs); the library takes a per-GPU lock for the duration of the
 * call, so calls from several host threads (each through its own context) are safe and run one after the other. */
#define FA_NOALLOC_MAX_ITEMS 4096
int fa_verify_many_dev_noalloc(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, const uint8_t* msg,
                               const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig, int norm_policy,
                               uint8_t* accept, void* stream);

/* ---- decoded-key form: verification::verify<N, beta2>(h, msg, mlen, sig), verification.hpp:20-26 ----
 * For callers that keep public keys decoded and verify many signatures per key (falcon.hpp:226-229,
 * benchmarks/bench_signing.cpp:105). A key is prepared once -- decoded and carried into the NTT domain,
 * N uint16 values in this library's own coefficient order and scaling (an opaque blob of fa_prepared_key_len bytes,
 * valid for any context of the same logn and the same build of the library) -- and items then name their key by index. Per item this saves
 * the key unpacking and one of the two forward transforms, and 897/1793 bytes of input.
 *
 * fa_prepare_keys: exactly one of `pk` (nkeys encoded keys, decode_pkey, decoding.hpp:18-48; a key with a
 * wrong header is marked and every item using it is rejected, as falcon.hpp:240-243 does) and `h` (nkeys
 * decoded polynomials, N uint16 each, natural order, any values: taken mod q as ff_t does) is non-NULL.
 * key_idx == NULL: item i uses key i (needs nkeys >= count). The host variant checks key_idx[i] < nkeys;
 * the device variant cannot and leaves that to the caller. */
size_t fa_prepared_key_len(int logn);
int fa_prepare_keys(fa_ctx* ctx, int logn, size_t nkeys, const uint8_t* pk, const uint16_t* h, uint16_t* prepared);
int fa_prepare_keys_dev(fa_ctx* ctx, int logn, size_t nkeys, const uint8_t* pk, const uint16_t* h, uint16_t* prepared,
                        void* stream); /* asynchronous on `stream` */
int fa_verify_many_prepared(fa_ctx* ctx, int logn, size_t count, const uint16_t* prepared, size_t nkeys,
                            const uint32_t* key_idx, const uint8_t* msg, const uint64_t* msg_off, uint32_t mlen,
                            const uint8_t* sig, int norm_policy, uint8_t* accept, const fa_debug* dbg);
int fa_verify_many_prepared_dev(fa_ctx* ctx, int logn, size_t count, const uint16_t* prepared, const uint32_t* key_idx,
                                const uint8_t* msg, const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig,
                                int norm_policy, uint8_t* accept, const fa_debug* dbg, void* stream);

/* ---- resident key sets: the decoded-key form for HOST callers (falcon.hpp:226-229, verification.hpp:20-26,
 * benchmarks/bench_signing.cpp:105: a caller that keeps `h` across calls) ----
 * fa_verify_many_prepared takes the prepared keys from caller memory and uploads them with every call -- N uint16 per
 * key, more bytes than the packed key it replaces. A key set keeps them in the GPU's memory instead: created once
 * from nkeys encoded keys (`pk`) or decoded polynomials (`h`), exactly one of the two as for fa_prepare_keys, it is
 * then named by any number of fa_verify_many_keyset calls, which send only signatures, messages and 4-byte key
 * indices over the host link (702 instead of 1,595 bytes per Falcon-512 item with a 32-byte message), through the same
 * two-set staging pipeline as fa_verify_many. key_idx == NULL: item i uses key i (needs nkeys >= count); every index is
 * checked against nkeys (FA_E_ARG). The set belongs to the GPU of the context that created it and may be used through
 * any context of that GPU (FA_E_ARG otherwise), from several threads at once (it is read-only); destroy it when no call
 * that names it is running. fa_keyset_device_keys: the device pointer of the prepared keys, for the device-resident
 * calls (fa_verify_many_prepared_dev). Wire format: the one in force in `ctx` when the set is created (keys) and when
 * it is used (signatures). */
typedef struct fa_keyset fa_keyset;
int fa_keyset_create(fa_ctx* ctx, int logn, size_t nkeys, const uint8_t* pk, const uint16_t* h, fa_keyset** out);
void fa_keyset_destroy(fa_keyset* ks);
size_t fa_keyset_size(const fa_keyset* ks);
int fa_keyset_logn(const fa_keyset* ks);
const uint16_t* fa_keyset_device_keys(const fa_keyset* ks);
int fa_verify_many_keyset(fa_ctx* ctx, const fa_keyset* ks, size_t count, const uint32_t* key_idx, const uint8_t* msg,
                          const uint64_t* msg_off, uint32_t mlen, const uint8_t* sig, int norm_policy, uint8_t* accept,
                          const fa_debug* dbg);

/* ---- hashing::hash_to_point<n>(salt, 40, msg, mlen, poly), hashing.hpp:16-50, as a batched primitive ----
 * (also the first step of signing, signing.hpp:48-49). salt: count*40 bytes; messages as in fa_verify_many;
 * c: count*N uint16, canonical residues in draw order. _dev: device pointers (4-byte aligned), work is
 * enqueued on `stream` and the call synchronises it once. */
int fa_hash_to_point_many(fa_ctx* ctx, int logn, size_t count, const uint8_t* salt, const uint8_t* msg,
                          const uint64_t* msg_off, uint32_t mlen, uint16_t* c);
int fa_hash_to_point_many_dev(fa_ctx* ctx, int logn, size_t count, const uint8_t* salt, const uint8_t* msg,
                              const uint64_t* msg_off, uint32_t mlen, uint16_t* c, void* stream);

/* ---- keygen::compute_public_key<N>(f, g, h), keygen.hpp:67-90: h = g * f^-1 in Z_q[x]/(x^n+1) ----
 * f, g: count*N int32 coefficients in [-6145, 6143] (the range the reference states, keygen.hpp:78-79);
 * h: count*N uint16, canonical. Where NTT(f) has a zero the quotient element is 0, as ff_t::inv gives
 * (ff.hpp:181-196); the reference's keygen never publishes such a key. _dev: asynchronous on `stream`. */
int fa_compute_public_keys(fa_ctx* ctx, int logn, size_t count, const int32_t* f, const int32_t* g, uint16_t* h);
int fa_compute_public_keys_dev(fa_ctx* ctx, int logn, size_t count, const int32_t* f, const int32_t* g, uint16_t* h,
                               void* stream);

/* Mixed Falcon-512 / Falcon-1024 batch, host buffers: logn[i] in {9,10}; item i's key, signature and
 * message are pk[pk_off[i]..], sig[sig_off[i]..], msg[msg_off[i]..msg_off[i+1]). The arrays are copied to the GPU
 * as they are (up to the last byte any record uses) and verified by fa_verify_mixed_dev. */
int fa_verify_mixed(fa_ctx* ctx, size_t count, const uint8_t* logn, const uint8_t* pk, const uint64_t* pk_off,
                    const uint8_t* msg, const uint64_t* msg_off, const uint8_t* sig, const uint64_t* sig_off,
                    int norm_policy, uint8_t* accept);

/* The same with every pointer a device pointer on ctx's GPU (falcon.hpp:230-249 serves both N from one template;
 * BASELINE.json configs[4]: the mixed 512+1024 batch). Records may sit at any byte offset; the offset arrays must
 * be 8-byte aligned. The item indices are split by N on the GPU (order-preserving), the call synchronises `stream`
 * once to learn the class sizes, then each class runs through the same passes as fa_verify_many_dev, reading its
 * records in place through the offsets -- nothing is gathered or copied. Reference wire format only. */
int fa_verify_mixed_dev(fa_ctx* ctx, size_t count, const uint8_t* logn, const uint8_t* pk, const uint64_t* pk_off,
                        const uint8_t* msg, const uint64_t* msg_off, const uint8_t* sig, const uint64_t* sig_off,
                        int norm_policy, uint8_t* accept, void* stream);

/* c[i] = a[i] * b[i] in Z_q[x]/(x^n+1), q = 12289, coefficients u16 (any value; taken mod q as the
 * ff_t constructor does, ff.hpp:69-76), results canonical. The composition ntt::ntt, polynomial::mul,
 * ntt::intt (ntt.hpp:120-200, polynomial.hpp:99-110) exercised by tests/test_ntt.cpp:48-57.
 * _dev: device pointers, asynchronous on `stream`. */
int fa_polymul_many(fa_ctx* ctx, int logn, size_t count, const uint16_t* a, const uint16_t* b, uint16_t* c);
int fa_polymul_many_dev(fa_ctx* ctx, int logn, size_t count, const uint16_t* a, const uint16_t* b, uint16_t* c,
                        void* stream);

/* ---- ntt::ntt<LOG2N>(poly) / ntt::intt<LOG2N>(poly), ntt.hpp:120-150 and :161-200, batched, out of place ----
 * a, out: count*N uint16. inverse == 0: natural order in, bit-reversed order out -- out[i][k] is element k of
 * the reference's array after ntt(); inverse != 0: the converse, with the 1/N scaling (ntt.hpp:193-199).
 * Inputs may be any uint16 (taken mod q as the ff_t constructor does, ff.hpp:69-76); results are canonical.
 * _dev: device pointers, asynchronous on `stream`. */
int fa_ntt_many(fa_ctx* ctx, int logn, size_t count, const uint16_t* a, uint16_t* out, int inverse);
int fa_ntt_many_dev(fa_ctx* ctx, int logn, size_t count, const uint16_t* a, uint16_t* out, int inverse, void* stream);

/* ---- decoding::decode_pkey<N>(pkey, h), decoding.hpp:18-48, batched ----
 * pk: count*pklen bytes; h: count*N uint16 = the raw 14-bit little-endian fields (up to 16383: the reference does
 * not range-check them); ok[i] = the reference's bool. A wrong header byte gives a zero row (decoding.hpp:29-32).
 * _dev: device pointers, asynchronous on `stream`. */
int fa_decode_pkeys(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, uint16_t* h, uint8_t* ok);
int fa_decode_pkeys_dev(fa_ctx* ctx, int logn, size_t count, const uint8_t* pk, uint16_t* h, uint8_t* ok, void* stream);

/* ---- decoding::decode_sig<N>(sig, salt, s2), decoding.hpp:369-391 (decompress_sig :267-361), batched ----
 * sig: count*siglen bytes; salt: count*40 bytes; s2: count*N int16 (the reference's int32_t values, all within
 * [-1023, 1023]); ok[i] = the reference's bool. Outputs are written exactly as far as the reference writes them:
 * a wrong header leaves salt[i] and s2[i] untouched (:380-382), a body that does not decode zeroes s2[i] and
 * leaves salt[i] untouched (:359, :385-387). _dev: device pointers (sig 4-byte aligned), asynchronous on
 * `stream`. */
int fa_decode_sigs(fa_ctx* ctx, int logn, size_t count, const uint8_t* sig, uint8_t* salt, int16_t* s2, uint8_t* ok);
int fa_decode_sigs_dev(fa_ctx* ctx, int logn, size_t count, const uint8_t* sig, uint8_t* salt, int16_t* s2, uint8_t* ok,
                       void* stream);

/* ---- encoding::encode_pkey<N>(h, pkey), encoding.hpp:12-35, batched (the inverse of decode_pkey) ----
 * h: count*N uint16, taken as they are (the reference packs h[i].v unmasked: a value >= 2^14 spills into the
 * neighbouring field exactly as its expressions make it); pk: count*pklen bytes, header byte log2 N.
 * _dev: device pointers (h 8-byte aligned), asynchronous on `stream`. */
int fa_encode_pkeys(fa_ctx* ctx, int logn, size_t count, const uint16_t* h, uint8_t* pk);
int fa_encode_pkeys_dev(fa_ctx* ctx, int logn, size_t count, const uint16_t* h, uint8_t* pk, void* stream);

/* ---- encoding::compress_sig<N, sbytelen>(poly_s, sig), encoding.hpp:169-230, batched (the inverse of decompress_sig) ----
 * s2: count*N int16 coefficients (the reference takes int32_t; the values a signer produces lie within [-2047, 2047]);
 * sig: count*siglen bytes of which only bytes 41.. are written, as the reference does (:164-165): header and salt are
 * the caller's; ok[i] = the reference's bool -- 0 when the code does not end before the last bit, and the body is then
 * all zeros (:226-229). A code longer than the reference's own 14 N / 8-byte work buffer (possible only for
 * coefficients far outside a signer's range) is undefined behaviour there; here it fails like any other code that is
 * too long. _dev: device pointers (s2 16-byte aligned), asynchronous on `stream`. */
int fa_compress_sigs(fa_ctx* ctx, int logn, size_t count, const int16_t* s2, uint8_t* sig, uint8_t* ok);
int fa_compress_sigs_dev(fa_ctx* ctx, int logn, size_t count, const int16_t* s2, uint8_t* sig, uint8_t* ok, void* stream);

/* Pinned host memory for the host-buffer entry points (no reference counterpart: the reference never
 * leaves the CPU). fa_host_alloc returns NULL on failure; fa_host_register pins an existing range
 * (page-granular, costly: worth it only for buffers that are reused; some hosts refuse it -- the MI355X boxes this
 * was measured on return hipErrorInvalidValue -- and the memory then simply stays pageable). */
void* fa_host_alloc(size_t bytes);
void fa_host_free(void* p);
int fa_host_register(void* p, size_t bytes);
int fa_host_unregister(void* p);

/* Record sizes, utils.hpp:9-17 and :39-49. */
size_t fa_pkey_len(int logn);
size_t fa_sig_len(int logn);

#ifdef __cplusplus
}
#endif
#endif /* FALCON_AMD_H */
