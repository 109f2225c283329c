/*
 * falcon_corpus.c -- synthetic corpus generator (bench/test INPUT infrastructure).
 *
 * Produces (public key, message, signature) triples in the reference's own wire format
 * without running the reference's keygen/sign (GMP NTRUSolve + floating-point ffSampling stay
 * out of scope). A triple is *derived*: pick small s1, s2, a salt
 * and a message, compute c = hash_to_point(salt || msg) and publish h = (c - s1) * s2^-1 in
 * Z_q[x]/(x^n+1); then s1 + s2*h = c holds exactly, so the triple verifies with squared norm
 * |s1|^2 + |s2|^2, and a verifier cannot tell it from an honest signature.
 *
 * The signer-side encoders are restated from the reference:
 *   encode_pkey   encoding.hpp:12-35   (14-bit little-endian packing, header = log2 n)
 *   compress_sig  encoding.hpp:169-230 (sign, low 7 bits, unary high part; must end before
 *                                       the last bit of the buffer)
 *   signature layout signing.hpp:116-117 / decoding.hpp:259-261 (0x30|log2 n, 40-byte salt)
 *
 * This file is deliberately self-contained: it shares no code with oracle/ (the checker) nor
 * with falcon_amd/csrc (the product), so three independent implementations must agree.
 * Everything is counter-based on (seed, item index): any shard can be regenerated alone.
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define Q 12289u

/* item classes (config 4 of BASELINE.json: "malformed-signature mix") */
enum {
  FC_VALID = 0,
  FC_BAD_SIG_HEADER = 1,
  FC_BAD_PK_HEADER = 2,
  FC_NEG_ZERO = 3,
  FC_TRAILING_BITS = 4,
  FC_UNARY_PAST_END = 5,
  FC_ENDS_AT_LAST_BIT = 6,
  FC_K_GE_8_VALID = 7,    /* >= 8 unary zeros: decodes to the same s2 by the reference's rule */
  FC_RANDOM_BODY = 8,
  FC_BIT_FLIP = 9,
  FC_TAMPERED_MSG = 10,
  FC_TAMPERED_SALT = 11,
  FC_PK_COEFF_GE_Q = 12,  /* h[i] + q stored raw: still verifies in the reference */
  FC_NUM_KINDS = 13
};

/* ---------------- counter-based RNG: splitmix64 stream per item ---------------- */
typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t* r)
{
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline double rng_unit(rng_t* r) { return ((double)(rng_next(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
/* Box-Muller: two rounded normal deviates per call */
static void rng_gauss2(rng_t* r, double sigma, int* a, int* b)
{
  const double u = rng_unit(r), v = rng_unit(r);
  const double m = sigma * sqrt(-2.0 * log(u));
  *a = (int)lround(m * cos(6.283185307179586 * v));
  *b = (int)lround(m * sin(6.283185307179586 * v));
}

/* ---------------- Z_q helpers (plain % arithmetic) ---------------- */
static inline uint32_t mq_mul(uint32_t a, uint32_t b) { return (a * b) % Q; }
static uint32_t mq_pow(uint32_t b, uint32_t e)
{
  uint32_t r = 1;
  for (; e; e >>= 1, b = mq_mul(b, b)) if (e & 1) r = mq_mul(r, b);
  return r;
}

/* psi-power tables for a textbook negacyclic NTT (natural in, bit-reversed out) */
static uint32_t g_psi[2][1024], g_ipsi[2][1024];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static unsigned brv(unsigned v, unsigned bits)
{
  unsigned r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}
static void init_tables(void)
{
  for (int t = 0; t < 2; t++) {
    const unsigned logn = 9 + (unsigned)t, n = 1u << logn;
    const uint32_t psi = t == 0 ? 49 : 7; /* primitive 2n-th roots of unity mod q */
    const uint32_t ipsi = mq_pow(psi, Q - 2);
    for (unsigned i = 0; i < n; i++) {
      g_psi[t][i] = mq_pow(psi, brv(i, logn));
      g_ipsi[t][i] = mq_pow(ipsi, brv(i, logn));
    }
  }
}
static void ntt_fwd(uint32_t* a, unsigned logn)
{
  const unsigned n = 1u << logn;
  const uint32_t* w = g_psi[logn - 9];
  for (unsigned len = n >> 1, k = 1; len >= 1; len >>= 1)
    for (unsigned start = 0; start < n; start += 2 * len, k++)
      for (unsigned i = start; i < start + len; i++) {
        const uint32_t t = mq_mul(w[k], a[i + len]);
        a[i + len] = (a[i] + Q - t) % Q;
        a[i] = (a[i] + t) % Q;
      }
}
static void ntt_inv(uint32_t* a, unsigned logn)
{
  const unsigned n = 1u << logn;
  const uint32_t* w = g_ipsi[logn - 9];
  for (unsigned len = 1; len < n; len <<= 1) {
    unsigned k = n / (2 * len);
    for (unsigned start = 0; start < n; start += 2 * len, k++)
      for (unsigned i = start; i < start + len; i++) {
        const uint32_t u = a[i], v = a[i + len];
        a[i] = (u + v) % Q;
        a[i + len] = mq_mul(w[k], (u + Q - v) % Q);
      }
  }
  const uint32_t ninv = mq_pow(n, Q - 2);
  for (unsigned i = 0; i < n; i++) a[i] = mq_mul(a[i], ninv);
}

/* ---------------- SHAKE256 (FIPS 202), compact form ---------------- */
static const uint64_t RC[24] = {
  0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
  0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
  0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
  0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
  0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
  0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL };
static void keccak(uint64_t s[25])
{
  static const int piln[24] = { 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1 };
  static const int rotc[24] = { 1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44 };
  for (int r = 0; r < 24; r++) {
    uint64_t bc[5], t;
    for (int i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
    for (int i = 0; i < 5; i++) {
      t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
      for (int j = 0; j < 25; j += 5) s[j + i] ^= t;
    }
    t = s[1];
    for (int i = 0; i < 24; i++) {
      const int j = piln[i];
      const uint64_t b = s[j];
      s[j] = (t << rotc[i]) | (t >> (64 - rotc[i]));
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = s[j + i];
      for (int i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    s[0] ^= RC[r];
  }
}
/* hash-to-point, Falcon spec algorithm 3 as the reference implements it (hashing.hpp:16-50):
 * 16-bit big-endian draws, accept t < 61445, value t mod q. */
static void hash_to_point(unsigned logn, const uint8_t* salt, const uint8_t* msg, size_t mlen, uint32_t* c)
{
  const unsigned n = 1u << logn;
  uint64_t s[25];
  uint8_t blk[136];
  size_t pos = 0;
  memset(s, 0, sizeof(s));
  for (size_t i = 0; i < 40 + mlen; i++) {
    const uint8_t b = i < 40 ? salt[i] : msg[i - 40];
    s[pos >> 3] ^= (uint64_t)b << (8 * (pos & 7));
    if (++pos == 136) { keccak(s); pos = 0; }
  }
  s[pos >> 3] ^= (uint64_t)0x1F << (8 * (pos & 7));
  s[16] ^= 0x8000000000000000ULL;
  unsigned got = 0;
  while (got < n) {
    keccak(s);
    for (int i = 0; i < 136; i++) blk[i] = (uint8_t)(s[i >> 3] >> (8 * (i & 7)));
    for (int off = 0; off < 136 && got < n; off += 2) {
      const uint32_t t = ((uint32_t)blk[off] << 8) | blk[off + 1];
      if (t < 61445u) c[got++] = t % Q;
    }
  }
}

/* ---------------- encoders ---------------- */
size_t fc_pkey_len(unsigned logn) { return 1 + ((((size_t)1 << logn) * 14) / 8); }
size_t fc_sig_len(unsigned logn) { return logn == 9 ? 666 : 1280; }

/* encode_pkey, encoding.hpp:12-35. Values may exceed q (class FC_PK_COEFF_GE_Q); 14 bits kept. */
static void encode_pkey(unsigned logn, const uint32_t* h, uint8_t* pk)
{
  const unsigned n = 1u << logn;
  pk[0] = (uint8_t)logn;
  memset(pk + 1, 0, fc_pkey_len(logn) - 1);
  for (unsigned i = 0; i < n; i++) {           /* coefficient i sits at bit 14*i, LSB first */
    const size_t bit = 14u * i;
    const uint32_t v = (h[i] & 0x3fffu) << (bit & 7);
    pk[1 + (bit >> 3)] |= (uint8_t)v;
    pk[1 + (bit >> 3) + 1] |= (uint8_t)(v >> 8);
    if ((bit & 7) > 2) pk[1 + (bit >> 3) + 2] |= (uint8_t)(v >> 16);
  }
}

typedef struct { uint8_t* buf; size_t nbits, cap; } bitw_t; /* MSB-first bit writer */
static inline void bw_put(bitw_t* w, unsigned bit)
{
  if (w->nbits < w->cap && bit) w->buf[w->nbits >> 3] |= (uint8_t)(0x80u >> (w->nbits & 7));
  w->nbits++;
}
static void bw_coeff(bitw_t* w, int v, int quirk_k8)
{
  const unsigned a = (unsigned)abs(v);
  bw_put(w, v < 0);
  for (int b = 6; b >= 0; b--) bw_put(w, (a >> b) & 1u);
  if (quirk_k8) { /* 8 zeros, a window 00000001, then the bit the decoder skips (decoding.hpp:300-320,335) */
    for (int i = 0; i < 15; i++) bw_put(w, 0);
    bw_put(w, 1);
    bw_put(w, 1);
    return;
  }
  for (unsigned i = 0; i < (a >> 7); i++) bw_put(w, 0);
  bw_put(w, 1);
}
/* Writes the compressed body (bytes 41..) and returns the number of bits the code needs; bits
 * beyond the buffer are counted but not stored. quirk_at < 0: none. */
static size_t encode_body(unsigned logn, const int* s2, uint8_t* sig, int quirk_at)
{
  const unsigned n = 1u << logn;
  const size_t cap = 8 * (fc_sig_len(logn) - 41);
  bitw_t w = { sig + 41, 0, cap };
  memset(sig + 41, 0, cap / 8);
  for (unsigned i = 0; i < n; i++) bw_coeff(&w, s2[i], (int)i == quirk_at);
  return w.nbits;
}
/* compress_sig, encoding.hpp:169-230: returns bits used, or 0 if the code does not fit (the
 * reference requires bit_idx < slen and zeroes the body otherwise, :226-229). */
static size_t compress_sig(unsigned logn, const int* s2, uint8_t* sig, int quirk_at)
{
  const size_t cap = 8 * (fc_sig_len(logn) - 41);
  const size_t used = encode_body(logn, s2, sig, quirk_at);
  if (used >= cap) { memset(sig + 41, 0, cap / 8); return 0; }
  return used;
}

/* ---------------- one item ---------------- */
static void make_item(unsigned logn, uint64_t seed, uint64_t index, uint32_t mlen, int kind,
                      uint8_t* pk, uint8_t* msg, uint8_t* sig)
{
  const unsigned n = 1u << logn;
  const double sigma = logn == 9 ? 165.7366171829776 : 168.38857144654395;
  const uint64_t beta2 = logn == 9 ? 34034726u : 70265242u;
  const size_t siglen = fc_sig_len(logn), body_bits = 8 * (siglen - 41);
  rng_t r = { seed ^ (index * 0xD6E8FEB86659FD93ULL) ^ ((uint64_t)logn << 56) };
  int s1[1024], s2[1024];
  uint32_t c[1024], a[1024], b[1024];
  size_t used = 0;
  int quirk_at = -1;

  for (uint32_t i = 0; i < mlen; i++) msg[i] = (uint8_t)rng_next(&r);
  sig[0] = (uint8_t)(0x30 | logn);
  for (int i = 0; i < 40; i++) sig[1 + i] = (uint8_t)rng_next(&r);

  uint64_t norm1 = 0;
  for (unsigned i = 0; i < n; i += 2) {
    rng_gauss2(&r, sigma, &s1[i], &s1[i + 1]);
    norm1 += (uint64_t)(s1[i] * s1[i]) + (uint64_t)(s1[i + 1] * s1[i + 1]);
  }
  for (;;) { /* resample s2 until the norm bound holds, the encoding fits and s2 is invertible */
    uint64_t norm = norm1;
    for (unsigned i = 0; i < n; i += 2) rng_gauss2(&r, sigma, &s2[i], &s2[i + 1]);
    for (unsigned i = 0; i < n; i++) {
      if (s2[i] > 1023) s2[i] = 1023;
      if (s2[i] < -1023) s2[i] = -1023;
      norm += (uint64_t)(s2[i] * s2[i]);
    }
    if (norm > beta2) {
      if (norm1 > beta2 / 2 + beta2 / 8) { /* unlucky s1: redraw it too */
        norm1 = 0;
        for (unsigned i = 0; i < n; i += 2) {
          rng_gauss2(&r, sigma, &s1[i], &s1[i + 1]);
          norm1 += (uint64_t)(s1[i] * s1[i]) + (uint64_t)(s1[i + 1] * s1[i + 1]);
        }
      }
      continue;
    }
    if (kind == FC_K_GE_8_VALID) { /* the quirk keeps only the low 7 bits: pick such a coefficient */
      quirk_at = (int)(rng_next(&r) % n);
      s2[quirk_at] %= 128;
    }
    if (kind == FC_NEG_ZERO) s2[rng_next(&r) % n] = 0;
    used = compress_sig(logn, s2, sig, quirk_at);
    if (used == 0) continue;
    for (unsigned i = 0; i < n; i++) b[i] = (uint32_t)(s2[i] < 0 ? s2[i] + (int)Q : s2[i]);
    ntt_fwd(b, logn);
    int invertible = 1;
    for (unsigned i = 0; i < n; i++) invertible &= b[i] != 0;
    if (invertible) break;
  }

  hash_to_point(logn, sig + 1, msg, mlen, c);
  for (unsigned i = 0; i < n; i++) a[i] = (c[i] + Q - (uint32_t)(s1[i] < 0 ? s1[i] + (int)Q : s1[i])) % Q;
  ntt_fwd(a, logn);
  { /* a <- a / b pointwise, one inversion for all (Montgomery's trick) */
    uint32_t pre[1024], acc = 1;
    for (unsigned i = 0; i < n; i++) { pre[i] = acc; acc = mq_mul(acc, b[i]); }
    acc = mq_pow(acc, Q - 2);
    for (int i = (int)n - 1; i >= 0; i--) {
      a[i] = mq_mul(a[i], mq_mul(acc, pre[i]));
      acc = mq_mul(acc, b[i]);
    }
  }
  ntt_inv(a, logn); /* a = h */
  if (kind == FC_PK_COEFF_GE_Q)
    for (unsigned i = 0; i < n; i++) if (a[i] + Q <= 16383u && (rng_next(&r) & 1)) a[i] += Q;
  encode_pkey(logn, a, pk);

  uint8_t* body = sig + 41;
  switch (kind) {
    case FC_BAD_SIG_HEADER: sig[0] ^= (uint8_t)(1u << (rng_next(&r) % 8)); break;
    case FC_BAD_PK_HEADER: pk[0] ^= (uint8_t)(1u << (rng_next(&r) % 8)); break;
    case FC_NEG_ZERO: { /* set the sign bit of a zero coefficient (decoding.hpp:327-330) */
      size_t bit = 0;
      for (unsigned i = 0; i < n; i++) {
        if (s2[i] == 0) { body[bit >> 3] |= (uint8_t)(0x80u >> (bit & 7)); break; }
        bit += 9 + ((unsigned)abs(s2[i]) >> 7);
      }
      break;
    }
    case FC_TRAILING_BITS: { /* a 1 somewhere after the last coefficient (decoding.hpp:342-357) */
      const size_t bit = used + rng_next(&r) % (body_bits - used);
      body[bit >> 3] |= (uint8_t)(0x80u >> (bit & 7));
      break;
    }
    case FC_UNARY_PAST_END: { /* drop the final terminator: the unary run never ends */
      const size_t bit = used - 1;
      body[bit >> 3] &= (uint8_t)~(0x80u >> (bit & 7));
      break;
    }
    case FC_ENDS_AT_LAST_BIT: { /* lengthen unary runs (each stays <= 7) until the code fills the
                                   buffer exactly: rejected by decoding.hpp:340 */
      size_t need = body_bits - used;
      for (int i = (int)n - 1; i >= 0 && need; i--) {
        const unsigned k = (unsigned)abs(s2[i]) >> 7;
        const unsigned add = (7 - k) < need ? (7 - k) : (unsigned)need;
        s2[i] += (s2[i] < 0 ? -128 : 128) * (int)add;
        need -= add;
      }
      encode_body(logn, s2, sig, -1);
      break;
    }
    case FC_RANDOM_BODY:
      for (size_t i = 0; i < siglen - 41; i++) body[i] = (uint8_t)rng_next(&r);
      break;
    case FC_BIT_FLIP: {
      const size_t bit = rng_next(&r) % body_bits;
      body[bit >> 3] ^= (uint8_t)(0x80u >> (bit & 7));
      break;
    }
    case FC_TAMPERED_MSG:
      if (mlen) msg[rng_next(&r) % mlen] ^= (uint8_t)(1u << (rng_next(&r) % 8));
      else sig[1] ^= 1;
      break;
    case FC_TAMPERED_SALT: sig[1 + rng_next(&r) % 40] ^= (uint8_t)(1u << (rng_next(&r) % 8)); break;
    default: break;
  }
}

/* class of item `index`: valid unless its draw falls in the malformed share, then one of the
 * 12 malformed classes, evenly. */
static int pick_kind(uint64_t seed, uint64_t index, uint32_t malformed_permille)
{
  rng_t r = { seed ^ 0xA5A5A5A5DEADBEEFULL ^ (index * 0x9E3779B97F4A7C15ULL) };
  if (rng_next(&r) % 1000 >= malformed_permille) return FC_VALID;
  return 1 + (int)(rng_next(&r) % (FC_NUM_KINDS - 1));
}

typedef struct {
  unsigned logn; uint64_t seed, first; size_t begin, end; uint32_t mlen, permille; int force_kind;
  uint8_t *pk, *msg, *sig, *kind;
} job_t;

static void* worker(void* arg)
{
  job_t* j = (job_t*)arg;
  const size_t pklen = fc_pkey_len(j->logn), siglen = fc_sig_len(j->logn);
  for (size_t i = j->begin; i < j->end; i++) {
    const uint64_t idx = j->first + i;
    const int kind = j->force_kind >= 0 ? j->force_kind : pick_kind(j->seed, idx, j->permille);
    make_item(j->logn, j->seed, idx, j->mlen, kind, j->pk + i * pklen, j->msg + i * j->mlen, j->sig + i * siglen);
    if (j->kind) j->kind[i] = (uint8_t)kind;
  }
  return NULL;
}

/* Fill pk[count*pklen], msg[count*mlen], sig[count*siglen], kind[count] (may be NULL) with items
 * first_index .. first_index+count-1 of stream `seed`. force_kind >= 0 makes every item that
 * class. Returns 0, or -1 on bad arguments. */
int fc_generate(unsigned logn, uint64_t seed, uint64_t first_index, size_t count, uint32_t mlen,
                uint32_t malformed_permille, int force_kind, int threads, uint8_t* pk, uint8_t* msg,
                uint8_t* sig, uint8_t* kind)
{
  if ((logn != 9 && logn != 10) || force_kind >= FC_NUM_KINDS || malformed_permille > 1000) return -1;
  pthread_once(&g_once, init_tables);
  if (threads < 1) threads = 1;
  if ((size_t)threads > count) threads = count ? (int)count : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  job_t* jobs = (job_t*)malloc(sizeof(job_t) * (size_t)threads);
  for (int t = 0; t < threads; t++) {
    jobs[t] = (job_t){ logn, seed, first_index, count * (size_t)t / (size_t)threads,
                       count * (size_t)(t + 1) / (size_t)threads, mlen, malformed_permille, force_kind,
                       pk, msg, sig, kind };
    if (t > 0) pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  worker(&jobs[0]);
  for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
  return 0;
}
