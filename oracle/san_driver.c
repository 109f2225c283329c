/*
 * san_driver.c -- TEST INFRASTRUCTURE: runs the CPU oracle and the corpus generator under
 * -fsanitize=address,undefined (tests/test_sanitizers.py, CPU only; SURVEY.md section 5 names this as the CPU-side
 * check: it is what exposed the int32 overflow of verification.hpp:76 and the over-read of decoding.hpp:217-218 in
 * the reference). Every record is handed over in a heap block of EXACTLY its size, so a read past a key or a
 * signature is an AddressSanitizer report.
 *
 * usage: san_driver <logn> <count> <mlen> <pk.bin> <msg.bin> <sig.bin>
 * prints a checksum of the results (accept flags, norms, decoded values) for the caller to compare with the
 * uninstrumented library.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* oracle/falcon_oracle.c */
typedef struct {
  uint16_t c[1024];
  int16_t s2[1024], s1[1024];
  uint64_t norm_exact;
  uint32_t norm_wrap32;
  int32_t status, hash_blocks;
} fo_debug_view;
size_t fo_debug_size(void);
int fo_verify(unsigned logn, const uint8_t* pkey, const uint8_t* msg, size_t mlen, const uint8_t* sig, void* dbg);
int fo_decode_pkey(unsigned logn, const uint8_t* pkey, uint16_t* h);
int fo_decompress_sig(unsigned logn, const uint8_t* sig, int32_t* poly_s);
int fo_decode_sig(unsigned logn, const uint8_t* sig, uint8_t* salt, int32_t* s2);
void fo_encode_pkey(unsigned logn, const uint16_t* h, uint8_t* pkey);
int fo_compress_sig(unsigned logn, const int32_t* poly_s, uint8_t* sig);
int fo_spec_decode_pkey(unsigned logn, const uint8_t* pkey, uint16_t* h);
int fo_spec_decompress_sig(unsigned logn, const uint8_t* sig, int32_t* poly_s);
/* corpus/falcon_corpus.c */
int fc_generate(unsigned logn, uint64_t seed, uint64_t first, size_t count, uint32_t mlen, uint32_t malformed_permille,
                int kind, int threads, uint8_t* pk, uint8_t* msg, uint8_t* sig, uint8_t* kinds);

static uint8_t* slurp(const char* path, size_t want)
{
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  uint8_t* p = (uint8_t*)malloc(want ? want : 1);
  if (fread(p, 1, want, f) != want) { fprintf(stderr, "%s: short file\n", path); exit(2); }
  fclose(f);
  return p;
}

static uint64_t mix(uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001b3ULL; }

int main(int argc, char** argv)
{
  if (argc != 7) { fprintf(stderr, "usage: %s logn count mlen pk.bin msg.bin sig.bin\n", argv[0]); return 2; }
  const unsigned logn = (unsigned)atoi(argv[1]);
  const size_t count = strtoull(argv[2], NULL, 10), mlen = strtoull(argv[3], NULL, 10), n = (size_t)1 << logn;
  const size_t pklen = 1 + 14 * n / 8, siglen = logn == 9 ? 666 : 1280;
  uint8_t* pk = slurp(argv[4], count * pklen);
  uint8_t* msg = slurp(argv[5], count * mlen);
  uint8_t* sig = slurp(argv[6], count * siglen);
  if (fo_debug_size() != sizeof(fo_debug_view)) { fprintf(stderr, "debug record layout\n"); return 2; }
  uint64_t sum = 0xcbf29ce484222325ULL;
  for (size_t i = 0; i < count; i++) {
    uint8_t* k = (uint8_t*)malloc(pklen);   /* exact sizes: an over-read is a report */
    uint8_t* s = (uint8_t*)malloc(siglen);
    uint8_t* m = (uint8_t*)malloc(mlen ? mlen : 1);
    memcpy(k, pk + i * pklen, pklen);
    memcpy(s, sig + i * siglen, siglen);
    memcpy(m, msg + i * mlen, mlen);
    fo_debug_view* d = (fo_debug_view*)malloc(sizeof(*d));
    const int flags = fo_verify(logn, k, m, mlen, s, d);
    sum = mix(mix(mix(sum, (uint64_t)flags), d->norm_exact), (uint64_t)d->status);
    uint16_t* h = (uint16_t*)malloc(2 * n);
    int32_t* s2 = (int32_t*)malloc(4 * n);
    uint8_t salt[40];
    sum = mix(sum, (uint64_t)fo_decode_pkey(logn, k, h));
    sum = mix(sum, (uint64_t)fo_spec_decode_pkey(logn, k, h));
    const int ok = fo_decode_sig(logn, s, salt, s2);
    sum = mix(sum, (uint64_t)ok);
    if (ok == 1) {  /* what decodes is encoded again (encoding.hpp:169-230) into exact-size buffers */
      uint8_t* back = (uint8_t*)calloc(siglen, 1);
      sum = mix(sum, (uint64_t)fo_compress_sig(logn, s2, back));
      for (size_t j = 41; j < siglen; j++) sum = mix(sum, back[j]);
      free(back);
      uint8_t* kb = (uint8_t*)malloc(pklen);
      if (fo_decode_pkey(logn, k, h)) {
        fo_encode_pkey(logn, h, kb);
        sum = mix(sum, kb[pklen - 1]);
      }
      free(kb);
    }
    sum = mix(sum, (uint64_t)fo_decompress_sig(logn, s, s2));
    sum = mix(sum, (uint64_t)fo_spec_decompress_sig(logn, s, s2));
    free(s2); free(h); free(d); free(m); free(s); free(k);
  }
  /* the corpus generator itself: every item class, two threads, exact-size outputs */
  for (int kind = 0; kind <= 12; kind++) {
    const size_t c = 24;
    uint8_t* gk = (uint8_t*)malloc(c * pklen);
    uint8_t* gs = (uint8_t*)malloc(c * siglen);
    uint8_t* gm = (uint8_t*)malloc(c * 33);
    uint8_t* kinds = (uint8_t*)malloc(c);
    if (fc_generate(logn, 77 + (uint64_t)kind, 5, c, 33, 0, kind, 2, gk, gm, gs, kinds)) { fprintf(stderr, "fc_generate failed\n"); return 3; }
    for (size_t i = 0; i < c; i++) sum = mix(sum, (uint64_t)fo_verify(logn, gk + i * pklen, gm + i * 33, 33, gs + i * siglen, NULL));
    free(kinds); free(gm); free(gs); free(gk);
  }
  printf("%016llx\n", (unsigned long long)sum);
  free(sig); free(msg); free(pk);
  return 0;
}
