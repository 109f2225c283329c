"""Test-side bindings: the CPU oracle (oracle/) and the corpus generator (corpus/).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import the oracle
binding; the product package falcon_amd never does.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KINDS = ["valid", "bad_sig_header", "bad_pk_header", "neg_zero", "trailing_bits", "unary_past_end",
         "ends_at_last_bit", "k_ge_8_valid", "random_body", "bit_flip", "tampered_msg", "tampered_salt",
         "pk_coeff_ge_q"]
ST_OK, ST_BAD_PK_HEADER, ST_BAD_SIG_HEADER, ST_BAD_ENCODING, ST_NORM_TOO_LARGE = range(5)
BETA2 = {9: 34034726, 10: 70265242}


def pk_len(logn):
    return 1 + (14 << logn) // 8


def sig_len(logn):
    return 666 if logn == 9 else 1280


def _build(subdir, lib):
    """Load <subdir>/<lib>, running make first when the source is newer. Serialised by a file lock: the ranks of a
    multi-process bench all come through here at once."""
    import fcntl
    path = os.path.join(ROOT, subdir, lib)
    src = os.path.join(ROOT, subdir, lib[3:-3] + ".c")
    with open(os.path.join(ROOT, subdir, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, subdir)], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(path)


class OracleDebug(ctypes.Structure):
    _fields_ = [("c", ctypes.c_uint16 * 1024), ("s2", ctypes.c_int16 * 1024), ("s1", ctypes.c_int16 * 1024),
                ("norm_exact", ctypes.c_uint64), ("norm_wrap32", ctypes.c_uint32), ("status", ctypes.c_int32),
                ("hash_blocks", ctypes.c_int32)]


_DBG_DTYPE = np.dtype([("c", "<u2", 1024), ("s2", "<i2", 1024), ("s1", "<i2", 1024), ("norm_exact", "<u8"),
                       ("norm_wrap32", "<u4"), ("status", "<i4"), ("hash_blocks", "<i4")], align=True)

_u8p = ctypes.POINTER(ctypes.c_uint8)


def _p(a, t=ctypes.c_uint8):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


class Oracle:
    def __init__(self, lib=None):
        self.lib = lib = lib if lib is not None else _build("oracle", "libfalcon_oracle.so")
        lib.fo_debug_size.restype = ctypes.c_size_t
        assert lib.fo_debug_size() == ctypes.sizeof(OracleDebug) == _DBG_DTYPE.itemsize
        lib.fo_verify.argtypes = [ctypes.c_uint, _u8p, _u8p, ctypes.c_size_t, _u8p, ctypes.c_void_p]
        lib.fo_verify_many.argtypes = [ctypes.c_uint, ctypes.c_size_t, _u8p, _u8p, ctypes.POINTER(ctypes.c_uint64),
                                       ctypes.c_size_t, _u8p, _u8p, ctypes.c_void_p, ctypes.c_int]
        lib.fo_verify_many.restype = None
        lib.fo_shake256_xof.argtypes = [_u8p, ctypes.c_size_t, _u8p, ctypes.c_size_t]
        lib.fo_hash_to_point.argtypes = [ctypes.c_uint, _u8p, ctypes.c_size_t, _u8p, ctypes.c_size_t,
                                         ctypes.POINTER(ctypes.c_uint16)]
        lib.fo_decode_pkey.argtypes = [ctypes.c_uint, _u8p, ctypes.POINTER(ctypes.c_uint16)]
        lib.fo_decompress_sig.argtypes = [ctypes.c_uint, _u8p, ctypes.POINTER(ctypes.c_int32)]
        lib.fo_decode_sig.argtypes = [ctypes.c_uint, _u8p, _u8p, ctypes.POINTER(ctypes.c_int32)]
        lib.fo_encode_pkey.argtypes = [ctypes.c_uint, ctypes.POINTER(ctypes.c_uint16), _u8p]
        lib.fo_encode_pkey.restype = None
        lib.fo_compress_sig.argtypes = [ctypes.c_uint, ctypes.POINTER(ctypes.c_int32), _u8p]
        lib.fo_spec_decode_pkey.argtypes = [ctypes.c_uint, _u8p, ctypes.POINTER(ctypes.c_uint16)]
        lib.fo_spec_encode_pkey.argtypes = [ctypes.c_uint, ctypes.POINTER(ctypes.c_uint16), _u8p]
        lib.fo_spec_decompress_sig.argtypes = [ctypes.c_uint, _u8p, ctypes.POINTER(ctypes.c_int32)]
        lib.fo_spec_compress_sig.argtypes = [ctypes.c_uint, ctypes.POINTER(ctypes.c_int32), _u8p]
        lib.fo_verify_many_wire.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_size_t, _u8p, _u8p,
                                            ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t, _u8p, _u8p, ctypes.c_void_p,
                                            ctypes.c_int]
        for f in (lib.fo_ntt, lib.fo_intt):
            f.argtypes = [ctypes.POINTER(ctypes.c_uint16), ctypes.c_uint]
            f.restype = None
        lib.fo_compute_public_key.argtypes = [ctypes.POINTER(ctypes.c_int32)] * 2 + [ctypes.POINTER(ctypes.c_uint16),
                                                                                    ctypes.c_uint]
        lib.fo_compute_public_key.restype = None
        for f in (lib.fo_polymul, lib.fo_polymul_schoolbook):
            f.argtypes = [ctypes.POINTER(ctypes.c_uint16)] * 3 + [ctypes.c_uint]
            f.restype = None

    def shake256(self, data: bytes, outlen: int) -> bytes:
        src = np.frombuffer(data, dtype=np.uint8).copy() if data else np.zeros(1, np.uint8)
        out = np.zeros(outlen, np.uint8)
        self.lib.fo_shake256_xof(_p(src), len(data), _p(out), outlen)
        return out.tobytes()

    def hash_to_point(self, logn, salt, msg):
        salt = np.ascontiguousarray(salt, np.uint8)
        msg = np.ascontiguousarray(msg, np.uint8)
        m = msg if msg.size else np.zeros(1, np.uint8)
        c = np.zeros(1 << logn, np.uint16)
        blocks = self.lib.fo_hash_to_point(logn, _p(salt), salt.size, _p(m), msg.size, _p(c, ctypes.c_uint16))
        return c, blocks

    def decode_pkey(self, logn, pk):
        pk = np.ascontiguousarray(pk, np.uint8)
        h = np.zeros(1 << logn, np.uint16)
        ok = self.lib.fo_decode_pkey(logn, _p(pk), _p(h, ctypes.c_uint16))
        return bool(ok), h

    def decompress_sig(self, logn, sig):
        sig = np.ascontiguousarray(sig, np.uint8)
        s = np.zeros(1 << logn, np.int32)
        ok = self.lib.fo_decompress_sig(logn, _p(sig), _p(s, ctypes.c_int32))
        return bool(ok), s

    def decode_sig(self, logn, sig, salt=None, s2=None):
        """decoding::decode_sig; salt / s2 may be passed in to observe what a failing call leaves untouched."""
        sig = np.ascontiguousarray(sig, np.uint8)
        salt = np.zeros(40, np.uint8) if salt is None else salt
        s2 = np.zeros(1 << logn, np.int32) if s2 is None else s2
        ok = self.lib.fo_decode_sig(logn, _p(sig), _p(salt), _p(s2, ctypes.c_int32))
        return ok == 1, salt, s2  # 0: wrong header, -1: body does not decode

    def encode_pkey(self, logn, h):
        """encoding::encode_pkey (encoding.hpp:12-35)."""
        h = np.ascontiguousarray(h, np.uint16)
        pk = np.zeros(pk_len(logn), np.uint8)
        self.lib.fo_encode_pkey(logn, _p(h, ctypes.c_uint16), _p(pk))
        return pk

    def compress_sig(self, logn, s2, sig=None):
        """encoding::compress_sig (encoding.hpp:169-230): returns (ok, sig); bytes 0..40 of a passed-in sig are kept."""
        s2 = np.ascontiguousarray(s2, np.int32)
        sig = np.zeros(sig_len(logn), np.uint8) if sig is None else sig
        ok = self.lib.fo_compress_sig(logn, _p(s2, ctypes.c_int32), _p(sig))
        return bool(ok), sig

    # ---- specification wire format (no counterpart in the reference; oracle/falcon_oracle.c "Specification wire format") ----
    def spec_decode_pkey(self, logn, pk):
        pk = np.ascontiguousarray(pk, np.uint8)
        h = np.zeros(1 << logn, np.uint16)
        return bool(self.lib.fo_spec_decode_pkey(logn, _p(pk), _p(h, ctypes.c_uint16))), h

    def spec_encode_pkey(self, logn, h):
        h = np.ascontiguousarray(h, np.uint16)
        pk = np.zeros(pk_len(logn), np.uint8)
        self.lib.fo_spec_encode_pkey(logn, _p(h, ctypes.c_uint16), _p(pk))
        return pk

    def spec_decompress_sig(self, logn, sig):
        sig = np.ascontiguousarray(sig, np.uint8)
        s = np.zeros(1 << logn, np.int32)
        return bool(self.lib.fo_spec_decompress_sig(logn, _p(sig), _p(s, ctypes.c_int32))), s

    def spec_compress_sig(self, logn, s2, header_and_salt=None):
        """Returns (fits, sig): bytes 41.. hold the code; bytes 0..40 are header_and_salt when given."""
        s2 = np.ascontiguousarray(s2, np.int32)
        sig = np.zeros(sig_len(logn), np.uint8)
        ok = bool(self.lib.fo_spec_compress_sig(logn, _p(s2, ctypes.c_int32), _p(sig)))
        if header_and_salt is not None:
            sig[:41] = header_and_salt
        return ok, sig

    def ntt(self, a, logn):
        a = np.array(a, np.uint16)
        self.lib.fo_ntt(_p(a, ctypes.c_uint16), logn)
        return a

    def intt(self, a, logn):
        a = np.array(a, np.uint16)
        self.lib.fo_intt(_p(a, ctypes.c_uint16), logn)
        return a

    def polymul(self, a, b, logn, schoolbook=False):
        a = np.ascontiguousarray(a, np.uint16)
        b = np.ascontiguousarray(b, np.uint16)
        c = np.zeros(1 << logn, np.uint16)
        f = self.lib.fo_polymul_schoolbook if schoolbook else self.lib.fo_polymul
        f(_p(a, ctypes.c_uint16), _p(b, ctypes.c_uint16), _p(c, ctypes.c_uint16), logn)
        return c

    def compute_public_key(self, logn, f, g):
        f = np.ascontiguousarray(f, np.int32)
        g = np.ascontiguousarray(g, np.int32)
        h = np.zeros(1 << logn, np.uint16)
        self.lib.fo_compute_public_key(_p(f, ctypes.c_int32), _p(g, ctypes.c_int32), _p(h, ctypes.c_uint16), logn)
        return h

    def verify(self, logn, pk, msg, sig):
        """One item. Returns (flags, debug record): flags bit0 = exact64 accept, bit1 = wrap32 accept."""
        pk = np.ascontiguousarray(pk, np.uint8)
        sig = np.ascontiguousarray(sig, np.uint8)
        msg = np.ascontiguousarray(msg, np.uint8)
        m = msg if msg.size else np.zeros(1, np.uint8)
        d = np.zeros(1, _DBG_DTYPE)
        r = self.lib.fo_verify(logn, _p(pk), _p(m), msg.size, _p(sig), d.ctypes.data)
        return r, d[0]

    def verify_many(self, logn, pk, msg, sig, mlen=None, msg_off=None, debug=False, threads=1, wire=0):
        """Batch. pk/sig: (count, len) uint8; msg: (count, mlen) or flat bytes with msg_off[count+1].
        Returns flags (count,) uint8 [bit0 exact64, bit1 wrap32] and the debug records (or None).
        wire=1: the specification's wire format instead of the reference's."""
        pk = np.ascontiguousarray(pk, np.uint8)
        sig = np.ascontiguousarray(sig, np.uint8)
        msg = np.ascontiguousarray(msg, np.uint8)
        count = pk.shape[0]
        assert pk.shape == (count, pk_len(logn)) and sig.shape == (count, sig_len(logn))
        if msg_off is None:
            mlen = msg.shape[1] if mlen is None else mlen
            off = None
        else:
            off = np.ascontiguousarray(msg_off, np.uint64)
            mlen = 0
        m = msg if msg.size else np.zeros(1, np.uint8)
        flags = np.zeros(count, np.uint8)
        dbg = np.zeros(count, _DBG_DTYPE) if debug else None
        self.lib.fo_verify_many_wire(logn, wire, count, _p(pk), _p(m), _p(off, ctypes.c_uint64) if off is not None else None,
                                     mlen, _p(sig), _p(flags), dbg.ctypes.data if debug else None, threads)
        return flags, dbg


class Corpus:
    def __init__(self):
        self.lib = lib = _build("corpus", "libfalcon_corpus.so")
        lib.fc_generate.argtypes = [ctypes.c_uint, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_uint32,
                                    ctypes.c_uint32, ctypes.c_int, ctypes.c_int, _u8p, _u8p, _u8p, _u8p]

    def generate(self, logn, count, seed=1, first=0, mlen=32, malformed_permille=0, kind=-1, threads=8):
        """Returns pk (count, pklen), msg (count, mlen), sig (count, siglen), kind (count,) as uint8 arrays."""
        pk = np.zeros((count, pk_len(logn)), np.uint8)
        msg = np.zeros((count, max(mlen, 1)), np.uint8)
        sig = np.zeros((count, sig_len(logn)), np.uint8)
        kinds = np.zeros(count, np.uint8)
        rc = self.lib.fc_generate(logn, seed, first, count, mlen, malformed_permille, kind, threads,
                                  _p(pk), _p(msg), _p(sig), _p(kinds))
        assert rc == 0
        return pk, msg[:, :mlen], sig, kinds


class Reference:
    """oracle/_ref/libfalcon_ref.so: the reference's own headers behind C entry points (oracle/ref_harness.cpp),
    built in the container where /root/reference exists. Used only to pin the oracle and the HIP path."""

    PATH = os.path.join(ROOT, "oracle", "_ref", "libfalcon_ref.so")

    def __init__(self):
        self.lib = lib = ctypes.CDLL(self.PATH)
        u16p, i32p, i16p = (ctypes.POINTER(t) for t in (ctypes.c_uint16, ctypes.c_int32, ctypes.c_int16))
        lib.fr_decode_pkey.argtypes = [ctypes.c_uint, _u8p, u16p]
        lib.fr_decompress_sig.argtypes = [ctypes.c_uint, _u8p, i32p]
        lib.fr_decode_sig.argtypes = [ctypes.c_uint, _u8p, _u8p, i32p]
        lib.fr_encode_pkey.argtypes = [ctypes.c_uint, u16p, _u8p]
        lib.fr_compress_sig.argtypes = [ctypes.c_uint, i32p, _u8p]
        lib.fr_ntt.argtypes = [ctypes.c_uint, u16p]
        lib.fr_intt.argtypes = [ctypes.c_uint, u16p]
        lib.fr_ff_op.argtypes = [ctypes.c_int, ctypes.c_size_t, u16p, u16p, u16p]
        lib.fr_verify_after_hash.argtypes = [ctypes.c_uint, u16p, u16p, i32p, i16p]
        lib.fr_verify_after_hash.restype = ctypes.c_uint64

    def decode_pkey(self, logn, pk):
        pk = np.ascontiguousarray(pk, np.uint8)
        h = np.full(1 << logn, 0xBEEF, np.uint16)
        ok = self.lib.fr_decode_pkey(logn, _p(pk), _p(h, ctypes.c_uint16))
        return bool(ok), h

    def decompress_sig(self, logn, sig):
        sig = np.ascontiguousarray(sig, np.uint8)
        s = np.full(1 << logn, 77, np.int32)
        ok = self.lib.fr_decompress_sig(logn, _p(sig), _p(s, ctypes.c_int32))
        return bool(ok), s

    def decode_sig(self, logn, sig, salt=None, s2=None):
        sig = np.ascontiguousarray(sig, np.uint8)
        salt = np.zeros(40, np.uint8) if salt is None else salt
        s2 = np.zeros(1 << logn, np.int32) if s2 is None else s2
        ok = self.lib.fr_decode_sig(logn, _p(sig), _p(salt), _p(s2, ctypes.c_int32))
        return bool(ok), salt, s2

    def encode_pkey(self, logn, h):
        h = np.ascontiguousarray(h, np.uint16)
        pk = np.zeros(pk_len(logn), np.uint8)
        self.lib.fr_encode_pkey(logn, _p(h, ctypes.c_uint16), _p(pk))
        return pk

    def compress_sig(self, logn, s2):
        """Returns (ok, sig) with bytes 0..40 zero: the reference's compress_sig leaves header and salt alone."""
        s2 = np.ascontiguousarray(s2, np.int32)
        sig = np.zeros(sig_len(logn), np.uint8)
        ok = self.lib.fr_compress_sig(logn, _p(s2, ctypes.c_int32), _p(sig))
        return bool(ok), sig

    def ntt(self, a, logn, inverse=False):
        a = np.array(a, np.uint16)
        (self.lib.fr_intt if inverse else self.lib.fr_ntt)(logn, _p(a, ctypes.c_uint16))
        return a

    def ff_op(self, op, a, b):
        a = np.ascontiguousarray(a, np.uint16)
        b = np.ascontiguousarray(b, np.uint16)
        out = np.zeros_like(a)
        self.lib.fr_ff_op(op, a.size, _p(a, ctypes.c_uint16), _p(b, ctypes.c_uint16), _p(out, ctypes.c_uint16))
        return out

    def verify_after_hash(self, logn, h, c, s2):
        """verification.hpp:36-79 with the reference's functions. Returns (s1 centred int16, exact norm)."""
        h = np.ascontiguousarray(h, np.uint16)
        c = np.ascontiguousarray(c, np.uint16)
        s2 = np.ascontiguousarray(s2, np.int32)
        s1 = np.zeros(1 << logn, np.int16)
        norm = self.lib.fr_verify_after_hash(logn, _p(h, ctypes.c_uint16), _p(c, ctypes.c_uint16), _p(s2, ctypes.c_int32),
                                             _p(s1, ctypes.c_int16))
        return s1, int(norm)


def to_spec_format(logn, pk, sig, orc=None):
    """Re-encodes reference-format items in the specification's wire format: keys field by field (values >= q stay
    as they are, which the specification rejects), signatures that decode through Algorithm 17; signatures the
    reference decoder rejects keep their bytes (arbitrary input for the other decoder)."""
    orc = orc or oracle()
    spk, ssig = np.zeros_like(pk), sig.copy()
    for i in range(pk.shape[0]):
        ok, h = orc.decode_pkey(logn, pk[i])
        spk[i] = orc.spec_encode_pkey(logn, h) if ok else pk[i]
        ok, _, s2 = orc.decode_sig(logn, sig[i])
        if ok:
            fits, enc = orc.spec_compress_sig(logn, s2, sig[i][:41])
            assert fits
            ssig[i] = enc
    return spk, ssig


_oracle = None
_corpus = None
_reference = None


def reference():
    """The reference-backed library, or None where it has not been built (no /root/reference at build time)."""
    global _reference
    if _reference is None and os.path.exists(Reference.PATH):
        _reference = Reference()
    return _reference



def oracle():
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle


_oracle_native = None


def oracle_native():
    """The TIMED build of the oracle (oracle/Makefile target `native`: -O3 -march=native -DFO_NATIVE), compiled on the
    machine that runs it -- with gcc and with clang (the reference's own default compiler, Makefile:1), whichever
    verifies faster here; None where neither can build it. bench.py's cpu_baseline times this one; the checker
    everywhere else is oracle(). The chosen compiler is left in `.built_with`."""
    global _oracle_native
    if _oracle_native is None:
        import fcntl
        import shutil
        import time
        cands = []
        for name, cc in (("gcc", shutil.which("gcc")), ("clang", shutil.which("clang") or "/opt/rocm/lib/llvm/bin/clang")):
            if not cc or not os.path.exists(cc):
                continue
            path = os.path.join(ROOT, "oracle", "_native", f"libfalcon_oracle_native_{name}.so")
            try:
                with open(os.path.join(ROOT, "oracle", ".build.lock"), "w") as lock:
                    fcntl.flock(lock, fcntl.LOCK_EX)
                    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "native", f"CC={cc}",
                                           f"NATIVE_OUT=_native/libfalcon_oracle_native_{name}.so"],
                                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                o = Oracle(ctypes.CDLL(path))
            except (OSError, subprocess.CalledProcessError):
                continue
            pk, msg, sig, _ = corpus().generate(9, 1500, seed=1)
            o.verify_many(9, pk[:100], msg[:100], sig[:100])
            t0 = time.perf_counter()
            o.verify_many(9, pk, msg, sig)
            o.built_with = f"{name} ({cc}) -O3 -march=native -DFO_NATIVE"
            cands.append((time.perf_counter() - t0, o))
        if not cands:
            return None
        _oracle_native = min(cands, key=lambda c: c[0])[1]
    return _oracle_native


def corpus():
    global _corpus
    if _corpus is None:
        _corpus = Corpus()
    return _corpus
