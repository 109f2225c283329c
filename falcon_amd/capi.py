"""ctypes binding of the C ABI in include/falcon_amd.h (falcon_amd/libfalcon_amd.so).

There is no CPU fallback: if the HIP library is missing or no GPU is present, calls raise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FALCON_AMD_LIB") or os.path.join(_HERE, "libfalcon_amd.so")  # override: A/B builds only

NORM_EXACT64, NORM_WRAP32 = 0, 1
ST_OK, ST_BAD_PK_HEADER, ST_BAD_SIG_HEADER, ST_BAD_ENCODING, ST_NORM_TOO_LARGE = range(5)
OPT_CHUNK_ITEMS, OPT_HASH_BLOCKS_512, OPT_HASH_BLOCKS_1024, OPT_PROFILE, OPT_OVERLAP, OPT_HOST_CHUNK_ITEMS, OPT_HOST_THREADS, OPT_WIRE_FORMAT, OPT_GRID_PER_CU, OPT_SMALL_BATCH_ITEMS, OPT_ASYNC = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
OPT_PAIR_HASH_ITEMS = 12
OPT_TINY_BATCH_ITEMS = 13
E_ARG, E_LOGN, E_ALIGN, E_INTERNAL, E_UNRESOLVED, E_TOO_MANY = -1, -2, -3, -4, -5, -6
NOALLOC_MAX_ITEMS = 4096
MAX_MSG_LEN = 0x7fffff00
WIRE_REFERENCE, WIRE_SPEC = 0, 1

# every symbol include/falcon_amd.h declares
EXPORTS = ["fa_ctx_create", "fa_ctx_destroy", "fa_set_option", "fa_get_profile", "fa_strerror", "fa_verify_many",
           "fa_verify_many_dev", "fa_verify_mixed", "fa_polymul_many", "fa_polymul_many_dev", "fa_pkey_len",
           "fa_sig_len", "fa_prepared_key_len", "fa_prepare_keys", "fa_prepare_keys_dev", "fa_verify_many_prepared",
           "fa_verify_many_prepared_dev", "fa_hash_to_point_many", "fa_hash_to_point_many_dev", "fa_compute_public_keys",
           "fa_compute_public_keys_dev", "fa_host_alloc", "fa_host_free", "fa_host_register", "fa_host_unregister",
           "fa_ntt_many", "fa_ntt_many_dev", "fa_decode_pkeys", "fa_decode_pkeys_dev", "fa_decode_sigs", "fa_decode_sigs_dev",
           "fa_finish", "fa_verify_many_dev_noalloc", "fa_verify_mixed_dev",
           "fa_encode_pkeys", "fa_encode_pkeys_dev", "fa_compress_sigs", "fa_compress_sigs_dev",
           "fa_keyset_create", "fa_keyset_destroy", "fa_keyset_size", "fa_keyset_logn", "fa_keyset_device_keys",
           "fa_verify_many_keyset"]


class FaDebug(ctypes.Structure):
    _fields_ = [("c", ctypes.c_void_p), ("s2", ctypes.c_void_p), ("s1", ctypes.c_void_p),
                ("norm_exact", ctypes.c_void_p), ("norm_wrap32", ctypes.c_void_p), ("status", ctypes.c_void_p)]


class FaProfile(ctypes.Structure):
    _fields_ = [("ms_decode", ctypes.c_double), ("ms_hash", ctypes.c_double), ("ms_verify", ctypes.c_double),
                ("ms_polymul", ctypes.c_double), ("launches_decode", ctypes.c_uint32),
                ("launches_hash", ctypes.c_uint32), ("launches_verify", ctypes.c_uint32),
                ("launches_polymul", ctypes.c_uint32), ("retried_items", ctypes.c_uint32)]


class FalconAmdError(RuntimeError):
    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


_lib = None


def load_library():
    """Load libfalcon_amd.so; raises FalconAmdError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FalconAmdError(f"{LIB_PATH} not found: build it with falcon_amd/csrc/build.sh "
                             "(__graft_entry__.build()); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    vp, u64, u32, i32, sz = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_size_t
    lib.fa_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    lib.fa_ctx_destroy.argtypes = [vp]
    lib.fa_ctx_destroy.restype = None
    lib.fa_set_option.argtypes = [vp, i32, ctypes.c_int64]
    lib.fa_get_profile.argtypes = [vp, ctypes.POINTER(FaProfile)]
    lib.fa_strerror.argtypes = [i32]
    lib.fa_strerror.restype = ctypes.c_char_p
    lib.fa_verify_many.argtypes = [vp, i32, sz, vp, vp, vp, u32, vp, i32, vp, ctypes.POINTER(FaDebug)]
    lib.fa_verify_many_dev.argtypes = [vp, i32, sz, vp, vp, vp, u32, vp, i32, vp, ctypes.POINTER(FaDebug), vp]
    lib.fa_verify_mixed.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.fa_verify_mixed_dev.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    lib.fa_verify_many_dev_noalloc.argtypes = [vp, i32, sz, vp, vp, vp, u32, vp, i32, vp, vp]
    lib.fa_finish.argtypes = [vp, vp, ctypes.POINTER(u64)]
    lib.fa_polymul_many.argtypes = [vp, i32, sz, vp, vp, vp]
    lib.fa_polymul_many_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp]
    lib.fa_prepared_key_len.argtypes = [i32]
    lib.fa_prepared_key_len.restype = sz
    lib.fa_prepare_keys.argtypes = [vp, i32, sz, vp, vp, vp]
    lib.fa_prepare_keys_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp]
    lib.fa_verify_many_prepared.argtypes = [vp, i32, sz, vp, sz, vp, vp, vp, u32, vp, i32, vp, ctypes.POINTER(FaDebug)]
    lib.fa_verify_many_prepared_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp, u32, vp, i32, vp, ctypes.POINTER(FaDebug), vp]
    lib.fa_keyset_create.argtypes = [vp, i32, sz, vp, vp, ctypes.POINTER(vp)]
    lib.fa_keyset_destroy.argtypes = [vp]
    lib.fa_keyset_destroy.restype = None
    lib.fa_keyset_size.argtypes = [vp]
    lib.fa_keyset_size.restype = sz
    lib.fa_keyset_logn.argtypes = [vp]
    lib.fa_keyset_device_keys.argtypes = [vp]
    lib.fa_keyset_device_keys.restype = vp
    lib.fa_verify_many_keyset.argtypes = [vp, vp, sz, vp, vp, vp, u32, vp, i32, vp, ctypes.POINTER(FaDebug)]
    lib.fa_hash_to_point_many.argtypes = [vp, i32, sz, vp, vp, vp, u32, vp]
    lib.fa_hash_to_point_many_dev.argtypes = [vp, i32, sz, vp, vp, vp, u32, vp, vp]
    lib.fa_compute_public_keys.argtypes = [vp, i32, sz, vp, vp, vp]
    lib.fa_compute_public_keys_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp]
    lib.fa_ntt_many.argtypes = [vp, i32, sz, vp, vp, i32]
    lib.fa_ntt_many_dev.argtypes = [vp, i32, sz, vp, vp, i32, vp]
    lib.fa_decode_pkeys.argtypes = [vp, i32, sz, vp, vp, vp]
    lib.fa_decode_pkeys_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp]
    lib.fa_decode_sigs.argtypes = [vp, i32, sz, vp, vp, vp, vp]
    lib.fa_decode_sigs_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp, vp]
    lib.fa_encode_pkeys.argtypes = [vp, i32, sz, vp, vp]
    lib.fa_encode_pkeys_dev.argtypes = [vp, i32, sz, vp, vp, vp]
    lib.fa_compress_sigs.argtypes = [vp, i32, sz, vp, vp, vp]
    lib.fa_compress_sigs_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp]
    lib.fa_host_alloc.argtypes = [sz]
    lib.fa_host_alloc.restype = vp
    lib.fa_host_free.argtypes = [vp]
    lib.fa_host_free.restype = None
    lib.fa_host_register.argtypes = [vp, sz]
    lib.fa_host_unregister.argtypes = [vp]
    lib.fa_pkey_len.argtypes = [i32]
    lib.fa_pkey_len.restype = sz
    lib.fa_sig_len.argtypes = [i32]
    lib.fa_sig_len.restype = sz
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().fa_strerror(rc).decode()
        raise FalconAmdError(f"{what} failed: {msg} (code {rc})", rc)


def _ptr(a):
    return a.ctypes.data if a is not None else None


_DUMMY = np.zeros(8, np.uint8)  # stands in for an empty message array (module-level: its address stays valid)


def _check_offsets(off, nbytes, what, rec_len=None):
    """off: monotone offsets that stay inside an array of nbytes bytes (rec_len: fixed record length per entry,
    else the last entry is the end of the last record)."""
    if off.size == 0:
        return
    if rec_len is None:
        if (np.diff(off.astype(np.int64)) < 0).any() or int(off[-1]) > nbytes:
            raise ValueError(f"{what}: offsets must be non-decreasing and end within the array")
    else:
        if (off.astype(np.int64) + np.asarray(rec_len, np.int64) > nbytes).any():
            raise ValueError(f"{what}: a record extends past the end of the array")


class PinnedArray:
    """A numpy array in pinned host memory (fa_host_alloc); `.array` is the view, close() frees it."""

    def __init__(self, shape, dtype=np.uint8):
        self.lib = load_library()
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = self.lib.fa_host_alloc(max(nbytes, 1))
        if not self.ptr:
            raise FalconAmdError(f"fa_host_alloc({nbytes}) failed")
        buf = (ctypes.c_uint8 * max(nbytes, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.lib.fa_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Keyset:
    """A set of public keys resident on one GPU in prepared form (fa_keyset); close() frees it."""

    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle

    def __len__(self):
        return int(self.lib.fa_keyset_size(self.handle))

    @property
    def logn(self):
        return int(self.lib.fa_keyset_logn(self.handle))

    @property
    def device_ptr(self):
        """Device address of the prepared keys (for verify_many_prepared_dev)."""
        return self.lib.fa_keyset_device_keys(self.handle)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.fa_keyset_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU's verifier context (fa_ctx)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        _check(self.lib.fa_ctx_create(device, ctypes.byref(h)), "fa_ctx_create")
        self.handle = h
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            self.lib.fa_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt, value):
        _check(self.lib.fa_set_option(self.handle, opt, value), "fa_set_option")

    def get_profile(self):
        p = FaProfile()
        _check(self.lib.fa_get_profile(self.handle, ctypes.byref(p)), "fa_get_profile")
        return {k: getattr(p, k) for k, _ in FaProfile._fields_}

    # ---- host-buffer entry points (numpy arrays) ----
    def verify_many(self, logn, pk, msg, sig, msg_off=None, norm_policy=NORM_EXACT64, debug=False, accept_out=None):
        """pk (count, pklen) u8, sig (count, siglen) u8; msg (count, mlen) u8, or flat bytes with
        msg_off (count+1) u64. Returns accept (count,) u8 and, with debug=True, a dict of intermediates."""
        n = 1 << logn
        pk = np.ascontiguousarray(pk, np.uint8)
        sig = np.ascontiguousarray(sig, np.uint8)
        msg = np.ascontiguousarray(msg, np.uint8)
        count = pk.shape[0]
        if pk.shape != (count, self.lib.fa_pkey_len(logn)) or sig.shape != (count, self.lib.fa_sig_len(logn)):
            raise ValueError("pk/sig shape does not match logn")
        if msg_off is None:
            if msg.ndim != 2 or msg.shape[0] != count:
                raise ValueError("msg must be (count, mlen) when msg_off is None")
            mlen, off = msg.shape[1], None
        else:
            off = np.ascontiguousarray(msg_off, np.uint64)
            if off.shape != (count + 1,):
                raise ValueError("msg_off must hold count+1 offsets")
            mlen = 0
        accept = np.zeros(count, np.uint8) if accept_out is None else accept_out  # e.g. a PinnedArray's view
        if accept.dtype != np.uint8 or accept.shape != (count,) or not accept.flags.c_contiguous:
            raise ValueError("accept_out must be a contiguous (count,) uint8 array")
        dbg = None
        out = None
        if debug:
            out = {"c": np.zeros((count, n), np.uint16), "s2": np.zeros((count, n), np.int16),
                   "s1": np.zeros((count, n), np.int16), "norm_exact": np.zeros(count, np.uint64),
                   "norm_wrap32": np.zeros(count, np.uint32), "status": np.zeros(count, np.uint8)}
            dbg = FaDebug(*[_ptr(out[k]) for k, _ in FaDebug._fields_])
        if off is not None:
            _check_offsets(off, msg.size, "msg_off")
        mp = _ptr(msg) if msg.size else (None if off is None and mlen == 0 else _ptr(_DUMMY))
        rc = self.lib.fa_verify_many(self.handle, logn, count, _ptr(pk), mp, _ptr(off), mlen, _ptr(sig), norm_policy,
                                     _ptr(accept), ctypes.byref(dbg) if dbg is not None else None)
        _check(rc, "fa_verify_many")
        return (accept, out) if debug else accept

    def _msg_args(self, msg, msg_off, count):
        msg = np.ascontiguousarray(msg, np.uint8)
        if msg_off is None:
            if msg.ndim != 2 or msg.shape[0] != count:
                raise ValueError("msg must be (count, mlen) when msg_off is None")
            mlen, off = msg.shape[1], None
        else:
            off = np.ascontiguousarray(msg_off, np.uint64)
            if off.shape != (count + 1,):
                raise ValueError("msg_off must hold count+1 offsets")
            _check_offsets(off, msg.size, "msg_off")
            mlen = 0
        mp = _ptr(msg) if msg.size else (None if off is None and mlen == 0 else _ptr(_DUMMY))
        return msg, off, mlen, mp

    def prepare_keys(self, logn, pk=None, h=None):
        """Prepared keys (NTT(h), opaque order) from encoded keys `pk` (nkeys, pklen) or decoded
        polynomials `h` (nkeys, N) uint16 -- verification::verify's decoded-key form."""
        if (pk is None) == (h is None):
            raise ValueError("give exactly one of pk, h")
        src = np.ascontiguousarray(pk, np.uint8) if pk is not None else np.ascontiguousarray(h, np.uint16)
        nkeys = src.shape[0]
        want = self.lib.fa_pkey_len(logn) if pk is not None else (1 << logn)
        if src.shape != (nkeys, want):
            raise ValueError("key array shape does not match logn")
        out = np.zeros((nkeys, 1 << logn), np.uint16)
        rc = self.lib.fa_prepare_keys(self.handle, logn, nkeys, _ptr(src) if pk is not None else None,
                                      _ptr(src) if pk is None else None, _ptr(out))
        _check(rc, "fa_prepare_keys")
        return out

    def verify_many_prepared(self, logn, prepared, msg, sig, key_idx=None, msg_off=None, norm_policy=NORM_EXACT64,
                             debug=False):
        n = 1 << logn
        prepared = np.ascontiguousarray(prepared, np.uint16)
        sig = np.ascontiguousarray(sig, np.uint8)
        count = sig.shape[0]
        if prepared.ndim != 2 or prepared.shape[1] != n or sig.shape != (count, self.lib.fa_sig_len(logn)):
            raise ValueError("prepared/sig shape does not match logn")
        kidx = None if key_idx is None else np.ascontiguousarray(key_idx, np.uint32)
        if kidx is not None and kidx.shape != (count,):
            raise ValueError("key_idx must hold count entries")
        msg, off, mlen, mp = self._msg_args(msg, msg_off, count)
        accept = np.zeros(count, np.uint8)
        dbg = out = None
        if debug:
            out = {"c": np.zeros((count, n), np.uint16), "s2": np.zeros((count, n), np.int16),
                   "s1": np.zeros((count, n), np.int16), "norm_exact": np.zeros(count, np.uint64),
                   "norm_wrap32": np.zeros(count, np.uint32), "status": np.zeros(count, np.uint8)}
            dbg = FaDebug(*[_ptr(out[k]) for k, _ in FaDebug._fields_])
        rc = self.lib.fa_verify_many_prepared(self.handle, logn, count, _ptr(prepared), prepared.shape[0], _ptr(kidx), mp,
                                              _ptr(off), mlen, _ptr(sig), norm_policy, _ptr(accept),
                                              ctypes.byref(dbg) if dbg is not None else None)
        _check(rc, "fa_verify_many_prepared")
        return (accept, out) if debug else accept

    def keyset_create(self, logn, pk=None, h=None):
        """fa_keyset_create: the keys stay on the GPU (prepared form); returns a Keyset."""
        if (pk is None) == (h is None):
            raise ValueError("give exactly one of pk, h")
        src = np.ascontiguousarray(pk, np.uint8) if pk is not None else np.ascontiguousarray(h, np.uint16)
        nkeys = src.shape[0]
        want = self.lib.fa_pkey_len(logn) if pk is not None else (1 << logn)
        if src.ndim != 2 or src.shape != (nkeys, want):
            raise ValueError("key array shape does not match logn")
        out = ctypes.c_void_p()
        rc = self.lib.fa_keyset_create(self.handle, logn, nkeys, _ptr(src) if pk is not None else None,
                                       _ptr(src) if pk is None else None, ctypes.byref(out))
        _check(rc, "fa_keyset_create")
        return Keyset(self.lib, out)

    def verify_many_keyset(self, keyset, msg, sig, key_idx=None, msg_off=None, norm_policy=NORM_EXACT64, debug=False,
                           accept_out=None):
        """fa_verify_many_keyset: host arrays of signatures, messages and key indices against a resident Keyset."""
        logn = keyset.logn
        n = 1 << logn
        sig = np.ascontiguousarray(sig, np.uint8)
        count = sig.shape[0]
        if sig.shape != (count, self.lib.fa_sig_len(logn)):
            raise ValueError("sig shape does not match the key set's logn")
        kidx = None if key_idx is None else np.ascontiguousarray(key_idx, np.uint32)
        if kidx is not None and kidx.shape != (count,):
            raise ValueError("key_idx must hold count entries")
        msg, off, mlen, mp = self._msg_args(msg, msg_off, count)
        accept = np.zeros(count, np.uint8) if accept_out is None else accept_out
        if accept.dtype != np.uint8 or accept.shape != (count,) or not accept.flags.c_contiguous:
            raise ValueError("accept_out must be a contiguous (count,) uint8 array")
        dbg = out = None
        if debug:
            out = {"c": np.zeros((count, n), np.uint16), "s2": np.zeros((count, n), np.int16),
                   "s1": np.zeros((count, n), np.int16), "norm_exact": np.zeros(count, np.uint64),
                   "norm_wrap32": np.zeros(count, np.uint32), "status": np.zeros(count, np.uint8)}
            dbg = FaDebug(*[_ptr(out[k]) for k, _ in FaDebug._fields_])
        rc = self.lib.fa_verify_many_keyset(self.handle, keyset.handle, count, _ptr(kidx), mp, _ptr(off), mlen, _ptr(sig),
                                            norm_policy, _ptr(accept), ctypes.byref(dbg) if dbg is not None else None)
        _check(rc, "fa_verify_many_keyset")
        return (accept, out) if debug else accept

    def hash_to_point_many(self, logn, salt, msg, msg_off=None):
        """hashing::hash_to_point for (count, 40) salts and messages -> (count, N) uint16."""
        salt = np.ascontiguousarray(salt, np.uint8)
        count = salt.shape[0]
        if salt.shape != (count, 40):
            raise ValueError("salt must be (count, 40)")
        msg, off, mlen, mp = self._msg_args(msg, msg_off, count)
        c = np.zeros((count, 1 << logn), np.uint16)
        _check(self.lib.fa_hash_to_point_many(self.handle, logn, count, _ptr(salt), mp, _ptr(off), mlen, _ptr(c)),
               "fa_hash_to_point_many")
        return c

    def compute_public_keys(self, logn, f, g):
        """keygen::compute_public_key: h = g / f mod (x^n + 1, q) for (count, N) int32 arrays f, g."""
        f = np.ascontiguousarray(f, np.int32)
        g = np.ascontiguousarray(g, np.int32)
        count = f.shape[0]
        if f.shape != (count, 1 << logn) or g.shape != f.shape:
            raise ValueError("f/g must be (count, n)")
        h = np.zeros(f.shape, np.uint16)
        _check(self.lib.fa_compute_public_keys(self.handle, logn, count, _ptr(f), _ptr(g), _ptr(h)), "fa_compute_public_keys")
        return h

    def verify_mixed(self, logn, pk, pk_off, msg, msg_off, sig, sig_off, norm_policy=NORM_EXACT64):
        logn = np.ascontiguousarray(logn, np.uint8)
        count = logn.shape[0]
        arrs = [np.ascontiguousarray(pk, np.uint8), np.ascontiguousarray(pk_off, np.uint64),
                np.ascontiguousarray(msg, np.uint8), np.ascontiguousarray(msg_off, np.uint64),
                np.ascontiguousarray(sig, np.uint8), np.ascontiguousarray(sig_off, np.uint64)]
        if arrs[1].shape != (count,) or arrs[5].shape != (count,) or arrs[3].shape != (count + 1,):
            raise ValueError("pk_off / sig_off hold count offsets, msg_off count + 1")
        n1024 = logn == 10
        _check_offsets(arrs[1], arrs[0].size, "pk_off", np.where(n1024, 1793, 897))
        _check_offsets(arrs[5], arrs[4].size, "sig_off", np.where(n1024, 1280, 666))
        _check_offsets(arrs[3], arrs[2].size, "msg_off")
        if arrs[2].size == 0:
            arrs[2] = _DUMMY
        accept = np.zeros(count, np.uint8)
        rc = self.lib.fa_verify_mixed(self.handle, count, _ptr(logn), *[_ptr(a) for a in arrs], norm_policy,
                                      _ptr(accept))
        _check(rc, "fa_verify_mixed")
        return accept

    def polymul_many(self, logn, a, b):
        a = np.ascontiguousarray(a, np.uint16)
        b = np.ascontiguousarray(b, np.uint16)
        count = a.shape[0]
        if a.shape != (count, 1 << logn) or b.shape != a.shape:
            raise ValueError("a/b must be (count, n)")
        c = np.zeros_like(a)
        _check(self.lib.fa_polymul_many(self.handle, logn, count, _ptr(a), _ptr(b), _ptr(c)), "fa_polymul_many")
        return c

    def ntt_many(self, logn, a, inverse=False):
        """ntt::ntt / ntt::intt (ntt.hpp:120-200) on every row of a (count, n) u16."""
        a = np.ascontiguousarray(a, np.uint16)
        count = a.shape[0]
        if a.shape != (count, 1 << logn):
            raise ValueError("a must be (count, n)")
        out = np.zeros_like(a)
        _check(self.lib.fa_ntt_many(self.handle, logn, count, _ptr(a), _ptr(out), 1 if inverse else 0), "fa_ntt_many")
        return out

    def decode_pkeys(self, logn, pk):
        """decoding::decode_pkey (decoding.hpp:18-48): returns (h (count, n) u16, ok (count,) u8)."""
        pk = np.ascontiguousarray(pk, np.uint8)
        count = pk.shape[0]
        if pk.shape != (count, self.lib.fa_pkey_len(logn)):
            raise ValueError("pk shape does not match logn")
        h = np.zeros((count, 1 << logn), np.uint16)
        ok = np.zeros(count, np.uint8)
        _check(self.lib.fa_decode_pkeys(self.handle, logn, count, _ptr(pk), _ptr(h), _ptr(ok)), "fa_decode_pkeys")
        return h, ok

    def decode_sigs(self, logn, sig, salt=None, s2=None):
        """decoding::decode_sig (decoding.hpp:369-391): returns (salt (count, 40) u8, s2 (count, n) i16, ok). salt / s2
        may be passed in to observe which rows the call leaves untouched (as the reference does on failure)."""
        sig = np.ascontiguousarray(sig, np.uint8)
        count = sig.shape[0]
        if sig.shape != (count, self.lib.fa_sig_len(logn)):
            raise ValueError("sig shape does not match logn")
        salt = np.zeros((count, 40), np.uint8) if salt is None else salt
        s2 = np.zeros((count, 1 << logn), np.int16) if s2 is None else s2
        if salt.dtype != np.uint8 or salt.shape != (count, 40) or not salt.flags.c_contiguous:
            raise ValueError("salt must be a contiguous (count, 40) uint8 array")
        if s2.dtype != np.int16 or s2.shape != (count, 1 << logn) or not s2.flags.c_contiguous:
            raise ValueError("s2 must be a contiguous (count, n) int16 array")
        ok = np.zeros(count, np.uint8)
        _check(self.lib.fa_decode_sigs(self.handle, logn, count, _ptr(sig), _ptr(salt), _ptr(s2), _ptr(ok)), "fa_decode_sigs")
        return salt, s2, ok

    def encode_pkeys(self, logn, h):
        """encoding::encode_pkey (encoding.hpp:12-35) for every row of h (count, n) u16 -> (count, pklen) u8."""
        h = np.ascontiguousarray(h, np.uint16)
        count = h.shape[0]
        if h.shape != (count, 1 << logn):
            raise ValueError("h must be (count, n)")
        pk = np.zeros((count, self.lib.fa_pkey_len(logn)), np.uint8)
        _check(self.lib.fa_encode_pkeys(self.handle, logn, count, _ptr(h), _ptr(pk)), "fa_encode_pkeys")
        return pk

    def compress_sigs(self, logn, s2, sig=None):
        """encoding::compress_sig (encoding.hpp:169-230) for every row of s2 (count, n) i16. sig (count, siglen) may be
        passed in: only bytes 41.. of each row are written. Returns (sig, ok)."""
        s2 = np.ascontiguousarray(s2, np.int16)
        count = s2.shape[0]
        if s2.shape != (count, 1 << logn):
            raise ValueError("s2 must be (count, n)")
        siglen = self.lib.fa_sig_len(logn)
        sig = np.zeros((count, siglen), np.uint8) if sig is None else sig
        if sig.dtype != np.uint8 or sig.shape != (count, siglen) or not sig.flags.c_contiguous:
            raise ValueError("sig must be a contiguous (count, siglen) uint8 array")
        ok = np.zeros(count, np.uint8)
        _check(self.lib.fa_compress_sigs(self.handle, logn, count, _ptr(s2), _ptr(sig), _ptr(ok)), "fa_compress_sigs")
        return sig, ok

    # ---- device-resident entry points (raw device pointers as ints) ----
    def verify_many_dev(self, logn, count, pk_ptr, msg_ptr, msg_off_ptr, mlen, sig_ptr, accept_ptr,
                        norm_policy=NORM_EXACT64, debug_ptrs=None, stream=0):
        dbg = None
        if debug_ptrs is not None:
            dbg = FaDebug(*[debug_ptrs.get(k) for k, _ in FaDebug._fields_])
        rc = self.lib.fa_verify_many_dev(self.handle, logn, count, pk_ptr, msg_ptr, msg_off_ptr, mlen, sig_ptr,
                                         norm_policy, accept_ptr, ctypes.byref(dbg) if dbg is not None else None,
                                         stream or None)
        _check(rc, "fa_verify_many_dev")

    def verify_many_dev_noalloc(self, logn, count, pk_ptr, msg_ptr, msg_off_ptr, mlen, sig_ptr, accept_ptr,
                                norm_policy=NORM_EXACT64, stream=0):
        """fa_verify_many_dev_noalloc: the same on the library's static scratch array, no device allocation."""
        rc = self.lib.fa_verify_many_dev_noalloc(self.handle, logn, count, pk_ptr, msg_ptr, msg_off_ptr, mlen, sig_ptr,
                                                 norm_policy, accept_ptr, stream or None)
        _check(rc, "fa_verify_many_dev_noalloc")

    def verify_mixed_dev(self, count, logn_ptr, pk_ptr, pk_off_ptr, msg_ptr, msg_off_ptr, sig_ptr, sig_off_ptr, accept_ptr,
                         norm_policy=NORM_EXACT64, stream=0):
        rc = self.lib.fa_verify_mixed_dev(self.handle, count, logn_ptr, pk_ptr, pk_off_ptr, msg_ptr, msg_off_ptr, sig_ptr,
                                          sig_off_ptr, norm_policy, accept_ptr, stream or None)
        _check(rc, "fa_verify_mixed_dev")

    def finish(self, stream=0):
        """fa_finish: synchronise `stream`; returns the number of unresolved items of the asynchronous calls since
        the last finish (0 in practice)."""
        n = ctypes.c_uint64(0)
        rc = self.lib.fa_finish(self.handle, stream or None, ctypes.byref(n))
        if rc not in (0, E_UNRESOLVED):
            _check(rc, "fa_finish")
        return int(n.value)

    def prepare_keys_dev(self, logn, nkeys, pk_ptr, h_ptr, prepared_ptr, stream=0):
        _check(self.lib.fa_prepare_keys_dev(self.handle, logn, nkeys, pk_ptr, h_ptr, prepared_ptr, stream or None),
               "fa_prepare_keys_dev")

    def verify_many_prepared_dev(self, logn, count, prepared_ptr, key_idx_ptr, msg_ptr, msg_off_ptr, mlen, sig_ptr,
                                 accept_ptr, norm_policy=NORM_EXACT64, stream=0):
        rc = self.lib.fa_verify_many_prepared_dev(self.handle, logn, count, prepared_ptr, key_idx_ptr, msg_ptr, msg_off_ptr,
                                                  mlen, sig_ptr, norm_policy, accept_ptr, None, stream or None)
        _check(rc, "fa_verify_many_prepared_dev")

    def polymul_many_dev(self, logn, count, a_ptr, b_ptr, c_ptr, stream=0):
        _check(self.lib.fa_polymul_many_dev(self.handle, logn, count, a_ptr, b_ptr, c_ptr, stream or None),
               "fa_polymul_many_dev")
