"""falcon_amd -- MI355X-native batched Falcon-512/1024 signature verification.

Host-side mirror of the reference's verify API (falcon.hpp:230-249): `verify(N, pkey, msg, sig)`
keeps the reference's argument meaning, `verify_many` is the batch entry point. Everything runs
in hand-written HIP kernels behind the C ABI of include/falcon_amd.h; there is no CPU path.
"""
import numpy as np

from .capi import (NORM_EXACT64, NORM_WRAP32, OPT_CHUNK_ITEMS, OPT_HASH_BLOCKS_512, OPT_HASH_BLOCKS_1024,
                   OPT_PROFILE, ST_BAD_ENCODING, ST_BAD_PK_HEADER, ST_BAD_SIG_HEADER, ST_NORM_TOO_LARGE, ST_OK,
                   Context, FalconAmdError, load_library)

_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def _logn(N):
    if N not in (512, 1024):
        raise ValueError("N must be 512 or 1024")  # falcon.hpp:236 `requires`
    return 9 if N == 512 else 10


def verify(N, pkey, msg, sig, device=0, norm_policy=NORM_EXACT64):
    """falcon::verify<N>(pkey, msg, mlen, sig) (falcon.hpp:230-249): True iff the signature verifies."""
    pk = np.frombuffer(bytes(pkey), np.uint8)[None, :]
    sg = np.frombuffer(bytes(sig), np.uint8)[None, :]
    m = np.frombuffer(bytes(msg), np.uint8)[None, :]
    return bool(default_context(device).verify_many(_logn(N), pk, m, sg, norm_policy=norm_policy)[0])


def verify_many(N, pkeys, msgs, sigs, msg_off=None, device=0, norm_policy=NORM_EXACT64, debug=False):
    """Batch form: pkeys (count, pklen), sigs (count, siglen), msgs (count, mlen) or flat + msg_off."""
    return default_context(device).verify_many(_logn(N), pkeys, msgs, sigs, msg_off=msg_off,
                                               norm_policy=norm_policy, debug=debug)
