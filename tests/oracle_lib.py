"""ctypes binding of oracle/libnolzss_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference algorithm (oracle/nolzss_oracle.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it; the
product package nolzss_amd never does.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_ROOT = Path(__file__).resolve().parent.parent
_SO = _ROOT / "oracle" / "libnolzss_oracle.so"

RC_MASK = 1 << 63


class OracleError(RuntimeError):
    pass


class OracleInvalidArgument(ValueError):
    pass


_FACTOR_DT = np.dtype([("start", "<u8"), ("length", "<u8"), ("ref", "<u8")])


def _load():
    src = _ROOT / "oracle" / "nolzss_oracle.c"
    if not _SO.exists() or (src.exists() and src.stat().st_mtime > _SO.stat().st_mtime):
        subprocess.check_call(["make", "-C", str(_ROOT / "oracle")], stdout=subprocess.DEVNULL)
    lib = C.CDLL(str(_SO))
    u8p, sz, szp = C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)
    vpp = C.POINTER(C.c_void_p)
    lib.oracle_last_error.restype = C.c_char_p
    lib.oracle_free.argtypes = [C.c_void_p]
    lib.oracle_suffix_array.argtypes = [u8p, sz, C.c_void_p]
    lib.oracle_lcp_array.argtypes = [u8p, sz, C.c_void_p, C.c_void_p]
    lib.oracle_factorize.argtypes = [u8p, sz, sz, vpp, szp]
    lib.oracle_count_factors.argtypes = [u8p, sz, sz, szp]
    lib.oracle_lpnf_all.argtypes = [u8p, sz, C.c_void_p, C.c_void_p]
    lib.oracle_prepare_multiple_dna_w_rc.argtypes = [
        C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), sz, vpp, szp, szp, vpp, szp]
    lib.oracle_factorize_multiple_dna_w_rc.argtypes = [u8p, sz, sz, vpp, szp]
    lib.oracle_count_factors_multiple_dna_w_rc.argtypes = [u8p, sz, sz, szp]
    lib.oracle_factorize_dna_w_rc.argtypes = [u8p, sz, vpp, szp]
    lib.oracle_lpnf_all_rc.argtypes = [u8p, sz, C.c_void_p, C.c_void_p]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _check(rc):
    if rc == 0:
        return
    msg = lib().oracle_last_error().decode("utf-8", "replace")
    if rc == 1:
        raise OracleInvalidArgument(msg)
    raise OracleError(msg)


def _buf(data):
    """bytes/bytearray/np.uint8 array -> (pointer, length, keepalive)."""
    if isinstance(data, np.ndarray):
        a = np.ascontiguousarray(data, dtype=np.uint8)
    else:
        a = np.frombuffer(bytes(data), dtype=np.uint8)
    return a.ctypes.data, a.size, a


def _take_factors(ptr, z):
    if not ptr.value or z == 0:
        if ptr.value:
            lib().oracle_free(ptr)
        return np.zeros(0, dtype=_FACTOR_DT)
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(z * 3,)).copy()
    lib().oracle_free(ptr)
    return arr.view(_FACTOR_DT)


def factors_array(data, start_pos=0):
    p, n, keep = _buf(data)
    out, z = C.c_void_p(), C.c_size_t()
    _check(lib().oracle_factorize(p, n, start_pos, C.byref(out), C.byref(z)))
    return _take_factors(out, z.value)


def factorize(data, start_pos=0):
    f = factors_array(data, start_pos)
    return list(zip(f["start"].tolist(), f["length"].tolist(), f["ref"].tolist()))


def count_factors(data, start_pos=0):
    p, n, keep = _buf(data)
    z = C.c_size_t()
    _check(lib().oracle_count_factors(p, n, start_pos, C.byref(z)))
    return z.value


def suffix_array(data):
    p, n, keep = _buf(data)
    sa = np.zeros(n, dtype=np.int32)
    _check(lib().oracle_suffix_array(p, n, sa.ctypes.data))
    return sa


def lcp_array(data, sa):
    p, n, keep = _buf(data)
    sa = np.ascontiguousarray(sa, dtype=np.int32)
    lcp = np.zeros(n, dtype=np.int32)
    _check(lib().oracle_lcp_array(p, n, sa.ctypes.data, lcp.ctypes.data))
    return lcp


def lpnf_all(data):
    p, n, keep = _buf(data)
    ln = np.zeros(n, dtype=np.uint32)
    rf = np.zeros(n, dtype=np.uint32)
    _check(lib().oracle_lpnf_all(p, n, ln.ctypes.data, rf.ctypes.data))
    return ln, rf


def prepare_multiple_dna_w_rc(seqs):
    seqs = [s.encode("ascii") if isinstance(s, str) else bytes(s) for s in seqs]
    k = len(seqs)
    arr = (C.c_char_p * max(k, 1))(*seqs)
    lens = (C.c_size_t * max(k, 1))(*[len(s) for s in seqs])
    S, S_len, orig = C.c_void_p(), C.c_size_t(), C.c_size_t()
    sp, ns = C.c_void_p(), C.c_size_t()
    _check(lib().oracle_prepare_multiple_dna_w_rc(arr, lens, k, C.byref(S), C.byref(S_len),
                                                   C.byref(orig), C.byref(sp), C.byref(ns)))
    s_bytes = C.string_at(S, S_len.value) if S.value else b""
    sent = []
    if sp.value:
        sent = np.ctypeslib.as_array(C.cast(sp, C.POINTER(C.c_uint64)), shape=(ns.value,)).tolist()
    lib().oracle_free(S)
    lib().oracle_free(sp)
    return s_bytes, orig.value, sent


def _rc_tuples(f):
    ref = f["ref"]
    is_rc = (ref >> np.uint64(63)).astype(bool)
    clean = ref & np.uint64(RC_MASK - 1)
    return list(zip(f["start"].tolist(), f["length"].tolist(), clean.tolist(), is_rc.tolist()))


def factorize_dna_w_rc(data):
    p, n, keep = _buf(data)
    out, z = C.c_void_p(), C.c_size_t()
    _check(lib().oracle_factorize_dna_w_rc(p, n, C.byref(out), C.byref(z)))
    return _rc_tuples(_take_factors(out, z.value))


def factors_array_multiple_dna_w_rc(S, start_pos=0):
    p, n, keep = _buf(S)
    out, z = C.c_void_p(), C.c_size_t()
    _check(lib().oracle_factorize_multiple_dna_w_rc(p, n, start_pos, C.byref(out), C.byref(z)))
    return _take_factors(out, z.value)


def factorize_multiple_dna_w_rc(S, start_pos=0):
    return _rc_tuples(factors_array_multiple_dna_w_rc(S, start_pos))


def count_factors_multiple_dna_w_rc(S, start_pos=0):
    p, n, keep = _buf(S)
    z = C.c_size_t()
    _check(lib().oracle_count_factors_multiple_dna_w_rc(p, n, start_pos, C.byref(z)))
    return z.value


def lpnf_all_rc(S):
    p, n, keep = _buf(S)
    N = n // 2 - 1
    ln = np.zeros(N, dtype=np.uint32)
    rf = np.zeros(N, dtype=np.uint64)
    _check(lib().oracle_lpnf_all_rc(p, n, ln.ctypes.data, rf.ctypes.data))
    return ln, rf
