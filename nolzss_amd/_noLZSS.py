"""Python-visible surface of the reference's compiled module `noLZSS._noLZSS`
(reference: src/cpp/bindings.cpp), bound to libnolzss_hip.so through ctypes.

Same names, argument meaning, return shapes and error behaviour as the pybind11 module; ctypes
releases the GIL around every native call just as the reference does with gil_scoped_release
(bindings.cpp:70).  All 42 functions of the reference module are present; every one of them
computes on the GPU through the C ABI (there is no CPU path).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import lib, check

__version__ = lib.nolzss_version().decode()

RC_MASK = 1 << 63
FACTOR_DTYPE = np.dtype([("start", "<u8"), ("length", "<u8"), ("ref", "<u8")])

_default_device = int(os.environ.get("NOLZSS_DEVICE", os.environ.get("LOCAL_RANK", "0")) or 0)


def set_device(device: int) -> None:
    """Select the HIP device used by the calls below (extension; default LOCAL_RANK or 0)."""
    global _default_device
    _default_device = int(device)


def get_device() -> int:
    return _default_device


class Factor:
    """reference: py::class_<Factor>, bindings.cpp:44-48 (ref is reported with RC_MASK stripped)"""
    __slots__ = ("start", "length", "_raw_ref")

    def __init__(self, start=0, length=0, ref=0):
        self.start, self.length, self._raw_ref = start, length, ref

    @property
    def ref(self):
        return self._raw_ref & (RC_MASK - 1)

    @property
    def is_rc(self):
        return bool(self._raw_ref & RC_MASK)


class FastaFactorizationResult:
    """reference: py::class_<FastaFactorizationResult>, bindings.cpp:51-53"""
    __slots__ = ("factors", "sentinel_factor_indices", "sequence_ids")

    def __init__(self, factors=(), sentinel_factor_indices=(), sequence_ids=()):
        self.factors = list(factors)
        self.sentinel_factor_indices = list(sentinel_factor_indices)
        self.sequence_ids = list(sequence_ids)


class FastaPerSequenceFactorizationResult:
    """reference: py::class_<FastaPerSequenceFactorizationResult>, bindings.cpp:1208-1213"""
    __slots__ = ("per_sequence_factors", "sequence_ids")

    def __init__(self, per_sequence_factors=(), sequence_ids=()):
        self.per_sequence_factors = list(per_sequence_factors)
        self.sequence_ids = list(sequence_ids)


def _as_buffer(data):
    """1-D, itemsize-1 buffer -> (address, nbytes, keepalive)  (bindings.cpp:59-67)."""
    try:
        mv = memoryview(data)
    except TypeError:
        raise TypeError(f"a bytes-like object is required, not '{type(data).__name__}'")
    if mv.itemsize != 1 or mv.ndim != 1:
        raise ValueError("data must be a 1-dimensional bytes-like object")
    if not mv.c_contiguous:
        mv = memoryview(bytes(mv))
    arr = np.frombuffer(mv, dtype=np.uint8)
    return arr.ctypes.data, arr.size, arr


class _Owned:
    """Frees a library-owned block when the last array that views it is gone."""

    def __init__(self, ptr):
        self.ptr = C.c_void_p(ptr.value)

    def __del__(self):
        lib.nolzss_free(self.ptr)


def _take(ptr, z):
    """library-owned factor array -> numpy structured array: a view of the block (a 2^30-base text has
    1.2 GB of factor records; copying them costs as much as downloading them), freed with the array."""
    if not ptr.value:
        return np.zeros(0, dtype=FACTOR_DTYPE)
    owner = _Owned(ptr)
    if z == 0:
        return np.zeros(0, dtype=FACTOR_DTYPE)
    raw = (C.c_uint64 * (3 * z)).from_address(ptr.value)
    raw._owner = owner
    return np.frombuffer(raw, dtype=FACTOR_DTYPE)


def _tuples3(f):
    return list(zip(f["start"].tolist(), f["length"].tolist(), f["ref"].tolist()))


def _tuples4(f):
    ref = f["ref"]
    is_rc = (ref >> np.uint64(63)).astype(bool)
    clean = ref & np.uint64(RC_MASK - 1)
    return list(zip(f["start"].tolist(), f["length"].tolist(), clean.tolist(), is_rc.tolist()))


# ---- plain mode --------------------------------------------------------------------------
def factorize_array(data, start_pos: int = 0) -> np.ndarray:
    """Extension: factors as a numpy structured array (start, length, ref), no tuple building."""
    p, n, keep = _as_buffer(data)
    out, z = C.c_void_p(), C.c_size_t()
    check(lib.nolzss_factorize(p, n, start_pos, _default_device, C.byref(out), C.byref(z)))
    return _take(out, z.value)


def factorize(data):
    """reference: m.def("factorize"), bindings.cpp:56-77 -> list[(start, length, ref)]"""
    return _tuples3(factorize_array(data))


def count_factors(data) -> int:
    """reference: m.def("count_factors"), bindings.cpp:122-141"""
    p, n, keep = _as_buffer(data)
    z = C.c_size_t()
    check(lib.nolzss_count_factors(p, n, 0, _default_device, C.byref(z)))
    return z.value


def factorize_file(path: str, reserve_hint: int = 0):
    """reference: m.def("factorize_file"), bindings.cpp:96-105 (reserve_hint is a host-vector
    hint in the reference and has no effect here)."""
    out, z = C.c_void_p(), C.c_size_t()
    check(lib.nolzss_factorize_file(os.fsencode(path), 0, _default_device, C.byref(out), C.byref(z)))
    return _tuples3(_take(out, z.value))


def count_factors_file(path: str) -> int:
    """reference: m.def("count_factors_file"), bindings.cpp:157-164"""
    z = C.c_size_t()
    check(lib.nolzss_count_factors_file(os.fsencode(path), 0, _default_device, C.byref(z)))
    return z.value


def factorize_device(data_ptr: int, n: int, stream: int = 0, emit: int = 2, start_pos: int = 0):
    """Extension used by bench.py / the shard dispatcher: the text is already in HBM
    (data_ptr = device address, e.g. torch.Tensor.data_ptr()).
    emit 0: count only; 1: build the factor records in HBM, no download; 2: download them.
    Returns (z, factor array or None)."""
    out, z = C.c_void_p(), C.c_size_t()
    check(lib.nolzss_factorize_device(data_ptr, n, start_pos, _default_device, stream or None, emit,
                                      C.byref(out) if emit == 2 else None, C.byref(z)))
    return z.value, (_take(out, z.value) if emit == 2 else None)


def factorize_batch(texts, devices=None, want_factors: bool = True, with_rc: bool = False):
    """Extension: the per-sequence shard unit of read_nucleotide_fasta
    (reference: genomics/fasta.py:110-122).  Returns (counts, [factor arrays] or None).
    with_rc: every record as factorize_dna_w_rc would (ref carries RC_MASK for reverse-complement
    factors; the per-record step of factorize_fasta_dna_w_rc_per_sequence)."""
    devices = list(devices) if devices is not None else [_default_device]
    bufs = [_as_buffer(t) for t in texts]
    m = len(bufs)
    ptrs = (C.c_void_p * max(m, 1))(*[b[0] for b in bufs])
    lens = (C.c_size_t * max(m, 1))(*[b[1] for b in bufs])
    devs = (C.c_int * len(devices))(*devices)
    out = C.POINTER(C.c_void_p)()
    zs = C.POINTER(C.c_size_t)()
    entry = lib.nolzss_factorize_batch_dna_w_rc if with_rc else lib.nolzss_factorize_batch
    check(entry(ptrs, lens, m, devs, len(devices), C.byref(out) if want_factors else None, C.byref(zs)))
    owner = _BatchResult(out if want_factors else None, zs, m)
    counts = np.ctypeslib.as_array(zs, shape=(m,)).tolist() if m else []
    if not want_factors:
        return counts, None
    # the arrays are views of the library's blocks (a merged run delivers the factors of thousands of
    # records in one block); the blocks live as long as any of the views
    arrays = []
    for j in range(m):
        if counts[j] == 0 or not out[j]:
            arrays.append(np.zeros(0, dtype=FACTOR_DTYPE))
        else:
            raw = (C.c_uint64 * (3 * counts[j])).from_address(out[j])
            raw._owner = owner
            arrays.append(np.frombuffer(raw, dtype=FACTOR_DTYPE))
    return counts, arrays


def factorize_batch_device(data_ptrs, lengths, emit: int = 0):
    """Extension (measurement): per-sequence batch over records resident in device memory
    (C ABI nolzss_factorize_batch_device); returns the factor count of every record."""
    m = len(data_ptrs)
    ptrs = (C.c_void_p * max(m, 1))(*data_ptrs)
    lens = (C.c_size_t * max(m, 1))(*lengths)
    zs = (C.c_size_t * max(m, 1))()
    check(lib.nolzss_factorize_batch_device(ptrs, lens, m, _default_device, emit, zs))
    return [zs[j] for j in range(m)]


class UnsupportedInput(Exception):
    """The native host-side reader does not take this input (NOLZSS_ERR_UNSUPPORTED): parse it in Python."""


class _FastaResult:
    """Frees a nolzss_nucleotide_fasta when the last array that views its blocks is gone."""

    def __init__(self, res):
        self.res = res

    def __del__(self):
        lib.nolzss_free_nucleotide_fasta(C.byref(self.res))


def read_nucleotide_fasta_arrays(path, devices=None, want_factors: bool = True, shard_index: int = 0,
                                 shard_count: int = 1):
    """Extension: the native form of genomics.read_nucleotide_fasta (reference: genomics/fasta.py:79-126;
    C ABI nolzss_read_nucleotide_fasta) -- the file is read, parsed and checked on the host by the
    library, every record of this shard factorized as one per-sequence batch.  Returns
    (ids, lengths, counts, owners, factor arrays or None); arrays of records this shard does not own are
    None.  Raises RuntimeError with the reference's FASTAError text, UnsupportedInput for non-ASCII files."""
    devices = list(devices) if devices is not None else [_default_device]
    devs = (C.c_int * len(devices))(*devices)
    res = _lib.NucleotideFasta()
    rc = lib.nolzss_read_nucleotide_fasta(os.fsencode(str(path)), devs, len(devices), 1 if want_factors else 0,
                                          shard_index, shard_count, C.byref(res))
    if rc == _lib.ERR_UNSUPPORTED:
        raise UnsupportedInput(lib.nolzss_last_error().decode("utf-8", "replace"))
    check(rc)
    owner = _FastaResult(res)
    m = res.num_sequences
    blob = C.string_at(res.sequence_ids, res.sequence_ids_bytes) if res.sequence_ids_bytes else b""
    ids = [x.decode("utf-8") for x in blob.split(b"\x00")[:m]]
    lengths = [res.lengths[j] for j in range(m)]
    counts = [res.counts[j] for j in range(m)]
    owners = [res.owners[j] for j in range(m)]
    arrays = None
    if want_factors:
        arrays = []
        for j in range(m):
            if owners[j] != shard_index:
                arrays.append(None)
            elif counts[j] == 0 or not res.factors[j]:
                arrays.append(np.zeros(0, dtype=FACTOR_DTYPE))
            else:
                raw = (C.c_uint64 * (3 * counts[j])).from_address(res.factors[j])
                raw._owner = owner
                arrays.append(np.frombuffer(raw, dtype=FACTOR_DTYPE))
    return ids, lengths, counts, owners, arrays


class _BatchResult:
    """Frees the result of nolzss_factorize_batch when the last array that views it is gone."""

    def __init__(self, out, zs, m):
        self.out, self.zs, self.m = out, zs, m

    def __del__(self):
        lib.nolzss_free_batch(self.out, self.zs, self.m)


# ---- reverse-complement DNA mode -----------------------------------------------------------
def factorize_dna_w_rc_array(data) -> np.ndarray:
    p, n, keep = _as_buffer(data)
    out, z = C.c_void_p(), C.c_size_t()
    check(lib.nolzss_factorize_dna_w_rc(p, n, _default_device, C.byref(out), C.byref(z)))
    return _take(out, z.value)


def factorize_dna_w_rc(data):
    """reference: bindings.cpp:207-228 -> list[(start, length, ref & ~RC_MASK, is_rc)]"""
    return _tuples4(factorize_dna_w_rc_array(data))


def factorize_dna_w_rc_device(data_ptr: int, n: int, stream: int = 0, emit: int = 2):
    """Extension (bench.py, BASELINE config 5): factorize_dna_w_rc of a text that is already in HBM
    (data_ptr = device address).  emit 0: count only; 1: factor records built in HBM, no download;
    2: download them.  Returns (z, factor array or None)."""
    out, z = C.c_void_p(), C.c_size_t()
    check(lib.nolzss_factorize_dna_w_rc_device(data_ptr, n, _default_device, stream or None, emit,
                                               C.byref(out) if emit == 2 else None, C.byref(z)))
    return z.value, (_take(out, z.value) if emit == 2 else None)


def count_factors_dna_w_rc(data) -> int:
    """reference: bindings.cpp:276-295"""
    p, n, keep = _as_buffer(data)
    z = C.c_size_t()
    check(lib.nolzss_count_factors_dna_w_rc(p, n, _default_device, C.byref(z)))
    return z.value


def factorize_multiple_dna_w_rc_array(data, start_pos: int = 0) -> np.ndarray:
    p, n, keep = _as_buffer(data)
    out, z = C.c_void_p(), C.c_size_t()
    check(lib.nolzss_factorize_multiple_dna_w_rc(p, n, start_pos, _default_device, C.byref(out), C.byref(z)))
    return _take(out, z.value)


def factorize_multiple_dna_w_rc(data):
    """reference: bindings.cpp:361-382"""
    return _tuples4(factorize_multiple_dna_w_rc_array(data))


def count_factors_multiple_dna_w_rc(data) -> int:
    """reference: bindings.cpp:427-446"""
    p, n, keep = _as_buffer(data)
    z = C.c_size_t()
    check(lib.nolzss_count_factors_multiple_dna_w_rc(p, n, 0, _default_device, C.byref(z)))
    return z.value


def prepare_multiple_dna_sequences_w_rc_bytes(sequences):
    """Extension: like prepare_multiple_dna_sequences_w_rc but returns the prepared string as
    bytes, so sentinel values >= 128 (more than ~61 sequences) survive."""
    seqs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in sequences]
    k = len(seqs)
    arr = (C.c_char_p * max(k, 1))(*seqs)
    lens = (C.c_size_t * max(k, 1))(*[len(s) for s in seqs])
    S, S_len, orig = C.c_void_p(), C.c_size_t(), C.c_size_t()
    sp, ns = C.c_void_p(), C.c_size_t()
    check(lib.nolzss_prepare_multiple_dna_w_rc(arr, lens, k, C.byref(S), C.byref(S_len), C.byref(orig),
                                               C.byref(sp), C.byref(ns)))
    try:
        data = C.string_at(S, S_len.value) if S.value else b""
        sent = []
        if sp.value and ns.value:
            sent = np.ctypeslib.as_array(C.cast(sp, C.POINTER(C.c_uint64)), shape=(ns.value,)).tolist()
    finally:
        lib.nolzss_free(S)
        lib.nolzss_free(sp)
    return data, orig.value, sent


def prepare_multiple_dna_sequences_w_rc(sequences):
    """reference: bindings.cpp:732-740 -> (prepared_string: str, original_length, sentinel_positions).
    The reference returns a std::string through pybind11, i.e. a UTF-8-decoded str; sentinel
    bytes >= 128 therefore raise UnicodeDecodeError there, and do so here as well."""
    data, orig, sent = prepare_multiple_dna_sequences_w_rc_bytes(sequences)
    return data.decode("utf-8"), orig, sent


# ---- reference + target factorization and v2 binary files -----------
def _str_arg(x, name):
    """pybind11 std::string argument: str (UTF-8 encoded) or bytes."""
    if isinstance(x, str):
        return x.encode("utf-8")
    if isinstance(x, (bytes, bytearray)):
        return bytes(x)
    raise TypeError(f"{name} must be str or bytes, not {type(x).__name__}")


def factorize_w_reference(reference_seq, target_seq):
    """reference: bindings.cpp:868-880 -> list[(start, length, ref)], absolute positions in
    reference + '\\x01' + target."""
    r, t = _str_arg(reference_seq, "reference_seq"), _str_arg(target_seq, "target_seq")
    out, z = C.c_void_p(), C.c_size_t()
    check(lib.nolzss_factorize_w_reference(r, len(r), t, len(t), _default_device, C.byref(out), C.byref(z)))
    return _tuples3(_take(out, z.value))


def factorize_w_reference_file(reference_seq, target_seq, out_path) -> int:
    """reference: bindings.cpp:907-915"""
    r, t = _str_arg(reference_seq, "reference_seq"), _str_arg(target_seq, "target_seq")
    z = C.c_size_t()
    check(lib.nolzss_factorize_w_reference_file(r, len(r), t, len(t), os.fsencode(out_path), _default_device,
                                                C.byref(z)))
    return z.value


def factorize_dna_w_reference_seq(reference_seq, target_seq):
    """reference: bindings.cpp:800-808 -> list[(start, length, ref & ~RC_MASK, is_rc)]"""
    r, t = _str_arg(reference_seq, "reference_seq"), _str_arg(target_seq, "target_seq")
    out, z = C.c_void_p(), C.c_size_t()
    check(lib.nolzss_factorize_dna_w_reference_seq(r, len(r), t, len(t), _default_device, C.byref(out), C.byref(z)))
    return _tuples4(_take(out, z.value))


def factorize_dna_w_reference_seq_file(reference_seq, target_seq, out_path) -> int:
    """reference: bindings.cpp:837-845"""
    r, t = _str_arg(reference_seq, "reference_seq"), _str_arg(target_seq, "target_seq")
    z = C.c_size_t()
    check(lib.nolzss_factorize_dna_w_reference_seq_file(r, len(r), t, len(t), os.fsencode(out_path),
                                                        _default_device, C.byref(z)))
    return z.value


def write_factors_binary_file(in_path, out_path) -> int:
    """reference: bindings.cpp:180-187 (input is a FILE path; v2 footer at the end)"""
    z = C.c_size_t()
    check(lib.nolzss_write_factors_binary_file(_str_arg(in_path, "in_path"), _str_arg(out_path, "out_path"),
                                               _default_device, C.byref(z)))
    return z.value


def write_factors_binary_file_dna_w_rc(in_path, out_path) -> int:
    """reference: bindings.cpp (write_factors_binary_file_dna_w_rc), factorizer.cpp:597-635"""
    z = C.c_size_t()
    check(lib.nolzss_write_factors_binary_file_dna_w_rc(_str_arg(in_path, "in_path"), _str_arg(out_path, "out_path"),
                                                        _default_device, C.byref(z)))
    return z.value


# ---- concatenated multi-sequence FASTA ---------------------------------------
def _sanitize_mode(mode: str) -> int:
    """reference: parse_fasta_sanitization_mode, bindings.cpp:29-37"""
    if mode == "remove_ambiguous":
        return 0
    if mode == "strict":
        return 1
    raise ValueError(f"Invalid sanitize_mode: '{mode}'. Expected 'remove_ambiguous' or 'strict'.")


def prepare_multiple_dna_sequences_no_rc_bytes(sequences):
    seqs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in sequences]
    k = len(seqs)
    arr = (C.c_char_p * max(k, 1))(*seqs)
    lens = (C.c_size_t * max(k, 1))(*[len(s) for s in seqs])
    S, S_len, orig = C.c_void_p(), C.c_size_t(), C.c_size_t()
    sp, ns = C.c_void_p(), C.c_size_t()
    check(lib.nolzss_prepare_multiple_dna_no_rc(arr, lens, k, C.byref(S), C.byref(S_len), C.byref(orig),
                                                C.byref(sp), C.byref(ns)))
    try:
        data = C.string_at(S, S_len.value) if S.value else b""
        sent = []
        if sp.value and ns.value:
            sent = np.ctypeslib.as_array(C.cast(sp, C.POINTER(C.c_uint64)), shape=(ns.value,)).tolist()
    finally:
        lib.nolzss_free(S)
        lib.nolzss_free(sp)
    return data, orig.value, sent


def prepare_multiple_dna_sequences_no_rc(sequences):
    """reference: bindings.cpp (prepare_multiple_dna_sequences_no_rc) -> (str, original_length,
    sentinel_positions); sentinel bytes >= 128 raise UnicodeDecodeError as they do through pybind11."""
    data, orig, sent = prepare_multiple_dna_sequences_no_rc_bytes(sequences)
    return data.decode("utf-8"), orig, sent


def _unpack_fasta_result(res):
    try:
        z = res.num_factors
        if z:
            raw = np.ctypeslib.as_array(C.cast(res.factors, C.POINTER(C.c_uint64)), shape=(z * 3,)).copy()
            factors = _tuples4(raw.view(FACTOR_DTYPE))
        else:
            factors = []
        sent = []
        if res.num_sentinels:
            sent = np.ctypeslib.as_array(C.cast(res.sentinel_factor_indices, C.POINTER(C.c_uint64)),
                                         shape=(res.num_sentinels,)).tolist()
        blob = C.string_at(res.sequence_ids, res.sequence_ids_bytes) if res.sequence_ids_bytes else b""
        ids = [x.decode("utf-8") for x in blob.split(b"\x00")[:res.num_sequences]]
    finally:
        lib.nolzss_free_fasta_result(C.byref(res))
    return factors, sent, ids


def _fasta_multiple(fasta_path, sanitize_mode, with_rc):
    res = _lib.FastaResult()
    check(lib.nolzss_factorize_fasta_multiple_dna(_str_arg(fasta_path, "fasta_path"), 1 if with_rc else 0,
                                                  _sanitize_mode(sanitize_mode), _default_device, C.byref(res)))
    return _unpack_fasta_result(res)


def factorize_dna_rc_w_ref_fasta_files(reference_fasta_path, target_fasta_path,
                                       sanitize_mode: str = "remove_ambiguous"):
    """reference: bindings.cpp:563-585 -> (factors, sentinel factor indices, sequence ids); the
    factors cover the target records only and may point into the reference records."""
    res = _lib.FastaResult()
    check(lib.nolzss_factorize_dna_rc_w_ref_fasta_files(
        _str_arg(reference_fasta_path, "reference_fasta_path"), _str_arg(target_fasta_path, "target_fasta_path"),
        _sanitize_mode(sanitize_mode), _default_device, C.byref(res)))
    return _unpack_fasta_result(res)


def write_factors_dna_w_reference_fasta_files_to_binary(reference_fasta_path, target_fasta_path, out_path,
                                                        sanitize_mode: str = "remove_ambiguous") -> int:
    """reference: fasta_processor.cpp:381-390"""
    z = C.c_size_t()
    check(lib.nolzss_write_factors_dna_w_reference_fasta_files_to_binary(
        _str_arg(reference_fasta_path, "reference_fasta_path"), _str_arg(target_fasta_path, "target_fasta_path"),
        _str_arg(out_path, "out_path"), _sanitize_mode(sanitize_mode), _default_device, C.byref(z)))
    return z.value


def parallel_write_factors_dna_w_reference_fasta_files_to_binary(reference_fasta_path, target_fasta_path, out_path,
                                                                 num_threads: int = 0,
                                                                 sanitize_mode: str = "remove_ambiguous") -> int:
    return write_factors_dna_w_reference_fasta_files_to_binary(reference_fasta_path, target_fasta_path, out_path,
                                                               sanitize_mode)


def _read_input_file(path):
    path = os.fsdecode(_str_arg(path, "path"))
    try:
        with open(path, "rb") as fh:
            return fh.read()
    except OSError:
        raise RuntimeError(f"Cannot open input file: {path}")   # factorizer.cpp:498-500


def factorize_file_dna_w_rc(path, reserve_hint: int = 0):
    """reference: noLZSS::factorize_file_dna_w_rc, factorizer.cpp:525-545"""
    return factorize_dna_w_rc(_read_input_file(path))


def count_factors_file_dna_w_rc(path) -> int:
    """reference: factorizer.cpp:575-577"""
    return count_factors_dna_w_rc(_read_input_file(path))


def factorize_file_multiple_dna_w_rc(path, reserve_hint: int = 0):
    """reference: factorizer.cpp:658-686 (the file holds an already prepared string)"""
    return factorize_multiple_dna_w_rc(_read_input_file(path))


def count_factors_file_multiple_dna_w_rc(path) -> int:
    """reference: factorizer.cpp:707-732"""
    return count_factors_multiple_dna_w_rc(_read_input_file(path))


def write_factors_binary_file_multiple_dna_w_rc(in_path, out_path) -> int:
    """reference: factorizer.cpp:751-790: no names, no sentinels, total_length = file size"""
    data = _read_input_file(in_path)
    f = factorize_multiple_dna_w_rc_array(data)
    f = np.ascontiguousarray(f)
    check(lib.nolzss_write_factor_file(_str_arg(out_path, "out_path"), f.ctypes.data if len(f) else None, len(f), 0, 0,
                                       len(data), None, 0))
    return len(f)


def factorize_fasta_multiple_dna_w_rc(fasta_path, sanitize_mode: str = "remove_ambiguous"):
    """reference: bindings.cpp:511-536 -> (factors as (start, length, ref, is_rc), sentinel factor
    indices, sequence ids)."""
    return _fasta_multiple(fasta_path, sanitize_mode, True)


def factorize_fasta_multiple_dna_no_rc(fasta_path, sanitize_mode: str = "remove_ambiguous"):
    """reference: bindings.cpp:611-636"""
    return _fasta_multiple(fasta_path, sanitize_mode, False)


def _write_fasta_multiple(fasta_path, out_path, sanitize_mode, with_rc):
    z = C.c_size_t()
    check(lib.nolzss_write_factors_binary_file_fasta_multiple_dna(
        _str_arg(fasta_path, "fasta_path"), _str_arg(out_path, "out_path"), 1 if with_rc else 0,
        _sanitize_mode(sanitize_mode), _default_device, C.byref(z)))
    return z.value


def write_factors_binary_file_fasta_multiple_dna_w_rc(fasta_path, out_path, sanitize_mode: str = "remove_ambiguous"):
    """reference: fasta_processor.cpp:345-351"""
    return _write_fasta_multiple(fasta_path, out_path, sanitize_mode, True)


def write_factors_binary_file_fasta_multiple_dna_no_rc(fasta_path, out_path, sanitize_mode: str = "remove_ambiguous"):
    """reference: fasta_processor.cpp:353-359"""
    return _write_fasta_multiple(fasta_path, out_path, sanitize_mode, False)


def _fasta_per_sequence(fasta_path, sanitize_mode, with_rc, want_factors, out_dir=None):
    res = _lib.FastaPerSequenceResult()
    check(lib.nolzss_factorize_fasta_per_sequence(
        _str_arg(fasta_path, "fasta_path"), 1 if with_rc else 0, _sanitize_mode(sanitize_mode),
        1 if want_factors else 0, _str_arg(out_dir, "out_dir") if out_dir is not None else None,
        _default_device, C.byref(res)))
    try:
        m = res.num_sequences
        counts = [res.counts[j] for j in range(m)]
        per_seq = None
        if want_factors:
            per_seq = []
            for j in range(m):
                if counts[j] == 0 or not res.factors[j]:
                    per_seq.append([])
                else:
                    raw = np.ctypeslib.as_array(C.cast(res.factors[j], C.POINTER(C.c_uint64)),
                                                shape=(counts[j] * 3,)).copy()
                    per_seq.append(_tuples4(raw.view(FACTOR_DTYPE)))
        blob = C.string_at(res.sequence_ids, res.sequence_ids_bytes) if res.sequence_ids_bytes else b""
        ids = [x.decode("utf-8") for x in blob.split(b"\x00")[:m]]
    finally:
        lib.nolzss_free_fasta_per_sequence_result(C.byref(res))
    return per_seq, counts, ids


def factorize_fasta_dna_w_rc_per_sequence(fasta_path, sanitize_mode: str = "remove_ambiguous"):
    """reference: bindings.cpp:1215-1237 -> (per-sequence factor lists, sequence ids)"""
    per_seq, _, ids = _fasta_per_sequence(fasta_path, sanitize_mode, True, True)
    return per_seq, ids


def factorize_fasta_dna_no_rc_per_sequence(fasta_path, sanitize_mode: str = "remove_ambiguous"):
    """reference: bindings.cpp (no-rc twin); the last base of every record is dropped as in
    fasta_processor.cpp:469-471"""
    per_seq, _, ids = _fasta_per_sequence(fasta_path, sanitize_mode, False, True)
    return per_seq, ids


def count_factors_fasta_dna_w_rc_per_sequence(fasta_path, sanitize_mode: str = "remove_ambiguous"):
    """reference: bindings.cpp:1372-1389 -> (counts, sequence ids, total)"""
    _, counts, ids = _fasta_per_sequence(fasta_path, sanitize_mode, True, False)
    return counts, ids, sum(counts)


def count_factors_fasta_dna_no_rc_per_sequence(fasta_path, sanitize_mode: str = "remove_ambiguous"):
    _, counts, ids = _fasta_per_sequence(fasta_path, sanitize_mode, False, False)
    return counts, ids, sum(counts)


def write_factors_binary_file_fasta_dna_w_rc_per_sequence(fasta_path, out_dir, sanitize_mode: str = "remove_ambiguous"):
    """reference: bindings.cpp:1312-1319 -> total number of factors; one <id>.bin per record"""
    _, counts, _ = _fasta_per_sequence(fasta_path, sanitize_mode, True, False, out_dir)
    return sum(counts)


def write_factors_binary_file_fasta_dna_no_rc_per_sequence(fasta_path, out_dir, sanitize_mode: str = "remove_ambiguous"):
    _, counts, _ = _fasta_per_sequence(fasta_path, sanitize_mode, False, False, out_dir)
    return sum(counts)


def parallel_write_factors_binary_file_fasta_dna_w_rc_per_sequence(fasta_path, out_dir, num_threads: int = 0,
                                                                   sanitize_mode: str = "remove_ambiguous"):
    return write_factors_binary_file_fasta_dna_w_rc_per_sequence(fasta_path, out_dir, sanitize_mode)


def parallel_write_factors_binary_file_fasta_dna_no_rc_per_sequence(fasta_path, out_dir, num_threads: int = 0,
                                                                    sanitize_mode: str = "remove_ambiguous"):
    return write_factors_binary_file_fasta_dna_no_rc_per_sequence(fasta_path, out_dir, sanitize_mode)


# ---- thread-parallel API: same results, num_threads is irrelevant on the GPU ---
def _write_arrays(out_path, f, total_length):
    f = np.ascontiguousarray(f)
    check(lib.nolzss_write_factor_file(os.fsencode(out_path), f.ctypes.data if len(f) else None, len(f), 0, 0,
                                       int(total_length), None, 0))
    return len(f)


def parallel_factorize_to_file(text, output_path, num_threads: int = 0, start_pos: int = 0) -> int:
    """reference: bindings.cpp:978-979 over parallel_factorizer.cpp:55-144 (footer :761-765:
    total_length = sum of factor lengths)."""
    data = _str_arg(text, "text")
    if len(data) == 0:
        return 0                                            # parallel_factorizer.cpp:57
    if start_pos >= len(data):
        raise ValueError("start_pos must be less than text length")   # :59-61
    f = factorize_array(data, start_pos=start_pos)
    return _write_arrays(output_path, f, int(f["length"].sum()))


def parallel_factorize_file_to_file(input_path, output_path, num_threads: int = 0, start_pos: int = 0) -> int:
    with open(os.fsdecode(_str_arg(input_path, "input_path")), "rb") as fh:
        return parallel_factorize_to_file(fh.read(), output_path, num_threads, start_pos)


def parallel_factorize_dna_w_rc_to_file(text, output_path, num_threads: int = 0) -> int:
    """reference: parallel_factorizer.cpp:1001-1017"""
    data = _str_arg(text, "text")
    if len(data) == 0:
        return 0
    f = factorize_dna_w_rc_array(data)
    return _write_arrays(output_path, f, int(f["length"].sum()))


def parallel_factorize_file_dna_w_rc_to_file(input_path, output_path, num_threads: int = 0) -> int:
    """reference: parallel_factorizer.cpp:1031-1041"""
    path = os.fsdecode(_str_arg(input_path, "input_path"))
    try:
        with open(path, "rb") as fh:
            data = fh.read()
    except OSError:
        raise RuntimeError(f"Cannot open input file: {path}")
    return parallel_factorize_dna_w_rc_to_file(data, output_path, num_threads)


def parallel_write_factors_binary_file_fasta_multiple_dna_w_rc(fasta_path, out_path, num_threads: int = 0,
                                                               sanitize_mode: str = "remove_ambiguous"):
    return _write_fasta_multiple(fasta_path, out_path, sanitize_mode, True)


def parallel_write_factors_binary_file_fasta_multiple_dna_no_rc(fasta_path, out_path, num_threads: int = 0,
                                                                sanitize_mode: str = "remove_ambiguous"):
    return _write_fasta_multiple(fasta_path, out_path, sanitize_mode, False)


# ---- measurement hooks ----------------------------------------------------------------------
def profile_enable(on: bool = True) -> None:
    check(lib.nolzss_profile_enable(_default_device, 1 if on else 0))


def profile_reset() -> None:
    check(lib.nolzss_profile_reset(_default_device))


def profile_report() -> dict:
    """{stage name: (launch count, total milliseconds, algorithmic bytes)} measured with HIP
    events on the pipeline stream."""
    buf = C.create_string_buffer(1 << 16)
    check(lib.nolzss_profile_report(_default_device, buf, len(buf)))
    res = {}
    for line in buf.value.decode().splitlines():
        name, count, ms, nbytes = line.split()
        res[name] = (int(count), float(ms), float(nbytes))
    return res


def device_count() -> int:
    c = C.c_int()
    check(lib.nolzss_device_count(C.byref(c)))
    return c.value


# ---- intermediate arrays for the parity tests ------------------------------------------------
def debug_arrays(data):
    """-> dict(sa, isa, lcp (n+1 entries), lstar) as computed on the device."""
    p, n, keep = _as_buffer(data)
    sa = np.zeros(n, dtype=np.uint32)
    isa = np.zeros(n, dtype=np.uint32)
    lcp = np.zeros(n + 1, dtype=np.uint32)
    lstar = np.zeros(n, dtype=np.uint32)
    check(lib.nolzss_debug_arrays(p, n, _default_device, sa.ctypes.data, isa.ctypes.data, lcp.ctypes.data,
                                  lstar.ctypes.data))
    return {"sa": sa, "isa": isa, "lcp": lcp, "lstar": lstar}


def debug_sort_pairs(keys, vals):
    keys = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    vals = np.ascontiguousarray(vals, dtype=np.uint32).copy()
    check(lib.nolzss_debug_sort_pairs(keys.ctypes.data, vals.ctypes.data, keys.size, _default_device))
    return keys, vals


def debug_arena():
    """(capacity, high-water mark) in bytes of the device arena."""
    cap, peak = C.c_size_t(), C.c_size_t()
    check(lib.nolzss_debug_arena(_default_device, C.byref(cap), C.byref(peak)))
    return cap.value, peak.value


def debug_parse_fasta(path, sanitize_mode: str = "remove_ambiguous"):
    """[(id, sequence bytes)] as the native FASTA reader sees the file (host only)."""
    ids, seqs = C.c_void_p(), C.c_void_p()
    nb_ids, nb_seqs, count = C.c_size_t(), C.c_size_t(), C.c_size_t()
    check(lib.nolzss_debug_parse_fasta(os.fsencode(str(path)), _sanitize_mode(sanitize_mode), C.byref(ids),
                                       C.byref(nb_ids), C.byref(seqs), C.byref(nb_seqs), C.byref(count)))
    try:
        a = C.string_at(ids, nb_ids.value).split(b"\0")[:-1] if nb_ids.value else []
        b = C.string_at(seqs, nb_seqs.value).split(b"\0")[:-1] if nb_seqs.value else []
    finally:
        lib.nolzss_free(ids)
        lib.nolzss_free(seqs)
    assert len(a) == len(b) == count.value
    return list(zip(a, b))


def debug_parse_nucleotide_fasta(path):
    """[(id, sequence bytes)] as the reader behind read_nucleotide_fasta_arrays sees the file (host only);
    raises what that entry point would."""
    ids, seqs = C.c_void_p(), C.c_void_p()
    nb_ids, nb_seqs, count = C.c_size_t(), C.c_size_t(), C.c_size_t()
    rc = lib.nolzss_debug_parse_nucleotide_fasta(os.fsencode(str(path)), C.byref(ids), C.byref(nb_ids), C.byref(seqs),
                                                 C.byref(nb_seqs), C.byref(count))
    if rc == _lib.ERR_UNSUPPORTED:
        raise UnsupportedInput(lib.nolzss_last_error().decode("utf-8", "replace"))
    check(rc)
    try:
        a = C.string_at(ids, nb_ids.value).split(b"\0")[:-1] if nb_ids.value else []
        b = C.string_at(seqs, nb_seqs.value).split(b"\0")[:-1] if nb_seqs.value else []
    finally:
        lib.nolzss_free(ids)
        lib.nolzss_free(seqs)
    assert len(a) == len(b) == count.value
    return list(zip(a, b))


def debug_lpt_plan(lengths, bins: int):
    """owners of the records under the library's shard plan (host only)."""
    m = len(lengths)
    lens = (C.c_size_t * max(m, 1))(*lengths)
    owners = (C.c_size_t * max(m, 1))()
    check(lib.nolzss_debug_lpt_plan(lens, m, bins, owners))
    return [owners[j] for j in range(m)]


def debug_batch_plan(lengths, n_dev: int, with_rc: bool = False):
    """The static plan of factorize_batch for these record lengths on n_dev devices (host only, no device touched):
    (chunk_of, device_of, n_chunks) -- chunk_of[j] = merged run of record j or -1, device_of[j] = slot in the device
    list of the run record j takes on its own or -1."""
    m = len(lengths)
    lens = (C.c_size_t * max(m, 1))(*lengths)
    chunk_of = (C.c_int32 * max(m, 1))()
    device_of = (C.c_int32 * max(m, 1))()
    n_chunks = C.c_size_t()
    check(lib.nolzss_debug_batch_plan(lens, m, n_dev, 1 if with_rc else 0, chunk_of, device_of, C.byref(n_chunks)))
    return [chunk_of[j] for j in range(m)], [device_of[j] for j in range(m)], n_chunks.value


def debug_trim_arenas() -> int:
    """Releases the idle device arenas of the default device; returns the bytes given back."""
    r = C.c_size_t()
    check(lib.nolzss_debug_trim_arenas(_default_device, C.byref(r)))
    return r.value


def debug_batch_counters():
    """(records factorized by merged runs, records factorized one pipeline run each) since load."""
    a, b = C.c_uint64(), C.c_uint64()
    lib.nolzss_debug_batch_counters(C.byref(a), C.byref(b))
    return a.value, b.value


def debug_scan(data, mode: int):
    data = np.ascontiguousarray(data, dtype=np.uint32).copy()
    check(lib.nolzss_debug_scan(data.ctypes.data, data.size, mode, _default_device))
    return data
