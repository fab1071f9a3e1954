"""Loader for libnolzss_hip.so, the gfx950 library behind this package.

There is deliberately no fallback: if the shared library is missing or cannot be loaded the
import fails loudly, and if no MI355X is visible every compute call raises (the library
returns NOLZSS_ERR_DEVICE).  Nothing in this package computes factors on the CPU.
"""
import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
# NOLZSS_LIB: another build of the same library (A/B measurements of compile-time variants, tools/ab_variant.sh)
LIB_PATH = Path(os.environ["NOLZSS_LIB"]).resolve() if os.environ.get("NOLZSS_LIB") else _PKG / "libnolzss_hip.so"

OK, ERR_INVALID_ARGUMENT, ERR_RUNTIME, ERR_NOMEM, ERR_DEVICE, ERR_IO, ERR_UNSUPPORTED = range(7)


class FastaResult(C.Structure):
    """Mirror of nolzss_fasta_result (include/nolzss_hip.h)."""
    _fields_ = [("factors", C.c_void_p), ("num_factors", C.c_size_t),
                ("sentinel_factor_indices", C.c_void_p), ("num_sentinels", C.c_size_t),
                ("sequence_ids", C.c_void_p), ("sequence_ids_bytes", C.c_size_t),
                ("num_sequences", C.c_size_t)]


class FastaPerSequenceResult(C.Structure):
    """Mirror of nolzss_fasta_per_sequence_result (include/nolzss_hip.h)."""
    _fields_ = [("factors", C.POINTER(C.c_void_p)), ("counts", C.POINTER(C.c_size_t)),
                ("sequence_ids", C.c_void_p), ("sequence_ids_bytes", C.c_size_t), ("num_sequences", C.c_size_t)]


class NucleotideFasta(C.Structure):
    """Mirror of nolzss_nucleotide_fasta (include/nolzss_hip.h)."""
    _fields_ = [("sequence_ids", C.c_void_p), ("sequence_ids_bytes", C.c_size_t), ("num_sequences", C.c_size_t),
                ("lengths", C.POINTER(C.c_size_t)), ("counts", C.POINTER(C.c_size_t)),
                ("owners", C.POINTER(C.c_size_t)), ("factors", C.POINTER(C.c_void_p)), ("keep", C.c_void_p)]


class Factor(C.Structure):
    """Mirror of nolzss_factor / the reference's struct Factor (factorizer.hpp:147-151)."""
    _fields_ = [("start", C.c_uint64), ("length", C.c_uint64), ("ref", C.c_uint64)]


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64; a process that initialises the system's HIP runtime
    first (through this library) and torch's afterwards ends up with two runtimes, and torch then reports no
    GPU.  When torch is installed, its runtime is loaded first so that libnolzss_hip.so binds to the same one
    (the loader resolves the SONAME to the copy already in the process) -- whatever the import order.
    NOLZSS_SYSTEM_HIP=1 keeps the system runtime."""
    import importlib.util
    import os
    if os.environ.get("NOLZSS_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = Path(spec.origin).resolve().parent / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _load():
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C nolzss_amd/csrc). "
            "nolzss_amd has no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(str(LIB_PATH))
    vp, sz = C.c_void_p, C.c_size_t
    szp, vpp = C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)
    lib.nolzss_last_error.restype = C.c_char_p
    lib.nolzss_version.restype = C.c_char_p
    lib.nolzss_free.argtypes = [vp]
    lib.nolzss_free.restype = None
    lib.nolzss_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.nolzss_factorize.argtypes = [vp, sz, sz, C.c_int, vpp, szp]
    lib.nolzss_count_factors.argtypes = [vp, sz, sz, C.c_int, szp]
    lib.nolzss_factorize_file.argtypes = [C.c_char_p, sz, C.c_int, vpp, szp]
    lib.nolzss_count_factors_file.argtypes = [C.c_char_p, sz, C.c_int, szp]
    lib.nolzss_factorize_device.argtypes = [vp, sz, sz, C.c_int, vp, C.c_int, vpp, szp]
    lib.nolzss_prepare_multiple_dna_w_rc.argtypes = [
        C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), sz, vpp, szp, szp, vpp, szp]
    lib.nolzss_factorize_multiple_dna_w_rc.argtypes = [vp, sz, sz, C.c_int, vpp, szp]
    lib.nolzss_count_factors_multiple_dna_w_rc.argtypes = [vp, sz, sz, C.c_int, szp]
    lib.nolzss_factorize_dna_w_rc.argtypes = [vp, sz, C.c_int, vpp, szp]
    lib.nolzss_count_factors_dna_w_rc.argtypes = [vp, sz, C.c_int, szp]
    lib.nolzss_factorize_dna_w_rc_device.argtypes = [vp, sz, C.c_int, vp, C.c_int, vpp, szp]
    lib.nolzss_factorize_w_reference.argtypes = [vp, sz, vp, sz, C.c_int, vpp, szp]
    lib.nolzss_factorize_dna_w_reference_seq.argtypes = [C.c_char_p, sz, C.c_char_p, sz, C.c_int, vpp, szp]
    lib.nolzss_write_factors_binary_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, szp]
    lib.nolzss_write_factors_binary_file_dna_w_rc.argtypes = [C.c_char_p, C.c_char_p, C.c_int, szp]
    lib.nolzss_factorize_w_reference_file.argtypes = [vp, sz, vp, sz, C.c_char_p, C.c_int, szp]
    lib.nolzss_factorize_dna_w_reference_seq_file.argtypes = [C.c_char_p, sz, C.c_char_p, sz, C.c_char_p,
                                                              C.c_int, szp]
    lib.nolzss_write_factor_file.argtypes = [C.c_char_p, vp, sz, C.c_uint64, C.c_uint64, C.c_uint64, vp, sz]
    lib.nolzss_prepare_multiple_dna_no_rc.argtypes = [
        C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), sz, vpp, szp, szp, vpp, szp]
    lib.nolzss_factorize_fasta_multiple_dna.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int,
                                                        C.POINTER(FastaResult)]
    lib.nolzss_free_fasta_result.argtypes = [C.POINTER(FastaResult)]
    lib.nolzss_free_fasta_result.restype = None
    lib.nolzss_write_factors_binary_file_fasta_multiple_dna.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                                                        C.c_int, szp]
    lib.nolzss_factorize_dna_rc_w_ref_fasta_files.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                                              C.POINTER(FastaResult)]
    lib.nolzss_write_factors_dna_w_reference_fasta_files_to_binary.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p,
                                                                               C.c_int, C.c_int, szp]
    lib.nolzss_factorize_fasta_per_sequence.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                                        C.POINTER(FastaPerSequenceResult)]
    lib.nolzss_free_fasta_per_sequence_result.argtypes = [C.POINTER(FastaPerSequenceResult)]
    lib.nolzss_free_fasta_per_sequence_result.restype = None
    lib.nolzss_factorize_batch.argtypes = [
        C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), sz, C.POINTER(C.c_int), sz,
        C.POINTER(C.POINTER(C.c_void_p)), C.POINTER(C.POINTER(C.c_size_t))]
    lib.nolzss_factorize_batch_dna_w_rc.argtypes = lib.nolzss_factorize_batch.argtypes
    lib.nolzss_factorize_batch_device.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), sz, C.c_int, C.c_int,
                                                  C.POINTER(C.c_size_t)]
    lib.nolzss_free_batch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), sz]
    lib.nolzss_free_batch.restype = None
    lib.nolzss_read_nucleotide_fasta.argtypes = [C.c_char_p, C.POINTER(C.c_int), sz, C.c_int, sz, sz,
                                                 C.POINTER(NucleotideFasta)]
    lib.nolzss_free_nucleotide_fasta.argtypes = [C.POINTER(NucleotideFasta)]
    lib.nolzss_free_nucleotide_fasta.restype = None
    lib.nolzss_debug_parse_nucleotide_fasta.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), szp, C.POINTER(C.c_void_p),
                                                        szp, szp]
    lib.nolzss_debug_lpt_plan.argtypes = [szp, sz, sz, szp]
    lib.nolzss_debug_batch_plan.argtypes = [szp, sz, sz, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), szp]
    lib.nolzss_profile_enable.argtypes = [C.c_int, C.c_int]
    lib.nolzss_profile_reset.argtypes = [C.c_int]
    lib.nolzss_profile_report.argtypes = [C.c_int, C.c_char_p, sz]
    lib.nolzss_debug_arrays.argtypes = [vp, sz, C.c_int, vp, vp, vp, vp]
    lib.nolzss_debug_sort_pairs.argtypes = [vp, vp, sz, C.c_int]
    lib.nolzss_debug_scan.argtypes = [vp, sz, C.c_int, C.c_int]
    lib.nolzss_debug_arena.argtypes = [C.c_int, szp, szp]
    lib.nolzss_debug_trim_arenas.argtypes = [C.c_int, szp]
    lib.nolzss_debug_parse_fasta.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), szp, C.POINTER(C.c_void_p),
                                             szp, szp]
    lib.nolzss_debug_batch_counters.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.nolzss_debug_batch_counters.restype = None
    return lib


lib = _load()

EXPORTED_SYMBOLS = [
    "nolzss_last_error", "nolzss_version", "nolzss_free", "nolzss_device_count",
    "nolzss_factorize", "nolzss_count_factors", "nolzss_factorize_file", "nolzss_count_factors_file",
    "nolzss_factorize_device", "nolzss_prepare_multiple_dna_w_rc",
    "nolzss_factorize_multiple_dna_w_rc", "nolzss_count_factors_multiple_dna_w_rc",
    "nolzss_factorize_dna_w_rc", "nolzss_count_factors_dna_w_rc", "nolzss_factorize_batch",
    "nolzss_factorize_w_reference", "nolzss_factorize_dna_w_reference_seq",
    "nolzss_write_factors_binary_file", "nolzss_write_factors_binary_file_dna_w_rc",
    "nolzss_factorize_w_reference_file", "nolzss_factorize_dna_w_reference_seq_file",
    "nolzss_write_factor_file", "nolzss_prepare_multiple_dna_no_rc", "nolzss_factorize_fasta_multiple_dna",
    "nolzss_free_fasta_result", "nolzss_write_factors_binary_file_fasta_multiple_dna",
    "nolzss_factorize_fasta_per_sequence", "nolzss_free_fasta_per_sequence_result",
    "nolzss_factorize_dna_rc_w_ref_fasta_files", "nolzss_write_factors_dna_w_reference_fasta_files_to_binary",
    "nolzss_free_batch", "nolzss_profile_enable", "nolzss_profile_reset", "nolzss_profile_report",
    "nolzss_debug_arrays", "nolzss_debug_sort_pairs", "nolzss_debug_scan", "nolzss_debug_arena",
    "nolzss_debug_batch_counters", "nolzss_factorize_batch_dna_w_rc",
    "nolzss_debug_trim_arenas", "nolzss_debug_parse_fasta",
    "nolzss_read_nucleotide_fasta", "nolzss_free_nucleotide_fasta", "nolzss_debug_parse_nucleotide_fasta", "nolzss_debug_lpt_plan", "nolzss_debug_batch_plan", "nolzss_factorize_batch_device",
    "nolzss_factorize_dna_w_rc_device",
]


def check(rc):
    """Map a status code to the exception the reference's pybind11 module would raise
    (std::invalid_argument -> ValueError, std::runtime_error -> RuntimeError;
    reference: src/cpp/bindings.cpp default exception translation)."""
    if rc == OK:
        return
    msg = lib.nolzss_last_error().decode("utf-8", "replace")
    if rc == ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)
