"""Child process of the full-size exact tests (tests/test_zz_gpu_fullsize_exact.py): the oracle on one
BASELINE configuration at its stated size, result arrays left in a directory.

    python tests/fullsize_oracle.py plain|rc OUTDIR [log2n]

plain: config 3, 2^30 bases of gen.repeat_dna(seed 0x5EED0003) through oracle_factorize
       (about 45 GB of host memory and a few minutes on one core);
rc:    config 5, 2^28 bases of gen.repeat_dna(seed 0x5EED0005) + reverse-complement strand through
       oracle_factorize_multiple_dna_w_rc (about 24 GB).
Writes start.npy / length.npy / ref.npy, then `done` (or `error` with the message).  Test infrastructure:
started by tests/conftest.py when the session holds the full-size tests, so that the CPU work overlaps
the rest of the GPU suite."""
import sys
import time
import traceback
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))

CONFIGS = {"plain": (30, 0x5EED0003), "rc": (28, 0x5EED0005)}


def text_of(mode: str, log2n: int = 0) -> np.ndarray:
    import gen
    log2, seed = CONFIGS[mode]
    return gen.repeat_dna(1 << (log2n or log2), seed=seed)


def main() -> int:
    mode, out = sys.argv[1], Path(sys.argv[2])
    log2n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    try:
        import oracle_lib as oracle
        t0 = time.time()
        text = text_of(mode, log2n)
        t1 = time.time()
        if mode == "plain":
            f = oracle.factors_array(text)
        else:
            S, _, _ = oracle.prepare_multiple_dna_w_rc([text.tobytes()])
            del text
            f = oracle.factors_array_multiple_dna_w_rc(S)
        for k in ("start", "length", "ref"):
            np.save(out / f"{k}.npy", np.ascontiguousarray(f[k]))
        (out / "done").write_text(f"{len(f)} factors, generate {t1 - t0:.1f} s, oracle {time.time() - t1:.1f} s\n")
        return 0
    except BaseException:  # noqa: BLE001 -- the parent reads the reason from the file
        (out / "error").write_text(traceback.format_exc())
        return 1


if __name__ == "__main__":
    sys.exit(main())
