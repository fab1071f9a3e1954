import os
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fullsize_oracle(mode): needs the oracle's result for a BASELINE configuration "
                                       "at full size; the child process is started when the session begins")


class _FullsizeOracles:
    """The oracle at the full size of BASELINE configs 3 and 5 takes minutes on one core: one child process
    per configuration (tests/fullsize_oracle.py) is started as soon as the selected tests are known and works
    while the rest of the suite runs; the tests that need a result wait for it."""

    # host memory one child needs at the full size (tests/fullsize_oracle.py) -- on top of the suite's own arrays
    NEED_GIB = {"plain": 45, "rc": 24}
    HEADROOM_GIB = 24

    def __init__(self):
        self.dir = None
        self.children = {}
        self.skip_reason = None  # set when the children are not started: the tests fall back to a prefix of the text

    @staticmethod
    def mem_available_gib():
        try:
            for line in open("/proc/meminfo"):
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / (1 << 20)
        except OSError:
            pass
        return None

    def start(self, modes):
        # The children are not started where they cannot finish: an explicit opt-out, a pytest-xdist worker (every
        # worker would start its own pair), or a host without the memory (the session would be OOM-killed).
        if os.environ.get("NOLZSS_SKIP_FULLSIZE"):
            self.skip_reason = "NOLZSS_SKIP_FULLSIZE is set"
        elif os.environ.get("PYTEST_XDIST_WORKER"):
            self.skip_reason = "running under pytest-xdist: the full-size oracle children are started by a plain session only"
        else:
            need = sum(self.NEED_GIB.get(m, 0) for m in modes) + self.HEADROOM_GIB
            have = self.mem_available_gib()
            if have is not None and have < need:
                self.skip_reason = f"MemAvailable {have:.0f} GiB < {need} GiB needed by the full-size oracle children"
        if self.skip_reason:
            return
        base = "/dev/shm" if os.path.isdir("/dev/shm") else None
        self.dir = Path(tempfile.mkdtemp(prefix="nolzss_fullsize_", dir=base))
        for mode in sorted(modes):
            out = self.dir / mode
            out.mkdir()
            log = open(out / "log.txt", "wb")
            self.children[mode] = subprocess.Popen(
                [sys.executable, str(ROOT / "tests" / "fullsize_oracle.py"), mode, str(out)],
                stdout=log, stderr=subprocess.STDOUT, cwd=str(ROOT))

    def result(self, mode, timeout_s):
        import numpy as np
        child = self.children[mode]
        out = self.dir / mode
        try:
            child.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            child.kill()
            pytest.fail(f"the oracle child for '{mode}' did not finish within {timeout_s} s")
        if not (out / "done").exists():
            reason = (out / "error").read_text() if (out / "error").exists() else (out / "log.txt").read_text()
            pytest.fail(f"the oracle child for '{mode}' failed (exit {child.returncode}): {reason[-2000:]}")
        return {k: np.load(out / f"{k}.npy") for k in ("start", "length", "ref")}, (out / "done").read_text().strip()

    def stop(self):
        for child in self.children.values():
            if child.poll() is None:
                child.kill()
                child.wait()
        if self.dir is not None:
            shutil.rmtree(self.dir, ignore_errors=True)


_oracles = _FullsizeOracles()


def pytest_collection_finish(session):
    if session.config.option.collectonly:
        return
    modes = {m.args[0] for item in session.items for m in item.iter_markers("fullsize_oracle")}
    if modes:
        _oracles.start(modes)


def pytest_sessionfinish(session, exitstatus):
    _oracles.stop()


@pytest.fixture
def oracle_children():
    return _oracles
