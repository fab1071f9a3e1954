"""GPU parity of the building blocks (device scans and radix sort) against numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from nolzss_amd import _noLZSS
    assert _noLZSS.device_count() >= 1, "no MI355X visible"
    return _noLZSS


@pytest.mark.parametrize("n", [1, 63, 64, 100, 4095, 4096, 4097, 1_000_003, (1 << 24) + 5])
def test_scans(native, n):
    rng = np.random.default_rng(n)
    x = rng.integers(0, 5, size=n, dtype=np.uint32)
    got = native.debug_scan(x, 0)
    exp = np.concatenate([[0], np.cumsum(x[:-1], dtype=np.uint64)]).astype(np.uint32)
    assert np.array_equal(got, exp)
    y = rng.integers(0, 1 << 31, size=n, dtype=np.uint32)
    y[rng.random(n) < 0.9] = 0
    assert np.array_equal(native.debug_scan(y, 1), np.maximum.accumulate(y))


@pytest.mark.parametrize("n", [1, 2, 255, 4096, 4097, 70_001, (1 << 22) + 17])
@pytest.mark.parametrize("kind", ["random", "fewkeys", "sorted"])
def test_radix_sort_pairs_stable(native, n, kind):
    rng = np.random.default_rng(n * 3 + len(kind))
    if kind == "random":
        keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n).astype(np.uint64)
    elif kind == "fewkeys":
        keys = rng.integers(0, 7, size=n).astype(np.uint64) << np.uint64(37)
    else:
        keys = np.arange(n, dtype=np.uint64) // np.uint64(3)
    vals = np.arange(n, dtype=np.uint32)
    k2, v2 = native.debug_sort_pairs(keys, vals)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k2, keys[order])
    assert np.array_equal(v2, vals[order])  # stability: equal keys keep input order
