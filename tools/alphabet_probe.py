"""Wall and stage times of count_factors on texts of other alphabets (2-, 4- and 8-bit packing): python tools/alphabet_probe.py <log2 n>"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import gen
from nolzss_amd import _noLZSS as native
n = 1 << int(sys.argv[1])
rng = np.random.default_rng(5)
cases = {
    "random DNA (2-bit)": gen.random_dna(n, 2),
    "repeat DNA (2-bit)": gen.repeat_dna(n, 3),
    "random sigma=16 (4-bit)": np.frombuffer(b"ACDEFGHIKLMNPQRS", dtype=np.uint8)[rng.integers(0, 16, n)],
    "random protein sigma=20 (8-bit)": np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)[rng.integers(0, 20, n)],
}
# protein with copied blocks
p = cases["random protein sigma=20 (8-bit)"].copy()
pos = 0
while pos < n:
    L = int(rng.integers(200, 5000))
    if rng.random() < 0.4 and pos > L:
        s = int(rng.integers(0, pos - L)); p[pos:pos + L] = p[s:s + L][: max(0, min(L, n - pos))]
    pos += L
cases["protein with 40 % copied blocks (8-bit)"] = p
if len(sys.argv) > 2:
    cases = {k: v for k, v in cases.items() if sys.argv[2] in k}
native.count_factors(gen.random_dna(1 << 16, 2))
native.profile_enable(True)
for name, t in cases.items():
    native.count_factors(t)
    native.profile_reset()
    t0 = time.time(); z = native.count_factors(t); dt = time.time() - t0
    st = native.profile_report()
    top = sorted(((k, v[1]) for k, v in st.items()), key=lambda kv: -kv[1])[:24]
    print(f"2^{sys.argv[1]} {name}: {dt*1e3:.1f} ms, z={z}\n     " + "  ".join(f"{k} {v:.1f}" for k, v in top), flush=True)
