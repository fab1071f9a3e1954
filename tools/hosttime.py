"""Host-buffer entry points at 2^30 bases: how much the upload / download add to the device time."""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests'))
import gen
from nolzss_amd import _noLZSS as native
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
text = gen.repeat_dna(1 << lg, seed=0x5EED0003)
native.count_factors(text[: 1 << 20])
for name, fn in (("count_factors (host buffer in)", lambda: native.count_factors(text)),
                 ("factorize_array (host buffer in, records out)", lambda: len(native.factorize_array(text)))):
    best = 1e9
    for rep in range(3):
        t0 = time.time(); z = fn(); best = min(best, time.time() - t0)
    print(f"2^{lg} bases, {name}: {best*1e3:.1f} ms = {(1<<lg)/best/1e9:.2f} Gbases/s (z={z})", flush=True)
native.profile_enable(True); native.profile_reset(); native.count_factors(text)
st = native.profile_report()
print({k: round(v[1], 2) for k, v in st.items() if k in ("text_h2d", "factors_d2h")})
import tempfile, os
path = os.path.join(tempfile.mkdtemp(), "text.bin")
text.tofile(path)
best = 1e9
for rep in range(3):
    t0 = time.time(); z = native.count_factors_file(path); best = min(best, time.time() - t0)
print(f"2^{lg} bases, count_factors_file (cached file in): {best*1e3:.1f} ms = {(1<<lg)/best/1e9:.2f} Gbases/s (z={z})", flush=True)
os.remove(path)
