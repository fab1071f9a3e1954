"""Group-size statistics of the K-symbol key sort: python tools/group_hist.py <log2 n> [K]
(how many suffixes share their first K symbols with how many others -- the input of the direct
comparison round).  Uses the debug entry point that returns SA / LCP."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gen
from nolzss_amd import _noLZSS as native

n = 1 << int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 17
text = gen.repeat_dna(n)
lcp = native.debug_arrays(text)["lcp"]
lcp = np.asarray(lcp)[:n]
tied = lcp >= K                      # rank r tied with rank r-1
# run lengths of consecutive tied flags -> group size = run + 1
edges = np.flatnonzero(np.diff(np.concatenate(([0], tied.view(np.int8), [0]))))
runs = edges[1::2] - edges[0::2]
gs = runs + 1
print(f"n={n} K={K}: {gs.sum()} tied suffixes in {len(gs)} groups; mean size {gs.mean():.2f}; "
      f"sum gs^2 / sum gs = {(gs.astype(np.float64)**2).sum() / gs.sum():.1f}")
for lo, hi in [(2, 2), (3, 3), (4, 4), (5, 8), (9, 16), (17, 32), (33, 64), (65, 256), (257, 1 << 30)]:
    sel = gs[(gs >= lo) & (gs <= hi)]
    print(f"  size {lo:>4}..{hi:<10}: {len(sel):>10} groups {sel.sum():>11} members ({100.0 * sel.sum() / gs.sum():5.1f} %)")
for depth in (81, 145, 273, 529):
    t2 = lcp >= depth
    e2 = np.flatnonzero(np.diff(np.concatenate(([0], t2.view(np.int8), [0]))))
    g2 = (e2[1::2] - e2[0::2]) + 1
    print(f"  still tied at {depth}: {g2.sum()} members, sum gs^2/sum gs = {(g2.astype(np.float64)**2).sum() / max(1, g2.sum()):.1f}")
