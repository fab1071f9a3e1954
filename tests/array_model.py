"""Array (SA / ISA / LCP) formulation of the per-position candidates that the HIP kernels
implement -- a slow, direct Python model used only by the CPU tests to validate the
*formulas*  against the oracle's tree walk before they are written as
kernels.  TEST INFRASTRUCTURE ONLY; nothing in nolzss_amd imports it.

Notation: r = ISA[i];  I(d) = maximal rank interval around r with LCP[lo+1..hi] >= d.
"""
import numpy as np

INF = 1 << 62


def _interval(lcp, n, r, d):
    lo = r
    while lo > 0 and lcp[lo] >= d:
        lo -= 1
    hi = r
    while hi + 1 < n and lcp[hi + 1] >= d:
        hi += 1
    return lo, hi


def _nearest(sa, lcp, n, r, pred):
    """(lcp to nearest rank above r satisfying pred, its SA) and the same below; lcp 0 if none."""
    res = []
    m = INF
    q = r - 1
    found = (0, None)
    while q >= 0:
        m = min(m, lcp[q + 1])
        if m == 0:
            break
        if pred(sa[q]):
            found = (m, int(sa[q]))
            break
        q -= 1
    res.append(found)
    m = INF
    q = r + 1
    found = (0, None)
    while q < n:
        m = min(m, lcp[q])
        if m == 0:
            break
        if pred(sa[q]):
            found = (m, int(sa[q]))
            break
        q += 1
    res.append(found)
    return res


def _range_lcp(lcp, isa, a, b):
    ra, rb = sorted((isa[a], isa[b]))
    return int(min(lcp[ra + 1:rb + 1]))


def lstar_plain(sa, isa, lcp, n, i, stats=None):
    """L*[i] = max{d : min SA[I(d)] + d <= i}: fast path from the two nearest-smaller
    neighbours, exact binary search only when the best neighbour overlaps position i."""
    r = isa[i]
    (lp, jp), (ls, js) = _nearest(sa, lcp, n, r, lambda v: v < i)
    M = max(lp, ls)
    if M == 0:
        return 0
    cands = [(l, j) for (l, j) in ((lp, jp), (ls, js)) if l == M]
    if any(i - j >= M for _, j in cands):
        return M
    if stats is not None:
        stats["fallback"] = stats.get("fallback", 0) + 1
    lo = max(min(l, i - j) for (l, j) in ((lp, jp), (ls, js)) if l > 0)
    hi = M
    while lo < hi:  # largest d in [lo, hi] with P(d); P(lo) is known true
        mid = (lo + hi + 1) // 2
        a, b = _interval(lcp, n, r, mid)
        if int(sa[a:b + 1].min()) + mid <= i:
            lo = mid
        else:
            hi = mid - 1
    return lo


def factor_plain(sa, isa, lcp, n, i):
    L = lstar_plain(sa, isa, lcp, n, i)
    if L == 0:
        return (1, i)
    a, b = _interval(lcp, n, isa[i], L)
    return (L, int(sa[a:b + 1].min()))


def factor_rc(sa, isa, lcp, m, N, i):
    """Emitted (len, ref, is_rc) of nolzss_multiple_dna_w_rc with the cursor at i < N."""
    r = isa[i]
    # forward: L_f = plain L* over S; then the explicit-node quirk
    Lf = lstar_plain(sa, isa, lcp, m, i)
    fwd, jf = 0, None
    if Lf > 0:
        a, b = _interval(lcp, m, r, Lf + 1)
        d_u = max(lcp[a], lcp[b + 1] if b + 1 < m else 0)
        a, b = _interval(lcp, m, r, d_u)
        jf = int(sa[a:b + 1].min())
        fwd = min(_range_lcp(lcp, isa, i, jf), i - jf)
    # reverse complement: nearest ranks whose suffix starts after 2N - i (and is not the
    # final sentinel): E = 2N - SA < i
    thr = 2 * N - i
    (lp, _), (ls, _) = _nearest(sa, lcp, m, r, lambda v: thr < v < m - 1)
    rc = max(lp, ls)
    end = None
    if rc > 0:
        a, b = _interval(lcp, m, r, rc)
        seg = sa[a:b + 1]
        seg = seg[(seg > thr) & (seg < m - 1)]
        end = 2 * N - int(seg.max())
    if fwd >= 1:
        if rc > fwd:
            return (rc, end - rc + 1, True)
        return (fwd, jf, False)
    if rc > 1:
        return (rc, end - rc + 1, True)
    return (1, i, False)


def factor_rc_fastpath(sa, isa, lcp, m, N, i):
    """Same result as factor_rc, organised the way rc_candidates_kernel / rc_fallback_kernel are:
    when a best forward neighbour does not overlap position i the forward length is final and no
    explicit-node bookkeeping is needed; reverse-complement candidates shorter than what the
    forward side already has are pruned."""
    r = isa[i]
    (lp, jp), (ls, js) = _nearest(sa, lcp, m, r, lambda v: v < i)
    M = max(lp, ls)
    final = M == 0 or any(l == M and i - j >= M for (l, j) in ((lp, jp), (ls, js)) if l > 0)
    thr = 2 * N - i
    (ru, _), (rd, _) = _nearest(sa, lcp, m, r, lambda v: v > thr)
    rc = max(ru, rd)
    if final:
        fwd = M
    else:
        Lf = lstar_plain(sa, isa, lcp, m, i)
        a, b = _interval(lcp, m, r, Lf + 1)
        d_u = max(lcp[a], lcp[b + 1] if b + 1 < m else 0)
        a, b = _interval(lcp, m, r, d_u)
        j = int(sa[a:b + 1].min())
        fwd = min(_range_lcp(lcp, isa, i, j), i - j)
    if fwd >= 1:
        use_rc = rc > fwd
    else:
        use_rc = rc > 1
        if not use_rc:
            return (1, i, False)
    L = rc if use_rc else fwd
    a, b = _interval(lcp, m, r, L)
    if use_rc:
        end = 2 * N - int(sa[a:b + 1].max())
        return (L, end - L + 1, True)
    return (L, int(sa[a:b + 1].min()), False)
