"""Definition-level brute-force models (pure Python, tiny inputs only).

Two independent checks of the oracle:
  * `plain_closed_form`  -- textbook LPnF:  L*[i] = max_{j<i} min(lcp(i,j), i-j),
    ref = leftmost occurrence of T[i:i+L*]  (SURVEY.md Appendix A1/A3);
  * `plain_tree_walk` / `rc_tree_walk` -- the reference's top-down walk over the
    explicit suffix-tree ancestors of leaf(i) (factorizer_core.hpp:66-109 and
    :241-379), with the tree replaced by occurrence sets computed by string
    comparison.  Nothing here shares code with oracle/ or with the HIP path.
"""

RC_MASK = 1 << 63
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def _lcp(t, a, b):
    h = 0
    n = len(t)
    while a + h < n and b + h < n and t[a + h] == t[b + h]:
        h += 1
    return h


def plain_closed_form_at(t, i):
    best = 0
    for j in range(i):
        best = max(best, min(_lcp(t, i, j), i - j))
    if best == 0:
        return (i, 1, i)
    pat = t[i:i + best]
    ref = t.find(pat)
    return (i, best, ref)


def _occ(t, i, d):
    """start positions p with t[p:p+d] == t[i:i+d] (requires i+d <= len(t))."""
    pat = t[i:i + d]
    out = []
    p = t.find(pat)
    while p != -1:
        out.append(p)
        p = t.find(pat, p + 1)
    return out


def _explicit_ancestors(t, i):
    """[(depth, occurrence list)] of the explicit internal ancestors of leaf(i), shallowest
    first, excluding the root.  A depth d is an explicit node iff the set of occurrences of
    t[i:i+d] strictly shrinks when the string is extended by one more symbol (the text's end
    acts as the unique terminator the reference's CST appends)."""
    n = len(t)
    res = []
    for d in range(1, n - i + 1):
        occ = _occ(t, i, d)
        if len(occ) < 2:
            break
        nxt = _occ(t, i, d + 1) if i + d + 1 <= n else []
        if len(nxt) < len(occ):
            res.append((d, occ))
    return res


def plain_tree_walk_at(t, i, u_min_stale=0):
    n = len(t)
    nodes = _explicit_ancestors(t, i) + [(n - i + 1, [i])]  # the leaf, depth counts terminator
    u_min, u_depth, u_root = u_min_stale, 0, True
    for depth, occ in nodes:
        v_min = min(occ)
        if v_min + depth - 1 < i:
            u_min, u_depth, u_root = v_min, depth, False
            continue
        if v_min == i:
            if u_root:
                return (i, 1, i), u_min
            return (i, u_depth, u_min), u_min
        l = min(_lcp(t, i, v_min), i - v_min)
        if l <= u_depth:
            return (i, u_depth, u_min), u_min
        return (i, l, v_min), u_min
    raise AssertionError("walk did not terminate")


def plain_factorize(t, walk=True, start_pos=0):
    out, i, stale = [], start_pos, 0
    while i < len(t):
        if walk:
            f, stale = plain_tree_walk_at(t, i, stale)
        else:
            f = plain_closed_form_at(t, i)
        out.append(f)
        i += f[1]
    return out


def revcomp(s):
    return "".join(_COMP[c] for c in reversed(s))


def sentinel_for(index):
    """k-th value of 1,2,3,... skipping 0 and A,C,G,T, wrapping past 255 back to 1
    (factorizer.cpp:110-125)."""
    s, count = 1, 0
    while True:
        if s != 0 and chr(s) not in "ACGT":
            if count == index:
                return s
            count += 1
        s = (s + 1) & 0xFF
        if s == 0:
            s = 1


def prepare_w_rc(seqs):
    """-> (S as a str of latin-1 code points, original_length, sentinel positions)."""
    seqs = [s.upper() for s in seqs if s]
    out, sent, k = [], [], 0
    pos = 0
    for s in seqs:
        out.append(s)
        pos += len(s)
        sent.append(pos)
        out.append(chr(sentinel_for(k)))
        pos += 1
        k += 1
    orig = pos
    for s in reversed(seqs):
        out.append(revcomp(s))
        pos += len(s)
        sent.append(pos)
        out.append(chr(sentinel_for(k)))
        pos += 1
        k += 1
    return "".join(out), orig, sent


def rc_tree_walk_at(S, N, i):
    INF = float("inf")
    m = len(S)
    nodes = _explicit_ancestors(S, i) + [(m - i + 1, [i])]
    have_f = have_r = False
    bf_start = bf_depth = br_end = br_pos = br_depth = 0
    for ell, occ in nodes:
        jF = min([p for p in occ if p < N], default=INF)
        ends = [(N - (p - (N + 1)) - 1, p) for p in occ if N + 1 <= p < m - 1]
        endRC, posR = min(ends, default=(INF, None))
        okF = jF != INF and jF + ell - 1 < i
        okR = endRC != INF and endRC < i
        if not okF and not okR:
            break
        if okF and ell > bf_depth:
            bf_depth, bf_start, have_f = ell, jF, True
        if okR and ell > br_depth:
            br_depth, br_end, br_pos, have_r = ell, endRC, posR, True
    if not have_f and not have_r:
        return (i, 1, i, False)
    fwd = min(_lcp(S, i, bf_start), i - bf_start) if have_f else 0
    rc = _lcp(S, i, br_pos) if have_r else 0
    if have_f and fwd >= 1:
        use_fwd, literal = not (have_r and rc > fwd), False
    else:
        use_fwd, literal = False, not (have_r and rc > 1)
    if literal:
        return (i, 1, i, False)
    if use_fwd:
        return (i, fwd, bf_start, False)
    return (i, rc, br_end - rc + 1, True)


def rc_factorize_prepared(S, start_pos=0):
    N = len(S) // 2 - 1
    out, i = [], start_pos
    while i < N:
        f = rc_tree_walk_at(S, N, i)
        out.append(f)
        i += f[1]
    return out


def rc_factorize(seq):
    S, _, _ = prepare_w_rc([seq])
    return rc_factorize_prepared(S)
