"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

A plain NumPy / pure-Python restatement of the per-window statistics path of
simonhmartin/genomics_general (reference snapshot 2026-05-29), used only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.

Parity pinning: the reference has no tests or golden vectors of its own, so
this restatement is pinned against OUTPUTS OF THE UNMODIFIED REFERENCE run in the build
container: tests/golden/*.csv were produced by tests/golden/make_golden.py, which invokes
/root/reference/{popgenWindows,ABBABABAwindows,distMat}.py on the committed .geno fixtures;
tests/test_oracle_golden.py checks every function below against those files.

Each function cites the reference file:line it restates (paths relative to the reference root).
Arithmetic dependency of the reference outside its tree: NumPy (unpinned; 2.2.6 here).
"""
import gzip
import itertools
import math
import string

import numpy as np

NUM_OF = {"A": 0, "C": 1, "G": 2, "T": 3, "N": -999}          # genomics.py:33
IUPAC_PAIR = {"A": "AA", "C": "CC", "G": "GG", "K": "GT", "M": "AC", "N": "NN",
              "S": "CG", "R": "AG", "T": "TT", "W": "AT", "Y": "CT"}  # genomics.py:14-15
HOMO_OF = {"A": "A", "C": "C", "G": "G", "T": "T"}              # genomics.py:16 (everything else -> N)


# ----------------------------------------------------------------------------------------------
# text -> sites -> windows
# ----------------------------------------------------------------------------------------------
def open_text(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "rt")


def read_sites(handle, header_line=None):
    """genomics.py:1914-1945 (GenoFileReader) + 1884-1904 (parseGenoLine): header first, '#' lines
    skipped, whitespace split.  Returns (names, [(scaffold, pos, [cells])...])."""
    it = iter(handle)
    header = header_line if header_line else next(it)
    names = header.split()[2:]
    sites = []
    for line in it:
        if not line or line[0] == "#":
            continue
        parts = line.split()
        if not parts:
            break                                   # the reference treats an empty line as EOF
        sites.append((parts[0], int(parts[1]), parts[2:]))
    return names, sites


class Win:
    """The fields of GenoWindow the drivers read (genomics.py:1721-1797)."""
    __slots__ = ("scaffold", "start", "end", "rows", "positions", "ID")

    def __init__(self, scaffold, start, end, rows, positions, ID):
        self.scaffold, self.start, self.end = scaffold, start, end
        self.rows, self.positions, self.ID = rows, positions, ID

    def mid(self):                                   # genomics.py:1795-1797
        if not self.positions:
            return float("nan")
        return int(round(sum(self.positions) / len(self.positions)))


def _scaffold_wanted(scaf, include, exclude):       # genomics.py:2016
    if not include and not exclude:
        return True
    if include:
        return scaf in include
    return scaf not in exclude


def coord_windows(sites, wind_size, step, include=None, exclude=None):
    """genomics.py:1971-2027 slidingCoordWindows, restated as the same state machine over a site cursor.
    Faithful quirks: empty windows are emitted; the window preceding a skipped (excluded / not included)
    scaffold is emitted a second time (the generator keeps the old window object across the skip,
    genomics.py:2016-2023); the EOF infinite loop under --include is NOT reproduced (we stop at EOF)."""
    out = []
    done = 0
    n = len(sites)
    i = 0
    w_scaf, lo, hi, rows, poss, wid = None, None, None, [], [], None
    while i < n:
        while i < n and sites[i][0] == w_scaf and sites[i][1] <= hi:
            if sites[i][1] >= lo:
                rows.append(sites[i][2])
                poss.append(sites[i][1])
            i += 1
        if w_scaf is not None:
            done += 1
            out.append(Win(w_scaf, lo, hi, rows[:], poss[:], wid))
        nxt = sites[i][0] if i < n else None
        if nxt == w_scaf:
            lo, hi = lo + step, hi + step                         # GenoWindow.slide 1767-1777
            k = 0
            while k < len(poss) and poss[k] < lo:
                k += 1
            rows, poss = rows[k:], poss[k:]
            wid = done + 1
        elif i >= n:
            break
        elif _scaffold_wanted(nxt, include, exclude):
            w_scaf, lo, hi, rows, poss, wid = nxt, 1, wind_size, [], [], done + 1
        else:
            while i < n and (sites[i][0] == nxt or not _scaffold_wanted(sites[i][0], include, exclude)):
                i += 1
    return out


def sites_windows(sites, wind_sites, overlap, max_dist=float("inf"), min_sites=None,
                  include=None, exclude=None):
    """genomics.py:2032-2108 slidingSitesWindows as the same state machine (same quirks as above;
    trim(leave=overlap) is positions[len-overlap:], genomics.py:1779-1788)."""
    if not min_sites:
        min_sites = wind_sites
    out = []
    done = 0
    n = len(sites)
    i = 0
    w_scaf, rows, poss, wid = None, [], [], None
    while True:
        while (i < n and sites[i][0] == w_scaf and len(poss) < wind_sites
               and (len(poss) == 0 or sites[i][1] - min(poss) <= max_dist)):
            rows.append(sites[i][2])
            poss.append(sites[i][1])
            i += 1
        nxt = sites[i][0] if i < n else None
        emitted = len(poss) >= min_sites
        if emitted:
            done += 1
            out.append(Win(w_scaf, min(poss), max(poss), rows[:], poss[:], wid))
        if nxt == w_scaf and i < n:
            if emitted:
                remove = len(poss) - overlap
                rows, poss = rows[remove:], poss[remove:]
                wid = done + 1
            else:
                rows, poss = rows[1:], poss[1:]
        elif i >= n:
            break
        elif _scaffold_wanted(nxt, include, exclude):
            w_scaf, rows, poss, wid = nxt, [], [], done + 1
        else:
            while i < n and (sites[i][0] == nxt or not _scaffold_wanted(sites[i][0], include, exclude)):
                i += 1
            if i >= n:
                break
    return out


def predefined_windows(sites, coords):
    """genomics.py:2112-2171 predefinedCoordWindows.  coords: [(scaffold, start, end[, ID])]."""
    all_scafs = [c[0] for c in coords]
    scafs = sorted(set(all_scafs), key=all_scafs.index)
    out = []
    i, n = 0, len(sites)
    cur_scaf, rows, poss = None, [], []
    for c in coords:
        wid = c[3] if len(c) > 3 else "NA"
        if cur_scaf is not None and cur_scaf == c[0]:
            k = 0
            while k < len(poss) and poss[k] < c[1]:
                k += 1
            rows, poss = rows[k:], poss[k:]
        else:
            cur_scaf, rows, poss = c[0], [], []
        lo, hi = c[1], c[2]
        widx = scafs.index(cur_scaf)
        while i < n and (sites[i][0] not in scafs or scafs.index(sites[i][0]) < widx):
            bad = sites[i][0]
            while i < n and sites[i][0] == bad:
                i += 1
        while i < n and sites[i][0] == cur_scaf and sites[i][1] < lo:
            i += 1
        while i < n and sites[i][0] == cur_scaf and lo <= sites[i][1] <= hi:
            rows.append(sites[i][2])
            poss.append(sites[i][1])
            i += 1
        out.append(Win(cur_scaf, lo, hi, rows[:], poss[:], wid))
        if i >= n:
            break
    return out


# ----------------------------------------------------------------------------------------------
# window -> alignment
# ----------------------------------------------------------------------------------------------
class Aln:
    """numArray / nanMask / names / sampleNames / groups of genomics.Alignment (genomics.py:808-869)."""

    def __init__(self, num, names, sample_names, groups):
        self.num = num                               # int64 [N][L], -999 = missing
        self.mask = num >= 0                         # genomics.py:834
        self.names = list(names)
        self.sample_names = list(sample_names)
        self.groups = list(groups)
        self.N, self.L = num.shape


def split_cell(cell, fmt, ploidy):
    """genomics.py:390-396 splitSeq applied to one cell (the reference zips whole columns; the
    per-cell view is identical when all cells have the expected width)."""
    if fmt == "diplo":
        cell = IUPAC_PAIR[cell]
    if fmt == "phased":
        alleles = cell[::2]
    else:
        alleles = cell
    assert len(alleles) == ploidy, "Sample ploidy (%d) doesn't match number of sequences (%d)" % (ploidy, len(alleles))
    return alleles


def window_to_aln(win, file_names, ind_names, pop_of, ploidy_of, fmt):
    """genomics.py:1101-1127 genoToAlignment on GenoWindow.seqDict() (1790-1793).
    ind_names: SampleData.indNames order; pop_of: name -> group label (or None)."""
    col = {nm: file_names.index(nm) for nm in ind_names}
    hap_names, samp, grp, seqs = [], [], [], []
    for nm in ind_names:
        pl = ploidy_of[nm]
        cells = [row[col[nm]] for row in win.rows]
        if pl is None:
            # ploidy left open (--inferPloidy): the number of sequences splitSeq returns for the window, genomics.py:1110 --
            # zip(*cells) stops at the shortest cell (390-396)
            short = min((len(IUPAC_PAIR[c]) if fmt == "diplo" else len(c)) for c in cells) if cells else 0
            pl = (short + 1) // 2 if fmt == "phased" else short
            cells = [c if fmt == "diplo" else c[:(2 * pl - 1 if fmt == "phased" else pl)] for c in cells]
        per_site = [split_cell(c, fmt, pl) for c in cells]
        if pl != 1:
            for k in range(pl):
                hap_names.append(nm + "_" + string.ascii_uppercase[k])
                samp.append(nm)
                grp.append(pop_of.get(nm))
                seqs.append([a[k] for a in per_site])
        else:
            hap_names.append(nm)
            samp.append(nm)
            grp.append(pop_of.get(nm))
            seqs.append([HOMO_OF.get(a[0], "N") for a in per_site])     # forceHomo 407-408
    order = np.argsort(hap_names)                                           # genomics.py:1122
    L = len(win.rows)
    num = np.full((len(seqs), L), -999, dtype=np.int64)
    for r, o in enumerate(order):
        s = seqs[o]
        for x in range(L):
            num[r, x] = NUM_OF.get(s[x], -999)      # non-ACGTN is undefined in the reference; we say missing
    return Aln(num, [hap_names[o] for o in order], [samp[o] for o in order], [grp[o] for o in order])


# ----------------------------------------------------------------------------------------------
# numeric core
# ----------------------------------------------------------------------------------------------
def pair_counts_loop(aln):
    """genomics.py:903-916 (pairDist/distMatrix) + 1219-1221 (numHamming) + 1042-1047 (pairNonNan),
    pair by pair exactly as the reference loops.  Returns integer D (both called & differ) and
    C (both called), symmetric, zero diagonal."""
    N = aln.N
    D = np.zeros((N, N), dtype=np.int64)
    C = np.zeros((N, N), dtype=np.int64)
    for i in range(N - 1):
        for j in range(i + 1, N):
            m = aln.mask[i] & aln.mask[j]
            dif = aln.num[i][m] - aln.num[j][m]
            D[i, j] = D[j, i] = int(np.sum(dif != 0))
            C[i, j] = C[j, i] = int(np.sum(m))
    return D, C


def pair_counts_gemm(aln, dtype=np.float64):
    """Same integers as pair_counts_loop via D = C - sum_b X_b X_b^T, C = V V^T (exact in float64 for
    counts < 2^53; SURVEY.md 8c verified the identity against Alignment.distMatrix).  dtype=np.float32 is exact as
    long as a window has fewer than 2^24 sites (every partial sum is an integer below 2^24) and halves time and memory
    for the 2000-haplotype windows of BASELINE.json's distMat configuration."""
    if dtype == np.float32:
        assert aln.L < (1 << 24)
    V = aln.mask.astype(dtype)
    C = V @ V.T
    same = np.zeros_like(C)
    for b in range(4):
        X = (aln.num == b).astype(dtype)
        same += X @ X.T
    D = C - same
    C = C.astype(np.int64)
    D = D.astype(np.int64)
    np.fill_diagonal(C, 0)
    np.fill_diagonal(D, 0)
    return D, C


def dist_from_counts(D, C):
    """distMatrix(): np.mean(bool) = count/len, nan when no jointly called site; diagonal 0."""
    with np.errstate(divide="ignore", invalid="ignore"):
        dm = D.astype(np.float64) / C.astype(np.float64)
    np.fill_diagonal(dm, 0.0)
    return dm


def nanmean_min(a, minimum=0):
    """genomics.py:88-90."""
    if a.size == 0:
        return np.nan
    if 1 - (1. * np.isnan(a).sum() / a.size) < minimum:
        return np.nan
    if np.all(np.isnan(a)):
        return np.nan
    return np.nanmean(a)


def group_dist_stats(aln, D, C, do_pairs=True, min_sites=None, min_data=0.01):
    """genomics.py:956-995 groupDistStats.  Returns (stats dict, masked distance matrix)."""
    dm = dist_from_counts(D, C)
    if min_sites:
        dm[C < min_sites] = np.nan
    np.fill_diagonal(dm, np.nan)
    pops, inv = np.unique(np.array(aln.groups), return_inverse=True)
    idx = [list(np.where(inv == x)[0]) for x in range(len(pops))]
    out = {}
    for x in range(len(pops)):
        out["pi_" + pops[x]] = nanmean_min(dm[np.ix_(idx[x], idx[x])], min_data)
    if len(pops) == 1 or not do_pairs:
        return out, dm
    for x in range(len(pops) - 1):
        for y in range(x + 1, len(pops)):
            dxy = nanmean_min(dm[np.ix_(idx[x], idx[y])], min_data)
            out["dxy_%s_%s" % (pops[x], pops[y])] = out["dxy_%s_%s" % (pops[y], pops[x])] = dxy
            nx, ny = len(idx[x]), len(idx[y])
            w = 1. * nx / (nx + ny)
            pi_s = w * out["pi_" + pops[x]] + (1 - w) * out["pi_" + pops[y]]
            both = idx[x] + idx[y]
            pi_t = nanmean_min(dm[np.ix_(both, both)], min_data)
            with np.errstate(divide="ignore", invalid="ignore"):
                fst = 1 - np.float64(pi_s) / np.float64(pi_t)
            out["Fst_%s_%s" % (pops[x], pops[y])] = out["Fst_%s_%s" % (pops[y], pops[x])] = fst
    return out, dm


def ind_pair_dists(aln, dm, include_same=False):
    """genomics.py:934-954 indPairDists on an (optionally already masked) distance matrix `dm`
    (the reference mutates its cached matrix, so a preceding groupDistStats leaves its mask)."""
    dm = dm.copy()
    if not include_same:
        np.fill_diagonal(dm, np.nan)
    names, first = [], {}
    for k, s in enumerate(aln.sample_names):         # uniqueIndices(preserveOrder) 1160-1164
        if s not in first:
            first[s] = []
            names.append(s)
        first[s].append(k)
    out = {a: {} for a in names}
    for a in names:
        for b in names:
            blk = dm[np.ix_(first[a], first[b])]
            out[a][b] = np.nan if np.all(np.isnan(blk)) else np.nanmean(blk)
    return out, dm


def sample_het(aln, dm, C):
    """genomics.py:918-929 sampleHet with a cached distance matrix, operator precedence included:
    `len(x)==2 & np.sum(mask & mask) >= _minSites` is `len(x) == (2 & C) >= 1`."""
    first = {}
    for k, s in enumerate(aln.sample_names):
        first.setdefault(s, []).append(k)
    out = {}
    for s, x in first.items():
        ok = False
        if len(x) >= 2:
            c = int(C[x[0], x[1]])
            ok = (len(x) == (2 & c)) and ((2 & c) >= 1)
        out["het_" + s] = dm[x[0], x[1]] if ok else np.nan
    return out


def h12_stats(aln, dm, max_dist=0):
    """genomics.py:1079-1098 H12stats + 1239-1261 distMat_to_cluster_sizes."""
    g = np.array(aln.groups, dtype=object)
    out = {}
    for name in np.unique(np.array(aln.groups)):
        idx = np.where(g == name)[0]
        with np.errstate(invalid="ignore"):
            match = dm[np.ix_(idx, idx)] <= max_dist
        sizes = []
        while match.shape[0] > 0:
            most = match.sum(axis=1).argmax()
            m = match[most, ].sum()
            if m > 1:
                sizes.append(m)
                keep = np.invert(match[most, ])
                match = match[np.ix_(keep, keep)]
            else:
                sizes += [1] * match.shape[0]
                break
        sizes = np.array(sizes)
        f = sizes / sizes.sum()
        H1 = (f ** 2).sum()
        if len(f) > 1:
            H12, H2 = H1 + 2 * f[0] * f[1], (f[1:] ** 2).sum()
        else:
            H12, H2 = H1, 0
        out["H1_" + name], out["H12_" + name], out["H2_" + name] = H1, H12, H2
    return out


def site_pop_counts(aln, members):
    """genomics.py:1049-1052 siteFreqs(asCounts) / 592-599 binBaseFreqs for the haplotype rows `members`:
    cnt[L][4] and n[L]."""
    sub = aln.num[members]
    cnt = np.stack([(sub == b).sum(axis=0) for b in range(4)], axis=1).astype(np.int64)
    return cnt, cnt.sum(axis=1)


def abbababa(aln, P1, P2, P3, P4, min_data):
    """genomics.py:1647-1695 ABBABABA(polarize=True) with f4/D/fd/fdm/ABBA/BABA (1409-1475,1565-1569)."""
    g = np.array(aln.groups, dtype=object)
    rows = [np.where(g == p)[0] for p in (P1, P2, P3, P4)]
    cnts, ns = zip(*[site_pop_counts(aln, r) for r in rows])
    tot = cnts[0] + cnts[1] + cnts[2] + cnts[3]
    biallelic = (tot > 0).sum(axis=1) == 2                                   # :1655
    enough = np.ones(aln.L, dtype=bool)
    for k in range(4):
        enough &= (ns[k] * 1. / len(rows[k]) >= min_data)                    # :1657-1660
    good = np.where(biallelic & enough)[0]
    if len(good) < 1:
        # :1693-1695 zips SIX names with SEVEN values ([nan]*6 + [0]): the 0 is dropped and sitesUsed is nan
        return dict(D=np.nan, fd=np.nan, fdM=np.nan, ABBA=np.nan, BABA=np.nan, sitesUsed=np.nan)
    with np.errstate(divide="ignore", invalid="ignore"):
        freqs = [1. * cnts[k][good] / ns[k][good][:, None] for k in range(4)]
        allf = 1. * tot[good] / (ns[0] + ns[1] + ns[2] + ns[3])[good][:, None]
    ai = np.where((allf > 0) & (freqs[3] == 0))                              # :1672
    p1, p2, p3, p4 = (f[ai[0], ai[1]] for f in freqs)

    def f4(a, b, c, d):
        return (1 - a) * b * c * (1 - d) - a * (1 - b) * c * (1 - d)
    abba_t = (1 - p1) * p2 * p3 * (1 - p4)
    baba_t = p1 * (1 - p2) * p3 * (1 - p4)
    with np.errstate(divide="ignore", invalid="ignore"):
        Dv = f4(p1, p2, p3, p4).sum() * 1. / (abba_t + baba_t).sum()
        pd = p2 * (p2 > p3) + p3 * (p3 >= p2)
        fdv = f4(p1, p2, p3, p4).sum() * 1. / f4(p1, pd, pd, p4).sum()
        a, b, x = (p3 > p1), (p3 > p2), (p1 > p2)
        y = ~x
        pdm1 = p3 * (x & a) + p1 * (~(x & a))
        pdm2 = p3 * (y & b) + p2 * (~(y & b))
        pdm3 = -p3 * (x & a) + p3 * (y & b) - p1 * (x & ~a) + p2 * (y & ~b)
        fdm = f4(p1, p2, p3, p4).sum() * 1. / f4(pdm1, pdm2, pdm3, p4).sum()
    return dict(D=Dv, fd=fdv, fdM=fdm, ABBA=abba_t.sum(), BABA=baba_t.sum(), sitesUsed=len(ai[0]))


FOURPOP_STATS = ["ABBA", "BABA", "ABAA", "BAAA", "D", "fd", "fd'", "fdm", "fdm'", "fdh", "fdh2", "fh"]   # fourPopWindows.py:241

# np.argsort(all4freqs, axis=1)[:, 2] on a row with two equal positive entries at bases (i, j): NumPy >= 2.0's x86 SIMD
# argsort network (AVX2 and AVX-512 alike) returns this base; the scalar fallback would return i.  The goldens were
# generated with the SIMD path, which is what current hardware runs.
MINOR_TIE = {(0, 1): 1, (0, 2): 0, (0, 3): 0, (1, 2): 1, (1, 3): 1, (2, 3): 2}


def four_pop(aln, P1, P2, P3, P4, min_data, polarize=False, fixed=False):
    """genomics.py:1585-1643 fourPop with the per-site terms of genomics.py:1409-1563."""
    g = np.array(aln.groups, dtype=object)
    rows = [np.where(g == p)[0] for p in (P1, P2, P3, P4)]
    cnts, ns = zip(*[site_pop_counts(aln, r) for r in rows])
    tot = cnts[0] + cnts[1] + cnts[2] + cnts[3]
    biallelic = (tot > 0).sum(axis=1) == 2                                   # :1593
    enough = np.ones(aln.L, dtype=bool)
    for k in range(4):
        enough &= (ns[k] * 1. / len(rows[k]) >= min_data)                    # :1595-1598
    good = np.where(biallelic & enough)[0]
    if len(good) < 1:
        out = {k: np.nan for k in FOURPOP_STATS}
        out["sitesUsed"] = 0
        return out
    with np.errstate(divide="ignore", invalid="ignore"):
        freqs = [1. * cnts[k][good] / ns[k][good][:, None] for k in range(4)]
    totg = tot[good]
    if polarize:
        ai = np.where((totg > 0) & (freqs[3] == 0))                          # :1610
    elif fixed:
        ai = np.where((totg > 0) & (freqs[3] == 0) & ((freqs[0] == 0) | (freqs[0] == 1)) &
                      ((freqs[1] == 0) | (freqs[1] == 1)) & ((freqs[2] == 0) | (freqs[2] == 1)))   # :1611-1614
    else:
        pick = np.zeros(len(good), dtype=np.int64)                            # :1615, second largest of four
        for r in range(len(good)):
            i, j = np.where(totg[r] > 0)[0]
            pick[r] = i if totg[r, i] < totg[r, j] else j if totg[r, j] < totg[r, i] else MINOR_TIE[(i, j)]
        ai = (np.arange(len(good)), pick)
    p1, p2, p3, p4 = (f[ai[0], ai[1]] for f in freqs)

    def f4(a, b, c, d):
        return (1 - a) * b * c * (1 - d) - a * (1 - b) * c * (1 - d)

    def f4c(a, b, c, d):
        return f4(a, b, c, d) + f4(1 - a, 1 - b, 1 - c, 1 - d)
    with np.errstate(divide="ignore", invalid="ignore"):
        abba_t = (1 - p1) * p2 * p3 * (1 - p4)
        baba_t = p1 * (1 - p2) * p3 * (1 - p4)
        pd = p2 * (p2 > p3) + p3 * (p3 >= p2)
        a, b, x = (p3 > p1), (p3 > p2), (p1 > p2)
        y = ~x
        pdm1 = p3 * (x & a) + p1 * (~(x & a))
        pdm2 = p3 * (y & b) + p2 * (~(y & b))
        pdm3 = -p3 * (x & a) + p3 * (y & b) - p1 * (x & ~a) + p2 * (y & ~b)
        num, numc = f4(p1, p2, p3, p4).sum(), f4c(p1, p2, p3, p4).sum()
        h4 = [f4c(p1, p3, p3, p4), f4c(p4, p2, p3, p4), f4c(p3, p2, p3, p4), f4c(p1, p4, p3, p4)]
        h8 = h4 + [f4c(p1, p2, p2, p4), f4c(p1, p2, p3, p1), f4c(p1, p2, p1, p4), f4c(p1, p2, p3, p2)]
        t1, t2 = np.abs(p1 - p2), np.abs(p3 - p4)
        out = {
            "D": num * 1. / (abba_t + baba_t).sum(),
            "fd": num * 1. / f4(p1, pd, pd, p4).sum(),
            "fd'": numc * 1. / f4c(p1, pd, pd, p4).sum(),
            "fdm": num * 1. / f4(pdm1, pdm2, pdm3, p4).sum(),
            "fdm'": numc * 1. / f4c(pdm1, pdm2, pdm3, p4).sum(),
            "fdh": numc * 1. / (np.amax(h4, axis=0).sum() if len(p1) else np.float64(0)),
            "fdh2": numc * 1. / (np.amax(h8, axis=0).sum() if len(p1) else np.float64(0)),
            "fh": numc * 1. / ((t1 * (t1 > t2) + t2 * (t2 >= t1)) ** 2).sum(),
            "ABBA": abba_t.sum(), "BABA": baba_t.sum(),
            "ABAA": ((1 - p1) * p2 * (1 - p3) * (1 - p4)).sum(), "BAAA": (p1 * (1 - p2) * (1 - p3) * (1 - p4)).sum(),
            "sitesUsed": len(ai[0]),
        }
    return out


def tajima_d(n, S, theta_pi):
    """genomics.py:619-632."""
    a = sum(1. / i for i in range(1, n))
    theta_w = 1. * S / a
    a2 = sum(1. / (i ** 2) for i in range(1, n))
    b1 = (n + 1.) / (3 * (n - 1))
    b2 = (2. * (n ** 2 + n + 3)) / (9 * n * (n - 1))
    c1 = b1 - (1. / a)
    c2 = b2 - ((n + 2) / (a * n)) + a2 / (a ** 2)
    e1 = c1 / a
    e2 = c2 / (a ** 2 + a2)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (theta_pi - theta_w) / np.sqrt(e1 * S + e2 * S * (S - 1))


def group_freq_stats(aln):
    """genomics.py:1002-1028 groupFreqStats + 609-616 baseCountPi.  Sites = no missing data in ANY
    haplotype of the whole alignment (:1010)."""
    out = {}
    full = np.where(np.all(aln.mask, axis=0))[0]
    g = np.array(aln.groups, dtype=object)
    for name in np.unique(np.array(aln.groups)):
        rows = np.where(g == name)[0]
        N = len(rows)
        l = len(full)
        if l >= 1:
            cnt, _ = site_pop_counts(aln, rows)
            c = cnt[full].astype(np.float64)
            pairs = (c[:, 0] * c[:, 1] + c[:, 0] * c[:, 2] + c[:, 0] * c[:, 3]
                     + c[:, 1] * c[:, 2] + c[:, 1] * c[:, 3] + c[:, 2] * c[:, 3])
            with np.errstate(divide="ignore", invalid="ignore"):
                site_pi = pairs / (.5 * N * (N - 1))
            S = int(np.sum(site_pi != 0.))
            theta_pi = 0.
            for v in site_pi:                                    # Python sum(), left to right (:1018)
                theta_pi = theta_pi + v
            with np.errstate(divide="ignore", invalid="ignore"):
                theta_w = S / np.sum(1. / np.arange(1, N))
            taj = tajima_d(N, S, theta_pi)
        else:
            S = theta_pi = theta_w = taj = np.nan
        out["l_" + name], out["S_" + name] = l, S
        out["thetaPi_" + name], out["thetaW_" + name], out["TajD_" + name] = theta_pi, theta_w, taj
    return out


# ----------------------------------------------------------------------------------------------
# drivers: the CSV each CLI writes
# ----------------------------------------------------------------------------------------------
def _fmt(v):
    return str(v)


def _round(v, k):
    if isinstance(v, (int, np.integer)):
        return round(v, k)
    return round(np.float64(v), k)


def make_windows(sites, wind_type, wind_size, step=None, overlap=0, max_dist=float("inf"),
                 min_sites=1, coords=None, include=None, exclude=None):
    if wind_type == "coordinate":
        return coord_windows(sites, wind_size, step or wind_size, include, exclude)
    if wind_type == "sites":
        return sites_windows(sites, wind_size, overlap, max_dist, min_sites, include, exclude)
    return predefined_windows(sites, coords)


def popgen_windows_csv(geno_path, fmt, pops, wind_size, step=None, min_sites=1, min_data=0.01,
                       analysis=("popDist", "popPairDist"), round_to=4, wind_type="coordinate",
                       overlap=0, max_dist=float("inf"), coords=None, add_id=False, write_failed=False,
                       ploidy=None, include=None, exclude=None, counts_fn=pair_counts_gemm, samples_only=None,
                       hap_dist=0):
    """popgenWindows.py:28-75 (stats_wrapper) + 216-354 (setup, header).  pops: [(name, [samples])] in
    CLI order, or None for the single population 'all'."""
    with open_text(geno_path) as fh:
        file_names, sites = read_sites(fh)
    if not min_sites:
        min_sites = wind_size
    if pops is None and not samples_only:
        pops = [("all", list(file_names))]                         # popgenWindows.py:284-291
    pops = pops or []
    pop_names = [p[0] for p in pops]
    ind_names = []
    for _, members in pops:
        for m in members:
            if m not in ind_names:
                ind_names.append(m)
    for m in (samples_only or []):                                 # --samples, popgenWindows.py:279-280
        if m not in ind_names:
            ind_names.append(m)
    if not pops and any(a in analysis for a in ("popFreq", "popDist", "popPairDist")):
        pops = [("all", list(ind_names))]
        pop_names = ["all"]
    pop_of = {}
    for nm in ind_names:
        hit = [p for p, mem in pops if nm in mem]
        pop_of[nm] = hit[0] if len(hit) == 1 else (tuple(hit) if hit else None)
    ploidy_of = {nm: (1 if fmt == "haplo" else 2) for nm in ind_names}
    if ploidy == "infer":                                          # --inferPloidy, popgenWindows.py:299-300
        ploidy_of = {nm: None for nm in ind_names}
    elif ploidy:
        ploidy_of.update(ploidy)
    stats = []
    if "popFreq" in analysis:
        for pre in ("l_", "S_", "thetaPi_", "thetaW_", "TajD_"):
            stats += [pre + n for n in pop_names]
    if "popDist" in analysis:
        stats += ["pi_" + n for n in pop_names]
    if "popPairDist" in analysis:
        stats += ["dxy_%s_%s" % (x, y) for x, y in itertools.combinations(pop_names, 2)]
        stats += ["Fst_%s_%s" % (x, y) for x, y in itertools.combinations(pop_names, 2)]
    if "indPairDist" in analysis:
        stats += ["_".join(["d", i, j]) for i, j in itertools.combinations_with_replacement(sorted(ind_names), 2)]
    if "indHet" in analysis:
        stats += ["het_" + n for n in ind_names]          # the reference's order is that of a Python set (hash dependent)
    if "hapStats" in analysis:
        for pre in ("H1_", "H12_", "H2_"):
            stats += [pre + n for n in pop_names]
    lines = [("windowID," if add_id else "") + "scaffold,start,end,mid,sites," + ",".join(stats)]
    wins = make_windows(sites, wind_type, wind_size, step, overlap, max_dist, min_sites, coords, include, exclude)
    for w in wins:
        n_sites = len(w.positions)
        if n_sites >= min_sites:
            good = True
            aln = window_to_aln(w, file_names, ind_names, pop_of, ploidy_of, fmt)
            sd = {}
            dm = None
            if "popFreq" in analysis:
                sd.update(group_freq_stats(aln))
            D, C = counts_fn(aln)
            if "popDist" in analysis or "popPairDist" in analysis:
                st, dm = group_dist_stats(aln, D, C, "popPairDist" in analysis, min_sites, min_data)
                sd.update(st)
            if "indPairDist" in analysis:
                base = dm if dm is not None else dist_from_counts(D, C)
                pdd, _ = ind_pair_dists(aln, base)
                for i, j in itertools.combinations_with_replacement(sorted(pdd.keys()), 2):
                    sd["_".join(["d", i, j])] = pdd[i][j]
            if "indHet" in analysis or "hapStats" in analysis:
                # the cached matrix as the reference leaves it at this point (genomics.py:959-963, 940)
                cache = dm.copy() if dm is not None else dist_from_counts(D, C)
                if dm is None and "indPairDist" in analysis:
                    np.fill_diagonal(cache, np.nan)
                if "indHet" in analysis:
                    sd.update(sample_het(aln, cache, C))
                if "hapStats" in analysis:
                    sd.update(h12_stats(aln, cache, hap_dist))
            vals = [_round(sd[s], round_to) for s in stats]
        else:
            good = False
            vals = [np.nan] * len(stats)
        row = ([w.ID] if add_id else []) + [w.scaffold, w.start, w.end, w.mid(), n_sites] + vals
        if good or write_failed:
            lines.append(",".join(_fmt(x) for x in row))
    return "\n".join(lines) + "\n"


def abbababa_windows_csv(geno_path, fmt, pops4, wind_size, step=None, min_sites=1, min_data=0.01, ploidy=None,
                         wind_type="coordinate", overlap=0, max_dist=float("inf"), coords=None,
                         add_id=False, write_failed=False, include=None, exclude=None):
    """ABBABABAwindows.py:27-52 (wrapper) + 244-245 (header).  pops4: [(name,[samples])]*4 = P1,P2,P3,O."""
    with open_text(geno_path) as fh:
        file_names, sites = read_sites(fh)
    if not min_sites:
        min_sites = wind_size
    ind_names = []
    for _, members in pops4:
        for m in members:
            if m not in ind_names:
                ind_names.append(m)
    pop_of = {nm: [p for p, mem in pops4 if nm in mem][0] for nm in ind_names}
    ploidy_of = {nm: (1 if fmt == "haplo" else 2) for nm in ind_names}
    if ploidy == "infer":                                          # --inferPloidy (ABBABABAwindows.py:224, fourPopWindows.py:228, distMat.py:214)
        ploidy_of = {nm: None for nm in ind_names}
        ploidy = None
    names4 = [p[0] for p in pops4]
    lines = [("windowID," if add_id else "") + "scaffold,start,end,mid,sites,sitesUsed,ABBA,BABA,D,fd,fdM"]
    wins = make_windows(sites, wind_type, wind_size, step, overlap, max_dist, min_sites, coords, include, exclude)
    for w in wins:
        n_sites = len(w.positions)
        used = np.nan
        good = False
        vals = [np.nan] * 5
        if n_sites >= min_sites:
            aln = window_to_aln(w, file_names, ind_names, pop_of, ploidy_of, fmt)
            sd = abbababa(aln, names4[0], names4[1], names4[2], names4[3], min_data)
            used = sd["sitesUsed"]
            if used >= min_sites:
                good = True
                vals = [round(np.float64(sd[s]), 4) for s in ("ABBA", "BABA", "D", "fd", "fdM")]
        row = ([w.ID] if add_id else []) + [w.scaffold, w.start, w.end, w.mid(), n_sites, used] + vals
        if good or write_failed:
            lines.append(",".join(_fmt(x) for x in row))
    return "\n".join(lines) + "\n"


def fourpop_windows_csv(geno_path, fmt, pops4, wind_size, step=None, min_sites=1, min_data=0.01, ploidy=None,
                        wind_type="coordinate", overlap=0, max_dist=float("inf"), coords=None,
                        add_id=False, write_failed=False, include=None, exclude=None, polarize=False, fixed=False):
    """fourPopWindows.py:28-55 (wrapper) + 238-243 (header).  pops4: [(name,[samples])]*4 = P1,P2,P3,O."""
    with open_text(geno_path) as fh:
        file_names, sites = read_sites(fh)
    if not min_sites:
        min_sites = wind_size
    ind_names = []
    for _, members in pops4:
        for m in members:
            if m not in ind_names:
                ind_names.append(m)
    pop_of = {nm: [p for p, mem in pops4 if nm in mem][0] for nm in ind_names}
    ploidy_of = {nm: (1 if fmt == "haplo" else 2) for nm in ind_names}
    if ploidy == "infer":                                          # --inferPloidy (ABBABABAwindows.py:224, fourPopWindows.py:228, distMat.py:214)
        ploidy_of = {nm: None for nm in ind_names}
        ploidy = None
    names4 = [p[0] for p in pops4]
    lines = [("windowID," if add_id else "") + "scaffold,start,end,mid,sites,sitesUsed," + ",".join(FOURPOP_STATS)]
    wins = make_windows(sites, wind_type, wind_size, step, overlap, max_dist, min_sites, coords, include, exclude)
    for w in wins:
        n_sites = len(w.positions)
        used = np.nan
        good = False
        vals = [np.nan] * len(FOURPOP_STATS)
        if n_sites >= min_sites:
            aln = window_to_aln(w, file_names, ind_names, pop_of, ploidy_of, fmt)
            sd = four_pop(aln, names4[0], names4[1], names4[2], names4[3], min_data, polarize, fixed)
            used = sd["sitesUsed"]
            if used >= min_sites:
                good = True
                vals = [round(np.float64(sd[s]), 4) for s in FOURPOP_STATS]
        row = ([w.ID] if add_id else []) + [w.scaffold, w.start, w.end, w.mid(), n_sites, used] + vals
        if good or write_failed:
            lines.append(",".join(_fmt(x) for x in row))
    return "\n".join(lines) + "\n"


def distmat_text(geno_path, fmt, wind_size=None, step=None, min_sites=1, wind_type="coordinate",
                 out_format="phylip", round_to=4, include_same=False, min_per_ind=None, samples=None, overlap=0,
                 max_dist=float("inf"), ploidy=None, write_failed=False, coords=None):
    """distMat.py:28-60 (stats_wrapper) + genomics.py:2288-2306 (matrix strings); coordinate, sites, predefined (distMat.py:187 keeps
    three columns of the window list) and cat windows.  A window
    that fails (too few sites, or an individual below --minPerInd) is a matrix of nan (distMat.py:47-50) and is written only
    with --writeFailedWindows (distMat.py:99-100)."""
    with open_text(geno_path) as fh:
        file_names, sites = read_sites(fh)
    ind_names = list(samples) if samples else list(file_names)
    ploidy_of = {nm: (1 if fmt == "haplo" else 2) for nm in ind_names}
    if ploidy == "infer":                                          # --inferPloidy (ABBABABAwindows.py:224, fourPopWindows.py:228, distMat.py:214)
        ploidy_of = {nm: None for nm in ind_names}
        ploidy = None
    if ploidy:
        ploidy_of.update(ploidy)
    pop_of = {}
    if wind_type == "cat":
        wins = [Win(None, None, None, [s[2] for s in sites], [float("nan")] * len(sites), None)]
        min_sites = 1
    else:
        if not min_sites:
            min_sites = wind_size
        if wind_type == "sites":
            wins = sites_windows(sites, wind_size, overlap, max_dist, min_sites)
        elif wind_type == "predefined":
            wins = predefined_windows(sites, [c[:3] for c in coords])
        else:
            wins = coord_windows(sites, wind_size, step or wind_size)
    n = len(ind_names)
    chunks = []
    for w in wins:
        good = len(w.positions) >= min_sites
        if good:
            aln = window_to_aln(w, file_names, ind_names, pop_of, ploidy_of, fmt)
            if min_per_ind and min(aln.mask.sum(axis=1)) < min_per_ind:
                good = False
        if not good and not write_failed:
            continue
        if good:
            D, C = pair_counts_gemm(aln)
            pdd, _ = ind_pair_dists(aln, dist_from_counts(D, C), include_same)
            M = np.zeros((n, n))
            for i, j in itertools.combinations_with_replacement(range(n), 2):
                M[i, j] = M[j, i] = pdd[ind_names[i]][ind_names[j]]
        else:
            M = np.full((n, n), np.nan)
        txt = M.round(round_to).astype(str)
        if out_format == "raw":
            chunks.append("\n".join(" ".join(r) for r in txt) + "\n")
        elif out_format == "phylip":
            chunks.append(str(n) + "\n" + "".join(ind_names[i] + "  " + " ".join(txt[i]) + "\n" for i in range(n)))
        else:
            s = "\nBEGIN Taxa;\nDIMENSIONS ntax=%d;\nTAXLABELS\n" % n
            s += "".join("[%d] '%s'\n" % (i + 1, ind_names[i]) for i in range(n))
            s += ";\nEND; [Taxa]\n"
            s += "\nBEGIN Distances;\nDIMENSIONS ntax=%d;\nFORMAT labels=left diagonal triangle=both;\nMATRIX\n" % n
            s += "".join("[%d] '%s'    " % (i + 1, ind_names[i]) + " ".join(txt[i]) + "\n" for i in range(n))
            s += ";\nEND; [Distances]\n"
            chunks.append(s)
    return "".join(chunks)


# ----------------------------------------------------------------------------------------------
# helpers shared by tests: engine one-hot codes -> reference-order alignment
# ----------------------------------------------------------------------------------------------
def aln_from_codes(codes_sites_by_hap, hap_names, sample_names, groups):
    """Build an Aln from engine one-hot int8 codes [L][H] (A=1,C=2,G=4,T=8,0=missing) given per-haplotype
    names; rows are sorted by haplotype name as genoToAlignment does (genomics.py:1122)."""
    lut = np.full(256, -999, dtype=np.int64)
    lut[1], lut[2], lut[4], lut[8] = 0, 1, 2, 3
    num = lut[np.asarray(codes_sites_by_hap).astype(np.uint8)].T
    order = np.argsort(hap_names)
    return Aln(np.ascontiguousarray(num[order]), [hap_names[o] for o in order],
               [sample_names[o] for o in order], [groups[o] for o in order]), order


# ----------------------------------------------------------------------------------------------
# freq.py (SURVEY.md 8f "next" row 1): per-site per-population base counts / target-allele frequencies
# ----------------------------------------------------------------------------------------------
def freq_tsv(geno_path, fmt, pops, target=None, as_counts=False, min_data=0, threshold=None, keep_nan=False,
             ploidy=None):
    """freq.py:30-113 (freqs_wrapper) + 222-224 (asCounts / keepNanLines / minData are overridden without --target) +
    297-298 (header).  pops: [(name, [samples])] in CLI order; `minor` target is not restated (the reference breaks ties with
    np.random.choice, freq.py/genomics.py:664-669)."""
    with open_text(geno_path) as fh:
        file_names, sites = read_sites(fh)
    if not target:
        as_counts, keep_nan, min_data = True, True, 0
    ind_names = []
    for _, members in pops:
        for m in members:
            if m not in ind_names:
                ind_names.append(m)
    pop_of = {nm: [p for p, mem in pops if nm in mem][0] for nm in ind_names}
    ploidy_of = {nm: 2 for nm in ind_names}
    if ploidy:
        ploidy_of.update(ploidy)
    win = Win(None, None, None, [s[2] for s in sites], [s[1] for s in sites], None)
    aln = window_to_aln(win, file_names, ind_names, pop_of, ploidy_of, "pairs" if fmt == "alleles" else fmt)
    g = np.array(aln.groups, dtype=object)
    pop_names = [p[0] for p in pops]
    counts = {}
    for p in pop_names:
        counts[p] = site_pop_counts(aln, np.where(g == p)[0])
    L = aln.L
    lines = ["scaffold\tposition\t" + "\t".join(pop_names)]
    if not target:
        cols = [[",".join(r) for r in counts[p][0].astype(str)] for p in pop_names]
        for i in range(L):
            lines.append("\t".join([sites[i][0], str(sites[i][1])] + [c[i] for c in cols]))
        return "\n".join(lines) + "\n"
    assert target == "derived"
    out_cnt = counts[pop_names[-1]][0]
    in_cnt = sum(counts[p][0] for p in pop_names[:-1])
    base = np.full(L, np.nan)
    for i in range(L):                                            # derivedAllele, genomics.py:636-662
        ina = np.where(in_cnt[i] > 0)[0]
        outa = np.where(out_cnt[i] > 0)[0]
        if len(outa) == 1 and len(ina) == 2 and outa[0] in ina:
            base[i] = ina[ina != outa[0]][0]
    good_site = ~np.isnan(base)
    cols = []
    for p in pop_names:
        cnt, n = counts[p]
        good = good_site & (n >= min_data)                        # freq.py:80 compares the COUNT with the proportion
        tf = np.zeros(L, dtype=int) if as_counts else np.full(L, np.nan)
        idx = np.where(good)[0]
        if len(idx):
            b = base[idx].astype(int)
            if as_counts:
                tf[idx] = cnt[idx, b]
            else:
                with np.errstate(divide="ignore", invalid="ignore"):
                    tf[idx] = 1. * cnt[idx, b] / n[idx]
        cols.append(np.around(tf, 4))
    allf = np.column_stack(cols)
    if threshold and not as_counts:
        hi_ = allf >= threshold
        lo_ = allf < threshold
        allf[hi_] = 1
        allf[lo_] = 0
    if keep_nan:
        keep = np.arange(L)
    elif not as_counts:
        keep = np.where(~np.all(np.isnan(allf), axis=1))[0]
    else:
        keep = np.where(~np.all(allf == 0, axis=1))[0]
    txt = allf.astype(str)
    for i in keep:
        lines.append("\t".join([sites[i][0], str(sites[i][1])] + list(txt[i])))
    return "\n".join(lines) + "\n"
