"""Counter-based synthetic genotype generator (host/NumPy statement of the spec).

The reference ships no data and no generator; SURVEY.md section 8(d) asks for a
deterministic, counter-based one so that any window of a huge device-resident data
set can be re-materialised as `.geno` text for the CPU oracle.  Everything here is
integer arithmetic on 64-bit hashes (no libm), so the HIP kernel `pg_synth_fill`
(csrc/popgen_kernels.hip) reproduces it bit for bit; tests/test_gpu_synth.py checks that.

Model (per site, keyed by (seed, scaffold id, position)):
  ref base, 10 % variable sites, alt base, 1 % of variable sites carry a third allele,
  base alt frequency p ~ U(0,1), per-population p_k = clip(p + 0.15 z_k) with z_k an
  Irwin-Hall(4) approximation of N(0,1), last population fixed ancestral w.p. 0.8,
  each haplotype allele ~ Bernoulli(p_k), each diploid genotype missing w.p. 5 %.

Allele codes are the engine's one-hot int8 codes: A=1 C=2 G=4 T=8, missing=0.
"""
import numpy as np

U64 = np.uint64
SEED_DEFAULT = 20260925

K_GOLD = U64(0x9E3779B97F4A7C15)
K_M1 = U64(0xBF58476D1CE4E5B9)
K_M2 = U64(0x94D049BB133111EB)
K_SCAF = U64(0xD6E8FEB86659FD93)
K_POP = U64(0xA24BAED4963EE407)
K_HAP = U64(0x9FB21C651E98DF25)
K_DIP = U64(0xC2B2AE3D27D4EB4F)

VAR_THR = 6554      # 0.10 * 65536
THIRD_THR = 655     # 0.01 * 65536
MISS_THR = 3277     # 0.05 * 65536
OUTFIX_THR = 52429  # 0.80 * 65536
USE3_THR = 26       # ~0.10 * 256
Z_SCALE = 17027     # 0.15 * 65536 / sd(IrwinHall4 of u16) * 65536

BASES = "ACGT"
ONEHOT = np.array([1, 2, 4, 8], dtype=np.int8)


def mix64(x):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=U64) + K_GOLD)
        z = (z ^ (z >> U64(30))) * K_M1
        z = (z ^ (z >> U64(27))) * K_M2
        return z ^ (z >> U64(31))


def site_keys(seed, scaf_id, pos):
    with np.errstate(over="ignore"):
        s = mix64(U64(seed) ^ (np.asarray(scaf_id, dtype=U64) * K_SCAF))
        return mix64(s ^ np.asarray(pos, dtype=U64))


def gen_codes(seed, scaf_id, pos, n_dip, n_pops, hap_index=None,
              var_thr=VAR_THR, miss_thr=MISS_THR):
    """One-hot int8 allele codes [n_sites][n_hap] for the sites (scaf_id[i], pos[i]).

    Haplotype g (generator order) is allele g%2 of diploid g//2; diploids are dealt to
    populations in contiguous equal blocks (diploid d -> pop d*n_pops//n_dip).
    `hap_index` optionally selects/reorders generator haplotypes (device slot -> g).
    """
    pos = np.asarray(pos, dtype=np.int64)
    scaf_id = np.broadcast_to(np.asarray(scaf_id, dtype=np.int64), pos.shape)
    n_hap_all = 2 * n_dip
    if hap_index is None:
        hap_index = np.arange(n_hap_all, dtype=np.int64)
    hap_index = np.asarray(hap_index, dtype=np.int64)
    ks = site_keys(seed, scaf_id, pos)                       # [L]
    ref = (ks & U64(3)).astype(np.int64)
    variable = ((ks >> U64(2)) & U64(0xFFFF)).astype(np.int64) < var_thr
    altoff = ((ks >> U64(18)) & U64(0xFF)).astype(np.int64) % 3
    alt = (ref + 1 + altoff) & 3
    has3 = ((ks >> U64(26)) & U64(0xFFFF)).astype(np.int64) < THIRD_THR
    third = (ref + 1 + (altoff + 1) % 3) & 3
    p16 = ((ks >> U64(42)) & U64(0xFFFF)).astype(np.int64)

    with np.errstate(over="ignore"):
        pk = np.empty((len(pos), n_pops), dtype=np.int64)
        for k in range(n_pops):
            kp = mix64(ks ^ (K_POP * U64(k + 1)))
            z = ((kp & U64(0xFFFF)).astype(np.int64) + ((kp >> U64(16)) & U64(0xFFFF)).astype(np.int64)
                 + ((kp >> U64(32)) & U64(0xFFFF)).astype(np.int64)
                 + ((kp >> U64(48)) & U64(0xFFFF)).astype(np.int64) - 131070)
            delta = (z * Z_SCALE) >> 16          # arithmetic shift == floor division
            v = np.clip(p16 + delta, 0, 65535)
            if k == n_pops - 1 and n_pops > 1:
                fixed = (mix64(kp) & U64(0xFFFF)).astype(np.int64) < OUTFIX_THR
                v = np.where(fixed, 0, v)
            pk[:, k] = v

        g = hap_index[None, :]                                # [1][H]
        dip = g // 2
        pop_of = (dip * n_pops) // n_dip
        kh = mix64(ks[:, None] ^ (K_HAP * (g + 1).astype(U64)))
        derived = (kh & U64(0xFFFF)).astype(np.int64) < np.take_along_axis(
            pk, np.broadcast_to(pop_of, (len(pos), g.shape[1])), axis=1)
        use3 = has3[:, None] & (((kh >> U64(16)) & U64(0xFF)).astype(np.int64) < USE3_THR)
        kd = mix64(ks[:, None] ^ (K_DIP * (dip + 1).astype(U64)))
        missing = ((kd >> U64(8)) & U64(0xFFFF)).astype(np.int64) < miss_thr

    allele = np.where(derived, np.where(use3, third[:, None], alt[:, None]), ref[:, None])
    allele = np.where(variable[:, None], allele, ref[:, None])
    codes = ONEHOT[allele]
    codes = np.where(missing, np.int8(0), codes).astype(np.int8)
    return codes


def dense_sites(n_sites, n_scaf):
    """Scaffold ids / positions of the dense layout used by the benchmarks: `n_scaf` equal
    scaffolds, every position 1..len present."""
    per = n_sites // n_scaf
    idx = np.arange(per * n_scaf, dtype=np.int64)
    return idx // per, idx % per + 1


class DenseRows:
    """The dense benchmark data set (n_scaf equal scaffolds, every position present) as a row-indexed input the multi-GPU plan can
    cut (genomics_general_amd.shardplan, the `.pgeno` interface: block headers + position arrays), without a file: bench.py
    --strong lets the ranks plan their window ranges on it exactly as the drivers do on a `.pgeno` file, and every rank then
    generates only its own rows on its device (the generator is counter-based)."""
    packed = True

    def __init__(self, n_sites, scaf_len, scaf_names):
        self.total, self.scaf_len, self.names = int(n_sites), int(scaf_len), list(scaf_names)
        self._rows = (0, self.total)

    def _index(self):
        return [(None, k * self.scaf_len, self.scaf_len, [0], [nm]) for k, nm in enumerate(self.names)], self.total

    def positions(self, blocks, a, b):
        return (np.arange(a, b, dtype=np.int64) % self.scaf_len + 1).astype(np.int32)

    def restrict_rows(self, blocks, a, b):
        self._rows = (int(a), int(b))

    def local_runs(self):
        """(run starts, run names, positions) of the rows this reader is restricted to, row 0 = its first row"""
        a, b = self._rows
        if b <= a:
            return np.zeros(0, dtype=np.int64), [], np.zeros(0, dtype=np.int32)
        k0, k1 = a // self.scaf_len, (b - 1) // self.scaf_len
        starts = np.array([max(k * self.scaf_len, a) - a for k in range(k0, k1 + 1)], dtype=np.int64)
        return starts, [self.names[k] for k in range(k0, k1 + 1)], self.positions(None, a, b)


def codes_to_letters(codes):
    lut = np.full(256, ord("N"), dtype=np.uint8)
    for b, c in zip(BASES, (1, 2, 4, 8)):
        lut[c] = ord(b)
    return lut[codes.astype(np.uint8)]


def write_geno(path_or_file, scaf_names, scaf_id, pos, codes, sample_names, sep="/", fmt="phased", haploid=()):
    """Render generator output as `.geno` text (README.md:32-40 of the reference).
    codes: [L][2*n_samples] in generator order; fmt in phased|pairs|diplo|haplo.  Samples whose index is in `haploid` get
    one-character cells (their first allele) in the phased / pairs formats (mixed ploidy, e.g. a sex chromosome)."""
    import gzip
    letters = codes_to_letters(codes)
    L, H = letters.shape
    close = False
    if isinstance(path_or_file, str):
        f = gzip.open(path_or_file, "wt") if path_or_file.endswith(".gz") else open(path_or_file, "wt")
        close = True
    else:
        f = path_or_file
    f.write("#CHROM\tPOS\t" + "\t".join(sample_names) + "\n")
    iupac = {"AA": "A", "CC": "C", "GG": "G", "TT": "T", "GT": "K", "TG": "K", "AC": "M", "CA": "M",
             "CG": "S", "GC": "S", "AG": "R", "GA": "R", "AT": "W", "TA": "W", "CT": "Y", "TC": "Y", "NN": "N"}
    for i in range(L):
        row = letters[i].tobytes().decode()
        if fmt == "haplo":
            cells = list(row)
        else:
            pairs = [row[j:j + 2] for j in range(0, H, 2)]
            if fmt == "phased":
                cells = [p[0] if k in haploid else p[0] + sep + p[1] for k, p in enumerate(pairs)]
            elif fmt == "pairs":
                cells = [p[0] if k in haploid else p for k, p in enumerate(pairs)]
            else:
                cells = [iupac[p] for p in pairs]
        f.write(scaf_names[int(scaf_id[i])] + "\t" + str(int(pos[i])) + "\t" + "\t".join(cells) + "\n")
    if close:
        f.close()


def write_geno_fast(path, codes, sample_names, scaf="chr1", pos0=1):
    """Vectorised `.geno` writer for benchmarks: phased cells `A/C` of fixed width, one scaffold, positions pos0, pos0+1, ...
    codes: one-hot int8 [L][2 * n_samples] in sample order (columns 2d, 2d+1 = the alleles of sample d)."""
    letters = codes_to_letters(codes)
    L, H = letters.shape
    n = H // 2
    head = (scaf + "\t").encode()
    with open(path, "wb") as f:
        f.write(("#CHROM\tPOS\t" + "\t".join(sample_names) + "\n").encode())
        step = 100000
        a = 0
        while a < L:
            # rows whose position has the same number of digits are one fixed-width byte matrix: prefix | digits | tab | cells
            p0 = pos0 + a
            nd = len(str(p0))
            b = min(L, a + step, 10 ** nd - pos0)
            rows = b - a
            line = np.empty((rows, len(head) + nd + 1 + 4 * n), dtype=np.uint8)
            line[:, :len(head)] = np.frombuffer(head, dtype=np.uint8)
            pos = np.arange(p0, p0 + rows, dtype=np.int64)
            for k in range(nd):
                line[:, len(head) + nd - 1 - k] = (pos // 10 ** k % 10 + ord("0")).astype(np.uint8)
            line[:, len(head) + nd] = ord("\t")
            cell = line[:, len(head) + nd + 1:].reshape(rows, n, 4)
            cell[:, :, 0] = letters[a:b, 0::2]
            cell[:, :, 1] = ord("/")
            cell[:, :, 2] = letters[a:b, 1::2]
            cell[:, :, 3] = ord("\t")
            cell[:, -1, 3] = ord("\n")
            f.write(line.data)
            a = b
