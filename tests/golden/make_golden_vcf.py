#!/usr/bin/env python
"""Golden outputs of the UNMODIFIED reference VCF_processing/parseVCF.py on two seeded synthetic VCF files.

    python tests/golden/make_golden_vcf.py        (needs /root/reference; writes tests/golden/vcf/*)

main.vcf.gz: diploid genotypes, phased and unphased, missing and half-missing calls, allele indices beyond the ALT list, MONO /
SNP / multi-allelic / indel / '*' sites, QUAL floats and '.', duplicated positions, FORMAT orders GT:DP:GQ, GT:GQ:DP:AD and GT.
hap.vcf.gz: one haploid sample (--ploidyFile) and a few genotypes of the wrong ploidy (--ploidyMismatchToMissing)."""
import gzip
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "vcf")
REF = "/root/reference/VCF_processing/parseVCF.py"

VCF_CASES = [
    ("main_skipindels", "main", ["--skipIndels"]),
    ("main_filters", "main", ["--skipIndels", "--minQual", "30", "--gtf", "flag=DP", "min=5", "--gtf", "flag=GQ", "min=20", "gtTypes=Het"]),
    ("main_subset_dups", "main", ["--skipIndels", "--excludeDuplicates", "--include", "chr1,chr3", "-s", "s3,s1,s5"]),
    ("main_partial_reftrack", "main", ["--skipIndels", "--keepPartial", "--exclude", "chr2", "--maxREFlen", "1", "--noHeader",
                                       "--outSep", " ", "--addRefTrack"]),
    ("main_missing_sitetypes", "main", ["--skipIndels", "--missing", "X", "--gtf", "flag=DP", "max=30", "siteTypes=SNP", "samples=s0,s2"]),
    ("main_ad_list", "main", ["--skipIndels", "--gtf", "flag=AD", "min=1", "gtTypes=Het,HomAlt"]),
    ("snps_default", "snps", []),
    ("hap_ploidyfile", "hap", ["--skipIndels", "--ploidyFile", "{dir}/hap.ploidy", "--ploidyMismatchToMissing"]),
    # cells of varying width: a --missing / --outSep of several characters
    ("main_missing_na", "main", ["--skipIndels", "--missing", "NA", "--gtf", "flag=DP", "min=5"]),
    ("hap_outsep_two_partial", "hap", ["--outSep", "::", "--missing", "??", "--keepPartial", "--ploidyFile", "{dir}/hap.ploidy",
                                       "--ploidyMismatchToMissing", "--addRefTrack"]),
    ("main_field_dp", "main", ["--field", "DP"]),
    ("main_field_gq_subset", "main", ["--field", "GQ", "--missing", "NA", "-s", "s3,s1", "--excludeDuplicates", "--minQual", "30",
                                      "--addRefTrack", "--exclude", "chr2", "--maxREFlen", "1"]),
    ("main_field_ad_noheader", "main", ["--field", "AD", "--noHeader", "--outSep", ",", "--include", "chr1,chr3"]),
    # freebayes-style records (multi-base REF / ALT haplotypes with a CIGAR per ALT in INFO): --simplifyALT rewrites every ALT to the
    # length of REF, --expandMulti writes one row per base (parseVCF.py:25-50, 74-77, 160-166, 380-388)
    ("cigar_simplify", "cigar", ["--simplifyALT"]),
    ("cigar_simplify_skipindels", "cigar", ["--simplifyALT", "--skipIndels", "--minQual", "20"]),
    ("cigar_expand", "cigar", ["--expandMulti"]),
    ("cigar_expand_skipindels_reftrack", "cigar", ["--expandMulti", "--skipIndels", "--addRefTrack", "--excludeDuplicates"]),
    ("cigar_expand_filters_subset", "cigar", ["--expandMulti", "--skipIndels", "--keepPartial", "--gtf", "flag=DP", "min=8", "--gtf", "flag=GQ",
                                              "min=30", "siteTypes=SNP", "gtTypes=Het", "-s", "s3,s1,s4", "--exclude", "chr2", "--outSep", " "]),
    ("cigar_simplify_hap", "cigarhap", ["--simplifyALT", "--ploidyFile", "{dir}/hap.ploidy", "--ploidyMismatchToMissing", "--maxREFlen", "3"]),
    # --field beside --simplifyALT / --expandMulti: the expansion loop (parseVCF.py:380-385) runs over the VALUES, one character a row
    ("cigar_field_gq_expand", "cigar", ["--field", "GQ", "--expandMulti", "--maxREFlen", "1", "--addRefTrack"]),
    ("cigar_field_phase_simplify", "cigar", ["--field", "phase", "--simplifyALT", "-s", "s3,s1"]),
    ("cigar_field_alleles_expand", "cigar", ["--field", "alleles", "--expandMulti", "--maxREFlen", "2", "--outSep", ","]),
]


def make_vcf(path, seed, n_samples=6, hap_sample=None, snps_only=False, wrong_ploidy=0.0):
    rng = np.random.default_rng(seed)
    names = ["s%d" % k for k in range(n_samples)]
    lines = ["##fileformat=VCFv4.2", "##contig=<ID=chr1,length=100000>", "##contig=<ID=chr2,length=50000>",
             "##source=make_golden_vcf", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names)]
    bases = "ACGT"
    for chrom, n in (("chr1", 220), ("chr2", 120), ("chr3", 90)):
        pos = 0
        for _ in range(n):
            pos += int(rng.integers(0 if rng.random() < 0.06 else 1, 40))        # a few duplicated positions
            pos = max(pos, 1)
            ref = bases[rng.integers(0, 4)]
            kind = rng.random()
            if kind < 0.15:
                alt = "."
            elif kind < 0.70 or snps_only:
                alt = bases[(bases.index(ref) + int(rng.integers(1, 4))) % 4]
            elif kind < 0.80:
                a1 = bases[(bases.index(ref) + 1) % 4]
                a2 = bases[(bases.index(ref) + 2) % 4]
                alt = a1 + "," + a2
            elif kind < 0.88:
                alt = ref + bases[rng.integers(0, 4)] + "," + bases[(bases.index(ref) + 1) % 4]     # insertion + SNP
            elif kind < 0.94:
                ref = ref + bases[rng.integers(0, 4)]                                                   # deletion
                alt = ref[0]
            else:
                alt = bases[(bases.index(ref) + 1) % 4] + ",*"
            n_alt = 0 if alt == "." else len(alt.split(","))
            qual = "." if rng.random() < 0.1 else ("%.1f" % (rng.random() * 100) if rng.random() < 0.7 else str(int(rng.integers(1, 99))))
            fk = rng.random()
            fmt = "GT:DP:GQ" if fk < 0.6 else ("GT:GQ:DP:AD" if fk < 0.9 else "GT")
            cells = []
            for s in range(n_samples):
                ploidy = 1 if s == hap_sample else 2
                if rng.random() < wrong_ploidy:
                    ploidy = 3 - ploidy
                al = []
                for _a in range(ploidy):
                    r = rng.random()
                    al.append("." if r < 0.08 else str(int(rng.integers(0, n_alt + 1)) if r < 0.97 else n_alt + 1))
                if rng.random() < 0.05:
                    al = ["."] * ploidy
                gt = ("|" if rng.random() < 0.4 else "/").join(al)
                dp = "." if rng.random() < 0.05 else str(int(rng.integers(0, 40)))
                gq = "." if rng.random() < 0.05 else str(int(rng.integers(0, 99)))
                ad = ",".join(str(int(rng.integers(0, 12))) for _ in range(n_alt + 1))
                cells.append({"GT:DP:GQ": "%s:%s:%s" % (gt, dp, gq), "GT:GQ:DP:AD": "%s:%s:%s:%s" % (gt, gq, dp, ad), "GT": gt}[fmt])
            lines.append("\t".join([chrom, str(pos), ".", ref, alt, qual, "PASS", "NS=%d" % n_samples, fmt] + cells))
        lines.append("")                                                                                 # an empty line
    with gzip.open(path, "wt") as f:
        f.write("\n".join(lines) + "\n")


def make_cigar_vcf(path, seed, n_samples=6, hap_sample=None, wrong_ploidy=0.0):
    """freebayes-style: SNPs (CIGAR 1X), MNPs (e.g. 1M1X1M, 3X), complex records whose ALT haplotypes carry insertions and deletions
    relative to REF (1M2D1M, 1M2I1M, 1X1M1D), several ALTs per record (one CIGAR each); every CIGAR is consistent with its REF / ALT
    pair (M + X + D = len(REF), M + X + I = len(ALT))"""
    rng = np.random.default_rng(seed)
    names = ["s%d" % k for k in range(n_samples)]
    lines = ["##fileformat=VCFv4.2", "##source=freeBayes-like", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names)]
    bases = "ACGT"

    def other(b):
        return bases[(bases.index(b) + int(rng.integers(1, 4))) % 4]

    def make_alt(ref):
        """an ALT haplotype and its CIGAR against ref"""
        alt, cig, i = "", [], 0
        while i < len(ref):
            r = rng.random()
            if r < 0.45:
                op, piece = "M", ref[i]
                i += 1
            elif r < 0.8:
                op, piece = "X", other(ref[i])
                i += 1
            elif r < 0.9 and 0 < i:
                op, piece = "I", bases[rng.integers(0, 4)]
            else:
                op, piece = "D", ""
                i += 1
            alt += piece
            if cig and cig[-1][1] == op:
                cig[-1][0] += 1
            else:
                cig.append([1, op])
        if not alt or all(op == "M" for _, op in cig):
            return make_alt(ref)
        return alt, "".join("%d%s" % (n, op) for n, op in cig)

    for chrom, n in (("chr1", 160), ("chr2", 60), ("chr3", 50)):
        pos = 0
        for _ in range(n):
            pos += int(rng.integers(0 if rng.random() < 0.05 else 4, 40))
            pos = max(pos, 1)
            kind = rng.random()
            reflen = 1 if kind < 0.5 else int(rng.integers(2, 5))
            ref = "".join(bases[rng.integers(0, 4)] for _ in range(reflen))
            if kind < 0.08:
                alts, cigs = [], []
            else:
                pairs = [make_alt(ref) for _ in range(1 if rng.random() < 0.75 else 2)]
                alts, cigs = [p[0] for p in pairs], [p[1] for p in pairs]
            n_alt = len(alts)
            qual = "." if rng.random() < 0.1 else "%.2f" % (rng.random() * 100)
            info = "NS=%d;CIGAR=%s;TYPE=%s" % (n_samples, ",".join(cigs) if cigs else "1M", "snp" if reflen == 1 else "complex")
            fmt = "GT:DP:GQ" if rng.random() < 0.8 else "GT:GQ:DP"
            cells = []
            for s_ in range(n_samples):
                ploidy = 1 if s_ == hap_sample else 2
                if rng.random() < wrong_ploidy:
                    ploidy = 3 - ploidy
                al = ["." if rng.random() < 0.07 else str(int(rng.integers(0, n_alt + 1))) for _a in range(ploidy)]
                if rng.random() < 0.04:
                    al = ["."] * ploidy
                gt = ("|" if rng.random() < 0.3 else "/").join(al)
                dp, gq = str(int(rng.integers(0, 40))), str(int(rng.integers(0, 99)))
                cells.append("%s:%s:%s" % ((gt, dp, gq) if fmt == "GT:DP:GQ" else (gt, gq, dp)))
            lines.append("\t".join([chrom, str(pos), ".", ref, ",".join(alts) if alts else ".", qual, ".", info, fmt] + cells))
    with gzip.open(path, "wt") as f:
        f.write("\n".join(lines) + "\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    make_cigar_vcf(os.path.join(OUT, "cigar.vcf.gz"), 21)
    make_cigar_vcf(os.path.join(OUT, "cigarhap.vcf.gz"), 22, hap_sample=2, wrong_ploidy=0.03)
    make_vcf(os.path.join(OUT, "main.vcf.gz"), 11)
    make_vcf(os.path.join(OUT, "snps.vcf.gz"), 12, snps_only=True)
    make_vcf(os.path.join(OUT, "hap.vcf.gz"), 13, hap_sample=2, wrong_ploidy=0.03)
    with open(os.path.join(OUT, "hap.ploidy"), "wt") as f:
        f.write("s2 1\ns4 2\n")
    only = set(sys.argv[1:])
    for name, vcf, argv in VCF_CASES:
        if only and name not in only:
            continue
        cmd = [sys.executable, REF, "-i", os.path.join(OUT, vcf + ".vcf.gz"), "-o", os.path.join(OUT, name + ".geno")]
        cmd += [a.format(dir=OUT) for a in argv]
        subprocess.run(cmd, check=True, timeout=300, stderr=subprocess.DEVNULL)
        print(name, os.path.getsize(os.path.join(OUT, name + ".geno")))


if __name__ == "__main__":
    main()
