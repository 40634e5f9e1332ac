"""Golden-case table shared by make_golden.py (runs the UNMODIFIED reference) and the tests.

fixture: name -> generator parameters (genomics_general_amd.synth).  case: tool, argv (reference CLI
syntax, with {geno} / {dir} placeholders), output file name."""

FIXTURES = {
    # BASELINE.json configs[0]: 10k sites (2 scaffolds x 5000), 8 diploids
    "c1": dict(seed=20260925, n_dip=8, n_pops=2, scaf_len=[5000, 5000], density=1.0, var_thr=6554, miss_thr=3277, fmt="phased", sep="/"),
    # sparse positions -> empty windows, gaps; 12 diploids / 3 pops; '|' separator
    "sparse": dict(seed=20260926, n_dip=12, n_pops=3, scaf_len=[6000, 3000, 2500], density=0.25, var_thr=30000, miss_thr=6000, fmt="phased", sep="|", gap=(2000, 3700)),
    # heavy missingness -> pairs below minSites, all-nan blocks
    "holes": dict(seed=20260927, n_dip=6, n_pops=2, scaf_len=[3000], density=0.5, var_thr=40000, miss_thr=36000, fmt="phased", sep="/"),
    # four populations for ABBA-BABA
    "abba": dict(seed=20260928, n_dip=16, n_pops=4, scaf_len=[8000, 4000], density=0.6, var_thr=45000, miss_thr=5000, fmt="phased", sep="/"),
    # same data as 'abba' rendered in the other genotype formats
    "abba_pairs": dict(seed=20260928, n_dip=16, n_pops=4, scaf_len=[8000, 4000], density=0.6, var_thr=45000, miss_thr=5000, fmt="pairs", sep=""),
    "abba_diplo": dict(seed=20260928, n_dip=16, n_pops=4, scaf_len=[8000, 4000], density=0.6, var_thr=45000, miss_thr=5000, fmt="diplo", sep=""),
    # haploid cells
    # mixed ploidy: samples 1, 6 and 9 have one-character cells (--haploid / --ploidyFile)
    "mixed": dict(seed=20260930, n_dip=10, n_pops=2, scaf_len=[3000, 1500], density=0.7, var_thr=30000, miss_thr=5000, fmt="phased", sep="/",
                  haploid=(1, 6, 9)),
    # many sites with three and four alleles (multi_frac of the sites get alleles drawn uniformly from A,C,G,T): the pairwise
    # difference counts of multi-allelic sites (k_pairD's virtual biallelic sites) against the reference's numHamming
    "multi": dict(seed=20261001, n_dip=9, n_pops=3, scaf_len=[2500, 1200], density=0.8, var_thr=30000, miss_thr=6000, fmt="phased", sep="/",
                  multi_frac=0.5),
    # ONE scaffold (with a gap: empty windows) and FOUR equal scaffolds: what the window-range shards of a multi-GPU launch are
    # tested on (2, 3 and 8 ranks; cuts between scaffold runs alone would leave ranks without data)
    "one": dict(seed=20261002, n_dip=8, n_pops=2, scaf_len=[12000], density=0.4, var_thr=30000, miss_thr=5000, fmt="phased", sep="/",
                gap=(5000, 7100)),
    "four": dict(seed=20261003, n_dip=8, n_pops=4, scaf_len=[3000, 3000, 3000, 3000], density=0.5, var_thr=40000, miss_thr=5000,
                 fmt="phased", sep="/"),
    "haplo": dict(seed=20260929, n_dip=5, n_pops=2, scaf_len=[4000], density=0.5, var_thr=30000, miss_thr=4000, fmt="haplo", sep=""),
    # positions beyond 32 bits (the reference parses Python integers, genomics.py:1884-1904; chromosomes of more than 2^31 bases
    # exist): the first scaffold's positions straddle 2^31, the second starts at 3 * 10^9
    "bigpos": dict(seed=20261004, n_dip=8, n_pops=4, scaf_len=[4000, 2000], density=0.5, var_thr=30000, miss_thr=5000, fmt="phased", sep="/",
                   pos_offset=[2 ** 31 - 2000, 3000000000]),
    # ploidy that changes along the file (--inferPloidy: per window and sample, genomics.py:1108-1111): four populations of
    # three.  haploid_where = (samples, scaffold index, first position, last position): their cells are one character there.
    # chr1: s2 and s8 from position 1001 on (a window boundary of -w 1000); chr2: the "males" s1, s4, s7, s10 everywhere (a sex
    # chromosome behind an autosome), s5 for positions 1500 .. 1800 (inside a window)
    "ploidyshift": dict(seed=20261005, n_dip=12, n_pops=4, scaf_len=[2400, 2600], density=0.8, var_thr=40000, miss_thr=4000, fmt="phased",
                        sep="/", haploid_where=[((2, 8), 0, 1001, 10 ** 9), ((1, 4, 7, 10), 1, 1, 10 ** 9), ((5,), 1, 1500, 1800)]),
    "ploidyshift_pairs": dict(seed=20261005, n_dip=12, n_pops=4, scaf_len=[2400, 2600], density=0.8, var_thr=40000, miss_thr=4000, fmt="pairs",
                              sep="", haploid_where=[((2, 8), 0, 1001, 10 ** 9), ((1, 4, 7, 10), 1, 1, 10 ** 9), ((5,), 1, 1500, 1800)]),
}


def pops_args(n_dip, n_pops, flag="-p"):
    per = n_dip // n_pops
    out = []
    for k in range(n_pops):
        out += [flag, "pop%d" % k, ",".join("s%d" % d for d in range(k * per, (k + 1) * per))]
    return out


def abba_args(n_dip):
    per = n_dip // 4
    out = []
    for flag, k in (("-P1", 0), ("-P2", 1), ("-P3", 2), ("-O", 3)):
        out += [flag, "pop%d" % k, ",".join("s%d" % d for d in range(k * per, (k + 1) * per))]
    return out


CASES = [
    # ---- popgenWindows.py ----
    dict(name="c1_popgen", tool="popgenWindows.py", fixture="c1",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "100"] + pops_args(8, 2)),
    dict(name="c1_popgen_T2_round12", tool="popgenWindows.py", fixture="c1",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "100", "--roundTo", "12", "-T", "2"] + pops_args(8, 2)),
    dict(name="c1_popgen_allpop", tool="popgenWindows.py", fixture="c1",
         argv=["-g", "{geno}", "-f", "phased", "-w", "2500", "-m", "50"]),
    # one population and only the pair statistics asked for: no statistic column at all (the header ends in a comma, the rows do not)
    dict(name="c1_popgen_pairdist_one_pop", tool="popgenWindows.py", fixture="c1",
         argv=["-g", "{geno}", "-f", "phased", "-w", "2500", "-m", "50", "--analysis", "popPairDist", "--addWindowID"]),
    dict(name="sparse_overlap_failed_id", tool="popgenWindows.py", fixture="sparse",
         argv=["-g", "{geno}", "-f", "phased", "-w", "500", "-s", "250", "-m", "10", "--writeFailedWindows",
               "--addWindowID", "--roundTo", "6"] + pops_args(12, 3)),
    dict(name="sparse_stepgap", tool="popgenWindows.py", fixture="sparse",
         argv=["-g", "{geno}", "-f", "phased", "-w", "300", "-s", "700", "-m", "5", "--writeFailedWindows"] + pops_args(12, 3)),
    dict(name="sparse_popfreq_indpair", tool="popgenWindows.py", fixture="sparse",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "20", "--analysis", "popFreq", "popDist", "popPairDist",
               "indPairDist", "--roundTo", "8"] + pops_args(12, 3)),
    dict(name="sparse_indpair_only", tool="popgenWindows.py", fixture="sparse",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1500", "-m", "20", "--analysis", "indPairDist", "--roundTo", "8",
               "--samples", "s0,s1,s5,s9"]),
    dict(name="sparse_sites_windows", tool="popgenWindows.py", fixture="sparse",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "sites", "-w", "200", "-O", "50", "-m", "100", "--roundTo", "8"] + pops_args(12, 3)),
    dict(name="sparse_sites_maxdist", tool="popgenWindows.py", fixture="sparse",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "sites", "-w", "100", "-D", "300", "-m", "40", "--roundTo", "8"] + pops_args(12, 3)),
    dict(name="sparse_predefined", tool="popgenWindows.py", fixture="sparse",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "predefined", "--windCoords", "{dir}/sparse_coords.txt", "-m", "5",
               "--writeFailedWindows", "--addWindowID", "--roundTo", "8"] + pops_args(12, 3)),
    dict(name="sparse_exclude", tool="popgenWindows.py", fixture="sparse",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "10", "--exclude", "{dir}/sparse_exclude.txt"] + pops_args(12, 3)),
    dict(name="sparse_include", tool="popgenWindows.py", fixture="sparse",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "10", "--include", "{dir}/sparse_include.txt"] + pops_args(12, 3)),
    dict(name="holes_minsites_nan", tool="popgenWindows.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "-w", "400", "-m", "60", "--minData", "0.5", "--roundTo", "8", "--writeFailedWindows"] + pops_args(6, 2)),
    dict(name="holes_sites_m0", tool="popgenWindows.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "sites", "-w", "150", "-m", "0", "--roundTo", "8"] + pops_args(6, 2)),
    dict(name="abba_pairs_popgen", tool="popgenWindows.py", fixture="abba_pairs",
         argv=["-g", "{geno}", "-f", "pairs", "-w", "2000", "-m", "50", "--roundTo", "8"] + pops_args(16, 4)),
    dict(name="abba_diplo_popgen", tool="popgenWindows.py", fixture="abba_diplo",
         argv=["-g", "{geno}", "-f", "diplo", "-w", "2000", "-m", "50", "--roundTo", "8"] + pops_args(16, 4)),
    dict(name="abba_phased_popgen", tool="popgenWindows.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "-w", "2000", "-m", "50", "--roundTo", "8"] + pops_args(16, 4)),
    dict(name="haplo_popgen", tool="popgenWindows.py", fixture="haplo",
         argv=["-g", "{geno}", "-f", "haplo", "-w", "1000", "-m", "20", "--roundTo", "8", "-p", "a", "s0_A,s0_B,s1_A,s1_B,s2_A",
               "-p", "b", "s2_B,s3_A,s3_B,s4_A,s4_B"]),
    dict(name="multi_popgen_all", tool="popgenWindows.py", fixture="multi",
         argv=["-g", "{geno}", "-f", "phased", "-w", "600", "-m", "20", "--roundTo", "8", "--analysis", "popDist", "popPairDist",
               "popFreq", "indPairDist", "indHet"] + pops_args(9, 3)),
    dict(name="multi_distmat", tool="distMat.py", fixture="multi",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1500", "-m", "20", "--outFormat", "raw"]),
    dict(name="multi_distmat_windows_id", tool="distMat.py", fixture="multi",
         argv=["-g", "{geno}", "-f", "phased", "-w", "700", "-m", "400", "--outFormat", "phylip", "--addWindowID", "--writeFailedWindows",
               "--windowDataOutFile", "{out}.windows"]),
    # ---- ABBABABAwindows.py ----
    dict(name="abba_windows", tool="ABBABABAwindows.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "20", "--minData", "0.5"] + abba_args(16)),
    dict(name="abba_windows_failed_id", tool="ABBABABAwindows.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "-w", "500", "-s", "250", "-m", "60", "--minData", "0.9", "--writeFailedWindows", "--addWindowID"] + abba_args(16)),
    # windows that have sites but not one good site: the reference prints sitesUsed = nan there (genomics.py:1693-1695), 0
    # only where good sites exist and none survives the derived-allele choice
    dict(name="abba_windows_none_good", tool="ABBABABAwindows.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "-w", "50", "-m", "1", "--minData", "1.0", "--writeFailedWindows"] + abba_args(16)),
    dict(name="abba_windows_sites", tool="ABBABABAwindows.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "sites", "-w", "400", "--overlap", "100", "-m", "50"] + abba_args(16)),
    dict(name="abba_windows_diplo", tool="ABBABABAwindows.py", fixture="abba_diplo",
         argv=["-g", "{geno}", "-f", "diplo", "-w", "1000", "-m", "20", "--minData", "0.5"] + abba_args(16)),
    # ---- less common flags ----
    dict(name="sparse_popsfile", tool="popgenWindows.py", fixture="sparse",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "10", "--popsFile", "{dir}/sparse_pops.txt",
               "-p", "north", "-p", "south", "--roundTo", "6"]),
    dict(name="mixed_haploid_flag", tool="popgenWindows.py", fixture="mixed",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "20", "--haploid", "s1,s6,s9", "--roundTo", "6",
               "--analysis", "popDist", "popPairDist", "popFreq", "indPairDist"] + pops_args(10, 2)),
    dict(name="mixed_inferploidy", tool="popgenWindows.py", fixture="mixed",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "20", "--inferPloidy", "--roundTo", "6",
               "--analysis", "popDist", "popPairDist", "indPairDist"] + pops_args(10, 2)),
    dict(name="mixed_ploidyfile_distmat", tool="distMat.py", fixture="mixed",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1500", "-m", "20", "--ploidyFile", "{dir}/mixed_ploidy.txt", "--includeSameWithSame"]),
    dict(name="abba_popsfile_exclude", tool="ABBABABAwindows.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "20", "--popsFile", "{dir}/abba_pops.txt",
               "--exclude", "{dir}/abba_exclude.txt", "-P1", "pop0", "-P2", "pop1", "-P3", "pop2", "-O", "pop3"]),
    dict(name="holes_distmat_minperind_samples", tool="distMat.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "10", "-Mi", "200", "--samples", "s0", "s2", "s3", "s5"]),
    dict(name="holes_distmat_sites", tool="distMat.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "sites", "-w", "300", "-O", "50", "-m", "100", "--outFormat", "nexus"]),
    dict(name="abba_freq_indfreqs", tool="freq.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "--indFreqs", "--target", "derived"]),
    # ---- --inferPloidy on a file whose cell widths change (missing #1) ----
    dict(name="ploidyshift_popgen", tool="popgenWindows.py", fixture="ploidyshift",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "20", "--inferPloidy", "--roundTo", "6",
               "--analysis", "popFreq", "popDist", "popPairDist"] + pops_args(12, 4)),
    dict(name="ploidyshift_popgen_sliding_ind", tool="popgenWindows.py", fixture="ploidyshift",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-s", "250", "-m", "20", "--inferPloidy", "--addWindowID", "--writeFailedWindows",
               "--analysis", "popDist", "popPairDist", "indPairDist", "hapStats"] + pops_args(12, 2)),
    # (-m 44 in windows of about 48 sites: pairs of haplotypes below the threshold, so what groupDistStats leaves in the cached distance
    # matrix -- its mask and the nan diagonal, genomics.py:959-963 -- shows in indPairDist and hapStats; the differential fuzz of round 6
    # found that state lost between the statistics of a window whose ploidies had been inferred)
    dict(name="ploidyshift_popgen_mask_state", tool="popgenWindows.py", fixture="ploidyshift",
         argv=["-g", "{geno}", "-f", "phased", "-w", "60", "-m", "44", "--inferPloidy", "--roundTo", "5",
               "--analysis", "popDist", "popPairDist", "indPairDist", "hapStats"] + pops_args(12, 3)),
    dict(name="ploidyshift_popgen_sites_pairs", tool="popgenWindows.py", fixture="ploidyshift_pairs",
         argv=["-g", "{geno}", "-f", "pairs", "--windType", "sites", "-w", "300", "--overlap", "100", "-m", "50", "--inferPloidy",
               "--roundTo", "5"] + pops_args(12, 3)),
    dict(name="ploidyshift_abba", tool="ABBABABAwindows.py", fixture="ploidyshift",
         argv=["-g", "{geno}", "-f", "phased", "-w", "800", "-m", "20", "--minData", "0.5", "--inferPloidy"] + abba_args(12)),
    dict(name="ploidyshift_fourpop", tool="fourPopWindows.py", fixture="ploidyshift_pairs",
         argv=["-g", "{geno}", "-f", "pairs", "-w", "800", "-m", "20", "--minData", "0.5", "--inferPloidy"] + abba_args(12)),
    dict(name="ploidyshift_distmat", tool="distMat.py", fixture="ploidyshift",
         argv=["-g", "{geno}", "-f", "phased", "-w", "700", "-m", "20", "--inferPloidy", "--includeSameWithSame"]),
    dict(name="ploidyshift_distmat_cat", tool="distMat.py", fixture="ploidyshift_pairs",
         argv=["-g", "{geno}", "-f", "pairs", "--windType", "cat", "--inferPloidy", "--outFormat", "raw"]),
    # ---- fourPopWindows.py (the reference needs np.NaN injected by the harness under NumPy 2) ----
    dict(name="fourpop_minor", tool="fourPopWindows.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "20", "--minData", "0.5"] + abba_args(16)),
    dict(name="fourpop_polarize", tool="fourPopWindows.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "20", "--polarize"] + abba_args(16)),
    dict(name="fourpop_fixed_failed_id", tool="fourPopWindows.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "-w", "2000", "-s", "1000", "-m", "8", "--fixed", "--writeFailedWindows", "--addWindowID"] + abba_args(16)),
    dict(name="fourpop_sites_diplo", tool="fourPopWindows.py", fixture="abba_diplo",
         argv=["-g", "{geno}", "-f", "diplo", "--windType", "sites", "-w", "400", "--overlap", "100", "-m", "50"] + abba_args(16)),
    dict(name="fourpop_holes_mindata0", tool="fourPopWindows.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "-w", "500", "-m", "5", "--minData", "0", "--writeFailedWindows",
               "-P1", "a", "s0", "-P2", "b", "s1", "-P3", "c", "s2,s3", "-O", "o", "s4,s5"]),
    # ---- distMat.py ----
    dict(name="holes_distmat_phylip", tool="distMat.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "10", "--windowDataOutFile", "{out}.windows"]),
    dict(name="holes_distmat_raw_same", tool="distMat.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1500", "-m", "10", "--outFormat", "raw", "--includeSameWithSame", "--roundTo", "6"]),
    # no -m: the default is 1 (distMat.py:123), not the window size
    dict(name="holes_distmat_default_minsites", tool="distMat.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "-w", "700", "--outFormat", "raw", "--writeFailedWindows", "--windowDataOutFile", "{out}.windows"]),
    dict(name="holes_distmat_cat_windows", tool="distMat.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "cat", "--outFormat", "raw", "--windowDataOutFile", "{out}.windows"]),
    dict(name="holes_distmat_cat_nexus", tool="distMat.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "cat", "--outFormat", "nexus", "--roundTo", "8"]),
]

CASES += [
    # ---- freq.py ----
    dict(name="abba_freq_counts", tool="freq.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased"] + pops_args(16, 4)),
    dict(name="abba_freq_derived", tool="freq.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "--target", "derived"] + pops_args(16, 4)),
    dict(name="abba_freq_derived_counts_keepnan", tool="freq.py", fixture="abba",
         argv=["-g", "{geno}", "-f", "phased", "--target", "derived", "--asCounts", "--keepNanLines", "--minData", "0.5"] + pops_args(16, 4)),
    dict(name="holes_freq_derived_threshold", tool="freq.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "--target", "derived", "--threshold", "0.5"] + pops_args(6, 2)),
    dict(name="abba_pairs_freq_alleles_allpop", tool="freq.py", fixture="abba_pairs",
         argv=["-g", "{geno}", "-f", "alleles"]),
]

CASES += [
    # ---- popgenWindows --analysis indHet / hapStats ----
    dict(name="sparse_het_hap_after_popdist", tool="popgenWindows.py", fixture="sparse",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "20", "--analysis", "popDist", "popPairDist", "indHet", "hapStats",
               "--hapDist", "0.05", "--roundTo", "8"] + pops_args(12, 3)),
    dict(name="holes_het_hap_only", tool="popgenWindows.py", fixture="holes",
         argv=["-g", "{geno}", "-f", "phased", "-w", "600", "-m", "30", "--analysis", "indHet", "hapStats", "--roundTo", "8"] + pops_args(6, 2)),
    # every population one cluster: H2 is the INTEGER 0 there (genomics.py:1092-1093), printed "0"
    dict(name="c1_hap_one_cluster", tool="popgenWindows.py", fixture="c1",
         argv=["-g", "{geno}", "-f", "phased", "-w", "2500", "-m", "50", "--analysis", "hapStats", "--hapDist", "0.9"] + pops_args(8, 2)),
    dict(name="c1_indpair_hap_exact", tool="popgenWindows.py", fixture="c1",
         argv=["-g", "{geno}", "-f", "phased", "-w", "2500", "-m", "50", "--analysis", "indPairDist", "hapStats", "--roundTo", "8"] + pops_args(8, 2)),
]

CASES += [
    # ---- one scaffold / four scaffolds: the multi-rank window-range plan (tests/test_dist.py) ----
    dict(name="one_popgen_overlap_failed_id", tool="popgenWindows.py", fixture="one",
         argv=["-g", "{geno}", "-f", "phased", "-w", "400", "-s", "150", "-m", "10", "--writeFailedWindows", "--addWindowID",
               "--roundTo", "6"] + pops_args(8, 2)),
    dict(name="one_popgen_stepgap", tool="popgenWindows.py", fixture="one",
         argv=["-g", "{geno}", "-f", "phased", "-w", "200", "-s", "500", "-m", "5", "--writeFailedWindows"] + pops_args(8, 2)),
    dict(name="one_popgen_sites", tool="popgenWindows.py", fixture="one",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "sites", "-w", "150", "-O", "50", "-m", "100", "--roundTo", "8",
               "--addWindowID"] + pops_args(8, 2)),
    dict(name="one_distmat_windows_id", tool="distMat.py", fixture="one",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1500", "-m", "20", "--outFormat", "raw", "--addWindowID", "--writeFailedWindows",
               "--windowDataOutFile", "{out}.windows"]),
    dict(name="four_popgen_id", tool="popgenWindows.py", fixture="four",
         argv=["-g", "{geno}", "-f", "phased", "-w", "500", "-m", "10", "--writeFailedWindows", "--addWindowID"] + pops_args(8, 4)),
    dict(name="four_abba_overlap", tool="ABBABABAwindows.py", fixture="four",
         argv=["-g", "{geno}", "-f", "phased", "-w", "600", "-s", "300", "-m", "10", "--minData", "0.5", "--writeFailedWindows",
               "--addWindowID"] + abba_args(8)),
    dict(name="four_fourpop", tool="fourPopWindows.py", fixture="four",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000", "-m", "10", "--minData", "0.5"] + abba_args(8)),
    # ---- predefined windows in the other drivers: a list out of file order, a scaffold the file does not hold, a window past
    #      the end of the file (ABBABABAwindows.py:31,328, distMat.py:33,118) ----
    dict(name="four_abba_predefined", tool="ABBABABAwindows.py", fixture="four",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "predefined", "--windCoords", "{dir}/four_coords.txt", "-m", "5",
               "--writeFailedWindows", "--addWindowID"] + abba_args(8)),
    dict(name="four_fourpop_predefined", tool="fourPopWindows.py", fixture="four",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "predefined", "--windCoords", "{dir}/four_coords.txt", "-m", "5",
               "--writeFailedWindows", "--addWindowID"] + abba_args(8)),
    dict(name="four_distmat_predefined", tool="distMat.py", fixture="four",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "predefined", "--windCoords", "{dir}/four_coords.txt", "-m", "5",
               "--outFormat", "raw", "--addWindowID", "--writeFailedWindows", "--windowDataOutFile", "{out}.windows"]),
    dict(name="four_abba_predefined_ordered", tool="ABBABABAwindows.py", fixture="four",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "predefined", "--windCoords", "{dir}/four_coords_ordered.txt", "-m", "5",
               "--writeFailedWindows", "--addWindowID"] + abba_args(8)),
]

AUX_FILES = {
    "four_coords.txt": ("chr2 100 900 a\nchr2 500 1500 b\nchr1 1 500 c\nchrX 1 100 d\nchr3 2000 2400 e\nchr4 2500 3500 f\n"
                        "chr4 5000 6000 g\n"),
    "four_coords_ordered.txt": "chr1 1 700 a\nchr1 400 1200 b\nchr2 100 900 c\nchr3 2000 2400 d\nchr4 2500 3500 e\nchr4 5000 6000 f\n",
    "sparse_coords.txt": "chr1 100 900 first\nchr1 500 1500 second\nchr1 4000 4100 third\nchr3 1 1000 onThree\nchr3 2000 2600 lastOne\n",
    "sparse_exclude.txt": "chr2\n",
    "sparse_include.txt": "chr1\nchr2\n",
    "sparse_pops.txt": "".join("s%d %s\n" % (d, "north" if d < 5 else "south" if d < 10 else "elsewhere") for d in range(12)),
    "abba_pops.txt": "".join("s%d pop%d\n" % (d, d // 4) for d in range(16)),
    "abba_exclude.txt": "chr2\n",
    "mixed_ploidy.txt": "".join("s%d %d\n" % (d, 1 if d in (1, 6, 9) else 2) for d in range(10)),
}


CASES += [
    # ---- positions beyond 2^31 (int64 from the tokenizers to the CSV) ----
    dict(name="bigpos_popgen_sites", tool="popgenWindows.py", fixture="bigpos",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "sites", "-w", "300", "-O", "100", "-m", "100", "--roundTo", "8",
               "--addWindowID"] + pops_args(8, 4)),
    dict(name="bigpos_popgen_coordinate", tool="popgenWindows.py", fixture="bigpos",
         argv=["-g", "{geno}", "-f", "phased", "-w", "1000000000", "-s", "500000000", "-m", "100", "--writeFailedWindows", "--addWindowID"] + pops_args(8, 4)),
    dict(name="bigpos_abba_sites", tool="ABBABABAwindows.py", fixture="bigpos",
         argv=["-g", "{geno}", "-f", "phased", "--windType", "sites", "-w", "400", "--overlap", "100", "-m", "50"] + abba_args(8)),
    dict(name="bigpos_freq", tool="freq.py", fixture="bigpos",
         argv=["-g", "{geno}", "-f", "phased"] + pops_args(8, 4)),
]
