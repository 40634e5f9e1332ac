"""Sample / population bookkeeping of the drop-in drivers.

Mirrors genomics.SampleData (genomics.py:1264-1290) and the haplotype naming / grouping that
genomics.genoToAlignment (genomics.py:1101-1127) derives from it, and turns them into the device slot
layout libpopgen_hip.so wants (populations contiguous, haplotypes of an individual adjacent).
"""
import string

import numpy as np


class SampleData:
    """Attribute-compatible stand-in for genomics.SampleData (genomics.py:1264-1290): `indNames` (individuals,
    caller-given first, then population members in order of first mention), `popNames`, `popNumbers`,
    `popInds` (keyed by both name and number) and `ploidy` (default 2)."""

    def __init__(self, indNames=None, popNames=None, popInds=None, popNumbers=None, ploidyDict=None):
        members = [list(m) for m in (popInds or [])]
        numbers = list(popNumbers) if popNumbers is not None else list(range(len(members)))
        names = list(popNames) if popNames is not None else [str(n) for n in numbers]
        if not (len(names) == len(members) == len(numbers)):
            raise AssertionError("Names, inds and numbers should be same length.")
        inds = list(indNames) if indNames is not None else []
        known = set(inds)
        self.popInds = {}
        for name, number, mem in zip(names, numbers, members):
            self.popInds[name] = mem
            self.popInds[number] = mem
            for ind in mem:
                if ind not in known:
                    known.add(ind)
                    inds.append(ind)
        self.popNames, self.popNumbers, self.indNames = names, numbers, inds
        self.ploidy = {ind: (ploidyDict[ind] if ploidyDict else 2) for ind in inds}

    def getPop(self, indName):
        """Population of an individual: None, a name, or a tuple when it was listed in several."""
        hits = tuple(p for p in self.popNames if indName in self.popInds[p])
        return None if not hits else (hits[0] if len(hits) == 1 else hits)


class HapLayout:
    """Device slot layout for the individuals of a SampleData, given the file's header names.

    slot order: populations in SampleData.popNames order, individuals inside a population in
    SampleData.indNames order, individuals without a population last; alleles of an individual adjacent.
    """

    def __init__(self, sampleData, file_names, genoFormat):
        self.sampleData = sampleData
        self.genoFormat = genoFormat
        file_names = list(file_names)
        pos_in_file = {}
        for k, nm in enumerate(file_names):
            pos_in_file.setdefault(nm, k)            # dict(zip(names, GTs)) keeps the LAST duplicate; headers are unique in practice
        missing = [nm for nm in sampleData.indNames if nm not in pos_in_file]
        if missing:
            raise KeyError("sample(s) not in the genotype file header: " + ",".join(missing[:5]))
        pop_id = {}
        for nm in sampleData.indNames:
            p = sampleData.getPop(nm)
            if isinstance(p, tuple):
                raise ValueError("sample %s is in more than one population (%s)" % (nm, ",".join(p)))
            pop_id[nm] = sampleData.popNames.index(p) if p is not None else -1
        order = sorted(range(len(sampleData.indNames)),
                       key=lambda k: (pop_id[sampleData.indNames[k]] if pop_id[sampleData.indNames[k]] >= 0 else 1 << 30, k))
        self.ind_order = [sampleData.indNames[k] for k in order]          # individuals in slot order
        self.hap_names, self.hap_sample_name, self.hap_group = [], [], []
        hap_pop, hap_sample = [], []
        self.max_ploidy = 1
        self.ind_slots = {}
        for s, nm in enumerate(self.ind_order):
            pl = sampleData.ploidy.get(nm)
            if pl is None:
                raise ValueError("ploidy of %s is not known: give --ploidy / --ploidyFile / --haploid (or --inferPloidy)" % nm)
            if pl < 1 or pl > 26:
                raise ValueError("ploidy of %s must be in 1..26" % nm)
            if genoFormat == "diplo" and pl != 2:
                raise AssertionError("Sample ploidy (%d) doesn't match number of sequences (2)" % pl)
            if genoFormat == "haplo" and pl != 1:
                raise AssertionError("Sample ploidy (%d) doesn't match number of sequences (1)" % pl)
            self.max_ploidy = max(self.max_ploidy, pl)
            self.ind_slots[nm] = list(range(len(hap_pop), len(hap_pop) + pl))
            for k in range(pl):
                # genomics.py:1114 / 1118: "ind_A","ind_B".. for ploidy > 1, bare name for haploids
                self.hap_names.append(nm + "_" + string.ascii_uppercase[k] if pl != 1 else nm)
                self.hap_sample_name.append(nm)
                self.hap_group.append(sampleData.popNames[pop_id[nm]] if pop_id[nm] >= 0 else None)
                hap_pop.append(pop_id[nm])
                hap_sample.append(s)
        self.n_hap = len(hap_pop)
        self.n_pops = len(sampleData.popNames)
        self.n_samp = len(self.ind_order)
        self.hap_pop = np.asarray(hap_pop, dtype=np.int32)
        self.hap_sample = np.asarray(hap_sample, dtype=np.int32)
        self.pop_sizes = [int(np.sum(self.hap_pop == p)) for p in range(self.n_pops)]
        # tokenizer column tables
        n_cols = len(file_names)
        self.col_ploidy = np.zeros(n_cols, dtype=np.int32)
        self.col_slot = np.full((n_cols, self.max_ploidy), -1, dtype=np.int32)
        for nm in self.ind_order:
            c = pos_in_file[nm]
            if self.col_ploidy[c]:
                raise ValueError("sample %s listed twice" % nm)
            sl = self.ind_slots[nm]
            self.col_ploidy[c] = len(sl)
            self.col_slot[c, :len(sl)] = sl
        # device-side expansion of packed `.pgeno` cells (pg_upload_packed_async): slot -> 2 * file column + allele index
        self.slot_src = np.full(self.n_hap, -1, dtype=np.int32)
        for c in range(n_cols):
            for k in range(int(self.col_ploidy[c])):
                self.slot_src[self.col_slot[c, k]] = 2 * c + k
        # order of the reference's Alignment rows: haplotypes sorted by name (genomics.py:1122)
        self.ref_order = np.argsort(np.array(self.hap_names))

    def sample_pair_index(self, s, t):
        """Index of unordered individual pair (slot-order indices) in pg_indpairdist output."""
        if s > t:
            s, t = t, s
        n = self.n_samp
        return s * n - s * (s - 1) // 2 + (t - s)

    def pop_pair_index(self, x, y):
        if x > y:
            x, y = y, x
        n = self.n_pops
        return x * n - x * (x - 1) // 2 + (y - x)
