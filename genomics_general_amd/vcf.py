"""Drop-in for the reference's VCF_processing/parseVCF.py (VCF -> `.geno`), the upstream producer of the engine's input, with
a second output route: `--packed out.pgeno` writes the tokenised, packed form directly, so that a VCF never has to exist as
`.geno` text.

The per-line work -- column split, REF/ALT allele table, site type, GT split, genotype filters, ploidy check, allele
look-up (VcfSite.__init__ / getGenotype, parseVCF.py:49-191) and the per-site filters of the main loop (parseVCF.py:367-370) --
runs in the native, multi-threaded pg_encode_vcf (csrc/pg_vcf.cpp); this module parses the command line (same flags as
parseVCF.py:268-303), streams the input in blocks and renders the rows.

Supported: -i/-o (.gz by suffix, stdin/stdout), -s/--samples, --include/--exclude(/File), --minQual, --gtf (repeatable), --skipIndels,
--excludeDuplicates, --maxREFlen, --ploidy, --ploidyFile, --ploidyMismatchToMissing, --keepPartial, --addRefTrack, --noHeader,
--missing, --outSep (of several characters: the line loop below, text only), and --field NAME (the values of another FORMAT field instead of genotypes:
plain text out, a line loop on the host -- that output is not an input of the engine), and --simplifyALT / --expandMulti (ALT
haplotypes of freebayes records rewritten to the length of REF from their CIGAR strings; one row per base with --expandMulti:
_cigar_main, likewise a line loop on the host).  Alleles longer than one base (indels without --skipIndels;
the homozygous-reference calls of a deletion site even with it) are printed as strings by the text route, exactly as the
reference prints them; the packed route stores such calls as missing (use --maxREFlen 1 to drop those sites altogether)."""
import argparse
import ctypes as C
import gzip
import json
import os
import queue
import sys
import threading
import time

import numpy as np

from . import _lib, genoio
from ._lib import check

SITE_TYPES = {"MONO": 1, "SNP": 2, "INDEL": 4}
GT_TYPES = {"Het": 1, "HomRef": 2, "Missing": 4, "HomAlt": 8}


class _Filter(C.Structure):
    _fields_ = [("flag", C.c_char_p), ("min", C.c_double), ("max", C.c_double), ("site_types", C.c_int), ("gt_types", C.c_int),
                ("samples", C.c_void_p)]


def _parse_gtf(tokens):
    """One --gtf option (the `key=value` words of parseVCF.py:255-266) -> the fields of a pg_vcf_filter: FORMAT flag, closed
    value range, and the optional selectors (site types, genotype types, samples) the filter is restricted to."""
    spec = {"min": -np.inf, "max": np.inf}
    for tok in tokens:
        key, eq, value = tok.partition("=")
        if not eq or key not in ("flag", "min", "max", "siteTypes", "gtTypes", "samples"):
            raise ValueError("Bad genotype filter specification. See help.")
        if key in ("min", "max"):
            try:
                spec[key] = float(value)
            except ValueError:
                raise ValueError("Bad genotype filter specification. See help.")
        elif key == "flag":
            spec[key] = value
        else:
            spec[key] = value.split(",")
    if "flag" not in spec:
        raise ValueError("Bad genotype filter specification. See help.")
    return spec


def _open_out(path):
    if not path:
        return sys.stdout.buffer
    # `-o out.geno.gz`: BGZF, deflated by the library's host threads -- what the reference's recipe `parseVCF.py ... | bgzip > out.geno.gz`
    # (VCF_processing/README.md:33) produces, a valid gzip file for its readers (gzip.open there), and the form the drivers inflate on
    # the GPU.  (parseVCF.py:358 itself would write one serial gzip stream.)
    return genoio.BgzfWriter(path) if path.endswith(".gz") else open(path, "wb")


class _AsyncOut:
    """writes of the output file on a thread of their own (for `.gz`: genoio.BgzfWriter's deflate pool + the file), two blocks deep"""

    def __init__(self, out):
        self.out = out
        self.q = queue.Queue(2)
        self.err = None
        self.th = threading.Thread(target=self._run, name="pg-vcf-write", daemon=True)
        self.th.start()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if self.err is None:
                try:
                    if isinstance(item, tuple):               # ("members", bytes): BGZF members the device deflated
                        self.out.write_members(item[1])
                    else:
                        self.out.write(memoryview(item))
                except BaseException as exc:                  # kept for the caller; the queue is still drained
                    self.err = exc

    def write(self, data):
        if self.err is not None:
            raise self.err
        self.q.put(data)

    def close(self):
        self.q.put(None)
        self.th.join()
        if self.err is not None:
            raise self.err

    def abort(self):
        self.err = self.err or RuntimeError("aborted")        # (what is still queued is not written)
        self.q.put(None)
        self.th.join()


def _text_blocks(reader, block_bytes, n_threads=0, plan=None, gz_rows=False):
    """the input in blocks of whole lines, as pairs (text, None) for the host parser or (raw, where) for the device's.

    A bgzip-compressed VCF (what `bgzip` / GATK / bcftools write) is read as spans of deflated members (genoio.BgzfFile.read_span).
    With a device and an option set its parser takes (`plan`: text output, no --packed; Engine.vcf_config), a block is handed on as it
    is -- a BgzfSpan, or a view of the memory-mapped plain file with its (descriptor, offset) -- and the caller submits it to the
    device (pg_vcf_dev_*): the text of a bgzipped VCF then never exists on the host.  Otherwise a span is inflated into a ring of
    buffers: ON THE DEVICE when there is one (the members cross PCIe deflated, k_inflate takes a wavefront per member, the text comes
    back into page-locked memory; PG_BGZF_DEVICE=0: never), else by the library's host threads; everything else: the reader's own
    blocks.  gz_rows: the device also deflates the rows it makes (`-o out.geno.gz`: k_deflate, BGZF members).  PG_VCF_DEVICE=0: the host parser always; =1: the device's even for a small file (tests)."""
    bg = isinstance(getattr(reader, "f", None), genoio.BgzfFile)
    made = {}
    maker = None
    force = os.environ.get("PG_VCF_DEVICE", "")
    want_dev = plan is not None and force != "0"
    # (a small file is over before a device context exists -- 0.1 - 0.3 s --: the host threads take it; PG_VCF_WAIT_FOR_DEVICE: always)
    wait = bool(os.environ.get("PG_VCF_WAIT_FOR_DEVICE")) or (want_dev and force == "1")
    gzs = isinstance(getattr(reader, "f", None), genoio.GzipStream)           # ONE gzip stream (`gzip in.vcf`): the host inflates it, chunks side by side
    size = getattr(reader.f, "size", 0) if bg else (reader.input_size() if reader.mm is not None else (4 * reader.input_size() if gzs else 0))
    worth = (bg or (want_dev and (reader.mm is not None or gzs))) and (size >= (32 << 20) or wait)
    def make():                                   # the device context takes 0.1 - 0.3 s: the first blocks do not wait for it
        try:
            from .engine import Engine
            eng = Engine(int(os.environ.get("PG_DEVICE", "0")))
            if want_dev:
                made["why_not"] = eng.vcf_config(plan)
                made["dev"] = made["why_not"] is None
                eng.vcf_set_output(bool(gz_rows))
            made["engine"] = eng
        except BaseException as exc:
            made["error"] = exc

    if worth and (os.environ.get("PG_BGZF_DEVICE", "1") != "0" or want_dev) and _lib.device_count() > 0:
        maker = threading.Thread(target=make, name="pg-vcf-context", daemon=True)
        maker.start()
    # a pipe (`bcftools view ... | parseVCF.py`): its size shows only as it arrives -- the context is made once a first block of 32 MB
    # has come in whole
    piped_in = want_dev and not worth and getattr(reader, "path", None) is None and os.environ.get("PG_BGZF_DEVICE", "1") != "0"
    if bg:
        reader.spans = True
    dev_bytes = int(os.environ.get("PG_VCF_DEVICE_BYTES", 256 << 20))       # (128 MB / 256 / 512 / 1 GB: 0.68 / 0.54 / 0.60 / 0.68 s for 6 GB of bgzipped VCF, profiles/r06)
    ring, turn = [None] * 4, 0                  # one block with the parser, two queued, one being filled
    info = {"bgzf": bg, "device_inflate": False, "inflate_kernel_ms": 0.0, "inflate_s": 0.0, "blocks": 0, "blocks_inflated_on_device": 0,
            "blocks_parsed_on_device": 0}
    try:
        while True:
            # blocks of 32 MB for the host parser while the device context is being made (the switch comes as soon as it exists), then
            # -- PG_VCF_DEVICE_BYTES -- blocks large enough to fill the chip: k_deflate takes a wavefront per 64 KiB of rows, a member
            # costs it 7.6 ms whatever else runs, and the rows of 128 MB of VCF text are 570 members on 3072 wave slots
            # (profiles/r06/vcf_gz_to_gz_kernel_stats*.csv)
            size = block_bytes
            if "PG_STREAM_BYTES" not in os.environ and (worth or piped_in) and want_dev:
                size = dev_bytes if (maker is None and made.get("dev")) else ((32 << 20) if (maker is not None or piped_in) else block_bytes)
            blk = reader.read_block(size)
            if len(blk) == 0:
                break
            info["blocks"] += 1
            if piped_in and len(blk) >= (32 << 20) and _lib.device_count() > 0:
                piped_in, worth = False, True
                maker = threading.Thread(target=make, name="pg-vcf-context", daemon=True)
                maker.start()
            if maker is not None and (not maker.is_alive() or wait):
                maker.join()
                maker = None
                if "error" in made:
                    raise made["error"]
                if bg:
                    reader.f.alloc = made["engine"].pinned.empty          # (the next spans arrive in page-locked memory)
                ring = [None] * 4
                _text_blocks.engine = made["engine"]
                if made.get("why_not"):
                    info["device_parser_not_taken"] = made["why_not"]
            eng = made.get("engine") if maker is None else None
            if eng is not None and made.get("dev") and os.environ.get("PG_BGZF_DEVICE", "1") != "0":
                info["blocks_parsed_on_device"] += 1
                yield blk, (reader.file_range(blk) if not isinstance(blk, genoio.BgzfSpan) else None) or ()
                continue
            if isinstance(blk, genoio.BgzfSpan):
                if os.environ.get("PG_BGZF_DEVICE", "1") == "0":
                    eng = None
                need = len(blk.head) + blk.members_text_len()
                buf = ring[turn % 4]
                if buf is None or buf.size < need:
                    buf = ring[turn % 4] = (eng.pinned.empty if eng is not None else np.empty)((need + need // 16 + (1 << 16),), np.uint8)
                turn += 1
                t0 = time.perf_counter()
                if eng is not None:
                    h = len(blk.head)
                    if h:
                        buf[:h] = np.frombuffer(blk.head, dtype=np.uint8)
                    info["inflate_kernel_ms"] += eng.inflate_members(blk.comp, blk.tab, buf[h:need])
                    info["blocks_inflated_on_device"] += 1
                    info["device_inflate"] = True
                    blk = buf[:len(blk)]
                else:
                    blk = blk.inflate_into(buf, None, n_threads)
                info["inflate_s"] += time.perf_counter() - t0
            yield blk, None
    finally:
        if maker is not None:
            maker.join()
        _text_blocks.last_info = info             # (the engine lives on: the parser may still be reading the last block in its memory)
        _text_blocks.engine = made.get("engine")


def _read_ahead(blocks, depth=2):
    """the blocks of an iterator, produced by a thread `depth` blocks ahead of the consumer"""
    q = queue.Queue(depth)
    stop = threading.Event()
    done = object()

    def run():
        try:
            for body in blocks:
                q.put(body)
                if stop.is_set():
                    break
            q.put(done)
        except BaseException as exc:
            q.put(exc)

    th = threading.Thread(target=run, name="pg-vcf-read", daemon=True)
    th.start()
    try:
        while True:
            body = q.get()
            if isinstance(body, BaseException):
                raise body
            if body is done:
                return
            yield body
    finally:
        stop.set()
        while th.is_alive():                                  # a reader blocked on a full queue
            try:
                q.get_nowait()
            except queue.Empty:
                pass
            th.join(0.05)


def _last_key(body):
    """CHROM and POS tokens of the last data line of a block (the duplicate test of the next block starts from them)"""
    mv = memoryview(body)
    end = len(mv)
    while end > 0:                                   # whole lines from the back, however long they are
        view = bytes(mv[max(end - (1 << 16), 0):end])
        nl = view.rfind(b"\n", 0, len(view) - 1 if view.endswith(b"\n") else len(view))
        span = 1 << 16
        while nl < 0 and end - span > 0:             # the line is longer than the tail looked at: look further back
            span *= 4
            view = bytes(mv[max(end - span, 0):end])
            nl = view.rfind(b"\n", 0, len(view) - 1 if view.endswith(b"\n") else len(view))
        line = view[nl + 1:]
        tok = line.split(None, 2)
        if len(tok) >= 2 and not line.startswith(b"#"):
            return tok[0], tok[1]
        end -= len(line)
    return None, None


def _contig_lists(args):
    """--include / --exclude and their file forms as two lists (parseIncludeExcludeArgs, parseVCF.py:306-330)"""
    include, exclude = [], []
    if args.include:
        include += args.include.split(",")
    if args.exclude:
        exclude += args.exclude.split(",")
    if args.includeFile:
        with open(args.includeFile, "rt") as f:
            include += [c.strip() for c in f.read().split("\n")]
    if args.excludeFile:
        with open(args.excludeFile, "rt") as f:
            exclude += [c.strip() for c in f.read().split("\n")]
    return include, exclude


def _field_main(args):
    """`--field NAME`: one row per site that passes the site filters (contigs, --minQual, --maxREFlen, --excludeDuplicates), the
    samples' values of FORMAT field NAME (parseVCF.py:371, getGenoField 185-191), `--missing` (default ".") where a sample has
    none.  Genotype filters, ploidy and indel options play no part, as in the reference; --simplifyALT / --expandMulti do, the way
    they do there: the records' INFO and CIGAR strings are read (and stop the run where they are not what that code expects), and
    --expandMulti prints a record as one row per base of REF holding one CHARACTER of every value."""
    if args.packed:
        raise SystemExit("parseVCF.py: --field writes text only (no --packed)")
    include, exclude = _contig_lists(args)
    include, exclude = set(include), set(exclude)
    absent = "." if args.missing is None else args.missing
    inp = (gzip.open(args.inFile, "rt") if args.inFile.endswith(".gz") else open(args.inFile, "rt")) if args.inFile else sys.stdin
    out = (gzip.open(args.outFile, "wt") if args.outFile.endswith(".gz") else open(args.outFile, "wt")) if args.outFile else sys.stdout
    cols = None
    for line in inp:
        if line.startswith("#CHROM"):
            cols = line.split()
            break
    assert cols is not None and len(cols) >= 9, "no #CHROM header line in the VCF"
    col_of = {nm: k for k, nm in enumerate(cols)}               # a repeated sample name: its last column, as dict(zip()) keeps it
    samples = args.samples.split(",") if args.samples else list(cols[9:])
    for s in samples:
        assert s in cols[9:], "Sample {} not in VCF header\n".format(s)
    where = [col_of[s] for s in samples]
    if not args.noHeader:
        out.write(args.outSep.join(["#CHROM", "POS"] + (["REF"] if args.addRefTrack else []) + samples) + "\n")
    import re
    split_gt = re.compile("[/|]")
    simplify = args.simplifyALT or args.expandMulti
    last = None
    for line in inp:
        f = line.split()
        if not f or f[0][0] == "#":
            continue
        if args.excludeDuplicates:
            if (f[0], f[1]) == last:
                continue
            last = (f[0], f[1])
        chrom, pos, ref, qual = f[0], int(f[1]), f[3], f[5]
        if simplify:
            # (VcfSite.__init__ reads INFO and rewrites the ALT haplotypes whatever is printed: a record without CIGAR, an INFO flag
            # without `=` or a malformed CIGAR string ends the run here as it does there)
            info = dict(x.split("=") for x in f[7].split(";"))
            cigars = info["CIGAR"].split(",")
            for k, a in enumerate(f[4].split(",") if f[4] != "." else []):
                _simplify_alt(a, cigars[k])
        if (exclude and chrom in exclude) or (include and chrom not in include):
            continue
        if args.minQual:
            try:
                if float(qual) < args.minQual:
                    continue
            except ValueError:
                pass
        if args.maxREFlen and len(ref) > args.maxREFlen:
            continue
        # a sample's FORMAT data as the reference holds them: dict(zip(names, values)) -- of a name that occurs twice the last value
        # the cell still has --, and next to a GT its "alleles" (a tuple) and "phase" under those names (VcfSite.__init__, 93-96)
        keys = f[8].split(":")
        occ = [i for i, nm in enumerate(keys) if nm == args.field]
        gt_occ = [i for i, nm in enumerate(keys) if nm == "GT"] if args.field in ("alleles", "phase") else []
        vals = []
        for c in where:
            parts = f[c].split(":")
            n = min(len(parts), len(keys))
            v = absent
            for i in reversed(occ):
                if i < n:
                    v = parts[i]
                    break
            for i in reversed(gt_occ):
                if i < n:
                    v = tuple(split_gt.split(parts[i])) if args.field == "alleles" else "|" if "|" in parts[i] else "/"
                    break
            vals.append(v)
        if args.expandMulti:
            # (parseVCF.py:380-385 runs for --field as well: row x of a record holds CHARACTER x of every value, and a value shorter
            # than REF ends the run with an IndexError, as there)
            for x in range(len(ref)):
                out.write(args.outSep.join([chrom, str(pos + x)] + ([ref[x]] if args.addRefTrack else []) + [v[x] for v in vals]) + "\n")
            continue
        out.write(args.outSep.join([chrom, str(pos)] + ([ref] if args.addRefTrack else []) + vals) + "\n")
    if out is not sys.stdout:
        out.close()
    return 0


def _simplify_alt(alt, cigar, missing="N"):
    """An ALT haplotype brought to the coordinates of REF with its CIGAR string (freebayes; parseVCF.py:25-46): matches and
    mismatches copy their bases, an insertion skips them, a deletion leaves `missing` in their place."""
    import re
    ops = re.findall(r"\d+|[MXDI]", cigar)
    out, at = [], 0
    if len(ops) % 2:
        raise ValueError("Malformed CIGAR: " + cigar)
    for count, op in zip(ops[0::2], ops[1::2]):
        if not count.isdigit() or op.isdigit():
            raise ValueError("Malformed CIGAR: " + cigar)
        n = int(count)
        if op in "MX":
            out.append(alt[at:at + n])
            at += n
        elif op == "I":
            at += n
        else:
            out.append(missing * n)
    return "".join(out)


def _gt_type(alleles):
    kinds = set(alleles)                                            # GTtype, parseVCF.py:13-18
    return "Het" if len(kinds) > 1 else "HomRef" if "0" in kinds else "Missing" if "." in kinds else "HomAlt"


def _cigar_main(args):
    """`--simplifyALT` / `--expandMulti` (the latter implies the former, parseVCF.py:337): every ALT haplotype of a freebayes
    record is rewritten to the length of REF from the CIGAR strings of its INFO column (VcfSite.__init__, parseVCF.py:72-77), the
    genotypes are looked up among the rewritten alleles (getGenotype, parseVCF.py:117-166), and with --expandMulti a record of a
    REF of n bases becomes n rows of single bases at POS .. POS + n - 1 (parseVCF.py:160-161, 380-386) -- rows the engine's
    tokenizers take.  A plain line loop on the host, like --field: records of this kind are a small share of a VCF, and their
    rows are of varying width.  Errors as the reference raises them: a record without CIGAR in INFO (KeyError), an INFO flag
    without `=` (ValueError), a ploidy mismatch without --ploidyMismatchToMissing (ValueError)."""
    import re
    if args.packed:
        raise SystemExit("parseVCF.py: --simplifyALT / --expandMulti write text only (no --packed)")
    include, exclude = _contig_lists(args)
    include, exclude = set(include), set(exclude)
    if include:
        sys.stderr.write("{} contigs will be included.".format(len(include)))
    if exclude:
        sys.stderr.write("{} contigs will be excluded.".format(len(exclude)))
    inp = (gzip.open(args.inFile, "rt") if args.inFile.endswith(".gz") else open(args.inFile, "rt")) if args.inFile else sys.stdin
    out = (gzip.open(args.outFile, "wt") if args.outFile.endswith(".gz") else open(args.outFile, "wt")) if args.outFile else sys.stdout
    cols = None
    for line in inp:
        if line.startswith("#CHROM"):
            cols = line.split()
            break
    assert cols is not None and len(cols) >= 9, "no #CHROM header line in the VCF"
    col_of = {nm: k for k, nm in enumerate(cols)}
    samples = args.samples.split(",") if args.samples else list(cols[9:])
    for s in samples:
        assert s in cols[9:], "Sample {} not in VCF header\n".format(s)
    where = [col_of[s] for s in samples]
    ploidy = {s: args.ploidy for s in samples}
    if args.ploidyFile:
        with open(args.ploidyFile, "rt") as pf:
            for f in (ln.split() for ln in pf):
                if f:
                    ploidy[f[0]] = int(f[1])
    filters = []
    for g in [_parse_gtf(g) for g in args.gtf] if args.gtf else []:
        filters.append(g)
    expand, must_match, keep_partial = args.expandMulti, args.skipIndels, args.keepPartial
    simplify = args.simplifyALT or args.expandMulti                 # neither: a --missing / --outSep of several characters (main)
    sep = args.outSep
    if not args.noHeader:
        out.write(sep.join(["#CHROM", "POS"] + (["REF"] if args.addRefTrack else []) + samples) + "\n")
    split_gt = re.compile("[/|]")
    last = None
    for line in inp:
        f = line.split()
        if not f or f[0][0] == "#":
            continue
        if args.excludeDuplicates:
            if (f[0], f[1]) == last:
                continue
            last = (f[0], f[1])
        chrom, pos, ref, qual = f[0], int(f[1]), f[3], f[5]
        alts = f[4].split(",") if f[4] != "." else []
        if simplify:
            info = dict(x.split("=") for x in f[7].split(";"))      # (a flag without `=` is a ValueError here as there)
            cigars = info["CIGAR"].split(",")
            alts = [_simplify_alt(a, cigars[k]) for k, a in enumerate(alts)]
        if (exclude and chrom in exclude) or (include and chrom not in include):
            continue
        if args.minQual:
            try:
                if float(qual) < args.minQual:
                    continue
            except ValueError:
                pass
        if args.maxREFlen and len(ref) > args.maxREFlen:
            continue
        alleles = {str(k): a for k, a in enumerate([ref] + alts)}
        same_len = {k: len(a) == len(ref) for k, a in alleles.items()}
        site_type = "MONO" if not alts else "SNP" if all(same_len.values()) else "INDEL"
        absent = args.missing if args.missing is not None else ("N" if not expand or len(ref) == 1 else ["N"] * len(ref))
        keys = f[8].split(":")
        cells = []
        for s, c in zip(samples, where):
            data = dict(zip(keys, f[c].split(":")))
            calls = tuple(split_gt.split(data["GT"]))
            phase = "|" if "|" in data["GT"] else "/"
            ok = True
            for g in filters:
                if "siteTypes" in g and site_type not in g["siteTypes"]:
                    continue
                if "gtTypes" in g and _gt_type(calls) not in g["gtTypes"]:
                    continue
                if "samples" in g and s not in g["samples"]:
                    continue
                try:
                    vals = [float(v) for v in data[g["flag"]].split(",")]
                    ok = all(g["min"] <= v for v in vals) and all(v <= g["max"] for v in vals)
                except (KeyError, ValueError):
                    ok = False
                if not ok:
                    break
            if ploidy[s] != len(calls):
                if not args.ploidyMismatchToMissing:
                    raise ValueError("Sample {} at {}:{} genotype {} does not match explected ploidy of {}".format(
                        s, chrom, pos, data["GT"], ploidy[s]))
                ok = False
            if ok:
                try:
                    got = [alleles[a] if (not must_match or same_len[a]) else absent for a in calls]
                    if not keep_partial and any(a is absent or a == absent for a in got):
                        got = [absent] * ploidy[s]
                except KeyError:                                    # `.` or an index beyond the ALT list
                    got = [absent] * ploidy[s]
            else:
                got = [absent] * ploidy[s]
            cells.append(tuple(phase.join(a[i] for a in got) for i in range(len(ref))) if expand else phase.join(got))
        if expand:
            for x in range(len(ref)):
                out.write(sep.join([chrom, str(pos + x)] + ([ref[x]] if args.addRefTrack else []) + [c[x] for c in cells]) + "\n")
        else:
            out.write(sep.join([chrom, str(pos)] + ([ref] if args.addRefTrack else []) + cells) + "\n")
    if out is not sys.stdout:
        out.close()
    return 0


class Plan:
    """The option set of one parseVCF run in the forms the native parsers take (pg_encode_vcf on the host, pg_vcf_dev_* on the device,
    tests/vcf_emul.cpp): selected sample columns and ploidies, genotype filters, contig list, option bits."""

    def __init__(self, args, vcf_samples):
        self.args = args
        include, exclude = [], []                                     # parseIncludeExcludeArgs, parseVCF.py:306-330
        if args.include:
            include += args.include.split(",")
        if args.exclude:
            exclude += args.exclude.split(",")
        if args.includeFile:
            with open(args.includeFile, "rt") as f:
                include += [c.strip() for c in f.read().split("\n")]
        if args.excludeFile:
            with open(args.excludeFile, "rt") as f:
                exclude += [c.strip() for c in f.read().split("\n")]
        if include:
            sys.stderr.write("{} contigs will be included.".format(len(set(include))))
        if exclude:
            sys.stderr.write("{} contigs will be excluded.".format(len(set(exclude))))
        self.vcf_samples = list(vcf_samples)
        samples = args.samples.split(",") if args.samples else None
        if samples:
            for s in samples:
                assert s in vcf_samples, "Sample {} not in VCF header\n".format(s)
        else:
            samples = list(vcf_samples)
        self.samples = samples
        ploidy = {s: args.ploidy for s in samples}
        if args.ploidyFile:
            with open(args.ploidyFile, "rt") as pf:
                for ln in pf:
                    f = ln.split()
                    if f and f[0] in ploidy:
                        ploidy[f[0]] = int(f[1])
        self.pl = np.array([ploidy[s] for s in samples], dtype=np.int32)
        if np.any((self.pl < 1) | (self.pl > 2)):
            raise SystemExit("parseVCF.py: this drop-in holds ploidy 1 or 2")
        # dict(zip(headers, elements)) keeps the LAST of duplicated sample names
        col_of = {nm: k for k, nm in enumerate(vcf_samples)}
        self.sel_col = np.array([col_of[s] for s in samples], dtype=np.int32)
        self.n_sel = len(samples)
        # ---- genotype filters ----
        gtf = [_parse_gtf(g) for g in args.gtf] if args.gtf else []
        self.keep_alive = []
        self.n_filters = len(gtf)
        self.Farr = Farr = (_Filter * max(len(gtf), 1))()
        for k, g in enumerate(gtf):
            Farr[k].flag = g["flag"].encode()
            Farr[k].min, Farr[k].max = g["min"], g["max"]
            Farr[k].site_types = sum(SITE_TYPES.get(t, 0) for t in set(g.get("siteTypes", [])))     # a set: a repeated name is one bit
            Farr[k].gt_types = sum(GT_TYPES.get(t, 0) for t in set(g.get("gtTypes", [])))
            if "siteTypes" in g and Farr[k].site_types == 0:
                Farr[k].site_types = 1 << 30                          # names that match no site type: the filter never applies
            if "gtTypes" in g and Farr[k].gt_types == 0:
                Farr[k].gt_types = 1 << 30
            if "samples" in g:
                m = np.array([1 if s in g["samples"] else 0 for s in samples], dtype=np.uint8)
                self.keep_alive.append(m)
                Farr[k].samples = m.ctypes.data
        self.flags = ((1 if args.skipIndels else 0) | (2 if args.keepPartial else 0) | (4 if args.ploidyMismatchToMissing else 0) |
                      (8 if args.excludeDuplicates else 0))
        self.contig_mode, self.contigs = 0, b""
        if include and exclude:                                       # the reference applies both: keep included minus excluded
            self.contig_mode, self.contigs = 1, "\n".join(sorted(set(include) - set(exclude))).encode()
        elif include:
            self.contig_mode, self.contigs = 1, "\n".join(sorted(set(include))).encode()
        elif exclude:
            self.contig_mode, self.contigs = 2, "\n".join(sorted(set(exclude))).encode()
        self.missing = args.missing if args.missing is not None else "N"
        self.sep = args.outSep
        self.bufs = {}

    def header_line(self):
        sep = self.sep.encode()
        return sep.join([b"#CHROM", b"POS"] + ([b"REF"] if self.args.addRefTrack else []) + [s.encode() for s in self.samples]) + b"\n"

    def site_args(self):
        """the arguments every native parser starts with behind its text (pg_encode_vcf's order)"""
        a = self.args
        vp = lambda x: C.c_void_p(x.ctypes.data)                      # noqa: E731
        return (len(self.vcf_samples), self.n_sel, vp(self.sel_col), vp(self.pl), self.flags, C.c_double(float(a.minQual or 0)),
                int(a.maxREFlen or 0), self.Farr, self.n_filters, C.c_char_p(self.contigs), len(self.contigs), self.contig_mode,
                C.c_char(self.missing.encode()))

    def _arr(self, name, shape, dtype):
        """arrays of the host parser's outputs, kept from block to block (pg_encode_vcf writes every field of a kept row)"""
        a = self.bufs.get(name)
        if a is None or a.shape[0] < shape[0]:
            a = self.bufs[name] = np.empty(shape, dtype=dtype)
        return a

    def host_parse(self, ptr, nbytes, prev_chrom=None, prev_pos=None, n_threads=0):
        """pg_encode_vcf over the block of whole lines at ptr: the number of kept sites, the calls longer than one base, and the
        arrays of its rows (for host_render / the packed writer)"""
        L = _lib.lib()
        fn = L.pg_encode_vcf
        fn.restype = C.c_int
        n_sel = self.n_sel
        nl = C.c_int64(0)
        check(L.pg_count_lines(ptr, nbytes, C.byref(nl)))
        cap = int(nl.value) + 1
        A = {"chars": self._arr("chars", (cap, 2 * n_sel), np.uint8), "aidx": self._arr("aidx", (cap, 2 * n_sel), np.int8),
             "phase": self._arr("phase", (cap, n_sel), np.uint8), "rflag": self._arr("rflag", (cap,), np.uint8),
             "pos": self._arr("pos", (cap,), np.int64), "coff": self._arr("coff", (cap,), np.int64), "clen": self._arr("clen", (cap,), np.int32),
             "roff": self._arr("roff", (cap,), np.int64), "rlen": self._arr("rlen", (cap,), np.int32),
             "aoff": self._arr("aoff", (cap,), np.int64), "alen": self._arr("alen", (cap,), np.int32)}
        vp = lambda a: C.c_void_p(a.ctypes.data)                      # noqa: E731
        n, nmb = C.c_int64(0), C.c_int64(0)
        check(fn(ptr, C.c_size_t(nbytes), *self.site_args(),
                 C.c_char_p(prev_chrom), len(prev_chrom or b""), C.c_char_p(prev_pos), len(prev_pos or b""),
                 vp(A["chars"]), vp(A["aidx"]), vp(A["phase"]), vp(A["rflag"]), vp(A["pos"]), vp(A["coff"]), vp(A["clen"]), vp(A["roff"]),
                 vp(A["rlen"]), vp(A["aoff"]), vp(A["alen"]), C.c_int64(cap), C.byref(n), C.byref(nmb), int(n_threads)))
        return int(n.value), int(nmb.value), A

    def host_render(self, ptr, k, A, n_threads=0):
        """the `.geno` text of the k rows host_parse left (pg_vcf_render_rows: a sizing call, then the bytes)"""
        L = _lib.lib()
        vp = lambda a: C.c_void_p(a.ctypes.data)                      # noqa: E731
        size = C.c_int64(0)
        rargs = (ptr, k, self.n_sel, vp(self.pl), vp(A["chars"]), vp(A["aidx"]), vp(A["phase"]), vp(A["rflag"]), vp(A["pos"]), vp(A["coff"]),
                 vp(A["clen"]), vp(A["roff"]), vp(A["rlen"]), vp(A["aoff"]), vp(A["alen"]), C.c_char(self.sep.encode()),
                 C.c_char(self.missing.encode()), 1 if self.args.addRefTrack else 0)
        check(L.pg_vcf_render_rows(*rargs, None, 0, C.byref(size), int(n_threads)))
        text = np.empty(size.value, dtype=np.uint8)
        check(L.pg_vcf_render_rows(*rargs, vp(text), size.value, C.byref(size), int(n_threads)))
        return text


def make_parser():
    ap = argparse.ArgumentParser(prog="parseVCF.py")
    ap.add_argument("-o", "--outFile", help="Output .geno file")
    ap.add_argument("-s", "--samples", help="sample names (separated by commas)")
    ap.add_argument("--include", help="include contigs (separated by commas)")
    ap.add_argument("--includeFile", help="File of contigs (one per line)")
    ap.add_argument("--exclude", help="exclude contigs (separated by commas)")
    ap.add_argument("--excludeFile", help="File of contigs (one per line)")
    ap.add_argument("--minQual", help="Minimum QUAL for a site", type=int)
    ap.add_argument("--gtf", help="Genotype filter. Syntax: flag=X min=X max=X siteTypes=X,X.. gtTypes=X,X.. samples=X,X..",
                    action="append", nargs="+")
    ap.add_argument("--skipIndels", help="Skip indels", action="store_true")
    ap.add_argument("--excludeDuplicates", help="Only include the first in a series of duplicated positions", action="store_true")
    ap.add_argument("--simplifyALT", action="store_true", help="Simplify multi-site alternate alleles using CIGAR (as in Freebayes output)")
    ap.add_argument("--expandMulti", action="store_true", help="Expand multi-site alleles (also sets simplifyALT)")
    ap.add_argument("--maxREFlen", help="Maximum length for reference allele", type=int)
    ap.add_argument("--ploidy", help="Ploidy for each sample", type=int, default=2)
    ap.add_argument("--ploidyFile", help="File with samples names and ploidy as columns")
    ap.add_argument("--ploidyMismatchToMissing", help="Set genotypes with mismatched ploidy to missing", action="store_true")
    ap.add_argument("--keepPartial", help="Keep genotypes where some but not all alleles are missing", action="store_true")
    ap.add_argument("--addRefTrack", help="Add a third column with the header REF and the reference allele", action="store_true")
    ap.add_argument("--noHeader", help="Output without header line", action="store_true")
    ap.add_argument("--field", help="Optional - format field to extract instead of genotypes")
    ap.add_argument("--missing", help="Value to use for missing data (one character)")
    ap.add_argument("--outSep", help="Output separator", default="\t")
    ap.add_argument("-i", "--inFile", help="Input vcf file")
    ap.add_argument("--packed", metavar="FILE.pgeno", help="also (or, without -o, only) write the packed form the engine's drivers read")
    ap.add_argument("--packedCodec", choices=("zlib", "none"), default="zlib",
                    help="cells of the --packed file: deflated chunks (smallest) or raw (1 byte per genotype: read by the drivers at PCIe speed)")
    ap.add_argument("--threads", type=int, default=0, help="host threads of the native parser (default: all)")
    return ap


def parse_vcf_main(argv=None):
    args = make_parser().parse_args(argv)
    if args.field is not None:
        return _field_main(args)
    if args.simplifyALT or args.expandMulti:
        return _cigar_main(args)
    missing = args.missing if args.missing is not None else "N"
    if len(missing) != 1 or len(args.outSep) != 1:
        # cells of varying width (`NA/NA`): the line loop of --simplifyALT writes them (text only); the native parser and the
        # packed format are for one-character cells
        if args.packed:
            raise SystemExit("parseVCF.py: --packed needs --missing and --outSep of one character")
        return _cigar_main(args)
    want_text = bool(args.outFile) or not args.packed
    if args.packed and args.addRefTrack and not want_text:
        raise SystemExit("parseVCF.py: --addRefTrack has no meaning for --packed output")

    # ---- header (parseHeaderLines, parseVCF.py:213-236) ----
    reader = genoio.BlockReader(args.inFile)
    head = None
    while True:
        line = reader.read_header()
        if not line:
            break
        if line.startswith(b"#CHROM"):
            head = line.decode("utf-8", "replace").split()
            break
    assert head is not None and len(head) >= 9, "no #CHROM header line in the VCF"
    plan = Plan(args, head[9:])
    samples, n_sel, pl, missing = plan.samples, plan.n_sel, plan.pl, plan.missing
    vcf_samples = plan.vcf_samples

    L = _lib.lib()
    out = _open_out(args.outFile) if want_text else None
    sep = args.outSep.encode()
    if out is not None and not args.noHeader:
        out.write(plan.header_line())
    packer = genoio.PackedWriter(args.packed, samples, [int(p) for p in pl], args.packedCodec) if args.packed else None
    lut = np.zeros(256, dtype=np.uint8)
    for ch, code in zip(b"ACGT", (1, 2, 4, 8)):
        lut[ch] = code
    block_bytes = int(os.environ.get("PG_STREAM_BYTES", 128 << 20))
    timing = {} if os.environ.get("PG_TIMING") else None
    t_start = time.perf_counter()

    def lap(name, t0):
        t1 = time.perf_counter()
        if timing is not None:
            timing[name] = timing.get(name, 0.0) + t1 - t0
        return t1

    # three steps side by side: a reader thread (the next blocks: file pages, or BGZF members inflated by the library's host threads),
    # this thread (pg_encode_vcf + pg_vcf_render_rows, both on host threads) and a writer thread (BGZF deflate + write)
    sink = _AsyncOut(out) if out is not None else None
    state = {"prev_chrom": None, "prev_pos": None, "n_multibase": 0}

    def host_block(body):
        """a block of text through the host parser: pg_encode_vcf + pg_vcf_render_rows on the host threads (+ the packed writer)"""
        t0 = time.perf_counter()
        ptr, nbytes, keep = _lib.text_ptr(body)
        k, nmb, A = plan.host_parse(ptr, nbytes, state["prev_chrom"], state["prev_pos"], int(args.threads))
        t0 = lap("parse_s", t0)
        state["n_multibase"] += nmb
        if args.excludeDuplicates:
            pc, pp = _last_key(body)
            if pc is not None:
                state["prev_chrom"], state["prev_pos"] = pc, pp
        if k == 0:
            return
        if sink is not None:
            text = plan.host_render(ptr, k, A, int(args.threads))
            t0 = lap("render_s", t0)
            sink.write(text)
            t0 = lap("wait_for_writer_s", t0)
        if packer is not None:
            chars, pos, coff, clen = A["chars"], A["pos"], A["coff"], A["clen"]
            # scaffold runs of the kept rows (names are read once per run)
            starts = np.zeros(k, dtype=np.int64)
            nr = C.c_int64(0)
            check(L.pg_scaffold_runs(ptr, coff, clen, k, starts, k, C.byref(nr)))
            starts = starts[:nr.value]
            run_names = [bytes(body[int(coff[i]):int(coff[i]) + int(clen[i])]) for i in starts]
            cells = lut[chars[:k, 0::2]] | (lut[chars[:k, 1::2]] << 4)
            packer.write_block(genoio.GenoData(None, pos[:k].copy(), starts.copy(), [nm.decode("utf-8", "replace") for nm in run_names]), cells)
            lap("pack_s", t0)

    def finish(pending):
        """the rows of a block the device parsed -> the writer; a block with a line the device does not take -> the host parser"""
        slot, raw, _keep = pending
        t0 = time.perf_counter()
        eng = _text_blocks.engine
        nbytes, _rows, line = eng.vcf_collect(slot)
        t0 = lap("device_wait_s", t0)
        if line < 0:
            if nbytes and gz_rows:
                members = eng.vcf_rows_bgzf(slot, eng.vcf_bgzf_bytes)
                t0 = lap("device_rows_s", t0)
                sink.write(("members", members))
                lap("wait_for_writer_s", t0)
            elif nbytes:
                text = eng.vcf_rows(slot, nbytes)
                t0 = lap("device_rows_s", t0)
                sink.write(text)
                lap("wait_for_writer_s", t0)
            return
        if timing is not None:
            timing["blocks_handed_to_the_host_parser"] = timing.get("blocks_handed_to_the_host_parser", 0) + 1
            timing.setdefault("first_line_handed_over", int(line))
        if args.excludeDuplicates:                               # (the data line before this block: the device carried it)
            key = eng.vcf_prev(slot)
            if key is not None:                                  # (None: the block before was the host's for its last line -- its key is in `state`)
                state["prev_chrom"], state["prev_pos"] = key
        host_block(eng.vcf_text(slot, len(raw)) if isinstance(raw, genoio.BgzfSpan) else raw)

    # the device's parser takes text output without --packed (rows as text are what it makes)
    dev_plan = plan if (sink is not None and packer is None) else None
    # `-o out.geno.gz`: the rows are deflated where they are made (PG_DEFLATE_DEVICE=0: by the host threads of genoio.BgzfWriter)
    gz_rows = dev_plan is not None and isinstance(out, genoio.BgzfWriter) and os.environ.get("PG_DEFLATE_DEVICE", "1") != "0"
    t0 = time.perf_counter()
    pending, slot = None, 0
    try:
        for body, where in _read_ahead(_text_blocks(reader, block_bytes, int(args.threads), dev_plan, gz_rows)):
            t0 = lap("wait_for_block_s", t0)
            if where is None:
                if pending is not None:
                    finish(pending)
                    pending = None
                host_block(body)
                state["prev_on_device"] = False
            else:
                # parse(k) is queued behind submit(k): while this thread waits for block k - 1 and hands its rows on, block k is
                # copied / inflated; the kernels of k are queued as soon as its line count is back
                eng = _text_blocks.engine
                if args.excludeDuplicates and not state.get("prev_on_device"):
                    eng.vcf_set_prev(state["prev_chrom"], state["prev_pos"])        # (of the blocks the host parsed so far)
                    state["prev_on_device"] = True
                keep = eng.vcf_submit(slot, body, where or None)
                t0 = lap("device_submit_s", t0)
                if pending is not None:
                    finish(pending)
                t0 = time.perf_counter()
                eng.vcf_parse(slot)
                lap("device_parse_queue_s", t0)
                pending = (slot, body, keep)
                slot ^= 1
            del body
            t0 = time.perf_counter()
        if pending is not None:
            finish(pending)
            pending = None
        n_multibase_total = state["n_multibase"]
    except BaseException:
        if sink is not None:                    # (the writer thread ends, whatever it had queued is dropped with the failed run)
            sink.abort()
        if out is not None and out is not sys.stdout.buffer:
            try:
                (out.abort if hasattr(out, "abort") else out.close)()       # (BGZF: no end-of-file member behind a truncated output)
            except Exception:
                pass
        raise
    if sink is not None:
        t0 = time.perf_counter()
        sink.close()
        lap("wait_for_writer_s", t0)
    if out is not None and out is not sys.stdout.buffer:
        out.close()
    elif out is not None:
        out.flush()
    if packer is not None:
        packer.close()
        if n_multibase_total:
            sys.stderr.write("%d allele calls longer than one base were stored as missing\n" % n_multibase_total)
    reader.close()
    if timing is not None:
        timing["total_s"] = time.perf_counter() - t_start
        try:                                                     # seconds since the process was started (interpreter, imports, header read, the run)
            with open("/proc/self/stat") as f:
                ticks = int(f.read().rsplit(")", 1)[1].split()[19])
            with open("/proc/uptime") as f:
                timing["since_process_start_s"] = float(f.read().split()[0]) - ticks / os.sysconf("SC_CLK_TCK")
        except Exception:
            pass
        timing.update(getattr(_text_blocks, "last_info", {}))
        sys.stderr.write("PG_TIMING " + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in timing.items()}) + "\n")
    return 0
