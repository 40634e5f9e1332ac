#!/usr/bin/env python
"""Audit of the hand-scheduled pair kernels (pg_pair_tile.hip): between a kernel's PG_AUDIT_BEGIN and PG_AUDIT_END markers (the
accumulation loop of the waves that hold accumulators; the drain `s_nop 11` carries the end marker), no
compiler-generated instruction may touch an accumulator register (a register written by a matrix instruction inside the asm
statements) -- hipcc pads no hazards around inline asm, so a copy of a fresh matrix result would read garbage on some launches.
Usage: audit_pair_tile_asm.py <pg_pair_tile-hip-amdgcn-amd-amdhsa-gfx950.s>   (from hipcc -save-temps)"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def main(path):
    lines = open(path).read().splitlines()
    bad = 0
    kernels = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*k_pair[CD]_tile.*:\s*(;.*)?$", l)]
    for k0 in kernels:
        end = next(i for i in range(k0, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = lines[k0:end]
        try:
            b0 = max(i for i, l in enumerate(body) if "PG_AUDIT_BEGIN" in l)
            b1 = max(i for i, l in enumerate(body) if "PG_AUDIT_END" in l)
        except (StopIteration, ValueError):
            print("no loop found in", lines[k0])
            bad += 1
            continue
        acc, inside = set(), False
        for l in body[b0:b1]:
            if "#ASMSTART" in l:
                inside = True
            elif "#ASMEND" in l:
                inside = False
            elif inside and "v_mfma" in l:
                acc |= regs(l.split(",")[0])
        # the block that zeroes the accumulators may be laid out behind the loop: basic blocks that hold a `v_mov vN, 0` of an
        # accumulator are initialisation, not readers of a result
        init, blk = set(), 0
        for i, l in enumerate(body[b0:b1]):
            if re.match(r"^\.LBB\d+_\d+:", l):
                blk = i
            m = re.match(r"\s*v_mov_b32_e32 v(\d+), 0\s*$", l.split(";")[0])
            if m and int(m.group(1)) in acc:
                init.add(blk)
        inside = False
        n, blk = 0, 0
        for i, l in enumerate(body[b0:b1]):
            if re.match(r"^\.LBB\d+_\d+:", l):
                blk = i
            if blk in init:
                continue
            if "#ASMSTART" in l:
                inside = True
            elif "#ASMEND" in l:
                inside = False
            elif not inside:
                code = l.split(";")[0]
                if code.strip() and not code.strip().endswith(":") and regs(code) & acc:
                    print("compiler instruction touches an accumulator:", l.strip())
                    n += 1
        print("%s: %d accumulator registers, %d offending instructions" % (lines[k0].split(":")[0][:60], len(acc), n))
        bad += n
    return 1 if bad or not kernels else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
