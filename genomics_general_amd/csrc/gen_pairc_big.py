#!/usr/bin/env python
"""Generates pg_pairc_big.inc: the main loops of k_pairC_big (gfx950 assembly, inline-asm operand syntax), one per plane width.

k_pairC_big computes the called counts of a window part (C = V V^T, exact MX fp4 products, see pg_pair_tile.hip) with ONE wave per
SIMD that owns up to 14 tiles of 32 x 32 and keeps their accumulators in the accumulator half of the register file: a fragment it
expands (7 VALU operations) then feeds up to 7 matrix instructions, and the whole loop -- LDS-DMA of the plane, fragment reads,
expansions, products -- is one hand-scheduled instruction stream, because the matrix pipe of a SIMD with a single wave is only
busy while that wave's next instruction is a product whose operands are ready.

Block = W waves (W = 1 for up to 10 tiles, 2 beyond) sharing a ring of NR pairs of word groups in LDS:
  * every wave copies the tile rows t = wave, wave + W, ... of a pair with `global_load_lds_dwordx4` (lanes 0..31 group 2p, lanes
    32..63 group 2p+1 of one tile row: 1 KiB in fragment order), NR pairs ahead; counted vmcnt + s_barrier (W = 2) hand a pair over;
  * RAW[2]: the 16-byte raw words (four K steps) of the wave's fragments for the current and the next pair (ds_read_b128);
  * FRAG[2]: the expanded fragments of the current and the next K step: dword m = (word >> m) & M, M = 0x11111111 or 0 for a
    lane half whose group lies beyond the part (both operands of a product in this one form: a product is 0.25, the
    accumulator count / 4);
  * K step k: the wave's products on FRAG[k & 1], interleaved with the expansion of step k+1 into FRAG[~k & 1] (of the next
    pair's word 0 at k = 3) and with the loop's bookkeeping (copy of pair r + NR in steps 0 - 2, reads of pair r + 2 in step 3:
    the only instructions outside the shadow of a product are one s_waitcnt and one s_barrier per pair); a VALU-written
    register is read by a product no sooner than two instructions later.
The loop is unrolled twice (RAW buffers swap roles).  Pairs past the end of the part are computed with M = 0 and copied from the
last real pair (the counted waits stay literal).

Operands of a generated statement (see k_pairC_big in pg_pair_big.hip):
    %[a0] .. %[a13]  (+a)  accumulators (v16f in AGPRs), tile n of the wave's list
    %[ga]    (v, 64-bit)   this lane's word of tile row 0, pair 0:  Vp + ((q0 + kb) * NPv + r) * 16
    %[lrd]   (v)           LDS byte address of the ring + 16 * lane
    %[kb]    (v)           lane >> 5
    %[lds]   (s)           LDS byte address of the ring
    %[stride] (s)          bytes per pair of groups (2 * NPv * 16)
    %[npair] (s)           pairs of the part;  %[qrem] (s): groups of the part;  %[wave] (s)
Fixed registers (clobbers): v[100:223], s[60:75], vcc, scc, m0 is saved and restored.

    python gen_pairc_big.py > pg_pairc_big.inc
"""
import os
import re

NR = 8                                    # ring depth in pairs
RAW = (100, 128)                          # 7 fragments x 4 words each
FRAG = (156, 184)                         # 7 fragments x 4 dwords each
VT0, VT1, VMA, VMB, VK1, VLR, VGA, VKB, VL16, VT2, VT3 = 212, 213, 214, 215, 216, 217, 218, 220, 221, 222, 223     # VGA is a pair (218:219)
S_SAVE, S_SLOTRD, S_SLOTWR, S_PISS, S_P2A, S_P2B, S_TRIPS, S_T, S_RING = 60, 61, 62, 63, 64, 65, 66, 67, 68
MFMA = "v_mfma_f32_32x32x64_f8f6f4 %s, v[%d:%d], v[%d:%d], %s cbsz:4 blgp:4"


def programs(T):
    """tile lists of the waves: the upper triangle row-major, cut into runs of at most 14"""
    tiles = [(i, j) for i in range(T) for j in range(i, T)]
    W = 1 if len(tiles) <= 14 else 2
    per = (len(tiles) + W - 1) // W
    return [tiles[w * per:(w + 1) * per] for w in range(W)]


def gen_wave(T, W, wave, tiles):
    frags = sorted(set(t for ij in tiles for t in ij))
    fpos = {f: k for k, f in enumerate(frags)}            # place of fragment f in the RAW / FRAG buffers
    mine = [t for t in range(T) if t % W == wave]         # tile rows this wave copies
    nm = len(mine)
    slot_bytes = T * 1024
    L = []

    def raw(buf, f, k):
        return RAW[buf] + 4 * fpos[f] + k

    def frag(buf, f):
        return FRAG[buf] + 4 * fpos[f]

    def expand(src_buf, k, dst_buf, vm):
        """VALU ops: FRAG[dst_buf][f] <- word k of RAW[src_buf][f], all fragments (no operation reads its predecessor's result)"""
        ops = []
        for f in frags:
            w, d = raw(src_buf, f, k), frag(dst_buf, f)
            ops += ["v_lshrrev_b32 v%d, 1, v%d" % (VT0, w), "v_lshrrev_b32 v%d, 2, v%d" % (VT1, w), "v_lshrrev_b32 v%d, 3, v%d" % (VT2, w),
                    "v_and_b32 v%d, v%d, v%d" % (d, w, vm), "v_and_b32 v%d, v%d, v%d" % (d + 1, VT0, vm),
                    "v_and_b32 v%d, v%d, v%d" % (d + 2, VT1, vm), "v_and_b32 v%d, v%d, v%d" % (d + 3, VT2, vm)]
        return ops

    def products(buf):
        out = []
        for n, (i, j) in enumerate(tiles):
            a = "%%[a%d]" % n
            out.append(MFMA % (a, frag(buf, i), frag(buf, i) + 3, frag(buf, j), frag(buf, j) + 3, a))
        return out

    def merge(a, b):
        """b spread evenly through a, both in their own order"""
        out, nb = [], 0
        for k, ins in enumerate(a):
            out.append(ins)
            want = len(b) * (k + 1) // len(a)
            out += b[nb:want]
            nb = want
        return out + b[nb:]

    def step(buf, valu, extra=()):
        """the products on FRAG[buf] with the VALU ops `valu` (and the bookkeeping `extra`) spread between them"""
        ms = products(buf)
        oth = merge(valu, list(extra))
        out = []
        for m, ins in enumerate(ms):
            out.append(ins)
            out += oth[len(oth) * m // len(ms):len(oth) * (m + 1) // len(ms)]
        return out + ["s_nop 1"]

    def copy_pair():
        """this wave's copies of pair S_PISS (clamped to the last real pair) into ring slot S_SLOTWR, then both advance"""
        out = ["s_add_u32 s%d, s%d, s%d" % (S_T, S_RING, S_SLOTWR)]
        for t in mine:
            # (the instruction offset moves the LDS side of the copy as well as the global side: row t lands at M0 + 512 t + 16 lane)
            out += ["s_add_u32 m0, s%d, %d" % (S_T, t * 512), "s_nop 0",
                    "global_load_lds_dwordx4 v[%d:%d], off offset:%d" % (VGA, VGA + 1, 512 * t)]
        out += ["s_add_u32 s%d, s%d, 1" % (S_PISS, S_PISS),
                "s_cmp_lt_u32 s%d, %%[npair]" % S_PISS,                     # the next pair to copy exists: move on, else stay
                "s_cselect_b32 s%d, %%[stride], 0" % S_T,
                "v_add_co_u32 v%d, vcc, s%d, v%d" % (VGA, S_T, VGA),
                "v_addc_co_u32 v%d, vcc, 0, v%d, vcc" % (VGA + 1, VGA + 1),
                "s_add_u32 s%d, s%d, %d" % (S_SLOTWR, S_SLOTWR, slot_bytes),
                "s_cmp_ge_u32 s%d, %d" % (S_SLOTWR, NR * slot_bytes),
                "s_cselect_b32 s%d, 0, s%d" % (S_SLOTWR, S_SLOTWR)]
        return out

    def read_pair(buf, vm, sp2):
        """RAW[buf] <- the ring slot S_SLOTRD (advanced), its mask: group 2p + kb of the part exists"""
        out = ["v_add_u32 v%d, s%d, v%d" % (VLR, S_SLOTRD, VL16)]
        for f in frags:
            out.append("ds_read_b128 v[%d:%d], v%d offset:%d" % (raw(buf, f, 0), raw(buf, f, 3), VLR, f * 1024))
        out += ["v_add_u32 v%d, s%d, v%d" % (VT3, sp2, VKB),
                "v_cmp_gt_u32 vcc, %%[qrem], v%d" % VT3,
                "v_cndmask_b32 v%d, 0, v%d, vcc" % (vm, VK1),
                "s_add_u32 s%d, s%d, 4" % (sp2, sp2),
                "s_add_u32 s%d, s%d, %d" % (S_SLOTRD, S_SLOTRD, slot_bytes),
                "s_cmp_ge_u32 s%d, %d" % (S_SLOTRD, NR * slot_bytes),
                "s_cselect_b32 s%d, 0, s%d" % (S_SLOTRD, S_SLOTRD)]
        return out

    def handover(n_in_flight, lgkm=False):
        """the own copies of the next pair to read have landed (n_in_flight later pairs stay in flight); everybody's have"""
        out = ["s_waitcnt vmcnt(%d)%s" % (n_in_flight * nm, " lgkmcnt(0)" if lgkm else "")]
        if W > 1:
            out.append("s_barrier")
        return out

    def half(cur, oth, vm_cur, vm_oth, sp2_cur):
        """Pair r: its four K steps from RAW[cur]; RAW[oth] holds pair r + 1 (read during the previous half's last step).  The copy
        of pair r + NR goes into the slot pair r came from -- every wave of the block finished reading that slot before the
        previous half's barrier --, its instructions spread through the first three steps; then one synchronisation point (RAW[oth]
        has arrived, the own copies of pair r + 2 have landed, everybody's have), and the reads of pair r + 2 into RAW[cur]
        -- free since the expansions of step 2 -- spread through the last step."""
        cp = copy_pair()
        c1, c2 = len(cp) // 3, 2 * len(cp) // 3
        # (the order of the copy's instructions is kept, and nothing else writes M0 or vcc in these steps)
        out = []
        out += step(0, expand(cur, 1, 1, vm_cur), cp[:c1])
        out += step(1, expand(cur, 2, 0, vm_cur), cp[c1:c2])
        out += step(0, expand(cur, 3, 1, vm_cur), cp[c2:])
        out += handover(NR - 2, lgkm=True)
        out += step(1, expand(oth, 0, 0, vm_oth), read_pair(cur, vm_cur, sp2_cur))
        return out

    # ---- prologue ----
    L += ["s_mov_b32 s%d, m0" % S_SAVE,
          "v_mov_b32 v%d, %%[lrd]" % VL16, "v_mov_b32 v%d, %%[kb]" % VKB, "v_mov_b32 v%d, 0x11111111" % VK1,
          "v_lshrrev_b64 v[%d:%d], 0, %%[ga]" % (VGA, VGA + 1),
          "s_mov_b32 s%d, %%[lds]" % S_RING, "s_mov_b32 s%d, 0" % S_SLOTRD, "s_mov_b32 s%d, 0" % S_SLOTWR, "s_mov_b32 s%d, 0" % S_PISS,
          "s_mov_b32 s%d, 0" % S_P2A, "s_mov_b32 s%d, 2" % S_P2B,
          "s_add_u32 s%d, %%[npair], 1" % S_TRIPS, "s_lshr_b32 s%d, s%d, 1" % (S_TRIPS, S_TRIPS)]
    for _ in range(NR):
        L += copy_pair()
    L += handover(NR - 1) + read_pair(0, VMA, S_P2A) + ["s_waitcnt lgkmcnt(0)"] + expand(0, 0, 0, VMA) + ["s_nop 1"]
    L += handover(NR - 2) + read_pair(1, VMB, S_P2B)
    # (from here on half r issues the copies of pair r + NR before its hand-over of pair r + 2: NR - 2 younger pairs in flight)
    L += ["1:"]
    body = half(0, 1, VMA, VMB, S_P2A) + half(1, 0, VMB, VMA, S_P2B)
    var = os.environ.get("PG_CBIG_VARIANT", "")          # timing experiments only (wrong results): drop a class of instructions
    drop = {"B": "s_barrier", "C": "global_load_lds", "R": "ds_read", "M": "v_mfma"}
    for k, pat in drop.items():
        if k in var:
            body = [i for i in body if pat not in i]
    if "X" in var:
        body = [i for i in body if not ((i.startswith("v_and_b32") or i.startswith("v_lshrrev_b32")))]
    if "V" in var:
        body = [re.sub(r"vmcnt\(\d+\) ", "", i) for i in body]
    L += body
    L += ["s_sub_u32 s%d, s%d, 1" % (S_TRIPS, S_TRIPS), "s_cmp_lg_u32 s%d, 0" % S_TRIPS, "s_cbranch_scc1 1b"]
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_mov_b32 m0, s%d" % S_SAVE, "s_nop 11"]
    return L


def emit(T):
    progs = programs(T)
    W = len(progs)
    L = []
    if W == 1:
        L = gen_wave(T, 1, 0, progs[0])
    else:
        w0, w1 = gen_wave(T, 2, 0, progs[0]), gen_wave(T, 2, 1, progs[1])
        # the loop label `1:` is local to each program; the outer labels are numbered apart
        L = ["s_cmp_lg_u32 %[wave], 0", "s_cbranch_scc1 8f"] + w0 + ["s_branch 9f", "8:"] + w1 + ["9:"]
    print("#define PG_CBIG_ASM_T%d \\" % T)
    for ln in L:
        print('    "%s\\n\\t" \\' % ln)
    print('    ""')
    print("#define PG_CBIG_WAVES_T%d %d" % (T, W))
    print("#define PG_CBIG_TILES_T%d { %s }" % (T, ", ".join("{ %s }" % ", ".join("{%d, %d}" % ij for ij in (p + [(-1, -1)] * (14 - len(p))))
                                                              for p in (progs + [[]])[:2])))


if __name__ == "__main__":
    print("// generated by gen_pairc_big.py -- do not edit")
    for T in range(1, 8):
        emit(T)
    clob = ["v%d" % k for k in range(RAW[0], VT3 + 1)] + ["s%d" % k for k in range(S_SAVE, S_RING + 1)] + ["vcc", "scc", "memory"]
    print("#define PG_CBIG_CLOBBERS " + ", ".join('"%s"' % c for c in clob))
    print("#define PG_CBIG_RING_PAIRS %d" % NR)
