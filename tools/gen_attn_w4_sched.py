#!/usr/bin/env python3
"""Emit videocof_amd/csrc/attn_w4_sched<PIPE>.inc: the hand-placed instruction schedule of one KV-tile interval of
attn_fwd_w4_kernel (4 waves x 64 query rows, one wave per SIMD).

One interval = 64 MFMA slots: 32 x S(t+1) = K(t+1).Q^T (segment B), then 32 x O += V^T(t).P(t) (segment C).  Every slot is
`MFMA ; sched_barrier ; fillers ; sched_barrier`, so hipcc keeps the placement (CDNA4 guide: with one wave per SIMD about
5 single-issue fillers hide under a 32-cycle MFMA, and the SAME multiset re-ordered is measurably slower).

Fillers per interval: 16 K-fragment + 16 V^T-fragment ds_read_b128 (each fragment feeds the two query blocks of the wave),
the 8 LDS-DMA pieces this wave stages for later tiles, and 160 softmax micro-ops (64 v_exp_f32, 64 row-sum adds,
32 v_cvt_pk_bf16_f32) in 8 groups of (8 exp, 8 add, 4 cvt), one group per P fragment.

The softmax stream is SHIFTED by half an interval so that every slot carries the same 2.5 micro-ops: segment B finishes
the current tile (P fragments tt = 2, 3 of S(t)), segment C already starts the NEXT tile (tt = 0, 1 of S(t+1), whose
accumulate chains end in slots B28 / B29: three MFMA slots and the barrier -- far more than the 12 wait states an 8-pass
MFMA result needs -- before the first v_exp_f32 that reads them in C00).  A P fragment is therefore always complete >= 1
slot before the PV step that reads it (asserted below), and all 160 micro-ops of a tile run in one continuous stream of
2.5 per slot from C00 of the previous interval to B31 of its own.

The macros (QK, PV, RDK, RDKN, RDV, G, E, A, C, MIDFENCE, MIDCHECK, SB) are defined next to the kernel in attn_fwd.hip.
The 8 DMA pieces go into segment B (every 4th slot) and are waited for at the fence that ends the interval.
Tunables: PDK / PDV = fragments kept in flight ahead of their first use; BAL = 1 gives the first slot of a fragment pair
(which carries no ds_read) 3 softmax micro-ops and the second 2, BAL = 0 the other way round.
usage: gen_attn_w4_sched.py PDK PDV BAL [PK [Q8]] > attn_w4_sched.inc

Q8 = 1 (attn_w4_sched_q8.inc): the QK^T product runs on the fp8 matrix pipe -- segment B is 8 slots of one 16-pass
v_mfma_scale_f32_32x32x64_f8f6f4 each (chains (qb, kt) of two links: d halves dh = 0, 1) instead of 32 8-pass slots; the softmax
stream is split 40 : 120 between the segments (5 micro-ops per B slot, 3.75 per C slot): tt = 3 of the tile in B, tt = 0, 1, 2 of the
next tile in C; K fragment f = 2 dh + kt is TWO ds_read_b128 (32 fp8 per lane) into its
own registers (RDK8), the wave stages 2 K pieces + 4 V^T pieces (G8(0..5)).  Segment C is unchanged.
"""
import sys

PIPE, PDK, PDV, BAL, PK = 0, 3, 3, 1, 0
if len(sys.argv) > 1:
    PDK, PDV, BAL = (int(v) for v in sys.argv[1:4])
Q8 = 0
if len(sys.argv) > 5:
    Q8 = int(sys.argv[5])
if len(sys.argv) > 4:
    PK = int(sys.argv[4])      # 1: the packed-shift form -- 4 more micro-ops per group (D = v_pk_fma_f32 on a score pair), 3 per slot


def take(ops, s):
    """micro-ops of slot s (0..31) out of an 80-op segment stream: 5 per slot pair, split 3 + 2 or 2 + 3
    (PK: 96 ops, 3 per slot).  BAL = 2 balances COST instead of counts: a v_exp_f32 costs 1.6 adds (tools/probe/valu_rates.hip), and
    the count-balanced stream puts three of them under one 32-cycle MFMA and none under the next -- here exactly ONE exponential per slot,
    its add one slot later, a pack every other slot.  Measured: no change for the power-limited bf16 kernel
    (profiles/r03/attn_cycle_balanced_sched_ab.log); the product files are BAL = 0."""
    if BAL == 3 and not PK and not Q8:      # as BAL = 2 with the packs on the odd slots (which also carry the fragment read)
        es = [o for o in ops if o.startswith("E(")]
        as_ = [o for o in ops if o.startswith("A(")]
        cs = [o for o in ops if o.startswith("C(")]
        line = [es[s]] + ([as_[s - 1]] if s >= 1 else []) + ([cs[(s - 3) // 2]] if s >= 3 and s % 2 == 1 else [])
        if s == 31:
            line += [as_[31], cs[15]]
        return line
    if BAL == 2 and not PK and not Q8:
        es = [o for o in ops if o.startswith("E(")]
        as_ = [o for o in ops if o.startswith("A(")]
        cs = [o for o in ops if o.startswith("C(")]
        assert len(es) == 32 and len(as_) == 32 and len(cs) == 16
        # group order of E / A / C is the same, C_k packs E_2k, E_2k+1
        line = [es[s]] + ([as_[s - 1]] if s >= 1 else []) + ([cs[(s - 2) // 2]] if s >= 2 and s % 2 == 0 else [])
        if s == 31:
            line += [as_[31], cs[15]]
        return line
    if PK:
        return ops[3 * s:3 * s + 3]
    a = (s >> 1) * 5
    first = 3 if BAL else 2
    return ops[a:a + first] if s % 2 == 0 else ops[a + first:a + 5]


def group(tt, qb):
    b = qb * 32 + tt * 8           # first score index of the group
    w = qb * 16 + tt * 4           # first packed word of P[qb][tt]
    E = lambda j: f"E({b + j})"
    A = lambda j: f"A({b + j})"
    C = lambda j: f"C({w + j})"
    if PK:
        # D(p): scores 2p, 2p + 1 -> exponent arguments (one packed fma); every D is >= 1 op ahead of the first E that reads it
        # (a VALU result feeding a transcendental the very next instruction costs a wait state)
        D = lambda k: f"D({(b + 2 * k) // 2})"
        return [D(0), D(1), E(0), E(1), E(2), E(3), D(2), A(0), A(1), E(4), C(0), A(2), E(5), D(3), C(1), A(3), E(6), A(4),
                E(7), A(5), C(2), A(6), A(7), C(3)]
    return [E(0), E(1), E(2), E(3), A(0), A(1), E(4), C(0), A(2), E(5), C(1), A(3), E(6), A(4), E(7), A(5), C(2),
            A(6), A(7), C(3)]


if Q8:
    # segment B is 8 slots x 64 cycles, segment C 32 x 32: the stream is split 40 : 120 like the MFMA time -- only tt = 3 of the
    # current tile in B, tt = 0, 1, 2 of the NEXT tile in C (its tt = 2 scores sit in the kt = 1 chains that end in B06 / B07)
    ops_b = sum((group(3, qb) for qb in (0, 1)), [])
    ops_c = sum((group(tt, qb) for tt in (0, 1, 2) for qb in (0, 1)), [])
    assert len(ops_b) == 40 and len(ops_c) == 120
else:
    ops_b = sum((group(tt, qb) for tt in (2, 3) for qb in (0, 1)), [])      # current tile, key half kt = 1
    ops_c = sum((group(tt, qb) for tt in (0, 1) for qb in (0, 1)), [])      # next tile, key half kt = 0
    assert len(ops_b) == len(ops_c) == (96 if PK else 80)

out = [f"// generated by tools/gen_attn_w4_sched.py {PDK} {PDV} {BAL}{' 1' if PK else (' 0' if Q8 else '')}{' 1' if Q8 else ''} -- do not edit by hand"]
if Q8:
    assert not PK and PIPE == 0
    out.append("// ---- top of the interval: the first two K fragments (their latency is exposed once per interval)")
    out.append("RDK8(0); RDK8(1); SB();")
elif PIPE == 0:
    out.append("// ---- top of the interval: first K fragments (their latency is exposed once per interval)")
    out.append(" ".join(f"RDK({f});" for f in range(PDK + 1)) + " SB();")
else:
    out.append("// ---- the first K fragments of this interval were requested at the end of the previous one")
g = 0
for s in range(8 if Q8 else 0):                          # ---- segment B, fp8 form: 8 x 16-pass MFMA, chains (qb, kt) of 2 links
    dh, kt, qb = s >> 2, (s >> 1) & 1, s & 1
    f = s >> 1
    line = [f"QK8({qb},{kt},{dh},{f});"]
    if s in (0, 2):
        line.append(f"RDK8({2 + (s >> 1)});")
    if s < 6:
        line.append(f"G8({s});")
    if s >= 8 - (PDV + 1):
        line.append(f"RDV({s - (8 - (PDV + 1))});")
    line += [o + ";" for o in ops_b[5 * s:5 * s + 5]]
    out.append(" ".join(line) + " SB();" + f"   // B{s:02d}")
for s in range(0 if Q8 else 32):                         # ---- segment B: S(t+1); 4 accumulate chains (qb, kt) interleaved
    ks, kt, qb = s >> 2, (s >> 1) & 1, s & 1
    f = s >> 1                                           # K fragment index = ks * 2 + kt
    line = [f"QK({qb},{kt},{ks},{f});"]
    if qb == 1 and f + PDK + 1 <= 15:
        line.append(f"RDK({f + PDK + 1});")
    if PIPE == 0 and s % 4 == 0:
        line.append(f"G({g});")
        g += 1
    if s >= 32 - (PDV + 1):
        line.append(f"RDV({s - (32 - (PDV + 1))});")
    line += [o + ";" for o in take(ops_b, s)]
    out.append(" ".join(line) + " SB();" + f"   // B{s:02d}")
if PIPE == 1:
    out.append("MIDFENCE(); SB();   // K(t+2), V(t+1) landed and visible; every wave is done reading K(t+1) and V(t-1)")
# all four P fragments and the row sums of the current tile are complete here and none has been consumed by a PV step yet:
# the place where the kernel checks that the tile stayed inside the window of its softmax reference (and repairs it if not)
out.append("MIDCHECK(); SB();")
for s in range(32):                                      # ---- segment C: O += V^T(t).P(t)
    tt, dt, qb = s >> 3, (s >> 1) & 3, s & 1
    f = s >> 1
    line = [f"PV({qb},{dt},{tt},{f});"]
    if qb == 1 and f + PDV + 1 <= 15:
        line.append(f"RDV({f + PDV + 1});")
    if PIPE == 1 and s % 4 == 0:
        line.append(f"G({g});")
        g += 1
    if PIPE == 1 and s >= 32 - (PDK + 1):
        line.append(f"RDKN({s - (32 - (PDK + 1))});")
    if Q8:
        line += [o + ";" for o in ops_c[(120 * s) // 32:(120 * (s + 1)) // 32]]
    else:
        line += [o + ";" for o in take(ops_c, s)]
    out.append(" ".join(line) + " SB();" + f"   // C{s:02d}")
assert g == (0 if Q8 else 8)
# ---- s_waitcnt lgkmcnt counts for the kernel form whose ds_reads are inline asm (fragments land in AGPRs; hipcc does not
# track them): LDS reads return in order, so the first MFMA that uses fragment X waits until at most N reads are still in
# flight, N = the reads issued after X's.  QK / PV get the count as a 5th argument (-1 = the fragment's second use, no wait).
import re
issued, waits = [], []
new_out = []
for line in out:
    if line.startswith("//") or "MIDCHECK" in line or "MIDFENCE" in line:
        new_out.append(line)
        continue
    toks = re.findall(r"(RDK8|RDK|RDV|QK8|QK|PV)\(([^)]*)\);", line)
    for name, args in toks:
        if name in ("RDK", "RDV"):
            issued.append((name[2], int(args)))
        elif name == "RDK8":
            issued += [("K", int(args))] * 2             # two ds_read_b128 per fp8 fragment
        else:
            a = [int(v) for v in args.split(",")]
            qb, f = a[0], a[3]
            kind = "K" if name in ("QK", "QK8") else "V"
            if qb == 0:
                idx = max(i for i, e in enumerate(issued) if e == (kind, f))
                w = len(issued) - 1 - idx
            else:
                w = -1
            line = line.replace(f"{name}({args});", f"{name}({args},{w});", 1)
    new_out.append(line)
out = new_out
text = "\n".join(out) + "\n"
# P fragments tt = 2, 3 of the current tile are complete at least one slot before the PV step that reads them
for tt in ((3,) if Q8 else (2, 3)):           # (Q8: tt = 2 was produced in the previous interval's segment C)
    for qb in range(2):
        last_c = f"C({qb * 16 + tt * 4 + 3});"
        first_pv = f"PV(0,0,{tt},"
        assert text.index(last_c) < text.index(first_pv), (tt, qb)
        assert text[text.index(last_c):text.index(first_pv)].count("SB();") >= 1, (tt, qb)
# the next tile's scores are read (segment C) only after segment B, whose last MFMAs on the tt = 0 chains are B28 / B29
assert text.index("E(0);") > text.index("// B07" if Q8 else "// B31")
sys.stdout.write(text)
