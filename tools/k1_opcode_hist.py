#!/usr/bin/env python3
"""Per-opcode DYNAMIC histogram of the filtered K1 traversal (k_sample_fast<false>): static instruction counts of the
traversal's blocks in the gfx950 ISA x how often the CPU emulator -- which runs the same traversal template -- passes
through them per brick (tests/perf/emu_event_rates.py -> profiles/r05_k1_event_rates_*.json).  What it answers: of the vector
instructions one brick issues while it walks the tree, which are arithmetic and which are compares / selects / moves / integer
-- and which opcodes make up the non-arithmetic part.

    python tools/k1_opcode_hist.py [profiles/r05_k1_event_rates_ico256.json] > profiles/r05_k1_opcode_histogram.txt

The block -> event mapping (k_sample_fast, the region between the first s_load of a node pair and the end of the pop loop):
  pair step   the block with two s_load_dwordx16 + ds_write_b16_d16_hi, split at its branches:
              [load, bounds, ballots] x pair_steps | [which child / both?] x (pair_steps - dead) | [near-first vote] x pushes |
              [push] x pushes;  the two small blocks behind it x (pair_steps - dead) and x dead
  leaf        prologue (error terms) x leaf_visits; the block with three s_load_dwordx16 split at its branches:
              [step 1: frame + rectangle bound + ballots] x filter_pairs_step1 | [step 2 + side 0] x filter_pairs_step2 |
              side-1 block x filter_pairs_step2; loop tail x filter_pairs_step1; leaf epilogue x leaf_visits
  pops        the ds_read_u16 block x pops; found-pop blocks x (pops - stale_pops)
Everything else of the kernel (brick set-up, the per-lane double tests after the traversal, the exact traversal of the rare lanes
the filter cannot serve, result write) is outside "one traversal step" and listed as a static total only."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FN = "_ZN2dg12_GLOBAL__N_113k_sample_fastILb0EEEvNS_12SampleParamsE"
ARITH_F32 = ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32",
             "v_max_f32", "v_min_f32", "v_med3_f32", "v_max3_f32", "v_min3_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32")
ARITH_F64 = ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_rcp_f64", "v_sqrt_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64")


def blocks(path):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith(FN + ":")][0]
    out, cur = [], None
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = [m.group(1), []]
            out.append(cur)
            continue
        t = l.strip()
        if not t or t.startswith((".", ";", "//")):
            continue
        if cur is None:
            cur = ["entry", []]
            out.append(cur)
        cur[1].append(re.sub(r"\s+", " ", t))
    return out


def segments(ins):
    """split a block behind every conditional branch"""
    segs, cur = [], []
    for t in ins:
        cur.append(t)
        if t.startswith("s_cbranch"):
            segs.append(cur)
            cur = []
    if cur:
        segs.append(cur)
    return segs


def kind(op):
    base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
    if base in ARITH_F32:
        return "valu f32 arithmetic"
    if base in ARITH_F64:
        return "valu f64 arithmetic"
    if op.startswith("v_"):
        return "valu other"
    if op.startswith("s_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    return "vmem"


def main():
    rates_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_k1_event_rates_ico256.json")
    ev = json.load(open(rates_path))
    r = ev["per_brick"]
    asm = "/tmp/dg_k1_hist.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-S",
                           os.path.join(ROOT, "discregrid_amd", "csrc", "dg_kernels_k1.hip"), "-o", asm], stderr=subprocess.DEVNULL)
    bs = blocks(asm)
    A = next(i for i, (_, ins) in enumerate(bs) if sum("s_load_dwordx16" in t for t in ins) == 2 and any(t.startswith("ds_write_b16") for t in ins))
    F = next(i for i, (_, ins) in enumerate(bs) if sum("s_load_dwordx16" in t for t in ins) == 3)
    P = next(i for i, (_, ins) in enumerate(bs) if i > F and any(t.startswith("ds_read_u16") for t in ins))
    assert F == A + 6 and P == F + 6, "the block layout of k_sample_fast changed: re-derive the mapping (docstring) -- A=%d F=%d P=%d" % (A, F, P)
    live = r["pair_steps"] - r["dead_steps"]
    found = r["pops"] - r["stale_pops"]
    weighted = []   # (what, instructions, weight)
    sa = segments(bs[A][1])
    assert len(sa) == 5, len(sa)
    weighted += [("pair step: loop head", sa[0], r["pair_steps"] + r["leaf_visits"]), ("pair step: record load, two bounds, reach ballots", sa[1], r["pair_steps"]),
                 ("pair step: one child or both?", sa[2], live), ("pair step: near-first vote", sa[3], r["pushes"]), ("pair step: push", sa[4], r["pushes"]),
                 ("pair step: descend", bs[A + 1][1], live), ("pair step: dead end", bs[A + 2][1], r["dead_steps"]),
                 ("pair step: loop control", bs[A - 1][1] + bs[A - 2][1], r["pair_steps"]),
                 ("leaf: error terms", bs[A + 3][1], r["leaf_visits"])]
    sf = segments(bs[F][1])
    assert len(sf) >= 3, len(sf)
    weighted += [("leaf: filter step 1 (frame, rectangle bound, ballots)", sf[0], r["filter_pairs_step1"]),
                 ("leaf: filter step 2 (distance to the sides, error interval) + side 0", [t for s in sf[1:] for t in s], r["filter_pairs_step2"]),
                 ("leaf: side 0 tail", bs[F + 1][1], r["filter_pairs_step2"]), ("leaf: side 1 (append)", bs[F + 2][1] + bs[F - 2][1], r["filter_pairs_step2"]),
                 ("leaf: pair loop tail", bs[F - 1][1], r["filter_pairs_step1"]), ("leaf: threshold update", bs[F + 3][1], r["leaf_visits"]),
                 ("pop: loop entry", bs[F + 4][1], r["leaf_visits"] + r["dead_steps"]), ("pop: parked bound, reach ballot", bs[P][1] + bs[P - 1][1], r["pops"]),
                 ("pop: found", bs[P + 1][1] + bs[P + 4][1][:3], found)]
    total = collections.Counter()
    kinds = collections.Counter()
    print("# k_sample_fast<false>, one brick's TRAVERSAL (gfx950 ISA x emulator event rates: %s %d^3, %d bricks)" % (ev["mesh"], ev["res"], ev["bricks"]))
    print("# event rates per brick: " + ", ".join("%s %.2f" % (k, v) for k, v in r.items()))
    print("#\n# %-66s %8s %8s %10s %10s" % ("segment", "instr", "x / brick", "VALU", "SALU+SMEM"))
    for what, ins, w in weighted:
        v = sum(t.startswith("v_") for t in ins)
        s_ = sum(t.startswith("s_") for t in ins)
        print("  %-66s %8d %8.2f %10.1f %10.1f" % (what, len(ins), w, v * w, s_ * w))
        for t in ins:
            op = t.split()[0]
            total[op] += w
            kinds[kind(op)] += w
    n_valu = sum(v for k, v in kinds.items() if k.startswith("valu"))
    print("#\n# per brick, traversal only: %.0f instructions, of them %.0f VALU" % (sum(kinds.values()), n_valu))
    for k in sorted(kinds, key=kinds.get, reverse=True):
        print("  %-22s %9.1f   %5.1f %% of all, %5.1f %% of VALU" % (k, kinds[k], 100 * kinds[k] / sum(kinds.values()), 100 * kinds[k] / n_valu if k.startswith("valu") else 0.0))
    print("#\n# the NON-ARITHMETIC vector instructions (compares, selects, moves, integer, conversions), most frequent first:")
    other = [(op, n) for op, n in total.items() if kind(op) == "valu other"]
    for op, n in sorted(other, key=lambda x: -x[1])[:12]:
        print("  %-28s %8.1f per brick   %5.1f %% of VALU" % (op, n, 100 * n / n_valu))
    print("#\n# the most frequent scalar instructions:")
    sc = [(op, n) for op, n in total.items() if kind(op) in ("salu", "smem")]
    for op, n in sorted(sc, key=lambda x: -x[1])[:10]:
        print("  %-28s %8.1f per brick" % (op, n))
    static_all = collections.Counter(kind(t.split()[0]) for _, ins in bs for t in ins)
    print("#\n# static instruction mix of the whole kernel (all %d blocks): %s" % (len(bs), dict(static_all)))


if __name__ == "__main__":
    main()
