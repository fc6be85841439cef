#!/usr/bin/env python3
"""The float filter's interval on a seeded campaign of (triangle, point) pairs, CPU only (the product's own filter
code through the emulator library), reproducible:

    python tests/perf/filter_campaign.py [--pairs 36000000] [--seed 2026] [--out profiles/r03_filter_campaign.json]

Part 1  the general generator of tests/test_emu.py::test_float_filter_interval_contains_the_double_value (random
        triangles, slivers / needles of every proportion, one very short side, scales 1e-3..1e3, offsets up to 1e3
        scales; points near sides / vertices / far away): the interval [q - err, q + err] must contain the DOUBLE
        value the reference computes (dg_geom.h: tri_closest) for every pair the filter accepts (valid == 1).
Part 2  the band around the filter's shape threshold: triangles with area2 / lmax^2 swept log-uniformly over
        1e-7 .. 1e-2 (threshold 1e-4: below it a triangle is DEGENERATE for the filter and never tested by it),
        acute, right and obtuse, points 1e-6 .. 1 side lengths off the plane, over the sharp corners and beyond them.
        Per decade of the shape ratio: pairs checked, violations, the largest |q - d2| / err.
Part 3  the reference-side error the interval has to absorb: |d2_reference - d2_true| / R^2 (R = |p - origin|_1 +
        mesh_l1, the scale the filter's error terms are made of), d2_true from an 80-bit evaluation, per decade.
        dg_geom.h derives |d2_reference - d2_true| <= (c1 u + c2 u^2 / rho^4) (D + l)^2 (u = 2^-53, D = |p - v0|, l the
        longest side, rho the shape ratio); the table holds the measured error on that scale for ALL triangles of the
        decade (accepted by the filter or not), the measured c2, and the margin to the kappa term (5 E^2 = 1.8e-11 R^2).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dgtest as T  # noqa: E402
import emu  # noqa: E402
import ctypes as C  # noqa: E402


def check(tri, pts, origin):
    L = emu.lib()
    L.emu_filter_check.restype = C.c_uint64
    L.emu_filter_check.argtypes = [T.c_dp, C.c_size_t, T.c_dp, C.c_size_t, T.c_dp, T.c_dp, T.c_u64p]
    worst = C.c_double(0.0)
    checked = C.c_uint64(0)
    tri = np.ascontiguousarray(tri.reshape(-1, 9))
    pts = np.ascontiguousarray(pts)
    bad = L.emu_filter_check(T.dp(tri), len(tri), T.dp(pts), len(pts), T.dp(np.ascontiguousarray(origin)), C.byref(worst),
                             C.byref(checked))
    return int(bad), int(checked.value), float(worst.value)


def sliver_band(rng, n_tri, rho_lo, rho_hi, scale):
    """triangles A, B, C with area2 / lmax^2 = rho (log-uniform), apex anywhere over (and beyond) the base"""
    A = rng.uniform(-1, 1, size=(n_tri, 3)) * scale
    d = rng.normal(size=(n_tri, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    Lb = scale * 10.0 ** rng.uniform(-1.5, 0.3, size=(n_tri, 1))
    B = A + d * Lb
    r = rng.normal(size=(n_tri, 3))
    perp = r - (r * d).sum(1, keepdims=True) * d
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    t = rng.uniform(-0.3, 1.3, size=(n_tri, 1))
    rho = 10.0 ** rng.uniform(np.log10(rho_lo), np.log10(rho_hi), size=(n_tri, 1))
    lmax = Lb * np.maximum(1.0, np.maximum(np.abs(t), np.abs(1.0 - t)))      # longest side (to first order in the height)
    h = rho * lmax * lmax / Lb
    Cc = A + d * (t * Lb) + perp * h
    return np.stack([A, B, Cc], axis=1)


def shape_ratio(tri):
    e0, e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], tri[:, 2] - tri[:, 1]
    a2 = np.linalg.norm(np.cross(e0, e1), axis=1)
    lm = np.maximum(np.linalg.norm(e0, axis=1), np.maximum(np.linalg.norm(e1, axis=1), np.linalg.norm(e2, axis=1)))
    return a2 / (lm * lm), lm


def adversarial_points(rng, tri, n_pts):
    """points near the plane of random triangles of the set: over the interior, the sides, the sharp corners, beyond them"""
    k = rng.integers(0, len(tri), n_pts)
    A, B, Cc = tri[k, 0], tri[k, 1], tri[k, 2]
    L = np.linalg.norm(B - A, axis=1, keepdims=True)
    n = np.cross(B - A, Cc - A)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-300)
    w = rng.dirichlet([0.35, 0.35, 0.35], size=n_pts)
    inplane = w[:, :1] * A + w[:, 1:2] * B + w[:, 2:] * Cc
    corner = np.where(rng.random((n_pts, 1)) < 0.5, A, B)
    along = (B - A) / L
    sgn = np.where((corner == A).all(1, keepdims=True), -1.0, 1.0)
    beyond = corner + sgn * along * L * 10.0 ** rng.uniform(-6, 0, size=(n_pts, 1))
    base = np.where(rng.random((n_pts, 1)) < 0.5, inplane, beyond)
    off = n * L * 10.0 ** rng.uniform(-6, 0, size=(n_pts, 1)) * rng.choice([-1.0, 1.0], size=(n_pts, 1))
    jitter = rng.normal(size=(n_pts, 3)) * L * 10.0 ** rng.uniform(-9, -2, size=(n_pts, 1))
    return base + off + jitter


def exact_d2(P, tri):
    """squared distance point - triangle in 80-bit arithmetic (projection + clamped sides): the 'true' value"""
    ld = np.longdouble
    p, a, b, c = (x.astype(ld) for x in (P, tri[:, 0], tri[:, 1], tri[:, 2]))

    def seg(p, a, b):
        ab = b - a
        t = ((p - a) * ab).sum(1) / np.maximum((ab * ab).sum(1), ld(1e-4000))
        t = np.clip(t, 0, 1)[:, None]
        q = a + t * ab
        return ((p - q) ** 2).sum(1)
    n = np.cross(b - a, c - a)
    nn = (n * n).sum(1)
    h = ((p - a) * n).sum(1)
    proj = p - n * (h / np.maximum(nn, ld(1e-4000)))[:, None]

    def side(u, v):
        return (np.cross(v - u, proj - u) * n).sum(1)
    inside = (side(a, b) >= 0) & (side(b, c) >= 0) & (side(c, a) >= 0)
    d_in = h * h / np.maximum(nn, ld(1e-4000))
    d_out = np.minimum(seg(p, a, b), np.minimum(seg(p, b, c), seg(p, c, a)))
    return np.where(inside, d_in, d_out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=36_000_000)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_filter_campaign.json"))
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    out = {"seed": a.seed, "threshold_area2_over_lmax2": 1e-4}
    # ---- part 1
    tot = dict(violations=0, checked=0, worst=0.0)
    while tot["checked"] < a.pairs // 2:
        r = emu.filter_interval_check(rng, n_tri=4000, n_pts=64)
        tot["violations"] += r["violations"]
        tot["checked"] += r["checked"]
        tot["worst"] = max(tot["worst"], r["worst"])
    out["general"] = tot
    print("general:", tot, "%.0f s" % (time.time() - t0), flush=True)
    # ---- part 2 + 3: the band around the threshold, per decade of the shape ratio
    decades = [(1e-7, 1e-5), (1e-5, 1e-4), (1e-4, 1e-3), (1e-3, 1e-2)]
    band = []
    per = a.pairs // 2 // len(decades)
    for lo, hi in decades:
        st = dict(rho_lo=lo, rho_hi=hi, violations=0, checked=0, worst=0.0, accepted_triangles=0, triangles=0,
                  ref_err_over_R2_max=0.0, ref_rel_err_max=0.0)
        rounds = 0
        while st["checked"] < per and rounds < max(4, per // 300_000):   # (below the threshold nothing is ever checked)
            rounds += 1
            scale = 10.0 ** rng.integers(-2, 3)
            tri = sliver_band(rng, 500, lo, hi, scale)
            shift = rng.uniform(-1, 1, size=3) * scale * 10.0 ** rng.integers(0, 3)
            tri = tri + shift
            origin = 0.5 * (tri.reshape(-1, 3).min(0) + tri.reshape(-1, 3).max(0))
            pts = adversarial_points(rng, tri, 256)
            bad, n, w = check(tri, pts, origin)
            rho, _ = shape_ratio(tri)
            st["violations"] += bad
            st["checked"] += n
            st["worst"] = max(st["worst"], w)
            st["triangles"] += len(tri)
            st["accepted_triangles"] += int((rho > 1e-4).sum())
            # reference-side error: each point against its own triangle; on ALL triangles (what the shape threshold
            # keeps away from the filter) and on the accepted ones
            k = rng.integers(0, len(tri), len(pts))
            V = tri.reshape(-1, 3)
            F = np.arange(len(V)).reshape(-1, 3)
            l1 = np.abs(V - origin).sum(1).max()
            R_all = np.abs(pts - origin).sum(1) + l1
            e_all = np.abs(T.oracle_point_triangle(V, F, k, pts).astype(np.longdouble) - exact_d2(pts, tri[k]))
            st["ref_err_over_R2_max_all_triangles"] = max(st.get("ref_err_over_R2_max_all_triangles", 0.0), float((e_all / (R_all * R_all)).max()))
            ok = rho[k] > 1e-4
            if ok.any():
                d_ref = T.oracle_point_triangle(V, F, k[ok], pts[ok])
                d_true = exact_d2(pts[ok], tri[k[ok]])
                R = R_all[ok]
                err = np.abs(d_ref.astype(np.longdouble) - d_true)
                st["ref_err_over_R2_max"] = max(st["ref_err_over_R2_max"], float((err / (R * R)).max()))
                big = d_true > 1e-6 * R * R
                if big.any():
                    st["ref_rel_err_max"] = max(st["ref_rel_err_max"], float((err[big] / d_true[big]).max()))
        st["kappa_slack_over_R2"] = 5 * 2.0 ** -38
        st["margin_to_slack"] = st["kappa_slack_over_R2"] / max(st["ref_err_over_R2_max"], 1e-300) if st["accepted_triangles"] else None
        band.append(st)
        print("band %.0e..%.0e:" % (lo, hi), st, "%.0f s" % (time.time() - t0), flush=True)
    out["band"] = band
    # ---- part 3: the reference's own rounding error on the scale of the derivation, by shape ratio (unshifted unit-scale
    # triangles: the error is relative to (D + l)^2, D = |p - v0|, l = the longest side)
    ref = []
    for lo, hi in [(1e-8, 1e-7), (1e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1e-4), (1e-4, 1e-3), (1e-3, 1e-2), (1e-2, 1.0)]:
        worst, c2 = 0.0, 0.0
        for _ in range(20):
            tri = sliver_band(rng, 500, lo, hi, 1.0)
            pts = adversarial_points(rng, tri, 4000)
            k = rng.integers(0, len(tri), len(pts))
            V = tri.reshape(-1, 3)
            F = np.arange(len(V)).reshape(-1, 3)
            e = np.abs(T.oracle_point_triangle(V, F, k, pts).astype(np.longdouble) - exact_d2(pts, tri[k]))
            rho, lm = shape_ratio(tri)
            Dl = np.linalg.norm(pts - tri[k, 0], axis=1) + lm[k]
            rel = (e / (Dl * Dl)).astype(np.float64)
            worst = max(worst, float(rel.max()))
            c2 = max(c2, float(((rel - 4.0 * 2.0 ** -53).clip(0) * rho[k] ** 4 / 2.0 ** -106).max()))
        ref.append(dict(rho_lo=lo, rho_hi=hi, ref_err_over_Dl2_max=worst, c2_max=c2))
        print("reference error %.0e..%.0e: %.3g (D + l)^2, c2 <= %.3g" % (lo, hi, worst, c2), flush=True)
    out["reference_error"] = ref
    out["pairs_checked"] = tot["checked"] + sum(b["checked"] for b in band)
    out["violations"] = tot["violations"] + sum(b["violations"] for b in band)
    out["seconds"] = time.time() - t0
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out, "pairs", out["pairs_checked"], "violations", out["violations"])
    return 1 if out["violations"] else 0


if __name__ == "__main__":
    sys.exit(main())
