#!/usr/bin/env python3
"""Turns the rocprofv3 databases profiles/collect.sh left under gpurun_out/prof_<tag>/ into
  profiles/<tag>_kernel_stats.txt   per-kernel durations (rocprofv3 --kernel-trace --stats)
  profiles/<tag>_pmc_summary.txt    mean PMC counter values per dispatch + derived figures
  profiles/counters.json            the derived figures bench.py puts on its JSON line, with the SHA-256 of
                                    discregrid_amd/csrc they were measured on
Usage: python profiles/summarize_pmc.py <dir> <tag>
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): separate --pmc passes; FETCH_SIZE is
reported in KiB and counts 64 B per 128-B request on gfx950 -> read bytes = 2 x FETCH_SIZE x 1024;
write bytes = WRITE_SIZE x 1024.  VALU busy = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x kernel
cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs."""
import glob
import hashlib
import json
import os
import re
import sqlite3
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["k_sample_fast", "k_sample_nodes", "k_heavy_subtrees", "k_expand_tiles", "k_heavy_finish", "k_interpolate_tiles", "k_tile_keys", "k_tile_bounds", "k_tile_row_items", "k_tile_items", "k_interpolate_binned", "k_interpolate_band", "k_interpolate_rows", "k_interpolate", "k_bin_probe",
           "k_bin_keys", "k_density_cells", "k_density_rows", "k_xmajor_copy", "k_xmajor_flags", "k_density_pairs", "k_density_bricks_lds", "k_density_bricks", "k_tile_flags", "k_field_check", "k_unpack_shards", "k_unpack_ranks", "k_expand_cells"]


def short(name):
    for k in KERNELS:
        m = re.search(r"\b" + k + r"(<[^>]*>)?", name)
        if m:
            return m.group(0)
    if "rocprim" in name or "radix" in name or "onesweep" in name:
        m = re.search(r"(\w*(sort|onesweep|histogram|scan)\w*)", name)
        return "rocprim::" + (m.group(1) if m else "kernel")
    return None


def csrc_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "discregrid_amd", "csrc")
    for f in sorted(os.listdir(d)):
        # (not the host-only sources: the copy pipeline, the CPU point query and the exchange glue do not change what the kernels do)
        if f.endswith((".hip", ".h", ".cpp")) and f not in ("dg_capi_host.cpp", "dg_host_query.cpp", "dg_capi_comm.cpp", "dg_capi_hostfield.cpp", "dg_capi_vmm.h", "dg_capi_shm.h"):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def find_db(d, stem):
    hits = sorted(glob.glob(os.path.join(d, "**", stem + "_results.db"), recursive=True)) or \
        sorted(glob.glob(os.path.join(d, "**", stem + "*.db"), recursive=True))
    return hits[-1] if hits else None


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = []
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        rows.append((name, calls, total * 1e3, avg * 1e3, pct))
    disp = list(c.execute("select name,duration,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size from kernels order by start"))
    return rows, disp


def counters(db):
    c = sqlite3.connect(db)
    out = {}
    for k, cn, v, n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        s = short(str(k))
        if s:
            out[(s, cn)] = (v, n)
    return out


def main():
    d, tag = sys.argv[1], sys.argv[2]
    stats_lines, pmc_lines = [], []
    derived = {}
    for w in ("k1", "k2", "k2r", "k2b", "k3", "u"):
        kt = find_db(d, w + "_kt")
        if not kt:
            continue
        rows, disp = kernel_stats(kt)
        stats_lines.append("# workload %s: rocprofv3 --kernel-trace --stats (%s), 1x MI355X" % (w, os.path.basename(kt)))
        stats_lines.append("%-80s %8s %16s %16s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        dur = {}
        for name, calls, total, avg, pct in rows:
            s = short(name)
            stats_lines.append("%-80s %8d %16.0f %16.0f %8.3f" % ((s or name)[:80], calls, total, avg, pct))
            if s:
                dur[s] = (avg, calls)
        seen = set()
        regs = {}
        for r in disp:
            s = short(r[0])
            if s and s not in seen:
                seen.add(s)
                regs[s] = int(r[4])
                stats_lines.append("#   %-40s grid=%d wg=%d vgpr=%d sgpr=%d lds=%d scratch=%d" % ((s,) + tuple(r[2:])))
        stats_lines.append("")
        allc = {}
        i = 0
        while True:
            i += 1
            db = find_db(d, "%s_pmc%d" % (w, i))
            if not db:
                break
            cs = counters(db)
            for (k, cn), (v, n) in sorted(cs.items()):
                pmc_lines.append("%-4s pmc%d  %-28s %-28s %16.6g %5d" % (w, i, k, cn, v, n))
                allc.setdefault(k, {})[cn] = v
        for k, cs in allc.items():
            if k not in dur:
                continue
            e = {"kernel_ms": dur[k][0] * 1e-6, "calls_in_trace": dur[k][1]}
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                e["hbm_read_bytes"] = 2.0 * cs["FETCH_SIZE"] * 1024.0
                e["hbm_write_bytes"] = cs["WRITE_SIZE"] * 1024.0
                e["hbm_bytes_per_launch"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
                e["hbm_gbs"] = e["hbm_bytes_per_launch"] / (e["kernel_ms"] * 1e-3) / 1e9
                e["hbm_frac"] = e["hbm_gbs"] / 8000.0
            if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs and cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"] > 0:
                e["l2_hit_rate"] = cs["TCC_HIT_sum"] / (cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"])
            if "GRBM_GUI_ACTIVE" in cs and "SQ_ACTIVE_INST_VALU" in cs and cs["GRBM_GUI_ACTIVE"] > 0:
                cyc = cs["GRBM_GUI_ACTIVE"] / 8.0
                e["kernel_cycles"] = cyc
                e["valu_busy"] = cs["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)
            if "SQ_WAVES" in cs and cs["SQ_WAVES"] > 0:
                wv = cs["SQ_WAVES"]
                e["waves"] = wv
                e["per_wave"] = {n_: cs[c_] / wv for n_, c_ in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("smem", "SQ_INSTS_SMEM"),
                                                              ("lds", "SQ_INSTS_LDS")) if c_ in cs}
                f64 = sum(cs.get(c_, 0.0) for c_ in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
                f32 = sum(cs.get(c_, 0.0) for c_ in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"))
                if f64:
                    e["per_wave"]["valu_f64"] = f64 / wv
                if f32:
                    e["per_wave"]["valu_f32"] = f32 / wv
                if "SQ_WAVE_CYCLES" in cs and cs["SQ_WAVE_CYCLES"] > 0:
                    e["wait_inst_any_frac"] = cs.get("SQ_WAIT_INST_ANY", 0.0) / cs["SQ_WAVE_CYCLES"]
                # efficiency, not utilisation: the share of the issued vector instructions that is floating-point arithmetic,
                # and the share of the scalar unit's issue slots (one per cycle and CU, 256 CUs) that scalar ALU + memory instructions fill
                if (f64 or f32) and cs.get("SQ_INSTS_VALU", 0) > 0:
                    e["arith_frac"] = (f64 + f32) / cs["SQ_INSTS_VALU"]
                if "GRBM_GUI_ACTIVE" in cs and cs["GRBM_GUI_ACTIVE"] > 0 and "SQ_INSTS_SALU" in cs:
                    e["salu_slot_frac"] = (cs["SQ_INSTS_SALU"] + cs.get("SQ_INSTS_SMEM", 0.0)) / (256.0 * cs["GRBM_GUI_ACTIVE"] / 8.0)
            if "GRBM_GUI_ACTIVE" in cs and cs["GRBM_GUI_ACTIVE"] > 0:
                cyc = cs["GRBM_GUI_ACTIVE"] / 8.0
                if "TA_TA_BUSY_sum" in cs:   # 256 texture-address / texture-data units, one per CU
                    e["ta_busy"] = cs["TA_TA_BUSY_sum"] / (256.0 * cyc)
                if "TD_TD_BUSY_sum" in cs:
                    e["td_busy"] = cs["TD_TD_BUSY_sum"] / (256.0 * cyc)
                if "TA_FLAT_READ_WAVEFRONTS_sum" in cs and cs["TA_FLAT_READ_WAVEFRONTS_sum"] > 0 and "TD_TD_BUSY_sum" in cs:
                    e["td_cycles_per_load_instruction"] = cs["TD_TD_BUSY_sum"] / cs["TA_FLAT_READ_WAVEFRONTS_sum"]
            if k in regs and regs[k] > 0:
                e["arch_vgprs"] = regs[k]   # (what rocprofv3 reports; accumulation registers used as spill space come on top)
            derived.setdefault(w, {})[k] = e
    k1 = None
    if "k1" in derived:
        # the dominant kernel of a K1 launch: the filtered kernel by default, the exact one with DG_FORCE=k1_fast=0
        for want in ("k_sample_fast", "k_sample_nodes"):
            for k, e in derived["k1"].items():
                if k1 is None and k.startswith(want):
                    k1 = dict(e)
                    k1["kernel"] = k
                    k1["per_brick"] = e.get("per_wave")
    out = {"csrc_sha256": csrc_hash(), "collected": time.strftime("%Y-%m-%d %H:%M:%S"), "tag": tag,
           "method": "rocprofv3 --pmc (separate passes) via profiles/collect.sh; HBM bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024; "
                     "valu_busy = SQ_ACTIVE_INST_VALU*4 / (1024 * GRBM_GUI_ACTIVE/8)",
           "k1": k1, "workloads": derived}
    json.dump(out, open(os.path.join(ROOT, "profiles", "counters.json"), "w"), indent=1, sort_keys=True)
    open(os.path.join(ROOT, "profiles", "%s_kernel_stats.txt" % tag), "w").write("\n".join(stats_lines) + "\n")
    head = ["# %s: profiles/collect.sh %s (rocprofv3 --pmc, one pass per counter group, counters only), 1x MI355X" % (tag, tag),
            "# values = mean per dispatch; FETCH_SIZE / WRITE_SIZE in KiB as reported",
            "%-4s %-5s %-28s %-28s %16s %5s" % ("wl", "pass", "kernel", "counter", "mean/dispatch", "n")]
    tail = ["", "# derived (see profiles/counters.json for all kernels)"]
    for w, ks in derived.items():
        for k, e in ks.items():
            parts = ["%.3f ms" % e["kernel_ms"]]
            if "hbm_bytes_per_launch" in e:
                parts.append("HBM %.3e B/launch = %.0f GB/s = %.3f of 8 TB/s" % (e["hbm_bytes_per_launch"], e["hbm_gbs"], e["hbm_frac"]))
            if "l2_hit_rate" in e:
                parts.append("L2 hit %.3f" % e["l2_hit_rate"])
            if "valu_busy" in e:
                parts.append("VALU busy %.3f" % e["valu_busy"])
            if "ta_busy" in e:
                parts.append("TA busy %.3f, TD busy %.3f" % (e["ta_busy"], e.get("td_busy", float("nan"))))
            if "per_wave" in e and "valu" in e["per_wave"]:
                parts.append("VALU/wave %.0f" % e["per_wave"]["valu"])
            tail.append("# %-3s %-28s %s" % (w, k, "; ".join(parts)))
    open(os.path.join(ROOT, "profiles", "%s_pmc_summary.txt" % tag), "w").write("\n".join(head + pmc_lines + tail) + "\n")
    print("\n".join(tail))


if __name__ == "__main__":
    main()
