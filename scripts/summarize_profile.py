#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by scripts/profile.sh into the small files
kept under profiles/:

  profiles/<tag>_kernel_trace.md   per kernel and grid size: calls, average duration
  profiles/<tag>_pmc.md            FETCH_SIZE / WRITE_SIZE per dispatch, calibration, bytes per unit
  profiles/traffic.json            HBM bytes per unit of work per opcode (read by bench.py)

Counter handling for gfx950: FETCH_SIZE is reported in KiB and counts half of
the bytes of a coalesced streaming read; the factor is re-measured here from the
membench kernels, whose byte counts are known, for 4-, 8- and 16-byte-per-lane
loads.  WRITE_SIZE (KiB) is calibrated the same way from the copy kernels.

    python scripts/summarize_profile.py gpurun_out/prof_r01 r01
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
os.makedirs(PROF, exist_ok=True)
csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"(rtdp::k_\w+<[^>]*>|k_read<[^>]*>|k_copy<[^>]*>|rtdp::k_stream_\w+)", name)
    return m.group(1) if m else None


def rows(pattern):
    for path in glob.glob(os.path.join(src, pattern)):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r


# ---- kernel trace
kt = defaultdict(list)
for r in rows("kt/*/*kernel_trace.csv"):
    k = short(r["Kernel_Name"])
    if k:
        grid = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
        kt[(k, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(PROF, tag + "_kernel_trace.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats, `python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline` (%s)\n\n" % tag)
    f.write("Per kernel and grid size (threads); the first launch of each shape is excluded from the average.\n\n")
    f.write("| kernel | grid (threads) | calls | avg us | min us | max us |\n|---|---|---|---|---|---|\n")
    for (k, grid), d in sorted(kt.items()):
        dd = d[1:] if len(d) > 1 else d
        f.write("| `%s` | %d | %d | %.1f | %.1f | %.1f |\n" % (k, grid, len(d), sum(dd) / len(dd) / 1e3, min(dd) / 1e3, max(dd) / 1e3))

# ---- kernel traces of the drivers above the op kernels
drivers = sorted(d for d in glob.glob(os.path.join(src, "kt_*")) if os.path.isdir(d))
if drivers:
    with open(os.path.join(PROF, tag + "_drivers_kernel_trace.md"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats of the drivers above the op kernels (%s)\n\n" % tag)
        f.write("`python3 bench.py --steps 3 --warmup 1 <arguments>`; library kernels only (torch's own kernels omitted), all calls averaged.\n")
        for d in drivers:
            per = defaultdict(list)
            for path in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")):
                with open(path, newline="") as fh:
                    for r in csv.DictReader(fh):
                        m = re.search(r"rtdp::(?:\(anonymous namespace\)::)?(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
                        if m:
                            per[m.group(1) + (m.group(2) or "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                        elif "rocprim" in r["Kernel_Name"]:   # the builder's radix sort and scans (hipcub)
                            per["rocprim sort / scan kernels"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            f.write("\n## %s\n\n| kernel | calls | avg us | total ms |\n|---|---|---|---|\n" % os.path.basename(d)[3:])
            for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                f.write("| `%s` | %d | %.1f | %.2f |\n" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))

# ---- counters
def pmc(pattern, counter):
    out = defaultdict(list)
    for r in rows(pattern):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == counter:
            out[(k, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return out


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


lines = ["# FETCH_SIZE / WRITE_SIZE passes (%s)\n" % tag,
         "Each counter was collected in a rocprofv3 run of its own (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`).\n",
         "## Calibration on membench (1 GiB buffers, known byte counts)\n",
         "| kernel | bytes moved | counter KiB (median) | bytes per counted KiB-unit |\n|---|---|---|---|"]
GiB = 1 << 30
fcal, wcal = {}, {}
for (k, grid), v in sorted(pmc("pmc_fetch_membench/*/*counter_collection.csv", "FETCH_SIZE").items()):
    if k.startswith("k_read"):
        factor = GiB / (med(v) * 1024.0)
        fcal[k] = factor
        lines.append("| `%s` | %d read | %.1f | x%.3f |" % (k, GiB, med(v), factor))
for (k, grid), v in sorted(pmc("pmc_write_membench/*/*counter_collection.csv", "WRITE_SIZE").items()):
    if k.startswith("k_copy"):
        factor = GiB / (med(v) * 1024.0)
        wcal[k] = factor
        lines.append("| `%s` | %d written | %.1f | x%.3f |" % (k, GiB, med(v), factor))
ffac = sorted(fcal.values())[len(fcal) // 2] if fcal else 2.0
wfac = sorted(wcal.values())[len(wcal) // 2] if wcal else 1.0
lines.append("\nFETCH_SIZE correction used: x%.3f (KiB -> bytes: x1024, then the factor).  WRITE_SIZE correction: x%.3f.\n" % (ffac, wfac))

# traffic.json keys: the layout bench.py runs by default is tiled, so "box" / "tri" passes see the tiled kernels and the
# "_plain" passes the kernels over one array per operand; bench.py looks the per-unit figures up as <opcode>_tiled / <opcode>
UNITS = {"box_tiled": (("rtdp::k_box_tiled<",), 156, 136, 20), "tri_tiled": (("rtdp::k_tri_tiled<",), 76, 64, 12),
         "box": (("rtdp::k_box<",), 156, 136, 20), "tri": (("rtdp::k_tri<",), 76, 64, 12),
         "euclid": (("rtdp::k_vec<true", "rtdp::k_vec_chain<true"), 136, 132, 4),
         "angular": (("rtdp::k_vec<false", "rtdp::k_vec_chain<false"), 76, 68, 8),
         "angular_short": (("rtdp::k_vec<false", "rtdp::k_vec_chain<false"), 76, 68, 8)}   # 2^22 beats: the shape short segments get
lines += ["## Datapath kernels (headline batch of each opcode)\n",
          "| opcode | kernel | units per launch | fetched B/unit (algorithmic) | written B/unit (algorithmic) | HBM B/unit (algorithmic) |\n|---|---|---|---|---|---|"]
traffic = {}
units_log = {}
for line in open(os.path.join(src, "log.txt")):
    if line.startswith("{"):
        try:
            d = json.loads(line)
        except Exception:
            continue
        for k, v in d.get("per_opcode", {}).items():
            if "units_per_gpu" in v and k not in units_log:   # the first line is the default benchmark line; later lines are single passes
                units_log[k] = v["units_per_gpu"]
        if d.get("config", {}).get("workload", "").startswith("angular, 2^22"):
            units_log["angular_short"] = d["config"]["ops_per_gpu"]
PASS = {"box_tiled": "box", "tri_tiled": "tri", "box": "box_plain", "tri": "tri_plain", "euclid": "euclid", "angular": "angular", "angular_short": "angular_short"}   # directory suffix of the passes
for op, (prefix, total, rd, wr) in UNITS.items():
    fe = {k: v for k, v in pmc("pmc_fetch_%s/*/*counter_collection.csv" % PASS[op], "FETCH_SIZE").items() if k[0].startswith(prefix)}
    we = {k: v for k, v in pmc("pmc_write_%s/*/*counter_collection.csv" % PASS[op], "WRITE_SIZE").items() if k[0].startswith(prefix)}
    base = op if op == "angular_short" else op.split("_")[0]
    if not fe or not we or base not in units_log:
        continue
    units_log[op] = units_log[base]
    key = max(fe, key=lambda k: k[1])  # the main launch (largest grid), not the tail launch
    n = units_log[op]
    fb = med(fe[key]) * 1024 * ffac / n
    wb = med(we[key]) * 1024 * wfac / n
    traffic[op] = {"hbm_bytes_per_unit": round(fb + wb, 2), "fetched": round(fb, 2), "written": round(wb, 2),
                   "algorithmic": total, "units": n, "kernel": key[0]}
    lines.append("| %s | `%s` | %d | %.2f (%d) | %.2f (%d) | %.2f (%d) |" % (op, key[0], n, fb, rd, wb, wr, fb + wb, total))
# ---- the record-stream front end: its kernels' bytes per record (2^24 records per call unless the pass says otherwise)
sf = pmc("pmc_fetch_stream/*/*counter_collection.csv", "FETCH_SIZE")
sw = pmc("pmc_write_stream/*/*counter_collection.csv", "WRITE_SIZE")
si = pmc("pmc_valu_stream/*/*counter_collection.csv", "SQ_INSTS_VALU")
sv = pmc("pmc_valu_stream/*/*counter_collection.csv", "SQ_WAVES")
if sf and sw:
    n_rec = 1 << 24
    lines += ["\n## Record stream, evaluate in place then fold (`bench.py --only stream --stream-mode 2`, 2^24 records per call)\n",
              "| kernel | fetched B/record | written B/record | VALU instructions per wave |\n|---|---|---|---|"]
    tf = tw = 0.0
    names = []
    for key in sorted(sf):
        if not key[0].startswith("rtdp::k_stream_") or key not in sw:
            continue
        fb, wb = med(sf[key]) * 1024 * ffac / n_rec, med(sw[key]) * 1024 * wfac / n_rec
        tf, tw = tf + fb, tw + wb
        names.append(key[0])
        vi = ("%.0f" % (med(si[key]) / med(sv[key]))) if key in si and key in sv and med(sv[key]) else "-"
        lines.append("| `%s` | %.2f | %.2f | %s |" % (key[0], fb, wb, vi))
    lines.append("| all | %.2f (336) | %.2f (84) | |" % (tf, tw))
    traffic["record_stream"] = {"hbm_bytes_per_unit": round(tf + tw, 2), "fetched": round(tf, 2), "written": round(tw, 2), "algorithmic": 420,
                                "units": n_rec, "kernel": " + ".join(names)}
# ---- VALU issue and clock
def pmc_timed(pattern, counter, prefix):
    vals, durs = [], []
    for r in rows(pattern):
        k = short(r["Kernel_Name"])
        if k and k.startswith(prefix) and r["Counter_Name"] == counter:
            vals.append((int(r["Grid_Size"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    if not vals:
        return None
    g = max(v[0] for v in vals)
    vals = [v for v in vals if v[0] == g]
    return med([v[1] for v in vals]), med([v[2] for v in vals])


PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 2  # 256 CUs x 4 SIMD-32: one wave64 VALU instruction per SIMD per 2 cycles at 2.4 GHz
lines += ["\n## VALU issue and clock (own counter passes: `--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES`, `--pmc GRBM_GUI_ACTIVE`)\n",
          "VALU issue fraction = SQ_INSTS_VALU / kernel time / 1.229 T wave-instructions/s (256 CUs x 4 SIMD-32, one wave64 instruction per 2 cycles,"
          " 2.4 GHz; the FP32 vector peak of 157.3 TFLOP/s is this rate x 64 lanes x 2 flops of an FMA).  A packed v_pk_add_f32 / v_pk_mul_f32 counts as one"
          " instruction here but occupies the SIMD like two, so the fraction understates kernels that use them.  Clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time.\n",
          "| opcode | VALU instructions per wave | per unit of work | VALU issue fraction | clock under the kernel |\n|---|---|---|---|---|"]
for op, (prefix, total, rd, wr) in UNITS.items():
    iv = pmc_timed("pmc_valu_%s/*/*counter_collection.csv" % PASS[op], "SQ_INSTS_VALU", prefix)
    wv = pmc_timed("pmc_valu_%s/*/*counter_collection.csv" % PASS[op], "SQ_WAVES", prefix)
    gr = pmc_timed("pmc_grbm_%s/*/*counter_collection.csv" % PASS[op], "GRBM_GUI_ACTIVE", prefix)
    if not iv or not wv or op not in units_log or op not in traffic:
        continue
    frac = iv[0] / (iv[1] * 1e-9) / PEAK_WAVE_INSTS
    clk = (gr[0] / 8 / (gr[1] * 1e-9) / 1e9) if gr else float("nan")
    lines.append("| %s | %.0f | %.1f | %.3f | %.2f GHz |" % (op, iv[0] / wv[0], iv[0] * 64 / units_log[op], frac, clk))
    traffic[op]["valu_issue_fraction"] = round(frac, 4)
open(os.path.join(PROF, tag + "_pmc.md"), "w").write("\n".join(lines) + "\n")
traffic["_source"] = "rocprofv3 --pmc passes %s: FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU, one counter set per run" % tag
json.dump(traffic, open(os.path.join(PROF, "traffic.json"), "w"), indent=1, sort_keys=True)
print("\n".join(lines))
