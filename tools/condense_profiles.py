#!/usr/bin/env python3
"""Condenses the rocprofv3 databases collected by tools/profile_on_gpu.sh (gpurun_out/prof/) into the
small summaries committed under profiles/:  python tools/condense_profiles.py r01"""
import csv
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.environ.get("VSLAM_PROFILES_DST") or os.path.join(ROOT, "profiles")   # on the GPU box: a directory under gpurun_out/


def kernel_stats(db_path, csv_path):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                           "from kernels group by name order by sum(end - start) desc"))
    total = sum(r[2] for r in rows) or 1
    with open(csv_path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, calls, tot, avg, mn, mx in rows:
            w.writerow([name, calls, tot, "%.1f" % avg, "%.2f" % (100.0 * tot / total), mn, mx])
    return {r[0]: {"calls": r[1], "avg_ns": r[3]} for r in rows}


def median_ms(db_path, name_part):
    """Median duration (ms) of the launches of the kernel whose name contains `name_part`."""
    db = sqlite3.connect(db_path)
    d = sorted(r[0] for r in db.execute("select end - start from kernels where name like ?", ("%" + name_part + "%",)))
    return d[len(d) // 2] / 1e6 if d else None


def trace_summary(db_path, txt_path, focus, reps):
    """Per-kernel totals over the last of `reps` equal repetitions in a kernel trace, the durations of the
    `focus` kernel and the idle gaps between its consecutive launches."""
    db = sqlite3.connect(db_path)
    rows = [r for r in db.execute("select name, start, end from kernels order by start") if "vslam" in r[0]]
    n = len(rows) // reps
    last = rows[-n:]
    t0, t1 = last[0][1], max(r[2] for r in last)
    tot = {}
    for name, a, b in last:
        k = name.split("(")[0][-70:]
        c = tot.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += (b - a) / 1e3
    busy = sum(v[1] for v in tot.values())
    with open(txt_path, "w") as f:
        f.write("last of %d repetitions: span %.3f ms over %d dispatches, kernels busy %.3f ms\n" % (reps, (t1 - t0) / 1e6, n, busy / 1e3))
        for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
            f.write("%-72s %5d calls %10.1f us  avg %8.1f  %5.1f %% of span\n" % (k, c, us, us / c, 100.0 * us / ((t1 - t0) / 1e3)))
        sel = [r for r in last if focus in r[0]]
        if sel:
            d = [(b - a) / 1e3 for _, a, b in sel]
            gaps = [(y[1] - x[2]) / 1e3 for x, y in zip(sel, sel[1:])]
            f.write("%s: %d launches, durations us %s\n" % (focus, len(sel), [round(x) for x in d]))
            f.write("gaps between consecutive launches us %s\n" % [round(x) for x in gaps])
            f.write("sum of durations %.1f us (%.1f %% of span), sum of gaps %.1f us\n" % (sum(d), 100.0 * sum(d) / ((t1 - t0) / 1e3), sum(gaps)))


def counters(db_path, names):
    db = sqlite3.connect(db_path)
    out = {}
    for kname, cname, mean, cnt in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                              "group by kernel_name, counter_name"):
        if any(k in kname for k in names):
            out.setdefault(kname, {})[cname] = mean
            out[kname]["launches"] = cnt
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(DST, exist_ok=True)
    stats = kernel_stats(os.path.join(SRC, "bench", "bench_results.db"), os.path.join(DST, tag + "_bench_kernel_stats.csv"))
    kernel_stats(os.path.join(SRC, "qr", "qr_results.db"), os.path.join(DST, tag + "_qr6012_kernel_stats.csv"))
    trace_summary(os.path.join(SRC, "qr", "qr_results.db"), os.path.join(DST, tag + "_qr6012_trace_summary.txt"), "qr_block_kernel", 3)
    for name, focus in (("bank_affine", "qr_block_kernel"), ("bank_iekf", "potf2_upper_kernel"), ("c4", "qr_block_kernel")):
        dbp = os.path.join(SRC, name, name + "_results.db")
        if os.path.exists(dbp):
            kernel_stats(dbp, os.path.join(DST, "%s_%s_kernel_stats.csv" % (tag, name)))
    with open(os.path.join(SRC, "bench.log")) as f:
        line = [l for l in f if l.startswith("{")][-1]
    with open(os.path.join(DST, tag + "_bench.json"), "w") as f:
        f.write(line)

    fetch = counters(os.path.join(SRC, "fetch", "fetch_results.db"), ["loglik"])
    write = counters(os.path.join(SRC, "write", "write_results.db"), ["loglik"])
    B, L, n = 512, 150, 462
    alg = (8 * n * n + 16 * n + 17 * L + 168) * B
    kern = {}
    total = 0.0
    for k in fetch:
        fkb, wkb = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
        kern[k] = {"FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb, "fetch_bytes_corrected": 2 * fkb * 1024, "write_bytes": wkb * 1024}
        total += 2 * fkb * 1024 + wkb * 1024
    traffic = {
        "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/run_loglik.py 512 150 3",
        "workload": "512 instances, L=150, n=462, value+gradient+Hessian (one point_loglik call = terms + assemble kernels)",
        "correction": "FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE taken as is; units KB",
        "kernels": kern, "hbm_bytes_per_launch": total, "algorithmic_bytes_per_launch": alg,
        # duration of the same kernel in the kernel trace of bench.py taken by the same script: bench.py replays the
        # byte counts above only while its own launches last this long
        "assemble_kernel_avg_ms": median_ms(os.path.join(SRC, "bench", "bench_results.db"), "loglik_assemble_kernel"),
        "terms_kernel_avg_ms": median_ms(os.path.join(SRC, "bench", "bench_results.db"), "loglik_terms_kernel"),
    }
    with open(os.path.join(DST, "traffic_point_loglik.json"), "w") as f:
        json.dump(traffic, f, indent=1)

    cf_db, cw_db = os.path.join(SRC, "corner_fetch", "fetch_results.db"), os.path.join(SRC, "corner_write", "write_results.db")
    if os.path.exists(cf_db) and os.path.exists(cw_db):
        names = ["corner_strip_kernel", "corner_tile_kernel", "sort_chunk_kernel", "merge_chunks_kernel", "emit_features_kernel"]
        cf, cw = counters(cf_db, names), counters(cw_db, names)
        ck, ctotal = {}, 0.0
        for k in cf:
            fkb, wkb = cf[k]["FETCH_SIZE"], cw.get(k, {}).get("WRITE_SIZE", 0.0)
            per_step = cf[k]["launches"] / 3.0          # run_corner.py 64 3 = three steps
            ck[k] = {"FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb, "launches_per_step": per_step,
                     "bytes_per_step": (2 * fkb + wkb) * 1024 * per_step}
            ctotal += ck[k]["bytes_per_step"]
        with open(os.path.join(DST, "traffic_corner.json"), "w") as f:
            json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/run_corner.py 64 3",
                       "workload": "64 frames 1920x1080, both detectors, keypoints out (one step = strip + sort + merge + emit kernels)",
                       "correction": "FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE taken as is; units KB",
                       "kernels": ck, "hbm_bytes_per_step": ctotal,
                       "algorithmic_bytes_per_step": 64 * 1920 * 1080 + 12 * 64 * (1873 + 6265)}, f, indent=1)
        print("corner HBM bytes per step %.0f" % ctotal)

    ff_db, fw_db = os.path.join(SRC, "fast_fetch", "fetch_results.db"), os.path.join(SRC, "fast_write", "write_results.db")
    if os.path.exists(ff_db) and os.path.exists(fw_db):
        names = ["fast9_tile_kernel", "sort_chunk_kernel", "merge_chunks_kernel", "emit_features_kernel"]
        ff, fw = counters(ff_db, names), counters(fw_db, names)
        fk, ftotal = {}, 0.0
        for k in ff:
            fkb, wkb = ff[k]["FETCH_SIZE"], fw.get(k, {}).get("WRITE_SIZE", 0.0)
            per_step = ff[k]["launches"] / 3.0          # run_fast9.py 64 3 = three steps
            fk[k] = {"FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb, "launches_per_step": per_step,
                     "bytes_per_step": (2 * fkb + wkb) * 1024 * per_step}
            ftotal += fk[k]["bytes_per_step"]
        with open(os.path.join(DST, "traffic_fast9.json"), "w") as f:
            json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/run_fast9.py 64 3",
                       "workload": "64 frames 1920x1080, FAST-9/16 threshold 20 with non-maximum suppression, keypoints out",
                       "correction": "FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE taken as is; units KB",
                       "kernels": fk, "hbm_bytes_per_step": ftotal}, f, indent=1)
        print("fast9 HBM bytes per step %.0f" % ftotal)

    sq = counters(os.path.join(SRC, "corner_sq", "sq_results.db"), ["corner_strip_kernel", "corner_tile_kernel"])
    with open(os.path.join(DST, tag + "_corner_sq_counters.json"), "w") as f:
        json.dump({"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU "
                              "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -- python3 tools/run_corner.py 64 3",
                   "workload": "64 frames 1920x1080, both detectors", "mean_per_launch": sq}, f, indent=1)
    mf_db = os.path.join(SRC, "qr_mfma", "mfma_results.db")
    if os.path.exists(mf_db) and not os.path.exists(os.path.join(SRC, "qr_mfma.failed")):
        mf = counters(mf_db, ["gemm_f64_kernel", "qr_block_kernel", "qr_panel_coop_kernel"])
        with open(os.path.join(DST, tag + "_qr6012_mfma_counters.json"), "w") as f:
            json.dump({"command": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA "
                                  "-- python3 tools/qr_bench.py 6012 6012 1",
                       "workload": "in-place QR 6012 x 6012 fp64", "mean_per_launch": mf}, f, indent=1)
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["avg_ns"] * kv[1]["calls"])[:8]:
        print("%-80s %6d %12.1f us" % (k[:80], v["calls"], v["avg_ns"] / 1e3))
    print("point_loglik HBM bytes per launch %.0f (algorithmic %.0f)" % (total, alg))


if __name__ == "__main__":
    main()
