"""Turn rocprofv3 outputs (rocpd sqlite databases) into the summaries committed under profiles/.

  python tools/profile_summary.py stats  <results.db> <out.csv> [invocations]
  python tools/profile_summary.py pmc    <write.db> <fetch.db> <out_prefix> <invocations> [config]

pmc: per-kernel sums of WRITE_SIZE / FETCH_SIZE (unit KB; FETCH_SIZE doubled on gfx950, which tallies 128-byte requests
as 64 bytes -- /opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section), grouped into the bench phases, per
build+solve+eval cycle, written to <out_prefix>_write_summary.csv / _fetch_summary.csv and merged into
profiles/pmc_traffic.json under the config key."""
import csv
import json
import os
import sqlite3
import sys

PHASES = [
    ("fit", ("small_fit_kernel",)),
    ("check", ("batch_check_kernel",)),
    ("gram", ("gram_mfma", "gram_diff_kernel", "gram_w_kernel")),
    ("factor", ("potrf_mega_kernel", "mega_status_kernel", "chol_update_kernel<128, 0>", "chol_update_kernel<64", "chol_diag")),
    ("eval", ("eval_fused_kernel", "eval_fused_split_kernel", "eval_combine_kernel", "eval_rows_kernel", "jac_assemble_kernel", "center_pad")),
    ("solve", ("chol_backsolve_kernel", "backsolve_persistent_kernel", "premul_kernel", "get_rhs_rows", "scatter_solution", "finish_lambda", "residual_kernel", "max_abs")),
]


FUSED = False  # set when the trace holds the fit's fused assembly kernel: the full-matrix kernel's dispatches then are bench.py's
               # side measurement of mrbf_gram (kernels.gram_full, seven launches per run), not part of the cycle
MEGA = False  # set when the trace holds the persistent factorisation: the blocked-Cholesky kernels then belong to the projection
              # (rank-2q update of K, Cholesky-QR of the tail basis), not to the factorisation


def phase_of(name):
    if FUSED and ("gram_mfma" in name or "gram_diff_kernel" in name):
        return "gram_full_side_measurement"
    if MEGA and ("chol_update_kernel" in name or "chol_diag" in name):
        return "project_misc"
    for ph, keys in PHASES:
        if any(k in name for k in keys):
            return ph
    return "project_misc"


def stats(db, out, inv):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "CallsPerCycle", "MsPerCycle"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 3), r[4], r[5], round(r[1] / inv, 2), round(r[2] / 1e6 / inv, 4)])
    print("wrote", out, "kernel time per cycle %.3f ms" % (tot / 1e6 / inv))


def pmc_sum(db, counter):
    con = sqlite3.connect(db)
    return con.execute("select name, count(*), sum(counter_value) from pmc_events where counter_name = ? group by name order by 3 desc", (counter,)).fetchall()


def pmc(wdb, fdb, prefix, inv, config):
    global MEGA, FUSED
    MEGA = any("potrf_mega_kernel" in r[0] for r in pmc_sum(wdb, "WRITE_SIZE"))
    FUSED = any("gram_w_kernel" in r[0] for r in pmc_sum(wdb, "WRITE_SIZE"))
    out = {}
    detail = {}
    for db, counter, tag, scale in ((wdb, "WRITE_SIZE", "write", 1.0), (fdb, "FETCH_SIZE", "fetch", 2.0)):
        rows = pmc_sum(db, counter)
        with open(f"{prefix}_{tag}_summary.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Dispatches", f"{counter}_KB_sum", "BytesPerCycle" + ("_x2_gfx950" if scale == 2.0 else ""), "Phase"])
            for name, cnt, val in rows:
                b = val * 1024.0 * scale / inv
                ph = phase_of(name)
                w.writerow([name, cnt, val, round(b, 1), ph])
                detail.setdefault(ph, {"write_bytes": 0.0, "fetch_bytes_corrected": 0.0})
                detail[ph]["write_bytes" if tag == "write" else "fetch_bytes_corrected"] += b
    for ph, d in detail.items():
        d["total"] = d["write_bytes"] + d["fetch_bytes_corrected"]
        out[ph] = d["total"]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[config] = out
    cur["_note"] = ("HBM-side bytes per cycle (gram, factor: one launch per cycle) from rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in separate "
                    "passes; counter unit KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)")
    cur.setdefault("_detail", {})
    cur["_detail"][config] = detail
    json.dump(cur, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


def timeline(db, which):
    """Kernels of fit number `which` (from the Gram kernel of that cycle to the end of its factorisation), offsets in us."""
    con = sqlite3.connect(db)
    cols = [c[0] for c in con.execute("select * from kernels limit 1").description]
    qcol = next((c for c in ("stream_id", "queue_id", "queue") if c in cols), None)
    rows = con.execute("select name, start, end%s from kernels order by start" % (", " + qcol if qcol else "")).fetchall()
    starts = [i for i, r in enumerate(rows) if "gram_w_kernel" in r[0] or "gram_mfma" in r[0] or "small_mean_kernel" in r[0]]
    if not starts:
        print("no gram kernel in trace; columns:", cols)
        return
    i0 = starts[min(which, len(starts) - 1)]
    # go back to the first kernel of the side chain of this fit (after the previous potrf_mega / eval)
    j = i0
    while j > 0 and rows[i0][1] - rows[j - 1][1] < 400000 and "potrf_mega" not in rows[j - 1][0] and "eval_" not in rows[j - 1][0]:
        j -= 1
    t0 = rows[j][1]
    for r in rows[j:]:
        print("%9.1f %8.1f  q=%s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "-", r[0][:90]))
        if "potrf_mega_kernel" in r[0] or (r[1] - t0) > 6e6:
            break


if __name__ == "__main__":
    if sys.argv[1] == "timeline":
        timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 1.0)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5]), sys.argv[6] if len(sys.argv) > 6 else "C3")
