#!/usr/bin/env python3
"""Condense rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into small text summaries for profiles/.

usage: rocprof_summarize.py <dir-with-results.db> [more dirs...]  > profiles/<name>.txt
Kernel-trace runs produce the per-kernel --stats table; --pmc runs produce per-kernel counter sums.
FETCH_SIZE / WRITE_SIZE are reported raw (KB) and, for FETCH_SIZE, with the gfx950 x2 correction from
/opt/skills/guides/MI355X_MICROARCH.md (HBM section).
"""
import glob
import os
import sqlite3
import sys


def short(name):
    return name if len(name) < 90 else name[:87] + "..."


def main():
    for d in sys.argv[1:]:
        for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
            con = sqlite3.connect(db)
            print("== %s" % os.path.relpath(db))
            rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
            if rows:
                print("-- kernel trace stats (durations in us, as the rocpd top_kernels view reports them)")
                print("%-92s %6s %16s %16s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
                for r in rows:
                    if r[4] < 0.0005:
                        continue
                    print("%-92s %6d %16.0f %16.0f %8.3f" % (short(r[0]), r[1], r[2], r[3], r[4]))
                geo = con.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, "
                                  "scratch_size from kernels where name like '%phe%' or name like '%k_decrypt%' "
                                  "group by name").fetchall()
                for g in geo:
                    print("   geometry %s: grid=%s wg=%s lds=%s vgpr=%s agpr=%s sgpr=%s scratch=%s" % ((short(g[0]),) + g[1:]))
                # the same kernel serves several legs (decrypt halves / key-owner halves, full batches / warm launches):
                # one row per (kernel, grid size), so that a leg's average launch time can be read off on its own
                try:
                    cols = [c[1] for c in con.execute("pragma table_info(kernels)").fetchall()]
                    dur = "duration" if "duration" in cols else "(end - start)"
                    per = con.execute("select name, grid_x, count(*), avg(%s), sum(%s) from kernels where name like '%%phe%%' or "
                                      "name like '%%k_decrypt%%' group by name, grid_x order by sum(%s) desc" % (dur, dur, dur)).fetchall()
                    print("-- per (kernel, grid) launch times (ns -> us)")
                    for r in per:
                        print("   %-86s grid=%-9s calls=%-5d avg_us=%14.1f total_us=%14.1f" % (short(r[0]), r[1], r[2], r[3] / 1e3, r[4] / 1e3))
                except sqlite3.OperationalError as exc:
                    print("   (no per-grid breakdown: %s; columns of `kernels`: %s)" % (exc, cols if "cols" in dir() else "?"))
            try:
                rows = con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                                   "group by kernel_name, counter_name").fetchall()
            except sqlite3.OperationalError:
                rows = []
            if not os.environ.get("PHE_SUMMARIZE_ALL"):               # (the microbenchmark's kernels are not in namespace phe)
                rows = [r for r in rows if "phe" in r[0] or "k_decrypt" in r[0]]
            if rows:
                print("-- PMC counters (summed over dispatches and over XCDs/SEs as rocprofv3 reports them)")
                for r in rows:
                    extra = ""
                    if r[1] == "FETCH_SIZE":
                        extra = "  KB raw; x2 gfx950 correction => %.1f MB" % (r[2] * 2 / 1024)
                    if r[1] == "WRITE_SIZE":
                        extra = "  KB raw => %.1f MB" % (r[2] / 1024)
                    print("%-60s %-22s %20.0f  dispatches=%d%s" % (short(r[0])[:60], r[1], r[2], r[3], extra))
            con.close()


if __name__ == "__main__":
    main()
