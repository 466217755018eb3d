#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace and/or PMC counters) as text.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_x_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    stats = {}
    for name, st, en in rows:
        d = (en - st) / 1e3
        s = stats.setdefault(name, [0, 0.0, 1e30, 0.0])
        s[0] += 1; s[1] += d; s[2] = min(s[2], d); s[3] = max(s[3], d)
    total = sum(s[1] for s in stats.values()) or 1.0
    print("# rocprofv3 --kernel-trace summary of %s" % path)
    print("%-90s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, s in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print("%-90s %8d %14.1f %12.1f %12.1f %12.1f %6.2f%%" % (name[:90], s[0], s[1], s[1] / s[0], s[2], s[3], 100 * s[1] / total))
    try:
        pm = cur.execute("select * from counters_collection limit 1").fetchall()
        if pm:
            ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            print("\n# PMC counters (columns: %s)" % ", ".join(ccols))
            kn = [c for c in ccols if "kernel" in c and "name" in c] or [c for c in ccols if c == "name"]
            cn = [c for c in ccols if "counter_name" in c or c == "counter"]
            vn = [c for c in ccols if "value" in c]
            if kn and cn and vn:
                q = "select %s, %s, count(*), sum(%s), avg(%s) from counters_collection group by 1, 2" % (kn[0], cn[0], vn[0], vn[0])
                print("%-70s %-28s %8s %20s %20s" % ("kernel", "counter", "n", "sum", "avg_per_dispatch"))
                for r in cur.execute(q):
                    print("%-70s %-28s %8d %20.1f %20.1f" % (str(r[0])[:70], r[1], r[2], r[3], r[4]))
    except sqlite3.Error as e:
        print("# no counter data:", e)


if __name__ == "__main__":
    main(sys.argv[1])
