#!/usr/bin/env python
"""
Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) result: per-kernel launch count / total / avg / min / max
duration (the `--kernel-trace --stats` table) and, when the run collected PMC counters, per-kernel average
counter values.  Writes a small text file that is committed under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_kt/bench_results.db [more.db ...] > profiles/r01_xxx.txt
"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f"# {path}")
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        name_col = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else cols[0])
        rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), "
                           f"max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for name, calls, tot, avg, mn, mx in rows:
            print(f"{name[:70]:70s} {calls:6d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} "
                  f"{100.0 * tot / total:6.2f}")
        try:
            pcols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            if pcols:
                kn = "kernel_name" if "kernel_name" in pcols else "name"
                rows = cur.execute(f"select {kn}, counter_name, count(*), avg(value), sum(value) from "
                                   f"counters_collection group by {kn}, counter_name order by 1, 2").fetchall()
                if rows:
                    print(f"\n{'kernel':70s} {'counter':>14s} {'dispatches':>10s} {'avg_per_dispatch':>18s}")
                    for name, counter, n, avg, tot in rows:
                        print(f"{name[:70]:70s} {counter:>14s} {n:10d} {avg:18.1f}")
        except sqlite3.Error as e:
            print("# no counter table:", e)
        print()


if __name__ == "__main__":
    main()
