#!/usr/bin/env python
"""Dump the per-kernel summary of a rocprofv3 rocpd (.db) result as markdown:
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.md"""
import sqlite3
import sys


def main(path, top=25):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for name, calls, tot, avg, pct in rows[:top]:
        short = name if len(name) < 90 else name[:87] + "..."
        print("| `%s` | %d | %.3f | %.2f | %.2f |" % (short, calls, tot / 1e3, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1])
