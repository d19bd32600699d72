#!/usr/bin/env python3
"""Print the per-kernel summary (calls, total us, average us, % of GPU time) of a rocprofv3 run stored in the
rocpd sqlite format (rocprofv3 --kernel-trace --stats -d DIR -o NAME  ->  DIR/NAME_results.db) as CSV."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("Name,Calls,TotalDurationUs,AverageUs,Percentage")
    for name, calls, total, avg, pct in rows:
        print(f"\"{name}\",{calls},{total:.1f},{avg:.2f},{pct:.3f}")


if __name__ == "__main__":
    main()
