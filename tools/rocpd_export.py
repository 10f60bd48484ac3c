#!/usr/bin/env python
"""Export the views of a rocprofv3 rocpd SQLite result (the default output format of ROCm 7.x) as CSV.

usage: python tools/rocpd_export.py stats   <results.db> <out.csv>   # per-kernel totals (the --stats summary)
       python tools/rocpd_export.py counter <results.db> <out.csv>   # one row per (dispatch, counter), the layout
                                                                     # tools/pmc_summary.py reads"""
import csv
import sqlite3
import sys


def main():
    what, db, out = sys.argv[1:4]
    con = sqlite3.connect(db)
    cur = con.cursor()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        if what == "stats":
            w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
            for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
                w.writerow(r)
        else:
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            for r in cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
                w.writerow(r)


if __name__ == "__main__":
    main()
