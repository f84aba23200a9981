#!/usr/bin/env python3
"""Per-dispatch counter values and durations of one kernel from a rocprofv3 --pmc sqlite file, in dispatch order.
usage: pmc_per_dispatch.py results.db kernel-substring"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
rows = con.execute("select dispatch_id, counter_name, sum(value), max(duration) from counters_collection where kernel_name like ? "
                   "group by dispatch_id, counter_name order by dispatch_id", ('%' + pat + '%',)).fetchall()
cur, line = None, ""
for d, n, v, dur in rows:
    if d != cur:
        if line: print(line)
        cur, line = d, "dispatch %5d  dur %8.1f us" % (d, dur / 1e3)
    line += "  %s %.4g" % (n, v)
if line: print(line)
