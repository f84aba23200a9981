#!/usr/bin/env python3
"""Per-kernel PMC counter averages from a rocprofv3 rocpd database (counter instances summed per dispatch).
usage: pmc_summary.py <db> [kernel-name-substring]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
q = """select kernel_name, grid_size, counter_name, avg(v), count(*), avg(d) from
       (select kernel_name, grid_size, counter_name, dispatch_id, sum(value) as v, max(duration) as d
        from counters_collection group by dispatch_id, counter_name) group by kernel_name, grid_size, counter_name"""
for r in con.execute(q).fetchall():
    if sub in r[0]:
        print("%-44s grid %-9s %-22s avg %.6g (n=%d) dur_us %.1f" % (r[0][:44], r[1], r[2], r[3], r[4], (r[5] or 0) / 1e3))
