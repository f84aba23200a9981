#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 (rocpd sqlite) trace: start offset, duration, idle gap before.
usage: tools/rocpd_timeline.py <results.db> [N]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = con.execute("select name, start, end from kernels order by start").fetchall()[-n:]
t0 = rows[0][1]
prev_end = None
for name, s, e in rows:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f us  dur %8.1f us  gap %7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name[:70]))
    prev_end = e
