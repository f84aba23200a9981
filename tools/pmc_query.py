import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
rows = con.execute("select counter_name, sum(value), count(distinct dispatch_id), max(duration) from counters_collection where kernel_name like ? group by counter_name", ('%'+pat+'%',)).fetchall()
for r in rows: print("%-28s total %.4g  dispatches %d  per-dispatch %.4g  dur_us %.1f" % (r[0], r[1], r[2], r[1]/r[2], r[3]/1e3))
