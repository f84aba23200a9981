#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/gpu_bandonly.py into the per-launch HBM traffic
of the level-0 band kernel, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as
64 B for wide coalesced reads -> x2; WRITE_SIZE as reported; both in KiB).
usage: pmc_level0.py <fetch.db> <write.db> <out.json>"""
import json, sqlite3, sys


def level0(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, sum(value), max(duration), max(start) from counters_collection "
                       "where kernel_name like '%band_kernel%' and counter_name = ? group by dispatch_id order by max(start)",
                       (counter,)).fetchall()
    # dispatches come in level order 0..6 per batch; level 0 is the largest value of each group of 7
    vals = [r[1] for r in rows]
    n = len(vals) // 7
    l0 = [vals[7 * i] for i in range(n)]
    d0 = [rows[7 * i][2] for i in range(n)]
    return sum(l0) / len(l0), sum(d0) / len(d0) / 1e3, n


fetch_kib, dur_f, n1 = level0(sys.argv[1], "FETCH_SIZE")
write_kib, dur_w, n2 = level0(sys.argv[2], "WRITE_SIZE")
out = {"kernel": "band_kernel<4> level 0, 60 frames per launch (3840x2160)",
       "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
       "read_bytes": 2.0 * fetch_kib * 1024, "write_bytes": write_kib * 1024,
       "traffic_bytes": 2.0 * fetch_kib * 1024 + write_kib * 1024,
       "algorithmic_bytes": 16.0 * (3840 * 2160 + 1920 * 1080) * 60,
       "launches_averaged": [n1, n2], "avg_duration_us_under_pmc": [dur_f, dur_w],
       "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE x1, KiB units"}
out["overfetch_ratio"] = out["traffic_bytes"] / out["algorithmic_bytes"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
