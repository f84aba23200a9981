#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/gpu_bandonly.py into the per-launch HBM traffic
of the dominant pyramid kernel (band2_kernel: levels 0+1 in one pass; with FVVDP_BAND_FUSE=0 the one-level band_kernel at
level 0), with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced reads -> x2; WRITE_SIZE as reported; both in KiB).
usage: pmc_level0.py <fetch.db> <write.db> <out.json>"""
import json, sqlite3, sys

W, H, N = 3840, 2160, 60
px = [W * H, (W // 2) * (H // 2), (W // 4) * (H // 4)]


def dominant(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, sum(value), max(duration), max(start), kernel_name from counters_collection "
                       "where (kernel_name like '%band2_kernel%' or kernel_name like '%band_kernel%') and counter_name = ? "
                       "group by dispatch_id order by max(start)", (counter,)).fetchall()
    b2 = [r for r in rows if "band2_kernel" in r[4]]
    if b2:                      # two-level kernel: at 4K levels 0+1 and levels 2+3 take it -> the large dispatches are the dominant launch
        mx = max(r[1] for r in b2)
        b2 = [r for r in b2 if r[1] > 0.5 * mx]
        vals, durs, fused = [r[1] for r in b2], [r[2] for r in b2], True
    else:                       # one-level kernels in level order 0..6 per batch
        n = len(rows) // 7
        vals, durs, fused = [rows[7 * i][1] for i in range(n)], [rows[7 * i][2] for i in range(n)], False
    return sum(vals) / len(vals), sum(durs) / len(durs) / 1e3, len(vals), fused


fetch_kib, dur_f, n1, fused = dominant(sys.argv[1], "FETCH_SIZE")
write_kib, dur_w, n2, _ = dominant(sys.argv[2], "WRITE_SIZE")
alg = 16.0 * ((px[0] + px[1]) + ((px[1] + px[2]) if fused else 0)) * N
moved = 16.0 * (px[0] + (px[2] if fused else px[1])) * N
out = {"kernel": ("band2_kernel<4> levels 0+1" if fused else "band_kernel<4> level 0") + ", 60 frames per launch (3840x2160)",
       "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
       "read_bytes": 2.0 * fetch_kib * 1024, "write_bytes": write_kib * 1024,
       "traffic_bytes": 2.0 * fetch_kib * 1024 + write_kib * 1024,
       "algorithmic_bytes": alg,
       "algorithmic_note": "SURVEY 8(d) streaming-pyramid figure of the levels the launch covers (every level read once, the "
                           "next written once)",
       "compulsory_bytes_of_this_kernel": moved,
       "compulsory_note": "what the launch has to move: level 0 read once + the level it writes (level 2 for the two-level "
                          "kernel, whose level 1 never leaves the chip); strips overlap by 20 of 128 columns (halo re-reads)",
       "launches_averaged": [n1, n2], "avg_duration_us_under_pmc": [dur_f, dur_w],
       "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE x1, KiB units"}
def temporal(db, counter):
    """the temporal kernel's dispatches of the same pass (present when the target ran whole predict() calls: STAGE=all)"""
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, sum(value), max(duration) from counters_collection where kernel_name like '%temporal_vec_kernel%' "
                       "and counter_name = ? group by dispatch_id", (counter,)).fetchall()
    return (sum(r[1] for r in rows) / len(rows), sum(r[2] for r in rows) / len(rows) / 1e3, len(rows)) if rows else None


kf, kw = temporal(sys.argv[1], "FETCH_SIZE"), temporal(sys.argv[2], "WRITE_SIZE")
if kf and kw:
    out["k1"] = {"kernel": "temporal_vec_kernel<8,4,u8>, 60 output frames per launch (3840x2160, 67 source frames read)",
                 "read_bytes": 2.0 * kf[0] * 1024, "write_bytes": kw[0] * 1024, "traffic_bytes": 2.0 * kf[0] * 1024 + kw[0] * 1024,
                 "algorithmic_bytes": (6.0 + 16.0) * px[0] * N, "history_bytes": 6.0 * px[0] * 7,
                 "launches_averaged": [kf[2], kw[2]], "avg_duration_us_under_pmc": [kf[1], kw[1]]}
out["traffic_over_algorithmic"] = out["traffic_bytes"] / alg
out["traffic_over_compulsory"] = out["traffic_bytes"] / moved
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
