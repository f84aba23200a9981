#!/usr/bin/env python3
"""Round 5: per-buffer counters of the temporal kernel from rocprofv3 --pmc passes of tools/microbench/k1_stream.hip ("pmc" mode: per
destination buffer one warm and one measured dispatch of the real kernel, then the same of the plain replay).  Every pass is its own
process = its own draw of allocations, so every pass gets its own table: one row per buffer (the measured dispatch), the duration
under the counters and, per counter, the sum over its instances, the largest instance and max/mean (imbalance); below each table the
correlation of every column with the duration across the buffers of that pass.
usage: pmc_k1_mode.py <kernel-substring> <stream-log-of-the-pass>:<results.db> ..."""
import math
import re
import sqlite3
import sys


def corr(x, y):
    n = len(x)
    if n < 3:
        return 0.0
    mx, my = sum(x) / n, sum(y) / n
    sxx = sum((a - mx) ** 2 for a in x)
    syy = sum((b - my) ** 2 for b in y)
    if sxx <= 0 or syy <= 0:
        return 0.0
    return sum((a - mx) * (b - my) for a, b in zip(x, y)) / math.sqrt(sxx * syy)


def main():
    pat = sys.argv[1]
    for spec in sys.argv[2:]:
        log, db = spec.split(":")
        kinds = []
        try:
            for line in open(log):
                m = re.match(r"^(\d+)\s+(\S+)\s+(0x[0-9a-f]+)", line)
                if m:
                    kinds.append((m.group(2), m.group(3)))
        except OSError:
            pass
        con = sqlite3.connect(db)
        rows = con.execute("select dispatch_id, counter_name, value, duration from counters_collection where kernel_name like ? "
                           "order by dispatch_id", ('%' + pat + '%',)).fetchall()
        disp = {}
        for d, name, val, dur in rows:
            e = disp.setdefault(d, {"dur": dur / 1e3, "c": {}})
            e["c"].setdefault(name, []).append(val)
        ids = sorted(disp)
        measured = ids[1::2]                      # warm, measured, warm, measured ...
        names = sorted({n for d in measured for n in disp[d]["c"]})
        print("### pass `%s` (%d dispatches of `%s`, %d counters: %s)" % (db.split("/")[-3] if db.count("/") >= 3 else db, len(ids), pat, len(names), " ".join(names)))
        print()
        hdr = ["buf", "kind", "us/frame"]
        for n in names:
            hdr += [n + " sum", "max inst", "max/mean"]
        print("| " + " | ".join(hdr) + " |")
        print("|" + "---|" * len(hdr))
        cols = {h: [] for h in hdr[2:]}
        for k, d in enumerate(measured):
            e = disp[d]
            row = [str(k), kinds[k][0] if k < len(kinds) else "?", "%.2f" % (e["dur"] / 60)]
            cols["us/frame"].append(e["dur"] / 60)
            for n in names:
                v = e["c"].get(n, [0.0])
                s, mx = sum(v), max(v)
                mean = s / len(v) if v else 0.0
                row += ["%.4g" % s, "%.4g" % mx, "%.3f" % (mx / mean if mean > 0 else 0.0)]
                cols[n + " sum"].append(s)
                cols.setdefault(n + " max", []).append(mx)
                cols.setdefault(n + " imb", []).append(mx / mean if mean > 0 else 0.0)
            print("| " + " | ".join(row) + " |")
        print()
        t = cols["us/frame"]
        print("correlation with us/frame over the %d buffers (spread of us/frame: %.2f .. %.2f):" % (len(t), min(t) if t else 0, max(t) if t else 0))
        for n in names:
            print("  %-44s sum %+.3f   largest instance %+.3f   max/mean %+.3f   (instances per dispatch: %d)" %
                  (n, corr(t, cols[n + " sum"]), corr(t, cols[n + " max"]), corr(t, cols[n + " imb"]), len(disp[measured[0]]["c"].get(n, []))))
        print()


if __name__ == "__main__":
    main()
