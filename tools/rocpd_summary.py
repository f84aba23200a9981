#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel table kept under profiles/.
usage: tools/rocpd_summary.py <results.db> [--band-levels N] [--only substr]
With --band-levels N the dispatches of band_kernel are labelled by pyramid level (they are issued in level order
0..N-1 for every batch), so the dominant level-0 launch gets its own row."""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    nlev = int(sys.argv[sys.argv.index("--band-levels") + 1]) if "--band-levels" in sys.argv else 0
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""
    con = sqlite3.connect(db)
    rows = con.execute("select name, duration, vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x, start "
                       "from kernels order by start").fetchall()
    agg, seq = {}, 0
    for name, dur, vg, sg, lds, gx, wg, _ in rows:
        label = name
        if nlev and "band2_kernel" in name:          # two levels per launch: counts as two positions of the level order
            label = "%s [levels %d+%d]" % (name, seq % nlev, seq % nlev + 1)
            seq += 2
        elif nlev and "band_kernel" in name:
            label = "%s [level %d]" % (name, seq % nlev)
            seq += 1
        a = agg.setdefault(label, [0, 0.0, 1e30, 0.0, vg, sg, lds, gx, wg])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values()) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | grid_x | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for label, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if only and only not in label:
            continue
        nm = label if len(label) < 110 else label[:107] + "..."
        print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s |" % (
            nm, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total, a[4], a[5], a[6], a[7], a[8]))


if __name__ == "__main__":
    main()
