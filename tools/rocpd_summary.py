#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel table kept under profiles/.
usage: tools/rocpd_summary.py <results.db> [--band-levels N] [--only substr] [--skip K] [--dispatches substr]

Columns: calls, total, avg / median / min / max over ALL dispatches, and `steady med` = the median after dropping the first
K dispatches of every kernel (--skip K, default 3: the launches that run while clocks and caches are cold).  The roofline
numbers of DESIGN.md / bench.py are quoted on medians; the average is kept so that warm-up and disturbed launches stay visible.
With --band-levels N the dispatches of band_kernel are labelled by pyramid level (they are issued in level order
0..N-1 for every batch), so the dominant level-0 launch gets its own row.
--dispatches substr additionally lists every dispatch of the kernels whose label contains substr, in launch order, with the
kernel that ran immediately before it (a launch behind a host->device copy or a different kernel mix shows up here)."""
import sqlite3
import sys


def median(v):
    v = sorted(v)
    n = len(v)
    return 0.0 if n == 0 else (v[n // 2] if n & 1 else 0.5 * (v[n // 2 - 1] + v[n // 2]))


def opt(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def main():
    db = sys.argv[1]
    nlev = int(opt("--band-levels", 0))
    only = opt("--only", "")
    skip = int(opt("--skip", 3))
    disp = opt("--dispatches", None)
    con = sqlite3.connect(db)
    rows = con.execute("select name, duration, vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x, start "
                       "from kernels order by start").fetchall()
    agg, seq = {}, 0
    order = []
    prev = "(first kernel)"
    for name, dur, vg, sg, lds, gx, wg, start in rows:
        label = name
        if nlev and "band2_kernel" in name:          # two levels per launch: counts as two positions of the level order
            label = "%s [levels %d+%d]" % (name, seq % nlev, seq % nlev + 1)
            seq += 2
        elif nlev and "band_kernel" in name:
            label = "%s [level %d]" % (name, seq % nlev)
            seq += 1
        a = agg.setdefault(label, {"d": [], "vg": vg, "sg": sg, "lds": lds, "gx": gx, "wg": wg})
        a["d"].append(dur)
        order.append((label, dur, start, prev))
        prev = name
    total = sum(sum(a["d"]) for a in agg.values()) or 1
    # the two register columns are what rocprofv3 records per dispatch (arch VGPRs in use as it counts them, SGPRs); they are NOT the
    # allocation that decides occupancy -- that is the code-object figure (tools/codeobj_report.py, "Registers, spills ..." table of
    # the bundle: e.g. band2_kernel<4, true> 156, temporal_vec_kernel<8, 4, 0, 1> 105)
    print("| kernel | calls | total ms | avg us | median us | steady med us (first %d dropped) | min us | max us | %% | vgpr (rocprof's count, not the allocation) | sgpr | lds B | grid_x | wg |" % skip)
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for label, a in sorted(agg.items(), key=lambda kv: -sum(kv[1]["d"])):
        if only and only not in label:
            continue
        d = a["d"]
        nm = label if len(label) < 110 else label[:107] + "..."
        steady = d[skip:] if len(d) > skip else d
        print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s |" % (
            nm, len(d), sum(d) / 1e6, sum(d) / len(d) / 1e3, median(d) / 1e3, median(steady) / 1e3, min(d) / 1e3, max(d) / 1e3,
            100.0 * sum(d) / total, a["vg"], a["sg"], a["lds"], a["gx"], a["wg"]))
    if disp:
        print()
        print("dispatches of `%s` in launch order (us; the kernel that ran before each):" % disp)
        t0 = rows[0][7] if rows else 0
        k = 0
        for label, dur, start, before in order:
            if disp in label:
                print("  #%-3d t=%10.1f ms  dur %9.1f  after %s" % (k, (start - t0) / 1e6, dur / 1e3, before[:60]))
                k += 1


if __name__ == "__main__":
    main()
