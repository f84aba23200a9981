#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel table kept under profiles/.
usage: tools/rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | grid_x | wg |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s |" % (
            name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10]))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(out)
    print(out)


if __name__ == "__main__":
    main()
