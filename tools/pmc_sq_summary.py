#!/usr/bin/env python3
"""SQ / TCC counter summary of rocprofv3 --pmc passes as a markdown table (kept under profiles/).
usage: pmc_sq_summary.py <kernel-substring> <results.db> [<results.db> ...]   (one db per --pmc pass; counters are merged)

Per kernel (name + grid size), averaged over its dispatches: every raw counter, and the ratios that say what bounds the kernel
(units: SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves, MI355X_MICROARCH.md):
  valu = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   share of a wave's resident time spent issuing VALU work; x waves per SIMD = how
                                                busy the SIMD's VALU is (>= ~0.9: VALU-bound)
  wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES           share parked in s_waitcnt / barriers (memory or LDS latency not hidden)
  stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES     share stalled at issue (pipe busy: another wave owns the VALU / LDS)
  lds = SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES, conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (or / SQ_ACTIVE_INST_LDS)
FETCH_SIZE is doubled (gfx950: 128-byte requests tallied at 64 B), WRITE_SIZE as reported, both KiB -> MB."""
import sqlite3
import sys


def main():
    pat = sys.argv[1]
    acc = {}
    for db in sys.argv[2:]:
        con = sqlite3.connect(db)
        rows = con.execute("select kernel_name, grid_size, counter_name, sum(value), max(duration), dispatch_id "
                           "from counters_collection where kernel_name like ? group by dispatch_id, counter_name",
                           ('%' + pat + '%',)).fetchall() if has_col(con, "grid_size") else \
            con.execute("select kernel_name, 0, counter_name, sum(value), max(duration), dispatch_id "
                        "from counters_collection where kernel_name like ? group by dispatch_id, counter_name",
                        ('%' + pat + '%',)).fetchall()
        for name, grid, ctr, val, dur, did in rows:
            k = (name, grid)
            a = acc.setdefault(k, {})
            c = a.setdefault(ctr, [0.0, 0, 0.0])
            c[0] += val
            c[1] += 1
            c[2] += dur
    for (name, grid), a in sorted(acc.items(), key=lambda kv: -max(c[2] / c[1] for c in kv[1].values())):
        avg = {ctr: c[0] / c[1] for ctr, c in a.items()}
        dur = max(c[2] / c[1] for c in a.values()) / 1e3
        n = max(c[1] for c in a.values())
        print("### `%s`  grid %s  (%d dispatches per pass, avg %.1f us under the counters)" % (name[:100], grid, n, dur))
        print()
        print("| counter | per dispatch |")
        print("|---|---|")
        for ctr in sorted(avg):
            print("| %s | %.4g |" % (ctr, avg[ctr]))
        wc = avg.get("SQ_WAVE_CYCLES")
        d = []
        if wc:
            for label, ctr in (("valu", "SQ_ACTIVE_INST_VALU"), ("any-inst", "SQ_ACTIVE_INST_ANY"), ("wait", "SQ_WAIT_ANY"),
                               ("stall", "SQ_WAIT_INST_ANY"), ("lds", "SQ_ACTIVE_INST_LDS"), ("salu", "SQ_ACTIVE_INST_SCA"),
                               ("vmem", "SQ_ACTIVE_INST_VMEM"), ("lds-stall", "SQ_WAIT_INST_LDS")):
                if ctr in avg:
                    d.append("%s %.3f" % (label, avg[ctr] / wc))
            if "SQ_BUSY_CYCLES" in avg:
                d.append("waves resident per SQ-busy cycle (WAVE_CYCLES / BUSY_CYCLES) %.2f" % (wc / avg["SQ_BUSY_CYCLES"]))
        if "SQ_INSTS_VALU" in avg and "SQ_ACTIVE_INST_VALU" in avg:
            d.append("quad-cycles per VALU instruction %.2f" % (avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_INSTS_VALU"]))
        if "SQ_LDS_BANK_CONFLICT" in avg:
            den = avg.get("SQ_LDS_IDX_ACTIVE") or avg.get("SQ_ACTIVE_INST_LDS")
            if den:
                d.append("LDS bank-conflict cycles / LDS active cycles %.3f" % (avg["SQ_LDS_BANK_CONFLICT"] / den))
        if "FETCH_SIZE" in avg:
            d.append("HBM read %.1f MB" % (2.0 * avg["FETCH_SIZE"] * 1024 / 1e6))
        if "WRITE_SIZE" in avg:
            d.append("HBM write %.1f MB" % (avg["WRITE_SIZE"] * 1024 / 1e6))
        print()
        print("shares of the waves' resident time: " + "; ".join(d))
        print()


def has_col(con, col):
    try:
        con.execute("select %s from counters_collection limit 1" % col)
        return True
    except sqlite3.Error:
        return False


if __name__ == "__main__":
    main()
