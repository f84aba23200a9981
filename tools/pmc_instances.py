#!/usr/bin/env python3
"""Round 5: per-INSTANCE counter values (the L2 channels of the 8 XCDs) of the temporal kernel from a rocprofv3 --pmc run with
--output-format json (the sqlite view sums the instances).  Per measured dispatch of tools/microbench/k1_stream.hip in "pmc" mode
(= per destination buffer): duration, and per counter the sum, the largest instance, max / mean and the spread per XCD.
usage: pmc_instances.py results.json stream-log [kernel-substring]"""
import json
import re
import sys


def walk_dims(rec):
    """dimension values of a counter record, whatever this rocprofv3 build calls them"""
    for key in ("dimensions", "dims", "dimension"):
        if key in rec:
            return rec[key]
    return None


def main():
    path, log = sys.argv[1], sys.argv[2]
    pat = sys.argv[3] if len(sys.argv) > 3 else "temporal_vec_kernel"
    kinds = []
    for line in open(log):
        m = re.match(r"^(\d+)\s+(\S+)\s+(0x[0-9a-f]+)", line)
        if m:
            kinds.append(m.group(2))
    d = json.load(open(path))
    root = d["rocprofiler-sdk-tool"]
    root = root[0] if isinstance(root, list) else root
    print("top-level keys:", sorted(root.keys()))
    cc = root.get("callback_records", {}).get("counter_collection") or root.get("buffer_records", {}).get("counter_collection")
    print("counter_collection records:", len(cc))
    if cc:
        print("one record:", json.dumps(cc[0])[:1500])
    # names of kernels and counters
    ksym = {}
    for k in root.get("kernel_symbols", []):
        ksym[k.get("kernel_id")] = k.get("formatted_kernel_name") or k.get("kernel_name")
    cinfo = {}
    for agent in root.get("counters", []):
        cinfo[agent.get("id", {}).get("handle", agent.get("id"))] = agent
    print("counters described:", len(cinfo), "example:", json.dumps(root.get("counters", [None])[0])[:800])
    rows = []
    for rec in cc:
        info = rec.get("dispatch_data", {}).get("dispatch_info", {})
        name = ksym.get(info.get("kernel_id"), "?")
        if pat not in str(name):
            continue
        vals = {}
        for r in rec.get("records", []):
            cid = r.get("counter_id", {}).get("handle", r.get("counter_id"))
            vals.setdefault(cid, []).append(r.get("value", 0.0))
        rows.append((info.get("dispatch_id"), rec.get("dispatch_data", {}).get("start_timestamp"), rec.get("dispatch_data", {}).get("end_timestamp"), vals))
    rows.sort(key=lambda t: t[0])
    measured = rows[1::2]
    print()
    print("| buf | kind | us/frame | counter | instances | sum | max | max/mean | per-XCD sums (if 128 instances: 16 consecutive each) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for k, (did, t0, t1, vals) in enumerate(measured):
        dur = ((t1 or 0) - (t0 or 0)) / 1e3 / 60
        for cid, v in sorted(vals.items()):
            nm = cinfo.get(cid, {}).get("name", str(cid))
            s, mx = sum(v), max(v)
            mean = s / len(v)
            per = ""
            if len(v) % 8 == 0 and len(v) >= 16:
                g = len(v) // 8
                per = " ".join("%.3g" % sum(v[i * g:(i + 1) * g]) for i in range(8))
            print("| %d | %s | %.2f | %s | %d | %.4g | %.4g | %.3f | %s |" % (k, kinds[k] if k < len(kinds) else "?", dur, nm, len(v), s, mx, mx / mean if mean else 0, per))
        if k < 3:
            for cid, v in sorted(vals.items()):
                print("|  |  |  | all instances of %s | | %s |" % (cinfo.get(cid, {}).get("name", str(cid)), " ".join("%.3g" % x for x in v)))


if __name__ == "__main__":
    main()
