#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel in the built libfvvdp_hip.so (the compiler's own code-object metadata).
usage: tools/codeobj_report.py [lib.so] [--only substr] [--spills]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import codeobj  # noqa: E402

only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--only"]
lib = args[0] if args else os.path.join(ROOT, "fovvideovdp_amd", "libfvvdp_hip.so")
md = codeobj.kernel_metadata(lib)
names = list(md)
dm = codeobj.demangle(names)
print("| kernel | vgpr | agpr | sgpr | sgpr spills | vgpr spills | scratch B | LDS B |")
print("|---|---|---|---|---|---|---|---|")
for d, n in sorted(zip(dm, names)):
    m = md[n]
    if only and only not in d:
        continue
    if "--spills" in sys.argv and not (m.get("sgpr_spill_count") or m.get("vgpr_spill_count") or m.get("private_segment_fixed_size")):
        continue
    print("| `%s` | %s | %s | %s | %s | %s | %s | %s |" % (d[:100], m.get("vgpr_count"), m.get("agpr_count"), m.get("sgpr_count"),
          m.get("sgpr_spill_count"), m.get("vgpr_spill_count"), m.get("private_segment_fixed_size"), m.get("group_segment_fixed_size")))
