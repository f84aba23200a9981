"""Reads the gfx950 code objects embedded in libfvvdp_hip.so and returns the per-kernel resource metadata the compiler
recorded (registers, spills, scratch, LDS).  Test / reporting helper: the library's `.hip_fatbin` section holds one clang
offload bundle per translation unit; each bundle entry for an amdgcn target is an ELF whose notes carry the metadata."""
import os
import re
import struct
import subprocess
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _section(path, name):
    out = subprocess.run([READELF, "-S", "-W", path], check=True, capture_output=True, text=True).stdout
    for line in out.splitlines():
        m = re.search(r"\]\s+%s\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)" % re.escape(name), line)
        if m:
            return int(m.group(2), 16), int(m.group(3), 16)
    raise RuntimeError("section %s not found in %s" % (name, path))


def code_objects(lib_path):
    """-> list of bytes objects, one per embedded amdgcn ELF"""
    off, size = _section(lib_path, ".hip_fatbin")
    with open(lib_path, "rb") as f:
        f.seek(off)
        blob = f.read(size)
    elfs = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            eoff, esize, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and esize > 0:
                elfs.append(blob[pos + eoff:pos + eoff + esize])
        pos = blob.find(MAGIC, pos + len(MAGIC))
    if not elfs:
        raise RuntimeError("no amdgcn code object found in %s (compressed bundle?)" % lib_path)
    return elfs


FIELDS = ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "agpr_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size", "wavefront_size")


def kernel_metadata(lib_path):
    """-> {demangled-ish kernel name: {field: int}} for every kernel of every code object in the library"""
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for i, elf in enumerate(code_objects(lib_path)):
            p = os.path.join(d, "co%d.elf" % i)
            with open(p, "wb") as f:
                f.write(elf)
            notes = subprocess.run([READELF, "--notes", p], check=True, capture_output=True, text=True).stdout
            cur = None                     # entries of amdhsa.kernels start with their first key (.agpr_count / .args)

            def close(c):
                if c and "name" in c:
                    res[c["name"]] = c

            for line in notes.splitlines():
                s = line.strip()
                if s.startswith("- .agpr_count:") or s.startswith("- .args:"):
                    close(cur)
                    cur = {}
                    s = s[2:]
                elif s.startswith("amdhsa.") and not s.startswith("amdhsa.kernels"):
                    close(cur)
                    cur = None
                    continue
                if cur is None:
                    continue
                m = re.match(r"\.(\w+):\s+(.*)$", s)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k in FIELDS:
                    try:
                        cur[k] = int(v)
                    except ValueError:
                        pass
                elif k == "name" and "name" not in cur and line.startswith("    .name") :
                    cur["name"] = v.strip("'\"")
            close(cur)
    return res


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return out.splitlines()
