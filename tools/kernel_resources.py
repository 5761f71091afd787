#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch figures of a built libsdr_amd.so, read from the metadata notes of the
gfx950 code objects bundled in it.
usage: kernel_resources.py lib.so [substring-of-demangled-name ...]"""
import os
import re
import subprocess
import sys
import tempfile

KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count")


def kernels_of(notes):
    """the notes list every kernel as a YAML block whose first key is '- .agpr_count'"""
    res, block = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s*(- )?\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        first, key, val = m.group(1), m.group(2), m.group(3).strip("'\" ")
        if first and key == "agpr_count":
            if "name" in block:
                res[block["name"]] = block
            block = {}
        if key == "name":
            if val.startswith("_Z"):
                block.setdefault("name", val)
        elif key in KEYS:
            block[key] = val
    if "name" in block:
        res[block["name"]] = block
    return res


def main():
    lib, pats = sys.argv[1], sys.argv[2:]
    data = open(lib, "rb").read()
    out = {}
    for m in re.finditer(b"\x7fELF\x02\x01\x01", data):
        off = m.start()
        if data[off + 18:off + 20] != b"\xe0\x00":      # e_machine = EM_AMDGPU
            continue
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(data[off:off + (64 << 20)])
            path = f.name
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
        os.unlink(path)
        out.update(kernels_of(notes))
    names = sorted(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for n, d in zip(names, dem):
        if pats and not all(p in d for p in pats):
            continue
        r = out[n]
        d = d.replace("sdr::", "").replace("(RxParams)", "")
        print(f"{d}: vgpr {r.get('vgpr_count')} agpr {r.get('agpr_count')} sgpr {r.get('sgpr_count')} lds {r.get('group_segment_fixed_size')} scratch {r.get('private_segment_fixed_size')} spilled {r.get('vgpr_spill_count')}")


if __name__ == "__main__":
    main()
