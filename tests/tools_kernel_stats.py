"""Developer tool: per-kernel register / scratch / LDS usage of a built HIP shared library (no GPU needed).

    python tests/tools_kernel_stats.py [path/to/lib.so] [substring filter]

Reads the clang offload bundle in the .hip_fatbin section, extracts the gfx950 code object and prints the AMDGPU metadata
(vgpr / agpr / sgpr counts, private segment = scratch bytes per lane, static LDS) of every kernel."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = 0
    while True:
        base = data.find(magic, pos)
        if base < 0:
            return
        n, = struct.unpack_from("<Q", data, base + 24)
        off = base + 32
        for _ in range(n):
            eoff, esize, tlen = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24: off + 24 + tlen].decode()
            off += 24 + tlen
            if "gfx" in triple and esize > 0:
                yield triple, data[base + eoff: base + eoff + esize]
        pos = base + 24


def kernel_stats(path):
    out = []
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for block in txt.split("  - .agpr_count:")[1:]:
            block = ".agpr_count:" + block
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", block) or [None, "?"])[1]
            out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"),
                            scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"),
                            spill_v=g("vgpr_spill_count"), spill_s=g("sgpr_spill_count")))
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "polympc_amd", "libpolympc_amd.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in sorted(kernel_stats(path), key=lambda k: k["name"]):
        name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
        if flt and flt not in name:
            continue
        print(f"vgpr {k['vgpr']:>4} agpr {k['agpr']:>4} sgpr {k['sgpr']:>4} scratch {k['scratch']:>6} lds {k['lds']:>6} spill v/s {k['spill_v']}/{k['spill_s']}  {name[:150]}")
