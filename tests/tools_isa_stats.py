"""Developer tool: per-kernel register / scratch / instruction-class counts from a hipcc -save-temps ISA listing (*.s).
   python tests/tools_isa_stats.py /tmp/pmpc_model_cstr-hip-amdgcn-amd-amdhsa-gfx950.s [name-filter]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
def dem(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    dn = dem(name)
    if flt and flt not in dn:
        continue
    g = lambda k: (re.search(r'\.amdhsa_' + k + r'\s+(\S+)', body) or [None, None])[1]
    # body of the function: from "name:" to ".Lfunc_end"
    fm = re.search(r'^' + re.escape(name) + r':[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M)
    counts = {}
    if fm:
        for line in fm.group(1).splitlines():
            t = line.strip().split()
            if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
                continue
            op = t[0]
            cls = ("mfma" if op.startswith("v_mfma") else "accvgpr" if op.startswith("v_accvgpr") else "readlane" if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) else
                   "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
                   "scratch" if op.startswith("scratch_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other")
            counts[cls] = counts.get(cls, 0) + 1
    print(dn[:140])
    print("   vgpr", g("next_free_vgpr"), "accum_offset", g("accum_offset"), "sgpr", g("next_free_sgpr"), "scratch", g("private_segment_fixed_size"), "lds", g("group_segment_fixed_size"), counts)
