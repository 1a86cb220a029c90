"""The register budget of the built kernels is what their launch bounds promise (no GPU needed: the code objects inside the product library are
read with the ROCm binutils). A kernel that asks for two wavefronts per SIMD but was compiled to more than 256 registers runs at one — silently:
hipcc only prints a -Wpass-failed remark (this happened to the two-wave HBM-factor kernels when a shared out-of-line function was compiled for the
one-wave kernels' budget)."""
import os
import re
import subprocess
import tempfile

import pytest

import polympc_amd as pa

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", pa.LIB_PATH, fat])
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    out = []
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):
        bundle = os.path.join(tmp, f"b{n}.bin"); open(bundle, "wb").write(data[a:b])
        targets = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={bundle}"], capture_output=True, text=True).stdout.split()
        for t in targets:
            if "gfx950" in t:
                co = os.path.join(tmp, f"b{n}.co")
                subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={bundle}", f"--targets={t}", f"--output={co}"])
                out.append(co)
    return out


def _kernels(co):
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        agpr = int(blk.split()[0])
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        yield name, vgpr, agpr


@pytest.mark.skipif(not (os.path.exists(f"{LLVM}/clang-offload-bundler") and os.path.exists(f"{LLVM}/llvm-readelf")), reason="ROCm binutils not installed")
def test_two_wave_kernels_fit_half_the_register_file():
    assert os.path.exists(pa.LIB_PATH)
    pat = re.compile(r"sqp_kernel<pmpc::(\w+), (\d+), (\d+), (true|false), (\d+), (true|false), (true|false), (true|false), (true|false), (true|false)>")
    seen = {"reg1": 0, "big2": 0, "big1": 0, "reg2": 0}
    with tempfile.TemporaryDirectory() as tmp:
        objs = _code_objects(tmp)
        assert objs, "no gfx950 code object in the library"
        for co in objs:
            ks = list(_kernels(co))
            names = subprocess.run(["c++filt"], input="\n".join(k[0] for k in ks), capture_output=True, text=True).stdout.splitlines()
            for (mangled, vgpr, agpr), dn in zip(ks, names):
                m = pat.search(dn)
                if not m:
                    continue
                nn, mm, khbm, w2 = int(m.group(2)), int(m.group(3)), m.group(6) == "true", m.group(7) == "true"
                # (.vgpr_count is the unified count: architected registers + accumulation file)
                if nn > 0 and nn + mm <= 64:      # one KKT row per lane: PMPC_SQP_WAVES = 2
                    seen["reg1"] += 1
                    assert vgpr <= 256, f"{dn[:120]}: {vgpr} registers, two wavefronts per SIMD need <= 256"
                elif khbm and w2:                 # HBM-factor kernel, two wavefronts per SIMD
                    seen["big2"] += 1
                    assert vgpr <= 256, f"{dn[:120]}: {vgpr} registers, two wavefronts per SIMD need <= 256"
                elif khbm:
                    seen["big1"] += 1
                    assert vgpr <= 512
                elif nn > 0:
                    seen["reg2"] += 1
                    assert vgpr <= 512
    assert seen["reg1"] > 0 and seen["big2"] > 0 and seen["big1"] > 0 and seen["reg2"] > 0, seen


@pytest.mark.skipif(not (os.path.exists(f"{LLVM}/clang-offload-bundler") and os.path.exists(f"{LLVM}/llvm-objdump")), reason="ROCm binutils not installed")
def test_block_structured_kernels_do_not_spill_inside_the_swept_inverse():
    """Compiler hazard 3: a VGPR spill placed inside a partial-EXEC region loses the inactive lanes' copies. The blocked sweep of RegKkt
    stages its tiles under lane predicates; the block-structured kernels (pmpc_qp_schur.hpp) call it with far more live state around it than the dense
    one-row-per-lane kernels do, and a build that put scratch traffic there returned wrong answers on the 16-node grid. Checked on the built code: no
    scratch instruction between the first and the last matrix-core instruction of any sqp_schur_kernel."""
    checked = 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in _code_objects(tmp):
            syms = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            if "sqp_schur_kernel" not in syms:
                continue
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            for blk in re.split(r"\n(?=[0-9a-f]+ <)", dis):
                head = blk.split("\n", 1)[0]
                if "sqp_schur_kernel" not in head:
                    continue
                lines = blk.split("\n")
                mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
                assert mf, head
                # no spill STORE anywhere inside, and no scratch access at all while EXEC is narrowed (between an s_and_saveexec and the s_or that restores it). A
                # reload issued with EXEC restored is harmless (round 5: the bordered parking build reloads a spilled zero constant there, the phase-timer build of the 7-node robot grid one operand since the KKT diagonal moved into registers).
                narrowed, bad = False, []
                for l in lines[mf[0]:mf[-1]]:
                    if "s_and_saveexec" in l or re.search(r"s_(and|andn2|mov)_b64 exec,", l): narrowed = True
                    if re.search(r"s_or_b64 exec, exec,", l) or re.search(r"s_mov_b64 exec, -1", l): narrowed = False
                    if "scratch_store" in l or ("scratch_" in l and narrowed): bad.append(l.strip())
                assert not bad, f"{head}: {len(bad)} scratch instructions under a narrowed EXEC / spill stores inside the swept inverse: {bad[:3]}"
                checked += 1
    assert checked >= 4


@pytest.mark.skipif(not (os.path.exists(f"{LLVM}/clang-offload-bundler") and os.path.exists(f"{LLVM}/llvm-objdump")), reason="ROCm binutils not installed")
def test_headline_kernel_has_no_scratch_and_condensed_kernels_none_in_the_admm_loop():
    """Two properties of the built code that cost a measurable share of a benchmark line when they break (compiler hazards 4 and 11): the
    headline kernel sqp_kernel<RobotOCP, 35, 21> has NO scratch traffic at all (round 4: the register-row BFGS spilled the head of the row of B behind its
    loads — 1.18 instead of 1.13 ms per 4096), and the condensed register kernels (CND = true: config B, the 16-node robot grid) have none between the first
    and the last DPP mat-vec instruction, i.e. inside the ADMM loop."""
    headline = cond = 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in _code_objects(tmp):
            syms = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            if "sqp_kernelINS_8RobotOCPELi35ELi21ELb0ELi0ELb0ELb0ELb0ELb0E" not in syms and not re.search(r"sqp_kernelINS_\d+\w+?ELi\d+ELi\d+ELb0ELi0ELb0ELb0ELb0ELb1E", syms):
                continue   # (disassembling every code object of the library takes half a minute)
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            for blk in re.split(r"\n(?=[0-9a-f]+ <)", dis):
                head = blk.split("\n", 1)[0]
                if "sqp_kernelINS_8RobotOCPELi35ELi21ELb0ELi0ELb0ELb0ELb0ELb0E" in head:
                    n = sum("scratch_" in l for l in blk.split("\n"))
                    assert n == 0, f"{head}: {n} scratch instructions in the headline kernel"
                    headline += 1
                elif re.search(r"sqp_kernelINS_\d+\w+?ELi\d+ELi\d+ELb0ELi0ELb0ELb0ELb0ELb1E", head):
                    lines = blk.split("\n")
                    dpp = [i for i, l in enumerate(lines) if "row_newbcast" in l]
                    assert dpp, head
                    inside = [l for l in lines[dpp[0]:dpp[-1]] if "scratch_" in l]
                    assert not inside, f"{head}: {len(inside)} scratch instructions inside the ADMM loop"
                    cond += 1
    assert headline == 1 and cond >= 2, (headline, cond)
