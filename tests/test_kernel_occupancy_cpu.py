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


def _exec_windows_of(co):
    """(kernels with an EXEC-window helper, helper bodies, [unproven bodies], [provable AGPR lane-validity violations]) of one code object — module-level so
    that a process pool can run it"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tools_exec_regions as ter
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--symbolize-operands", co], capture_output=True, text=True).stdout
    kernels = bodies = 0
    bad, bad_agpr = [], []
    for name, found, unproven, agpr in ter.full_report(dis):
        if found:
            kernels += 1; bodies += len(found)
        assert all(f["restore"] in ("exec, -1", "saved") for f in found), name
        bad += [(name[:120], f["block"], f["text"]) for f in unproven]
        bad_agpr += [(name[:120],) + tuple(v) for v in agpr]
    return kernels, bodies, bad, bad_agpr


@pytest.mark.skipif(not (os.path.exists(f"{LLVM}/clang-offload-bundler") and os.path.exists(f"{LLVM}/llvm-objdump")), reason="ROCm binutils not installed")
def test_exec_window_helpers_start_with_every_lane_enabled_and_no_spill_loses_lanes():
    """Two properties of the BUILT code of every shipped kernel, decided by a forward data-flow over each kernel's control-flow graph (tests/tools_exec_regions.py:
    EXEC and every saved copy of it as symbolic masks, through s_*_saveexec, if / else / loop lowering and SGPR spill lanes; conservative: unknown = fail):
    (1) The one inline-asm statement of the product that writes EXEC (pmpc_qp_reg.hpp `pivot_lane_setup`: a one-lane window, restored to the constant -1) is
        sound only where it is emitted with EXEC all-ones — inside the then-block of a lane-divergent if / else the structurizer's `s_or_saveexec ; s_xor`
        would compute an empty else mask and the else lanes would silently lose their work. VERDICT round 5 suspected exactly that behind round 5's
        miscompiled hook build; EXPERIMENTS.md round 6: refuted (every body proven at full EXEC in that build too, and the fault is the same with no EXEC
        write at all). Here: every helper body of the library starts in a state PROVEN to be full EXEC.
    (2) What that fault really was — compiler hazard 3 of DESIGN.md, now with the instruction sequence: a VGPR -> AGPR spill (`v_accvgpr_write_b32 a116, v40`,
        the lane id) emitted inside the else-block of a lane-divergent if / else and read back at full EXEC; the lanes of the then-side come back as stale
        register content. Here: no accumulation register of any shipped kernel is written under a provably narrowed EXEC and read back with provably more
        lanes enabled (the faulty build: three such reads in exactly the faulty kernel, tests/experiments/exec_probe_run.sh). The same rule over the private
        (scratch) slots with a constant offset — the other place a spill can go: about 10 000 spill stores and 17 000 reloads in the library, none of them loses lanes."""
    from concurrent.futures import ProcessPoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        objs = _code_objects(tmp)
        assert objs
        with ProcessPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
            res = list(ex.map(_exec_windows_of, objs))
    kernels = sum(r[0] for r in res); bodies = sum(r[1] for r in res)
    bad = [b for r in res for b in r[2]]
    bad_agpr = [b for r in res for b in r[3]]
    assert kernels >= 100 and bodies >= 5000, (kernels, bodies)     # every register-resident QP kernel carries them (35 + 21: 32 bodies per inverse site)
    assert not bad and not bad_agpr, (f"{len(bad)} EXEC-window helper bodies where EXEC is not proven full: {bad[:3]}; {len(bad_agpr)} accumulation-register / scratch-slot reads with "
                                      f"lanes enabled that the last write provably did not cover (a spill inside a partial-EXEC block): {bad_agpr[:3]}")
    assert not bad_agpr, f"{len(bad_agpr)} accumulation-register / scratch-slot reads with lanes enabled that the last write provably did not cover (a spill inside a partial-EXEC block): {bad_agpr[:3]}"


def test_exec_region_analysis_sees_the_two_fault_patterns():
    """The analysis itself on three hand-written listings: the faulty build's sequence (condensed: a spill inside the else-block of a divergent if / else, read back
    at full EXEC) is reported, an EXEC-window helper inside a then-block is reported as not proven, and the same code outside the blocks is clean."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tools_exec_regions as ter
    head = "\t.text\nkern:\n\tv_cmp_lt_i32_e32 vcc, 32, v0\n\ts_and_saveexec_b64 s[6:7], vcc\n\ts_xor_b64 s[6:7], exec, s[6:7]\n"
    helper = "\ts_lshl_b64 exec, 1, 4\n\tv_mov_b64 v[2:3], 0\n\ts_mov_b64 exec, -1\n"
    spill = "\tv_accvgpr_write_b32 a116, v40\n"
    mid = "\tv_add_u32_e32 v2, 1, v0\n\ts_or_saveexec_b64 s[6:7], s[6:7]\n\ts_xor_b64 exec, exec, s[6:7]\n"
    tail = "\ts_or_b64 exec, exec, s[6:7]\n\tv_accvgpr_read_b32 v161, a116\n\ts_endpgm\n.Lfunc_end0:\n"
    # (a) the spill in the else-block, read back behind the join
    (name, found, unproven, agpr), = ter.full_report(head + mid + spill + tail)
    assert len(agpr) == 1 and "a116" in agpr[0][1] and not found
    # (b) the helper inside the then-block
    (name, found, unproven, agpr), = ter.full_report(head + helper + mid + tail.replace("\tv_accvgpr_read_b32 v161, a116\n", ""))
    assert len(found) == 1 and len(unproven) == 1
    # (c) both behind the join, at full EXEC: clean
    (name, found, unproven, agpr), = ter.full_report(head + mid + "\ts_or_b64 exec, exec, s[6:7]\n" + helper + spill + "\tv_accvgpr_read_b32 v161, a116\n\ts_endpgm\n.Lfunc_end0:\n")
    assert len(found) == 1 and not unproven and not agpr
    # (d) a mask that travels through an SGPR spill lane and a loop (LLVM's SI_LOOP lowering) still restores full EXEC
    loop = ("\t.text\nkern:\n\tv_cmp_lt_i32_e32 vcc, 32, v0\n\ts_mov_b64 s[2:3], exec\n\tv_writelane_b32 v250, s2, 5\n\tv_writelane_b32 v250, s3, 6\n\ts_and_b64 s[2:3], s[2:3], vcc\n"
            "\ts_mov_b64 exec, s[2:3]\n\ts_cbranch_execz .LBB0_3\n\ts_mov_b64 s[8:9], 0\n.LBB0_2:\n\tv_cmp_lt_i32_e32 vcc, 3, v1\n\ts_or_b64 s[8:9], vcc, s[8:9]\n\ts_andn2_b64 exec, exec, s[8:9]\n"
            "\ts_cbranch_execnz .LBB0_2\n\ts_or_b64 exec, exec, s[8:9]\n.LBB0_3:\n\tv_readlane_b32 s4, v250, 5\n\tv_readlane_b32 s5, v250, 6\n\ts_or_b64 exec, exec, s[4:5]\n" + helper + "\ts_endpgm\n.Lfunc_end0:\n")
    (name, found, unproven, agpr), = ter.full_report(loop)
    assert len(found) == 1 and not unproven
    # (e) the same lane loss through a private (scratch) slot: a spill store inside the else-block, the reload behind the join — reported; the store behind the join
    #     (full EXEC), a reload inside the same block, and a slot addressed through a register (a private array, not followed) — clean
    sst, sld = "\tscratch_store_dwordx2 off, v[40:41], off offset:72\n", "\tscratch_load_dwordx2 v[160:161], off, off offset:72\n"
    fin = "\ts_endpgm\n.Lfunc_end0:\n"
    res = ter.full_report(head + mid + sst + "\ts_or_b64 exec, exec, s[6:7]\n" + sld + fin)
    assert len(res) == 1 and len(res[0][3]) == 2 and "offset:72" in res[0][3][0][1]   # (one report per dword slot)
    assert not ter.full_report(head + mid + "\ts_or_b64 exec, exec, s[6:7]\n" + sst + sld + fin)
    assert not ter.full_report(head + mid + sst + sld + "\ts_or_b64 exec, exec, s[6:7]\n" + fin)
    assert not ter.full_report(head + mid + "\tscratch_store_dwordx2 v5, v[40:41], off offset:72\n\ts_or_b64 exec, exec, s[6:7]\n\tscratch_load_dwordx2 v[160:161], v5, off offset:72\n" + fin)
