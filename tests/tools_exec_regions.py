"""Developer tool + library for tests/test_kernel_occupancy_cpu.py: is an EXEC-writing inline-asm helper ever emitted where EXEC may be narrowed?

The register QP kernels contain ONE kind of inline assembly that writes EXEC behind the compiler's back (pmpc_qp_reg.hpp `pivot_lane_setup`: a one-lane EXEC
window around a handful of moves). Its original form ended with `s_mov_b64 exec, -1`, which is only sound where EXEC is all-ones on entry: inside the
then-block of a lane-divergent if / else the structurizer's `s_or_saveexec ; s_xor exec` sequence would then compute an EMPTY else mask, i.e. the else
lanes silently lose their work. This module decides that question on the built code.

Method: forward data-flow over the basic blocks of a kernel (GCN assembly text — `hipcc -S` output or `llvm-objdump -d --symbolize-operands`), abstract
state = (EXEC known to be all-ones?, set of SGPR pairs known to hold an all-ones mask). Every instruction that writes EXEC or an SGPR pair updates it;
joins are intersections. CONSERVATIVE: anything not understood (masks travelling through SGPR spills, v_cmpx, ...) makes EXEC "possibly narrowed".
A helper body reported as `full` is therefore proven to start with every lane enabled; `maybe-narrowed` needs a look.

    python tests/tools_exec_regions.py file.s [kernel-name-substring]
"""
import re
import sys

_LABEL = re.compile(r"^(\.?[A-Za-z_$][\w$.]*|<L\d+>|<[\w$.+]+>):")
_SREG = re.compile(r"^s\[(\d+):(\d+)\]$|^s(\d+)$|^(vcc|exec)$")


def split_functions(text):
    """yield (name, [lines]) for every function body of a `hipcc -S` file or an `llvm-objdump -d` listing"""
    if ".amdgcn_target" in text or "\t.text" in text:      # compiler output
        cur, name = None, None
        for ln in text.split("\n"):
            m = re.match(r"^([A-Za-z_$][\w$.]*):", ln)
            if m and not m.group(1).startswith(".L"):
                if cur is not None:
                    yield name, cur
                name, cur = m.group(1), []
                continue
            if cur is not None:
                if ln.startswith(".Lfunc_end"):
                    yield name, cur
                    cur, name = None, None
                else:
                    cur.append(ln)
        if cur is not None:
            yield name, cur
    else:                                                    # objdump
        for blk in re.split(r"\n(?=[0-9a-f]+ <(?!L\d+>)[^>]+>:)", text):   # (<L12>: the basic-block labels of --symbolize-operands stay inside their function)
            m = re.match(r"[0-9a-f]+ <([^>]+)>:", blk)
            if m:
                yield m.group(1), blk.split("\n")[1:]


def _regs(op):
    """SGPR numbers named by an operand (vcc = 106/107 by convention here; exec handled separately)"""
    op = op.strip().rstrip(",")
    m = _SREG.match(op)
    if not m:
        return None
    if m.group(1):
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    if m.group(3):
        return {int(m.group(3))}
    if m.group(4) == "vcc":
        return {1000, 1001}
    return None


def _parse(lines):
    """-> blocks: list of dict(label, insts[(mnemonic, operands, raw)], succ labels, falls_through)"""
    blocks, cur = [], {"label": "<entry>", "insts": [], "succ": [], "fall": True}
    for raw in lines:
        s = raw.split(";")[0].split("//")[0].rstrip()
        if not s.strip():
            continue
        st = s.strip()
        st = re.sub(r"^[0-9a-f]+:?\s+(?=<)", "", st)        # objdump: "0000000000005170 <L0>:" -> "<L0>:"
        m = _LABEL.match(st)
        if m and not st.startswith("\t"):
            if cur["insts"] or cur["label"] == "<entry>":
                blocks.append(cur)
            cur = {"label": m.group(1), "insts": [], "succ": [], "fall": True}
            continue
        if st.startswith("."):
            continue
        parts = st.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur["insts"].append((mn, ops, st))
        if mn.startswith("s_cbranch") or mn == "s_branch":
            tgt = ops[-1].split()[-1] if ops else None
            if tgt:
                cur["succ"].append(tgt.strip())
            if mn == "s_branch":
                cur["fall"] = False
            blocks.append(cur)
            cur = {"label": None, "insts": [], "succ": [], "fall": True}
        elif mn in ("s_endpgm", "s_setpc_b64"):
            cur["fall"] = False
            blocks.append(cur)
            cur = {"label": None, "insts": [], "succ": [], "fall": True}
    if cur["insts"]:
        blocks.append(cur)
    return blocks


# ---- abstract masks --------------------------------------------------------------------------------------------------------------------------------
# FULL                      all 64 lanes
# ("sub", B, id, pol)       EXACTLY  B & cond_id  (pol = 1)  or  B & ~cond_id  (pol = 0), B another abstract mask, id = the instruction that narrowed
# ("in", B)                 some subset of B (inexact: the result of a join of different subsets of B, e.g. EXEC at the header of a lane-divergent loop)
# None                      unknown
FULL = ("full",)
_MAXDEPTH = 12
TRACK_AGPR = False   # set by agpr_report(): also follow under which EXEC every accumulation register was last written


def _depth(v):
    d = 0
    while v is not None and v != FULL:
        v = v[1]; d += 1
    return d


def _exact(v):
    while v is not None and v != FULL:
        if v[0] != "sub":
            return False
        v = v[1]
    return v == FULL


def _contains(big, small):
    """is `small` provably a subset of `big`?"""
    v = small
    while v is not None:
        if v == big:
            return True
        if v == FULL:
            return False
        v = v[1]
    return False


def _union(a, b):
    if a == FULL or b == FULL:
        return FULL
    if a is None or b is None:
        return None
    if _contains(a, b):
        return a
    if _contains(b, a):
        return b
    if a[0] == "sub" and b[0] == "sub" and a[1] == b[1] and a[2] == b[2] and a[3] != b[3]:
        return a[1]                              # the two halves of one split
    return None


def _join(a, b):
    if a == b:
        return a
    if a is None or b is None:
        return None
    # closest common ancestor -> "some subset of it"
    anc = []
    v = a
    while v is not None:
        anc.append(v)
        if v == FULL:
            break
        v = v[1]
    v = b
    while v is not None:
        if v in anc:
            if v != FULL and v[0] == "in":
                return v                         # a subset of "some subset of B" is some subset of B
            return ("in", v) if _depth(v) < _MAXDEPTH else None
        if v == FULL:
            break
        v = v[1]
    return None


def _narrow(v, ident, pol=1):
    if v is None or _depth(v) >= _MAXDEPTH:
        return None
    return ("sub", v, ident, pol)


def _step(state, mn, ops, ident):
    """abstract transfer of one instruction; state = (exec value, {key: (value, half)}) with key = an SGPR number (a 64-bit mask is tracked as its two
    32-bit halves, so that it can be followed through the SGPR spill lanes of a VGPR: v_writelane / v_readlane) or ("slot", vgpr, lane)"""
    ex, regs = state
    regs = dict(regs)

    def val(op):
        if op == "-1":
            return FULL
        if op == "exec":
            return ex
        r = _regs(op)
        if r is not None and len(r) == 2:
            lo = min(r)
            a, b = regs.get(lo), regs.get(lo + 1)
            if a is not None and b is not None and a[0] == b[0] and a[1] == 0 and b[1] == 1:
                return a[0]
        return None

    def kill(op):
        r = _regs(op)
        if r:
            for x in r:
                regs.pop(x, None)

    def setp(op, v):
        kill(op)
        r = _regs(op)
        if r and len(r) == 2 and v is not None:
            regs[min(r)] = (v, 0); regs[min(r) + 1] = (v, 1)

    if not ops:
        return (ex, regs)
    d = ops[0]
    if mn == "s_and_saveexec_b64":
        old = ex; setp(d, old); ex = _narrow(old, ident, 1)
    elif mn == "s_andn2_saveexec_b64":            # d <- exec ; exec <- src & ~exec
        old = ex; src = val(ops[1])
        if src is not None and old is not None and src != FULL and old != FULL and src[0] == "sub" and old[0] == "sub" and src[1] == old[1] and src[2] == old[2] and src[3] != old[3]:
            new = src                              # (else lanes) & ~(then lanes) = else lanes
        elif src is not None:
            new = ("in", src) if _depth(src) < _MAXDEPTH else None
        else:
            new = None
        setp(d, old); ex = new
    elif mn == "s_or_saveexec_b64":               # d <- exec ; exec <- src | exec
        old = ex; src = val(ops[1]); setp(d, old); ex = _union(src, old)
    elif mn.endswith("_saveexec_b64"):
        old = ex; setp(d, old); ex = None
    elif d == "exec":
        if mn == "s_mov_b64":
            ex = val(ops[1])
        elif mn == "s_or_b64":
            ex = _union(val(ops[1]), val(ops[2]))
        elif mn == "s_xor_b64":                    # exec ^ (a half of it) = the other half
            a, b = val(ops[1]), val(ops[2])
            if a is not None and b is not None and b != FULL and b[0] == "sub" and b[1] == a:
                ex = ("sub", a, b[2], 1 - b[3])
            elif a is not None and b is not None and a != FULL and a[0] == "sub" and a[1] == b:
                ex = ("sub", b, a[2], 1 - a[3])
            else:
                ex = None
        elif mn in ("s_and_b64", "s_andn2_b64"):
            a = val(ops[1]) if ops[1] == "exec" else (val(ops[2]) if ops[2] == "exec" else None)
            ex = _narrow(a, ident, 1)
        else:
            ex = None                               # (incl. the helper's own s_lshl / s_bfm window: it restores EXEC itself)
    elif d in ("exec_lo", "exec_hi") or mn.startswith("v_cmpx"):
        ex = None
    elif mn == "s_mov_b64" and len(ops) > 1:
        setp(d, val(ops[1]))
    elif mn == "s_mov_b32" and len(ops) > 1:       # a mask copied one half at a time
        r, q = _regs(d), _regs(ops[1])
        h = regs.get(min(q)) if q and len(q) == 1 else None
        kill(d)
        if r and len(r) == 1 and h is not None:
            regs[min(r)] = h
    elif mn == "s_xor_b64" and len(ops) > 2:       # d <- exec ^ d (the else mask of an if / else)
        a, b = val(ops[1]), val(ops[2])
        if a is not None and b is not None and a != FULL and a[0] == "sub" and a[1] == b:
            setp(d, ("sub", b, a[2], 1 - a[3]))
        elif a is not None and b is not None and b != FULL and b[0] == "sub" and b[1] == a:
            setp(d, ("sub", a, b[2], 1 - b[3]))
        else:
            kill(d)
    elif mn == "s_and_b64" and len(ops) > 2:       # d <- a & b: a subset of whichever operand is a known mask
        a, b = val(ops[1]), val(ops[2])
        setp(d, _narrow(a if a is not None else b, ident, 1))
    elif mn == "s_andn2_b64" and len(ops) > 2:     # d <- a & ~b
        setp(d, _narrow(val(ops[1]), ident, 1))
    elif mn == "s_or_b64" and len(ops) > 2:
        setp(d, _union(val(ops[1]), val(ops[2])))
    elif mn == "v_writelane_b32" and len(ops) > 2:  # SGPR spill into a lane of a VGPR
        q = _regs(ops[1])
        key = ("slot", d, ops[2])
        h = regs.get(min(q)) if q and len(q) == 1 else None
        if h is not None:
            regs[key] = h
        else:
            regs.pop(key, None)
    elif mn == "v_readlane_b32" and len(ops) > 2:
        h = regs.get(("slot", ops[1], ops[2]))
        kill(d)
        r = _regs(d)
        if r and len(r) == 1 and h is not None:
            regs[min(r)] = h
    elif mn.startswith("s_") and not mn.startswith(("s_cmp", "s_bitcmp", "s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio", "s_cbranch", "s_branch", "s_store", "s_sendmsg", "s_setreg", "s_dcache", "s_icache", "s_endpgm", "s_trap", "s_set_gpr", "s_setpc", "s_sethalt", "s_inst_prefetch", "s_clause", "s_ttrace", "s_code_end", "s_version", "s_delay_alu", "s_wait")):
        kill(d)                                  # SALU / SMEM write of some other SGPR (range)
    elif mn == "v_readfirstlane_b32" or mn.startswith("v_cmp") or mn in ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_div_scale_f64", "v_div_scale_f32", "v_mad_u64_u32", "v_mad_i64_i32"):
        kill(d)
        if len(ops) > 1:
            kill(ops[1])                         # carry-out / second scalar destination
    # ---- lane validity of the accumulation registers (compiler hazard 3, DESIGN.md: a VGPR -> AGPR spill emitted inside a partial-EXEC block keeps only the
    # active lanes' copies): every write of aN records the EXEC it ran under; _agpr_reads() checks each read against it
    if TRACK_AGPR and d and d[0] == "a" and (d[1:].isdigit() or d.startswith("a[")):
        m2 = re.match(r"a\[(\d+):(\d+)\]$|a(\d+)$", d)
        if m2:
            lo, hi = (int(m2.group(1)), int(m2.group(2))) if m2.group(1) else (int(m2.group(3)), int(m2.group(3)))
            for r_ in range(lo, hi + 1):
                prev = regs.get(("a", r_))
                # a write under EXEC e on top of data valid under w leaves valid: w | e — kept when one contains the other, else just e
                if prev is not None and ex is not None and _contains(prev[0], ex):
                    continue
                if ex is None:
                    regs.pop(("a", r_), None)
                else:
                    regs[("a", r_)] = (ex, 2)
    # ---- the same for private (scratch) slots addressed by a constant offset — `scratch_store_dwordxN off, v[..], off offset:K`: a VGPR spilled to scratch inside a
    # partial-EXEC block keeps only the active lanes' copies, exactly like the AGPR case. Slots addressed through a register (private arrays) are not followed.
    if TRACK_AGPR and mn.startswith("scratch_store_dword"):
        sl = _scratch_slots(mn, ops, 0)
        if sl:
            for r_ in sl:
                prev = regs.get(("m", r_))
                if prev is not None and ex is not None and _contains(prev[0], ex):
                    continue
                if ex is None:
                    regs.pop(("m", r_), None)
                else:
                    regs[("m", r_)] = (ex, 2)
    if mn == "s_swappc_b64":                     # a call: the callee may clobber the caller-saved SGPRs; EXEC is preserved by the ABI
        regs = {x: v for x, v in regs.items() if not isinstance(x, int) or x >= 30 or x == -1}
    return (ex, regs)


def _scratch_slots(mn, ops, first):
    """dword slots of a scratch access with a constant address (`off ... off offset:K`), else None. ops: [off, data, off offset:K] (store) / [data, off, off offset:K] (load)"""
    addr = [o.strip() for o in ops[first:first + 1] + ops[first + 2:]] if first == 0 else [o.strip() for o in ops[1:]]
    if len(addr) < 2 or addr[0] != "off" or not addr[1].startswith("off"):
        return None
    m2 = re.search(r"offset:(\d+)", " ".join(addr))
    k = int(m2.group(1)) if m2 else 0
    w = {"": 1, "x2": 2, "x3": 3, "x4": 4}.get(mn.split("dword")[-1], None)
    if w is None or k % 4:
        return None
    return list(range(k // 4, k // 4 + w))


def _run_block(b, st, i, visit=None):
    """abstract execution of one basic block. LLVM's loop lowering (SI_IF_BREAK / SI_LOOP / SI_END_CF) is recognised as a unit:
           s_or_b64 sA, cond, sA ; s_andn2_b64 exec, exec, sA ; s_cbranch_execnz header ; s_or_b64 exec, exec, sA
    — sA accumulates the lanes that left the loop, so that behind the final s_or EXEC is what it was on entry to the loop (the base of the header's
    "some subset of E0" value). The final s_or may open the next block; the pending restore travels in the state as register key -1."""
    insts = b["insts"]
    for k, (mn, ops, raw) in enumerate(insts):
        if visit:
            visit(k, mn, ops, raw, st)
        if mn == "s_andn2_b64" and ops and ops[0] == "exec" and ops[1] == "exec" and k + 1 < len(insts) and insts[k + 1][0] in ("s_cbranch_execnz", "s_cbranch_execz"):
            v = st[0]
            e0 = v[1] if (v is not None and v != FULL and v[0] == "in") else v
            st = _step(st, mn, ops, (i, k))
            regs = dict(st[1]); regs[-1] = (ops[2], e0)
            st = (st[0], regs)
            continue
        if mn == "s_or_b64" and ops and ops[0] == "exec" and ops[1] == "exec" and -1 in st[1] and st[1][-1][0] == ops[2]:
            e0 = st[1][-1][1]
            regs = {r: v for r, v in st[1].items() if r != -1}
            st = (e0, regs)
            continue
        if mn not in ("s_cbranch_execnz", "s_cbranch_execz") and -1 in st[1]:
            # the pending restore survives instructions that touch neither EXEC nor the accumulated mask (the scheduler moves unrelated scalar moves in between)
            pend = _regs(st[1][-1][0]) or set()
            dst = ops[0].strip() if ops else ""
            wr = _regs(dst) or set()
            if dst.startswith("exec") or (wr & pend) or mn.endswith("_saveexec_b64") or mn.startswith(("v_cmpx", "s_swappc", "s_setpc", "s_branch", "s_cbranch")):
                st = (st[0], {r: v for r, v in st[1].items() if r != -1})
        st = _step(st, mn, ops, (i, k))
    return st


def analyse(lines):
    """-> ([{block, state 'full' | 'maybe-narrowed', restore 'exec, -1' | 'saved', saves_exec, text}] for every EXEC-window helper body, blocks, entry states)"""
    blocks = _parse(lines)
    idx = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    n = len(blocks)
    if n == 0:
        return [], blocks, []
    succ = [[] for _ in range(n)]
    for i, b in enumerate(blocks):
        for t in b["succ"]:
            t2 = t if t in idx else ("<" + t.strip("<>") + ">" if ("<" + t.strip("<>") + ">") in idx else None)
            if t2 is not None:
                succ[i].append(idx[t2])
        if b["fall"] and i + 1 < n:
            succ[i].append(i + 1)
    entry = [None] * n
    entry[0] = (FULL, {})
    work = [0]
    rounds = 0
    while work:
        rounds += 1
        if rounds > 200 * n:
            break
        i = work.pop()
        st = _run_block(blocks[i], entry[i], i)
        for j in succ[i]:
            if entry[j] is None:
                new = (st[0], dict(st[1]))
            else:
                ex = _join(entry[j][0], st[0])
                regs = {}
                for r, v in entry[j][1].items():
                    if r in st[1]:
                        w = st[1][r]
                        if r == -1:
                            if v == w:
                                regs[r] = v
                        elif v[1] == w[1]:
                            jv = _join(v[0], w[0])
                            if jv is not None:
                                regs[r] = (jv, v[1])
                new = (ex, regs)
            if entry[j] is None or new != entry[j]:
                entry[j] = new
                work.append(j)
    found = []
    for i, b in enumerate(blocks):
        if entry[i] is None:
            continue
        insts = b["insts"]

        def visit(k, mn, ops, raw, st, b=b, i=i, insts=insts):
            # the helper bodies: a scalar shift / bit-field mask INTO exec (nothing the compiler itself emits)
            if ops and ops[0] == "exec" and mn in ("s_lshl_b64", "s_lshr_b64", "s_bfm_b64"):
                tail = " | ".join(r for _, _, r in insts[k:k + 14])
                restore = "exec, -1" if re.search(r"s_mov_b64 exec, -1", tail) else ("saved" if re.search(r"s_mov_b64 exec, s\[", tail) else "?")
                found.append({"block": b["label"] or f"#{i}", "state": "full" if st[0] == FULL else "maybe-narrowed", "restore": restore, "text": raw})
        _run_block(b, entry[i], i, visit)
    return found, blocks, entry


def full_report(text, want=None):
    """both checks in one pass over the listing: -> [(kernel, helper bodies found, helper bodies not proven at full EXEC, provable AGPR lane-validity violations)]"""
    global TRACK_AGPR
    TRACK_AGPR = True
    try:
        out = []
        for name, lines in split_functions(text):
            if want and want not in name:
                continue
            found, blocks, entry = analyse(lines)
            bad_agpr = _agpr_violations(blocks, entry) if any(mn.startswith(("v_accvgpr", "scratch_load")) for b in blocks for mn, _, _ in b["insts"]) else []
            if found or bad_agpr:
                out.append((name, found, [f for f in found if f["state"] != "full"], bad_agpr))
        return out
    finally:
        TRACK_AGPR = False


def _agpr_violations(blocks, entry):
    bad = []
    for i, b in enumerate(blocks):
        if entry[i] is None:
            continue

        def visit(k, mn, ops, raw, st, b=b, i=i):
            if mn.startswith("scratch_load_dword"):
                for r_ in (_scratch_slots(mn, ops, 1) or []):
                    w = st[1].get(("m", r_))
                    if w is None or w[0] == FULL or st[0] is None or not _exact(w[0]) or not _exact(st[0]):
                        continue
                    if not _contains(w[0], st[0]):
                        bad.append((b["label"] or f"#{i}", raw, str(w[0])[:80], str(st[0])[:80]))
            srcs = ops[1:] if not mn.startswith(("ds_write", "global_store", "scratch_store", "flat_store", "buffer_store")) else ops
            for o in srcs:
                m2 = re.match(r"a\[(\d+):(\d+)\]$|a(\d+)$", o.strip())
                if not m2:
                    continue
                lo, hi = (int(m2.group(1)), int(m2.group(2))) if m2.group(1) else (int(m2.group(3)), int(m2.group(3)))
                for r_ in range(lo, hi + 1):
                    w = st[1].get(("a", r_))
                    if w is None or w[0] == FULL:
                        continue                      # not written on this path as far as the analysis knows / written with every lane enabled
                    if st[0] is None or not _exact(w[0]) or not _exact(st[0]):
                        continue                      # only PROVABLE violations: both masks exact (no join-widened "some subset of" value on either side)
                    if not _contains(w[0], st[0]):
                        bad.append((b["label"] or f"#{i}", raw, str(w[0])[:80], str(st[0])[:80]))
        _run_block(b, entry[i], i, visit)
    return bad


def agpr_report(text, want=None):
    """Per kernel: the reads of an accumulation register (v_accvgpr_read, or aN as a source operand) that PROVABLY run with lanes enabled for which the
    register's last write was not enabled — the signature of a VGPR -> AGPR spill placed inside a partial-EXEC block (round 5's miscompiled hook build:
    `v_accvgpr_write_b32 a116, v40` in the else-block of a lane-divergent if, read back under full EXEC). -> [(kernel, [(block, instruction, written-under, read-under)])]"""
    return [(name, bad) for name, found, unproven, bad in full_report(text, want)]


def report(text, want=None):
    out = []
    for name, lines in split_functions(text):
        if want and want not in name:
            continue
        found, blocks, entry = analyse(lines)
        if not found:
            continue
        bad = [f for f in found if f["state"] != "full"]
        out.append((name, len(found), bad, found))
    return out


if __name__ == "__main__":
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else None
    if want == "--agpr":
        want = sys.argv[3] if len(sys.argv) > 3 else None
        for name, bad in agpr_report(txt, want):
            print(f"{name[:150]}: {len(bad)} AGPR reads with lanes enabled that the last write did not cover")
            for f in bad[:6]:
                print("    ", f)
        sys.exit(0)
    for name, nf, bad, found in report(txt, want):
        kinds = sorted({f["restore"] for f in found})
        print(f"{name[:150]}: {nf} EXEC-window helper bodies, {len(bad)} where EXEC may be narrowed; forms {kinds}")
        for f in bad[:10]:
            print("    ", f["block"], f["text"])
