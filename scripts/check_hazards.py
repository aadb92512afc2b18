#!/usr/bin/env python3
"""Static audit of the software-managed instruction hazards in the gfx950 device code.

hipcc pads the wait states ("s_nop") that gfx950 needs between certain producer/consumer pairs, but it does NOT
look inside an `asm` statement (its hazard recognizer treats the statement as one opaque instruction).  The kernels
use inline asm for DPP FMAs, v_permlane swaps, literal-vcc selects and fused row sums; a missing wait state there
gives silently wrong numbers on a few instances per ten thousand (round 1, commits 9980a4f / 31ce26a).  This script
compiles csrc/ftmpc_capi.hip to device assembly (no GPU needed) and re-checks EVERY instruction pair of the final
stream -- compiler-scheduled code and asm bodies alike -- against the rule table below, walking backwards through
the control-flow graph (fall-through and branch edges) for up to the longest rule's wait states.

Rule table (gfx940/gfx950; names of the LLVM GCNHazardRecognizer checks they restate):
  dpp_vgpr        VALU writes VGPR -> DPP instruction reads it                      2   checkDPPHazards (DppVgprWaitStates)
  dpp_exec        VALU writes EXEC -> DPP instruction                               5   checkDPPHazards (DppExecWaitStates)
  trans_use       transcendental result -> non-transcendental VALU reads it         1   checkVALUHazards (TransDefWaitstates)
  permlane_swap   VALU writes VGPR -> v_permlane16/32_swap reads/writes it          2   checkPermlaneHazards (gfx950)
  readlane_vgpr   VALU writes VGPR -> v_readlane / v_readfirstlane reads it         1   VALUWriteVGPRReadlaneRead
  sgpr_valu       VALU writes SGPR/VCC -> VALU reads it (incl. implicit vcc)        2   VALUWriteSGPRVALURead
  lane_select     VALU writes SGPR -> v_readlane/v_writelane lane select            4   checkRWLaneHazards
  div_fmas        VALU writes VCC -> v_div_fmas                                     4   checkDivFMasHazards
  sgpr_vmem       VALU writes SGPR -> VMEM reads it                                 5   VALUWriteSGPRVMEMRead
  mfma_use        MFMA writes D -> VALU read/write, LDS / VMEM read of D            f32 16x16x4: 10, 32x32x2: 18, 4x4x1: 4
                                                                                    f64 16x16x4: 11 (memory read 18), 4x4x4: 6 (9)
                                                                                    checkMAIVALUHazards (SMFMA N-pass: N+2; DMFMA)
  mfma_srcab      MFMA writes D -> another MFMA reads it as its A or B operand      as mfma_use (checkMAIHazards90A, SrcA/B overlap)
  valu_mfma       VALU writes VGPR -> MFMA reads it as A/B/C                        2   (reported for asm producers only)
A wait state is one issued instruction; `s_nop N` is N+1.

Pairs the compiler scheduled itself are reported separately ("compiler-only").  For most of the project's life that
list was empty and served as the calibration of the rule table -- until the workgroup kernel's helper loops produced
    v_mfma_f32_16x16x4_f32 a[0:3], ...   (end of a chain)
    v_mfma_f32_16x16x4_f32 a[4:7], ...
    s_cbranch_execnz .LBB15_3508
  .LBB15_3508:
    s_waitcnt vmcnt(1)
    v_accvgpr_read_b32 v7, a3            (3 wait states after the producer; the table says 10)
and the kernel returned wrong factors on the GPU (3861 of 4096 instances not converged) until the consumer was padded.
So `--elide` also PADS every compiler-only pair up to the table, and the audit of its output fails on any pair left,
whichever side wrote it.  Exit code 1 when a pair with at least one side inside an asm body violates a rule, or (after
--elide) when any pair does.
"""
import argparse
import re
import subprocess
import sys
import tempfile
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "fault-tolerant-mpc_amd" / "csrc"

TRANS = re.compile(r"^v_(rsq|rcp|sqrt|exp|log|sin|cos|rcp_iflag)_(f16|f32|f64|legacy_f32)")
MFMA_WAIT = {  # mnemonic prefix -> (VALU / WAW wait states, memory-read wait states)
    "v_mfma_f32_16x16x4_f32": (10, 10), "v_mfma_f32_16x16x4f32": (10, 10),
    "v_mfma_f32_32x32x2_f32": (18, 18), "v_mfma_f32_32x32x2f32": (18, 18),
    "v_mfma_f32_4x4x1_16b_f32": (4, 4), "v_mfma_f32_4x4x1f32": (4, 4),
    "v_mfma_f64_16x16x4_f64": (11, 18), "v_mfma_f64_16x16x4f64": (11, 18),
    "v_mfma_f64_4x4x4_4b_f64": (6, 9), "v_mfma_f64_4x4x4f64": (6, 9),
}
MAXW = 18
TWO_DST = re.compile(r"^v_(add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co|div_scale|mad_u64_u32|mad_i64_i32)")
ACCUM_DST = re.compile(r"^v_(fmac|mac|pk_fmac|dot2c|dot4c|dot8c|fmamk|fmaak|movrel|cndmask.*dpp|writelane)")
REG = re.compile(r"\b(v|a|s)\[(\d+):(\d+)\]|\b(v|a|s)(\d+)\b|\b(vcc_lo|vcc_hi|vcc|exec_lo|exec_hi|exec|m0)\b")
SPECIAL = {"vcc": [("s", 106), ("s", 107)], "vcc_lo": [("s", 106)], "vcc_hi": [("s", 107)], "exec": [("s", 126), ("s", 127)],
           "exec_lo": [("s", 126)], "exec_hi": [("s", 127)], "m0": [("s", 124)]}


def regs_of(text):
    out = []
    for m in REG.finditer(text):
        if m.group(1):
            out += [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
        elif m.group(4):
            out.append((m.group(4), int(m.group(5))))
        else:
            out += SPECIAL[m.group(6)]
    return out


def split_ops(s):
    ops, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


class Inst:
    __slots__ = ("mn", "ops", "line", "in_asm", "wr", "rd", "ws", "kind", "dpp", "lanesel", "text", "ab", "target", "callee")

    def __init__(self, text, line, in_asm):
        self.text, self.line, self.in_asm = text, line, in_asm
        self.target = None      # s_setpc_b64 that ends a relaxed long branch: the label it jumps to
        self.callee = None      # s_swappc_b64: the function it calls
        body = text.split(";")[0].strip()
        parts = body.split(None, 1)
        self.mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        self.ops = split_ops(rest)
        mn = self.mn
        self.dpp = ("_dpp" in mn) or bool(re.search(r"\b(row_|quad_perm|wave_|row_newbcast|dpp8)", rest))
        self.ws = 1
        self.lanesel = []
        if mn == "s_nop":
            self.ws = int(self.ops[0], 0) + 1
        if mn.startswith("v_"):
            self.kind = "mfma" if mn.startswith("v_mfma") or mn.startswith("v_smfmac") else "valu"
        elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
            self.kind = "vmem"
        elif mn.startswith("ds_"):
            self.kind = "lds"
        elif mn.startswith("s_"):
            self.kind = "salu"
        else:
            self.kind = "other"
        wr, rd = [], []
        ops = self.ops
        opregs = [regs_of(o) for o in ops]
        if self.kind in ("valu", "mfma"):
            if mn.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap_b32")):
                wr = opregs[0] + opregs[1]
                rd = opregs[0] + opregs[1]
            else:
                ndst = 2 if TWO_DST.match(mn) else 1
                if mn.startswith("v_cmpx"):
                    wr = SPECIAL["exec"]
                    ndst = 1 if (opregs and ops[0].startswith(("s", "vcc", "exec"))) else 0
                for i, r in enumerate(opregs):
                    (wr if i < ndst else rd).extend(r)
                if ACCUM_DST.match(mn) or self.dpp or "_sdwa" in mn:
                    rd += opregs[0] if opregs else []
                if re.match(r"^v_(cndmask_b32(_e32|_dpp|_sdwa)?$|addc_co_u32_e32|subb_co_u32_e32|subbrev_co_u32_e32|div_fmas)", mn) and "vcc" not in rest:
                    rd += SPECIAL["vcc"]
                if mn.startswith(("v_readlane", "v_writelane")) and len(ops) >= 3:
                    self.lanesel = [r for r in opregs[2] if r[0] == "s"]
        elif self.kind == "salu":
            if mn.startswith(("s_cmp", "s_bitcmp", "s_waitcnt", "s_nop", "s_branch", "s_cbranch", "s_barrier", "s_endpgm", "s_setprio",
                              "s_sleep", "s_sethalt", "s_setreg", "s_icache", "s_dcache", "s_store", "s_buffer_store", "s_trap")):
                for r in opregs:
                    rd += r
            else:
                for i, r in enumerate(opregs):
                    (wr if i == 0 else rd).extend(r)
        elif self.kind in ("vmem", "lds"):
            is_load = ("load" in mn or "_read" in mn or "bpermute" in mn or "permute" in mn or ("atomic" in mn and " glc" in rest)) and "lds" not in rest.split()
            for i, r in enumerate(opregs):
                (wr if (i == 0 and is_load) else rd).extend(r)
        self.wr, self.rd = set(wr), set(rd)
        self.ab = set(opregs[1] + opregs[2]) if self.kind == "mfma" and len(opregs) >= 3 else set()


LONG_BRANCH = re.compile(r"\((\.LBB\w+)-\.Lpost_getpc\w*\)")
CALL_SYM = re.compile(r"\b(_Z\w+)@rel32@")


def parse(path):
    """-> {function: items}; items = list of Inst or ('label', name).
    Two pseudo-branches of the final stream are resolved here, so that the backward walk crosses them:
      * a RELAXED LONG BRANCH  s_getpc_b64 sX / s_add_u32 sX, sX, (.LBBn-.Lpost_getpc)&.. / s_addc_u32 / s_setpc_b64 sX
        is an unconditional branch to .LBBn (Inst.target of the s_setpc_b64);
      * a CALL  s_getpc_b64 / s_add_u32 .., SYM@rel32@lo+4 / s_addc_u32 .., SYM@rel32@hi+12 / s_swappc_b64  enters SYM
        (Inst.callee); the callee's  s_setpc_b64 s[30:31]  returns behind it."""
    funcs, cur, name, in_asm = {}, None, None, False
    pend_label, pend_sym = None, None
    for ln, raw in enumerate(open(path), 1):
        s = raw.rstrip("\n")
        st = s.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is None:
            continue
        if st.startswith(".Lfunc_end"):
            funcs[name] = cur
            cur = None
            continue
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.L\w+):", st)
        if m:
            cur.append(("label", m.group(1)))
            continue
        if not st or st.startswith((";", ".", "//")):
            continue
        if re.match(r"^[a-z_0-9]+(\s|$)", st):
            it = Inst(st, ln, in_asm)
            m = LONG_BRANCH.search(st)
            if m and it.mn.startswith("s_add"):
                pend_label = m.group(1)
            m = CALL_SYM.search(st)
            if m and it.mn.startswith("s_add"):
                pend_sym = m.group(1)
            if it.mn == "s_setpc_b64":
                it.target, pend_label = pend_label, None
            if it.mn == "s_swappc_b64":
                it.callee, pend_sym = pend_sym, None
            cur.append(it)
    return funcs


def predecessors(items):
    label_pos = {it[1]: i for i, it in enumerate(items) if isinstance(it, tuple)}
    preds = defaultdict(list)
    for i, it in enumerate(items):
        if isinstance(it, tuple):
            continue
        if it.mn.startswith(("s_branch", "s_cbranch")):
            tgt = it.ops[-1] if it.ops else None
            if tgt in label_pos:
                preds[label_pos[tgt]].append(i)
        elif it.mn == "s_setpc_b64" and it.target in label_pos:
            preds[label_pos[it.target]].append(i)
    return preds


class Program:
    """All functions of the code object with what the backward walk needs to cross function boundaries."""

    def __init__(self, funcs):
        self.funcs = funcs
        self.preds = {n: predecessors(items) for n, items in funcs.items()}
        self.callsites = defaultdict(list)     # callee -> [(caller, position of the s_swappc_b64)]
        self.rets = defaultdict(list)          # function -> positions of its returns (s_setpc_b64 without a label target)
        self.unresolved = []
        for n, items in funcs.items():
            for i, it in enumerate(items):
                if isinstance(it, tuple):
                    continue
                if it.mn == "s_swappc_b64":
                    if it.callee in funcs:
                        self.callsites[it.callee].append((n, i))
                    else:
                        self.unresolved.append((n, it))
                elif it.mn == "s_setpc_b64" and it.target is None:
                    self.rets[n].append(i)


def walk_back(prog, fname, start, budget):
    """Yields (inst, wait states between it and the consumer) for every instruction reachable backwards from position
    `start` of function `fname` (exclusive) with fewer than `budget` wait states in between: across labels (fall-through,
    branches, relaxed long branches), out of a function's entry into every call site, and from behind a call into the
    callee's returns."""
    stack = [(fname, start - 1, 0, False)]     # (function, position, distance, arrived at this s_swappc_b64 from its callee's entry)
    seen = set()

    def push(fn, pos, dist, from_entry=False):
        key = (fn, pos, dist, from_entry)
        if key not in seen:
            seen.add(key)
            stack.append(key)

    while stack:
        fn, pos, dist, from_entry = stack.pop()
        items, preds = prog.funcs[fn], prog.preds[fn]
        while dist < budget:
            if pos < 0:    # function entry: the code in front of every call of this function ran before (kernels: nothing did)
                for cf, cpos in prog.callsites.get(fn, ()):
                    push(cf, cpos, dist, True)
                break
            it = items[pos]
            if isinstance(it, tuple):
                for b in preds.get(pos, ()):
                    push(fn, b, dist)
                pos -= 1
                # fall-through into the label is impossible behind an unconditional transfer
                if pos >= 0 and not isinstance(items[pos], tuple) and items[pos].mn in ("s_branch", "s_endpgm", "s_setpc_b64"):
                    break
                continue
            if it.mn == "s_swappc_b64" and it.callee in prog.funcs and not from_entry:
                # coming from behind the call: what ran last is the callee -- its returns, its body, its entry, and through
                # prog.callsites the s_swappc_b64 itself and the code in front of it
                for r in prog.rets.get(it.callee, ()):
                    push(it.callee, r, dist)
                break
            from_entry = False
            yield it, dist
            dist += it.ws
            pos -= 1


def check_consumer(prog, fname, i, found):
    """Appends the violations whose CONSUMER is instruction i of function fname."""
    items = prog.funcs[fname]
    c = items[i]
    if isinstance(c, tuple) or c.kind in ("salu", "other"):
        return

    def report(rule, need, prod, cons, dist, reg):
        found.append(dict(rule=rule, need=need, have=dist, reg=f"{reg[0]}{reg[1]}", prod=prod, cons=cons,
                          asm=prod.in_asm or cons.in_asm))

    rules = []   # (name, need, producer predicate, registers of interest)
    vrd = {r for r in c.rd if r[0] in "va"}
    srd = {r for r in c.rd if r[0] == "s"}
    if c.kind in ("valu", "mfma"):
        if c.dpp:
            rules.append(("dpp_vgpr", 2, lambda p: p.kind in ("valu", "mfma"), vrd))
            rules.append(("dpp_exec", 5, lambda p: p.kind == "valu", set(SPECIAL["exec"])))
        if c.kind == "valu" and not TRANS.match(c.mn):
            rules.append(("trans_use", 1, lambda p: p.kind == "valu" and bool(TRANS.match(p.mn)), vrd))
        if c.mn.startswith(("v_permlane16_swap", "v_permlane32_swap")):
            rules.append(("permlane_swap", 2, lambda p: p.kind in ("valu", "mfma"), vrd))
        if c.mn.startswith(("v_readlane", "v_readfirstlane")):
            rules.append(("readlane_vgpr", 1, lambda p: p.kind in ("valu", "mfma"), vrd))
        if c.lanesel:
            rules.append(("lane_select", 4, lambda p: p.kind == "valu", set(c.lanesel)))
        if c.mn.startswith("v_div_fmas"):
            rules.append(("div_fmas", 4, lambda p: p.kind == "valu", set(SPECIAL["vcc"])))
        sv = {r for r in srd if r[1] < 124 or r[1] in (106, 107)}
        if sv:
            rules.append(("sgpr_valu", 2, lambda p: p.kind == "valu", sv))
        if c.kind == "mfma":
            rules.append(("valu_mfma", 2, lambda p: p.kind == "valu" and p.in_asm, vrd))
    if c.kind == "vmem" and srd:
        rules.append(("sgpr_vmem", 5, lambda p: p.kind == "valu", srd))
    touched = (c.rd | c.wr) if c.kind == "valu" else c.rd
    vt = {r for r in touched if r[0] in "va"}
    budget = max([r[1] for r in rules] + [MAXW if (vt or c.kind == "mfma") else 0])
    if budget == 0:
        return
    for p, dist in walk_back(prog, fname, i, budget):
        for name, need, pred, regs in rules:
            if dist < need and regs and pred(p):
                hit = p.wr & regs
                if hit:
                    report(name, need, p, c, dist, sorted(hit)[0])
        if p.kind == "mfma" and c.kind == "mfma" and c.ab:
            # an MFMA result read as the A or B operand of another MFMA (srcC of the same shape is forwarded by the hardware)
            for pre, (wv, wm) in MFMA_WAIT.items():
                if p.mn.startswith(pre):
                    hit = p.wr & c.ab
                    if hit and dist < wv:
                        report("mfma_srcab", wv, p, c, dist, sorted(hit)[0])
                    break
        if p.kind == "mfma" and vt and c.kind != "mfma":
            for pre, (wv, wm) in MFMA_WAIT.items():
                if p.mn.startswith(pre):
                    need = wm if c.kind in ("vmem", "lds") else wv
                    hit = p.wr & vt
                    if hit and dist < need:
                        report("mfma_use", need, p, c, dist, sorted(hit)[0])
                    break


def check_function(prog, fname):
    found = []
    for i in range(len(prog.funcs[fname])):
        check_consumer(prog, fname, i, found)
    return found


# ---------------------------------------------------------------------------------------------------------
# Wait-state elision.  Every asm body opens with a fixed `s_nop` that covers "whatever hipcc scheduled right before"
# (the author cannot know); in the final stream most of them are not needed.  Each asm-side s_nop is lowered to the
# smallest count for which every consumer in the window behind it still passes the rule table above (greedy, in
# program order, against the CFG-aware checker; kept as written when a branch follows within the window).
# ---------------------------------------------------------------------------------------------------------
ELIDE_WINDOW = 24


def elide_function(prog, fname):
    items = prog.funcs[fname]
    saved = removed = 0
    for i, it in enumerate(items):
        if isinstance(it, tuple) or it.mn != "s_nop" or not it.in_asm:
            continue
        win = items[i + 1:i + 1 + ELIDE_WINDOW]
        if any((not isinstance(x, tuple)) and x.mn.startswith(("s_branch", "s_cbranch", "s_setpc", "s_swappc", "s_endpgm", "s_call")) for x in win):
            continue
        orig = it.ws
        for w in range(0, orig + 1):
            it.ws = w
            found = []
            for j in range(i + 1, min(len(items), i + 1 + ELIDE_WINDOW)):
                check_consumer(prog, fname, j, found)
                if found:
                    break
            if not found:
                break
        saved += orig - it.ws
        removed += 1 if it.ws == 0 else 0
    return saved, removed


def elide(asm_in, asm_out, lower=True):
    """Writes asm_out = asm_in with the asm-side wait states minimised; returns statistics.  The result is audited again."""
    funcs = parse(asm_in)
    prog = Program(funcs)
    new_ws = {}
    stats = {}
    for name, items in funcs.items():
        before = sum(it.ws for it in items if not isinstance(it, tuple) and it.mn == "s_nop" and it.in_asm)
        saved, removed = elide_function(prog, name) if lower else (0, 0)
        stats[name] = dict(asm_wait_states=before, saved=saved, nops_removed=removed)
        for it in items:
            if not isinstance(it, tuple) and it.mn == "s_nop" and it.in_asm:
                new_ws[it.line] = it.ws
    # pairs the compiler scheduled itself below the table (seen once: an MFMA chain, a branch, v_accvgpr_read of the
    # result three wait states later): pad the consumer rather than argue about which side is right
    pad = {}
    for name, items in funcs.items():
        for f in check_function(prog, name):
            if not f["asm"]:
                ln = f["cons"].line
                pad[ln] = max(pad.get(ln, 0), f["need"] - f["have"])
        stats[name]["padded"] = sum(1 for it in items if not isinstance(it, tuple) and it.line in pad)
    out = []
    for ln, raw in enumerate(open(asm_in), 1):
        if ln in new_ws:
            w = new_ws[ln]
            if w == 0:
                continue
            raw = re.sub(r"s_nop\s+\S+", f"s_nop {w - 1}", raw, count=1)
        if ln in pad:
            w = pad[ln]
            while w > 0:
                out.append(f"\ts_nop {min(w, 8) - 1}\n")
                w -= min(w, 8)
        out.append(raw)
    Path(asm_out).write_text("".join(out))
    return stats


def build_asm(out):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-mllvm",
           "-disable-machine-licm", "-Wno-unused-function", "-Wno-unused-command-line-argument", "-o", str(out), "ftmpc_capi.hip"]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def audit(asm_path):
    funcs = parse(asm_path)
    prog = Program(funcs)
    if prog.unresolved:
        raise SystemExit("call targets not resolved (the audit cannot follow them): " + ", ".join(f"{n} L{it.line}" for n, it in prog.unresolved))
    summary = {}
    for name, items in funcs.items():
        found = check_function(prog, name)
        insts = [it for it in items if not isinstance(it, tuple)]
        summary[name] = dict(n_inst=len(insts), n_asm_inst=sum(1 for it in insts if it.in_asm),
                             asm=[f for f in found if f["asm"]], compiler=[f for f in found if not f["asm"]])
    return summary


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--asm", help="device assembly to audit (default: compile csrc/ftmpc_capi.hip)")
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--elide", metavar="OUT", help="write the assembly with minimised asm-side wait states (and padded compiler-side pairs) to OUT, then audit OUT")
    ap.add_argument("--pad-only", action="store_true", help="with --elide: keep the asm-side wait states as written, only pad")
    args = ap.parse_args()
    if args.asm:
        path = Path(args.asm)
    else:
        path = Path(tempfile.mkdtemp(prefix="ftmpc_haz_")) / "ftmpc_dev.s"
        build_asm(path)
    if args.elide:
        st = elide(path, args.elide, lower=not args.pad_only)
        for name, v in st.items():
            if v["asm_wait_states"]:
                print(f"elide {re.sub(r'^_ZN5ftmpc[0-9]+', '', name)[:50]:52s} asm wait states {v['asm_wait_states']:6d} -> {v['asm_wait_states'] - v['saved']:6d}"
                      f"  ({v['nops_removed']} s_nop removed)" + (f", {v['padded']} compiler-side pairs padded" if v.get("padded") else ""))
        path = Path(args.elide)
    summary = audit(path)
    bad = 0
    for name, s in summary.items():
        short = re.sub(r"^_ZN5ftmpc\d+", "", name)[:60]
        print(f"{short:62s} {s['n_inst']:7d} instructions, {s['n_asm_inst']:5d} inside asm: "
              f"{len(s['asm'])} asm-side violations, {len(s['compiler'])} compiler-only")
        for f in s["asm"][:50]:
            bad += 1
            print(f"   !! {f['rule']}: need {f['need']} wait states, have {f['have']} on {f['reg']}\n"
                  f"        producer L{f['prod'].line}{' [asm]' if f['prod'].in_asm else ''}: {f['prod'].text}\n"
                  f"        consumer L{f['cons'].line}{' [asm]' if f['cons'].in_asm else ''}: {f['cons'].text}")
        if args.elide:
            bad += len(s["compiler"])
        if args.verbose or (args.elide and s["compiler"]):
            by = defaultdict(list)
            for f in s["compiler"]:
                by[f["rule"]].append(f)
            for r, fs in by.items():
                f = fs[0]
                print(f"   (compiler-only) {r}: {len(fs)} compiler pairs below {f['need']}; e.g. have {f['have']}: L{f['prod'].line} {f['prod'].text}  ->  L{f['cons'].line} {f['cons'].text}")
    print("asm-side violations:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
