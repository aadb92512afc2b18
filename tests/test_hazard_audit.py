"""CPU: static audit of the software-managed instruction hazards of the gfx950 code object (scripts/check_hazards.py).
Inline asm is invisible to hipcc's hazard recognizer; a missing wait state there once corrupted a few instances per
ten thousand on the GPU.  The audit compiles the kernels to device assembly here (no GPU) and checks every
producer/consumer pair of the final instruction stream; a second pass proves the checker can see such faults by
deleting the wait states on purpose."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "scripts"))
import check_hazards as ch  # noqa: E402


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    p = tmp_path_factory.mktemp("haz") / "ftmpc_dev.s"
    ch.build_asm(p)
    return p


def test_no_hazard_in_any_kernel(device_asm):
    summary = ch.audit(device_asm)
    kernels = [n for n in summary if "ftmpc_solve_f32_kernel" in n]
    assert len(kernels) == 3 and all(summary[n]["n_asm_inst"] > 2000 for n in kernels)     # the asm sites were seen
    assert any("ftmpc_solve_f64_kernel" in n and summary[n]["n_asm_inst"] > 100 for n in summary)
    bad = [(n, f["rule"], f["prod"].text, f["cons"].text) for n, s in summary.items() for f in s["asm"]]
    assert not bad, bad[:5]
    # Pairs the compiler scheduled itself satisfy the table everywhere except at one pattern of the workgroup kernel: an
    # MFMA chain, a branch, v_accvgpr_read of the result 3 wait states later (table: 10).  That one is real -- the GPU
    # returned wrong factors until the consumer was padded (scripts/check_hazards.py header) -- so the build pads it.
    comp = [(n, f) for n, s in summary.items() for f in s["compiler"]]
    assert all(f["rule"] in ("mfma_use", "mfma_srcab") and any(k in n for k in ("ftmpc_solve_wg32_kernel", "ftmpc_solve_ws32_kernel", "ftmpc_solve_hull32_kernel", "ftmpc_solve_f32_kernelILi8"))
               for n, f in comp), [(n, f["rule"]) for n, f in comp if "wg32" not in n and "ws32" not in n][:5]
    # (the workgroup kernels 7 and 8 share the factorisation where the pattern sits; kernel 11 with the terminal set has one such
    # read behind its float64 / fp32 branch; kernel 2 <8> since the polish: one v_mov over the first register of a finished MFMA's
    # destination 9 wait states after it, table 10 -- the build pads every one of them, see test_elided_stream_is_clean_and_shorter)


def test_elided_stream_is_clean_and_shorter(device_asm, tmp_path):
    """What the build ships: the asm-side wait states lowered to what the final schedule needs, compiler-side pairs below
    the table padded (csrc/Makefile step 2).  The result passes the audit with NO pair left on either side and drops
    most of the conservative s_nops of the fp32 kernels."""
    out = tmp_path / "final.s"
    stats = ch.elide(device_asm, out)
    summary = ch.audit(out)
    assert not [f for s in summary.values() for f in s["asm"]] and sum(len(s["compiler"]) for s in summary.values()) == 0
    k8 = next(v for n, v in stats.items() if "ftmpc_solve_f32_kernelILi8" in n)
    assert k8["saved"] > 0.5 * k8["asm_wait_states"]
    assert sum(v.get("padded", 0) for v in stats.values()) == len({f["cons"].line for s in ch.audit(device_asm).values() for f in s["compiler"]})
    # every change is an s_nop line lowered, dropped or added: nothing else differs
    a = [l for l in device_asm.read_text().split("\n") if not l.strip().startswith("s_nop")]
    b = [l for l in out.read_text().split("\n") if not l.strip().startswith("s_nop")]
    assert a == b


def test_checker_sees_deleted_wait_states(device_asm, tmp_path):
    lines, out, in_asm = device_asm.read_text().split("\n"), [], False
    for l in lines:
        st = l.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
        elif st.startswith(";;#ASMEND"):
            in_asm = False
        if in_asm and st.startswith("s_nop"):
            continue
        out.append(l)
    mut = tmp_path / "mut.s"
    mut.write_text("\n".join(out))
    rules = {f["rule"] for s in ch.audit(mut).values() for f in s["asm"]}
    # (whether a v_rsq lands right in front of an asm body depends on the compiler's schedule of the day: the
    # transcendental rule is pinned by the hand-written sequence below instead)
    assert {"permlane_swap", "dpp_vgpr"} <= rules


def test_rule_table_on_hand_written_sequences(tmp_path):
    def run(body):
        p = tmp_path / "t.s"
        p.write_text("_Z1kv:\n" + "\n".join("\t" + l for l in body) + "\n.Lfunc_end0:\n")
        return sorted({f["rule"] for f in ch.check_function(ch.Program(ch.parse(p)), "_Z1kv")})
    assert run(["v_rsq_f32_e32 v1, v2", "v_mul_f32_e32 v3, v1, v1"]) == ["trans_use"]
    assert run(["v_rsq_f32_e32 v1, v2", "s_nop 0", "v_mul_f32_e32 v3, v1, v1"]) == []
    assert run(["v_mov_b32_e32 v1, v2", "s_nop 0", "v_add_f32_dpp v3, v1, v1 row_ror:8 row_mask:0xf bank_mask:0xf"]) == ["dpp_vgpr"]
    assert run(["v_mov_b32_e32 v1, v2", "s_nop 1", "v_add_f32_dpp v3, v1, v1 row_ror:8 row_mask:0xf bank_mask:0xf"]) == []
    assert run(["v_mov_b32_e32 v1, v2", "v_permlane32_swap_b32_e32 v1, v4"]) == ["permlane_swap"]
    assert run(["v_cmp_lt_f32_e32 vcc, v1, v2", "v_cndmask_b32_e32 v3, v4, v5, vcc"]) == ["sgpr_valu"]
    assert run(["s_mov_b32 vcc_lo, 1", "s_mov_b32 vcc_hi, 0", "v_cndmask_b32_e32 v3, v4, v5, vcc"]) == []      # SALU writer: interlocked
    assert run(["v_readlane_b32 s3, v2, 1", "s_nop 2", "v_readlane_b32 s4, v5, s3"]) == ["lane_select"]
    assert run(["v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]"] + ["s_nop 7"] + ["v_add_f32_e32 v9, v0, v0"]) == ["mfma_use"]
    assert run(["v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]"] + ["s_nop 7", "s_nop 1"] + ["v_add_f32_e32 v9, v0, v0"]) == []
    assert run(["v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]", "v_mfma_f32_16x16x4_f32 v[0:3], v6, v7, v[0:3]"]) == []   # accumulate chain
    # the sequence hipcc emitted in the workgroup kernel (wrong results on the GPU until padded)
    assert run(["v_mfma_f32_16x16x4_f32 a[0:3], v25, v7, a[0:3]", "v_mfma_f32_16x16x4_f32 a[4:7], v25, v15, a[4:7]", "s_cbranch_execnz .LBB0_8",
                "s_branch .LBB0_9", ".LBB0_8:", "s_waitcnt vmcnt(1)", "v_accvgpr_read_b32 v7, a3", ".LBB0_9:"]) == ["mfma_use"]
    # across a RELAXED LONG BRANCH (s_getpc / s_add / s_addc / s_setpc): four wait states of SALU between the MFMA and the read
    assert run(["v_mfma_f32_16x16x4_f32 a[0:3], v25, v7, a[0:3]", "s_getpc_b64 s[4:5]", ".Lpost_getpc3:", "s_add_u32 s4, s4, (.LBB0_9-.Lpost_getpc3)&4294967295",
                "s_addc_u32 s5, s5, (.LBB0_9-.Lpost_getpc3)>>32", "s_setpc_b64 s[4:5]", ".LBB0_8:", "s_nop 0", ".LBB0_9:", "v_accvgpr_read_b32 v7, a3"]) == ["mfma_use"]
    # across a branch edge: the producer sits before the branch, the consumer behind the label
    assert run(["v_mov_b32_e32 v1, v2", "s_cbranch_scc1 .LBB0_2", "s_nop 3", ".LBB0_2:", "v_add_f32_dpp v3, v1, v1 row_ror:8 row_mask:0xf bank_mask:0xf"]) == ["dpp_vgpr"]


def test_walk_crosses_calls_and_returns(tmp_path):
    """Producer in the caller / consumer at the callee's entry, and producer at the callee's end / consumer behind the call:
    the audit follows s_swappc_b64 into the callee and s_setpc_b64 s[30:31] back (function boundaries were blind)."""
    def run(caller, callee):
        p = tmp_path / "c.s"
        call = ["s_getpc_b64 s[0:1]", "s_add_u32 s0, s0, _Z1fv@rel32@lo+4", "s_addc_u32 s1, s1, _Z1fv@rel32@hi+12", "s_swappc_b64 s[30:31], s[0:1]"]
        body = []
        for l in caller:
            body += call if l == "CALL" else [l]
        p.write_text("_Z1kv:\n" + "\n".join("\t" + l for l in body) + "\n\ts_endpgm\n.Lfunc_end0:\n_Z1fv:\n" +
                     "\n".join("\t" + l for l in callee) + "\n\ts_setpc_b64 s[30:31]\n.Lfunc_end1:\n")
        prog = ch.Program(ch.parse(p))
        assert not prog.unresolved and prog.callsites["_Z1fv"] and prog.rets["_Z1fv"]
        return {n: sorted({f["rule"] for f in ch.check_function(prog, n)}) for n in ("_Z1kv", "_Z1fv")}
    # MFMA result written in front of the call, read by the callee's first instruction (4 SALU wait states in between)
    r = run(["v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]", "CALL"], ["v_add_f32_e32 v9, v0, v0"])
    assert r["_Z1fv"] == ["mfma_use"]
    r = run(["v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]", "s_nop 7", "CALL"], ["v_add_f32_e32 v9, v0, v0"])
    assert r["_Z1fv"] == []
    # transcendental at the callee's end, consumed right behind the call: only the return lies in between
    r = run(["CALL", "v_add_f32_dpp v3, v1, v1 row_ror:8 row_mask:0xf bank_mask:0xf"], ["v_mov_b32_e32 v1, v2"])
    assert r["_Z1kv"] == ["dpp_vgpr"]
    r = run(["CALL", "s_nop 0", "v_add_f32_dpp v3, v1, v1 row_ror:8 row_mask:0xf bank_mask:0xf"], ["v_mov_b32_e32 v1, v2"])
    assert r["_Z1kv"] == []
