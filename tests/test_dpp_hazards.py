"""CPU check of the generated gfx950 ISA: no VALU-write -> DPP-read hazard around the inline-asm DPP instructions.

`v_fmac_f64_dpp` (mpc_wave.hip, rollout_fast.hip) and `v_fmac_f32_dpp` / `v_mul_f32_dpp` (lqr_tiled_core.h) are inline asm, and LLVM's hazard
recognizer does not pad hazards around inline asm: the two wait states a DPP read of a VGPR needs after a VALU write of it come
from a separate `asm("s_nop 1" : "+v"(v))` statement in the source.  Nothing in the compiler keeps a register-allocator copy
(`v_mov`) from landing between the two statements, which would silently give wrong values.  So this test compiles the
translation units to assembly (`hipcc --cuda-device-only -S`, no GPU needed) and checks, for every `*_dpp` instruction, that no
VALU instruction wrote its DPP source operand within the previous two wait states (an instruction in between = 1 wait state,
`s_nop N` = N + 1)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zopt_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-pragma-unroll-threshold=1000000", "--cuda-device-only", "-S"]
REG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def _regs(op):
    """VGPR numbers named by an operand like v7 or v[40:41]; empty for anything else (SGPRs, literals, vcc, ...)"""
    m = REG.match(op.strip())
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def _instructions(asm):
    """(mnemonic, operands) of every instruction line; directives, labels and comments dropped"""
    for ln in asm.splitlines():
        ln = ln.split(";")[0].strip()
        if not ln or ln.startswith(".") or ln.endswith(":") or ln.startswith("//"):
            continue
        parts = ln.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        # DPP modifiers ride on the last operand ("v[92:93] row_newbcast:0 row_mask:0xf ..."): keep the register part
        ops = [o.split()[0] if o else o for o in ops]
        yield mn, ops


def scan(asm):
    """-> (number of DPP instructions, list of violations)"""
    window = []          # (wait states this instruction contributes, VGPRs it writes as a VALU instruction)
    n_dpp, bad = 0, []
    for mn, ops in _instructions(asm):
        if mn.endswith("_dpp"):
            n_dpp += 1
            src0 = _regs(ops[1]) if len(ops) > 1 else set()
            ws = 0
            for states, written in reversed(window):
                if ws >= 2:
                    break
                if written & src0:
                    bad.append((mn, ops, ws))
                    break
                ws += states
        if mn == "s_nop":
            window.append((int(ops[0], 0) + 1, set()))
        else:
            written = _regs(ops[0]) if (mn.startswith("v_") and ops) else set()
            window.append((1, written))
        if len(window) > 8:
            window.pop(0)
    return n_dpp, bad


def test_scanner_sees_a_planted_hazard():
    ok = "\tv_add_f64 v[2:3], v[4:5], v[6:7]\n\ts_nop 1\n\tv_fmac_f64_dpp v[8:9], v[2:3], v[10:11] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
    bad = "\tv_add_f64 v[2:3], v[4:5], v[6:7]\n\ts_nop 0\n\tv_fmac_f64_dpp v[8:9], v[2:3], v[10:11] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
    copy = "\tv_add_f64 v[2:3], v[4:5], v[6:7]\n\ts_nop 1\n\tv_mov_b32_e32 v3, v20\n\tv_fmac_f64_dpp v[8:9], v[2:3], v[10:11] row_newbcast:0\n"
    assert scan(ok) == (1, [])
    assert scan(bad)[1] and scan(copy)[1]


@pytest.mark.parametrize("src", ["mpc_wave.hip", "lqr_backward_tiled_f32.hip", "rollout_fast.hip"])
def test_no_valu_write_to_dpp_read_hazard_in_generated_isa(src, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path / (src + ".s")
    p = subprocess.run([HIPCC] + FLAGS + [os.path.join(CSRC, src), "-o", str(out)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    n_dpp, bad = scan(out.read_text())
    assert n_dpp > 100, f"{src}: expected the inline-asm DPP instructions in the ISA, found {n_dpp}"
    assert not bad, f"{src}: {len(bad)} DPP reads within two wait states of a VALU write of their source, e.g. {bad[:3]}"
