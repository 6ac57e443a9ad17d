"""Static ISA checks on the compiled kernels (no GPU needed: hipcc cross-compiles gfx950).

The instruction-level sorting networks (csrc/la_sort32.h, la_sort64.h) place their own wait states around DPP /
v_permlane*_swap reads and declare what their SALU ops clobber; the compiler does not look inside asm statements.
Two bugs of that class were found the hard way in round 1 (a missing "scc" clobber, a DPP read one wait state
after a compiler-generated write), both invisible to most parity tests.  This test recompiles the translation
units that use the networks and runs tools/check_dpp_hazards.py over the ISA.
"""
import importlib.util
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kafka_lag_based_assignor_amd", "csrc")
UNITS = ["la_block.hip", "la_large.hip", "la_wave_tile_l32.hip", "la_wave_tile_l8.hip"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(_hipcc()):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa")

    def compile_one(unit):
        dst = os.path.join(str(out), unit.replace(".hip", ".s"))
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-o", dst, os.path.join(CSRC, unit)], stderr=subprocess.DEVNULL)
        return dst

    with ThreadPoolExecutor(max_workers=4) as pool:
        return list(pool.map(compile_one, UNITS))


def test_no_dpp_read_within_two_wait_states_of_a_valu_write(isa, capsys):
    spec = importlib.util.spec_from_file_location("check_dpp_hazards", os.path.join(ROOT, "tools", "check_dpp_hazards.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    for path in isa:
        assert chk.check(path) == 0, path
    out = capsys.readouterr().out
    checked = sum(int(m) for m in re.findall(r"(\d+) DPP reads", out))
    assert checked > 10000          # the networks really are in these units


def _asm_statements(src):
    """Every asm volatile( ... ) of a source text, by parenthesis matching (string literals skipped), with its line."""
    for m in re.finditer(r"asm\s+volatile\s*\(", src):
        i, depth = m.end(), 1
        while depth:
            c = src[i]
            if c == '"':
                i += 1
                while src[i] != '"':
                    i += 2 if src[i] == "\\" else 1
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
            i += 1
        yield src.count("\n", 0, m.start()) + 1, src[m.end():i - 1]


def test_asm_statements_with_salu_logic_declare_scc():
    # s_and / s_or / s_xor / s_andn2 ... write SCC; an asm statement that contains one must say so, or the
    # compiler keeps a live SCC across it (the bug: a loop condition's s_cmp result straddling a network block).
    # The compare-exchange statements take their "keep my own record" fix-up as a macro parameter (FIX: LA_FIX_S is
    # the s_xor_b64 form, LA_FIX_V the VALU form), so a statement with a FIX slot counts as one that may hold SALU logic.
    pat = re.compile(r"\bs_(and|or|xor|andn2|orn2|nand|nor|xnor|not|add|sub|cmp|bfe|lshl|lshr)[a-z0-9_]*\b")
    offenders = []
    seen = 0
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith((".h", ".hip")):
            continue
        src = open(os.path.join(CSRC, name)).read()
        fix_macros = {m.group(1) for m in re.finditer(r"#define\s+(LA_FIX_[A-Z0-9_]+)\b[^\n]*", src) if pat.search(m.group(0))}
        for line, body in _asm_statements(src):
            may_hold_salu = bool(pat.search(body)) or bool(re.search(r"\bFIX\b", body)) or any(f in body for f in fix_macros)
            if may_hold_salu:
                seen += 1
                if '"scc"' not in body:
                    offenders.append("%s:%d: %s" % (name, line, " ".join(body.split())[:100]))
    assert seen >= 3            # the check is looking at the right statements
    assert not offenders, offenders
    # and the SALU fix-up macro exists where the statements expect it
    assert "s_xor_b64 vcc" in open(os.path.join(CSRC, "la_sort64.h")).read()


def test_lds_exchanges_stay_lds_instructions(isa):
    """Round 2 regression: the two exchange buffers of the greedy's cross-wavefront steps were first kept as an array of
    two pointers indexed at run time; hipcc put the array in scratch memory, the pointers came back as generic ones and
    every LDS access of the exchange became a flat_load / flat_store (the full network got 15 % slower, silently).  No
    kernel of this library dereferences a pointer it cannot place: the ISA must hold no flat_* memory instruction, and
    the hot kernels no dynamically indexed scratch."""
    for path in isa:
        text = open(path).read()
        flat = re.findall(r"^\s*(flat_(?:load|store|atomic)\w*)", text, re.M)
        # (la_block: one flat_load per kernel entry -- the topic index comes from a device list or from the kernel
        # arguments, a select between two address spaces, once per workgroup)
        allowed = 2 if os.path.basename(path) == "la_block.s" else 0
        assert len(flat) <= allowed, "%s: %d flat memory instructions (%s ...)" % (os.path.basename(path), len(flat), flat[0])
        # spills are tolerated (a handful in the one-workgroup greedy at its 128-VGPR cap); scratch addressed through an
        # SGPR offset is a private ARRAY in memory, which is what the bug looked like
        dyn = re.findall(r"^\s*scratch_(?:load|store)\w*\s+[^;\n]*\bs\d+\b[^;\n]*$", text, re.M)
        assert not dyn, "%s: dynamically addressed scratch: %s" % (os.path.basename(path), dyn[0].strip())


def test_the_greedy_chain_keeps_nothing_in_scratch(isa):
    """Round 4 regression: the one-workgroup greedy (1 024 threads: a 128-VGPR cap) had hipcc hoist what a round derives from
    the thread index out of the loop of rounds and park it in scratch memory -- reloads in the middle of a dependent chain.
    The thread index is opaque once per round now (greedy_rounds_packed): no kernel of the chain may need scratch at all."""
    text = open([p for p in isa if os.path.basename(p) == "la_large.s"][0]).read()
    found = 0
    for m in re.finditer(r"^(_ZN[^\n:]*greedy_rounds_kernel[^\n:]*):\s*;.*?^; ScratchSize: (\d+)", text, re.M | re.S):
        found += 1
        assert int(m.group(2)) == 0, "%s keeps %s bytes per lane in scratch" % (m.group(1), m.group(2))
    assert found >= 4, "greedy_rounds_kernel<1 / 2 / 4 / 8> not found in the ISA (%d)" % found
