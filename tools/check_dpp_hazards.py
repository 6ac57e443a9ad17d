#!/usr/bin/env python3
"""Static check of the gfx950 "VALU write -> DPP / v_permlane*_swap read needs 2 wait states" rule on
compiled kernels (the compiler does not look inside asm statements, la_sort32.h / la_sort64.h place the
pads by hand).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S -o /tmp/x.s kafka_lag_based_assignor_amd/csrc/la_wave_tile.hip
    python tools/check_dpp_hazards.py /tmp/x.s [more.s ...]

Linear scan per function.  A label resets nothing (fall-through is checked); a branch target is also
reached with whatever was issued before the jump, which this tool does not follow -- the blocks in question
are straight-line code.  Wait states: every instruction issued in between counts 1, `s_nop N` counts N+1.
"""
import re
import sys

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_ops(rest):
    rest = re.split(r"\s+(quad_perm|row_|wave_|bank_mask|row_mask|bound_ctrl|dst_sel|src0_sel|offset|off\b|sc0|sc1|nt\b)", rest)[0]
    return [o.strip() for o in rest.split(",") if o.strip()]


def check(path):
    bad = 0
    fn = None
    recent = []            # list of (wait_states_ago_counter, written vgprs) -- we keep (age, regs)
    n_dpp = n_swap = 0
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        if not s or s.startswith((";", "//", ".")) and not s.endswith(":"):
            continue
        if s.endswith(":"):
            if s.startswith("_Z") or not s.startswith("."):
                fn = s[:-1]
                recent = []
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        parts = s.split(None, 1)
        mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        is_dpp = bool(re.search(r"quad_perm|row_shl|row_shr|row_ror|row_mirror|row_half_mirror|row_bcast|wave_", rest))
        is_swap = mn.startswith("v_permlane") and "swap" in mn
        ops = split_ops(rest)
        # reads that are subject to the rule
        hazard_reads = set()
        if is_dpp and ops:
            n_dpp += 1
            srcs = ops[1:]
            if srcs and srcs[0] in ("vcc", "vcc_lo"):
                srcs = srcs[1:]
            if srcs:
                hazard_reads = regs(srcs[0])
        elif is_swap:
            n_swap += 1
            for o in ops[:2]:
                hazard_reads |= regs(o)
        for age, written in recent:
            if age < 2 and hazard_reads & written:
                bad += 1
                print("%s:%d: in %s: `%s` reads v%s through DPP/permlane %d wait state(s) after its VALU write"
                      % (path, ln, fn, s, sorted(hazard_reads & written), age))
        # age everything by what this instruction contributes
        step = 1
        if mn == "s_nop":
            step = int(ops[0], 0) + 1 if ops else 1
        recent = [(a + step, w) for a, w in recent if a + step < 4]
        # VALU writes
        if mn.startswith("v_") and not mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
            w = regs(ops[0]) if ops else set()
            if is_swap and len(ops) > 1:
                w |= regs(ops[1])
            if w:
                recent.append((0, w))
    print("%s: %d DPP reads, %d permlane swaps checked, %d hazards" % (path, n_dpp, n_swap, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
