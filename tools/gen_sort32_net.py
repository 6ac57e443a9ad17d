#!/usr/bin/env python3
"""Generates kafka_lag_based_assignor_amd/csrc/la_sort32_net.h: the bitonic sort of 64 x E 32-bit keys held by ONE wavefront
(E keys per lane, element index i = lane * E + r, ascending on exit) as ONE asm statement per E, scheduled by this script.

Why generated.  la_sort32.h builds the same sort from small asm blocks (one compare-exchange each) that the compiler strings
together: every block pads for the worst case of what may precede it (s_nop), and the 128-key sort of the block path's greedy
round came out at ~300 issue slots.  A lone wavefront issues one instruction every ~4 cycles whatever its kind, so the chain of
rounds is a count of issue slots.  Here the whole network is laid out at once: the moves of all registers of a step first, then
their v_med3_u32, and a wait state (s_nop) only where the "VALU write -> DPP / v_permlane*_swap read needs 2 wait states" rule of
the CDNA ISA is not already met by the instructions in between -- the script tracks the distance of every register's last write.

Network: the classic bitonic sorter (stage K = 2 .. N, steps j = K/2 .. 1, partner i ^ j, ascending where (i & K) == 0).  With
v_med3_u32 the direction is data: med3(a, b, 0) = min, med3(a, b, ~0) = max, so a step is
    DIR = D[j] ^ D[K]                       (one VALU per step; D[x] = all ones on the lanes whose element index has bit x set)
    t = key of lane ^ (j / E)               (DPP move; two moves for lane ^ 4; v_mov + v_permlane{16,32}_swap for lane ^ 16 / ^ 32)
    key = med3(key, t, DIR)
and an in-lane step (j < E) is med3(a, b, DIRK) / med3(a, b, ~DIRK) on the two registers.
"""
import os
import sys

DPP = {1: "quad_perm:[1,0,3,2]", 2: "quad_perm:[2,3,0,1]", 8: "row_ror:8"}
TAIL = " row_mask:0xf bank_mask:0xf"


class Sched:
    def __init__(self):
        self.out = []
        self.last_write = {}          # register -> index (in self.out) of the instruction that wrote it last

    def emit(self, text, writes=(), dpp_reads=()):
        need = 0
        for r in dpp_reads:
            if r in self.last_write:
                between = len(self.out) - 1 - self.last_write[r]
                need = max(need, 2 - between)
        if need > 0:
            self.out.append("s_nop %d" % (need - 1))      # s_nop N = N + 1 wait states, one issue slot
        self.out.append(text)
        for r in writes:
            self.last_write[r] = len(self.out) - 1


def gen(E):
    N = 64 * E
    s = Sched()
    X = ["%%[x%d]" % r for r in range(E)]
    T = ["%%[t%d]" % r for r in range(E)]
    D = {1: "%[d1]", 2: "%[d2]", 4: "%[d4]", 8: "%[d8]", 16: "%[d16]", 32: "%[d32]"}
    DIR, NDIR = "%[dir]", "%[ndir]"
    # registers written by compiler code before the statement: treat as written "just now"
    for r in range(E):
        s.last_write[X[r]] = -1
    s.out.append("; sort of %d keys" % N)
    s.last_write = {k: 0 for k in X}      # index 0 = the comment line: distance counts from here
    K = 2
    while K <= N:
        j = K // 2
        while j >= 1:
            # direction of element i: ascending where (i & K) == 0; the lower index of a pair (bit j clear) keeps the min there
            if j >= E:
                lj = j // E
                if K == N:
                    dirv = D[lj]
                elif K >= E:
                    s.emit("v_xor_b32 %s, %s, %s" % (DIR, D[lj], D[K // E]), writes=[DIR])
                    dirv = DIR
                else:
                    raise AssertionError
                if lj in DPP:
                    for r in range(E):
                        s.emit("v_mov_b32_dpp %s, %s %s%s" % (T[r], X[r], DPP[lj], TAIL), writes=[T[r]], dpp_reads=[X[r]])
                elif lj == 4:
                    for r in range(E):
                        s.emit("v_mov_b32_dpp %s, %s row_half_mirror%s" % (T[r], X[r], TAIL), writes=[T[r]], dpp_reads=[X[r]])
                    for r in range(E):
                        s.emit("v_mov_b32_dpp %s, %s quad_perm:[3,2,1,0]%s" % (T[r], T[r], TAIL), writes=[T[r]], dpp_reads=[T[r]])
                else:
                    swap = "v_permlane16_swap_b32" if lj == 16 else "v_permlane32_swap_b32"
                    for r in range(E):
                        s.emit("v_mov_b32 %s, %s" % (T[r], X[r]), writes=[T[r]])
                    for r in range(E):
                        # after the swap both lanes of a pair hold (lower's, upper's) key in (x, t)
                        s.emit("%s %s, %s" % (swap, X[r], T[r]), writes=[X[r], T[r]], dpp_reads=[X[r], T[r]])
                for r in range(E):
                    s.emit("v_med3_u32 %s, %s, %s, %s" % (X[r], X[r], T[r], dirv), writes=[X[r]])
            else:
                # in-lane step: registers r and r | j
                if K < E:
                    pairs = [(r, r | j, ((r & K) != 0)) for r in range(E) if (r & j) == 0]
                    for a, b, desc in pairs:
                        lo, hi = (b, a) if desc else (a, b)
                        s.emit("v_min_u32 %s, %s, %s" % (T[a], X[a], X[b]), writes=[T[a]])
                        s.emit("v_max_u32 %s, %s, %s" % (X[hi], X[a], X[b]), writes=[X[hi]])
                        s.emit("v_mov_b32 %s, %s" % (X[lo], T[a]), writes=[X[lo]])
                elif K == N:
                    for r in range(E):
                        if (r & j) == 0:
                            a, b = r, r | j
                            s.emit("v_min_u32 %s, %s, %s" % (T[a], X[a], X[b]), writes=[T[a]])
                            s.emit("v_max_u32 %s, %s, %s" % (X[b], X[a], X[b]), writes=[X[b]])
                            s.emit("v_mov_b32 %s, %s" % (X[a], T[a]), writes=[X[a]])
                else:
                    dk = D[K // E]
                    s.emit("v_not_b32 %s, %s" % (NDIR, dk), writes=[NDIR])
                    for r in range(E):
                        if (r & j) == 0:
                            a, b = r, r | j
                            s.emit("v_med3_u32 %s, %s, %s, %s" % (T[a], X[a], X[b], dk), writes=[T[a]])      # lower register
                            s.emit("v_med3_u32 %s, %s, %s, %s" % (X[b], X[a], X[b], NDIR), writes=[X[b]])
                            s.emit("v_mov_b32 %s, %s" % (X[a], T[a]), writes=[X[a]])
            j //= 2
        K *= 2
    return [l for l in s.out if not l.startswith(";")]


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "kafka_lag_based_assignor_amd", "csrc", "la_sort32_net.h")
    out = []
    out.append("// la_sort32_net.h -- GENERATED by tools/gen_sort32_net.py; do not edit.  The bitonic sort of 64 x E 32-bit keys of one")
    out.append("// wavefront (E keys per lane, element index i = lane * E + r, ascending on exit) as one asm statement per E: see the")
    out.append("// generator for the network, the scheduling and the hazard rule it keeps.")
    out.append("#pragma once")
    out.append("#include <hip/hip_runtime.h>")
    out.append("#include <stdint.h>")
    out.append("")
    out.append("namespace la {")
    out.append("")
    out.append("// d[b]: all ones on the lanes whose lane-id bit (1 << b) is set, b = 0 .. 5 (loop-invariant; sort_net_dirs fills it)")
    out.append("__device__ __forceinline__ void sort_net_dirs(int lane, uint32_t (&d)[6]) {")
    out.append("#pragma unroll")
    out.append("    for (int b = 0; b < 6; ++b) d[b] = (lane & (1 << b)) ? 0xFFFFFFFFu : 0u;")
    out.append("}")
    out.append("")
    counts = {}
    for E in (1, 2, 4):
        ins = gen(E)
        counts[E] = (len(ins), sum(1 for x in ins if x.startswith("s_nop")))
        out.append("// %d keys: %d issue slots (%d of them s_nop)" % (64 * E, counts[E][0], counts[E][1]))
        out.append("__device__ __forceinline__ void sort_net_u32_e%d(uint32_t (&x)[%d], const uint32_t (&d)[6]) {" % (E, E))
        out.append("    uint32_t " + ", ".join("t%d" % r for r in range(E)) + ", dir, ndir;")
        out.append("    asm volatile(")
        for l in ins:
            out.append('        "%s\\n\\t"' % l)
        outs = ", ".join(['[x%d] "+v"(x[%d])' % (r, r) for r in range(E)] + ['[t%d] "=&v"(t%d)' % (r, r) for r in range(E)] +
                         ['[dir] "=&v"(dir)', '[ndir] "=&v"(ndir)'])
        ins_ = ", ".join('[d%d] "v"(d[%d])' % (1 << b, b) for b in range(6))
        out.append("        : %s" % outs)
        out.append("        : %s);" % ins_)
        out.append("}")
        out.append("")
    out.append("}  // namespace la")
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")
    print(path, counts)


if __name__ == "__main__":
    sys.exit(main())


# ---- a lane-level simulator of exactly the instructions the generator emits (tests/test_kernel_models_cpu.py runs it) --------
def simulate(E, keys, ins=None):
    """keys: list of 64 * E ints (element i = lane * E + r) -> the keys after the generated instruction stream."""
    import re
    ins = ins or gen(E)
    M = 0xFFFFFFFF
    reg = {"%%[x%d]" % r: [keys[l * E + r] & M for l in range(64)] for r in range(E)}
    for b in range(6):
        reg["%%[d%d]" % (1 << b)] = [M if (l >> b) & 1 else 0 for l in range(64)]

    def src_lane(ctrl, l):
        row, i = l & ~15, l & 15
        if ctrl.startswith("quad_perm"):
            p = [int(x) for x in re.findall(r"\d", ctrl)]
            return row | (i & ~3) | p[i & 3]
        if ctrl == "row_ror:8":
            return row | ((i + 8) & 15)
        if ctrl == "row_half_mirror":
            return row | (i & 8) | (7 - (i & 7))
        raise ValueError(ctrl)

    for text in ins:
        op, rest = text.split(" ", 1) if " " in text else (text, "")
        if op == "s_nop":
            continue
        m = re.match(r"(%\[\w+\]), (%\[\w+\])(?:, (%\[\w+\]))?(?:, (%\[\w+\]))?(?: (.*))?$", rest)
        dst, a, b, c, tail = m.groups()
        if op == "v_mov_b32_dpp":
            ctrl = tail.replace(" row_mask:0xf bank_mask:0xf", "")
            s = list(reg[a])
            reg[dst] = [s[src_lane(ctrl, l)] for l in range(64)]
        elif op == "v_mov_b32":
            reg[dst] = list(reg[a])
        elif op == "v_xor_b32":
            reg[dst] = [x ^ y for x, y in zip(reg[a], reg[b])]
        elif op == "v_not_b32":
            reg[dst] = [x ^ M for x in reg[a]]
        elif op == "v_min_u32":
            reg[dst] = [min(x, y) for x, y in zip(reg[a], reg[b])]
        elif op == "v_max_u32":
            reg[dst] = [max(x, y) for x, y in zip(reg[a], reg[b])]
        elif op == "v_med3_u32":
            reg[dst] = [sorted((x, y, z))[1] for x, y, z in zip(reg[a], reg[b], reg[c])]
        elif op in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
            d = 16 if "16" in op else 32
            vd, vs = list(reg[dst]), list(reg[a])
            nd, ns = list(vd), list(vs)
            for l in range(64):
                if l & d:                      # an "odd" row / half of vdst <-> the "even" one of vsrc
                    nd[l] = vs[l ^ d]
                    ns[l ^ d] = vd[l]
            reg[dst], reg[a] = nd, ns
        else:
            raise ValueError(text)
    return [reg["%%[x%d]" % (i % E)][i // E] for i in range(64 * E)]
