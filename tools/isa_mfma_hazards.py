#!/usr/bin/env python3
"""Static check of a `hipcc -S` listing for the hazard the compiler cannot see behind inline-asm MFMAs: a non-MFMA instruction that
reads (or overwrites) a register an earlier v_mfma writes, before the MFMA's result exists.

    usage: tools/isa_mfma_hazards.py file.s [<substring of the mangled kernel name>]     (exit code 1 if anything is flagged)

Rule (gfx940 / gfx950, LLVM GCNHazardRecognizer: "XDL write VGPR -> VALU / VMEM / LDS / FLAT read, VALU write"): after an MFMA
of P passes, P + 3 wait states must lie between it and such an instruction.  For builtin MFMAs the compiler inserts the
s_nops itself; for MFMAs issued from inline asm nothing does -- and a hand-placed `s_nop` only protects a reader the compiler
cannot schedule in front of it (round 4: a drain without a data dependence on the accumulators).

Model: time in issue slots (4 clocks).  Every instruction takes one slot, `s_nop k` takes k + 1; an MFMA cannot ISSUE before
the matrix pipe has finished accepting the previous one (P slots after that one's issue), which is what makes "two unrelated
MFMAs in between" a sufficient distance.  P = 8 for the 32x32 shapes, 4 for 16x16 (measured issue intervals).
The scan is linear over the kernel's listing (fall-through paths only: the state is dropped behind an unconditional branch, a
hazard that only exists across a taken branch or a loop back-edge is not seen; waits and barriers are counted as one slot, i.e.
never in the kernel's favour)."""
import re
import sys

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
LOAD_LIKE = ("ds_read", "ds_load", "global_load", "buffer_load", "scratch_load", "flat_load")
STORE_LIKE = ("global_store", "buffer_store", "scratch_store", "flat_store", "ds_write", "ds_store", "global_atomic", "ds_add")


def regs_of(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def passes(op):
    if "32x32" in op:
        return 8
    if "16x16" in op:
        return 4
    return 2 if "4x4" in op else 8


def kernels(lines):
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            start = (m.group(1), i)
        elif start and l.strip().startswith("s_endpgm"):
            yield start[0], start[1], i
            start = None


def scan(lines, lo, hi):
    t, pipe_free = 0, 0
    inflight = []        # (issue slot, P, dest registers, line no, text)
    found = []
    for ln in range(lo, hi + 1):
        s = lines[ln].split(";")[0].strip()
        if not s or s.endswith(":") or s.startswith((".", "#", "//")):
            continue
        op, _, rest = s.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if op == "s_nop":
            t += int(ops[0], 0) + 1
            continue
        if op in ("s_branch", "s_setpc_b64", "s_endpgm"):      # what follows is reached by jumps only: nothing known in flight
            inflight = []
            t += 1
            continue
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            t = max(t, pipe_free)
            p = passes(op)
            pipe_free = t + p
            inflight.append((t, p, regs_of(ops[0]), ln + 1, s))
            inflight = [f for f in inflight if t - f[0] < 64]
            t += 1
            continue
        # sources and destinations alike (RAW and WAW) -- except the destination of a load: its data arrives hundreds of clocks
        # later, long after any MFMA in flight has written the register (the allocator re-uses dead accumulators this way)
        is_load = op.startswith(LOAD_LIKE) and not op.startswith(STORE_LIKE)
        touched = set()
        for o in (ops[1:] if is_load else ops):
            touched |= regs_of(o)
        if touched:
            for (ti, p, dst, mln, mtext) in inflight:
                if touched & dst and (t - ti - 1) < p + 3:
                    found.append((ln + 1, s, mln, mtext, t - ti - 1, p + 3))
        t += 1
    return found


def main():
    path = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().split("\n")
    bad = 0
    for name, lo, hi in kernels(lines):
        if key not in name:
            continue
        found = scan(lines, lo, hi)
        print(f"{name}: {len(found)} flagged")
        for ln, s, mln, mtext, have, need in found[:20]:
            print(f"  line {ln}: `{s}`  touches the result of line {mln} `{mtext[:60]}`: {have} wait states, {need} needed")
        bad += len(found)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
