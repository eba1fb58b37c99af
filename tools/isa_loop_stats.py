#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S listing (no GPU needed).
usage: tools/isa_loop_stats.py file.s <substring of the mangled kernel name> [--dump]
For every backward branch (a loop) prints the instruction histogram of the code between the label and the branch:
MFMA, VALU by class, LDS, VMEM, s_waitcnt, s_nop (with total nop states), s_barrier, v_mov / v_accvgpr moves, scratch."""
import re, sys, collections

def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel") or lines[i].strip() == "s_endpgm" and i > start + 50)
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", l)
        if m: labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", l) or re.match(r"\s+s_branch\s+(\.LBB\w+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    print(f"kernel at line {start + 1}, {len(body)} lines, loops: {[(a + start + 1, b + start + 1) for a, b in loops]}")
    for a, b in loops:
        h = collections.Counter(); nops = 0
        for l in body[a:b + 1]:
            l = l.strip()
            if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"): continue
            op = l.split()[0]
            if op == "s_nop":
                nops += int(l.split()[1]) + 1
            if op.startswith("v_mfma"): c = op
            elif op.startswith("ds_"): c = op
            elif op.startswith(("buffer_", "global_", "flat_")): c = op
            elif op.startswith("scratch_"): c = op
            elif op.startswith("v_"): c = op
            elif op.startswith("s_waitcnt"): c = "s_waitcnt"
            elif op in ("s_nop", "s_barrier", "s_setprio"): c = op
            elif op.startswith("s_cbranch") or op == "s_branch": c = "s_branch*"
            else: c = "salu"
            h[c] += 1
        tot = sum(h.values())
        valu = sum(v for k, v in h.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        mfma = sum(v for k, v in h.items() if k.startswith("v_mfma"))
        print(f"--- loop lines {a + start + 1}-{b + start + 1}: {tot} instr, {mfma} MFMA, {valu} VALU, nop states {nops}")
        for k, v in sorted(h.items(), key=lambda kv: -kv[1]):
            print(f"   {v:5d}  {k}")
        if dump:
            print("\n".join(body[a:b + 1]))

if __name__ == "__main__":
    main()
