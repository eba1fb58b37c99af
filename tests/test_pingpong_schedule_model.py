"""Executable form of the hazard argument in `gemm_fl_kernel`'s ping-pong schedule (ctrlora_amd/csrc/gemm.hip,
PRIO == 3): a barrier-generation model of the two staggered wave groups checks, for every pipeline
length, that

  RAW  every LDS fragment read of a stage happens after a barrier that follows the counted-vmcnt wait of EVERY
       wave for that stage's LDS-DMA (an LDS-DMA is ordered for a ds_read only by the issuer's vmcnt + a barrier the
       reader has passed: MI355X_MICROARCH.md / cdna_hip_programming.md "Pipelining across barriers"), and
  WAR  every DMA into a ring slot is issued after a barrier that follows the lgkmcnt(0) retiring the LAST reads of
       the stage that occupied the slot,

with three ring slots.  The model replays the schedule exactly as the kernel writes it (prologue, stagger barrier,
L0 | M0 | L1 | M1 per stage, tail variants without DMA issue, balancing barrier).  A section between a group's
k-th and (k+1)-th barrier can only start after barrier generation k completed (all waves arrived) and ends before
the group arrives at generation k+1 -- so "X in section i of one group, Y in section j >= i+1 of any group" is the
only ordering the hardware guarantees, and that is what the checks use.  No GPU needed.
"""
import pytest

SLOTS = 3


def schedule(group: int, total: int):
    """[(section_index, action, stage)] for one wave group; section index = generation of the last barrier passed."""
    ev, sec = [], 0

    def barrier():
        nonlocal sec
        sec += 1

    # prologue (both groups together): DMA of stage 0, stage 1, A-part of stage 2; wait for stage 0; common barrier
    for s in range(min(total, 2)):
        ev += [(sec, "issueA", s), (sec, "issueB", s)]
    if total > 2:
        ev.append((sec, "issueA", 2))
    ev.append((sec, "wait", 0))
    barrier()
    ev.append((sec, "read", (0, 0)))          # k-half 0 of stage 0; retired by lgkmcnt(0) before the next barrier
    if group == 1:
        barrier()                             # the stagger
    for s in range(total):
        ib, ia = s + 2 < total, s + 3 < total
        # L0
        ev.append((sec, "read", (s, 1)))
        if ib:
            ev.append((sec, "issueB", s + 2))
        if s + 1 < total:
            ev.append((sec, "wait", s + 1))   # vmcnt(G) / vmcnt(0): own DMA of stage s+1 (A and B) has landed
        barrier()                             # M0
        barrier()
        # L1
        if s + 1 < total:
            ev.append((sec, "read", (s + 1, 0)))
        if ia:
            ev.append((sec, "issueA", s + 3))
        barrier()                             # M1
        barrier()
    if group == 0:
        barrier()                             # balance the stagger
    return ev, sec


@pytest.mark.parametrize("total", list(range(1, 14)))
def test_pingpong_ring_has_no_raw_or_war_hazard(total):
    evs, counts = zip(*(schedule(g, total) for g in (0, 1)))
    assert counts[0] == counts[1], "both groups must execute the same number of barriers"
    waits = [{st: sec for sec, a, st in ev if a == "wait"} for ev in evs]
    reads = [[(sec, st) for sec, a, st in ev if a == "read"] for ev in evs]
    issues = [[(sec, a, st) for sec, a, st in ev if a.startswith("issue")] for ev in evs]
    for g in (0, 1):
        # every stage is waited for exactly once and both operand parts were issued before that wait (program order)
        assert sorted(waits[g]) == list(range(total))
        for st in range(total):
            for part in ("issueA", "issueB"):
                sec_issue = [sec for sec, a, s2 in issues[g] if a == part and s2 == st]
                assert len(sec_issue) == 1 and sec_issue[0] <= waits[g][st], (g, st, part)
    for g in (0, 1):
        for sec_r, (st, half) in reads[g]:
            for h in (0, 1):                  # RAW: the read section starts after a barrier following every group's wait
                assert sec_r >= waits[h][st] + 1, f"RAW: group {g} reads stage {st} half {half} in section {sec_r}, " \
                                                   f"group {h} waits in section {waits[h][st]}"
    last_read = {}
    for g in (0, 1):
        for sec_r, (st, _) in reads[g]:
            last_read[st] = max(last_read.get(st, -1), sec_r)
    for g in (0, 1):
        for sec_i, part, st in issues[g]:
            if st >= SLOTS:                   # WAR: the slot still holds stage st - 3 until its last read retired
                assert sec_i >= last_read[st - SLOTS] + 1, f"WAR: group {g} {part}({st}) in section {sec_i}, " \
                                                            f"stage {st - SLOTS} last read in section {last_read[st - SLOTS]}"
    # every stage is read completely (both k-halves) by both groups
    for g in (0, 1):
        assert sorted(st for _, st in reads[g]) == [(s, h) for s in range(total) for h in (0, 1)]


def test_model_catches_a_two_slot_ring():
    """Sanity of the checker itself: with only two slots the same schedule must show a WAR hazard."""
    global SLOTS
    SLOTS = 2
    try:
        with pytest.raises(AssertionError, match="WAR"):
            test_pingpong_ring_has_no_raw_or_war_hazard(8)
    finally:
        SLOTS = 3
