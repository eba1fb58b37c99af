"""Executable form of the LDS-DMA ring arguments of the d_head-40 attention kernels (no GPU needed), in the style of
tests/test_pingpong_schedule_model.py:

  * `attn_fwd40_kernel` (csrc/attention_fwd40.hip, run_fast): K / V tiles two ahead into a ring of four stages, ONE barrier per
    tile, `s_waitcnt vmcnt(0)` in front of it; K(t+1) is read in iteration t, the V(t+1) fragment reads are ISSUED in the
    second half of iteration t and retire (`lgkmcnt`) only at the top of iteration t + 1 -- behind that iteration's barrier;
  * the fold backward kernels (csrc/attention_tr.hip, PIPE): Q / dO (or K / V) tiles two ahead into a ring of three stages,
    one barrier per tile, a COUNTED `vmcnt` in front of it (the requests of the tile after this one stay in flight); every
    read of tile t is issued and retired inside iteration t.

All waves of a workgroup run the same sequence, so one event list stands for all of them.  A section = the code between the
k-th and (k+1)-th barrier; what the hardware orders is "X in section i, Y in section j >= i + 1" (and program order within
a wave).  Checked, for every tile count:

  RAW  a tile is read in a section that starts after a barrier which follows every wave's wait for that tile's DMA
       (an LDS-DMA is ordered for a ds_read only by the issuer's vmcnt + a barrier the reader has passed);
  WAR  a DMA into a ring stage is issued after a barrier that follows the RETIREMENT (lgkmcnt) of the last read of the tile
       that occupied the stage;
  the counted wait really covers the tile it is meant for (loads retire in issue order: `vmcnt <= n` leaves exactly the n
  youngest requests in flight).
"""
import pytest


def fwd40_schedule(nt: int, ring: int):
    """(events, sections): events = (section, kind, tile); kinds: issue, wait (vmcnt reaches this tile), read_issue, read_retire."""
    ev, sec = [], 0
    outstanding = []                      # DMA requests in flight, oldest first (per wave)

    def issue(t):
        ev.append((sec, "issue", t)); outstanding.append(t)

    def wait_all():
        while outstanding:
            ev.append((sec, "wait", outstanding.pop(0)))

    issue(0); issue(1)                    # (nt >= 2: launcher)
    wait_all()
    sec += 1                              # __syncthreads
    ev += [(sec, "read_issue", ("K", 0)), (sec, "read_retire", ("K", 0))]
    ev.append((sec, "read_issue", ("V", 0)))                     # V(0) fragments: retired at the top of iteration 0
    for t in range(nt):
        wait_all()                        # s_waitcnt vmcnt(0): tile t + 1 (requested in iteration t - 1)
        sec += 1                          # s_barrier
        if t + 2 < nt:
            issue(t + 2)
        has_next = t + 1 < nt
        if has_next:
            ev.append((sec, "read_issue", ("K", t + 1)))
        ev.append((sec, "read_retire", ("V", t)))                # lgkm_wait at the top: V(t) fragments landed
        if has_next:
            ev.append((sec, "read_retire", ("K", t + 1)))        # lgkm_wait<0> before the S MFMAs
            ev.append((sec, "read_issue", ("V", t + 1)))         # gaps 16..27 of the second half
    return ev, sec


def bwd_schedule(nt: int, ring: int, per_tile: int = 4):
    """The PIPE form of attn_bwd_dkv / attn_bwd_dq: counted vmcnt, three stages."""
    ev, sec = [], 0
    outstanding = []                      # (tile, instruction) requests in flight, oldest first

    def issue(t):
        ev.append((sec, "issue", t))
        outstanding.extend([t] * per_tile)

    def wait_count(n):                    # s_waitcnt vmcnt(n): everything but the n youngest requests has landed
        while len(outstanding) > n:
            t = outstanding.pop(0)
            if t not in outstanding:
                ev.append((sec, "wait", t))

    issue(0)
    if nt > 1:
        issue(1)
    for t in range(nt):
        wait_count(per_tile if t + 1 < nt else 0)
        sec += 1                          # __syncthreads
        if t + 2 < nt:
            issue(t + 2)
        ev += [(sec, "read_issue", ("T", t)), (sec, "read_retire", ("T", t))]   # fragment reads of tile t: all inside iteration t
    return ev, sec


def check(ev, ring):
    waits = {t: sec for sec, k, t in ev if k == "wait"}
    issues = {t: sec for sec, k, t in ev if k == "issue"}
    assert sorted(waits) == sorted(issues), "every requested tile is waited for"
    for t in issues:
        assert issues[t] <= waits[t]
    last_retire = {}
    for sec, k, what in ev:
        if k == "read_issue":
            t = what[1]
            assert t in waits, f"tile {t} read but never requested"
            assert sec >= waits[t] + 1, f"RAW: {what} read in section {sec}, its DMA is waited for in section {waits[t]}"
        if k == "read_retire":
            last_retire[what[1]] = max(last_retire.get(what[1], -1), sec)
    for t, sec_i in issues.items():
        if t >= ring:                     # the stage still holds tile t - ring until its last read has retired
            assert sec_i >= last_retire[t - ring] + 1, \
                f"WAR: tile {t} requested in section {sec_i}, tile {t - ring} last read retires in section {last_retire[t - ring]}"
    return waits, issues


@pytest.mark.parametrize("nt", [2, 3, 4, 5, 8, 64])
def test_fwd40_ring_has_no_raw_or_war_hazard(nt):
    ev, _ = fwd40_schedule(nt, 4)
    check(ev, 4)
    check(ev, 3)          # (three stages would do: the kernel's fourth is slack, not safety)
    tiles_read = sorted({what for _, k, what in ev if k == "read_retire"})
    assert tiles_read == sorted([("K", t) for t in range(nt)] + [("V", t) for t in range(nt)])


@pytest.mark.parametrize("nt", [1, 2, 3, 4, 7, 64])
def test_backward_three_stage_ring_with_counted_vmcnt(nt):
    ev, _ = bwd_schedule(nt, 3)
    waits, _ = check(ev, 3)
    # the counted wait at the top of iteration t is the one that retires tile t (not earlier, not later)
    for t in range(nt):
        assert waits[t] == t, (t, waits[t])


def test_models_catch_a_ring_that_is_too_small():
    with pytest.raises(AssertionError, match="WAR"):
        check(fwd40_schedule(8, 2)[0], 2)
    with pytest.raises(AssertionError, match="WAR"):
        check(bwd_schedule(8, 2)[0], 2)


def test_model_catches_a_missing_wait():
    """A counted wait that leaves one request too many in flight reads a tile that has not landed."""
    ev, sec = [], 0
    ev += [(0, "issue", 0), (0, "issue", 1), (0, "wait", 0)]
    ev += [(1, "read_issue", ("T", 0)), (1, "read_retire", ("T", 0))]
    ev += [(2, "read_issue", ("T", 1)), (2, "read_retire", ("T", 1)), (2, "wait", 1)]     # waited for in the section it is read in
    with pytest.raises(AssertionError, match="RAW"):
        check(ev, 3)
