"""The x-stationary kernel (csrc/gemm_xs.hip) guards its LDS ring with ONE counted `s_waitcnt vmcnt(n)` per chunk that spans
three kinds of vector-memory operations -- the chunk DMAs, the residual loads and the output stores -- relying on gfx9's single,
in-order counter.  `n` is computed from the iteration's position ("allow" in the kernel).  This is a replay of the kernel's issue
order, per wave, for every form that ships (ring depth 2 / 3; plain, residual, GEGLU epilogues) and every column-run length:

  * RAW: when the wave passes the wait of iteration c, every operation up to and including its DMA of chunk c has retired;
  * no over-wait: `n` is exactly the number of operations issued after that DMA (one less would stall on a newer operation);
  * WAR: the DMA that refills a slot is issued after the barrier that follows the last reader of that slot;
  * the residual wait inside an iteration covers the residual loads and nothing newer than them but this iteration's DMA;
  * every block is stored exactly once, its residual loaded exactly once before it.
No GPU."""
import pytest

DPC_OF = {20: 5, 28: 7, 40: 10, 48: 12}          # DMA instructions per wave and chunk (K / 64)


def kernel_allow(c, nch, D, DPC, CPB, RPB, SPB):
    """`allow` as csrc/gemm_xs.hip computes it at the top of iteration c."""
    done_before = lambda i: i >= CPB and i % CPB == 0
    allow = 0
    for j in range(1, D):                      # chunks the prologue issued right behind DMA(c)
        allow += DPC if (c + j < D and c + j < nch) else 0
    for k in range(D, 0, -1):
        i = c - k
        if i < 0:
            continue
        if k != D:
            allow += (RPB if done_before(i) else 0) + (DPC if i + D < nch else 0)
        allow += SPB if done_before(i) else 0
    return allow


def replay(nob, RING, KS, epi):
    CPB = 2 if epi == "geglu" else 1
    RPB = 2 if epi == "res" else 0
    SPB, D, DPC = 2, RING - 1, DPC_OF[KS]
    nch = nob * CPB
    ops = []                       # issue order of one wave: (kind, index)
    dma_pos, res_pos, store_pos = {}, {}, {}
    barrier_after_op = {}          # barrier of iteration c sits after this many issued operations
    waits = []                     # (iteration, allow, ops issued so far)
    inner_waits = []
    for s in range(min(D, nch)):
        dma_pos[s] = len(ops); ops += [("dma", s)] * DPC
        dma_pos[s] = len(ops) - 1                                   # position of the chunk's LAST instruction
    for c in range(nch):
        waits.append((c, kernel_allow(c, nch, D, DPC, CPB, RPB, SPB), len(ops)))
        barrier_after_op[c] = len(ops)
        fin = c >= CPB and c % CPB == 0
        if fin and RPB:
            ops += [("res", c // CPB - 1)] * RPB
            res_pos[c // CPB - 1] = len(ops) - 1
        if c + D < nch:
            ops += [("dma", c + D)] * DPC
            dma_pos[c + D] = len(ops) - 1
        if fin:
            if RPB:
                inner_waits.append((c // CPB - 1, DPC if c + D < nch else 0, len(ops)))
            ops += [("store", c // CPB - 1)] * SPB
            store_pos[c // CPB - 1] = len(ops) - 1
    if RPB:
        ops += [("res", nob - 1)] * RPB
        res_pos[nob - 1] = len(ops) - 1
        inner_waits.append((nob - 1, 0, len(ops)))
    ops += [("store", nob - 1)] * SPB
    store_pos[nob - 1] = len(ops) - 1
    return dict(ops=ops, dma_pos=dma_pos, res_pos=res_pos, store_pos=store_pos, waits=waits, inner_waits=inner_waits,
                barrier_after_op=barrier_after_op, nch=nch, D=D, RING=RING, CPB=CPB)


@pytest.mark.parametrize("RING,KS,epi", [(3, 20, "plain"), (2, 28, "plain"), (3, 40, "plain"), (3, 48, "plain"),
                                          (3, 20, "res"), (2, 28, "res"), (3, 40, "res"),
                                          (3, 20, "geglu"), (2, 28, "geglu"), (3, 48, "geglu")])
def test_counted_vmcnt_of_the_x_stationary_kernel(RING, KS, epi):
    for nob in list(range(1, 14)) + [40, 80 if epi != "geglu" else 40]:
        r = replay(nob, RING, KS, epi)
        ops, nch, D = r["ops"], r["nch"], r["D"]
        for c, allow, issued in r["waits"]:
            # in-order retirement: after vmcnt(allow) the oldest (issued - allow) operations are complete
            retired = issued - allow
            assert r["dma_pos"][c] < retired, (nob, c, "RAW: chunk c not guaranteed to have landed")
            assert r["dma_pos"][c] == retired - 1, (nob, c, "over-wait: the count also covers an operation newer than DMA(c)")
        for b, allow, issued in r["inner_waits"]:
            retired = issued - allow
            assert r["res_pos"][b] == retired - 1, (nob, b, "residual wait")
            assert r["res_pos"][b] < r["store_pos"][b]
        # WAR on the ring: chunk c + D goes into the slot chunk c - 1 was read from (iteration c - 1 ends before barrier c)
        first = {}
        for pos, (kind, idx) in enumerate(ops):
            if kind == "dma":
                first.setdefault(idx, pos)
        for ch, pos in first.items():
            if ch >= D:
                reader = ch - RING                                   # previous occupant of the slot
                assert (ch % RING) == (reader % RING) or reader < 0
                assert pos >= r["barrier_after_op"][ch - D], (nob, ch, "DMA issued before the barrier that frees its slot")
                assert ch - D >= reader + 1                          # ... and that barrier is behind the reader's iteration
        # every block stored once, each residual loaded once and before its store
        stores = [i for k, i in ops if k == "store"]
        assert sorted(set(stores)) == list(range(nob)) and len(stores) == 2 * nob
        if epi == "res":
            loads = [i for k, i in ops if k == "res"]
            assert sorted(set(loads)) == list(range(nob)) and len(loads) == 2 * nob
        assert len([1 for k, _ in ops if k == "dma"]) == nch * DPC_OF[KS]
        assert max(a for _, a, _ in r["waits"]) <= 32                # the kernel's switch covers vmcnt(0) .. vmcnt(32)


def test_a_wrong_count_is_caught_by_the_model():
    """The model has teeth: one store less per block in the formula (SPB = 1) over-waits nowhere but breaks RAW."""
    RING, KS = 3, 20
    D, DPC = RING - 1, DPC_OF[KS]
    r = replay(8, RING, KS, "plain")
    bad = 0
    for c, _, issued in r["waits"]:
        allow = kernel_allow(c, r["nch"], D, DPC, 1, 0, 3)          # the formula believing in THREE stores per block
        if r["dma_pos"][c] >= issued - allow:
            bad += 1
    assert bad > 0
