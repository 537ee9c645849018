"""Ring protocol of the experimental LDS-DMA chain kernel (k_dense_dma, csrc/dense.hip), modelled for one wave (all waves issue the same sequence):
memory operations complete in issue order, and `s_waitcnt vmcnt(N)` guarantees that everything but the N youngest has completed.  The model
follows the kernel's control flow -- prologue order, one counted wait + one barrier per k-step, the waits skipped for WU - 1 k-steps behind an
epilogue (which starts with vmcnt(0)), the epilogue's stores and the 4-byte shift DMAs in between -- and checks that (1) a ring unit is complete
when its fragments are read, (2) a slot is overwritten only after its previous unit was used.  WU / RU / pieces per wave as in the kernel."""
import pytest


def run(NTM, NTN, ks, n_pass_tiles, stores_per_epi, shift_dma):
    FP, RT = 128*NTM, 64*NTN
    WU, RU = 4, (5 if NTN == 4 else 6)
    kWP, kRP = FP//128, RT//128
    kWait = (WU-2)*(kWP+kRP) + kRP
    ops = []            # (kind, unit) in issue order
    done = 0            # ops[:done] are known complete
    w_next = r_next = 0
    slot_w = {}; slot_r = {}     # slot -> unit currently (being) loaded
    used_w = set(); used_r = set()
    def issue_w():
        nonlocal w_next
        s = w_next % WU
        if s in slot_w: assert slot_w[s] in used_w, ("w slot overwritten before use", slot_w[s])
        slot_w[s] = w_next
        for _ in range(kWP): ops.append(('w', w_next))
        w_next += 1
    def issue_r():
        nonlocal r_next
        s = r_next % RU
        if s in slot_r: assert slot_r[s] in used_r, ("r slot overwritten before use", slot_r[s])
        slot_r[s] = r_next
        for _ in range(kRP): ops.append(('r', r_next))
        r_next += 1
    def wait(n):
        nonlocal done
        done = max(done, len(ops) - n)
    def complete(kind, unit):
        idx = [i for i, o in enumerate(ops) if o == (kind, unit)]
        return len(idx) > 0 and max(idx) < done
    if shift_dma: ops.extend([('s', -1)] * shift_dma)
    for u in range(RU - WU): issue_r()
    for u in range(WU - 1): issue_w(); issue_r()
    since = WU - 1
    j = 0
    for p in range(n_pass_tiles):
        for step in range(2 * ks):
            if since >= WU - 1: wait(kWait)
            else: since += 1
            # barrier B(j): unit j-1 has been used by everyone
            issue_w(); issue_r()
            assert complete('w', j), ("weights unit not complete at use", p, step, j)
            assert complete('r', j), ("ray unit not complete at use", p, step, j)
            used_w.add(j); used_r.add(j)
            j += 1
        wait(0); since = 0                                # epilogue: vmcnt(0) + barrier
        ops.extend([('s', -1)] * shift_dma)               # next tile's shifts
        ops.extend([('st', -1)] * stores_per_epi)         # stores
    return len(ops)


@pytest.mark.parametrize("ntm,ntn,ks", [(2, 4, 5), (2, 4, 16), (2, 4, 21), (3, 2, 16), (3, 2, 12)])
def test_counted_waits_cover_every_unit_and_no_slot_is_overwritten_early(ntm, ntn, ks):
    for stores in (0, 5, 32, 40):              # epilogue stores per wave (ragged tiles issue fewer)
        for shift_dma in (0, 1, 3):            # shift-table pieces per wave
            assert run(ntm, ntn, ks, 4, stores, shift_dma) > 0


def test_the_model_notices_a_wait_that_is_one_piece_short():
    # the same protocol with kWait + 1 must fail: the weights of the unit in use would be among the ops not waited for
    code = compile(open(__file__).read().replace("kWait = (WU-2)*(kWP+kRP) + kRP", "kWait = (WU-2)*(kWP+kRP) + kRP + 1"), __file__ + "<+1>", "exec")
    ns = {"__name__": "model_plus_one", "__file__": __file__}
    exec(code, ns)
    with pytest.raises(AssertionError):
        ns["run"](2, 4, 16, 2, 0, 0)
