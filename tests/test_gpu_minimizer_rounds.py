"""Round 5: the placing protocol of the one-pass minimizer kernels (nthash_amd/csrc/block_rounds.hpp) when its assumption --
every block of the grid resident -- does NOT hold.  The kernels go out as cooperative launches now; what is left is a leader
that waits too long, sets `abort`, and the host redoing the call on the kernels that need no protocol.  Nothing exercised
that path before: here a knob oversizes the grid on a plain launch (NTHIP_TUNE_MZ_GRID) and another
shortens the wait (NTHIP_TUNE_MZ_TIMEOUT_US), and two contexts run minimizers on one device at the same time."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx_with(env):
    import nthash_amd
    for k_, v in env.items():
        os.environ[k_] = str(v)
    try:
        return nthash_amd.Context(0)
    finally:
        for k_ in env:
            os.environ.pop(k_, None)


def _brute(oracle, data, n, L, k, w):
    """(offsets, positions, hashes) of the minimizers of n fixed-length reads, from the oracle's stream"""
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data[: n * L], offs, k, 1, want_pos=True)
    nwin = L - k + 1
    ww = min(w, nwin)
    exp_off, exp_pos, exp_h = [0], [], []
    o = 0
    for r in range(n):
        c = int(want["counts"][r])
        p, h = want["pos"][o:o + c].astype(np.int64), want["hashes"][o:o + c].ravel()
        picked = set()
        for s in range(0, nwin - ww + 1):
            lo, hi = np.searchsorted(p, s), np.searchsorted(p, s + ww)
            if hi > lo:
                picked.add(int(p[lo + int(np.argmin(h[lo:hi]))]))
        picked = sorted(picked)
        look = dict(zip(p.tolist(), h.tolist()))
        exp_pos += picked
        exp_h += [look[q] for q in picked]
        exp_off.append(len(exp_pos))
        o += c
    return np.array(exp_off, np.uint64), np.array(exp_pos, np.uint32), np.array(exp_h, np.uint64)


def _device_minimizers(ctx, d_in, n, L, k, w, cap):
    d_h, d_p, d_o = ctx.malloc(cap * 8), ctx.malloc(cap * 4), ctx.malloc((n + 1) * 8)
    ctx.set_profiling(True)
    total = ctx.minimizers_ptr(d_in, n, L, 0, k, w, d_h, d_p, d_o, cap)
    name = ctx.last_kernel_ms()[1]
    ctx.set_profiling(False)
    h, p, o = np.zeros(total, np.uint64), np.zeros(total, np.uint32), np.zeros(n + 1, np.uint64)
    ctx.d2h(h, d_h)
    ctx.d2h(p, d_p)
    ctx.d2h(o, d_o)
    for q in (d_h, d_p, d_o):
        ctx.free(q)
    return total, o, p, h, name


@pytest.mark.parametrize("L,k,w,dirty", [(150, 31, 10, False), (150, 31, 10, True), (150, 31, 19, False)])   # the record form; with N's; the any-run-length form
def test_minimizers_when_the_grid_is_not_resident(oracle, L, k, w, dirty):
    """a plain launch of 8 x the blocks the device holds, leaders that give up after 300 us: the look-back of a resident
    block waits for a block that is not running, `abort` is set, the kernel's output is garbage (inside the arrays) and the
    host redoes the call on the kernels without a protocol -- the minimizers are those of the default context, and those of
    the brute force over the oracle's stream"""
    n = 600_000
    good = _ctx_with({})
    bad = _ctx_with({"NTHIP_TUNE_MZ_GRID": 2048, "NTHIP_TUNE_MZ_TIMEOUT_US": 300})
    d_in = good.malloc(n * L)
    good.synth_reads_ptr(d_in, 0, n, L, 77)
    if dirty:
        rng = np.random.default_rng(5)
        enn = np.array([78], np.uint8)
        for p_ in rng.choice(n * L, 300, replace=False):
            good.h2d(d_in + int(p_), enn)
    cap = n * (2 * (L - k + 1) // (w + 1) + 8)
    t0, o0, p0, h0, name0 = _device_minimizers(good, d_in, n, L, k, w, cap)
    assert name0 in ("minimizer_w_kernel", "minimizer_fused_kernel"), name0
    t1, o1, p1, h1, name1 = _device_minimizers(bad, d_in, n, L, k, w, cap)
    assert name1 not in ("minimizer_w_kernel", "minimizer_fused_kernel"), name1     # the call was redone
    assert t1 == t0 and (o1 == o0).all() and (p1 == p0).all() and (h1 == h0).all()
    m = 1500
    head = np.zeros(m * L, np.uint8)
    good.d2h(head, d_in)
    eo, ep, eh = _brute(oracle, head, m, L, k, w)
    assert (o1[: m + 1] == eo).all() and (p1[: int(eo[-1])] == ep).all() and (h1[: int(eo[-1])] == eh).all()
    # and the oversized grid WITH time to wait: the blocks take their turns on the CUs a resident block's look-back needs, so
    # whether this finishes inside the wait depends on the scheduler -- either way the answer is the same
    slow = _ctx_with({"NTHIP_TUNE_MZ_GRID": 2048, "NTHIP_TUNE_MZ_TIMEOUT_US": 20000})
    t2, o2, p2, h2, _ = _device_minimizers(slow, d_in, n, L, k, w, cap)
    assert t2 == t0 and (o2 == o0).all() and (p2 == p0).all() and (h2 == h0).all()
    good.free(d_in)
    for c in (good, bad, slow):
        c.close()


def test_minimizers_from_two_contexts_at_once(oracle):
    """two contexts (two streams) run the one-pass kernel on the same device at the same time, five times over: a cooperative
    launch has its blocks resident or is refused, a refused or late launch is redone -- both always answer like one alone"""
    n, L, k, w = 2_000_000, 150, 31, 10
    a, b = _ctx_with({}), _ctx_with({})
    d_a, d_b = a.malloc(n * L), b.malloc(n * L)
    a.synth_reads_ptr(d_a, 0, n, L, 11)
    b.synth_reads_ptr(d_b, n, n, L, 11)
    cap = n * (2 * (L - k + 1) // (w + 1) + 8)
    ref_a = _device_minimizers(a, d_a, n, L, k, w, cap)
    ref_b = _device_minimizers(b, d_b, n, L, k, w, cap)
    out, err = {}, []

    def run(key, ctx, d_in, reps):
        try:
            out[key] = [_device_minimizers(ctx, d_in, n, L, k, w, cap) for _ in range(reps)]
        except Exception as e:  # noqa: BLE001
            err.append(e)
    ta = threading.Thread(target=run, args=("a", a, d_a, 5))
    tb = threading.Thread(target=run, args=("b", b, d_b, 5))
    ta.start()
    tb.start()
    ta.join()
    tb.join()
    assert not err, err
    for key, ref in (("a", ref_a), ("b", ref_b)):
        for got in out[key]:
            assert got[0] == ref[0] and (got[1] == ref[1]).all() and (got[2] == ref[2]).all() and (got[3] == ref[3]).all()
    m = 800
    head = np.zeros(m * L, np.uint8)
    b.d2h(head, d_b)
    eo, ep, eh = _brute(oracle, head, m, L, k, w)
    assert (ref_b[1][: m + 1] == eo).all() and (ref_b[2][: int(eo[-1])] == ep).all() and (ref_b[3][: int(eo[-1])] == eh).all()
    a.free(d_a)
    b.free(d_b)
    a.close()
    b.close()
