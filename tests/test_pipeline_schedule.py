"""Hazard check of the hand-scheduled LDS-DMA pipelines of the eight-phase GEMM kernels (csrc/pv_gemm9.hip, pv_gemm9h.hip).

The kernels' correctness rests on arithmetic nobody sees in a numerics test until it races: counted `s_waitcnt vmcnt(N)` that
leave the N youngest memory operations of a thread in flight, raw `s_barrier`s, LDS units that are overwritten by a DMA "two
phases after their last read", two halves of the workgroup running one barrier apart, and an epilogue whose stores sit in the
same in-order queue.  This test reads the SCHEDULE out of the kernel sources (prologue issue order, which unit every phase
requests, every wait immediate, which units a phase reads, where the stream advances) and replays it in a small model:

  * per thread, memory operations complete in issue order; after `vmcnt(N)` all but the N youngest are complete;
  * a barrier is a rendezvous of both halves; an event happens-before another one of the other half only across a rendezvous;
  * fragment reads complete at the thread's next wait (the ISA carries `lgkmcnt(0)` in the same instruction:
    tests/test_isa_counts.py);

and asserts, for several output tiles per workgroup, every reduction length class and both store counts:
  RAW  every fragment read of a unit happens after EVERY thread's share of the DMAs that fill it is complete;
  WAR  every DMA into a unit is issued after every thread has finished reading the unit's previous contents;
  and the unit a phase requests is the one the stream's K tile belongs in (parity / buffer index).
(The epilogue-table DMAs are left out: extra operations in the queue only make a counted wait stricter.)
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorchvideo_amd", "csrc")


class Prog:
    """Event list of one half of the workgroup: ("issue", slot, tile) | ("store",) | ("wait", n) | ("barrier",) | ("read", slot, tile)."""

    def __init__(self):
        self.ev = []

    def add(self, *e):
        self.ev.append(e)


def _check(progs):
    info = []
    for p in progs:
        kb, k = [], 0
        for e in p.ev:
            kb.append(k)
            if e[0] == "barrier":
                k += 1
        info.append(kb)
    assert info[0][-1] + (progs[0].ev[-1][0] == "barrier") == info[1][-1] + (progs[1].ev[-1][0] == "barrier"), "barrier counts differ"

    def hb(c1, i1, c2, i2):            # event i1 of half c1 happens-before event i2 of half c2
        return i1 < i2 if c1 == c2 else info[c1][i1] < info[c2][i2]

    # per half: {(slot, tile): index of the first wait that covers the unit's LAST DMA (the thread's whole share has landed)}
    done = []
    for c, p in enumerate(progs):
        last_issue = {}
        for i, e in enumerate(p.ev):
            if e[0] == "issue":
                last_issue[(e[1], e[2])] = i
        vm, comp = [], {}
        pending = dict(last_issue)
        for i, e in enumerate(p.ev):
            if e[0] in ("issue", "store"):
                vm.append(i)
            elif e[0] == "wait":
                cutoff = len(vm) - e[1]
                covered = set(vm[:max(cutoff, 0)])
                for key, li in list(pending.items()):
                    if li in covered:
                        comp[key] = i
                        del pending[key]
        done.append(comp)

    nreads = 0
    for c, p in enumerate(progs):
        nxt_wait = {}
        w = None
        for i in range(len(p.ev) - 1, -1, -1):
            if p.ev[i][0] == "wait":
                w = i
            nxt_wait[i] = w
        for i, e in enumerate(p.ev):
            if e[0] == "read":
                nreads += 1
                key = (e[1], e[2])
                for c2 in range(2):
                    assert key in done[c2], "RAW: unit %s of K tile %d is read but half %d never completes it" % (e[1], e[2], c2)
                    assert hb(c2, done[c2][key], c, i), "RAW: %s/K%d read by half %d before half %d's share has landed" % (e[1], e[2], c, c2)
            if e[0] == "issue":
                # every read of the slot's previous contents, by both halves, is complete before this DMA is issued
                for c2, p2 in enumerate(progs):
                    for i2, e2 in enumerate(p2.ev):
                        if e2[0] == "read" and e2[1] == e[1] and e2[2] < e[2]:
                            fin = next((k for k in range(i2 + 1, len(p2.ev)) if p2.ev[k][0] == "wait"), None)
                            assert fin is not None
                            assert hb(c2, fin, c, i), "WAR: DMA of K%d into %s (half %d) can overtake half %d's read of K%d" % (
                                e[2], e[1], c, c2, e2[2])
    assert nreads > 0


# ------------------------------------------------------------------------------------------------ pv_gemm9.hip (256 x 256 tile)
def _quad_schedule():
    src = open(os.path.join(CSRC, "pv_gemm9.hip")).read()
    pro = src[src.index("// ---- prologue:"):src.index("const bool half_b")]
    prologue = []
    for m in re.finditer(r"issue_a\(\d, unit_a\((\d), (\d)\), -1\)|issue_b\(\d, unit_b\((\d), (\d)\), -1|advance\(\)|vml\((\d+)\)", pro):
        if m.group(0).startswith("issue_a"):
            prologue.append(("A", int(m.group(1)), int(m.group(2))))
        elif m.group(0).startswith("issue_b"):
            prologue.append(("B", int(m.group(3)), int(m.group(4))))
        elif m.group(0).startswith("advance"):
            prologue.append("advance")
        else:
            prologue.append(("wait", int(m.group(5))))
    iss = {}
    for m in re.finditer(r"#define ISS_P(\d)_(\d)\(J\) issue_([ab])\(\d, unit_[ab]\((\d), (\d)\)", src):
        iss[(int(m.group(1)), int(m.group(2)))] = (m.group(3).upper(), int(m.group(4)), int(m.group(5)))
    assert len(iss) == 8
    m = re.search(r"vml\(\(DM \? (\d+) : (\d+)\) \+ \(YF32 \? (\d+) : (\d+)\)\)", src)
    wait_dm, wait_plain, st32, st16 = map(int, m.groups())
    kt = src[src.index("#define PV9_KTILE(P, FIRST)"):src.index("#define ISS_P0_0")]
    body = []
    for m in re.finditer(r"PV9_READ_B\(b\d, P, (\d)\)|PV9_READ_A\(af, P, (\d)\)|PV9_PHASE\(FIRST, ISS_P(\d)_##P|advance\(\)", kt):
        t = m.group(0)
        body.append(("readB", int(m.group(1))) if t.startswith("PV9_READ_B") else ("readA", int(m.group(2))) if t.startswith("PV9_READ_A")
                    else ("phase", int(m.group(3))) if t.startswith("PV9_PHASE") else "advance")
    assert [b for b in body if b != "advance" and b[0] == "phase"] == [("phase", 0), ("phase", 1), ("phase", 2), ("phase", 3)]
    ph = src[src.index("#define PV9_PHASE("):src.index("#define PV9_READ_A")]
    assert ph.index("if (!DM) { ISS(-1); }") < ph.index("PV9_WAIT_B1(FIRST)") < ph.index("PV9_M1(")    # plain: DMAs before the wait
    assert src.index("PV9_KTILE(0, true);") < src.index("PV9_KTILE(1, false);")
    return prologue, iss, (wait_dm, wait_plain), (st16, st32), body


def _quad_program(half_b, nk, tiles, dm, stores):
    prologue, iss, (wait_dm, wait_plain), _, body = _quad_schedule()
    p, stream = Prog(), 0                      # stream: global index of the K tile the DMA stream is on
    for e in prologue:
        if e == "advance":
            stream += 1
        elif e[0] == "wait":
            p.add("wait", e[1])
            p.add("barrier")
        else:
            assert e[1] == stream % 2          # the unit's parity is the K tile's
            p.add("issue", e, stream)
            p.add("issue", e, stream)
    if half_b:
        p.add("barrier")
    g, stores_behind = 0, False
    for t in range(tiles):
        for k in range(nk):
            par, first = k % 2, k == 0
            for b in body:
                if b == "advance":
                    stream += 1
                elif b[0] == "readB":
                    p.add("read", ("B", par, b[1]), g)
                elif b[0] == "readA":
                    p.add("read", ("A", par, b[1]), g)
                else:
                    slot = iss[(b[1], par)]
                    assert slot[1] == stream % 2, "phase %d of parity %d requests a unit of the wrong parity" % (b[1], par)
                    extra = stores if (first and stores_behind) else 0
                    if not dm:
                        p.add("issue", slot, stream)
                        p.add("issue", slot, stream)
                        p.add("wait", wait_plain + extra)
                        p.add("barrier")
                    else:
                        p.add("wait", wait_dm + extra)
                        p.add("barrier")
                        p.add("issue", slot, stream)
                        p.add("issue", slot, stream)
                    p.add("barrier")
            g += 1
        if not half_b:
            p.add("barrier")
        for _ in range(stores):
            p.add("store")
        stores_behind = True
        if half_b:
            p.add("barrier")
    if not half_b:
        p.add("barrier")
    p.add("wait", 0)
    return p


@pytest.mark.parametrize("dm", [False, True])
@pytest.mark.parametrize("stores", [16, 32])
@pytest.mark.parametrize("nk", [4, 6, 12])
def test_quad_kernel_schedule_has_no_lds_hazard(nk, stores, dm):
    _, _, _, (st16, st32), _ = _quad_schedule()
    assert (st16, st32) == (16, 32)
    _check([_quad_program(False, nk, 3, dm, stores), _quad_program(True, nk, 3, dm, stores)])


def test_the_model_sees_a_wait_that_is_one_too_weak():
    """The checker is not vacuous: the same schedule with every main-loop wait relaxed by one DMA pair races."""
    import unittest.mock as mock
    prologue, iss, waits, st, body = _quad_schedule()
    with mock.patch(__name__ + "._quad_schedule", lambda: (prologue, iss, (waits[0] + 2, waits[1] + 2), st, body)):
        with pytest.raises(AssertionError, match="RAW"):
            _check([_quad_program(False, 6, 2, True, 16), _quad_program(True, 6, 2, True, 16)])


# ------------------------------------------------------------------------------------------- pv_gemm9h.hip (128 x 256 / 256 x 128)
def _half_schedule():
    src = open(os.path.join(CSRC, "pv_gemm9h.hip")).read()
    pro = src[src.index("// ---- prologue:"):src.index("const bool half_b")]
    prologue = []
    for m in re.finditer(r"issue_s\((\d), (\d), (\d), gq\)|issue_w\((\d), (\d), gq\)|advance\(\)|vml\((\d+)\)", pro):
        t = m.group(0)
        if t.startswith("issue_s"):
            prologue.append(("S%d" % int(m.group(1)), int(m.group(2))))
        elif t.startswith("issue_w"):
            prologue.append(("W", int(m.group(4))))
        elif t.startswith("advance"):
            prologue.append("advance")
        else:
            prologue.append(("wait", int(m.group(6))))
    kt = src[src.index("#define PVH_KTILE(Q, F0, F1)"):src.index("const int nk3")]
    assert "constexpr int QW = ((Q) + 2) % 3;" in kt
    body = []
    for m in re.finditer(r"advance\(\)|PVH_READ_B\(Q\)|PVH_READ_A\(Q, (\d)\)|PVH_PHASE\(F(\d), (\d+), (\d), ([^;]*)\);", kt):
        t = m.group(0)
        if t.startswith("advance"):
            body.append("advance")
        elif t.startswith("PVH_READ_B"):
            body.append(("read", "W"))
        elif t.startswith("PVH_READ_A"):
            body.append(("read", "S%d" % int(m.group(1))))
        else:
            units = []
            for u in re.finditer(r"issue_s\((\d), QW, \d, gq\)|issue_w\(QW, \d, gq\)", m.group(5)):
                units.append("S%d" % int(u.group(1)) if u.group(0).startswith("issue_s") else "W")
            assert m.group(2) == m.group(4)                     # F0 flags phase 0, F1 phase 1
            body.append(("phase", int(m.group(3)), int(m.group(4)), units))
    flags = re.findall(r"PVH_KTILE\((\d), (true|false), (true|false)\);", src)
    flags = [(int(q), a == "true", b == "true") for q, a, b in flags]
    first = flags[:3]                                  # the three K tiles after an epilogue; the loop body carries no allowance
    assert [f[0] for f in first] == [0, 1, 2] and flags[3:6] == [(0, False, False), (1, False, False), (2, False, False)]
    m = re.search(r"constexpr int kStores = YF32 \? (\d+) : (\d+);", src)
    ph = src[src.index("#define PVH_PHASE("):src.index("#define PVH_READ_A")]
    assert ph.index("PVH_WAIT_B1(FIRST, N)") < ph.index("PVH_SLOT(D0)") < ph.index("PVH_SLOT(D2)")     # DMAs inside the MFMA window
    return prologue, body, first, (int(m.group(2)), int(m.group(1)))


def _half_program(half_b, nk, tiles, stores):
    prologue, body, first, _ = _half_schedule()
    p, stream = Prog(), 0
    for e in prologue:
        if e == "advance":
            stream += 1
        elif e[0] == "wait":
            p.add("wait", e[1])
            p.add("barrier")
        else:
            assert e[1] == stream % 3                  # buffer index = K tile index mod 3
            p.add("issue", (e[0], e[1]), stream)
    if half_b:
        p.add("barrier")
    g, stores_behind = 0, False
    for t in range(tiles):
        for k in range(nk):
            q = k % 3
            fl = {0: first[0][1], 1: first[0][2]} if k == 0 else {0: first[1][1], 1: first[1][2]} if k == 1 else {0: False, 1: False}
            for b in body:
                if b == "advance":
                    stream += 1
                elif b[0] == "read":
                    p.add("read", (b[1], q), g)
                else:
                    _, n, phase, units = b
                    assert (q + 2) % 3 == stream % 3, "the stream's K tile does not belong in buffer (Q + 2) % 3"
                    p.add("wait", n + (stores if (fl[phase] and stores_behind) else 0))
                    p.add("barrier")
                    for u in units:
                        p.add("issue", (u, (q + 2) % 3), stream)
                    p.add("barrier")
            g += 1
        if not half_b:
            p.add("barrier")
        for _ in range(stores):
            p.add("store")
        stores_behind = True
        if half_b:
            p.add("barrier")
    if not half_b:
        p.add("barrier")
    p.add("wait", 0)
    return p


@pytest.mark.parametrize("stores", [8, 16])
@pytest.mark.parametrize("nk", [6, 9, 18])
def test_half_height_kernel_schedule_has_no_lds_hazard(nk, stores):
    _, _, _, st = _half_schedule()
    assert st == (8, 16)
    _check([_half_program(False, nk, 3, stores), _half_program(True, nk, 3, stores)])


def test_half_height_first_flags_are_exactly_the_phases_with_stores_behind():
    """Dropping the store allowance from any of the three flagged phases is caught; adding it to the fourth would over-wait
    (correct but slower) -- the flags in the source are the minimal set."""
    import unittest.mock as mock
    prologue, body, first, st = _half_schedule()
    assert first == [(0, True, True), (1, True, False), (2, False, False)]
    # allowance on a phase whose covered units were requested AFTER the stores: the wait would leave the unit in flight
    bad = [(0, True, True), (1, True, True), (2, False, False)]
    with mock.patch(__name__ + "._half_schedule", lambda: (prologue, body, bad, st)):
        with pytest.raises(AssertionError, match="RAW"):
            _check([_half_program(False, 6, 2, 8), _half_program(True, 6, 2, 8)])
