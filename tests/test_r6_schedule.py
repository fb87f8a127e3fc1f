"""The resident-plane walk (R6) of csrc/gemm_bf16_256.hip has a 12-phase LDS-DMA schedule with counted waits; this test
reads the schedule OUT OF THE KERNEL SOURCE -- the product order, the slot tables, what every phase issues, the vmcnt
immediates, the prologue -- and replays it: every fragment read must find the plane / half / K-tile it expects in its slot
(content), every half image must be guaranteed landed by a wait at least one phase before its first read (RAW, with the
in-order vmcnt semantics: a wait for N leaves the N newest pieces in flight, two pieces per half image), and every
overwrite must be issued at least one phase after the last read of what it replaces (WAR; reads retire before a phase's
first barrier and the lagging row group is one barrier behind -- the rule of the round-1 schedule).  A pitfall the GPU
tests cannot be trusted to catch: an early read passes whenever the DMA happens to land first."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "collaborative-deep-metric-learning_amd", "csrc", "gemm_bf16_256.hip")).read()


def _ints(name, text):
    m = re.search(name + r"\s*=\s*\{([^;]*?)\}[,;]", text)
    assert m, name
    return [int(x) for x in re.findall(r"-?\d+", m.group(1))]


def _schedule():
    i = SRC.index("auto phase = [&](auto phc, auto parc")
    body = SRC[i:SRC.index("auto period = [&]", i)]
    pa, pb = _ints(r"constexpr int PA\[6\]", body)[:6], _ints(r"PB\[6\]", body)[:6]
    sa = [int(x) for x in re.findall(r"\d+", re.search(r"SLOT_A\[2\]\[3\]\s*=\s*\{(.*?)\}\};", body).group(1))]
    slot_a = [sa[:3], sa[3:6]]
    assert len(sa) == 6
    vm = _ints(r"constexpr int VM\[12\]", body)
    assert len(vm) == 12
    issue = {}
    for m in re.finditer(r"if constexpr \(PH == (\d+)\) issue\((\d), (\d), (\d), ([^,]+), (kw[ab]_[cn])\);", body):
        issue[int(m.group(1))] = (int(m.group(2)), int(m.group(3)), int(m.group(4)), m.group(5).strip(), m.group(6)[-1] == "n")
    assert sorted(issue) == list(range(12))
    sb = re.search(r"constexpr int sb = (.+?);", body).group(1)          # B slot of the plane read at this phase
    rdb = re.search(r"constexpr bool rdB = HALF == 0 && \((.+?)\);", body).group(1)
    rdb_steps = [int(x) for x in re.findall(r"S == (\d)", rdb)]
    j = SRC.index("// prologue: the steady state at phase 0 of the first K-tile")
    pro = [(int(a), int(b), int(c), int(d)) for a, b, c, d in
           re.findall(r"issue\((\d), (\d), (\d), (\d), k[ab]_?\);", SRC[j:SRC.index("CDML_BARRIER();", j)])]
    pro_vm = int(re.search(r"vmcnt\((\d+)\)", SRC[j:SRC.index("CDML_BARRIER();", j)]).group(1))
    return pa, pb, slot_a, vm, issue, sb, rdb_steps, pro, pro_vm


def test_r6_schedule_hazards():
    PA, PB, SLOT_A, VM, ISSUE, SB, RDB, PRO, PRO_VM = _schedule()
    assert sorted(zip(PA, PB)) == sorted([(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]), "the six plane products"
    n_per = 7
    seq = []                                                        # (time, (operand, slot, half), (operand, plane, half, K-tile))
    for n, (img, pl, hh, slot) in enumerate(PRO):
        seq.append((n - len(PRO), (img, slot, hh), (img, pl, hh, 0)))
    for w in range(n_per):
        PAR = w & 1
        for ph in range(12):
            img, pl, hh, slot_expr, nxt = ISSUE[ph]
            slot = eval(slot_expr, {"SLOT_A": SLOT_A, "PAR": PAR})
            seq.append((w * 12 + ph, (img, slot, hh), (img, pl, hh, w + (1 if nxt else 0))))
    reads = []
    for w in range(n_per):
        PAR = w & 1
        for ph in range(12):
            S, HALF = ph >> 1, ph & 1
            reads.append((w * 12 + ph, (0, SLOT_A[PAR][PA[S]], HALF), (0, PA[S], HALF, w)))
            if HALF == 0 and S in RDB:
                slot = eval(SB, {"PB": PB, "S": S, "PAR": PAR})
                reads += [(w * 12 + ph, (1, slot, hh), (1, PB[S], hh, w)) for hh in (0, 1)]
    assert sum(1 for r in reads if r[1][0] == 1) == n_per * 6, "B fragments are read on three of the six steps"

    def landed_by(idx):                                             # the earliest wait after which load #idx is guaranteed done
        for T in range(-1, n_per * 12):
            issued = sum(1 for tt, _, _ in seq if tt <= T)
            in_flight = (PRO_VM if T == -1 else VM[T % 12]) // 2    # half images allowed to stay in flight
            if idx < issued - in_flight:
                return T
        return None

    for T, key, want in reads:
        last = [(i, tt, tag) for i, (tt, k, tag) in enumerate(seq) if k == key and tt < T][-1]
        assert last[2] == want, "phase %d reads %s: the slot holds %s, not %s" % (T, key, last[2], want)
        lt = landed_by(last[0])
        assert lt is not None and lt <= T - 1, "RAW: %s read at phase %d, guaranteed landed only by the wait of phase %s" % (want, T, lt)
    for tt, key, tag in seq:
        assert not [T for T, k, _ in reads if k == key and T == tt], "WAR: %s overwritten in the phase that still reads it (%d)" % (key, tt)
    # every phase issues exactly one half image, 6 plane images per K-tile
    assert len(seq) == len(PRO) + 12 * n_per


def _narrow_schedule():
    i = SRC.index("auto phase_n = [&](auto sc, auto parc")
    body = SRC[i:SRC.index("auto period_n = [&]", i)]
    pa, pb = _ints(r"constexpr int PA\[6\]", body)[:6], _ints(r"PB\[6\]", body)[:6]
    sa = [int(x) for x in re.findall(r"\d+", re.search(r"NSLOT_A\[2\]\[3\]\s*=\s*\{(.*?)\}\};", body).group(1))]
    assert len(sa) == 6
    slot_a = [sa[:3], sa[3:6]]
    vm = _ints(r"constexpr int NVM\[6\]", body)
    assert len(vm) == 6
    rdb_steps = [int(x) for x in re.findall(r"S == (\d)", re.search(r"constexpr bool rdB = (.+?);", body).group(1))]
    issue = {}
    for m in re.finditer(r"if constexpr \(S == (\d)\) \{(.*?)\}\n", body):
        issue[int(m.group(1))] = [(int(a), int(b), int(c), d.strip(), e[-1] == "n") for a, b, c, d, e in
                                  re.findall(r"issue_n\((\d), (\d), (\d), ([^,]+), (kw[ab]_[cn])\);", m.group(2))]
    assert sorted(issue) == list(range(6))
    j = SRC.index("// narrow prologue: the steady state at step 0 of the first K-tile")
    seg = SRC[j:SRC.index("CDML_BARRIER();", j)]
    pro = [(int(a), int(b), int(c), int(d)) for a, b, c, d in re.findall(r"issue_n\((\d), (\d), (\d), (\d), k[ab]_?\);", seg)]
    pro_vm = int(re.search(r"vmcnt\((\d+)\)", seg).group(1))
    return pa, pb, slot_a, vm, issue, rdb_steps, pro, pro_vm


def test_half_tile_schedule_hazards():
    """The 128 x 256 half tile of the same walk (one phase per step, A rows in ONE half image per plane: four 16-KiB A slots,
    three B slots): content, RAW and WAR replayed from the source like the full tile's schedule."""
    PA, PB, SLOT_A, VM, ISSUE, RDB, PRO, PRO_VM = _narrow_schedule()
    assert sorted(zip(PA, PB)) == sorted([(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]), "the six plane products"
    assert RDB == [0, 3, 5]
    n_per = 7
    # a slot: A = (0, slot) one half image (hh always 0); B = (1, slot, hh).  LDS: A slots 0..3 at slot * IMG, B slots at
    # 4 IMG + slot * 2 IMG + hh * IMG -- disjoint by construction for slots 0..3 / 0..2
    seq = []                                                        # (time, key, content)
    for n, (img, pl, hh, slot) in enumerate(PRO):
        assert img == 1 or hh == 0
        seq.append((n - len(PRO), (img, slot, hh), (img, pl, hh, 0)))
    for w in range(n_per):
        PAR = w & 1
        for S in range(6):
            for img, pl, hh, slot_expr, nxt in ISSUE[S]:
                slot = eval(slot_expr, {"NSLOT_A": SLOT_A, "PAR": PAR})
                assert (img == 0 and hh == 0 and 0 <= slot <= 3) or (img == 1 and 0 <= slot <= 2)
                seq.append((w * 6 + S, (img, slot, hh), (img, pl, hh, w + (1 if nxt else 0))))
    assert len(seq) == len(PRO) + 9 * n_per, "nine half images per K-tile: three A, six B"
    reads = []
    for w in range(n_per):
        PAR = w & 1
        for S in range(6):
            reads.append((w * 6 + S, (0, SLOT_A[PAR][PA[S]], 0), (0, PA[S], 0, w)))
            if S in RDB:
                reads += [(w * 6 + S, (1, PB[S], hh), (1, PB[S], hh, w)) for hh in (0, 1)]   # plane p lives in B slot p

    def landed_by(idx):
        for T in range(-1, n_per * 6):
            issued = sum(1 for tt, _, _ in seq if tt <= T)
            in_flight = (PRO_VM if T == -1 else VM[T % 6]) // 2
            if idx < issued - in_flight:
                return T
        return None

    for T, key, want in reads:
        last = [(i, tt, tag) for i, (tt, k, tag) in enumerate(seq) if k == key and tt < T][-1]
        assert last[2] == want, "step %d reads %s: the slot holds %s, not %s" % (T, key, last[2], want)
        lt = landed_by(last[0])
        assert lt is not None and lt <= T - 1, "RAW: %s read at step %d, guaranteed landed only by the wait of step %s" % (want, T, lt)
    for tt, key, tag in seq:
        assert not [T for T, k, _ in reads if k == key and T == tt], "WAR: %s overwritten in the step that still reads it (%d)" % (key, tt)
    # (the in-order issue sequence inside a step is the source order: the vmcnt table counts on it)


def _r3_schedule(sch):
    i = SRC.index("auto phase3 = [&](auto phc, auto parc")
    body = SRC[i:SRC.index("auto period3 = [&]", i)]
    pa, pb = _ints(r"constexpr int PA3\[3\]", body)[:3], _ints(r"PB3\[3\]", body)[:3]
    sa = [int(x) for x in re.findall(r"\d+", re.search(r"SLOT_A3\[2\]\[2\]\s*=\s*\{(.*?)\}\};", body).group(1))]
    assert len(sa) == 4
    slot_a = [sa[:2], sa[2:4]]
    vm_all = [int(x) for x in re.findall(r"\d+", re.search(r"constexpr int VM3\[2\]\[6\]\s*=\s*\{(.*?)\}\};", body).group(1))]
    assert len(vm_all) == 12
    vm = vm_all[6 * sch:6 * sch + 6]
    rdb_steps = [int(x) for x in re.findall(r"S == (\d)", re.search(r"constexpr bool rdB = HALF == 0 && \((.+?)\);", body).group(1))]
    assert re.search(r"constexpr int sb = PB3\[S\];", body), "plane p of B lives in B slot p"
    issue = {ph: [] for ph in range(6)}
    for m in re.finditer(r"if constexpr \(SCH == (\d) && PH == (\d)\) \{(.*?)\}\n", body):
        if int(m.group(1)) == sch:
            issue[int(m.group(2))] = [(int(a), int(b), int(c), d.strip(), e[-1] == "n") for a, b, c, d, e in
                                      re.findall(r"issue\((\d), (\d), (\d), ([^,]+), (kw[ab]_[cn])\);", m.group(3))]
    j = SRC.index("// R3 prologue: the steady state at phase 0 of the first K-tile")
    seg = SRC[j:SRC.index("CDML_BARRIER();", j)]
    pro = [(int(a), int(b), int(c), int(d)) for a, b, c, d in re.findall(r"issue\((\d), (\d), (\d), (\d), k[ab]3\);", seg)]
    pro_vm = int(re.search(r"vmcnt\((\d+)\)", seg).group(1))
    return pa, pb, slot_a, vm, issue, rdb_steps, pro, pro_vm


import pytest  # noqa: E402


@pytest.mark.parametrize("sch", [0, 1])
def test_r3_schedule_hazards(sch):
    """The three-product period of the two-plane fp16 form (gemm_f16x2_256.hip compiles the same source): 8 half images in 6
    phases -- content, RAW and WAR replayed from the source like the six-product walk's, for both placements of the loads the
    source holds (0: the k-strided kernel's, 1: the k-contiguous kernels')."""
    PA, PB, SLOT_A, VM, ISSUE, RDB, PRO, PRO_VM = _r3_schedule(sch)
    assert sorted(zip(PA, PB)) == sorted([(0, 0), (1, 0), (0, 1)]), "hi.hi, lo.hi, hi.lo"
    assert RDB == [0, 2]
    n_per = 7
    seq = []                                                        # (time, (operand, slot, half), content) in issue order
    for n, (img, pl, hh, slot) in enumerate(PRO):
        seq.append((n - len(PRO), (img, slot, hh), (img, pl, hh, 0)))
    for w in range(n_per):
        PAR = w & 1
        for ph in range(6):
            for img, pl, hh, slot_expr, nxt in ISSUE[ph]:
                slot = eval(slot_expr, {"SLOT_A3": SLOT_A, "PAR": PAR})
                assert (img == 0 and 0 <= slot <= 2) or (img == 1 and 0 <= slot <= 1), "three A slots, two B slots: 160 KiB"
                seq.append((w * 6 + ph, (img, slot, hh), (img, pl, hh, w + (1 if nxt else 0))))
    assert len(seq) == len(PRO) + 8 * n_per, "eight half images per K-tile: every plane image once"
    reads = []
    for w in range(n_per):
        PAR = w & 1
        for ph in range(6):
            S, HALF = ph >> 1, ph & 1
            reads.append((w * 6 + ph, (0, SLOT_A[PAR][PA[S]], HALF), (0, PA[S], HALF, w)))
            if HALF == 0 and S in RDB:
                reads += [(w * 6 + ph, (1, PB[S], hh), (1, PB[S], hh, w)) for hh in (0, 1)]

    def landed_by(idx):
        for T in range(-1, n_per * 6):
            issued = sum(1 for tt, _, _ in seq if tt <= T)
            in_flight = (PRO_VM if T == -1 else VM[T % 6]) // 2    # half images allowed to stay in flight (two pieces each)
            if idx < issued - in_flight:
                return T
        return None

    for T, key, want in reads:
        last = [(i, tt, tag) for i, (tt, k, tag) in enumerate(seq) if k == key and tt < T][-1]
        assert last[2] == want, "phase %d reads %s: the slot holds %s, not %s" % (T, key, last[2], want)
        lt = landed_by(last[0])
        assert lt is not None and lt <= T - 1, "RAW: %s read at phase %d, guaranteed landed only by the wait of phase %s" % (want, T, lt)
    for tt, key, tag in seq:
        assert not [T for T, k, _ in reads if k == key and T == tt], "WAR: %s overwritten in the phase that still reads it (%d)" % (key, tt)
