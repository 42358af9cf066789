"""CPU model of the tile-16 kernels' lane -> address map (zopt_amd/csrc/lqr_backward.hip, kernel prologue).

Re-states the index arithmetic of the kernel in Python and proves, for every supported (n, m) and every lane,
that each load stays inside the step's own matrix and that valid lanes read exactly the element the MFMA
layout needs.  A wrong map is a GPU memory fault, so it is checked here, without a GPU."""
import itertools

import pytest


def lane_map(n, m):
    KS = (n + 3) // 4
    NP = 4 * KS
    out = []
    for lane in range(64):
        g, c = lane >> 4, lane & 15
        cA = c < n
        cB = NP <= c < NP + m
        row0 = g < n
        laneA, laneB = row0 and cA, row0 and cB
        dF = 4 * n if laneA else 4 * m if laneB else 0
        q4n = 4 * n if laneA else 0
        pF0 = ("A", g * n + c) if laneA else ("B", g * m + (c - NP)) if laneB else ("A", 0)
        pQ0 = g * n + c if laneA else 0
        rec = dict(lane=lane, g=g, c=c, F=[], Q=[])
        for s in range(KS):
            row = 4 * s + g
            rowok = row < n
            arr, off = pF0
            off = off + (s * dF if rowok else 0)
            vF = rowok and (cA or cB)
            rec["F"].append((arr, off, vF, row))
            rec["Q"].append((pQ0 + (s * q4n if rowok else 0), rowok and cA, row))
        vRm = g < m and cB
        rec["Rm"] = ((g * m + (c - NP)) if vRm else 0, vRm)
        vBt = cA and g < m
        rec["Bt"] = ((c * m + g) if vBt else 0, vBt)
        vRt = c < m and g < m
        rec["Rt"] = ((c * m + g) if vRt else 0, vRt)
        rec["L"] = (g * n + c, g < m and cA)
        out.append(rec)
    return KS, NP, out


@pytest.mark.parametrize("n,m", list(itertools.product(range(1, 13), range(1, 5))))
def test_every_lane_in_bounds_and_reads_the_right_element(n, m):
    KS, NP, lanes = lane_map(n, m)
    assert NP + m <= 16
    size = {"A": n * n, "B": n * m}
    for r in lanes:
        g, c = r["g"], r["c"]
        for (arr, off, valid, row) in r["F"]:
            assert 0 <= off < size[arr]
            if valid:
                assert (arr, off) == (("A", row * n + c) if c < n else ("B", row * m + (c - NP)))
        for (off, valid, row) in r["Q"]:
            assert 0 <= off < n * n
            if valid:
                assert off == row * n + c
        off, valid = r["Rm"]
        assert 0 <= off < m * m and (not valid or off == g * m + (c - NP))
        off, valid = r["Bt"]
        assert 0 <= off < n * m and (not valid or off == c * m + g)
        off, valid = r["Rt"]
        assert 0 <= off < m * m and (not valid or off == c * m + g)
        off, valid = r["L"]
        assert not valid or 0 <= off < m * n
    # every element of every matrix is read by exactly one valid lane/K-step
    seenA = sorted(off for r in lanes for (arr, off, v, _) in r["F"] if v and arr == "A")
    seenB = sorted(off for r in lanes for (arr, off, v, _) in r["F"] if v and arr == "B")
    assert seenA == list(range(n * n)) and seenB == list(range(n * m))
    assert sorted(off for r in lanes for (off, v, _) in r["Q"] if v) == list(range(n * n))
    assert sorted(r["Rm"][0] for r in lanes if r["Rm"][1]) == list(range(m * m))
    assert sorted(r["L"][0] for r in lanes if r["L"][1]) == list(range(m * n))
