"""The kill matrix of the oracle's known answers (parity unpinned -- this is the most that can be done for it here).

`tests/golden/kat.json` holds hand-derived known answers for the restated OpenCV 4.4 arithmetic.  A known-answer set pins a quirk of
Appendix A only if a restatement that gets THAT quirk wrong fails at least one of them.  tests/oracle_mutants.py is a switchable
restatement: unmutated it equals the oracle bit for bit; each of its mutants gets one quirk wrong (round half up, an FMA in the blend
or in `x * inv - 0.5f`, weights after the clamp, a pad on the indivisible axis only, the clip limit in float, redistribution until
stable, `255 / total`, ...).  This test asserts that EVERY mutant is killed by at least one known answer, prints the mutant x KAT
matrix, and checks that every known answer is reproduced by both oracles and -- for the round-5 entries -- by the exact-arithmetic
derivation of tests/golden/derive_kats.py, which shares no code with oracle/.

The reference's own check (1frameMeasure.cpp:91-100: absdiff + analyzeDiff with a tolerance of 1) is evaluated beside it: the last
column says which mutants IT would have let through on the same inputs."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle

HERE = Path(__file__).parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE / "golden"))
import derive_kats  # noqa: E402
import oracle_mutants as M  # noqa: E402

KAT = json.loads((HERE / "golden" / "kat.json").read_text())


def _src(k):
    h, w = k["shape"]
    if "src" in k:
        return np.array(k["src"], np.uint8).reshape(h, w)
    if "src_runs" in k:
        return np.concatenate([np.full(n, v, np.uint8) for v, n in k["src_runs"]]).reshape(h, w)
    if "src_const" in k:
        return np.full((h, w), k["src_const"], np.uint8)
    if "src_arange" in k:
        return np.arange(k["src_arange"], dtype=np.uint8).reshape(h, w)
    q = k["src_quadrants"]
    a = np.empty((h, w), np.uint8)
    a[: h // 2, : w // 2], a[: h // 2, w // 2:], a[h // 2:, : w // 2], a[h // 2:, w // 2:] = q
    return a


def _all_kats():
    """(id, op, kat) for every entry that is a vector; EQ-6 (a 4K constant frame) is represented by its 8x8 corner here -- the
    full-size form runs in test_oracle.py and on the GPU."""
    out = [(k["id"], "equalize", k) for k in KAT["equalize"]]
    out += [(k["id"], "clahe", k) for k in KAT["clahe"]]
    out += [(k["id"], k["op"], k) for k in KAT["derived"]]
    return out


def _verdict(k, op, eq_fn, clahe_fn):
    """(passes, worst absolute difference or None) of an implementation on one known answer."""
    if k["id"] == "CL-3":                                 # geometry KAT: the LUT stage must see a 24x16 image, tiles 3x2
        return None, None
    src = _src(k)
    if src.size > 1 << 16:
        src = src[:8, :8]
    out = eq_fn(src) if op == "equalize" else clahe_fn(src, k["clip"], *k["tiles"])
    ok, worst = True, 0
    if "dst" in k:
        want = np.array(k["dst"], np.uint8).reshape(src.shape)
        ok &= bool(np.array_equal(out, want))
        worst = int(np.abs(out.astype(int) - want.astype(int)).max())
    if "dst_const" in k:
        ok &= bool((out == k["dst_const"]).all())
        worst = max(worst, int(np.abs(out.astype(int) - k["dst_const"]).max()))
    if "dst_arange" in k:
        want = np.arange(k["dst_arange"]).reshape(src.shape)
        ok &= bool(np.array_equal(out, want))
        worst = max(worst, int(np.abs(out.astype(int) - want).max()))
    if "lut" in k:
        for v, want in k["lut"].items():
            sel = out[src == int(v)]
            ok &= bool((sel == want).all())
            worst = max(worst, int(np.abs(sel.astype(int) - want).max()))
    return ok, worst


def test_unmutated_harness_is_the_oracle():
    """The switchable restatement with no switch thrown must be the oracle itself, or the matrix below says nothing about it."""
    rng = np.random.default_rng(20261005)
    for it in range(60):
        h, w = (int(v) for v in rng.integers(1, 48, 2))
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        if it % 3 == 0:
            a = rng.choice(rng.choice(256, 3, replace=False), (h, w)).astype(np.uint8)
        assert np.array_equal(M.equalize_hist(a), oracle.np_equalize_hist(a)) and np.array_equal(M.equalize_hist(a), oracle.equalize_hist(a))
        tx, ty = (int(v) for v in rng.integers(1, 9, 2))
        cl = float(rng.choice([0.0, 0.7, 2.0, 3.0, 40.0]))
        got = M.clahe(a, cl, tx, ty)
        assert np.array_equal(got, oracle.np_clahe(a, cl, tx, ty)) and np.array_equal(got, oracle.clahe(a, cl, tx, ty)), (h, w, tx, ty, cl)


@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_both_oracles_reproduce_every_known_answer(impl):
    eq = oracle.equalize_hist if impl == "c" else oracle.np_equalize_hist
    cl = oracle.clahe if impl == "c" else oracle.np_clahe
    for kid, op, k in _all_kats():
        ok, _ = _verdict(k, op, eq, cl)
        assert ok in (True, None), kid


def test_derived_answers_follow_from_appendix_a_in_exact_arithmetic():
    """kat.json's round-5 answers are what tests/golden/derive_kats.py derives from Appendix A with exact rationals and explicit
    binary32 roundings (no numpy, no code of oracle/): the file was not edited by hand afterwards and not produced by the oracle."""
    for k in KAT["derived"]:
        out = derive_kats.derived_answer(k)
        assert [v for row in out for v in row] == k["dst"], k["id"]
    # ... and the derivation also reproduces the older, hand-written answers it can express
    assert derive_kats.equalize_hist([[50, 50, 100, 200]]) == [[0, 0, 128, 255]]                      # EQ-1
    assert derive_kats.equalize_hist([[10, 10, 10, 20]]) == [[0, 0, 0, 255]]                           # EQ-4
    assert derive_kats.clahe([[7] * 4] * 4, 2.0, 1, 1) == [[32] * 4] * 4                               # CL-1
    k4 = next(k for k in KAT["clahe"] if k["id"] == "CL-4")
    assert [v for row in derive_kats.clahe(_src(k4).tolist(), k4["clip"], *k4["tiles"]) for v in row] == k4["dst"]


def test_every_mutant_is_killed_by_a_known_answer(capsys):
    kats = _all_kats()
    ids = [kid for kid, _, _ in kats]
    rows, survivors, ref_check_blind = [], [], []
    for name in M.MUTANTS:
        eq = lambda s, n=name: M.equalize_hist(s, n)
        cl = lambda s, c, tx, ty, n=name: M.clahe(s, c, tx, ty, mutant=n)
        killed_by, max_err = [], 0
        for kid, op, k in kats:
            if (op == "equalize") != name.startswith("eq_"):
                continue
            if kid == "CL-3":
                ew, eh, tw, th, _ = M.clahe_geometry(k["shape"][1], k["shape"][0], 2.0, *k["tiles"], name)
                if [ew, eh] != k["ext"] or [tw, th] != k["tile"]:
                    killed_by.append(kid)
                    max_err = max(max_err, 255)            # a different geometry is not a +-1 matter
                continue
            ok, worst = _verdict(k, op, eq, cl)
            if not ok:
                killed_by.append(kid)
                max_err = max(max_err, worst)
        rows.append((name, killed_by, max_err))
        if not killed_by:
            survivors.append(name)
        if killed_by and max_err <= 1:
            ref_check_blind.append(name)
    with capsys.disabled():
        print("\nkill matrix: mutant of the restated OpenCV 4.4 arithmetic -> known answers (tests/golden/kat.json) it FAILS")
        print(f"  {'mutant':34s} {'killed by':44s} worst |diff|   the reference's own +-1 check (1frameMeasure.cpp:91-100)")
        for name, killed_by, max_err in rows:
            print(f"  {name:34s} {','.join(killed_by) or 'SURVIVES':44s} {max_err:5d}         {'would pass it' if max_err <= 1 else 'would catch it'}")
        print(f"  {len(rows)} mutants, {len(survivors)} survive; {len(ref_check_blind)} of the killed ones differ by at most 1 grey level on every known answer "
              f"-- a +-1 tolerance sees none of them")
    assert survivors == [], f"known answers discriminate nothing about: {survivors}"
    # every round-5 entry kills the mutant(s) it was made for
    rowmap = {name: killed_by for name, killed_by, _ in rows}
    for k in KAT["derived"]:
        for m in k["guards"]:
            assert k["id"] in rowmap[m], (k["id"], m)
    assert set(ids) >= {"EQ-1", "EQ-2", "EQ-7", "CL-1", "CL-5", "CL-19"}
