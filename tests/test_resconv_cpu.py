"""CPU: the static tile schedule of efts_resconv5 (host logic behind efts_resconv5_plan) and its argument checks.
No device work: the plan query never touches the GPU, and efts_resconv5 rejects bad arguments before any launch."""
import ctypes as C

import pytest
from hypothesis import given, settings, strategies as st

from efficient_tts_amd import build as B
from efficient_tts_amd import lib as L
from efficient_tts_amd import ops as P


@pytest.fixture(scope="module", autouse=True)
def lib():
    B.build(verbose=False)
    return L.load()


def cover(units):          # rows `units` half units (32 window rows) yield in ceil(units / 8) tiles
    return 32 * units - 4 * ((units + 7) // 8)


def check_plan(m, n, cus):
    groups, classes = P.resconv5_plan(m, n, cus)
    ntn = n // 256
    slots = max(1, cus // ntn)
    assert 1 <= len(classes) <= 4 and groups >= 1
    units = []
    for rows, tiles in classes:
        assert 1 <= len(tiles) <= 8 and all(2 <= t <= 8 for t in tiles)
        assert rows == sum(32 * t - 4 for t in tiles)                     # every row of the class is in exactly one tile
        assert max(tiles) - min(tiles) <= 1                               # as even as possible: no short, LDS-DMA-bound tile beside a tall one
        units.append(sum(tiles))
    # coverage: group g belongs to class g % len(classes) and owns `rows` consecutive rows
    per_round = sum(r for r, _ in classes)
    covered = (groups // len(classes)) * per_round + sum(r for r, _ in classes[:groups % len(classes)])
    assert covered >= m
    assert covered - m < max(r for r, _ in classes) + min(r for r, _ in classes)      # no group without rows
    # one round: never more workgroups than CUs, and the longest group is as short as the row count allows
    u_min = 2
    while cover(u_min) * slots < m:
        u_min += 1
    if u_min <= 64:
        assert groups <= slots
        assert max(units) == u_min
        assert min(units) >= u_min - 1
    return groups, classes


def test_plans_of_the_baseline_shapes():
    # (rows, expected classes): B x (T2 + 2) rows of BASELINE configs 2 / 3 / 5 on 256 CUs, 512 channels
    g, c = check_plan(64 * 802, 512, 256)
    assert [t for _, t in c] == [[7, 6], [6, 7]] and g <= 128            # 13 half units each (6.5 x 64 rows; 64-row units needed 7)
    g, c = check_plan(32 * 802, 512, 256)
    assert [t for _, t in c] == [[7], [6]]
    check_plan(16 * 1202, 512, 256)
    check_plan(16 * 1502, 512, 256)
    check_plan(1 * 802, 512, 256)


@settings(max_examples=300, deadline=None)
@given(m=st.integers(1, 400_000), n=st.sampled_from([256, 512, 768, 1024]), cus=st.sampled_from([64, 128, 256, 304]))
def test_plan_properties(m, n, cus):
    check_plan(m, n, cus)


def test_plan_argument_errors(lib):
    buf = (C.c_int32 * L.RC_PLAN_INTS)()
    assert lib.efts_resconv5_plan(0, 512, 256, buf, L.RC_PLAN_INTS) == -2          # EFTS_ESHAPE
    assert lib.efts_resconv5_plan(100, 500, 256, buf, L.RC_PLAN_INTS) == -2
    assert lib.efts_resconv5_plan(100, 512, 256, buf, 10) == -1                    # EFTS_EINVAL: buffer too small
    assert lib.efts_resconv5_plan(100, 512, 256, None, L.RC_PLAN_INTS) == -1


def _args(m=1000, plan=None):
    g = L.ResConv5Args()
    g.x, g.w, g.y = 0x1000, 0x2000, 0x3000           # never dereferenced: every call below fails before the launch
    g.ldx = g.ldw = g.ldy = 1024
    g.w_tap_stride = 512 * 1024
    g.split, g.m, g.n, g.nchunk, g.y_split = 1, m, 512, 8, 1
    if plan is not None:
        g.plan = plan
    return g


def test_resconv5_rejects_bad_arguments_before_launching(lib):
    g = _args(); g.n = 500
    assert lib.efts_resconv5(C.byref(g), None) == -2 and b"256" in lib.efts_last_error()
    g = _args(); g.split = 3
    assert lib.efts_resconv5(C.byref(g), None) == -1
    g = _args(); g.x = 0x1004
    assert lib.efts_resconv5(C.byref(g), None) == -3                                 # EFTS_EALIGN
    g = _args(); g.y = None
    assert lib.efts_resconv5(C.byref(g), None) == -1 and b"no output" in lib.efts_last_error()
    g = _args(); g.split, g.x_lo = 2, 0x4000
    assert lib.efts_resconv5(C.byref(g), None) == -1
    # explicit plans are validated on the host
    short = P.make_plan(500, [[4], [2]])                                             # covers 500 rows, not 1000
    g = _args(1000, short)
    assert lib.efts_resconv5(C.byref(g), None) == -1 and b"cover" in lib.efts_last_error()
    bad = P.make_plan(1000, [[8, 6]]); bad[2] += 1                                   # rows != sum of the tiles
    g = _args(1000, bad)
    assert lib.efts_resconv5(C.byref(g), None) == -1 and b"sum" in lib.efts_last_error()
    bad = P.make_plan(1000, [[8, 6]]); bad[4] = 9                                    # a tile height of 9 half units
    g = _args(1000, bad)
    assert lib.efts_resconv5(C.byref(g), None) == -1
    bad = P.make_plan(1000, [[8, 6]]); bad[4] = 1                                    # ... and of 1
    g = _args(1000, bad)
    assert lib.efts_resconv5(C.byref(g), None) == -1


def test_make_plan_round_trip():
    for m, classes in ((51328, [[7, 6], [6, 7]]), (1000, [[2]]), (25664, [[8], [7], [4, 3]])):
        buf = P.make_plan(m, classes)
        rows = [sum(32 * t - 4 for t in cl) for cl in classes]
        assert buf[1] == len(classes)
        covered = (buf[0] // len(classes)) * sum(rows) + sum(rows[:buf[0] % len(classes)])
        assert covered >= m and covered - m < max(rows) + min(rows)


def test_generated_main_loop_is_what_the_generator_writes(tmp_path):
    """csrc/efts_rc4_loop.inc (the hand-scheduled main loop of efts_resconv5's one-wave-per-SIMD kernel) is generated: the committed
    file must be exactly what tools/gen_rc4_asm.py produces, and the stream must have the shape the kernel's C++ side assumes --
    per tile height h: three chunk bodies of 5 steps of 8 h MFMAs, every MFMA on its own accumulator block per k-slice, one barrier per
    step plus the prologue's, staging loads / stores paired, all registers inside the declared clobber ranges."""
    import importlib.util
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_rc4_asm", os.path.join(root, "tools", "gen_rc4_asm.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    committed = open(g.OUT).read()
    g.OUT = str(tmp_path / "loop.inc")
    g.main()
    assert open(g.OUT).read() == committed, "efts_rc4_loop.inc is stale: run python tools/gen_rc4_asm.py"
    for h in range(2, 9):
        lines = g.Gen(h).build()
        mf = [ln for ln in lines if ln.startswith("v_mfma")]
        assert len(mf) == 3 * 5 * 8 * h
        assert sum(1 for ln in lines if ln == "s_barrier") == 3 * 5 - 1 + 1          # one per step but the tile's last; + the prologue's
        # per k-slice of 2 h MFMAs every accumulator block appears exactly once
        for q in range(0, len(mf), 2 * h):
            accs = [re.match(r"v_mfma_f32_32x32x16_bf16 a\[(\d+):", ln).group(1) for ln in mf[q:q + 2 * h]]
            assert sorted(map(int, accs)) == [16 * b for b in range(2 * h)]
        assert sum(1 for ln in mf if ln.endswith(", 0")) == 2 * h                      # the tile's first k-slice starts from zero
        # staging: as many ds_write_b128 of weight pieces as buffer_loads of them (+ the prologue's 8 loads, - the final step's missing re-loads)
        wl = sum(1 for ln in lines if ln.startswith("buffer_load_dwordx4") and f"v{g.VOW}" <= ln.split(", ")[1] <= f"v{g.VOW + 7}")
        ww = sum(1 for ln in lines if ln.startswith("ds_write_b128") and f"v{g.WRW}," in ln)
        assert ww == 3 * 5 * 8 and wl == ww
        for ln in lines:                                                               # registers stay inside what the statement clobbers
            for r in re.findall(r"\bv\[?(\d+)", ln):
                assert g.V0 <= int(r) < g.VEND, ln
            for r in re.findall(r"\bs\[?(\d+)", ln):
                assert g.S0 <= int(r) < g.SEND, ln
