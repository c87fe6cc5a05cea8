"""CPU checks around the constant region of the fisheye mask (csrc/superpoint.hip sp_plan_mask_skip, csrc/conv.hip tile_origin; the GPU side is
tests/test_gpu_mask_skip.py): the integer arithmetic of the tile walk -- the decode of a tile number into (image, tile row, tile column) over the
tiles that run, by multiply-high divisions -- enumerates every tile outside the rectangle exactly once, for every rectangle of several grids; the
plan bench.py mirrors for its FLOP accounting gives the rectangles worked out by hand for 600 x 480."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _magic(d):
    return ((1 << 32) + d - 1) // d if d > 1 else 0          # ceil(2^32 / d); 0 = divisor 1 (2^32 does not fit 32 bits)


def _umulhi(a, b):
    return (a * b) >> 32


def _walk(tiles_x, tiles_y, sy0, sy1, sx0, sx1, batch):
    """conv.hip: launch_conv_pp_abl (the parameters) + tile_origin (the decode), in Python integers"""
    skip = sy1 > sy0 and sx1 > sx0
    y0, y1, x0, w = (sy0, sy1, sx0, sx1 - sx0) if skip else (0, 0, 0, 0)
    bw = tiles_x - w
    act = tiles_x * tiles_y - (y1 - y0) * w
    n_above = y0 * tiles_x if skip else act
    n_upto = n_above + (y1 - y0) * bw
    m_tpi, m_tx, m_bw = _magic(act), _magic(tiles_x), _magic(bw)
    assert batch * act * tiles_x * tiles_y < 1 << 32             # the launcher's exactness condition
    out = []
    for t in range(batch * act):
        b = _umulhi(t, m_tpi) if m_tpi else t
        r = t - b * act
        if r < n_above or r >= n_upto:
            base = 0
            if r >= n_upto:
                r, base = r - n_upto, y1
            ry = _umulhi(r, m_tx) if m_tx else r
            rx, ry = r - ry * tiles_x, ry + base
        else:
            r -= n_above
            q = _umulhi(r, m_bw) if m_bw else r
            c = r - q * bw
            ry, rx = y0 + q, (c if c < x0 else c + w)
        out.append((b, ry, rx))
    return out, (y0, y1, x0, x0 + w, skip)


def test_tile_walk_enumerates_the_tiles_outside_the_rectangle_once():
    for tx, ty in [(19, 60), (10, 30), (5, 15), (1, 1), (2, 3), (3, 8)]:
        rects = [(0, 0, 0, 0)] + [(a, b, c, d) for a in range(ty) for b in range(a + 1, ty + 1) for c in range(tx) for d in range(c + 1, tx + 1)
                                  if (b - a) * (d - c) < tx * ty and (ty <= 20 or (a % 7 == 4 and b % 5 == 4))]
        for sy0, sy1, sx0, sx1 in rects:
            seen, (y0, y1, x0, x1, skip) = _walk(tx, ty, sy0, sy1, sx0, sx1, 2)
            want = [(b, y, x) for b in range(2) for y in range(ty) for x in range(tx) if not (skip and y0 <= y < y1 and x0 <= x < x1)]
            assert sorted(seen) == want and len(set(seen)) == len(seen), (tx, ty, sy0, sy1, sx0, sx1)


def test_plan_for_600x480_is_the_one_worked_out_by_hand(monkeypatch):
    monkeypatch.delenv("OMNI_SP_MASK_SKIP", raising=False)
    spec = importlib.util.spec_from_file_location("bench_for_plan", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    plan = bench.mask_skip_plan(480, 600)
    # rows 360-479 are blanked; conv1a is constant on rows 361-479; conv1b on 362-478 x 1-598 -> tile rows 46-58 (13 of 60), tile columns 1-17 (of 19)
    assert abs(plan["conv1b"][0] - 13 * 17 / (60 * 19)) < 1e-12 and plan["conv1b"][1] == 2.0 * 480 * 600 * 64 * 64 * 9
    assert abs(plan["conv2a"][0] - 6 * 8 / (30 * 10)) < 1e-12            # 240 x 300: rows 182-237 x columns 2-297 -> tile rows 23-28, columns 1-8
    assert abs(plan["conv2b"][0] - 6 * 8 / (30 * 10)) < 1e-12
    assert abs(plan["conv3a"][0] - 2 * 3 / (15 * 5)) < 1e-12             # 120 x 150: rows 93-116 x columns 3-146 -> tile rows 12-13, columns 1-3
    assert bench.mask_skip_plan(64, 96) == {} or all(v[0] == 0 for v in bench.mask_skip_plan(64, 96).values())      # the band is thinner than a tile row
    monkeypatch.setenv("OMNI_SP_MASK_SKIP", "0")
    assert bench.mask_skip_plan(480, 600) == {}
