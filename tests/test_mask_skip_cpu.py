"""CPU checks around the constant region of the fisheye mask (csrc/superpoint.hip sp_plan_mask_skip, csrc/conv.hip tile_origin; the GPU side is
tests/test_gpu_mask_skip.py): the integer arithmetic of the tile walk -- the decode of a tile number into (image, tile row, tile column) over the
tiles that run, by multiply-high divisions -- enumerates every tile outside the rectangle exactly once, for every rectangle of several grids; the
plan bench.py mirrors for its FLOP accounting gives the rectangles worked out by hand for 600 x 480 (docs/history/rounds_2_to_5.md, round 3)."""
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


def _walk_vlad(tiles_x, tiles_y, sy0, sy1, sx0, sw, batch):
    """vlad_s.hip: launch_sb (the parameters) + full_tile (the t-th tile that runs -> its number in the full grid), in Python integers"""
    skip = sy1 > sy0 and sw > 0
    y0, y1, x0, w = (sy0, sy1, sx0, sw) if skip else (0, 0, 0, 0)
    tiles_img = tiles_x * tiles_y
    act = tiles_img - (y1 - y0) * w
    bw = tiles_x - w
    above = y0 * tiles_x if skip else act
    upto = above + (y1 - y0) * bw
    m_act, m_bw = _magic(act), _magic(bw)
    out = []
    for t in range(batch * act):
        if y1 <= y0:
            out.append(t)
            continue
        tb = _umulhi(t, m_act) if m_act else t
        r = t - tb * act
        if r < above:
            ttr = r
        elif r < upto:
            q = r - above
            ry = _umulhi(q, m_bw) if m_bw else q
            c = q - ry * bw
            ttr = (y0 + ry) * tiles_x + (c if c < x0 else c + w)
        else:
            ttr = r - upto + y1 * tiles_x
        out.append(tb * tiles_img + ttr)
    return out, (y0, y1, x0, x0 + w, skip)


def test_mobilenetvlad_tile_walk_enumerates_the_tiles_outside_the_rectangle_once():
    """the persistent block kernel of MobileNetVLAD under the fisheye mask (round 6): rectangles up to the full width of the grid (no tile left in a band row)"""
    for tx, ty in [(19, 30), (10, 15), (5, 8), (1, 2), (3, 4)]:
        rects = [(0, 0, 0, 0)] + [(a, b, c, d - c) for a in range(ty) for b in range(a + 1, ty + 1) for c in range(tx) for d in range(c + 1, tx + 1)
                                  if (b - a) * (d - c) < tx * ty and (ty <= 8 or (a % 5 == 3 and b % 4 == 1))]
        for sy0, sy1, sx0, sw in rects:
            seen, (y0, y1, x0, x1, skip) = _walk_vlad(tx, ty, sy0, sy1, sx0, sw, 3)
            want = [b * tx * ty + y * tx + x for b in range(3) for y in range(ty) for x in range(tx) if not (skip and y0 <= y < y1 and x0 <= x < x1)]
            assert seen == want, (tx, ty, sy0, sy1, sx0, sw)              # (in order: the walk is monotonic)


def test_plan_for_600x480_is_the_one_worked_out_by_hand():
    """omni_sp_mask_skip_plan (the library's own plan, pure arithmetic: callable without a device) -- bench.py takes its executed-FLOP accounting from the
    same function through omni_sp_stage_tiles_left_out"""
    import omni_loader
    capi = omni_loader.load().capi
    plan = lambda h, w, prec, layer: capi.sp_mask_skip_plan(w, h, prec, layer)
    F16, SPLIT, F32 = capi.PREC_F16, capi.PREC_SPLIT, capi.PREC_F32
    # rows 360-479 are blanked; conv1a is constant on rows 361-479; conv1b on 362-478 x 1-598 -> tile rows 46-58 (13 of 60), tile columns 1-17 (of 19)
    assert plan(480, 600, F16, 1) == ((46, 59, 1, 18), 13 * 17 / (60 * 19))
    assert plan(480, 600, F16, 2) == ((23, 29, 1, 9), 6 * 8 / (30 * 10))            # 240 x 300: rows 182-237 x columns 2-297 -> tile rows 23-28, columns 1-8
    assert plan(480, 600, F16, 3) == ((23, 29, 1, 9), 6 * 8 / (30 * 10))
    assert plan(480, 600, F16, 4) == ((12, 14, 1, 4), 2 * 3 / (15 * 5))             # 120 x 150: rows 93-116 x columns 3-146 -> tile rows 12-13, columns 1-3
    assert plan(480, 600, F16, 0)[1] == 0.0                                          # conv1a is fused into conv1b on the fp16 path
    # OMNI_PREC_SPLIT: conv1a in 8-row tile rows over the whole width (rows 361-479 -> tile rows 46-59), the cin = 64 layers in 4 x 32 tiles
    assert plan(480, 600, SPLIT, 0) == ((46, 60, 0, 19), 14 / 60)
    assert plan(480, 600, SPLIT, 1) == ((91, 119, 1, 18), 28 * 17 / (120 * 19))      # rows 362-478 -> 4-row tile rows 91-118
    assert plan(480, 600, SPLIT, 2) == ((46, 59, 1, 9), 13 * 8 / (60 * 10))          # rows 182-237
    assert plan(480, 600, SPLIT, 5) == ((47, 58, 1, 4), 11 * 3 / (60 * 5))           # conv3b, 120 x 150, 2 x 32 tiles: rows 94-115 x columns 4-145
    assert plan(480, 600, F16, 5) == ((16, 19, 1, 4), 3 * 3 / (20 * 5))              # conv3b on the fp16 register-stationary kernel: 6 x 32 tiles, rows 96-113
    for layer in range(6):
        assert plan(64, 96, F16, layer)[1] == 0.0                                    # the band is thinner than a tile row
        assert plan(480, 600, F32, layer)[1] == 0.0
