"""CPU: the library's tile bookkeeping (terra::crop snapping, Step-3 tile grid, tiles.create
boxes, cellFromXY) must agree BIT-EXACTLY with the oracle's literal restatement, and the
oracle must satisfy the structural invariants of V73:656-895 (G6/G7 of SURVEY.md 8c)."""
import numpy as np
import pytest

import machisplin_amd as mhs
from oracle import tiles as ot

GRIDS = [  # nrow, ncol, tile_edge, xmin, ymax, res
    (1500, 1500, 1500, -78.0, -5.0, 1 / 1200),
    (1501, 1501, 1500, -78.0, -5.0, 1 / 1200),
    (2476, 3264, 1500, -77.7435765934, -5.809416782 + 2476 * 0.0008333333 - 2476 * 0.0008333333, 0.0008333333),
    (2000, 2000, 1500, -78.0, -5.0, 1 / 1200),
    (10000, 10000, 1500, -78.0, -5.0, 1 / 1200),
    (300, 400, 100, -78.0, -5.0, 1 / 1200),
    (250, 330, 100, 10.25, 45.5, 0.01),       # positive coordinates: the snapping flips direction
    (333, 217, 64, -0.37, 0.21, 0.003),        # straddles the origin
]


@pytest.mark.parametrize("nrow,ncol,edge,xmin,ymax,res", GRIDS)
def test_step3_windows_bit_exact(nrow, ncol, edge, xmin, ymax, res):
    g = mhs.Geometry(xmin, ymax, res, res, nrow, ncol)
    og = ot.Geom(xmin, ymax, res, res, nrow, ncol)
    nRx, nCx, fit, keep = mhs.tiles.step3_tile_windows(g, edge)
    onR, onC, ofw, okw = ot.step3_windows(og, edge)
    assert (nRx, nCx) == (onR, onC) == (-(-nrow // edge), -(-ncol // edge))
    assert np.array_equal(fit, np.array(ofw)) and np.array_equal(keep, np.array(okw))
    # invariants: every cell is kept by 1..4 tiles; keep inside fit; tile 0 is the south-west one
    cover = np.zeros((nrow, ncol), dtype=np.int8)
    for f, k in zip(fit, keep):
        assert f[0] <= k[0] < k[1] <= f[1] and f[2] <= k[2] < k[3] <= f[3]
        cover[k[0]:k[1], k[2]:k[3]] += 1
    assert cover.min() == 1 and cover.max() <= 4
    assert keep[0][1] == nrow and keep[0][2] == 0 and keep[-1][0] == 0 and keep[-1][3] == ncol


def test_crop_window_matches_oracle_on_random_extents():
    rng = np.random.default_rng(0)
    for xmin, ymax, res in [(-78.0, -5.0, 1 / 1200), (3.5, 60.0, 0.0125), (-0.4, 0.3, 0.001)]:
        g = mhs.Geometry(xmin, ymax, res, res, 700, 900)
        og = ot.Geom(xmin, ymax, res, res, 700, 900)
        for _ in range(300):
            x = np.sort(rng.uniform(g.xmin - 0.1, g.xmax + 0.1, 2))
            y = np.sort(rng.uniform(g.ymin - 0.1, g.ymax + 0.1, 2))
            if rng.random() < 0.5:  # edges exactly on cell centres: the fragile half-cell case
                x = g.x_from_col(np.sort(rng.integers(0, 900, 2)))
                y = g.y_from_row(np.sort(rng.integers(0, 700, 2))[::-1])
            e = (x[0], x[1], y[0], y[1])
            want = ot.crop_window(og, e)
            if want is None:
                with pytest.raises(mhs.MhsError):
                    mhs.tiles.crop_window(g, e)
            else:
                assert mhs.tiles.crop_window(g, e) == want


def test_tiles_create_matches_oracle():
    rng = np.random.default_rng(1)
    g = mhs.Geometry(-78.0, -5.0, 1 / 1200, 1 / 1200, 1000, 1300)
    og = ot.Geom(-78.0, -5.0, 1 / 1200, 1 / 1200, 1000, 1300)
    xy = np.column_stack([rng.uniform(g.xmin, g.xmax, 500), rng.uniform(g.ymin, g.ymax, 500)])
    t = mhs.tiles.tiles_create(g, xy, out_ncol=3, out_nrow=2, feather_d=50)
    boxes, wins, sel = ot.tiles_create(og, xy, 3, 2, 50)
    assert np.array_equal(t["e.ext"], np.array(boxes)) and np.array_equal(t["win"], np.array(wins))
    assert all(np.array_equal(a, b) for a, b in zip(t["dat"], sel))
    # feather.d = 50 pixels => neighbouring tiles overlap by ~50 columns / rows
    assert 48 <= t["win"][0][3] - t["win"][1][2] <= 52


def test_cells_from_xy():
    g = mhs.Geometry(-78.0, -5.0, 0.5, 0.25, 4, 3)
    og = ot.Geom(-78.0, -5.0, 0.5, 0.25, 4, 3)
    pts = np.array([[-77.75, -5.125], [-76.5, -6.0], [-78.0, -5.0], [-79.0, -5.5], [-77.2, np.nan], [-76.51, -5.99]])
    rows, cols = mhs.tiles.cells_from_xy(g, pts)
    assert list(rows) == [0, 3, 0, -1, -1, 3] and list(cols) == [0, 2, 0, -1, -1, 2]
    for (x, y), r, c in zip(pts[:4], rows, cols):
        oc, orr = og.col_from_x(x), og.row_from_y(y)
        assert (r, c) == ((orr, oc) if oc >= 0 and orr >= 0 else (-1, -1))


def test_feather_kat_two_constant_tiles_give_a_linear_ramp():
    """G7: tiles A = 0 (west), B = 1 (east) => the strip is a linear ramp 0..1 along x."""
    og = ot.Geom(10.0, 50.0, 0.01, 0.01, 120, 300)  # positive coordinates
    nRx, nCx, fw, kw = ot.step3_windows(og, 150)
    assert (nRx, nCx) == (1, 2)
    tiles = [np.zeros((kw[0][1] - kw[0][0], kw[0][3] - kw[0][2])), np.ones((kw[1][1] - kw[1][0], kw[1][3] - kw[1][2]))]
    layers = [ot.extend_full(og, kw[h], tiles[h]) for h in range(2)]
    base = ot.mosaic_mean(layers[::-1])
    fin = ot.feather_and_merge(og, nRx, nCx, kw, tiles, base)
    assert not np.isnan(fin).any()
    w, strip = ot._feather_pair(og, layers[0], layers[1], "x")
    ncs = w[3] - w[2]
    assert np.allclose(strip[0], np.linspace(0, 1, ncs)) and np.allclose(strip, strip[0][None, :])
    assert np.array_equal(fin[w[0]:w[1], w[2]:w[3]], strip)
    # rows / columns of the overlap that the half-cell snap drops keep the plain mean (0.5)
    inter = fin[:, kw[1][2]:kw[0][3]]
    assert set(np.unique(inter[(inter != 0.5)]).round(12)) <= set(np.linspace(0, 1, ncs).round(12))
    assert (fin[:, :kw[1][2]] == 0).all() and (fin[:, kw[0][3]:] == 1).all()


def test_seam_count():
    import ctypes as C
    n = C.c_int64()
    assert mhs._lib.load().mhs_seam_count(7, 7, C.byref(n)) == 0 and n.value == 6 * 7 + 7 * 6
    assert mhs._lib.load().mhs_seam_count(1, 2, C.byref(n)) == 0 and n.value == 1
