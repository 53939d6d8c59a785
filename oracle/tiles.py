"""Oracle: tile bookkeeping of machisplin.mltps Step 3/4/5 and machisplin.tiles.create/merge,
restated LITERALLY on a mini-terra (numpy arrays + extents) so each statement can be read
against the R source.  TEST INFRASTRUCTURE ONLY (oracle/__init__.py).  PARITY UNPINNED vs
R/terra: `terra` is a CRAN dependency (NAMESPACE:11, unpinned, not vendored); its snapping
rules below are restated from its published C++ (SpatRaster::origin/align/crop, colFromX,
rowFromY, xFromCol, yFromRow) and cannot be executed here.

Reference code followed (V73):
  tile boxes            V73:649-681        per-tile TPS           V73:687-738
  mean mosaic           V73:739-747        single-tile branch     V73:748-753
  vertical feather      V73:764-806        horizontal feather     V73:815-877
  combine               V73:880-895        sum / extract / R2     V73:906-930
  tiles.create          V73:1165-1256      tiles.merge            V73:1392-1548
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import tps as otps


# ------------------------------------------------------------------ mini terra --
def c_round(x: float) -> float:
    """C's round(): half away from zero (terra is C++; Python's round() is half-to-even)."""
    return math.copysign(math.floor(abs(x) + 0.5), x)


@dataclass(frozen=True)
class Geom:
    xmin: float
    ymax: float
    xres: float
    yres: float
    nrow: int
    ncol: int

    @property
    def xmax(self):
        return self.xmin + self.ncol * self.xres

    @property
    def ymin(self):
        return self.ymax - self.nrow * self.yres

    def ext(self):
        return (self.xmin, self.xmax, self.ymin, self.ymax)

    def x_from_col(self, col):
        return self.xmin + (np.asarray(col, dtype=np.float64) + 0.5) * self.xres

    def y_from_row(self, row):
        return self.ymax - (np.asarray(row, dtype=np.float64) + 0.5) * self.yres

    def origin(self):
        """SpatRaster::origin(): the grid-line intersection nearest (0, 0)."""
        x = self.xmin - self.xres * c_round(self.xmin / self.xres)
        y = self.ymax - self.yres * c_round(self.ymax / self.yres)
        if math.isclose(self.xres + x, abs(x), rel_tol=0, abs_tol=1e-12 * self.xres):
            x = abs(x)
        if math.isclose(self.yres + y, abs(y), rel_tol=0, abs_tol=1e-12 * self.yres):
            y = abs(y)
        return x, y

    def col_from_x(self, x: float) -> int:
        """colFromX: -1 outside; the east edge belongs to the last column."""
        if x == self.xmax:
            return self.ncol - 1
        if x < self.xmin or x > self.xmax:
            return -1
        return int(math.floor((x - self.xmin) / self.xres))

    def row_from_y(self, y: float) -> int:
        if y == self.ymin:
            return self.nrow - 1
        if y < self.ymin or y > self.ymax:
            return -1
        return int(math.floor((self.ymax - y) / self.yres))


def align_near(g: Geom, e):
    """SpatRaster::align(e, "near"): snap an extent to the raster's cell boundaries."""
    ox, oy = g.origin()
    xmn = c_round((e[0] - ox) / g.xres) * g.xres + ox
    xmx = c_round((e[1] - ox) / g.xres) * g.xres + ox
    ymn = c_round((e[2] - oy) / g.yres) * g.yres + oy
    ymx = c_round((e[3] - oy) / g.yres) * g.yres + oy
    if xmn == xmx:
        if xmn <= e[0]:
            xmx = xmx + g.xres
        else:
            xmn = xmn - g.xres
    if ymn == ymx:
        if ymn <= e[2]:
            ymx = ymx + g.yres
        else:
            ymn = ymn - g.yres
    return (xmn, xmx, ymn, ymx)


def crop_window(g: Geom, e):
    """terra::crop(x, e): align, intersect with x's extent, then the cell window is found from
    the cell centres half a cell inside the snapped extent.  Returns (r0, r1, c0, c1) half-open,
    or None if the extents do not overlap."""
    a = align_near(g, e)
    xmn, xmx = max(a[0], g.xmin), min(a[1], g.xmax)
    ymn, ymx = max(a[2], g.ymin), min(a[3], g.ymax)
    if not (xmn < xmx and ymn < ymx):
        return None
    c0 = g.col_from_x(xmn + 0.5 * g.xres)
    c1 = g.col_from_x(xmx - 0.5 * g.xres)
    r0 = g.row_from_y(ymx - 0.5 * g.yres)
    r1 = g.row_from_y(ymn + 0.5 * g.yres)
    return (r0, r1 + 1, c0, c1 + 1)


def window_geom(g: Geom, w) -> Geom:
    """Geometry of a cropped raster: its extent is rebuilt from the parent's edge and the
    window (terra keeps the snapped extent; cell centres derive from it)."""
    r0, r1, c0, c1 = w
    return Geom(g.xmin + c0 * g.xres, g.ymax - r0 * g.yres, g.xres, g.yres, r1 - r0, c1 - c0)


def extend_full(g: Geom, w, vals):
    """terra::extend(x, full): NA outside x's window."""
    out = np.full((g.nrow, g.ncol), np.nan)
    r0, r1, c0, c1 = w
    out[r0:r1, c0:c1] = vals
    return out


def mosaic_mean(layers):
    """terra::mosaic(sprc, fun="mean"): NA-aware mean, summed in collection order."""
    s = np.zeros_like(layers[0])
    n = np.zeros(layers[0].shape, dtype=np.int64)
    for a in layers:
        ok = ~np.isnan(a)
        s = np.where(ok, s + np.where(ok, a, 0.0), s)
        n += ok
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(n > 0, s / n, np.nan)


def mosaic_first(layers):
    out = layers[0].copy()
    for a in layers[1:]:
        out = np.where(np.isnan(out), a, out)
    return out


def points_extent(g: Geom, full):
    """terra::ext(terra::as.points(r)): bounding box of the CENTRES of the non-NA cells."""
    rows, cols = np.nonzero(~np.isnan(full))
    if rows.size == 0:
        return None
    x = g.x_from_col(np.array([cols.min(), cols.max()]))
    y = g.y_from_row(np.array([rows.max(), rows.min()]))
    return (float(x[0]), float(x[1]), float(y[0]), float(y[1]))


# ------------------------------------------------------------ Step 3 tile grid --
def step3_tile_boxes(g: Geom, tile_edge: int = 1500, fit_overlap: float = 0.2, keep_overlap: float = 0.025):
    """V73:656-681: nRx x nCx tiles, row-major from the south-west; fit box = cell +- 0.2 of the
    tile size, keep box = cell +- 0.025."""
    nRx = int(math.ceil(g.nrow / tile_edge))
    nCx = int(math.ceil(g.ncol / tile_edge))
    xmin, xmax, ymin, ymax = g.ext()
    longDist = (xmax - xmin) / nCx
    latDist = (ymax - ymin) / nRx
    fit, keep = [], []
    for j in range(1, nRx + 1):
        for h in range(1, nCx + 1):
            fit.append((xmin + ((longDist * (h - 1)) - (longDist * fit_overlap)), xmin + ((longDist * h) + (longDist * fit_overlap)),
                        (ymin + ((latDist * (j - 1))) - (latDist * fit_overlap)), (ymin + ((latDist * j)) + (latDist * fit_overlap))))
    for j in range(1, nRx + 1):
        for h in range(1, nCx + 1):
            keep.append((xmin + ((longDist * (h - 1)) - (longDist * keep_overlap)), xmin + ((longDist * h) + (longDist * keep_overlap)),
                         (ymin + ((latDist * (j - 1))) - (latDist * keep_overlap)), (ymin + ((latDist * j)) + (latDist * keep_overlap))))
    return nRx, nCx, fit, keep


def step3_windows(g: Geom, tile_edge=1500):
    """Integer windows of every tile: fit window = crop(rast_stack, b) (V73:699); keep window =
    crop(pred, d) (V73:728) -- a crop of the FIT raster, so it is intersected with it."""
    nRx, nCx, fit, keep = step3_tile_boxes(g, tile_edge)
    fw, kw = [], []
    for b, d in zip(fit, keep):
        wf = crop_window(g, b)
        gf = window_geom(g, wf)
        wk = crop_window(gf, d)
        fw.append(wf)
        kw.append((wf[0] + wk[0], wf[0] + wk[1], wf[2] + wk[2], wf[2] + wk[3]))
    return nRx, nCx, fw, kw


def stations_in_window(g: Geom, w, xy, cov1_full):
    """terra::extract(rb[[1]], Full.cords) then complete.cases (V73:701-706): stations whose
    cell lies in the cropped raster and whose first covariate there is not NA."""
    r0, r1, c0, c1 = w
    gw = window_geom(g, w)
    sel = []
    for i, (x, y) in enumerate(xy):
        c, r = gw.col_from_x(x), gw.row_from_y(y)
        if c < 0 or r < 0:
            continue
        if cov1_full is not None and np.isnan(cov1_full[r0 + r, c0 + c]):
            continue
        sel.append(i)
    return np.array(sel, dtype=np.int64)


def step3_tps_tiles(g: Geom, xy, resid, cov1_full=None, tile_edge=1500, gcv_mode="fields", lam=None):
    """V73:687-753.  Returns (nRx, nCx, keep windows, per-tile keep-window arrays, rast.mosaic)."""
    nRx, nCx, fw, kw = step3_windows(g, tile_edge)
    if nRx * nCx == 1:
        m = otps.fit(xy, resid, lam=lam, gcv_mode=gcv_mode)
        full = otps.predict_grid(m, g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol)
        return nRx, nCx, [(0, g.nrow, 0, g.ncol)], [full], full
    tiles = []
    for h in range(nRx * nCx):
        sel = stations_in_window(g, fw[h], xy, cov1_full)
        r0, r1, c0, c1 = kw[h]
        if sel.size < 10:  # V73:710-721: zero tile
            tiles.append(np.zeros((r1 - r0, c1 - c0)))
            continue
        m = otps.fit(xy[sel], resid[sel], lam=lam, gcv_mode=gcv_mode)
        # terra::interpolate(terra::rast(rb), tps): cell centres of the FIT raster
        gf = window_geom(g, fw[h])
        wk = (r0 - fw[h][0], r1 - fw[h][0], c0 - fw[h][2], c1 - fw[h][2])
        tiles.append(otps.predict_grid(m, gf.xmin, gf.ymax, gf.xres, gf.yres, gf.nrow, gf.ncol, *wk))
    full_layers = [extend_full(g, kw[h], tiles[h]) for h in range(nRx * nCx)]
    # ie <- sprc(raster_sTPS[[j]], ie) prepends: the collection runs from the LAST tile to the first
    rast_mosaic = mosaic_mean(full_layers[::-1])
    return nRx, nCx, kw, tiles, rast_mosaic


# ---------------------------------------------------------------- Step 4 feather --
def _feather_pair(g: Geom, A_full, B_full, axis: str):
    """One seam (V73:772-799 / 829-868): A = raster_sTPS[[v]] (west or south tile), B = its east
    or north neighbour, both extended to the full grid.  Returns (window, strip values) or None."""
    AAA = A_full + B_full
    e = points_extent(g, AAA)
    if e is None:
        return None
    w = crop_window(g, e)
    if w is None:
        return None
    r0, r1, c0, c1 = w
    gs = window_geom(g, w)
    bR1, bR2 = A_full[r0:r1, c0:c1], B_full[r0:r1, c0:c1]
    if axis == "x":
        coord = np.tile(gs.x_from_col(np.arange(gs.ncol)), (gs.nrow, 1))
    else:
        coord = np.tile(gs.y_from_row(np.arange(gs.nrow))[:, None], (1, gs.ncol))
    with np.errstate(invalid="ignore", divide="ignore"):
        delta_L1 = coord.max() - coord.min()
        stD1 = (coord - coord.min()) / delta_L1
        stD1 = 1 - stD1
        delta_L2 = coord.max() - coord.min()
        stD2 = (coord - coord.min()) / delta_L2
        feath = bR2 * stD2 + bR1 * stD1
    return w, feath


def feather_and_merge(g: Geom, nRx, nCx, kw, tiles, rast_mosaic, merge_mode=False):
    """V73:760-895 (Step 4) or, with merge_mode, machisplin.tiles.merge V73:1399-1546.
    Returns final.TPS on the full grid."""
    n = nRx * nCx
    if n == 1:
        return rast_mosaic
    layers = [extend_full(g, kw[h], tiles[h]) for h in range(n)]
    strips = []  # creation order
    for j in range(1, nRx + 1):
        for h in range(1, nCx + 1):
            v = h + (j * nCx) - nCx
            if h < nCx:
                res = _feather_pair(g, layers[v - 1], layers[v], "x")
                if res is not None:
                    strips.append(extend_full(g, res[0], res[1]))
    f_clock = 0
    for j in range(1, nRx + 1):
        for h in range(1, nCx + 1):
            f_clock += 1
            f_timer = (nRx * nCx) - nCx + 1
            v = h + (j * nCx) - nCx
            if f_clock < f_timer:
                res = _feather_pair(g, layers[v - 1], layers[v + nCx - 1], "y")
                if res is not None:
                    strips.append(extend_full(g, res[0], res[1]))
    if not strips:
        return rast_mosaic
    if merge_mode:
        # feath.ras.out <- c(feath.ras.out, feath.ras): creation order; ic prepends => reversed
        coll = strips[::-1]
    else:
        # feath.ras.TPS <- rast(list(feath.ras, feath.ras.TPS)) prepends; ic prepends again
        coll = strips
    if n > 2:
        feath_mosaic = mosaic_mean(coll)
        return mosaic_first([feath_mosaic, rast_mosaic])
    return mosaic_first([strips[0], rast_mosaic])  # terra::merge(feath, mosaic), V73:891-892


# --------------------------------------------------------------------- Step 5 --
def step5_combine(g: Geom, pred_elev, final_tps, xy, resp, rsq_model=None):
    """V73:906-930: sum, extract at the stations, R^2, keep the sum iff it beats the ensemble."""
    total = pred_elev + final_tps  # app(sum): NA if either is NA
    rows = np.array([g.row_from_y(y) for y in xy[:, 1]])
    cols = np.array([g.col_from_x(x) for x in xy[:, 0]])
    f_actual = total[rows, cols]
    tss = np.sum((resp - resp.mean()) ** 2)
    rss_final = np.sum((resp - f_actual) ** 2)
    rsq_final = 1 - rss_final / tss
    if rsq_model is None:
        rss_m = np.sum((resp - pred_elev[rows, cols]) ** 2)
        rsq_model = 1 - rss_m / tss
    final = total if rsq_final > rsq_model else pred_elev
    return final, rsq_model, rsq_final, resp - f_actual


# ------------------------------------------------------- machisplin.tiles.create --
def tiles_create_boxes(g: Geom, out_ncol=3, out_nrow=3, feather_d=50):
    """V73:1165-1197: tile box = grid cell +- feather.d/2 PIXELS; row-major from the south-west."""
    feather_d = feather_d / 2
    xmin, xmax, ymin, ymax = g.ext()
    nRx, nCx = out_nrow, out_ncol
    long_pix = (xmax - xmin) / g.ncol
    lat_pix = (ymax - ymin) / g.nrow
    longDist = (xmax - xmin) / nCx
    latDist = (ymax - ymin) / nRx
    boxes = []
    for j in range(1, nRx + 1):
        for h in range(1, nCx + 1):
            boxes.append((xmin + ((longDist * (h - 1)) - (long_pix * feather_d)), xmin + ((longDist * h) + (long_pix * feather_d)),
                          (ymin + ((latDist * (j - 1))) - (lat_pix * feather_d)), (ymin + ((latDist * j)) + (lat_pix * feather_d))))
    return boxes


def tiles_create(g: Geom, xy, out_ncol=3, out_nrow=3, feather_d=50):
    """Returns per tile: crop window of the rasters (V73:1205-1208) and the indices of the
    stations inside the tile extent (terra::crop(points, ext): borders inclusive, V73:1240-1243)."""
    boxes = tiles_create_boxes(g, out_ncol, out_nrow, feather_d)
    wins, sel = [], []
    for b in boxes:
        wins.append(crop_window(g, b))
        inside = (xy[:, 0] >= b[0]) & (xy[:, 0] <= b[1]) & (xy[:, 1] >= b[2]) & (xy[:, 1] <= b[3])
        sel.append(np.flatnonzero(inside))
    return boxes, wins, sel


def tiles_merge(g: Geom, wins, tiles, in_ncol, in_nrow):
    """machisplin.tiles.merge (V73:1392-1548): tiles[h] covers window wins[h] of the full grid."""
    layers = [extend_full(g, wins[h], tiles[h]) for h in range(len(tiles))]
    rast_mosaic_in = mosaic_mean(layers[::-1])  # ig prepends
    return feather_and_merge(g, in_nrow, in_ncol, wins, tiles, rast_mosaic_in, merge_mode=True)
