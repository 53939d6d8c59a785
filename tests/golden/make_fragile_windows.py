#!/usr/bin/env python3
"""Which of the reference's tile windows hinge on one rounding?  

terra::crop snaps an extent with round((e - origin) / res) (oracle/tiles.py::align_near, restated from terra's C++; not
verifiable without R).  The Step-3 boxes (V73:656-681) and the machisplin.tiles.create boxes (V73:1165-1197) are sums of
fractions of the raster extent, so a box edge can land within an ulp of a cell CENTRE -- where round() flips.  For
every box of the G6 grids, every edge is moved by +-1 ulp and by +-1e-9 cell sizes (one edge at a time) and the crop
window is recomputed; a window that moves is FRAGILE: a differently rounded R / terra build may disagree with the oracle
there by one row or column.  Output: tests/golden/fragile_windows.json (checked by tests/test_window_fragility.py; the
fragile boxes are the first thing tests/golden/capture_from_R.R should capture)."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tiles as ot  # noqa: E402

S = (-78.0, -5.0, 1.0 / 1200.0)                      # the synthetic grids' origin and cell size (machisplin_amd.synth.grid)
B = (-77.7439933, -5.8090002, 0.0016666666)          # the bundled rasters' half-resolution geometry (cfg1_extdata.npz)
GRIDS = [("1500x1500", S, 1500, 1500), ("1501x1501", S, 1501, 1501), ("2000x2000", S, 2000, 2000), ("10000x10000", S, 10000, 10000),
         ("bundled 1238x1632", B, 1238, 1632), ("bundled 2476x3264", (B[0], B[1], B[2] / 2), 2476, 3264)]


def probes(box, res):
    """(label, box') for every single-edge perturbation."""
    out = []
    for k in range(4):
        for lab, f in (("+1ulp", lambda v: np.nextafter(v, np.inf)), ("-1ulp", lambda v: np.nextafter(v, -np.inf)),
                       ("+1e-9res", lambda v: v + 1e-9 * res), ("-1e-9res", lambda v: v - 1e-9 * res)):
            b = list(box)
            b[k] = float(f(b[k]))
            out.append(("xmin xmax ymin ymax".split()[k] + lab, tuple(b)))
    return out


def fragile(g, box, parent=None):
    base = ot.crop_window(parent or g, box)
    moved = []
    for lab, b in probes(box, g.xres):
        w = ot.crop_window(parent or g, b)
        if w != base:
            moved.append({"probe": lab, "window": list(w) if w else None})
    return base, moved


def main():
    res = {}
    for name, (xmin, ymax, r), nrow, ncol in GRIDS:
        g = ot.Geom(xmin, ymax, r, r, nrow, ncol)
        entry = {"geom": [xmin, ymax, r, r, nrow, ncol], "boxes": 0, "probes": 0, "fragile": []}
        nRx, nCx, fit, keep = ot.step3_tile_boxes(g, 1500)
        for m, (b, d) in enumerate(zip(fit, keep)):
            wf, mv = fragile(g, b)
            entry["boxes"] += 2; entry["probes"] += 32
            if mv:
                entry["fragile"].append({"kind": "step3 fit box (V73:699)", "tile": m, "box": list(b), "window": list(wf), "moves": mv})
            gf = ot.window_geom(g, wf)
            wk, mv = fragile(g, d, parent=gf)
            if mv:
                entry["fragile"].append({"kind": "step3 keep box (V73:728)", "tile": m, "box": list(d), "window": list(wk), "moves": mv})
        for (oc, orow) in ((3, 3), (2, 2), (2, 3)):
            for m, b in enumerate(ot.tiles_create_boxes(g, oc, orow, 50)):
                w, mv = fragile(g, b)
                entry["boxes"] += 1; entry["probes"] += 16
                if mv:
                    entry["fragile"].append({"kind": "tiles.create %dx%d box (V73:1205)" % (oc, orow), "tile": m, "box": list(b),
                                             "window": list(w), "moves": mv})
        res[name] = entry
    return res


if __name__ == "__main__":
    out = main()
    path = os.path.join(ROOT, "tests", "golden", "fragile_windows.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        print("%-20s boxes %4d  probes %5d  fragile boxes %3d" % (k, v["boxes"], v["probes"], len(v["fragile"])))
