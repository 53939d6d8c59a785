"""Regenerates tests/golden/cfg1_extdata.npz.  Run from the repo root IN THE BUILD CONTAINER:

    python tests/golden/make_golden_cfg1.py

BASELINE.json configs[0] ("bundled sampling.csv + alt/slope/TWI extdata rasters") as DATA: the reference's
station table /root/reference/data-raw/sampling.csv (813 rows: long, lat, bio_1, bio_12) and level 0 of the
two rasters that are present in the checkout, inst/extdata/TWI.tif.ovr and slope.tif.ovr (1632 x 1238 INT2S,
1/600 degree; alt.tif and the full-resolution rasters are missing blobs, .MISSING_LARGE_BLOBS:6-9), decoded with
libtiff via Pillow, plus the georeference from TWI.tfw.  No reference code is involved.  tests/test_cfg1_gpu.py
reads this file on the GPU box, where /root/reference does not exist.
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
Image.MAX_IMAGE_PIXELS = None


def main():
    tab = np.loadtxt(os.path.join(REF, "data-raw", "sampling.csv"), delimiter=",", skiprows=1)
    planes = {}
    for name in ("TWI", "slope"):
        a = np.array(Image.open(os.path.join(REF, "inst", "extdata", name + ".tif.ovr")))
        assert a.shape == (1238, 1632) and a.min() >= -32768 and a.max() <= 32767
        planes[name] = a.astype(np.int16)
    w = [float(v) for v in open(os.path.join(REF, "inst", "extdata", "TWI.tfw")).read().split()]
    # world file: x scale, 0, 0, -y scale, x and y of the CENTRE of the north-west cell of the base raster; the
    # overview level is 2x coarser with the same north-west corner
    xres, yres = w[0], -w[3]
    xmin, ymax = w[4] - 0.5 * xres, w[5] + 0.5 * yres
    np.savez_compressed(os.path.join(HERE, "cfg1_extdata.npz"), sampling=tab, TWI=planes["TWI"], slope=planes["slope"],
                        geom=np.array([xmin, ymax, 2 * xres, 2 * yres, 1238, 1632], dtype=np.float64),
                        nodata=np.float64(-32768))
    print("wrote cfg1_extdata.npz", os.path.getsize(os.path.join(HERE, "cfg1_extdata.npz")), "bytes")


if __name__ == "__main__":
    main()
