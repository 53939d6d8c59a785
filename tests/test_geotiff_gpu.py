"""GPU: GeoTIFF -> device planes (decode on host threads, band-wise overlapped upload) and device
plane -> FLT4S GeoTIFF, against Pillow/libtiff; then the planes feed the predictors unchanged."""
import numpy as np
import pytest
from PIL import Image

from machisplin_amd import io as mio

pytestmark = pytest.mark.gpu
Image.MAX_IMAGE_PIXELS = None


def test_read_dev_multi_band_int16_lzw(hip, tmp_path):
    rng = np.random.default_rng(0)
    # 6000 x 6000 INT2S = 72 MB: three upload bands of ~32 MB, double-buffered
    a = (np.cumsum(rng.integers(-2, 3, (6000, 6000), dtype=np.int16), axis=1, dtype=np.int64) % 3000 - 500).astype(np.int16)
    a[::97, ::89] = -32768
    p = str(tmp_path / "alt.tif")
    Image.fromarray(a.view(np.uint16)).save(p, compression="tiff_lzw", tiffinfo={339: 2, 42113: "-32768"})
    open(str(tmp_path / "alt.tfw"), "w").write("0.0008333333\n0\n0\n-0.0008333333\n-77.7435765934\n-5.8094167820\n")
    geom, plane, nodata = mio.read_raster(p)
    assert plane.dtype.__str__() == "torch.int16" and nodata == -32768
    assert np.array_equal(plane.cpu().numpy(), a)
    assert geom is not None and geom.xres == pytest.approx(0.0008333333) and (geom.nrow, geom.ncol) == (6000, 6000)
    assert geom.xmin == pytest.approx(-77.7435765934 - 0.5 * 0.0008333333)


def test_stack_from_files_feeds_predict_and_write_back(hip, tmp_path):
    import torch
    rng = np.random.default_rng(1)
    g = hip.Geometry(-78.0, -5.0, 1 / 1200, 1 / 1200, 300, 420)
    paths = []
    for k in range(3):
        a = rng.integers(-200, 4000, (300, 420)).astype(np.int16)
        a[rng.random(a.shape) < 0.01] = -32768
        p = str(tmp_path / f"cov{k}.tif")
        Image.fromarray(a.view(np.uint16)).save(p, compression="tiff_adobe_deflate", tiffinfo={339: 2, 42113: "-32768"})
        paths.append(p)
    stack = mio.read_stack(paths, geom=g)
    assert stack.n_layers == 3 and stack.nodata == -32768
    m = hip.models.Gam([1.0, 0.5, -0.25, 0.125, 2.0, 3.0])
    pred = hip.predict(stack, m)
    host = np.stack([np.array(Image.open(p)).astype(np.float64) for p in paths])
    host[host == -32768] = np.nan
    x, y = g.x_from_col(np.arange(420)), g.y_from_row(np.arange(300))
    want = 1.0 + 0.5 * host[0] - 0.25 * host[1] + 0.125 * host[2] + 2.0 * x[None, :] + 3.0 * y[:, None]
    assert np.allclose(pred.cpu().numpy(), want, rtol=1e-14, atol=0, equal_nan=True)
    out = str(tmp_path / "bio_1.tif")
    mio.write_geotiff(out, g, pred, nodata=-3.4e38)
    back = np.array(Image.open(out))
    ref = np.where(np.isnan(want), np.float32(-3.4e38), want.astype(np.float32))
    assert back.dtype == np.float32 and np.array_equal(back, ref)
    gg = mio.geometry_of(out)
    assert (gg.xmin, gg.ymax, gg.nrow, gg.ncol) == (-78.0, -5.0, 300, 420)
