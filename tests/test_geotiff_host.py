"""CPU: the GeoTIFF reader/writer (host functions of the C ABI, no GPU) against libtiff via Pillow:
the reference's own bundled rasters (tiled, deflate, INT2S, NoData -32768, four overview levels) when
/root/reference is present, and Pillow-written fixtures for strips / LZW / deflate / float32."""
import os

import numpy as np
import pytest
from PIL import Image

import machisplin_amd as mhs
from machisplin_amd import io as mio

REF = "/root/reference/inst/extdata"
Image.MAX_IMAGE_PIXELS = None


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "TWI.tif.ovr")), reason="reference data not mounted")
@pytest.mark.parametrize("name", ["TWI.tif.ovr", "slope.tif.ovr"])
def test_bundled_overviews_decode_like_libtiff(name):
    path = os.path.join(REF, name)
    info = mio.tiff_info(path)
    assert (info["width"], info["height"], info["bits"], info["sample_format"]) == (1632, 1238, 16, 2)
    im = Image.open(path)
    assert info["compression"] == 8 and info["n_ifd"] == 4 and info["dtype"] == mhs._lib.I16
    assert info["nodata"] == float(im.tag_v2[42113])  # GDAL_NODATA: -32768 for TWI (slope.tif.ovr carries 65536)
    for ifd in range(4):
        im.seek(ifd)
        want = np.array(im)
        got = mio.read_host(path, ifd)
        assert got.dtype == np.int16 and got.shape == want.shape
        assert np.array_equal(got.astype(np.int64), want.astype(np.int64))
    # georeference from the .tfw sidecar of the base raster (inst/extdata/TWI.tfw, alt.tfw)
    g = mio.geometry_of(path)
    w = mio.world_file(os.path.join(REF, name.split(".")[0] + ".tfw"))
    assert w["xres"] == pytest.approx(0.0008333333) and g.xres == pytest.approx(2 * w["xres"])
    assert g.xmin == pytest.approx(-77.7435765934 - 0.5 * 0.0008333333) and (g.nrow, g.ncol) == (1238, 1632)


@pytest.mark.parametrize("comp", ["raw", "tiff_lzw", "tiff_adobe_deflate"])
@pytest.mark.parametrize("kind", ["i16", "f32", "u8"])
def test_pillow_written_strips(tmp_path, comp, kind):
    rng = np.random.default_rng(5)
    if kind == "i16":
        a = (np.cumsum(rng.integers(-3, 4, (217, 333)), axis=1) * 7 - 2000).astype(np.int16)
        a[rng.random(a.shape) < 0.01] = -32768
        im = Image.fromarray(a.view(np.uint16))
        extra = {339: 2}  # SampleFormat = signed
    elif kind == "f32":
        a = rng.standard_normal((150, 97)).astype(np.float32)
        im, extra = Image.fromarray(a), {}
    else:
        a = rng.integers(0, 255, (64, 1000)).astype(np.uint8)
        im, extra = Image.fromarray(a), {}
    p = str(tmp_path / f"t_{kind}_{comp}.tif")
    im.save(p, compression=comp, tiffinfo=extra)
    got = mio.read_host(p)
    assert got.dtype == a.dtype and np.array_equal(got, a)


def test_writer_roundtrip_and_tags(tmp_path):
    rng = np.random.default_rng(1)
    g = mhs.Geometry(-78.0, -5.0, 1 / 1200, 1 / 1200, 431, 517)
    a = rng.standard_normal((431, 517)) * 100
    a[rng.random(a.shape) < 0.02] = np.nan
    for compress in (True, False):
        p = str(tmp_path / f"out_{compress}.tif")
        mio.write_geotiff(p, g, a, nodata=-9999.0, compress=compress)
        im = Image.open(p)
        back = np.array(im)
        want = np.where(np.isnan(a), -9999.0, a).astype(np.float32)
        assert back.dtype == np.float32 and np.array_equal(back, want)
        tags = im.tag_v2
        one = lambda v: v[0] if isinstance(v, tuple) else v
        assert one(tags[339]) == 3 and one(tags[258]) == 32 and one(tags[259]) == (8 if compress else 1)
        assert tuple(tags[33550])[:2] == pytest.approx((1 / 1200, 1 / 1200))
        assert tuple(tags[33922]) == pytest.approx((0, 0, 0, -78.0, -5.0, 0))
        assert float(tags[42113]) == -9999.0
        assert tuple(tags[34735])[-4:] == (2048, 0, 1, 4326)
        # and our own reader agrees, including the georeference
        assert np.array_equal(mio.read_host(p), want)
        gg = mio.geometry_of(p)
        assert (gg.xmin, gg.ymax, gg.nrow, gg.ncol) == (-78.0, -5.0, 431, 517) and gg.xres == pytest.approx(1 / 1200)
    # NaN stays NaN when no nodata value is given
    p = str(tmp_path / "nan.tif")
    mio.write_geotiff(p, g, a)
    assert np.array_equal(np.isnan(np.array(Image.open(p))), np.isnan(a))


def test_errors(tmp_path):
    p = str(tmp_path / "junk.tif")
    open(p, "wb").write(b"not a tiff at all")
    with pytest.raises(mhs.MhsError):
        mio.tiff_info(p)
    with pytest.raises(mhs.MhsError):
        mio.tiff_info(str(tmp_path / "missing.tif"))
    rgb = str(tmp_path / "rgb.tif")
    Image.fromarray(np.zeros((4, 4, 3), dtype=np.uint8)).save(rgb)
    with pytest.raises(mhs.MhsError):  # three samples per pixel
        mio.tiff_info(rgb)


def _patch_tag(raw: bytearray, tag: int, value: int):
    """overwrite the inline value of a classic little-endian TIFF directory entry"""
    import struct
    ifd = struct.unpack_from("<I", raw, 4)[0]
    n = struct.unpack_from("<H", raw, ifd)[0]
    for e in range(n):
        p = ifd + 2 + 12 * e
        t, typ = struct.unpack_from("<HH", raw, p)
        if t == tag:
            if typ == 3:
                struct.pack_into("<H", raw, p + 8, value & 0xFFFF)
            else:
                struct.pack_into("<I", raw, p + 8, value & 0xFFFFFFFF)
            return
    raise KeyError(tag)


@pytest.mark.parametrize("tag,value", [(256, 0x7FFFFFF0), (257, 0x7FFFFFF0), (279, 0x7FFFFFFF), (273, 0x7FFFFF00),
                                       (278, 0)])
def test_crafted_header_fields_are_rejected_not_trusted(tmp_path, tag, value):
    """width / height / StripByteCounts / StripOffsets that do not fit the file: an error code, never an
    over-read, an attacker-sized allocation or an exception out of the C ABI (geotiff.hip)."""
    a = np.arange(64 * 48, dtype=np.int16).reshape(48, 64)
    p = str(tmp_path / "ok.tif")
    Image.fromarray(a.view(np.uint16)).save(p, compression="raw", tiffinfo={339: 2, 278: 48})
    raw = bytearray(open(p, "rb").read())
    assert np.array_equal(mio.read_host(p), a)
    _patch_tag(raw, tag, value)
    q = str(tmp_path / "bad.tif")
    open(q, "wb").write(bytes(raw))
    if tag == 278:   # RowsPerStrip = 0 means "one strip" and stays readable
        assert np.array_equal(mio.read_host(q), a)
        return
    with pytest.raises(mhs.MhsError):
        mio.read_host(q)


def test_truncated_file_and_huge_tag_count(tmp_path):
    import struct
    a = np.arange(64 * 48, dtype=np.int16).reshape(48, 64)
    p = str(tmp_path / "ok.tif")
    Image.fromarray(a.view(np.uint16)).save(p, compression="tiff_lzw", tiffinfo={339: 2})
    raw = bytearray(open(p, "rb").read())
    q = str(tmp_path / "cut.tif")
    open(q, "wb").write(bytes(raw[: len(raw) // 2]))
    with pytest.raises(mhs.MhsError):
        mio.read_host(q)
    # a directory entry whose count field claims 4e9 values
    ifd = struct.unpack_from("<I", raw, 4)[0]
    struct.pack_into("<I", raw, ifd + 2 + 4, 0xF0000000)
    open(q, "wb").write(bytes(raw))
    with pytest.raises(mhs.MhsError):
        mio.tiff_info(q)
