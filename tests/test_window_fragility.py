"""How much of the tile bookkeeping hinges on a single rounding?  terra::crop's snapping is
restated from terra's C++ and cannot be checked against R here, so every box of the G6 grids is probed:
each edge moved by +-1 ulp and +-1e-9 cell sizes, crop window recomputed (tests/golden/make_fragile_windows.py).  The
committed table tests/golden/fragile_windows.json is what DESIGN.md section 5 quotes and what capture_from_R.R targets
first; this test keeps it in step with the oracle and pins the headline: NO Step-3 window (V73:656-728) of any BASELINE
geometry is fragile; the only fragile boxes are machisplin.tiles.create's on the odd 1501 x 1501 grid, whose box edges
fall exactly on cell centres."""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("make_fragile_windows", os.path.join(HERE, "golden", "make_fragile_windows.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.main()


def test_fragile_window_table_is_current_and_step3_is_robust():
    got = json.loads(json.dumps(_gen()))
    with open(os.path.join(HERE, "golden", "fragile_windows.json")) as f:
        want = json.load(f)
    assert got == want, "oracle/tiles.py changed: regenerate with python tests/golden/make_fragile_windows.py"
    for name, e in got.items():
        kinds = {f["kind"].split(" (")[0] for f in e["fragile"]}
        assert not any(k.startswith("step3") for k in kinds), (name, kinds)
        if name != "1501x1501":
            assert e["fragile"] == [], name
    # every fragile window differs from its probes by exactly one row or column
    for f in got["1501x1501"]["fragile"]:
        for mv in f["moves"]:
            assert sum(abs(a - b) for a, b in zip(mv["window"], f["window"])) == 1


def test_hip_windows_agree_with_the_oracle_on_the_fragile_boxes():
    """The C ABI's own crop (tiles.hip, host code) lands on the oracle's side of every fragile rounding."""
    import machisplin_amd as mhs
    from oracle import tiles as ot
    with open(os.path.join(HERE, "golden", "fragile_windows.json")) as f:
        tab = json.load(f)
    n = 0
    for name, e in tab.items():
        xmin, ymax, xres, yres, nrow, ncol = e["geom"]
        g = mhs.Geometry(xmin, ymax, xres, yres, int(nrow), int(ncol))
        og = ot.Geom(xmin, ymax, xres, yres, int(nrow), int(ncol))
        for fr in e["fragile"]:
            if fr["kind"].startswith("tiles.create"):
                assert tuple(mhs.tiles.crop_window(g, tuple(fr["box"]))) == tuple(fr["window"]) == ot.crop_window(og, tuple(fr["box"]))
                n += 1
    assert n == 10
