"""CPU: host-side logic of the library's multi-device drivers (csrc/multi.hip) -- the row-band plan."""
import numpy as np
import pytest

from machisplin_amd import multi, sharded


@pytest.mark.parametrize("nrow", [16, 100, 333, 1500, 10000, 20000])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 8, 16])
@pytest.mark.parametrize("share", [None, 0.0, 0.03, 0.1, 0.5])
def test_row_band_plan_covers_the_grid_in_aligned_chunks(nrow, n, share):
    bands, band, lead = multi.plan_row_bands(nrow, n, share)
    assert len(bands) == n and bands[0][0] == 0 and max(b for _, b in bands) == nrow
    rows = np.zeros(nrow, dtype=int)
    for a, b in bands:
        assert 0 <= a <= b <= nrow
        rows[a:b] += 1
    assert (rows == 1).all()                                              # every row exactly once
    if n > 1 and -(-nrow // n) >= 16:
        assert all(a % 16 == 0 for a, b in bands if b > a)               # whole 16 x 16 tiles of gbm's coherent kernel per band
    # the in-place all-gather's layout: chunk k = `band` rows, slot 0's rows at the END of its chunk -> chunk k starts at grid row k band - lead
    assert bands[0][1] - bands[0][0] <= band and lead == (band - (bands[0][1] - bands[0][0]) if n > 1 else 0)
    for k, (a, b) in enumerate(bands):
        if b > a:
            assert b - a <= band and a == (0 if k == 0 else k * band - lead)


def test_row_band_plan_matches_the_torch_driver():
    """csrc/multi.hip plan_bands and sharded.row_bands are the same rule (one process per GPU / one process, N slots)."""
    for nrow, n, share in ((10000, 8, 0.04), (10000, 8, None), (333, 4, 0.12), (100, 4, 0.1), (20000, 8, 0.0)):
        bands, band, lead = multi.plan_row_bands(nrow, n, share)
        tb, tbands = sharded.row_bands(nrow, n, share)
        assert band == tb and bands == [(int(a), int(b)) for a, b in tbands], (nrow, n, share)
