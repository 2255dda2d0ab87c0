"""ar_quantize_static - the entry point that allocates nothing - against ar_quantize, through the C ABI: same
arguments, byte-identical ASCII planes, on two calls in a row (the second on a smaller grid, so that whatever the
first call left in the shared staging array lies beyond what the second one writes), with smoothing on and off, on
a second context of the same GPU, and with its size limit enforced.  (Named so that it is collected last.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _grid(rng, cols, rows):
    lv = rng.choice([0, 40, 90, 140, 250], size=(rows * cols, 1)).astype(np.uint8)
    rgba = np.concatenate([lv, lv // 2 + 10, lv, np.full_like(lv, 255)], axis=1)
    ov = rng.random(rows * cols) < 0.05
    rgba[ov, 3] = rng.integers(0, 256, size=ov.sum())           # UI override cells and the alphas around their window
    return rgba


def test_static_quantize_equals_quantize_on_two_calls_in_a_row(pkg):
    rng = np.random.default_rng(17)
    R = pkg.Renderer(0)
    R2 = pkg.Renderer(0)
    try:
        for mode in ((True, 2, 12), (False, 2, 12), (True, 3, 20)):
            R.set_ascii("@%#*+=-:. ", *mode)
            R2.set_ascii("@%#*+=-:. ", *mode)
            first = True
            for cols, rows in ((1920, 1080), (333, 77), (96, 36), (1, 1)):     # the largest grid it accepts, then smaller ones
                rgba = _grid(rng, cols, rows)
                want = R.quantize(rgba, cols, rows)
                got = R.quantize_static(rgba, cols, rows)
                assert np.array_equal(got, want), (mode, cols, rows, int(np.count_nonzero(got != want)))
                assert len(np.unique(got)) > (3 if cols > 1 else 0)
                if not first:
                    # another context on the same GPU shares the array and gives the same answer
                    assert np.array_equal(R2.quantize_static(rgba, cols, rows), want), (mode, cols, rows)
                first = False
        with pytest.raises(pkg.AsciiRastError, match="exceed"):
            R.quantize_static(np.zeros((1081 * 1920, 4), dtype=np.uint8), 1920, 1081)
        with pytest.raises(pkg.AsciiRastError):
            R._check(R._L.ar_quantize_static(R._h, None, 4, 4, None))
        # the context is still good after the refusals
        rgba = _grid(rng, 64, 16)
        assert np.array_equal(R.quantize_static(rgba, 64, 16), R.quantize(rgba, 64, 16))
    finally:
        R.close()
        R2.close()


def test_static_quantize_leaves_device_memory_alone(pkg):
    """What the entry point exists for: no device allocation on the way.  A fresh context that has never run
    ar_quantize (so the growable staging buffers do not exist) makes calls at several grid sizes and parameter sets; the
    device's free memory, read with hipMemGetInfo before and after, must not move.  (A first context warms the same
    kernels up beforehand: loading a kernel's code is the runtime's allocation, made once per process.)"""
    import torch
    rng = np.random.default_rng(23)
    modes = ((True, 2, 12), (False, 2, 12), (True, 1, 3), (True, 3, 20))
    grids = ((640, 360), (1920, 1080), (97, 31))
    cases = [(m, g, _grid(rng, *g)) for m in modes for g in grids]
    warm = pkg.Renderer(0)
    try:
        for mode, (cols, rows), rgba in cases:
            warm.set_ascii("@%#*+=-:. ", *mode)
            warm.quantize_static(rgba, cols, rows)
        want = {}
        for k, (mode, (cols, rows), rgba) in enumerate(cases):
            warm.set_ascii("@%#*+=-:. ", *mode)
            want[k] = warm.quantize(rgba, cols, rows)
    finally:
        warm.close()
    R = pkg.Renderer(0)
    try:
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        got = {}
        for k, (mode, (cols, rows), rgba) in enumerate(cases):
            R.set_ascii("@%#*+=-:. ", *mode)
            got[k] = R.quantize_static(rgba, cols, rows)
        torch.cuda.synchronize()
        free_after = torch.cuda.mem_get_info(0)[0]
        assert free_after == free_before, "ar_quantize_static changed the device's free memory by %d bytes" % (free_before - free_after)
        for k in got:
            assert np.array_equal(got[k], want[k]), cases[k][:2]
        # the control: the allocating entry point does move it on this context (its staging buffers do not exist yet)
        R.quantize(cases[1][2], 1920, 1080)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(0)[0] < free_before
    finally:
        R.close()
