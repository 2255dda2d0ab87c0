"""RGBA grid -> ASCII codes through ar_quantize (AsciiPass.drawFromBuffer's index half): blocking calls per second with
pageable and page-locked host arrays, plus the GPU time of the kernels under HIP events.  AR_LIB selects a library variant.
python scripts/quantize_rate.py"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
if os.environ.get("AR_LIB"):
    ar.binding.LIB_PATH = os.environ["AR_LIB"]
R = ar.Renderer(0)
rng = np.random.default_rng(12345)
print("library:", ar.binding.LIB_PATH)
for cols, rows in ((1920, 1080), (3840, 2160)):
    for override_density in (0.0, 0.05):
        src = rng.integers(0, 256, size=(rows, cols, 4), dtype=np.uint8)
        src[..., 3] = 255
        ov = rng.random((rows, cols)) < override_density
        src[ov, 3] = rng.integers(2, 255, size=int(ov.sum()))
        for pinned in (False, True):
            if pinned:
                grid = R.host_array((rows, cols, 4), np.uint8); grid[...] = src
                out = R.host_array((rows, cols), np.uint8)
            else:
                grid, out = src.copy(), np.zeros((rows, cols), np.uint8)
            R.quantize_into(grid, cols, rows, out)
            n = 30
            t0 = time.perf_counter()
            for _ in range(n):
                R.quantize_into(grid, cols, rows, out)
            dt = (time.perf_counter() - t0) / n
            R.set_profiling(True); R.kernel_stats()
            for _ in range(10):
                R.quantize_into(grid, cols, rows, out)
            st = R.kernel_stats(); R.set_profiling(False)
            k = st.get("k_ascii", (0, 0.0))
            print("%4dx%-4d overrides %3.0f%%  %-9s %8.1f calls/s  (%.3f ms per call; smoothing kernels %.1f us)"
                  % (cols, rows, 100 * override_density, "pinned" if pinned else "pageable", 1.0 / dt, 1e3 * dt, 1e3 * k[1] / max(k[0], 1)))
R.close()
