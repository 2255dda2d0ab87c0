"""Throughput of the glyph-compositing kernel (k_present): bytes of image written per second."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
atlas = np.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "atlas_8x16.bin"), dtype=np.uint8)
R = ar.Renderer(0)
rng = np.random.default_rng(0)
for cols, rows in ((96, 36), (480, 270), (960, 540)):
    rgba = rng.integers(0, 256, size=(rows, cols, 4), dtype=np.uint8); rgba[:, :, 3] = 255
    out = R.host_array((rows * 16, cols * 8, 4), np.uint8)
    for _ in range(2): R.present(rgba, cols, rows, atlas, 8, 16, 1.2, out=out)
    R.set_profiling(True); R.kernel_stats()
    for _ in range(10): R.present(rgba, cols, rows, atlas, 8, 16, 1.2, out=out)
    n, ms = R.kernel_stats()["k_present"]
    R.set_profiling(False)
    b = cols * 8 * rows * 16 * 4
    print("%4dx%-4d cells -> %5dx%-5d px: k_present %.1f us, %.0f GB/s written (%.1f%% of 8 TB/s)" % (
        cols, rows, cols * 8, rows * 16, 1e3 * ms / n, b / (1e-3 * ms / n) / 1e9, 100 * b / (1e-3 * ms / n) / 8e12), flush=True)
