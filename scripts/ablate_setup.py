"""k_setup duration of an ablation variant: frames with 7 point lights switch the ablation on (the
buffers keep the valid contents of the earlier, normal frames)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
ar.binding.LIB_PATH = os.environ["AR_LIB"]
sc = ar.scenes.random_scene(1000000, 1920, 1080)
R = ar.Renderer(0)
R.set_option("overlap", 0)
ar.scenes.upload(R, sc)
for _ in range(6): R.render_async(1920, 1080)
R.sync()
for npl in (0, 7):
    L = dict(sc["lights"]); L["pl_pos"] = np.ones((npl, 3), np.float32); L["pl_col"] = np.ones((npl, 3), np.float32)
    R.set_lights(L)
    R.set_profiling(True); R.kernel_stats()
    for _ in range(50): R.render_async(1920, 1080)
    st = R.kernel_stats()
    print(os.path.basename(os.environ["AR_LIB"]), "n_pl", npl, {k: round(1e3 * ms / n, 2) for k, (n, ms) in st.items() if n}, flush=True)
