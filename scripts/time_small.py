"""k_tile duration on the small reference scenes for a library variant (AR_LIB=path): icosphere C2, cube C1."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
if os.environ.get("AR_LIB"):
    ar.binding.LIB_PATH = os.environ["AR_LIB"]
R = ar.Renderer(0)
R.set_option("overlap", 0)
out = {}
for name, sc in (("ico", ar.scenes.icosphere_scene()), ("ico_s5", ar.scenes.icosphere_scene(subdiv=5)), ("cube", ar.scenes.cube_scene())):
    ar.scenes.upload(R, sc)
    for _ in range(20): R.render_async(sc["cols"], sc["rows"])
    R.sync(); R.set_profiling(True); R.kernel_stats()
    for _ in range(200): R.render_async(sc["cols"], sc["rows"])
    st = R.kernel_stats(); R.set_profiling(False)
    out[name] = round(1e3 * st["k_tile"][1] / st["k_tile"][0], 1)
print(os.path.basename(os.environ.get("AR_LIB", "default")), "k_tile us:", out, flush=True)
