"""Scene ingest: seconds from an indexed triangle list on the host to a mesh the renderer can draw.
  host build  : host.mesh_to_vbo (vectorised pushTri) + ar_set_mesh (uploads 120 B per triangle)
  device build: ar_set_mesh_indexed (uploads positions + indices + colours, builds the vertex buffer on the GPU)
Both end with the same bytes on the GPU (tests/test_gpu_parity.py)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
R = ar.Renderer(0)
for n in (1_000_000, 10_000_000):
    P, I, C = ar.geometry.random_mesh(0xA5C11, n, 1920, 1080, 1.0)
    P = np.ascontiguousarray(P, dtype=np.float32); I = np.ascontiguousarray(I, dtype=np.uint32); C = np.ascontiguousarray(C, dtype=np.float32)
    R.set_mesh_indexed(P[:300], I[:100], C[:100])  # warm up
    t0 = time.perf_counter(); vbo, idx = ar.host.mesh_to_vbo(P, I, C); t1 = time.perf_counter(); R.set_mesh(vbo, idx); t2 = time.perf_counter()
    R.set_mesh_indexed(P, I, C); t3 = time.perf_counter()
    print("%9d triangles: host build %.3f s + upload/derive %.3f s = %.3f s;  device build %.3f s  (%.1fx)" %
          (n, t1 - t0, t2 - t1, t2 - t0, t3 - t2, (t2 - t0) / (t3 - t2)), flush=True)
    del vbo, idx
