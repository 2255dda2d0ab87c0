"""A second, independent statement of what the raster stage computes, used only to cross-check the CPU oracle.

Written from the OpenGL ES 3.0 rules that the reference's GL state selects (js/render/backends/raster.js:100-102,
276-279,355: depth test LESS against a cleared 1.0, back faces culled with counter-clockwise front faces, viewport
= the whole grid, one sample at each cell centre) and from the text of its shaders (raster_shader.js:17-22,42-62),
NOT from oracle/cpu_raster.c.  It deliberately takes the textbook route where the oracle takes another:

  * everything in binary64 (the oracle fixes binary32 / binary64 per operation);
  * an explicit perspective divide to normalised device coordinates and then to window coordinates (the oracle
    never divides: it works on homogeneous edge functions);
  * Sutherland-Hodgman clipping of the triangle against the near plane into a polygon, which is then fanned (the
    oracle has its own one-or-two-triangle case analysis);
  * barycentric coordinates from window-space areas, depth interpolated linearly in the window, varyings
    perspective-corrected through 1/w.

The two can therefore only differ where rounding decides: cells whose centre lies within a hair of an edge, and
near-equal depths.  render() reports those cells in `ambiguous` so that a comparison can leave them out.
"""
import numpy as np

EDGE_TOL = 2e-3      # cells: window coordinates of binary32 clip positions are good to about 1e-4 of a cell
DEPTH_TOL = 2e-6


def _clip_near(poly):
    """Sutherland-Hodgman against z_c + w_c >= 0.  poly: list of (clip[4], attr[9])."""
    out = []
    n = len(poly)
    for k in range(n):
        c0, a0 = poly[k]
        c1, a1 = poly[(k + 1) % n]
        d0, d1 = c0[2] + c0[3], c1[2] + c1[3]
        if d0 >= 0:
            out.append((c0, a0))
        if (d0 >= 0) != (d1 >= 0):
            t = d0 / (d0 - d1)
            out.append((c0 + (c1 - c0) * t, a0 + (a1 - a0) * t))
    return out


def render(vbo9, idx, proj, view, lights, cols, rows, model=None):
    """-> dict(rgba u8 [rows, cols, 4], depth f64 [rows, cols], covered bool, ambiguous bool), row 0 = top."""
    vbo = np.asarray(vbo9, dtype=np.float64).reshape(-1, 9)
    idx = np.asarray(idx, dtype=np.int64).reshape(-1, 3)
    P = np.asarray(proj, dtype=np.float64).reshape(4, 4).T      # column-major arrays -> matrices
    V = np.asarray(view, dtype=np.float64).reshape(4, 4).T
    PV = P @ V
    if model is not None:
        # a uModel uniform: clip = P V M p, world position = M p, normal = mat3(M) n
        M = np.asarray(model, dtype=np.float64).reshape(4, 4).T
        vbo = vbo.copy()
        vbo[:, 0:3] = vbo[:, 0:3] @ M[:3, :3].T + M[:3, 3]
        vbo[:, 3:6] = vbo[:, 3:6] @ M[:3, :3].T
    nv = vbo.shape[0]
    depth = np.ones((rows, cols))                                # GL window rows: j = 0 is the bottom
    second = np.ones((rows, cols))
    attr = np.zeros((rows, cols, 9))
    covered = np.zeros((rows, cols), dtype=bool)
    ambiguous = np.zeros((rows, cols), dtype=bool)
    for tri in idx:
        if np.any(tri < 0) or np.any(tri >= nv):
            continue
        a = vbo[tri]                                             # [3, 9]
        if not np.all(np.isfinite(a[:, :3])):
            continue
        c = (PV @ np.concatenate([a[:, :3], np.ones((3, 1))], axis=1).T).T
        poly = _clip_near([(c[k], a[k]) for k in range(3)])
        if len(poly) < 3:
            continue
        for f in range(1, len(poly) - 1):
            cc = np.stack([poly[0][0], poly[f][0], poly[f + 1][0]])
            aa = np.stack([poly[0][1], poly[f][1], poly[f + 1][1]])
            w = cc[:, 3]
            if np.any(w <= 0) or not np.all(np.isfinite(cc)):
                continue
            ndc = cc[:, :3] / w[:, None]
            xw = (ndc[:, 0] + 1.0) * cols / 2.0
            yw = (ndc[:, 1] + 1.0) * rows / 2.0
            zw = (ndc[:, 2] + 1.0) / 2.0
            area2 = (xw[1] - xw[0]) * (yw[2] - yw[0]) - (xw[2] - xw[0]) * (yw[1] - yw[0])
            if not area2 > 0:                                    # back face or zero area
                continue
            i0 = max(0, int(np.floor(xw.min() - 0.5 - EDGE_TOL)))
            i1 = min(cols - 1, int(np.ceil(xw.max() - 0.5 + EDGE_TOL)))
            j0 = max(0, int(np.floor(yw.min() - 0.5 - EDGE_TOL)))
            j1 = min(rows - 1, int(np.ceil(yw.max() - 0.5 + EDGE_TOL)))
            if i0 > i1 or j0 > j1:
                continue
            px = np.arange(i0, i1 + 1)[None, :] + 0.5
            py = np.arange(j0, j1 + 1)[:, None] + 0.5
            lam, dist = [], []
            for k in range(3):
                p, q = (k + 1) % 3, (k + 2) % 3                  # the edge opposite vertex k
                e = (xw[p] - px) * (yw[q] - py) - (xw[q] - px) * (yw[p] - py)
                lam.append(e / area2)
                dist.append(e / max(np.hypot(xw[q] - xw[p], yw[q] - yw[p]), 1e-300))
            dmin = np.minimum(np.minimum(dist[0], dist[1]), dist[2])
            inside = dmin >= 0
            z = lam[0] * zw[0] + lam[1] * zw[1] + lam[2] * zw[2]
            sub_depth = depth[j0:j1 + 1, i0:i1 + 1]
            sub_second = second[j0:j1 + 1, i0:i1 + 1]
            # a cell centre within EDGE_TOL of this triangle's boundary: rounding may decide either way - it matters
            # whenever this triangle would be (or come close to being) the visible one there
            near_edge = (np.abs(dmin) < EDGE_TOL) & (z < sub_depth + DEPTH_TOL) & (z >= -DEPTH_TOL) & (z < 1 + DEPTH_TOL)
            ambiguous[j0:j1 + 1, i0:i1 + 1] |= near_edge
            ok = inside & (z >= 0) & (z < 1)
            win = ok & (z < sub_depth)
            lose = ok & ~win
            sub_second[win] = sub_depth[win]
            sub_second[lose] = np.minimum(sub_second[lose], z[lose])
            if not win.any():
                continue
            sub_depth[win] = z[win]
            covered[j0:j1 + 1, i0:i1 + 1] |= win
            den = lam[0] / w[0] + lam[1] / w[1] + lam[2] / w[2]
            sub_attr = attr[j0:j1 + 1, i0:i1 + 1]
            for q in range(9):
                v = (lam[0] * aa[0, q] / w[0] + lam[1] * aa[1, q] / w[1] + lam[2] * aa[2, q] / w[2]) / den
                sub_attr[..., q][win] = v[win]
    ambiguous |= covered & (second - depth < DEPTH_TOL)
    # raster_shader.js:42-62
    pos, nrm, col = attr[..., 0:3], attr[..., 3:6], attr[..., 6:9]
    ln = np.sqrt(np.sum(nrm * nrm, axis=-1, keepdims=True))
    N = np.divide(nrm, ln, out=np.zeros_like(nrm), where=ln > 0)
    amb = np.asarray(lights["ambient"], dtype=np.float64)
    ldir = np.asarray(lights["dir"], dtype=np.float64)
    lcol = np.asarray(lights["dir_color"], dtype=np.float64)
    out = col * amb
    out = out + col * lcol * np.maximum(np.sum(N * -ldir, axis=-1, keepdims=True), 0.0)
    plpos = np.asarray(lights.get("pl_pos", np.zeros((0, 3))), dtype=np.float64).reshape(-1, 3)
    plcol = np.asarray(lights.get("pl_col", np.zeros((0, 3))), dtype=np.float64).reshape(-1, 3)
    for k in range(min(plpos.shape[0], 8)):
        Lv = plpos[k] - pos
        d2 = np.maximum(np.sum(Lv * Lv, axis=-1, keepdims=True), 1e-4)
        L = Lv / np.sqrt(d2)
        ndl = np.maximum(np.sum(N * L, axis=-1, keepdims=True), 0.0)
        out = out + col * plcol[k] * (ndl * (1.0 / (1.0 + d2 * 0.05)))
    byte = np.floor(np.clip(out, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    rgba = np.zeros((rows, cols, 4), dtype=np.uint8)
    rgba[..., 3] = 255
    rgba[..., :3][covered] = byte[covered]
    # colour bytes that sit on a rounding boundary may legitimately come out one apart: report how close they are
    frac = np.clip(out, 0.0, 1.0) * 255.0 + 0.5
    flip = slice(None, None, -1)                                 # output row 0 = top of the image
    return {"rgba": rgba[flip], "depth": depth[flip], "covered": covered[flip], "ambiguous": ambiguous[flip],
            "byte_margin": np.min(np.abs(frac - np.round(frac)), axis=-1)[flip]}
