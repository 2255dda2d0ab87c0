// geometry.js - mesh generators (JavaScript twin of ascii-renderer_amd/geometry.py).
//
// The reference ships an empty js/geometry.js, so these are this project's own generators.  They
// emit the interchange shape SceneBuilder.addMesh accepts (js/render/scene_api.js:169-192):
// `positions` (3 floats per vertex) and `indices` (3 per triangle, counter-clockwise seen from
// outside) as typed arrays.  Both language twins draw from the same mulberry32 stream in the same
// order with the same literal constants and produce bit-identical binary32 meshes.
export const TAN40 = 0.8390996311772799;   // tan(40 deg): half of the fixed 80 deg vertical fov

export function mulberry32(a) {
  return function () {
    a |= 0; a = (a + 0x6D2B79F5) | 0;
    let t = Math.imul(a ^ (a >>> 15), 1 | a);
    t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t;
    return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
  };
}

// Column-major model matrix (Float32Array(16)): uniform scale, then rotation by `angle` about `axis`, then
// translation - what a spinning mesh hands to HipRasterBackend.setModelMatrix per frame (same arithmetic as
// geometry.py: spin_matrix).
export function spinMatrix(angle, axis, translate, scale) {
  const a = axis || [0, 1, 0], t = translate || [0, 0, 0], k = scale === undefined ? 1.0 : +scale;
  const L = Math.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  const x = a[0] / L, y = a[1] / L, z = a[2] / L, c = Math.cos(angle), s = Math.sin(angle);
  return new Float32Array([
    (c + x * x * (1 - c)) * k, (y * x * (1 - c) + z * s) * k, (z * x * (1 - c) - y * s) * k, 0,
    (x * y * (1 - c) - z * s) * k, (c + y * y * (1 - c)) * k, (z * y * (1 - c) + x * s) * k, 0,
    (x * z * (1 - c) + y * s) * k, (y * z * (1 - c) - x * s) * k, (c + z * z * (1 - c)) * k, 0,
    t[0], t[1], t[2], 1]);
}

export function cube(half) {
  const h = half === undefined ? 0.5 : +half;
  const positions = new Float32Array([-h, -h, -h, h, -h, -h, h, h, -h, -h, h, -h, -h, -h, h, h, -h, h, h, h, h, -h, h, h]);
  const indices = new Uint32Array([4, 5, 6, 4, 6, 7, 1, 0, 3, 1, 3, 2, 5, 1, 2, 5, 2, 6, 0, 4, 7, 0, 7, 3, 7, 6, 2, 7, 2, 3, 0, 1, 5, 0, 5, 4]);
  return { positions, indices };
}

export function icosphere(subdiv, radius) {
  const R = radius === undefined ? 1.0 : +radius;
  const t = (1.0 + Math.sqrt(5.0)) / 2.0, s = Math.sqrt(1.0 + t * t);
  let verts = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]]
    .map((v) => [v[0] / s, v[1] / s, v[2] / s]);
  let faces = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
    [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]];
  for (let level = 0; level < (subdiv | 0); level++) {
    const cache = new Map(), out = [];
    const mid = (i, j) => {
      const lo = Math.min(i, j), hi = Math.max(i, j), key = lo * 4294967296 + hi;
      let k = cache.get(key);
      if (k === undefined) {
        const a = verts[lo], b = verts[hi];
        const m = [(a[0] + b[0]) * 0.5, (a[1] + b[1]) * 0.5, (a[2] + b[2]) * 0.5];
        const L = Math.sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
        verts.push([m[0] / L, m[1] / L, m[2] / L]);
        k = verts.length - 1;
        cache.set(key, k);
      }
      return k;
    };
    for (const f of faces) {
      const ab = mid(f[0], f[1]), bc = mid(f[1], f[2]), ca = mid(f[2], f[0]);
      out.push([f[0], ab, ca], [f[1], bc, ab], [f[2], ca, bc], [ab, bc, ca]);
    }
    faces = out;
  }
  const positions = new Float32Array(verts.length * 3), indices = new Uint32Array(faces.length * 3);
  verts.forEach((v, i) => { positions[3 * i] = v[0] * R; positions[3 * i + 1] = v[1] * R; positions[3 * i + 2] = v[2] * R; });
  faces.forEach((f, i) => { indices[3 * i] = f[0]; indices[3 * i + 1] = f[1]; indices[3 * i + 2] = f[2]; });
  return { positions, indices };
}

// Screen-uniform cloud of small random triangles in front of a camera at the origin looking along
// +X (yaw = pitch = 0; screen right = +Z, up = +Y).  Per triangle, 15 consecutive draws: depth x in
// [d0, d1]; centre y, z uniform over the frustum cross-section at that depth; three vertices =
// centre + x*rho*u with u uniform in [-1,1]^3 and rho = cells * (2*TAN40/rows); then an albedo in
// [0.2, 1]^3.  Winding is as generated (about half the triangles are back faces).
export function randomMesh(seed, nTris, cols, rows, pixelAspect, depthRange, cells) {
  const pa = pixelAspect === undefined ? 1.0 : pixelAspect, d0 = depthRange ? depthRange[0] : 2.0, d1 = depthRange ? depthRange[1] : 60.0;
  const aspect = (cols / rows) * pa, rho = (cells === undefined ? 2.0 : cells) * (2.0 * TAN40 / rows);
  const rnd = mulberry32(seed);
  const positions = new Float32Array(nTris * 9), colors = new Float32Array(nTris * 3), indices = new Uint32Array(nTris * 3);
  for (let t = 0; t < nTris; t++) {
    const x = d0 + (d1 - d0) * rnd();
    const cy = (2.0 * rnd() - 1.0) * x * TAN40;
    const cz = (2.0 * rnd() - 1.0) * x * TAN40 * aspect;
    const k = x * rho;
    for (let v = 0; v < 3; v++) {
      const ox = (2.0 * rnd() - 1.0) * k, oy = (2.0 * rnd() - 1.0) * k, oz = (2.0 * rnd() - 1.0) * k;
      positions[9 * t + 3 * v] = x + ox; positions[9 * t + 3 * v + 1] = cy + oy; positions[9 * t + 3 * v + 2] = cz + oz;
      indices[3 * t + v] = 3 * t + v;
    }
    colors[3 * t] = 0.2 + 0.8 * rnd(); colors[3 * t + 1] = 0.2 + 0.8 * rnd(); colors[3 * t + 2] = 0.2 + 0.8 * rnd();
  }
  return { positions, indices, colors };
}
