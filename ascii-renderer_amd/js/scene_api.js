// scene_api.js - scene authoring: SceneBuilder -> plain "unified" scene object (schema version 2).
//
// Same public surface and output shape as the reference's js/render/scene_api.js (MaterialIds :11-19,
// SceneBuilder :52-258 with caps 64 spheres / 4096 tris / 4096 quads, createSceneBuilder :261,
// fromObject :266-319), so scenes written against it keep working.  Added for large meshes:
// toRasterMesh(), which hands the backend typed arrays instead of one object per triangle.
export const MaterialIds = Object.freeze({ LIGHT: 0, WHITE: 1, GREEN: 2, RED: 3, GLASS: 6, MIRROR: 7 });

const DEFAULT_MAT_ID = MaterialIds.WHITE;
const isFinite3 = (a) => Array.isArray(a) && a.length === 3 && a.every(Number.isFinite);
const copy3 = (v) => [+v[0], +v[1], +v[2]];
const u16 = (x) => { const n = x | 0; return n < 0 ? 0 : (n > 0xFFFF ? 0xFFFF : n); };
const u32 = (x) => { let n = Math.floor(+x); if (!Number.isFinite(n) || n < 0) n = 0; return n >>> 0; };
const uv2 = (uv) => [u16(uv[0] || 0), u16(uv[1] || 0)];
const orDefault = (v, d) => (v === undefined || v === null ? d : v);

function mkMaterial(def) {
  const d = def || {};
  const albedo = orDefault(d.albedo, [0.8, 0.8, 0.8]), emission = orDefault(d.emission, [0, 0, 0]);
  return {
    name: String(d.name || ''),
    albedo: copy3(albedo).map((v) => Math.min(1, Math.max(0, +v))),
    emissive: !!d.emissive,
    emission: copy3(emission).map((v) => +v),
    reflective: !!d.reflective,
    roughness: Math.min(1, Math.max(0, +orDefault(d.roughness, 0.0))),
  };
}

const defaultLights = () => ({ env: { color: [0, 0, 0], intensity: 0.0 }, area: { center: [3, 2.8, 3], radius: 0.5, auto: true } });
const defaultCamera = () => ({ pos: [2.78, 2.73, -8.00], yaw: 0.0, pitch: 0.0, fovY: 80 * Math.PI / 180 });

export class SceneBuilder {
  constructor(maxSpheres, maxTris, maxQuads) {
    this._maxS = orDefault(maxSpheres, 64) | 0;
    this._maxT = orDefault(maxTris, 4096) | 0;
    this._maxQ = orDefault(maxQuads, 4096) | 0;
    this._materials = new Map();
    this._geom = { spheres: [], tris: [], quads: [] };
    this._atlas = { width: 0, height: 0 };
    this._lights = defaultLights();
    this._camera = defaultCamera();
    this.addMaterial(MaterialIds.LIGHT, { name: 'LIGHT', albedo: [1, 1, 1], emissive: true, emission: [16.86, 10.76, 8.2] });
    this.addMaterial(MaterialIds.WHITE, { name: 'WHITE', albedo: [0.7295, 0.7355, 0.7290], roughness: 0.6 });
    this.addMaterial(MaterialIds.GREEN, { name: 'GREEN', albedo: [0.1170, 0.4125, 0.1150], roughness: 0.6 });
    this.addMaterial(MaterialIds.RED, { name: 'RED', albedo: [0.6110, 0.0555, 0.0620], roughness: 0.6 });
    this.addMaterial(MaterialIds.GLASS, { name: 'GLASS', albedo: [1, 1, 1], reflective: true });
    this.addMaterial(MaterialIds.MIRROR, { name: 'MIRROR', albedo: [1, 1, 1], reflective: true });
  }

  addMaterial(id, def) { const k = u32(id); this._materials.set(k, mkMaterial(def)); return k; }
  hasMaterial(id) { return this._materials.has(u32(id)); }
  getMaterial(id) { return this._materials.get(u32(id)); }
  _mat(id) { return this.hasMaterial(id) ? u32(id) : DEFAULT_MAT_ID; }

  setCameraPose(pos, opts) {
    const p = orDefault(pos, [2.78, 2.73, -8.00]), o = opts || {};
    const yaw = orDefault(o.yaw, 0.0), pitch = orDefault(o.pitch, 0.0), fovYDeg = orDefault(o.fovYDeg, 80);
    if (!isFinite3(p) || !Number.isFinite(yaw) || !Number.isFinite(pitch)) throw new Error('setCameraPose: bad args');
    this._camera = { pos: copy3(p), yaw: +yaw, pitch: +pitch, fovY: (+fovYDeg) * Math.PI / 180 };
    return this;
  }

  setEnvLight(color, intensity) {
    const c = orDefault(color, [0, 0, 0]), k = orDefault(intensity, 0.0);
    if (!isFinite3(c) || !Number.isFinite(k)) throw new Error('setEnvLight: bad args');
    this._lights.env = { color: copy3(c), intensity: +k };
    return this;
  }

  setAreaLight(center, radius, opts) {
    const c = orDefault(center, [3, 2.8, 3]), r = orDefault(radius, 0.5), auto = orDefault((opts || {}).auto, true);
    if (!isFinite3(c) || !Number.isFinite(r)) throw new Error('setAreaLight: bad args');
    this._lights.area = { center: copy3(c), radius: +r, auto: !!auto };
    return this;
  }

  setTextureAtlasSize(width, height) {
    this._atlas = { width: Math.max(0, width | 0), height: Math.max(0, height | 0) };
    return this;
  }

  addSphere(center, radius, materialId) {
    const c = orDefault(center, [0, 0, 0]), r = orDefault(radius, 1.0);
    if (!isFinite3(c) || !Number.isFinite(r)) throw new Error('addSphere: bad args');
    if (this._geom.spheres.length >= this._maxS) return this;
    this._geom.spheres.push({ p: copy3(c), r: +r, matId: this._mat(orDefault(materialId, DEFAULT_MAT_ID)) });
    return this;
  }

  addTriangle(a, b, c, materialId, uvA, uvB, uvC) {
    const A = orDefault(a, [0, 0, 0]), B = orDefault(b, [1, 0, 0]), C = orDefault(c, [0, 1, 0]);
    if (!isFinite3(A) || !isFinite3(B) || !isFinite3(C)) throw new Error('addTriangle: bad args');
    if (this._geom.tris.length >= this._maxT) return this;
    this._geom.tris.push({
      a: copy3(A), b: copy3(B), c: copy3(C), matId: this._mat(orDefault(materialId, DEFAULT_MAT_ID)),
      uvA: uv2(orDefault(uvA, [0, 0])), uvB: uv2(orDefault(uvB, [0, 0])), uvC: uv2(orDefault(uvC, [0, 0])),
    });
    return this;
  }

  addQuad(a, b, c, d, materialId, uv0, uv1, uv2_, uv3) {
    const A = orDefault(a, [0, 0, 0]), B = orDefault(b, [1, 0, 0]), C = orDefault(c, [1, 1, 0]), D = orDefault(d, [0, 1, 0]);
    if (!isFinite3(A) || !isFinite3(B) || !isFinite3(C) || !isFinite3(D)) throw new Error('addQuad: bad args');
    if (this._geom.quads.length >= this._maxQ) return this;
    this._geom.quads.push({
      a: copy3(A), b: copy3(B), c: copy3(C), d: copy3(D), matId: this._mat(orDefault(materialId, DEFAULT_MAT_ID)),
      uv0: uv2(orDefault(uv0, [0, 0])), uv1: uv2(orDefault(uv1, [0, 0])), uv2: uv2(orDefault(uv2_, [0, 0])), uv3: uv2(orDefault(uv3, [0, 0])),
    });
    return this;
  }

  addRect(p00, p10, p11, p01, materialId, uv00, uv10, uv11, uv01) {
    return this.addQuad(p00, p10, p11, p01, materialId, uv00, uv10, uv11, uv01);
  }

  // positions: 3 numbers per vertex; indices: optional triples; uvs: optional uint16 pairs per vertex.
  // Plain arrays as in the reference, typed arrays accepted as well.
  addMesh(mesh) {
    const positions = mesh.positions, indices = orDefault(mesh.indices, null), uvs = orDefault(mesh.uvs, null);
    const materialId = orDefault(mesh.materialId, DEFAULT_MAT_ID);
    const isArr = Array.isArray(positions) || ArrayBuffer.isView(positions);
    if (!isArr || positions.length % 3 !== 0) return this;
    const V = positions.length / 3;
    const getV = (i) => [positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]];
    const getUV = (i) => ((!uvs || uvs.length < 2 * (i + 1)) ? [0, 0] : [u16(uvs[2 * i] | 0), u16(uvs[2 * i + 1] | 0)]);
    if (indices && indices.length % 3 === 0) {
      for (let t = 0; t < indices.length; t += 3) {
        const i0 = indices[t] | 0, i1 = indices[t + 1] | 0, i2 = indices[t + 2] | 0;
        if (i0 < 0 || i0 >= V || i1 < 0 || i1 >= V || i2 < 0 || i2 >= V) continue;
        this.addTriangle(getV(i0), getV(i1), getV(i2), materialId, getUV(i0), getUV(i1), getUV(i2));
      }
    } else {
      for (let i = 0; i + 8 < positions.length; i += 9) {
        this.addTriangle([positions[i], positions[i + 1], positions[i + 2]], [positions[i + 3], positions[i + 4], positions[i + 5]],
          [positions[i + 6], positions[i + 7], positions[i + 8]], materialId, [0, 0], [0, 0], [0, 0]);
      }
    }
    return this;
  }

  toUnified() {
    const table = {};
    for (const [id, m] of this._materials.entries()) {
      table[id] = { name: m.name, albedo: copy3(m.albedo), emissive: m.emissive, emission: copy3(m.emission), reflective: m.reflective, roughness: m.roughness };
    }
    const i2 = (p) => [p[0] | 0, p[1] | 0];
    return {
      version: 2,
      camera: Object.assign({}, this._camera, { pos: copy3(this._camera.pos) }),
      atlas: { width: this._atlas.width | 0, height: this._atlas.height | 0 },
      materials: { table },
      geometry: {
        spheres: this._geom.spheres.map((s) => ({ p: copy3(s.p), r: +s.r, matId: s.matId >>> 0 })),
        tris: this._geom.tris.map((t) => ({ a: copy3(t.a), b: copy3(t.b), c: copy3(t.c), matId: t.matId >>> 0, uvA: i2(t.uvA), uvB: i2(t.uvB), uvC: i2(t.uvC) })),
        quads: this._geom.quads.map((q) => ({ a: copy3(q.a), b: copy3(q.b), c: copy3(q.c), d: copy3(q.d), matId: q.matId >>> 0, uv0: i2(q.uv0), uv1: i2(q.uv1), uv2: i2(q.uv2), uv3: i2(q.uv3) })),
      },
      lights: {
        env: { color: copy3(this._lights.env.color), intensity: +this._lights.env.intensity },
        area: { center: copy3(this._lights.area.center), radius: +this._lights.area.radius, auto: !!this._lights.area.auto },
      },
    };
  }

  toPathTracer() { return this.toUnified(); }
  toObject() { return this.toPathTracer(); }

  reset() {
    this._geom = { spheres: [], tris: [], quads: [] };
    this._atlas = { width: 0, height: 0 };
    this._lights = defaultLights();
    this._camera = defaultCamera();
    return this;
  }
}

export function createSceneBuilder(maxSpheres, maxTris, maxQuads) {
  return new SceneBuilder(maxSpheres, maxTris, maxQuads);
}

export function fromObject(obj) {
  const sb = new SceneBuilder();
  if (!obj || typeof obj !== 'object') return sb;
  if (obj.camera) {
    const c = obj.camera;
    const fovY = Number.isFinite(c.fovY) ? +c.fovY : 80 * Math.PI / 180;
    sb.setCameraPose(orDefault(c.pos, [2.78, 2.73, -8.00]), {
      yaw: Number.isFinite(c.yaw) ? +c.yaw : 0.0, pitch: Number.isFinite(c.pitch) ? +c.pitch : 0.0, fovYDeg: fovY * 180 / Math.PI,
    });
  }
  if (obj.atlas) sb.setTextureAtlasSize(obj.atlas.width | 0, obj.atlas.height | 0);
  if (obj.materials && obj.materials.table) {
    for (const k of Object.keys(obj.materials.table)) sb.addMaterial(u32(k), obj.materials.table[k]);
  }
  if (obj.lights) {
    if (obj.lights.env) sb.setEnvLight(orDefault(obj.lights.env.color, [0, 0, 0]), orDefault(obj.lights.env.intensity, 0.0));
    if (obj.lights.area) sb.setAreaLight(orDefault(obj.lights.area.center, [3, 2.8, 3]), +obj.lights.area.radius || 0.5, { auto: !!obj.lights.area.auto });
  }
  const g = obj.geometry || {};
  for (const s of (g.spheres || [])) sb.addSphere(orDefault(s.p, [0, 0, 0]), Number(s.r) || 1, u32(orDefault(s.matId, DEFAULT_MAT_ID)));
  for (const t of (g.tris || [])) {
    sb.addTriangle(orDefault(t.a, [0, 0, 0]), orDefault(t.b, [1, 0, 0]), orDefault(t.c, [0, 1, 0]), u32(orDefault(t.matId, DEFAULT_MAT_ID)),
      orDefault(t.uvA, [0, 0]), orDefault(t.uvB, [0, 0]), orDefault(t.uvC, [0, 0]));
  }
  for (const q of (g.quads || [])) {
    sb.addQuad(orDefault(q.a, [0, 0, 0]), orDefault(q.b, [1, 0, 0]), orDefault(q.c, [1, 1, 0]), orDefault(q.d, [0, 1, 0]), u32(orDefault(q.matId, DEFAULT_MAT_ID)),
      orDefault(q.uv0, [0, 0]), orDefault(q.uv1, [0, 0]), orDefault(q.uv2, [0, 0]), orDefault(q.uv3, [0, 0]));
  }
  return sb;
}

// Wraps typed-array geometry as a scene the raster backend ingests without per-triangle objects:
// { mesh: { positions, indices, colors? }, lights?, ... }.
export function meshScene(mesh, extra) {
  return Object.assign({ mesh: { positions: mesh.positions, indices: mesh.indices, colors: mesh.colors || null } }, extra || {});
}
