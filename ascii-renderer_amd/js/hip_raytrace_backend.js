// hip_raytrace_backend.js - the ray backend behind the router: the reference's deterministic single-bounce
// ray tracer (js/render/backends/raytrace.js, raytrace_shader.js) with its host half in JavaScript and the
// per-cell work in the HIP library (ar_rt_set_scene / ar_rt_render through the N-API addon).
//
// Contract (raytrace.js:1-17,135-138,316-421): name 'raytrace'; setScene(scene|null);
// render(timeSec, framebuffer, appState) -> undefined; renderRaw; getPixels; dispose.  The scene is packed at
// setScene into the same uniform arrays the reference uploads (unified scenes as they are, legacy ones adapted
// as _resolveUnified does); the camera is read live from camera.js at render time; the framebuffer is filled
// synchronously, row 0 = top, and left alone when its length does not match.
// Beyond the reference: appState.ascii (Uint8Array cols*rows), when present, receives the ASCII plane.
import { config } from './config.js';
import { camera as LiveCamera } from './camera.js';
import { native } from './native.js';
import { flipAndCopy, rampBytes, modeParams } from './hip_backend.js';

const LEGACY_PALETTE = { 0: [5, 5, 5], 1: [0.9, 0.9, 0.9], 2: [0.7, 0.9, 0.7], 3: [0.95, 0.45, 0.45], 6: [0.9, 0.95, 1.0] };   // raytrace.js:156-162

// raytrace.js:52-61
export function raytraceLimits() {
  const LIM = (config.PATH_TRACER && config.PATH_TRACER.LIMITS) || {};
  const pick = (v, d) => ((v === undefined || v === null ? d : v) | 0);
  return { MAX_SPHERES: pick(LIM.MAX_SPHERES, 32), MAX_PLANES: pick(LIM.MAX_PLANES, 16), MAX_TRIS: pick(LIM.MAX_TRIS, 64), MAX_MATS: 64, MAX_PL: 8, MAX_DL: 2 };
}

// The legacy scene shape {spheres: [{p, r, m}], planes: [{p: [nx, ny, nz, d], m}], tris: [{a, b, c, m}], envLight,
// dirLight} in unified form (raytrace.js:140-193): every primitive gets a material of its own, made from its legacy
// material number m (0 counts as 1) - palette colour or grey, a mirror when m is above 4.  Scenes that already have a
// geometry, materials or lights member pass through untouched.
const LEGACY_KINDS = [
  ['spheres', (s, mat) => ({ p: s.p, r: s.r, mat })],
  ['planes', (pl, mat) => ({ n: [pl.p[0], pl.p[1], pl.p[2]], d: pl.p[3], mat })],
  ['tris', (t, mat) => ({ a: t.a, b: t.b, c: t.c, mat })],
];

export function resolveUnified(scene) {
  const legacy = scene || {};
  if (legacy.geometry || legacy.materials || legacy.lights) return legacy;
  const materials = [], geometry = {};
  for (const [kind, convert] of LEGACY_KINDS) {
    geometry[kind] = (legacy[kind] || []).map((prim) => {
      const number = (prim.m || 1) | 0;
      materials.push({ albedo: LEGACY_PALETTE[number] || [0.8, 0.8, 0.8], emissive: false, emission: [0, 0, 0], reflective: number > 4, roughness: 0 });
      return convert(prim, materials.length - 1);
    });
  }
  const sky = legacy.envLight, sun = legacy.dirLight;
  return {
    camera: legacy.camera || null, geometry, materials,
    lights: {
      points: [],
      directionals: sun ? [{ dir: sun.dir || [0, -1, 0], color: sun.color || [1, 1, 1], intensity: sun.intensity || 0 }] : [],
      env: sky ? { color: sky.color || [0, 0, 0], intensity: sky.intensity || 0 } : { color: [0, 0, 0], intensity: 0 },
    },
  };
}

// A scene value as the reference coerces it on its way into a Float32Array: anything that is not a number (missing,
// NaN) becomes 0.
const scalar = (v) => +v || 0;
const component = (vec, k) => (vec === undefined || vec === null ? 0 : scalar(vec[k]));
const scaled = (vec, k, factor) => scalar(vec[k]) * factor;

// One row per uniform array the reference's ray tracer uploads (raytrace.js:195-312): which list of the unified scene
// feeds it, which limit caps it, and how one list item becomes the array's elements.  `columns` name the typed arrays
// of the packed scene: [key, array type, elements per item, item -> elements].
const RT_GROUPS = [
  { count: 'sphereCount', limit: 'MAX_SPHERES', items: (u) => (u.geometry || {}).spheres || [], columns: [
    ['sphereGeom', Float32Array, 4, (s) => [component(s.p, 0), component(s.p, 1), component(s.p, 2), scalar(s.r)]],
    ['sphereMaterial', Int32Array, 1, (s) => [s.mat | 0]]] },
  { count: 'planeCount', limit: 'MAX_PLANES', items: (u) => (u.geometry || {}).planes || [], columns: [
    ['planeGeom', Float32Array, 4, (p) => { const n = p.n || [0, 1, 0]; return [scalar(n[0]), scalar(n[1]), scalar(n[2]), scalar(p.d)]; }],
    ['planeMaterial', Int32Array, 1, (p) => [p.mat | 0]]] },
  { count: 'triCount', limit: 'MAX_TRIS', items: (u) => (u.geometry || {}).tris || [], columns: [
    ['triA', Float32Array, 3, (t) => [0, 1, 2].map((k) => component(t.a, k))],
    ['triB', Float32Array, 3, (t) => [0, 1, 2].map((k) => component(t.b, k))],
    ['triC', Float32Array, 3, (t) => [0, 1, 2].map((k) => component(t.c, k))],
    ['triMaterial', Int32Array, 1, (t) => [t.mat | 0]]] },
  // (a materials member that is not an array has no length: the minimum is NaN, no entry is read and the count reads 0;
  // an empty array gets one default material, below)
  { count: 'materialCount', limit: 'MAX_MATS', items: (u) => u.materials || [], holes: true, columns: [
    ['materialAlbedo', Float32Array, 3, (m) => [0, 1, 2].map((k) => component(m.albedo, k))],
    ['materialMirror', Int32Array, 1, (m) => [m.reflective ? 1 : 0]]] },
  { count: 'pointCount', limit: 'MAX_PL', items: (u) => (Array.isArray((u.lights || {}).points) ? u.lights.points : []), columns: [
    ['pointPos', Float32Array, 3, (l) => { const p = l.p || [0, 0, 0]; return [0, 1, 2].map((k) => scalar(p[k])); }],
    ['pointColor', Float32Array, 3, (l) => { const c = l.color || [1, 1, 1], w = scalar(l.intensity); return [0, 1, 2].map((k) => scaled(c, k, w)); }]] },
  { count: 'dirCount', limit: 'MAX_DL', items: (u) => (Array.isArray((u.lights || {}).directionals) ? u.lights.directionals : []), columns: [
    ['dirDir', Float32Array, 3, (l) => { const d = l.dir || [0, -1, 0]; return [0, 1, 2].map((k) => scalar(d[k])); }],
    ['dirColor', Float32Array, 3, (l) => { const c = l.color || [1, 1, 1], w = scalar(l.intensity); return [0, 1, 2].map((k) => scaled(c, k, w)); }]] },
];

// Scene -> the fixed-size arrays and counts of the ray tracer's uniforms, byte for byte what the reference uploads
// (raytrace.js:195-312), under this build's own names: sphereCount / sphereGeom (x, y, z, r) / sphereMaterial,
// planeCount / planeGeom (n, d) / planeMaterial, triCount / triA / triB / triC / triMaterial, materialCount /
// materialAlbedo / materialMirror, pointCount / pointPos / pointColor, dirCount / dirDir / dirColor, environment.
export function packRaytraceScene(scene, limits) {
  const caps = limits || raytraceLimits(), unified = resolveUnified(scene);
  const packed = {};
  for (const group of RT_GROUPS) {
    const items = group.items(unified), cap = caps[group.limit];
    const n = Math.min(items.length, cap);
    for (const [key, ArrayType, width] of group.columns) packed[key] = new ArrayType(cap * width);
    for (let i = 0; i < n; i++) {
      const item = group.holes ? (items[i] || {}) : items[i];
      for (const [key, , width, elements] of group.columns) packed[key].set(elements(item), i * width);
    }
    packed[group.count] = n;
  }
  if (packed.materialCount === 0) {                     // no materials at all: one diffuse grey (raytrace.js:237-241)
    packed.materialAlbedo.fill(0.8, 0, 3);
    packed.materialMirror[0] = 0;
    packed.materialCount = 1;
  }
  for (const group of RT_GROUPS) packed[group.count] |= 0;
  const env = (unified.lights || {}).env, weight = env ? scalar(env.intensity) : 0;
  packed.environment = new Float32Array(env ? [0, 1, 2].map((k) => component(env.color, k) * weight) : [0, 0, 0]);
  return packed;
}

// Frame uniforms of raytrace.js:330-346.
export function raytraceFrameUniforms() {
  const cam = LiveCamera ? { pos: LiveCamera.pos, yaw: LiveCamera.yaw, pitch: LiveCamera.pitch } : { pos: { x: 0, y: 0, z: 5 }, yaw: 0, pitch: 0 };
  const fovRad = ((config.FOVY_DEG || 80) * Math.PI) / 180;
  const pixAsp = (config.PATH_TRACER && Number.isFinite(config.PATH_TRACER.PIXEL_ASPECT)) ? Math.max(1e-6, config.PATH_TRACER.PIXEL_ASPECT) : 1.0;
  return { camPos: new Float32Array([cam.pos.x, cam.pos.y, cam.pos.z]), yaw: cam.yaw || 0, pitch: cam.pitch || 0, fovY: fovRad, pixAspect: pixAsp };
}

export class HipRaytraceBackend {
  constructor(opts) {
    this.name = 'raytrace';
    this._device = (opts && opts.device) | 0;
    this._h = null;
    this._scene = null;
    this._packed = null;
    this._width = 0; this._height = 0;
    this._lastTopDown = null;
  }

  _ensure() { if (!this._h) this._h = native().create(this._device); }

  dispose() {
    if (this._h) native().destroy(this._h);
    this._h = null;
    this._lastTopDown = null;
  }

  setScene(scene) {
    this._scene = scene || null;
    this._ensure();
    this._packed = packRaytraceScene(this._scene);       // the reference packs a null scene too (raytrace.js:135-138)
    native().rtSetScene(this._h, this._packed);
  }

  render(_timeSec, framebuffer, appState) {
    this._ensure();
    const w = ((appState && appState.cols) | 0) || (config.VIRTUAL_GRID_WIDTH | 0) || 1;
    const h = ((appState && appState.rows) | 0) || (config.VIRTUAL_GRID_HEIGHT | 0) || 1;
    this._width = w; this._height = h;
    if (!this._scene) return;                            // raytrace.js:321
    const N = native(), F = raytraceFrameUniforms(), mode = modeParams();
    N.setAscii(this._h, rampBytes(config.ASCII_RAMP || '@%#*+=-:. '), mode.on, mode.radius, mode.thresh);
    const ascii = (appState && appState.ascii && appState.ascii.length === w * h) ? appState.ascii : null;
    if (!this._own || this._own.length !== w * h * 4) this._own = new Uint8Array(N.allocHost(this._h, w * h * 4));
    N.rtRender(this._h, w, h, F.camPos, F.yaw, F.pitch, F.fovY, F.pixAspect, this._own, ascii);
    this._lastTopDown = this._own;
    if (framebuffer && framebuffer.length === this._own.length) framebuffer.set(this._own);
  }

  renderRaw(args) {
    if (args && args.framebuffer && args.appState) return this.render(args.time || 0, args.framebuffer, args.appState);
    return undefined;
  }

  // The reference keeps the bottom-up readPixels image (raytrace.js:412-421); this backend keeps it top-down.
  getPixels(args) {
    const a = args || {};
    if (!this._lastTopDown) return null;
    const w = a.width || this._width, h = a.height || this._height;
    const flipY = a.flipY === undefined ? true : !!a.flipY;
    if (a.framebuffer && a.framebuffer.length >= w * h * 4) {
      if (flipY) a.framebuffer.set(this._lastTopDown); else flipAndCopy(this._lastTopDown, a.framebuffer, w, h);
      return a.framebuffer;
    }
    const out = new Uint8Array(w * h * 4);
    flipAndCopy(this._lastTopDown, out, w, h);
    return out;
  }
}
