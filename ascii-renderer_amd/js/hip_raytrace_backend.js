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

// raytrace.js:140-193: unified scenes pass through; the legacy shape gets one material per primitive
export function resolveUnified(scene) {
  const S = scene || {};
  if (S.geometry || S.materials || S.lights) return S;
  const uni = { camera: S.camera || null, geometry: { spheres: [], planes: [], tris: [] }, materials: [],
    lights: { points: [], directionals: [], env: { color: [0, 0, 0], intensity: 0 } } };
  const pushMatFromM = (m) => {
    const idx = uni.materials.length;
    uni.materials.push({ albedo: LEGACY_PALETTE[m | 0] || [0.8, 0.8, 0.8], emissive: false, emission: [0, 0, 0], reflective: (m | 0) > 4, roughness: 0 });
    return idx;
  };
  (S.spheres || []).forEach((s) => uni.geometry.spheres.push({ p: s.p, r: s.r, mat: pushMatFromM(s.m || 1) }));
  (S.planes || []).forEach((p) => uni.geometry.planes.push({ n: [p.p[0], p.p[1], p.p[2]], d: p.p[3], mat: pushMatFromM(p.m || 1) }));
  (S.tris || []).forEach((t) => uni.geometry.tris.push({ a: t.a, b: t.b, c: t.c, mat: pushMatFromM(t.m || 1) }));
  if (S.envLight) uni.lights.env = { color: S.envLight.color || [0, 0, 0], intensity: S.envLight.intensity || 0 };
  if (S.dirLight) uni.lights.directionals.push({ dir: S.dirLight.dir || [0, -1, 0], color: S.dirLight.color || [1, 1, 1], intensity: S.dirLight.intensity || 0 });
  return uni;
}

const num = (v) => +v || 0;
const at = (a, i) => (a === undefined || a === null ? undefined : a[i]);

// raytrace.js:195-312: fixed-size arrays plus counts, exactly as they go to the uniforms
export function packRaytraceScene(scene, limits) {
  const L = limits || raytraceLimits(), U = resolveUnified(scene);
  const G = U.geometry || {};
  const spheres = G.spheres || [], planes = G.planes || [], tris = G.tris || [], mats = U.materials || [], lights = U.lights || {};
  const S4 = new Float32Array(L.MAX_SPHERES * 4), Si = new Int32Array(L.MAX_SPHERES);
  const P4 = new Float32Array(L.MAX_PLANES * 4), Pi = new Int32Array(L.MAX_PLANES);
  const Ta = new Float32Array(L.MAX_TRIS * 3), Tb = new Float32Array(L.MAX_TRIS * 3), Tc = new Float32Array(L.MAX_TRIS * 3), Ti = new Int32Array(L.MAX_TRIS);
  const Ma = new Float32Array(L.MAX_MATS * 3), Mr = new Int32Array(L.MAX_MATS);
  let numMats = Math.min(mats.length, L.MAX_MATS);            // NaN when materials is not an array: no entry is read
  for (let i = 0; i < numMats; i++) {
    const m = mats[i] || {};
    Ma[i * 3] = num(at(m.albedo, 0)); Ma[i * 3 + 1] = num(at(m.albedo, 1)); Ma[i * 3 + 2] = num(at(m.albedo, 2));
    Mr[i] = m.reflective ? 1 : 0;
  }
  if (numMats === 0) { Ma[0] = Ma[1] = Ma[2] = 0.8; Mr[0] = 0; numMats = 1; }
  const numS = Math.min(spheres.length, L.MAX_SPHERES);
  for (let i = 0; i < numS; i++) {
    const s = spheres[i];
    S4[i * 4] = num(at(s.p, 0)); S4[i * 4 + 1] = num(at(s.p, 1)); S4[i * 4 + 2] = num(at(s.p, 2)); S4[i * 4 + 3] = num(s.r);
    Si[i] = s.mat | 0;
  }
  const numP = Math.min(planes.length, L.MAX_PLANES);
  for (let i = 0; i < numP; i++) {
    const p = planes[i], n = p.n || [0, 1, 0];
    P4[i * 4] = num(n[0]); P4[i * 4 + 1] = num(n[1]); P4[i * 4 + 2] = num(n[2]); P4[i * 4 + 3] = num(p.d);
    Pi[i] = p.mat | 0;
  }
  const numT = Math.min(tris.length, L.MAX_TRIS);
  for (let i = 0; i < numT; i++) {
    const t = tris[i];
    for (let k = 0; k < 3; k++) { Ta[i * 3 + k] = num(at(t.a, k)); Tb[i * 3 + k] = num(at(t.b, k)); Tc[i * 3 + k] = num(at(t.c, k)); }
    Ti[i] = t.mat | 0;
  }
  const Ppos = new Float32Array(L.MAX_PL * 3), Pcol = new Float32Array(L.MAX_PL * 3), Ddir = new Float32Array(L.MAX_DL * 3), Dcol = new Float32Array(L.MAX_DL * 3);
  let numPL = 0, numDL = 0;
  if (Array.isArray(lights.points)) {
    numPL = Math.min(lights.points.length, L.MAX_PL);
    for (let i = 0; i < numPL; i++) {
      const Lp = lights.points[i], p = Lp.p || [0, 0, 0], c = Lp.color || [1, 1, 1], k = num(Lp.intensity);
      for (let q = 0; q < 3; q++) { Ppos[i * 3 + q] = num(p[q]); Pcol[i * 3 + q] = num(c[q]) * k; }
    }
  }
  if (Array.isArray(lights.directionals)) {
    numDL = Math.min(lights.directionals.length, L.MAX_DL);
    for (let i = 0; i < numDL; i++) {
      const Ld = lights.directionals[i], d = Ld.dir || [0, -1, 0], c = Ld.color || [1, 1, 1], k = num(Ld.intensity);
      for (let q = 0; q < 3; q++) { Ddir[i * 3 + q] = num(d[q]); Dcol[i * 3 + q] = num(c[q]) * k; }
    }
  }
  const e = lights.env, ek = e ? num(e.intensity) : 0;
  const envCol = new Float32Array(e ? [num(at(e.color, 0)) * ek, num(at(e.color, 1)) * ek, num(at(e.color, 2)) * ek] : [0, 0, 0]);
  return { numS: numS | 0, S4, Si, numP: numP | 0, P4, Pi, numT: numT | 0, Ta, Tb, Tc, Ti, numMats: numMats | 0, Ma, Mr,
    numPL: numPL | 0, Ppos, Pcol, numDL: numDL | 0, Ddir, Dcol, envCol };
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
    this._cache = null;
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
    this._cache = packRaytraceScene(this._scene);       // the reference packs a null scene too (raytrace.js:135-138)
    native().rtSetScene(this._h, this._cache);
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
