// hip_backend.js - the raster backend behind the router: host half in JavaScript, device half in the
// HIP library (through the N-API addon).
//
// Contract (reference js/render/backends/raster.js:51-71,116-133,269-272,364-375):
//   name; setScene(scene|null); render(timeSec, framebuffer, appState) -> undefined;
//   renderRaw({time, framebuffer, appState}); getPixels({framebuffer, width, height, flipY=true});
//   dispose() (idempotent).
// The caller owns `framebuffer` (Uint8Array / Uint8ClampedArray of cols*rows*4, row 0 = top); it is
// filled synchronously, and silently left alone when its length does not match (raster.js:360).
// Geometry is snapshotted at setScene; the camera is read live from camera.js at render time and the
// grid size from appState.cols / rows (raster.js:271-272, 284-286).
// Beyond the reference: appState.depth (Float32Array cols*rows) and appState.ascii (Uint8Array
// cols*rows) are filled when present, and a scene may carry `mesh` = {positions: Float32Array,
// indices: Uint32Array, colors?: Float32Array (per triangle)} as a typed-array fast path.
// Also beyond it: renderAsync(timeSec, appState, framebuffer?) -> token and collect(token, framebuffer?) - the
// reference's caller already keeps one frame in flight and two framebuffers (js/main.js:243-248,364-375) around a
// render that blocks; with these two calls several frames render and copy out while the caller works.
import { config } from './config.js';
import { camera as LiveCamera } from './camera.js';
import { native, AR_OPT_PATH } from './native.js';

export const MAX_POINT_LIGHTS = 8;                       // raster_shader.js:5
const DEFAULT_AMBIENT = [0.15, 0.18, 0.22];              // raster.js:66
const DEFAULT_LIGHT_DIR = [0.25, -1.0, 0.15];            // raster.js:67 (not normalized; kept)
const DEFAULT_LIGHT_COLOR = [1.2, 1.15, 1.1];            // raster.js:68
const LEGACY_COLORS = { 0: [5, 5, 5], 1: [0.9, 0.9, 0.9], 2: [0.7, 0.9, 0.7], 3: [0.95, 0.45, 0.45], 6: [0.9, 0.95, 1.0] };
const FALLBACK_COLOR = [0.8, 0.8, 0.8];
const SPHERE_LAT = 12, SPHERE_LON = 16;                  // raster.js:197
const PLANE_HALF_SIZE = 20.0;                            // raster.js:187

/* ------------------------------- small math -------------------------------- */
function cross(a, b) { return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]; }
function dot(a, b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
function normalize(v) { const L = Math.hypot(v[0], v[1], v[2]) || 1; return [v[0] / L, v[1] / L, v[2] / L]; }
const clamp01 = (v) => Math.min(1, Math.max(0, v));

// raster.js:15-22, column-major
export function mat4Perspective(fovyRad, aspect, near, far) {
  const f = 1 / Math.tan(Math.max(1e-6, fovyRad * 0.5)), nf = 1 / (near - far);
  const out = new Float32Array(16);
  out[0] = f / aspect; out[5] = f; out[10] = (far + near) * nf; out[11] = -1; out[14] = 2 * far * near * nf;
  return out;
}

// raster.js:41-45: yaw = pitch = 0 looks along +X
export function yawPitchToDir(yaw, pitch) {
  const cp = Math.cos(pitch), sp = Math.sin(pitch), cy = Math.cos(yaw), sy = Math.sin(yaw);
  return [cp * cy, sp, cp * sy];
}

// raster.js:30-39, column-major
export function mat4LookAt(eye, center, up) {
  const f = normalize([center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]]);
  const s = normalize(cross(f, up));
  const u = cross(s, f);
  const out = new Float32Array(16);
  out[0] = s[0]; out[1] = u[0]; out[2] = -f[0];
  out[4] = s[1]; out[5] = u[1]; out[6] = -f[1];
  out[8] = s[2]; out[9] = u[2]; out[10] = -f[2];
  out[12] = -dot(s, eye); out[13] = -dot(u, eye); out[14] = dot(f, eye); out[15] = 1;
  return out;
}

// Projection and view exactly as RasterBackend.render uploads them (raster.js:284-304).  The
// reference reads config.FOVY_DEG, which does not exist, so the field of view is always 80 degrees.
export function cameraMatrices(cam, cols, rows) {
  const eye = [cam.pos.x, cam.pos.y, cam.pos.z];
  const dir = yawPitchToDir(cam.yaw || 0, cam.pitch || 0);
  const center = [eye[0] + dir[0], eye[1] + dir[1], eye[2] + dir[2]];
  const pixAsp = (config.PATH_TRACER && Number.isFinite(config.PATH_TRACER.PIXEL_ASPECT))
    ? Math.max(1e-6, config.PATH_TRACER.PIXEL_ASPECT) : 1.0;
  const fovRad = ((config.FOVY_DEG || 80) * Math.PI / 180);
  const aspect = Math.max(1e-6, (cols / Math.max(1, rows)) * pixAsp);
  return { proj: mat4Perspective(fovRad, aspect, 0.05, 100.0), view: mat4LookAt(eye, center, [0, 1, 0]) };
}

// Light uniforms as raster.js:307-342 packs them.
export function packLights(scene) {
  const U = scene || {}, L = U.lights || null;
  let ambient = DEFAULT_AMBIENT, dirDir = DEFAULT_LIGHT_DIR, dirCol = DEFAULT_LIGHT_COLOR;
  if (L && L.env) {
    const e = L.env, k = e.intensity || 0, c = e.color || [];
    ambient = [(c[0] || 0) * k, (c[1] || 0) * k, (c[2] || 0) * k];
  } else if (U.envLight && U.envLight.color) {
    ambient = U.envLight.color;
  }
  if (L && L.directionals && L.directionals.length) {
    const d0 = L.directionals[0], k = d0.intensity || 0, c = d0.color || [];
    dirDir = d0.dir || dirDir;
    dirCol = [(c[0] || 1) * k, (c[1] || 1) * k, (c[2] || 1) * k];
  } else if (U.dirLight && U.dirLight.dir) {
    dirDir = U.dirLight.dir;
    dirCol = U.dirLight.color || dirCol;
  }
  const pts = (L && L.points && L.points.length) ? L.points : [];
  const n = Math.min(MAX_POINT_LIGHTS, pts.length);
  const plPos = new Float32Array(n * 3), plCol = new Float32Array(n * 3);
  for (let i = 0; i < n; i++) {
    const p = pts[i].p || [0, 0, 0], c = pts[i].color || [1, 1, 1], k = +pts[i].intensity || 0;
    plPos[i * 3] = p[0]; plPos[i * 3 + 1] = p[1]; plPos[i * 3 + 2] = p[2];
    plCol[i * 3] = c[0] * k; plCol[i * 3 + 1] = c[1] * k; plCol[i * 3 + 2] = c[2] * k;
  }
  return {
    ambient: new Float32Array([ambient[0], ambient[1], ambient[2]]),
    dir: new Float32Array([dirDir[0], dirDir[1], dirDir[2]]),
    dirColor: new Float32Array([dirCol[0], dirCol[1], dirCol[2]]),
    plPos, plCol,
  };
}

// raster.js:135-146
export function resolveColor(scene, mat, m) {
  if (scene && Array.isArray(scene.materials) && Number.isFinite(mat)) {
    const M = scene.materials[mat | 0];
    if (M && M.albedo) return [clamp01(+M.albedo[0] || 0), clamp01(+M.albedo[1] || 0), clamp01(+M.albedo[2] || 0)];
  }
  if (Number.isFinite(m)) return LEGACY_COLORS[m | 0] || FALLBACK_COLOR;
  return FALLBACK_COLOR;
}

// Texel coordinate as the path tracer uploads it, (v|0)>>>0 (pathtrace.js:186-195,230-234), kept to the 16 bits
// SceneBuilder stores (scene_api.js:28-31).
function u16(v) { const n = v | 0; return n < 0 ? 0 : (n > 0xFFFF ? 0xFFFF : n); }
function uvOf(uv) { return uv ? [u16(uv[0] || 0), u16(uv[1] || 0)] : [0, 0]; }

// Albedo of a SceneBuilder primitive: materials.table[matId].albedo (scene_api.js:196-207) when the scene
// carries such a table, else raster.js's own rule.
export function resolveColorTable(scene, prim) {
  const table = scene && scene.materials && !Array.isArray(scene.materials) ? scene.materials.table : null;
  if (table && Number.isFinite(prim.matId)) {
    const M = table[String(prim.matId | 0)];
    if (M && M.albedo) return [clamp01(+M.albedo[0] || 0), clamp01(+M.albedo[1] || 0), clamp01(+M.albedo[2] || 0)];
  }
  return resolveColor(scene, prim.mat, prim.m);
}

// Growing interleaved vertex buffer (position, normal, colour: nine floats per vertex) plus index list: what the scene
// walkers below emit into.  Values are stored as binary32 on the way in, exactly as a Float32Array built at the end
// from JavaScript numbers would hold them.
class MeshAssembler {
  constructor() {
    this.floats = new Float32Array(9 * 256);
    this.indices = new Uint32Array(3 * 256);
    this.vertexCount = 0;
    this.indexCount = 0;
    this.texelRows = [];          // [triangle number, eight u16 values] (textured builds only)
  }

  get triangleCount() { return this.indexCount / 3; }

  vertex(position, normal, color) {
    if (9 * (this.vertexCount + 1) > this.floats.length) {
      const grown = new Float32Array(2 * this.floats.length);
      grown.set(this.floats);
      this.floats = grown;
    }
    this.floats.set(position, 9 * this.vertexCount);
    this.floats.set(normal, 9 * this.vertexCount + 3);
    this.floats.set(color, 9 * this.vertexCount + 6);
    return this.vertexCount++;
  }

  triangle(i, j, k) {
    if (this.indexCount + 3 > this.indices.length) {
      const grown = new Uint32Array(2 * this.indices.length);
      grown.set(this.indices);
      this.indices = grown;
    }
    this.indices[this.indexCount++] = i; this.indices[this.indexCount++] = j; this.indices[this.indexCount++] = k;
  }

  // three unshared vertices carrying the face normal normalize(cross(q - p, r - p)), or (0, 1, 0) for a face of
  // length <= 1e-8 (raster.js:159-169)
  flatTriangle(p, q, r, color) {
    let n = cross([q[0] - p[0], q[1] - p[1], q[2] - p[2]], [r[0] - p[0], r[1] - p[1], r[2] - p[2]]);
    const len = Math.hypot(n[0], n[1], n[2]);
    n = len > 1e-8 ? [n[0] / len, n[1] / len, n[2] / len] : [0, 1, 0];
    this.triangle(this.vertex(p, n, color), this.vertex(q, n, color), this.vertex(r, n, color));
  }

  texels(row) { this.texelRows.push([this.triangleCount, row]); }   // for the triangle emitted next

  finish(withTexels) {
    const out = { vbo: this.floats.slice(0, 9 * this.vertexCount), idx: this.indices.slice(0, this.indexCount) };
    if (withTexels) {
      out.uv8 = new Uint16Array(this.triangleCount * 8);
      for (const [t, row] of this.texelRows) out.uv8.set(row, 8 * t);
    }
    return out;
  }
}

// The four corners of a plane's stand-in rectangle in the order the two triangles use them: signs of the two
// in-plane axes (raster.js:187-193).
const PLANE_CORNER_SIGNS = [[-1, -1], [1, -1], [1, 1], [-1, 1]];
const PLANE_TRIANGLES = [[0, 1, 2], [0, 2, 3]];

function emitPlane(mesh, plane, color) {
  const unit = normalize(plane.n ? [plane.n[0], plane.n[1], plane.n[2]] : [0, 1, 0]);
  const offset = Number.isFinite(plane.d) ? plane.d : 0;
  const foot = [-offset * unit[0], -offset * unit[1], -offset * unit[2]];           // the plane's point nearest the origin
  const helper = Math.abs(unit[1]) < 0.99 ? [0, 1, 0] : [1, 0, 0];
  const axisU = normalize(cross(unit, helper)), axisV = normalize(cross(unit, axisU));
  const corners = PLANE_CORNER_SIGNS.map(([su, sv]) => foot.map((f, k) => f + (su * PLANE_HALF_SIZE * axisU[k] + sv * PLANE_HALF_SIZE * axisV[k])));
  for (const [i, j, k] of PLANE_TRIANGLES) mesh.flatTriangle(corners[i], corners[j], corners[k], color);
}

// Latitude / longitude sphere with smooth normals: (SPHERE_LAT + 1) rings of (SPHERE_LON + 1) vertices, ring 0 at
// the +Y pole, two triangles per cell wound (here, below, right), (right, below, below-right) (raster.js:196-219).
function emitSphere(mesh, sphere, color) {
  const centre = sphere.p || [0, 0, 0], radius = Number.isFinite(sphere.r) ? sphere.r : 1;
  const ring = SPHERE_LON + 1;
  let first = -1;
  for (let lat = 0; lat <= SPHERE_LAT; lat++) {
    const polar = (lat / SPHERE_LAT) * Math.PI, sinPolar = Math.sin(polar), cosPolar = Math.cos(polar);
    for (let lon = 0; lon <= SPHERE_LON; lon++) {
      const around = (lon / SPHERE_LON) * 2 * Math.PI;
      const out = [Math.cos(around) * sinPolar, cosPolar, Math.sin(around) * sinPolar];
      const v = mesh.vertex([centre[0] + radius * out[0], centre[1] + radius * out[1], centre[2] + radius * out[2]], out, color);
      if (first < 0) first = v;
    }
  }
  for (let lat = 0; lat < SPHERE_LAT; lat++) {
    for (let lon = 0; lon < SPHERE_LON; lon++) {
      const here = first + lat * ring + lon, below = here + ring;
      mesh.triangle(here, below, here + 1);
      mesh.triangle(here + 1, below, below + 1);
    }
  }
}

// A quad as the path tracer sees it (pathtrace.js:215-218, shader_utils.js:190-191): halves (a, b, c) and (a, c, d),
// two-sided - emitted once per winding, of which back-face culling keeps one pair.  Corner numbers per emitted triangle:
const QUAD_TRIANGLES = [[0, 1, 2], [0, 2, 3], [0, 2, 1], [0, 3, 2]];

function emitQuad(mesh, quad, color) {
  const corner = [quad.a || [0, 0, 0], quad.b || [1, 0, 0], quad.c || [1, 1, 0], quad.d || [0, 1, 0]];
  const texel = [uvOf(quad.uv0), uvOf(quad.uv1), uvOf(quad.uv2), uvOf(quad.uv3)];
  for (const tri of QUAD_TRIANGLES) {
    // sampleQuadFetch (shader_utils.js:120-132) samples a half unless all three of its coordinate pairs are zero; the
    // mirrored copy of a half has the same three pairs
    const sampled = tri.some((c) => texel[c][0] !== 0 || texel[c][1] !== 0) ? 1 : 0;
    mesh.texels([...texel[tri[0]], ...texel[tri[1]], ...texel[tri[2]], sampled, 0]);
    mesh.flatTriangle(corner[tri[0]], corner[tri[1]], corner[tri[2]], color);
  }
}

// Scene object -> { vbo: Float32Array(9 * V) (position, normal, colour), idx: Uint32Array(3 * T) }, the buffers
// RasterBackend._rebuildGeometry uploads (raster.js:148-257): tris (flat normal), then planes (two triangles of
// half-size 20), then spheres (12 x 16 latitude / longitude mesh); quads are ignored as in the reference.
//
// opts.textured goes beyond the reference's raster backend (which draws neither quads nor textures) and adds `uv8`
// to the result: quads are appended after everything else (emitQuad); colours resolve through materials.table when
// the scene has one; uv8 holds one row (u0, v0, u1, v1, u2, v2, flags, 0) per triangle for ar_set_mesh_uv, flags bit
// 0 set where the path tracer would sample the atlas: every `tris` item (sampleTriFetch), a quad half unless its
// three coordinate pairs are all zero.
export function buildGeometry(scene, opts) {
  const textured = !!(opts && opts.textured);
  if (!scene) return { vbo: new Float32Array(0), idx: new Uint32Array(0), uv8: textured ? new Uint16Array(0) : undefined };
  if (scene.mesh && scene.mesh.positions && scene.mesh.indices) return meshToVbo(scene.mesh.positions, scene.mesh.indices, scene.mesh.colors);
  const colorOf = textured ? (prim) => resolveColorTable(scene, prim) : (prim) => resolveColor(scene, prim.mat, prim.m);
  const nested = scene.geometry || {};
  const listOf = (kind) => scene[kind] || nested[kind] || [];
  const mesh = new MeshAssembler();
  for (const t of listOf('tris')) {
    if (textured) mesh.texels([...uvOf(t.uvA), ...uvOf(t.uvB), ...uvOf(t.uvC), 1, 0]);
    mesh.flatTriangle(t.a || [0, 0, 0], t.b || [1, 0, 0], t.c || [0, 1, 0], colorOf(t));
  }
  for (const plane of listOf('planes')) emitPlane(mesh, plane, colorOf(plane));
  for (const sphere of listOf('spheres')) emitSphere(mesh, sphere, colorOf(sphere));
  if (textured) for (const quad of listOf('quads')) emitQuad(mesh, quad, colorOf(quad));
  return mesh.finish(textured);
}

// Typed-array fast path: indexed mesh -> the reference's unshared, flat-shaded layout without any
// per-triangle JS objects.  Normal = normalize(cross(b - a, c - a)) in binary64, (0, 1, 0) when its
// length is <= 1e-8 (raster.js:159-163); colour = per-triangle albedo or (0.8, 0.8, 0.8).
export function meshToVbo(positions, indices, colors) {
  const T = (indices.length / 3) | 0, vbo = new Float32Array(T * 27), idx = new Uint32Array(T * 3);
  for (let t = 0; t < T; t++) {
    const ia = indices[3 * t] * 3, ib = indices[3 * t + 1] * 3, ic = indices[3 * t + 2] * 3;
    const ax = positions[ia], ay = positions[ia + 1], az = positions[ia + 2];
    const ux = positions[ib] - ax, uy = positions[ib + 1] - ay, uz = positions[ib + 2] - az;
    const vx = positions[ic] - ax, vy = positions[ic + 1] - ay, vz = positions[ic + 2] - az;
    let nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const L = Math.sqrt(nx * nx + ny * ny + nz * nz);
    if (L > 1e-8) { nx /= L; ny /= L; nz /= L; } else { nx = 0; ny = 1; nz = 0; }
    const r = colors ? colors[3 * t] : FALLBACK_COLOR[0], g = colors ? colors[3 * t + 1] : FALLBACK_COLOR[1], b = colors ? colors[3 * t + 2] : FALLBACK_COLOR[2];
    for (let k = 0; k < 3; k++) {
      const src = indices[3 * t + k] * 3, o = (3 * t + k) * 9;
      vbo[o] = positions[src]; vbo[o + 1] = positions[src + 1]; vbo[o + 2] = positions[src + 2];
      vbo[o + 3] = nx; vbo[o + 4] = ny; vbo[o + 5] = nz;
      vbo[o + 6] = r; vbo[o + 7] = g; vbo[o + 8] = b;
      idx[3 * t + k] = 3 * t + k;
    }
  }
  return { vbo, idx };
}

// context.js:117-124
export function flipAndCopy(src, dst, w, h) {
  const row = w * 4;
  for (let y = 0; y < h; y++) dst.set(src.subarray((h - 1 - y) * row, (h - 1 - y) * row + row), y * row);
}

export function rampBytes(ramp) {
  const s = String(ramp), out = new Uint8Array(s.length);
  for (let i = 0; i < s.length; i++) out[i] = s.charCodeAt(i) & 0xFF;
  return out;
}

// Modal-smoothing uniforms from config, as AsciiPass derives them (js/ascii_pass.js:149-152, 294-299).
export function modeParams() {
  const k = Math.max(3, (config.ASCII_MODE_KERNEL | 0) || 3);
  return {
    on: !!config.ASCII_MODE_FILTER,
    radius: Math.max(1, Math.min(3, Math.max(1, ((k - 1) / 2) | 0))),
    thresh: Math.max(1, (config.ASCII_MODE_THRESH | 0) || 5),
  };
}

/* -------------------------------- backend ---------------------------------- */
export class HipRasterBackend {
  constructor(opts) {
    this.name = 'hip';
    this._device = (opts && opts.device) | 0;
    this._h = (opts && opts.nativeHandle) || null;      // a member of a group: the group owns the native context
    this._borrowed = !!this._h;
    this._scene = null;
    this._hasGeometry = false;
    this._width = 0; this._height = 0;
    this._lastTopDown = null;   // last frame, row 0 = top
    this._forceDirect = !!(opts && opts.directPath);
    // Beyond the reference's raster backend: draw the scene's quads (two-sided) and apply scene.atlas through the
    // texel coordinates of tris and quads, as the path tracer's primary ray does.  Off by default, which keeps
    // frames identical to the reference's rasterizer; may be set on the instance before setScene.
    this.renderQuads = !!(opts && opts.renderQuads);
    this._pinned = new WeakSet();  // page-locked arrays handed out by allocFramebuffer / allocAsciiPlane / allocDepthPlane
    // Frames enqueued by renderAsync and not yet collected, by staging slot, and per slot the page-locked rgba / ascii /
    // depth planes that stand in for callers' plain arrays.  The GPU works on three frames at a time; a queue of six
    // keeps every one of its three frame streams fed while the host prepares the next frame.
    this._maxInFlight = Math.max(1, Math.min(16, ((opts && opts.framesInFlight) | 0) || 6));
    this._flights = new Array(this._maxInFlight).fill(null);
    this._staging = this._flights.map(() => ({}));
    this._nextFlight = 0;
    this._rampCache = { text: null, bytes: null };
  }

  // Uint8ClampedArray(cols*rows*4) over page-locked host memory.  render() fills such a framebuffer by
  // DMA straight from the GPU (no staging copy, no second copy from the backend's own image); getPixels
  // then serves from it, so it must not be overwritten before the next render if getPixels is used.
  // Plain typed arrays (the reference's fbA / fbB, js/main.js:243-248) keep working as before.
  allocFramebuffer(cols, rows) {
    this._ensure();
    const fb = new Uint8ClampedArray(native().allocHost(this._h, cols * rows * 4));
    this._pinned.add(fb);
    return fb;
  }

  // Page-locked planes for the optional appState.ascii (Uint8Array) / appState.depth (Float32Array) outputs.
  allocAsciiPlane(cols, rows) { this._ensure(); const a = new Uint8Array(native().allocHost(this._h, cols * rows)); this._pinned.add(a); return a; }
  allocDepthPlane(cols, rows) { this._ensure(); const a = new Float32Array(native().allocHost(this._h, cols * rows * 4)); this._pinned.add(a); return a; }

  _ensure() {
    if (this._h) return;
    this._h = native().create(this._device);
    if (this._forceDirect) native().setOption(this._h, AR_OPT_PATH, 1);
  }

  dispose() {
    if (this._h && !this._borrowed) native().destroy(this._h);   // waits for frames in flight
    this._h = null;
    this._lastTopDown = null;
    this._own = null;
    this._hasGeometry = false;
    this._flights = new Array(this._maxInFlight).fill(null);
    this._staging = this._flights.map(() => ({}));
    this._nextFlight = 0;
  }

  setScene(scene) {
    this._scene = scene || null;
    this._ensure();
    const N = native();
    const M = this._scene && this._scene.mesh;
    if (M && M.positions instanceof Float32Array && M.indices instanceof Uint32Array && (!M.colors || M.colors instanceof Float32Array)) {
      // typed-array scene: the unshared flat-shaded vertex buffer (what meshToVbo returns) is built on the GPU
      N.setMeshIndexed(this._h, M.positions, M.indices, M.colors || null);
      N.setTexture(this._h, null, 0, 0);
      this._hasGeometry = M.indices.length >= 3;
      return;
    }
    const g = buildGeometry(this._scene, { textured: this.renderQuads });
    N.setMesh(this._h, g.vbo, g.idx);
    this._hasGeometry = g.idx.length > 0;
    const A = this.renderQuads && this._scene ? this._scene.atlas : null;   // {width, height, pixels}: pathtrace.js:341-355
    if (g.uv8 && g.uv8.length && A && A.pixels && A.width && A.height) {
      N.setMeshUv(this._h, g.uv8);
      N.setTexture(this._h, A.pixels instanceof Uint8Array ? A.pixels : new Uint8Array(A.pixels), A.width | 0, A.height | 0);
    } else {
      N.setTexture(this._h, null, 0, 0);
    }
  }

  // New vertex data for the mesh uploaded by the last setScene, in place (ar_update_positions): a Float32Array of 9 floats
  // per vertex (position, normal, colour - the layout buildGeometry / meshToVbo produce) or 3 (positions only).  For a
  // typed-array scene the vertices are the unshared ones meshToVbo lays out, three per triangle in index order;
  // opts.flatNormals recomputes their face normals on the GPU, opts.resort picks a new spatial order.  A deforming mesh
  // without another setScene - the reference would rebuild and upload everything (raster.js:130-133).
  updatePositions(data, opts) {
    this._ensure();
    if (!(data instanceof Float32Array)) throw new TypeError('updatePositions: Float32Array expected');
    const perVertex = opts && opts.floatsPerVertex ? opts.floatsPerVertex | 0 : (opts && opts.positionsOnly === false ? 9 : 3);
    native().updatePositions(this._h, data, perVertex, ((opts && opts.flatNormals) ? 1 : 0) | ((opts && opts.resort) ? 2 : 0));
  }

  // A model matrix for the mesh (column-major, 16 numbers; null removes it), applied on the GPU from the next frame
  // on: rigid motion of the scene without another setScene.  The reference's rasterizer has none
  // (raster_shader.js:12-13); see ar_set_model.
  setModelMatrix(m) {
    this._ensure();
    native().setModel(this._h, m ? (m instanceof Float32Array ? m : new Float32Array(m)) : null);
  }

  // The per-frame uniforms: camera read live (raster.js:284-286), lights packed from the retained scene object every
  // frame as the reference does (raster.js:307-342: a light moved in place shows in the next frame), ASCII parameters
  // from config.
  _uploadFrameState(w, h, appState) {
    const cam = LiveCamera
      ? { pos: LiveCamera.pos, yaw: LiveCamera.yaw, pitch: LiveCamera.pitch }
      : ((appState && appState.camera) || { pos: { x: 0, y: 0, z: 5 }, yaw: 0, pitch: 0 });
    const m = cameraMatrices(cam, w, h);
    const L = packLights(this._scene);
    const mode = modeParams();
    const rampText = String(config.ASCII_RAMP || '@%#*+=-:. ');
    if (this._rampCache.text !== rampText) this._rampCache = { text: rampText, bytes: rampBytes(rampText) };
    const N = native();
    N.setCamera(this._h, m.proj, m.view);
    N.setLights(this._h, L.ambient, L.dir, L.dirColor, L.plPos, L.plCol);
    N.setAscii(this._h, this._rampCache.bytes, mode.on, mode.radius, mode.thresh);
  }

  render(timeSec, framebuffer, appState) {
    this._ensure();
    const w = ((appState && appState.cols) | 0) || (config.VIRTUAL_GRID_WIDTH | 0) || 1;
    const h = ((appState && appState.rows) | 0) || (config.VIRTUAL_GRID_HEIGHT | 0) || 1;
    this._width = w; this._height = h;
    if (!this._hasGeometry) return;                      // raster.js:274
    this._uploadFrameState(w, h, appState);
    const N = native();
    const depth = (appState && appState.depth && appState.depth.length === w * h) ? appState.depth : null;
    const ascii = (appState && appState.ascii && appState.ascii.length === w * h) ? appState.ascii : null;
    if (framebuffer && framebuffer.length === w * h * 4 && this._pinned.has(framebuffer)) {
      N.render(this._h, w, h, framebuffer, depth, ascii);
      this._lastTopDown = framebuffer;
      return;
    }
    if (!this._own || this._own.length !== w * h * 4) this._own = new Uint8Array(N.allocHost(this._h, w * h * 4));
    this._lastTopDown = this._own;
    N.render(this._h, w, h, this._lastTopDown, depth, ascii);
    if (framebuffer && framebuffer.length === this._lastTopDown.length) framebuffer.set(this._lastTopDown);
  }

  renderRaw(args) {
    if (args && args.framebuffer && args.appState) return this.render(args.time || 0, args.framebuffer, args.appState);
    return undefined;
  }

  // render() split in two, for callers that keep frames in flight (the reference's own loop does: kickGPU's
  // gpuInFlight guard and the fbA / fbB pair, js/main.js:243-248,364-375).  renderAsync reads the camera, uploads the
  // frame's uniforms, enqueues the frame and its copies to the host, and returns a token without waiting; collect
  // waits for that frame and delivers it.  Up to framesInFlight frames (constructor option, default 6) may be
  // outstanding; one more renderAsync throws.
  //   framebuffer  a page-locked array from allocFramebuffer: the GPU writes it directly (keep it untouched until
  //                collect); undefined: the image is staged in the backend's own page-locked memory and copied into
  //                the array given to collect; null: no RGBA read-back at all.
  //   appState.ascii / appState.depth: as for render(); page-locked planes (allocAsciiPlane / allocDepthPlane) are
  //                written directly, plain arrays are filled at collect.
  // Returns null when there is nothing to draw (raster.js:274); collect(null) is a no-op returning false.
  renderAsync(timeSec, appState, framebuffer) {
    this._ensure();
    const w = ((appState && appState.cols) | 0) || (config.VIRTUAL_GRID_WIDTH | 0) || 1;
    const h = ((appState && appState.rows) | 0) || (config.VIRTUAL_GRID_HEIGHT | 0) || 1;
    this._width = w; this._height = h;
    if (!this._hasGeometry) return null;
    const slot = this._nextFlight;
    if (this._flights[slot]) throw new Error('renderAsync: ' + this._maxInFlight + ' frames are in flight; collect the oldest first');
    this._uploadFrameState(w, h, appState);
    const N = native(), S = this._staging[slot];
    const staged = (key, Type, length, bytes) => {
      if (!S[key] || S[key].length !== length) S[key] = new Type(N.allocHost(this._h, bytes));
      return S[key];
    };
    const flight = { slot, w, h, rgba: null, rgbaStaged: false, ascii: null, asciiOut: null, depth: null, depthOut: null, ticket: 0 };
    if (framebuffer === undefined) { flight.rgba = staged('rgba', Uint8Array, w * h * 4, w * h * 4); flight.rgbaStaged = true; }
    else if (framebuffer !== null) {
      if (framebuffer.length !== w * h * 4 || !this._pinned.has(framebuffer)) throw new Error('renderAsync: the framebuffer must come from allocFramebuffer(cols, rows) of this backend (or be undefined / null)');
      flight.rgba = framebuffer;
    }
    const wantAscii = appState && appState.ascii && appState.ascii.length === w * h ? appState.ascii : null;
    if (wantAscii) {
      if (this._pinned.has(wantAscii)) flight.ascii = wantAscii;
      else { flight.ascii = staged('ascii', Uint8Array, w * h, w * h); flight.asciiOut = wantAscii; }
    }
    const wantDepth = appState && appState.depth && appState.depth.length === w * h ? appState.depth : null;
    if (wantDepth) {
      if (this._pinned.has(wantDepth)) flight.depth = wantDepth;
      else { flight.depth = staged('depth', Float32Array, w * h, w * h * 4); flight.depthOut = wantDepth; }
    }
    N.renderAsync(this._h, w, h);
    flight.ticket = N.readAsync(this._h, w, h, flight.rgba, flight.depth, flight.ascii, null);
    this._flights[slot] = flight;
    this._nextFlight = (slot + 1) % this._flights.length;
    return flight;
  }

  // Waits for the frame behind `token` and hands it over: a staged image is copied into `framebuffer` (when its length
  // matches, as render() does), staged ascii / depth planes into the arrays appState named.  True when a frame was
  // delivered.  Tokens are collected in the order they were issued.
  collect(token, framebuffer) {
    if (!token) return false;
    if (this._flights[token.slot] !== token) throw new Error('collect: unknown or already collected token');
    native().wait(this._h, token.ticket);
    this._flights[token.slot] = null;
    if (token.rgba) {
      this._lastTopDown = token.rgba;
      if (token.rgbaStaged && framebuffer && framebuffer.length === token.rgba.length) framebuffer.set(token.rgba);
    }
    if (token.asciiOut) token.asciiOut.set(token.ascii);
    if (token.depthOut) token.depthOut.set(token.depth);
    return true;
  }

  // Several frames that differ in their camera only, handed over at once (ar_render_batch_async): the frames share one
  // chain of kernel launches, which is most of what a small scene's frame costs.  cameras: [{pos: {x, y, z} | [x, y, z],
  // yaw, pitch}, ...] - the values the live camera would have had at each of the frames.  Returns the number of frames
  // enqueued (0 when there is nothing to draw); readBatch(frame, planes) then copies frame `frame` out (it waits).
  renderBatch(cameras, appState) {
    this._ensure();
    const w = ((appState && appState.cols) | 0) || (config.VIRTUAL_GRID_WIDTH | 0) || 1;
    const h = ((appState && appState.rows) | 0) || (config.VIRTUAL_GRID_HEIGHT | 0) || 1;
    this._width = w; this._height = h;
    if (!this._hasGeometry || !cameras || !cameras.length) return 0;
    this._uploadFrameState(w, h, appState);
    const packed = new Float32Array(32 * cameras.length);
    cameras.forEach((c, k) => {
      const p = Array.isArray(c.pos) ? { x: c.pos[0], y: c.pos[1], z: c.pos[2] } : c.pos;
      const m = cameraMatrices({ pos: p, yaw: c.yaw, pitch: c.pitch }, w, h);
      packed.set(m.proj, 32 * k); packed.set(m.view, 32 * k + 16);
    });
    native().renderBatchAsync(this._h, w, h, packed, 0);
    return cameras.length;
  }

  // Planes of frame `frame` of the last renderBatch: {rgba, depth, ascii, ids} typed arrays to fill, any may be absent.
  readBatch(frame, planes) {
    const p = planes || {};
    native().readBatch(this._h, this._width, this._height, frame | 0, p.rgba || null, p.depth || null, p.ascii || null, p.ids || null);
  }

  // True when collect(token) would not wait.
  ready(token) { return !token || native().poll(this._h, token.ticket); }

  // Per-kernel GPU time (HIP events on the frame streams): profile(true) starts it, kernelStats() returns
  // { kernel: { launches, ms } } for the launches since the last call, waiting for the frames in flight.
  profile(on) { this._ensure(); native().setProfiling(this._h, !!on); }
  kernelStats() { this._ensure(); return native().kernelStats(this._h); }

  // Multi-GPU and interop plumbing, as the C ABI offers it (addresses are BigInt): restrict frames to a strip of rows,
  // run them on a caller's HIP stream, store the ASCII plane in a caller's device buffer, hand outputs over in stream
  // order, and read the planes of the last frame on demand.
  setStrip(row0, row1) { this._ensure(); native().setStrip(this._h, row0 | 0, row1 | 0); }
  setStream(stream) { this._ensure(); native().setStream(this._h, stream || null); }
  setAsciiTarget(address, firstRow) { this._ensure(); native().setAsciiTarget(this._h, address || null, firstRow | 0); }
  streamWaitFrame(stream) { this._ensure(); native().streamWaitFrame(this._h, stream); }
  frameWaitStream(stream) { this._ensure(); native().frameWaitStream(this._h, stream); }
  deviceBuffer(which) { this._ensure(); return native().deviceBuffer(this._h, { rgba: 0, depth: 1, ascii: 2 }[which]); }
  sync() { this._ensure(); native().sync(this._h); }
  setIdOutput(on) { this._ensure(); native().setIdOutput(this._h, !!on); }
  // Planes of the most recent frame (any may be null); ids needs setIdOutput(true) before that frame.
  read(planes) {
    this._ensure();
    const p = planes || {};
    native().read(this._h, this._width, this._height, p.rgba || null, p.depth || null, p.ascii || null, p.ids || null);
  }
  // Text colours of the most recent frame (packed r << 16 | g << 8 | b, see AsciiPass.colorsFromBuffer).
  readColors(out, steps, grayscale) { this._ensure(); native().readColors(this._h, this._width, this._height, steps | 0, !!grayscale, out); return out; }

  // The reference keeps the bottom-up readPixels image and flips on request (raster.js:366-375);
  // this backend keeps the top-down image, so flipY = true is the plain copy.
  getPixels(args) {
    const a = args || {};
    if (!this._lastTopDown) return null;
    const w = a.width || this._width, h = a.height || this._height;
    const flipY = a.flipY === undefined ? true : !!a.flipY;
    const bottomUp = () => { const out = new Uint8Array(w * h * 4); flipAndCopy(this._lastTopDown, out, w, h); return out; };
    if (a.framebuffer && a.framebuffer.length >= w * h * 4) {
      if (flipY) a.framebuffer.set(this._lastTopDown);
      else flipAndCopy(this._lastTopDown, a.framebuffer, w, h);
      return a.framebuffer;
    }
    return bottomUp();                                   // raw image as read back: bottom-up
  }

  // The vertex and index buffers as the GPU holds them (what the reference hands to gl.bufferData).
  readMesh() { this._ensure(); return native().readMesh(this._h); }

  frameCounters() {
    this._ensure();
    const c = native().frameCounters(this._h);
    return { triangles: c[0], mediumRecords: c[1], largeRecords: c[2], binEntries: c[3], poolOverflow: c[4], tiled: c[5] };
  }
}

/* ------------------------------ several GPUs -------------------------------- */
// One frame on several GPUs of this machine, from one Node process (ar_group_*): the rows are cut into one strip per
// device, each device renders its strip from its own copy of the scene on a native host thread, and every `batch` frames
// RCCL all-gathers combine the ASCII strips, after which every device holds every frame of the batch.  The reference has
// one GPU behind its router (js/gpu_renderer.js:92-95); this is the build's scaling layer for its hot path.
//   const group = new HipRasterGroup({ devices: [0, 1, 2, 3], batch: 16 });
//   group.setScene(scene);                       // every member, through the ordinary backend
//   group.renderAsync(t, { cols, rows });        // live camera, returns at once
//   group.sync(); group.read(frame)              // Uint8Array(cols * rows) of ASCII codes
export class HipRasterGroup {
  constructor(opts) {
    const devices = (opts && opts.devices) || [0];
    this.batch = Math.max(1, ((opts && opts.batch) | 0) || 16);
    this._g = native().groupCreate(Int32Array.from(devices), this.batch);
    this.members = devices.map((d, i) => new HipRasterBackend({ device: d, nativeHandle: native().groupCtx(this._g, i) }));
    this.cols = 0; this.rows = 0;
  }

  static stripRows(rows, nStrips, strip) { return native().stripRows(rows | 0, nStrips | 0, strip | 0); }

  setScene(scene) { for (const m of this.members) m.setScene(scene); }

  // One frame with the live camera (js/camera.js) on every member; lights and ASCII parameters as render() uploads them.
  renderAsync(timeSec, appState) {
    const w = ((appState && appState.cols) | 0) || (config.VIRTUAL_GRID_WIDTH | 0) || 1;
    const h = ((appState && appState.rows) | 0) || (config.VIRTUAL_GRID_HEIGHT | 0) || 1;
    if (!this.members[0]._hasGeometry) return false;
    if (!this._uniformsSent) {        // (scene-level uniforms: the members are idle now)
      for (const m of this.members) m._uploadFrameState(w, h, appState);
      this._uniformsSent = true;
    }
    const cam = { pos: LiveCamera.pos, yaw: LiveCamera.yaw, pitch: LiveCamera.pitch };
    const mats = cameraMatrices(cam, w, h);
    native().groupRenderAsync(this._g, w, h, mats.proj, mats.view);
    this.cols = w; this.rows = h;
    return true;
  }

  // Gathers what is left of the current batch and waits; lights / ASCII parameters are sent again with the next frame.
  sync() { native().groupSync(this._g); this._uniformsSent = false; }

  get framesGathered() { return native().groupFramesGathered(this._g); }

  // ASCII grid of frame `frame` (0 = oldest) of the last gathered batch as member `member` holds it (call sync() first).
  read(frame, member, out) {
    const grid = out || new Uint8Array(this.cols * this.rows);
    if (grid.length !== this.cols * this.rows) throw new Error('HipRasterGroup.read: array does not match the grid');
    native().groupRead(this._g, member | 0, frame | 0, grid);
    return grid;
  }

  dispose() {
    if (!this._g) return;
    for (const m of this.members) m.dispose();
    native().groupDestroy(this._g);
    this._g = null;
  }
}
