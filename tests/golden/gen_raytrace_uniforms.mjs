// Generates ref_raytrace_uniforms.json by IMPORTING the reference's ray-trace backend (read-only tree at
// /root/reference) behind a recording fake WebGL context and writing down every uniform it uploads for a
// frame: scene packing (js/render/backends/raytrace.js:140-312) and camera / frame uniforms (:330-389).
// Runs only in the build container; no reference source text is copied.
//
//   cd tests/golden && node --harmony-optional-chaining --harmony-nullish \
//        --experimental-loader ./ref_loader.mjs gen_raytrace_uniforms.mjs
import fs from 'fs';

const REF = '/root/reference/js/';
const b64 = (ta) => Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength).toString('base64');

function mulberry32(a) {
  return function () {
    a |= 0; a = (a + 0x6D2B79F5) | 0;
    let t = Math.imul(a ^ (a >>> 15), 1 | a);
    t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t;
    return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
  };
}

function makeFakeGL(log) {
  return new Proxy({}, {
    get(_, prop) {
      if (typeof prop !== 'string') return undefined;
      if (/^[A-Z0-9_]+$/.test(prop)) return prop;
      return (...args) => {
        log.push({ fn: prop, args });
        switch (prop) {
          case 'getShaderParameter': case 'getProgramParameter': return true;
          case 'checkFramebufferStatus': return 'FRAMEBUFFER_COMPLETE';
          case 'getAttribLocation': return 0;
          case 'getUniformLocation': return args[1];
          case 'getExtension': return {};
          case 'isContextLost': return false;
          default:
            if (prop.startsWith('create')) return { kind: prop };
            return undefined;
        }
      };
    },
  });
}

function capture(RaytraceBackend, camera, config, scene, cam, cols, rows, pixelAspect) {
  const log = [];
  globalThis.document = { createElement: () => ({ getContext: () => makeFakeGL(log) }) };
  if (pixelAspect === null) delete config.PATH_TRACER.PIXEL_ASPECT; else config.PATH_TRACER.PIXEL_ASPECT = pixelAspect;
  camera.pos.x = cam.pos[0]; camera.pos.y = cam.pos[1]; camera.pos.z = cam.pos[2];
  camera.yaw = cam.yaw; camera.pitch = cam.pitch;
  const be = new RaytraceBackend();
  be.setScene(scene);
  be.render(0, new Uint8ClampedArray(cols * rows * 4), { cols, rows });
  const out = { cols, rows, pixelAspect, camera: cam, limits: be._limits, uniforms: {} };
  for (const { fn, args } of log) {
    if (fn === 'uniform1i') out.uniforms[args[0]] = args[1];
    else if (fn === 'uniform1f') out.uniforms[args[0]] = b64(new Float32Array([args[1]]));
    else if (fn === 'uniform3f') out.uniforms[args[0]] = b64(new Float32Array([args[1], args[2], args[3]]));
    else if (fn === 'uniform3fv' || fn === 'uniform4fv') out.uniforms[args[0]] = b64(new Float32Array(args[1]));
    else if (fn === 'uniform1iv') out.uniforms[args[0]] = b64(new Int32Array(args[1]));
  }
  be.dispose();
  return out;
}

(async () => {
  const { config } = await import(REF + 'config.js');
  const { camera } = await import(REF + 'camera.js');
  const { createScene } = await import(REF + 'scene.js');
  const { RaytraceBackend } = await import(REF + 'render/backends/raytrace.js');

  const rnd = mulberry32(0x7ACE);
  const r3 = (s) => [(rnd() * 2 - 1) * s, (rnd() * 2 - 1) * s, (rnd() * 2 - 1) * s];
  // unified scene: materials array (diffuse and mirrors), every primitive kind, point and directional lights, environment
  const unified = {
    materials: [
      { albedo: [0.8, 0.8, 0.8] }, { albedo: [0.9, 0.2, 0.2] }, { albedo: [0.2, 0.9, 0.3], reflective: false },
      { albedo: [0.95, 0.95, 1.0], reflective: true }, { albedo: [1.5, -0.1, 0.5], reflective: 1 }, { name: 'no-albedo' },
    ],
    geometry: { spheres: [], planes: [], tris: [] },
    lights: {
      env: { color: [0.3, 0.4, 0.6], intensity: 0.5 },
      points: [{ p: [2, 4, 1], color: [1, 0.9, 0.8], intensity: 12 }, { p: [-3, 2, -2], color: [0.4, 0.5, 1.0], intensity: 8 }, { intensity: 1 }],
      directionals: [{ dir: [0.3, -1, 0.2], color: [1, 1, 0.9], intensity: 0.8 }, { dir: [-1, -0.5, 0], intensity: 0.3 }, { dir: [0, -1, 0], intensity: 9 }],
    },
  };
  for (let i = 0; i < 9; i++) unified.geometry.spheres.push({ p: r3(4).map((v, k) => (k === 1 ? Math.abs(v) * 0.5 + 0.4 : v)), r: 0.3 + rnd() * 0.7, mat: i % 6 });
  unified.geometry.spheres.push({ r: 0.5 });                       // default centre, mat undefined -> 0
  unified.geometry.planes.push({ n: [0, 1, 0], d: 0, mat: 0 }, { n: [0, 0, 1], d: 6, mat: 2 }, { n: [1, 0, 0], d: 7, mat: 3 }, { mat: 1, d: -9 });
  for (let i = 0; i < 20; i++) { const c = r3(3.5); c[1] = Math.abs(c[1]) + 0.2; unified.geometry.tris.push({ a: [c[0] + rnd(), c[1], c[2] + rnd()], b: [c[0] - rnd(), c[1] + rnd(), c[2]], c: [c[0], c[1] + 1 + rnd(), c[2] - rnd()], mat: (i * 7) % 8 }); }
  unified.geometry.tris.push({ mat: 3 });                          // all defaults -> zero triangle
  // more primitives than the limits allow (32 spheres, 64 triangles)
  const crowded = { materials: [{ albedo: [0.7, 0.7, 0.7] }, { albedo: [1, 1, 1], reflective: true }], geometry: { spheres: [], tris: [] }, lights: { points: [{ p: [0, 5, 0], color: [1, 1, 1], intensity: 20 }] } };
  for (let i = 0; i < 40; i++) crowded.geometry.spheres.push({ p: [(i % 8) - 3.5, 0.4, ((i / 8) | 0) - 2], r: 0.35, mat: i % 2 });
  for (let i = 0; i < 70; i++) crowded.geometry.tris.push({ a: [i * 0.2 - 7, 0, -4], b: [i * 0.2 - 6.8, 0, -4], c: [i * 0.2 - 6.9, 1.5, -4], mat: 0 });
  // legacy shape (no geometry / materials / lights keys): adapted by _resolveUnified
  const legacy = {
    spheres: [{ p: [0, 1, 0], r: 1, m: 6 }, { p: [2, 0.5, 1], r: 0.5, m: 3 }, { p: [-2, 0.5, -1], r: 0.5 }],
    planes: [{ p: [0, 1, 0, 0], m: 1 }, { p: [0, 0, 1, 5], m: 2 }],
    tris: [{ a: [-1, 0, 2], b: [1, 0, 2], c: [0, 2, 2], m: 2 }, { a: [3, 0, 0], b: [3, 0, 2], c: [3, 2, 1], m: 7 }],
    envLight: { color: [0.2, 0.3, 0.5], intensity: 1.5 },
    dirLight: { dir: [0.4, -1, 0.3], color: [1, 0.95, 0.9], intensity: 1.2 },
  };
  const scenes = { unified, crowded, legacy, default_scene: createScene(), empty: {} };
  const cams = [
    { name: 'front', cam: { pos: [0, 1.5, 6], yaw: -Math.PI / 2, pitch: -0.1 }, cols: 96, rows: 36, pa: 0.5 },
    { name: 'orbit', cam: { pos: [5, 3, -4], yaw: 2.4, pitch: -0.35 }, cols: 160, rows: 60, pa: null },
    { name: 'zenith', cam: { pos: [0.5, 9, 0.25], yaw: 0.3, pitch: -Math.PI / 2 }, cols: 64, rows: 48, pa: 0.4817 },
  ];
  const out = { note: 'uniforms captured from the reference RaytraceBackend through a recording fake WebGL context', scenes, cases: [] };
  for (const [sname, scene] of Object.entries(scenes)) {
    for (const c of cams) {
      if (sname !== 'unified' && c.name !== 'front') continue;
      const cap = capture(RaytraceBackend, camera, config, scene, c.cam, c.cols, c.rows, c.pa);
      cap.scene = sname; cap.cam_name = c.name;
      out.cases.push(cap);
    }
  }
  fs.writeFileSync('ref_raytrace_uniforms.json', JSON.stringify(out));
  for (const c of out.cases) console.log(c.scene, c.cam_name, JSON.stringify(c.limits), Object.keys(c.uniforms).length, c.uniforms.uNumS, c.uniforms.uNumP, c.uniforms.uNumT, c.uniforms.uNumMats, c.uniforms.uNumPL, c.uniforms.uNumDL);
})().catch((e) => { console.error(e); process.exit(1); });
