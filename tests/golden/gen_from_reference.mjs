// Generates the golden fixtures in this directory by IMPORTING the reference's own CPU-side
// code (read-only tree at /root/reference) and recording what it computes. Runs only in the
// build container; nothing here is needed at test time and no reference source text is copied.
//
//   cd tests/golden && node --harmony-optional-chaining --harmony-nullish \
//        --experimental-loader ./ref_loader.mjs gen_from_reference.mjs
//
// What is recorded (all data, no code):
//   ref_raster_host.json  - for several scenes/cameras: the interleaved VBO + IBO that
//                           RasterBackend._rebuildGeometry uploads (js/render/backends/raster.js:148-266)
//                           and the uniforms RasterBackend.render sets (raster.js:284-342).
//   ref_quantize.json     - compositeAndQuantize chars (js/renderer.js:21-99) for exhaustive and
//                           random RGBA grids, several ramps, with and without UI overrides.
//   ref_scene_builder.json- SceneBuilder.addMesh / toObject outputs (js/render/scene_api.js:169-236)
//                           and the default createScene() object (js/scene.js).
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

/* ---- recording fake WebGL: constants echo their name, methods are logged ---- */
function makeFakeGL(log) {
  const target = {};
  return new Proxy(target, {
    get(_, prop) {
      if (typeof prop !== 'string') return undefined;
      if (/^[A-Z0-9_]+$/.test(prop)) return prop;
      return (...args) => {
        log.push({ fn: prop, args });
        switch (prop) {
          case 'getShaderParameter': case 'getProgramParameter': return true;
          case 'checkFramebufferStatus': return 'FRAMEBUFFER_COMPLETE';
          case 'getAttribLocation': return ['a_pos', 'a_nrm', 'a_col'].indexOf(args[1]);
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

function captureRaster(RasterBackend, camera, config, scene, cam, cols, rows, pixelAspect) {
  const log = [];
  globalThis.document = { createElement: () => ({ getContext: () => makeFakeGL(log) }) };
  if (pixelAspect === null) delete config.PATH_TRACER.PIXEL_ASPECT;
  else config.PATH_TRACER.PIXEL_ASPECT = pixelAspect;
  camera.pos.x = cam.pos[0]; camera.pos.y = cam.pos[1]; camera.pos.z = cam.pos[2];
  camera.yaw = cam.yaw; camera.pitch = cam.pitch;
  const be = new RasterBackend();
  be.setScene(scene);
  const fb = new Uint8ClampedArray(cols * rows * 4);
  be.render(0, fb, { cols, rows });
  const out = { cols, rows, pixelAspect, camera: cam, uniforms: {} };
  let bound = null;
  for (const { fn, args } of log) {
    if (fn === 'bindBuffer') bound = args[0];
    else if (fn === 'bufferData') {
      if (args[0] === 'ARRAY_BUFFER') out.vbo_f32 = b64(args[1]);
      else if (args[0] === 'ELEMENT_ARRAY_BUFFER') { out.ibo_u32 = b64(args[1]); out.index_ctor = args[1].constructor.name; }
    } else if (fn === 'uniformMatrix4fv') out.uniforms[args[0]] = b64(new Float32Array(args[2]));
    else if (fn === 'uniform3f') out.uniforms[args[0]] = b64(new Float32Array([args[1], args[2], args[3]]));
    else if (fn === 'uniform3fv') out.uniforms[args[0]] = b64(new Float32Array(args[1]));
    else if (fn === 'uniform1i') out.uniforms[args[0]] = args[1];
    else if (fn === 'drawElements') out.draw = { mode: args[0], count: args[1], type: args[2], offset: args[3] };
    else if (fn === 'enable') (out.enabled = out.enabled || []).push(args[0]);
    else if (fn === 'cullFace') out.cullFace = args[0];
    else if (fn === 'clearColor') out.clearColor = args;
    else if (fn === 'viewport') out.viewport = args;
    else if (fn === 'renderbufferStorage') out.depthFormat = args[1];
  }
  be.dispose();
  return out;
}

(async () => {
  const { config } = await import(REF + 'config.js');
  const { camera } = await import(REF + 'camera.js');
  const { createScene } = await import(REF + 'scene.js');
  const sceneApi = await import(REF + 'render/scene_api.js');
  const { compositeAndQuantize } = await import(REF + 'renderer.js');
  const { RasterBackend } = await import(REF + 'render/backends/raster.js');

  /* ------------------------------ raster host half ------------------------------ */
  const rnd = mulberry32(0xC0FFEE);
  const r3 = (s) => [(rnd() * 2 - 1) * s, (rnd() * 2 - 1) * s, (rnd() * 2 - 1) * s];
  const legacyTris = [];
  for (let i = 0; i < 24; i++) {
    const c = r3(4);
    const t = { a: [c[0] + rnd(), c[1] + rnd(), c[2] + rnd()], b: [c[0] - rnd(), c[1] + rnd(), c[2]], c: [c[0], c[1] - rnd(), c[2] + rnd()] };
    if (i % 4 === 0) t.mat = i % 3; else if (i % 4 === 1) t.m = [0, 1, 2, 3, 6, 5][i % 6]; else if (i % 4 === 2) { t.mat = 1; t.m = 2; }
    legacyTris.push(t);
  }
  legacyTris.push({ a: [1, 1, 1], b: [1, 1, 1], c: [1, 1, 1], mat: 0 });          // degenerate -> normal (0,1,0)
  legacyTris.push({ a: [0, 0, 0], b: [1e-5, 0, 0], c: [0, 1e-5, 0], m: 3 });      // |N| = 1e-10 <= 1e-8
  legacyTris.push({ mat: 2 });                                                     // default a/b/c
  const pts = [];
  for (let i = 0; i < 10; i++) pts.push({ p: r3(6), color: [rnd(), rnd(), rnd()], intensity: 5 + 20 * rnd() });
  pts[3] = { intensity: 2 };            // default p / color
  pts[5] = { p: [1, 2, 3], color: [1, 1, 1] }; // intensity missing -> 0
  const scenes = {
    default_scene: createScene(),
    legacy_full: {
      materials: [{ albedo: [0.9, 0.1, 0.2] }, { albedo: [1.5, -0.2, 0.5] }, { name: 'no-albedo' }],
      tris: legacyTris,
      planes: [{ n: [0, 1, 0], d: 0, m: 1 }, { n: [0.3, 0.2, 0.9], d: -2, mat: 1 }, { n: [0, -2, 0.01], d: 3 }, {}],
      spheres: [{ p: [1, 2, 3], r: 0.5, m: 3 }, { mat: 0 }, { p: [-2, 0.5, 1], r: 1.25, mat: 7 }],
      lights: {
        env: { color: [0.2, 0.3, 0.4], intensity: 0.5 },
        directionals: [{ dir: [-0.3, -0.8, 0.5], color: [1, 0.9, 0.8], intensity: 1.5 }, { dir: [1, 0, 0], intensity: 9 }],
        points: pts,
      },
    },
    legacy_lights_old_style: {
      tris: legacyTris.slice(0, 5),
      envLight: { color: [0.05, 0.06, 0.07] },
      dirLight: { dir: [0.1, -0.9, -0.2], color: [0.8, 0.7, 0.6] },
    },
    legacy_dir_no_color: {
      tris: legacyTris.slice(0, 3),
      lights: { directionals: [{ dir: [0, -1, 0], color: [0, 0.5, 1], intensity: 2 }], points: [{ p: [0, 3, 0], color: [1, 1, 1], intensity: 10 }] },
      geometry: { tris: [{ a: [9, 9, 9], b: [8, 9, 9], c: [9, 8, 9] }] }, // ignored: scene.tris wins
    },
    geometry_only: { geometry: { tris: legacyTris.slice(5, 9), spheres: [{ p: [0, 0, 0], r: 2, m: 6 }] } },
    empty: {},
  };
  const cams = [
    { name: 'scene_cam', cam: { pos: [0, 1.5, 6], yaw: 0, pitch: 0 }, cols: 96, rows: 36, pa: null },
    { name: 'orbit_a', cam: { pos: [2.5, 1.0, -1.75], yaw: 2.5, pitch: -0.3 }, cols: 80, rows: 40, pa: 0.5 },
    { name: 'steep', cam: { pos: [-4, 7, 0.25], yaw: -1.1, pitch: 1.4707963267948966 }, cols: 640, rows: 360, pa: 0.4817 },
    { name: 'hd', cam: { pos: [0, 0, 0], yaw: 0, pitch: 0 }, cols: 1920, rows: 1080, pa: 1.0 },
  ];
  const rasterOut = { note: 'captured from reference RasterBackend through a recording fake WebGL context', cases: [] };
  for (const [sname, scene] of Object.entries(scenes)) {
    for (const c of cams) {
      if (sname !== 'default_scene' && sname !== 'legacy_full' && c.name !== 'orbit_a') continue;
      const cap = captureRaster(RasterBackend, camera, config, scene, c.cam, c.cols, c.rows, c.pa);
      cap.scene = sname; cap.cam_name = c.name;
      rasterOut.cases.push(cap);
    }
  }
  rasterOut.scenes = scenes;
  fs.writeFileSync('ref_raster_host.json', JSON.stringify(rasterOut));

  /* --------------------------------- quantizer --------------------------------- */
  const ramps = ['@%#*+=-:. ', ' .:-=+*#%@', 'ab', 'X', '@&#+=~-;:",.',
    ' .\'`^",:;Il!i><~+_-?][}{1)(|\\/tfjrxnuvczXYUJCLQ0OZmwqpdbkhao*#MW&8%B@$'];
  const quant = { note: 'compositeAndQuantize outputs from the reference', cases: [] };
  function runQuant(ramp, cols, rows, rgba, ui) {
    config.ASCII_RAMP = ramp;
    const fb = new Uint8ClampedArray(rgba);
    const uiBuffer = [];
    for (let y = 0; y < rows; y++) { const row = []; for (let x = 0; x < cols; x++) row.push(ui ? ui[y * cols + x] : null); uiBuffer.push(row); }
    const st = { cols, rows, framebuffer: fb, uiBuffer, dataBuffer_chars: new Uint16Array(cols * rows), dataBuffer_colors: new Uint32Array(cols * rows) };
    compositeAndQuantize(st);
    return { chars: st.dataBuffer_chars, colors: st.dataBuffer_colors, alpha_after: fb.filter((_, i) => i % 4 === 3) };
  }
  // exhaustive: every sum 0..765 (three different splits of each sum)
  {
    const cols = 766, rows = 3;
    const rgba = new Uint8Array(cols * rows * 4);
    for (let s = 0; s <= 765; s++) {
      const r0 = Math.min(255, s), g0 = Math.min(255, s - r0), b0 = s - r0 - g0;
      const b1 = Math.min(255, s), g1 = Math.min(255, s - b1), r1 = s - b1 - g1;
      const g2 = Math.floor(s / 3), r2 = Math.floor((s - g2) / 2), b2 = s - g2 - r2;
      const put = (row, r, g, b, a) => { const p = (row * cols + s) * 4; rgba[p] = r; rgba[p + 1] = g; rgba[p + 2] = b; rgba[p + 3] = a; };
      put(0, r0, g0, b0, 255); put(1, r1, g1, b1, 0); put(2, r2, g2, b2, 1);
    }
    for (const ramp of ramps) {
      const o = runQuant(ramp, cols, rows, rgba, null);
      quant.cases.push({ name: 'exhaustive_sums', ramp, cols, rows, rgba: b64(rgba), chars_u16: b64(o.chars), colors_u32: b64(o.colors) });
    }
  }
  // random grids with UI overrides
  {
    const rq = mulberry32(12345);
    for (const [cols, rows] of [[80, 40], [96, 36], [7, 5], [1, 1]]) {
      const rgba = new Uint8Array(cols * rows * 4);
      for (let i = 0; i < rgba.length; i++) rgba[i] = (rq() * 256) | 0;
      const ui = [];
      for (let i = 0; i < cols * rows; i++) { const u = rq(); ui.push(u < 0.15 ? String.fromCharCode(32 + ((rq() * 95) | 0)) : null); }
      for (const ramp of ramps.slice(0, 3)) {
        const o = runQuant(ramp, cols, rows, rgba, ui);
        quant.cases.push({ name: 'random_ui', ramp, cols, rows, rgba: b64(rgba), ui: ui.map((c) => (c === null ? 0 : c.charCodeAt(0))), chars_u16: b64(o.chars), colors_u32: b64(o.colors), alpha_after_u8: b64(new Uint8Array(o.alpha_after)) });
      }
    }
  }
  config.ASCII_RAMP = '@%#*+=-:. ';
  fs.writeFileSync('ref_quantize.json', JSON.stringify(quant));

  /* ------------------------------- scene builder ------------------------------- */
  const sb = sceneApi.createSceneBuilder();
  sb.addMesh({ positions: [0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0.5], indices: [0, 1, 2, 2, 1, 3, 0, 9, 1], uvs: [0, 0, 5, 0, 0, 7, 70000, -3], materialId: sceneApi.MaterialIds.RED });
  sb.addMesh({ positions: [0, 0, 1, 1, 0, 1, 0, 1, 1, 5, 5, 5, 6, 5, 5, 5, 6, 5], materialId: 99 });
  sb.addTriangle([0, 0, 2], [1, 0, 2], [0, 1, 2], sceneApi.MaterialIds.GREEN, [1, 2], [3, 4], [5, 6]);
  const capped = sceneApi.createSceneBuilder(1, 2, 1);
  for (let i = 0; i < 4; i++) { capped.addTriangle([i, 0, 0], [i + 1, 0, 0], [i, 1, 0]); capped.addSphere([i, 0, 0], 1); capped.addQuad([0, 0, i], [1, 0, i], [1, 1, i], [0, 1, i]); }
  const sbOut = {
    note: 'SceneBuilder outputs from the reference',
    material_ids: sceneApi.MaterialIds,
    add_mesh: sb.toObject(),
    capped: capped.toObject(),
    default_scene: createScene(),
    round_trip: sceneApi.fromObject(sb.toObject()).toObject(),
  };
  fs.writeFileSync('ref_scene_builder.json', JSON.stringify(sbOut));
  console.log('wrote fixtures:', rasterOut.cases.length, 'raster cases,', quant.cases.length, 'quantize cases');
})().catch((e) => { console.error(e); process.exit(1); });
