// CPU-side checks of the JavaScript host (no GPU, the addon is never asked to create a context).
// Run by tests/test_js_host.py:  node tests/js/test_host.mjs <golden dir> <out json>
import fs from 'fs';
import path from 'path';
import crypto from 'crypto';
import assert from 'assert';
import { fileURLToPath } from 'url';

const here = path.dirname(fileURLToPath(import.meta.url));
const JS = path.join(here, '..', '..', 'ascii-renderer_amd', 'js');
const goldenDir = process.argv[2], outPath = process.argv[3];

const f32 = (b64) => { const b = Buffer.from(b64 || '', 'base64'); return new Float32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.length)); };
const u32 = (b64) => { const b = Buffer.from(b64 || '', 'base64'); return new Uint32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.length)); };
const bytesEqual = (a, b) => a.length === b.length && Buffer.compare(Buffer.from(a.buffer, a.byteOffset, a.byteLength), Buffer.from(b.buffer, b.byteOffset, b.byteLength)) === 0;
const sha = (ta) => crypto.createHash('sha256').update(Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength)).digest('hex');

(async () => {
  const { config } = await import(path.join(JS, 'config.js'));
  const { camera, setCamera } = await import(path.join(JS, 'camera.js'));
  const hb = await import(path.join(JS, 'hip_backend.js'));
  const router = await import(path.join(JS, 'gpu_renderer.js'));
  const sceneApi = await import(path.join(JS, 'scene_api.js'));
  const geometry = await import(path.join(JS, 'geometry.js'));
  const asciiPass = await import(path.join(JS, 'ascii_pass.js'));
  const results = { checks: 0 };
  const ok = (cond, msg) => { assert.ok(cond, msg); results.checks++; };

  /* ---- host half vs values recorded from the reference's RasterBackend ---- */
  const ref = JSON.parse(fs.readFileSync(path.join(goldenDir, 'ref_raster_host.json'), 'utf8'));
  for (const c of ref.cases) {
    const scene = ref.scenes[c.scene];
    const g = hb.buildGeometry(scene);
    ok(bytesEqual(g.vbo, f32(c.vbo_f32)), 'VBO bytes differ: ' + c.scene);
    ok(bytesEqual(g.idx, u32(c.ibo_u32)), 'IBO bytes differ: ' + c.scene);
    if (!c.uniforms.uProj) continue;
    if (c.pixelAspect === null) delete config.PATH_TRACER.PIXEL_ASPECT; else config.PATH_TRACER.PIXEL_ASPECT = c.pixelAspect;
    setCamera(c.camera.pos, c.camera.yaw, c.camera.pitch);
    const m = hb.cameraMatrices(camera, c.cols, c.rows);
    ok(bytesEqual(m.proj, f32(c.uniforms.uProj)), 'uProj differs: ' + c.scene + '/' + c.cam_name);
    ok(bytesEqual(m.view, f32(c.uniforms.uView)), 'uView differs: ' + c.scene + '/' + c.cam_name);
    const L = hb.packLights(scene), n = c.uniforms.uPLCount;
    ok(bytesEqual(L.ambient, f32(c.uniforms.uAmbient)) && bytesEqual(L.dir, f32(c.uniforms.uLightDir)) && bytesEqual(L.dirColor, f32(c.uniforms.uLightColor)), 'light uniforms differ: ' + c.scene);
    ok(L.plPos.length === 3 * n && bytesEqual(L.plPos, f32(c.uniforms['uPLPos[0]']).subarray(0, 3 * n)) && bytesEqual(L.plCol, f32(c.uniforms['uPLCol[0]']).subarray(0, 3 * n)), 'point lights differ: ' + c.scene);
  }
  delete config.PATH_TRACER.PIXEL_ASPECT;

  /* ---- SceneBuilder vs objects recorded from the reference's builder ---- */
  const sbRef = JSON.parse(fs.readFileSync(path.join(goldenDir, 'ref_scene_builder.json'), 'utf8'));
  assert.deepStrictEqual(Object.assign({}, sceneApi.MaterialIds), sbRef.material_ids); results.checks++;
  const sb = sceneApi.createSceneBuilder();
  sb.addMesh({ positions: [0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0.5], indices: [0, 1, 2, 2, 1, 3, 0, 9, 1], uvs: [0, 0, 5, 0, 0, 7, 70000, -3], materialId: sceneApi.MaterialIds.RED });
  sb.addMesh({ positions: [0, 0, 1, 1, 0, 1, 0, 1, 1, 5, 5, 5, 6, 5, 5, 5, 6, 5], materialId: 99 });
  sb.addTriangle([0, 0, 2], [1, 0, 2], [0, 1, 2], sceneApi.MaterialIds.GREEN, [1, 2], [3, 4], [5, 6]);
  assert.deepStrictEqual(JSON.parse(JSON.stringify(sb.toObject())), sbRef.add_mesh); results.checks++;
  const capped = sceneApi.createSceneBuilder(1, 2, 1);
  for (let i = 0; i < 4; i++) { capped.addTriangle([i, 0, 0], [i + 1, 0, 0], [i, 1, 0]); capped.addSphere([i, 0, 0], 1); capped.addQuad([0, 0, i], [1, 0, i], [1, 1, i], [0, 1, i]); }
  assert.deepStrictEqual(JSON.parse(JSON.stringify(capped.toObject())), sbRef.capped); results.checks++;
  assert.deepStrictEqual(JSON.parse(JSON.stringify(sceneApi.fromObject(sbRef.add_mesh).toObject())), sbRef.round_trip); results.checks++;
  assert.throws(() => sb.addTriangle([0, 0], [1, 0, 0], [0, 1, 0]), /addTriangle: bad args/); results.checks++;
  // the reference's default scene rebuilt through this builder gives the same raster geometry
  const rebuilt = sceneApi.fromObject(sbRef.default_scene).toObject();
  const gDefault = hb.buildGeometry(rebuilt), cDefault = ref.cases.find((c) => c.scene === 'default_scene');
  ok(bytesEqual(gDefault.vbo, f32(cDefault.vbo_f32)) && gDefault.idx.length === 2304, 'default scene through the builder');
  // typed-array meshes go through addMesh too
  const cubeMesh = geometry.cube(0.5);
  ok(sceneApi.createSceneBuilder().addMesh(cubeMesh).toObject().geometry.tris.length === 12, 'addMesh with typed arrays');

  /* ---- quads + texel coordinates (beyond the reference's rasterizer) vs the path tracer's recorded uploads ---- */
  const uvRef = JSON.parse(fs.readFileSync(path.join(goldenDir, 'ref_pathtrace_uv.json'), 'utf8'));
  results.textured = {};
  for (const c of uvRef.cases) {
    const plain = hb.buildGeometry(c.scene), g = hb.buildGeometry(c.scene, { textured: true });
    ok(plain.uv8 === undefined && g.uv8.length === (g.idx.length / 3) * 8, 'uv8 shape: ' + c.name);
    const nT = c.uniforms.uNumTris, nQ = c.uniforms.uNumQuads, first = plain.idx.length / 3;
    ok(g.idx.length / 3 === first + 4 * nQ && bytesEqual(g.idx.subarray(0, plain.idx.length), plain.idx), 'quads are appended: ' + c.name);
    for (let t = 0; t < nT; t++) {
      const want = ['A', 'B', 'C'].flatMap((k) => Array.from(u32(c.uniforms['uTriUV' + k + '[0]']).subarray(2 * t, 2 * t + 2)));
      assert.deepStrictEqual(Array.from(g.uv8.subarray(8 * t, 8 * t + 8)), [...want, 1, 0]);
    }
    results.checks += nT;
    for (let q = 0; q < nQ; q++) {
      const u = [0, 1, 2, 3].map((k) => Array.from(u32(c.uniforms['uQuadUV' + k + '[0]']).subarray(2 * q, 2 * q + 2)));
      const any = (...v) => (v.flat().some((x) => x !== 0) ? 1 : 0);
      const want = [[...u[0], ...u[1], ...u[2], any(u[0], u[1], u[2]), 0], [...u[0], ...u[2], ...u[3], any(u[0], u[2], u[3]), 0],
        [...u[0], ...u[2], ...u[1], any(u[0], u[1], u[2]), 0], [...u[0], ...u[3], ...u[2], any(u[0], u[2], u[3]), 0]];
      for (let k = 0; k < 4; k++) assert.deepStrictEqual(Array.from(g.uv8.subarray(8 * (first + 4 * q + k), 8 * (first + 4 * q + k) + 8)), want[k]);
    }
    results.checks += nQ;
    results.textured[c.name] = { uv8: sha(g.uv8), idx: sha(g.idx), verts: g.vbo.length / 9 };
  }

  /* ---- ray backend host half vs the uniforms recorded from the reference's RaytraceBackend ---- */
  const rt = await import(path.join(JS, 'hip_raytrace_backend.js'));
  const rtRef = JSON.parse(fs.readFileSync(path.join(goldenDir, 'ref_raytrace_uniforms.json'), 'utf8'));
  const i32 = (b64) => { const b = Buffer.from(b64 || '', 'base64'); return new Int32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.length)); };
  assert.deepStrictEqual(rt.raytraceLimits(), rtRef.cases[0].limits); results.checks++;
  for (const c of rtRef.cases) {
    const P = rt.packRaytraceScene(rtRef.scenes[c.scene], c.limits), U = c.uniforms, tag = ' (' + c.scene + ')';
    const counts = { uNumS: P.sphereCount, uNumP: P.planeCount, uNumT: P.triCount, uNumMats: P.materialCount, uNumPL: P.pointCount, uNumDL: P.dirCount };
    for (const [k, v] of Object.entries(counts)) ok(U[k] === v, k + tag);
    const arrays = [['uSpheres[0]', P.sphereGeom, P.sphereCount, f32], ['uSm[0]', P.sphereMaterial, P.sphereCount, i32],
      ['uPlanes[0]', P.planeGeom, P.planeCount, f32], ['uPm[0]', P.planeMaterial, P.planeCount, i32],
      ['uTa[0]', P.triA, P.triCount, f32], ['uTb[0]', P.triB, P.triCount, f32], ['uTc[0]', P.triC, P.triCount, f32], ['uTm[0]', P.triMaterial, P.triCount, i32],
      ['uMatAlbedo[0]', P.materialAlbedo, P.materialCount, f32], ['uMatReflective[0]', P.materialMirror, P.materialCount, i32],
      ['uPLPos[0]', P.pointPos, P.pointCount, f32], ['uPLCol[0]', P.pointColor, P.pointCount, f32],
      ['uDLDir[0]', P.dirDir, P.dirCount, f32], ['uDLCol[0]', P.dirColor, P.dirCount, f32]];
    for (const [name, got, n, dec] of arrays) {
      if (n > 0) ok(bytesEqual(got, dec(U[name])), name + tag);       // the reference uploads an array only when its count is positive
      else ok(U[name] === undefined, name + ' must not be uploaded' + tag);
    }
    ok(bytesEqual(P.environment, f32(U.uEnv)), 'uEnv' + tag);
    if (c.pixelAspect === null) delete config.PATH_TRACER.PIXEL_ASPECT; else config.PATH_TRACER.PIXEL_ASPECT = c.pixelAspect;
    setCamera(c.camera.pos, c.camera.yaw, c.camera.pitch);
    const F = rt.raytraceFrameUniforms();
    ok(bytesEqual(F.camPos, f32(U.uCamPos)) && Math.fround(F.yaw) === f32(U.uYaw)[0] && Math.fround(F.pitch) === f32(U.uPitch)[0] &&
      Math.fround(F.fovY) === f32(U.uFovY)[0] && Math.fround(F.pixAspect) === f32(U.uPixAspect)[0], 'frame uniforms' + tag);
  }
  delete config.PATH_TRACER.PIXEL_ASPECT;

  /* ---- router semantics (js/gpu_renderer.js:52-111 of the reference) ---- */
  assert.throws(() => router.registerBackend('', () => ({})), /registerBackend\(name, factory\): invalid args/);
  assert.throws(() => router.registerBackend('x', 42), /invalid args/);
  assert.throws(() => router.setBackend('nope'), /Unknown backend "nope"/);
  const log = [];
  const mk = (tag) => () => ({ name: tag, setScene(s) { log.push(tag + ':setScene:' + (s ? s.tag : 'null')); }, render(t, fb, st) { log.push(tag + ':render:' + st.cols); return undefined; }, dispose() { log.push(tag + ':dispose'); throw new Error('swallowed'); } });
  router.registerBackend('FakeA', mk('a'));
  router.registerBackend('fakeb', mk('b'));
  ok(router.listBackends().includes('fakea') && router.listBackends().includes('hip'), 'names are lower-cased');
  ok(router.listBackends().includes('raytrace'), 'the ray backend is registered under the reference name');
  ok(router.setBackend('FAKEA') === 'fakea' && router.getBackend() === 'fakea', 'setBackend returns the canonical name');
  router.setScene({ tag: 's1' });
  router.renderScene(0, null, { cols: 7, rows: 3 });
  router.setBackend('fakeb');                            // disposes a (error swallowed), replays s1 on b
  ok(router.renderRaw({}) === undefined && router.getPixels({}) === undefined, 'optional methods may be missing');
  assert.deepStrictEqual(log, ['a:setScene:s1', 'a:render:7', 'a:dispose', 'b:setScene:s1']); results.checks++;
  assert.throws(() => router.disposeGPU(), /swallowed/);  // disposeGPU does not swallow (reference: no try)
  for (const alias of ['raster', 'r', 'rasterizer']) ok(true, alias);

  /* ---- ASCII pass surface ---- */
  const atlas = asciiPass.atlasDescriptor(8, 16);
  assert.deepStrictEqual(atlas, { tileW: 12, tileH: 20, tilesPerRow: 16, count: 256, pad: 2, rows: 16 }); results.checks++;
  assert.deepStrictEqual(asciiPass.atlasTile(atlas, 65), { x: 12, y: 80 }); results.checks++;
  const mp = hb.modeParams();
  assert.deepStrictEqual(mp, { on: true, radius: 2, thresh: 12 }); results.checks++;
  const pass = new asciiPass.AsciiPass({ ramp: 'ab' });
  ok(pass.ramp === 'ab' && pass.cols === 96 && pass.rows === 36 && pass.modeRadius === 2 && pass.modeThresh === 12, 'AsciiPass defaults');
  pass.blit(new Uint8Array(5 * 4 * 4), 5, 4);
  ok(pass.cols === 5 && pass.rows === 4 && pass.codes.length === 20, 'blit auto-resizes');

  /* ---- geometry twins: hashes compared with the Python generators by the caller ---- */
  const ico = geometry.icosphere(3, 1.0), rm = geometry.randomMesh(0xA5C11, 5000, 1920, 1080, 1.0);
  const rnd = geometry.mulberry32(12345);
  results.hashes = {
    cube_pos: sha(cubeMesh.positions), cube_idx: sha(cubeMesh.indices),
    ico_pos: sha(ico.positions), ico_idx: sha(ico.indices),
    rm_pos: sha(rm.positions), rm_col: sha(rm.colors),
    rm_vbo: sha(hb.meshToVbo(rm.positions, rm.indices, rm.colors).vbo),
    ico_vbo: sha(hb.meshToVbo(ico.positions, ico.indices, null).vbo),
  };
  results.mulberry = [rnd(), rnd(), rnd()];
  fs.writeFileSync(outPath, JSON.stringify(results));
  console.log('js host checks passed:', results.checks);
})().catch((e) => { console.error(e); process.exit(1); });
