// Generates ref_pathtrace_uv.json by IMPORTING the reference's path-tracer backend (read-only tree at
// /root/reference) behind a recording fake WebGL2 context and writing down what it uploads for the
// atlas and the per-vertex texel coordinates.  Runs only in the build container; no reference source
// text is copied.
//
//   cd tests/golden && node --harmony-optional-chaining --harmony-nullish \
//        --experimental-loader ./ref_loader.mjs gen_pathtrace_uv.mjs
//
// Recorded (data only): the scene objects (SceneBuilder.toObject of js/scene.js's createScene and of a
// builder scene with textured triangles and quads, js/render/scene_api.js:139-236), the synthetic atlas
// attached to them, and from TracerPass.uploadScene / PathtraceBackend._ensureAtlasTexture
// (js/render/backends/pathtrace.js:115-250,341-355): uNumTris / uNumQuads, uAtlasSize, the uTriUV* and
// uQuadUV* integer uniforms and the texImage2D arguments.
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

globalThis.WebGL2RenderingContext = function WebGL2RenderingContext() {};
function makeFakeGL2(log) {
  return new Proxy(Object.create(globalThis.WebGL2RenderingContext.prototype), {
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

// atlas with every texel class: alpha 0 (clear), 1 (solid), 32..126 (glyph), and values the shader treats as "no texture"
function syntheticAtlas(w, h, seed) {
  const rnd = mulberry32(seed), px = new Uint8Array(w * h * 4);
  for (let i = 0; i < w * h; i++) {
    const u = rnd();
    const a = u < 0.2 ? 0 : u < 0.5 ? 1 : u < 0.9 ? 32 + ((rnd() * 95) | 0) : [2, 31, 127, 200, 255][(rnd() * 5) | 0];
    px[4 * i] = (rnd() * 256) | 0; px[4 * i + 1] = (rnd() * 256) | 0; px[4 * i + 2] = (rnd() * 256) | 0; px[4 * i + 3] = a;
  }
  return px;
}

function capture(PathtraceBackend, scene, cols, rows) {
  const log = [];
  globalThis.document = { createElement: () => ({ getContext: () => makeFakeGL2(log) }) };
  const be = new PathtraceBackend();
  be.setScene(scene);
  be.render(0, new Uint8ClampedArray(cols * rows * 4), { cols, rows });
  const out = { uniforms: {}, tex_uploads: [] };
  for (const { fn, args } of log) {
    if (fn === 'uniform2uiv' && /UV/.test(String(args[0]))) out.uniforms[args[0]] = b64(new Uint32Array(args[1]));
    else if (fn === 'uniform1i' && /^uNum/.test(String(args[0]))) out.uniforms[args[0]] = args[1];
    else if (fn === 'uniform2i' && args[0] === 'uAtlasSize') out.uniforms.uAtlasSize = [args[1], args[2]];
    else if (fn === 'texImage2D' && args.length === 9 && args[8]) out.tex_uploads.push({ width: args[3], height: args[4], format: args[6], type: args[7], data: b64(new Uint8Array(args[8])) });
  }
  be.dispose();
  return out;
}

(async () => {
  const { createScene } = await import(REF + 'scene.js');
  const sceneApi = await import(REF + 'render/scene_api.js');
  const { PathtraceBackend } = await import(REF + 'render/backends/pathtrace.js');

  const out = { note: 'texel-coordinate and atlas uploads captured from the reference PathtraceBackend through a recording fake WebGL2 context', cases: [] };

  // 1. the default scene with a 32x32 atlas attached the way js/main.js:226-231 does
  {
    const scene = createScene();
    const pixels = syntheticAtlas(32, 32, 0xA71A5);
    scene.atlas = { width: 32, height: 32, pixels };
    const cap = capture(PathtraceBackend, scene, 96, 36);
    const plain = JSON.parse(JSON.stringify({ ...scene, atlas: { width: 32, height: 32 } }));
    out.cases.push({ name: 'default_scene_atlas32', scene: plain, atlas_rgba: b64(pixels), ...cap });
  }
  // 2. builder scene: textured triangles (addTriangle, addMesh with uvs incl. out-of-range values) and quads with
  //    zero, partly zero and full texel coordinates
  {
    const sb = sceneApi.createSceneBuilder();
    sb.setCameraPose([0, 1, 4], { yaw: -Math.PI / 2, pitch: 0 });
    sb.setTextureAtlasSize(20, 12);
    sb.addTriangle([-2, 0, 0], [0, 0, 0], [-1, 2, 0], sceneApi.MaterialIds.GREEN, [0, 0], [19, 0], [10, 11]);
    sb.addTriangle([0, 0, 0.5], [2, 0, 0.5], [1, 2, 0.5], sceneApi.MaterialIds.RED);                    // default uvs (0,0)
    sb.addTriangle([0, 2, -1], [2, 2, -1], [1, 3, -1], 99, [70000, -3], [2.9, 7.2], [65535, 65536]);    // clamped by _u16
    sb.addMesh({ positions: [-1, -1, 1, 1, -1, 1, 1, 1, 1, -1, 1, 1], indices: [0, 1, 2, 0, 2, 3], uvs: [0, 11, 19, 11, 19, 0, 0, 0], materialId: sceneApi.MaterialIds.WHITE });
    sb.addMesh({ positions: [3, 0, 0, 4, 0, 0, 3, 1, 0], indices: [0, 1, 2], uvs: [5, 5], materialId: 2 }); // short uvs array
    sb.addQuad([-3, 0, -2], [3, 0, -2], [3, 3, -2], [-3, 3, -2], sceneApi.MaterialIds.WHITE, [0, 12], [20, 12], [20, 0], [0, 0]);
    sb.addQuad([-3, 0, -3], [3, 0, -3], [3, 3, -3], [-3, 3, -3], sceneApi.MaterialIds.RED);               // all zero: never sampled
    sb.addQuad([-5, 0, 2], [-5, 0, -2], [-5, 3, -2], [-5, 3, 2], sceneApi.MaterialIds.GREEN, [0, 0], [0, 0], [0, 0], [7, 3]); // first half zero
    sb.addQuad([5, 0, 2], [5, 0, -2], [5, 3, -2], [5, 3, 2], sceneApi.MaterialIds.MIRROR, [4, 4], [0, 0], [0, 0], [0, 0]);
    sb.addRect([-8, 0, -8], [8, 0, -8], [8, 0, 8], [-8, 0, 8], sceneApi.MaterialIds.WHITE, [0, 0], [19, 0], [19, 11], [0, 11]);
    sb.addSphere([0, 1, 0], 0.75, sceneApi.MaterialIds.GLASS);
    const scene = sb.toObject();
    const pixels = syntheticAtlas(20, 12, 0xBEEF);
    scene.atlas = { width: 20, height: 12, pixels };
    const cap = capture(PathtraceBackend, scene, 80, 40);
    const plain = JSON.parse(JSON.stringify({ ...scene, atlas: { width: 20, height: 12 } }));
    out.cases.push({ name: 'builder_textured', scene: plain, atlas_rgba: b64(pixels), ...cap });
  }
  // 3. a scene without an atlas: uAtlasSize falls back to 1 x 1 (texturing off)
  {
    const scene = createScene();
    const cap = capture(PathtraceBackend, scene, 96, 36);
    out.cases.push({ name: 'default_scene_no_atlas', scene: JSON.parse(JSON.stringify(scene)), atlas_rgba: null, ...cap });
  }
  fs.writeFileSync('ref_pathtrace_uv.json', JSON.stringify(out));
  for (const c of out.cases) console.log(c.name, Object.keys(c.uniforms).join(' '), c.uniforms.uAtlasSize, c.tex_uploads.map((t) => t.width + 'x' + t.height).join(','));
})().catch((e) => { console.error(e); process.exit(1); });
