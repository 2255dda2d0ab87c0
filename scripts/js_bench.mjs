// Frame rate of the drop-in path as a JavaScript caller sees it: router -> HipRasterBackend.render ->
// N-API -> HIP kernels -> readback into the caller's framebuffer, blocking, one frame at a time
// (the reference's render loop, js/main.js:364-375).  node scripts/js_bench.mjs [tris cols rows]
import path from 'path';
import { fileURLToPath } from 'url';
const here = path.dirname(fileURLToPath(import.meta.url));
const JS = path.join(here, '..', 'ascii-renderer_amd', 'js');
(async () => {
  const geometry = await import(path.join(JS, 'geometry.js'));
  const sceneApi = await import(path.join(JS, 'scene_api.js'));
  const { setCamera } = await import(path.join(JS, 'camera.js'));
  const { HipRasterBackend } = await import(path.join(JS, 'hip_backend.js'));
  const tris = +(process.argv[2] || 1000000), cols = +(process.argv[3] || 1920), rows = +(process.argv[4] || 1080);
  const t0 = Date.now();
  const mesh = geometry.randomMesh(0xA5C11, tris, 1920, 1080, 1.0);
  const backend = new HipRasterBackend();
  backend.setScene(sceneApi.meshScene(mesh));
  setCamera({ x: 0, y: 0, z: 0 }, 0, 0);
  console.log(`scene of ${tris} triangles built and uploaded in ${Date.now() - t0} ms`);
  const plain = new Uint8ClampedArray(cols * rows * 4), pinned = backend.allocFramebuffer(cols, rows);
  const ascii = new Uint8Array(cols * rows);
  for (const [label, fb, state] of [['plain framebuffer', plain, { cols, rows }], ['page-locked framebuffer', pinned, { cols, rows }],
    ['page-locked framebuffer + ascii plane', pinned, { cols, rows, ascii }],
    ['page-locked framebuffer + page-locked ascii and depth planes', pinned, { cols, rows, ascii: backend.allocAsciiPlane(cols, rows), depth: backend.allocDepthPlane(cols, rows) }]]) {
    for (let i = 0; i < 5; i++) backend.render(0, fb, state);
    const n = 100, s = process.hrtime.bigint();
    for (let i = 0; i < n; i++) backend.render(i / 60, fb, state);
    const dt = Number(process.hrtime.bigint() - s) / 1e9;
    console.log(`${label}: ${(n / dt).toFixed(1)} frames/s (${(1e3 * dt / n).toFixed(3)} ms/frame)`);
  }
  backend.dispose();
})().catch((e) => { console.error(e); process.exit(1); });
