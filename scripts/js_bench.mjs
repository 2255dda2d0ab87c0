// Frame rate of the drop-in path as a JavaScript caller sees it: router -> HipRasterBackend.render ->
// N-API -> HIP kernels -> readback into the caller's framebuffer, blocking, one frame at a time
// (the reference's render loop, js/main.js:364-375); then the same with frames in flight
// (renderAsync / collect, three and six outstanding), a new camera every frame.  node scripts/js_bench.mjs [tris cols rows]
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
  setCamera([0, 0, 0], 0, 0);
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
  // frames in flight: enqueue, and collect the oldest once three are outstanding
  const inFlight = (label, make, n, depth) => {
    const run = (count) => {
      const q = [];
      for (let i = 0; i < count; i++) {
        setCamera([0, 0, 0], 0.01 * Math.sin(i * 0.1), 0);
        if (q.length === (depth || 6)) { const p = q.shift(); backend.collect(p.token, p.fb); }
        q.push(make(i));
      }
      while (q.length) { const p = q.shift(); backend.collect(p.token, p.fb); }
    };
    run(30);
    const s = process.hrtime.bigint();
    run(n);
    const dt = Number(process.hrtime.bigint() - s) / 1e9;
    console.log(`${depth || 6} in flight, ${label}: ${(n / dt).toFixed(1)} frames/s (${(1e3 * dt / n).toFixed(4)} ms/frame)`);
  };
  const pinnedFbs = [0, 1, 2, 3, 4, 5].map(() => backend.allocFramebuffer(cols, rows));
  const pinnedAscii = [0, 1, 2, 3, 4, 5].map(() => backend.allocAsciiPlane(cols, rows));
  const plainFb = new Uint8ClampedArray(cols * rows * 4), plainAscii = new Uint8Array(cols * rows);
  inFlight('outputs resident (no read-back)', (i) => ({ token: backend.renderAsync(i / 60, { cols, rows }, null), fb: null }), 3000, 3);
  inFlight('outputs resident (no read-back)', (i) => ({ token: backend.renderAsync(i / 60, { cols, rows }, null), fb: null }), 3000);
  inFlight('page-locked ascii plane only', (i) => ({ token: backend.renderAsync(i / 60, { cols, rows, ascii: pinnedAscii[i % 6] }, null), fb: null }), 1500);
  inFlight('page-locked framebuffer + ascii plane', (i) => ({ token: backend.renderAsync(i / 60, { cols, rows, ascii: pinnedAscii[i % 6] }, pinnedFbs[i % 6]), fb: null }), 600);
  inFlight('plain ascii plane only (staged, copied at collect)', (i) => ({ token: backend.renderAsync(i / 60, { cols, rows, ascii: plainAscii }, null), fb: null }), 1000);
  inFlight('plain framebuffer + ascii plane (staged, copied at collect)', (i) => ({ token: backend.renderAsync(i / 60, { cols, rows, ascii: plainAscii }), fb: plainFb }), 300);
  // per-kernel GPU time as the addon reports it (HIP events), frames in flight
  backend.profile(true);
  backend.kernelStats();
  inFlight('outputs resident, per-kernel events on', (i) => ({ token: backend.renderAsync(i / 60, { cols, rows }, null), fb: null }), 600);
  const ks = backend.kernelStats();
  backend.profile(false);
  for (const name of Object.keys(ks)) console.log(`  ${name}: ${ks[name].launches} launches, ${(1e3 * ks[name].ms / ks[name].launches).toFixed(2)} us each`);
  backend.dispose();
})().catch((e) => { console.error(e); process.exit(1); });
