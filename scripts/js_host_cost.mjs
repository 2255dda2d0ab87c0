// Where a JavaScript caller's frame time goes with frames in flight: per-call host cost of the pieces of
// HipRasterBackend.renderAsync / collect at the bench workload.  node scripts/js_host_cost.mjs [tris cols rows]
import path from 'path';
import { fileURLToPath } from 'url';
const here = path.dirname(fileURLToPath(import.meta.url));
const JS = path.join(here, '..', 'ascii-renderer_amd', 'js');
(async () => {
  const geometry = await import(path.join(JS, 'geometry.js'));
  const sceneApi = await import(path.join(JS, 'scene_api.js'));
  const { setCamera, camera } = await import(path.join(JS, 'camera.js'));
  const hb = await import(path.join(JS, 'hip_backend.js'));
  const { native } = await import(path.join(JS, 'native.js'));
  const tris = +(process.argv[2] || 1000000), cols = +(process.argv[3] || 1920), rows = +(process.argv[4] || 1080);
  const backend = new hb.HipRasterBackend();
  backend.setScene(sceneApi.meshScene(geometry.randomMesh(0xA5C11, tris, 1920, 1080, 1.0)));
  setCamera([0, 0, 0], 0, 0);
  const N = native(), h = backend._h;
  const acc = {}, t = (name, fn) => { const s = process.hrtime.bigint(); const r = fn(); acc[name] = (acc[name] || 0) + Number(process.hrtime.bigint() - s); return r; };
  const n = 3000, tickets = [];
  for (let pass = 0; pass < 2; pass++) {
    for (const k of Object.keys(acc)) acc[k] = 0;
    const s = process.hrtime.bigint();
    for (let i = 0; i < n; i++) {
      t('setCamera (JS)', () => setCamera([0, 0, 0], 0.01 * Math.sin(i * 0.1), 0));
      if (tickets.length === 3) t('wait', () => N.wait(h, tickets.shift()));
      const m = t('cameraMatrices (JS)', () => hb.cameraMatrices(camera, cols, rows));
      const L = t('packLights (JS)', () => hb.packLights(backend._scene));
      t('setCamera+setLights+setAscii (addon)', () => { N.setCamera(h, m.proj, m.view); N.setLights(h, L.ambient, L.dir, L.dirColor, L.plPos, L.plCol); N.setAscii(h, hb.rampBytes('@%#*+=-:. '), true, 2, 12); });
      t('renderAsync (addon: five launches)', () => N.renderAsync(h, cols, rows));
      tickets.push(t('readAsync (addon: event record)', () => N.readAsync(h, cols, rows, null, null, null, null)));
    }
    while (tickets.length) N.wait(h, tickets.shift());
    const total = Number(process.hrtime.bigint() - s);
    if (pass === 1) {
      console.log(`${n} frames in flight, ${(total / n / 1e3).toFixed(2)} us per frame (${(1e9 * n / total).toFixed(0)} frames/s); host time per frame by piece:`);
      for (const k of Object.keys(acc)) console.log(`  ${k}: ${(acc[k] / n / 1e3).toFixed(2)} us`);
    }
  }
  backend.dispose();
})().catch((e) => { console.error(e); process.exit(1); });
