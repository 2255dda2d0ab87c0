// gpu_renderer.js - the backend plugin boundary: a table of named backend factories and the facade the
// application renders through.
//
// Exports and behaviour follow the reference's router contract (js/gpu_renderer.js:52-111):
//   registerBackend(name, factory)   stored under the lower-cased name; bad arguments throw
//   listBackends() / getBackend()    names in registration order / name of the live backend
//   setBackend(name)                 unknown names throw; the old backend's dispose() and the replay of the last
//                                    scene on the new one may fail without the switch failing; factory() gets no
//                                    arguments; returns the canonical name
//   setScene(scene)                  remembered (null for a falsy scene) and handed to the live backend
//   renderScene(timeSec, framebuffer, appState) / renderRaw(args) / getPixels(args)
//   disposeGPU()                     drops the live backend; the remembered scene stays
// Any call that needs a backend while none is live starts the default one first.
// What differs from the reference: the table starts with the HIP rasterizer ('hip'; 'raster' and the reference's
// aliases for it lead there, as does 'gpu') and the HIP ray tracer ('raytrace', aliases 'rt' and 'ray'); 'hip' is
// the default; there is no path tracer.
import { HipRasterBackend } from './hip_backend.js';
import { HipRaytraceBackend } from './hip_raytrace_backend.js';

class BackendRouter {
  constructor(defaultName) {
    this.defaultName = defaultName;
    this.entries = new Map();     // canonical name -> { make, aliases: Set }
    this.live = null;             // { name, backend } while a backend is up
    this.scene = null;            // the scene every new backend is brought up with
  }

  add(name, make, aliases) {
    this.entries.set(name, { make, aliases: new Set(aliases || []) });
    return this;
  }

  // canonical name for a name or alias, or undefined
  resolve(name) {
    const wanted = String(name || '').toLowerCase();
    if (this.entries.has(wanted)) return wanted;
    for (const [canonical, entry] of this.entries) if (entry.aliases.has(wanted)) return canonical;
    return undefined;
  }

  names() { return [...this.entries.keys()]; }

  // the live backend, starting the default one when there is none
  backend() {
    if (!this.live) this.switchTo(this.defaultName);
    return this.live.backend;
  }

  // calls backend[method](...args) when the backend has such a method
  call(backend, method, args, quiet) {
    if (!backend || typeof backend[method] !== 'function') return undefined;
    if (!quiet) return backend[method](...args);
    try { return backend[method](...args); } catch (err) { return undefined; }
  }

  switchTo(name) {
    const canonical = this.resolve(name);
    if (canonical === undefined) throw new Error(`Unknown backend "${name}". Known: ${this.names().join(', ')}`);
    if (this.live) this.call(this.live.backend, 'dispose', [], true);
    const backend = this.entries.get(canonical).make();
    this.live = { name: canonical, backend };
    if (this.scene) this.call(backend, 'setScene', [this.scene], true);
    return canonical;
  }

  shutDown() {
    if (this.live) this.call(this.live.backend, 'dispose', [], false);
    this.live = null;
  }
}

const router = new BackendRouter('hip')
  .add('hip', () => new HipRasterBackend(), ['raster', 'r', 'rasterizer', 'gpu'])
  .add('raytrace', () => new HipRaytraceBackend(), ['rt', 'ray']);

export function registerBackend(name, factory) {
  if (!name || typeof factory !== 'function') throw new Error('registerBackend(name, factory): invalid args');
  const key = String(name).toLowerCase(), known = router.entries.get(key);
  router.add(key, factory, known ? known.aliases : []);   // a replaced backend keeps the aliases that led to it
}

export const listBackends = () => router.names();

export function getBackend() {
  router.backend();
  return router.live.name;
}

export const setBackend = (name) => router.switchTo(name);

export function setScene(scene) {
  router.scene = scene || null;
  router.call(router.backend(), 'setScene', [router.scene], false);
}

// (render is the one method every backend must have: called without a check, as the reference does)
export const renderScene = (timeSec, framebuffer, appState) => router.backend().render(timeSec, framebuffer, appState);

export const renderRaw = (args) => router.call(router.backend(), 'renderRaw', [args], false);

export const getPixels = (args) => router.call(router.backend(), 'getPixels', [args], false);

export const disposeGPU = () => router.shutDown();

// Not in the reference: the live backend object, for hosts that use what a backend offers beyond the five-method
// contract (HipRasterBackend.renderAsync / collect, allocFramebuffer, kernelStats ...).
export const activeBackend = () => router.backend();
