// gpu_renderer.js - backend plugin registry and the stable render facade.
//
// Same exports and semantics as the reference's router (js/gpu_renderer.js:52-111):
//   registerBackend(name, factory)  name lower-cased; throws on bad arguments
//   listBackends() / getBackend() / setBackend(name)   setBackend disposes the previous backend
//       (errors swallowed), calls factory() with no arguments and replays the last scene
//   setScene(scene) / renderScene(timeSec, framebuffer, appState) / renderRaw(args) /
//   getPixels(args) / disposeGPU()
// Differences: the built-in registry holds the HIP raster backend ('hip') and the HIP ray backend ('raytrace',
// aliases 'rt' and 'ray' as in the reference); 'raster' and the reference's aliases for it resolve to 'hip';
// the default backend is 'hip'; there is no path tracer.
import { HipRasterBackend } from './hip_backend.js';
import { HipRaytraceBackend } from './hip_raytrace_backend.js';

const _registry = new Map([
  ['hip', () => new HipRasterBackend()],
  ['raytrace', () => new HipRaytraceBackend()],
]);

const _alias = new Map([
  ['raster', 'hip'], ['r', 'hip'], ['rasterizer', 'hip'], ['gpu', 'hip'],
  ['rt', 'raytrace'], ['ray', 'raytrace'],
]);

let _active = null;
let _activeName = null;
let _lastScene = null;

function _canonical(name) {
  const n = String(name || '').toLowerCase();
  if (_registry.has(n)) return n;
  const a = _alias.get(n);
  return _registry.has(a) ? a : null;
}

function _ensureActive() {
  if (_active) return;
  setBackend('hip');
}

export function registerBackend(name, factory) {
  if (!name || typeof factory !== 'function') {
    throw new Error('registerBackend(name, factory): invalid args');
  }
  _registry.set(String(name).toLowerCase(), factory);
}

export function listBackends() {
  return Array.from(_registry.keys());
}

export function getBackend() {
  _ensureActive();
  return _activeName;
}

export function setBackend(name) {
  const key = _canonical(name);
  if (!key) throw new Error('Unknown backend "' + name + '". Known: ' + listBackends().join(', '));
  if (_active && _active.dispose) {
    try { _active.dispose(); } catch (e) { /* as the reference: ignored */ }
  }
  _active = _registry.get(key)();
  _activeName = key;
  if (_lastScene && _active && _active.setScene) {
    try { _active.setScene(_lastScene); } catch (e) { /* as the reference: ignored */ }
  }
  return _activeName;
}

export function setScene(scene) {
  _lastScene = scene || null;
  _ensureActive();
  if (_active && _active.setScene) _active.setScene(_lastScene);
}

export function renderScene(timeSec, framebuffer, appState) {
  _ensureActive();
  return _active.render(timeSec, framebuffer, appState);
}

export function renderRaw(args) {
  _ensureActive();
  return _active.renderRaw ? _active.renderRaw(args) : undefined;
}

export function getPixels(args) {
  _ensureActive();
  return _active.getPixels ? _active.getPixels(args) : undefined;
}

export function disposeGPU() {
  if (_active && _active.dispose) _active.dispose();
  _active = null;
  _activeName = null;
}
