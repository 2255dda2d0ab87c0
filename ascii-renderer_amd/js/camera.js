// camera.js - the live camera singleton the backend reads at render time (reference js/camera.js:7-13
// and js/render/backends/raster.js:284-286).  yaw = pitch = 0 looks along +X.  Input handling
// (updateCamera) belongs to the application and is not part of the raster path.
export const camera = {
  pos: { x: 0, y: 0, z: 5 },
  yaw: 0,
  pitch: 0,
  speed: 2.5,
  sensitivity: 1.5,
};

// pos: [x, y, z] or {x, y, z} (the shape of camera.pos itself)
export function setCamera(pos, yaw, pitch) {
  const p = Array.isArray(pos) || ArrayBuffer.isView(pos) ? pos : [pos.x, pos.y, pos.z];
  if (!(Number.isFinite(+p[0]) && Number.isFinite(+p[1]) && Number.isFinite(+p[2]))) throw new TypeError('setCamera: position must be [x, y, z] or {x, y, z} of finite numbers');
  camera.pos.x = +p[0]; camera.pos.y = +p[1]; camera.pos.z = +p[2];
  camera.yaw = +yaw || 0;
  camera.pitch = +pitch || 0;
  return camera;
}
