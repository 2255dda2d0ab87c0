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

export function setCamera(pos, yaw, pitch) {
  camera.pos.x = +pos[0]; camera.pos.y = +pos[1]; camera.pos.z = +pos[2];
  camera.yaw = +yaw || 0;
  camera.pitch = +pitch || 0;
  return camera;
}
