// config.js - the one mutable configuration object, with the keys the raster / ASCII path reads.
// Same names and defaults as the reference's js/config.js:7-61 (keys of unrelated subsystems - path
// tracer sampling, ripples, PI digits - are not carried).
const SCREEN_SCALE = 4;
const FONT_HEIGHT_WIDTH_RATIO = 1.5;

export const config = {
  TARGET_FPS: 60,
  DEFAULT_BACKEND: 'hip',
  USE_GRAYSCALE: false,

  // index 0 = darkest. (The reference declares this key twice, js/config.js:14-15; the later wins.)
  ASCII_RAMP: '@%#*+=-:. ',

  VIRTUAL_GRID_WIDTH: 16 * SCREEN_SCALE * FONT_HEIGHT_WIDTH_RATIO, // 96
  VIRTUAL_GRID_HEIGHT: 9 * SCREEN_SCALE,                           // 36

  // modal (majority) smoothing of glyphs: js/config.js:26-34
  ASCII_MODE_FILTER: true,
  ASCII_MODE_KERNEL: 5,
  ASCII_MODE_THRESH: 5 * 5 * 0.5,

  CAMERA: { FOVY_DEG: 80 },

  // PIXEL_ASPECT (cell width / cell height) is filled in at run time by the application
  // (js/main.js:217); unset means 1.0 (js/render/backends/raster.js:293-295).
  PATH_TRACER: {},
};
