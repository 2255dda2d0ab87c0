// ascii_pass.js - RGBA grid -> ASCII-code grid (== glyph-atlas tile indices), on the GPU.
//
// API surface of the reference's AsciiPass (js/ascii_pass.js:130-344): constructor(canvas?, opts),
// setGridSize, blit, draw, drawFromBuffer, setCellSize, and the atlas descriptor of buildAtlas
// ({tileW, tileH, tilesPerRow, count, pad, rows}, ascii_pass.js:20-86).  draw() computes the index half of the
// reference's fragment shader (js/ascii_pass_shader.js:140-188): alpha 2..254 is a literal code, otherwise ramp
// quantisation with optional modal smoothing, then ramp index -> ASCII code; the result is `codes`, one byte per
// cell, where the reference would look up tile (code % 16, floor(code / 16)).  present() runs the shader's other half
// as well (lines 190-230: glyph texels composited onto a canvas-sized RGBA image) with an atlas the caller supplies -
// baking one needs a font rasterizer, which is the only part left out.  colorsFromBuffer() gives the text colours of
// the reference's DOM presentation.
import { config } from './config.js';
import { native } from './native.js';
import { rampBytes } from './hip_backend.js';

export function atlasDescriptor(cellW, cellH) {
  const count = 256, pad = 2, tilesPerRow = Math.ceil(Math.sqrt(count));
  return { tileW: cellW + pad * 2, tileH: cellH + pad * 2, tilesPerRow, count, pad, rows: Math.ceil(count / tilesPerRow) };
}

// Tile origin (in atlas pixels, before padding) of ASCII code `code`: ascii_pass.js:56-60.
export function atlasTile(atlas, code) {
  return { x: (code % atlas.tilesPerRow) * atlas.tileW, y: Math.floor(code / atlas.tilesPerRow) * atlas.tileH };
}

export class AsciiPass {
  constructor(arg1, maybeOpts) {
    // the reference accepts (canvas, opts) or (opts); a canvas has no meaning here and is ignored
    const isOpts = arg1 && typeof arg1 === 'object' && typeof arg1.getContext !== 'function';
    const opts = (isOpts ? arg1 : maybeOpts) || {};
    this.canvas = isOpts ? null : (arg1 || null);
    this.ramp = String(opts.ramp !== undefined && opts.ramp !== null ? opts.ramp
      : (config.ASCII_RAMP !== undefined && config.ASCII_RAMP !== null ? config.ASCII_RAMP : '@%#*+=-:. '));
    this.grayscaleText = !!opts.grayscaleText;
    this.alphaGamma = Number.isFinite(opts.alphaGamma) ? opts.alphaGamma : 1.20;
    this.alphaBias = Number.isFinite(opts.alphaBias) ? opts.alphaBias : 0.045;
    this.transparentBackground = opts.transparentBackground !== undefined ? !!opts.transparentBackground : true;
    // not in the reference: draw() through the library's fixed staging array (ar_quantize_static), for hosts that
    // must not cause device allocations once set up; grids beyond 1920 x 1080 cells then throw
    this.noAlloc = !!opts.noAlloc;
    // modal smoothing controls come from config at construction (ascii_pass.js:149-152)
    this.modeFilter = !!config.ASCII_MODE_FILTER;
    const k = Math.max(3, (config.ASCII_MODE_KERNEL | 0) || 3);
    this.modeRadius = Math.max(1, ((k - 1) / 2) | 0);
    this.modeThresh = Math.max(1, (config.ASCII_MODE_THRESH | 0) || 5);
    this.cols = (config.VIRTUAL_GRID_WIDTH | 0) || 1;
    this.rows = (config.VIRTUAL_GRID_HEIGHT | 0) || 1;
    this.cwCSS = Number.isFinite(opts.cellWidth) ? opts.cellWidth : 8;    // the reference measures a DOM element
    this.chCSS = Number.isFinite(opts.cellHeight) ? opts.cellHeight : 16;
    this.dpr = 1;
    this.cellWDevice = Math.max(1, Math.round(this.cwCSS * this.dpr));
    this.cellHDevice = Math.max(1, Math.round(this.chCSS * this.dpr));
    this._atlas = atlasDescriptor(this.cellWDevice, this.cellHDevice);
    this._device = (opts.device | 0) || 0;
    this._h = null;
    this._src = null;
    this.codes = new Uint8Array(this.cols * this.rows);
  }

  get atlas() { return this._atlas; }

  _ensure() {
    if (!this._h) this._h = native().create(this._device);
  }

  dispose() {
    if (this._h) native().destroy(this._h);
    this._h = null;
  }

  setGridSize(cols, rows) {
    const c = cols | 0, r = rows | 0;
    if (c <= 0 || r <= 0) return;
    if (c === this.cols && r === this.rows) return;
    this.cols = c; this.rows = r;
    this.codes = new Uint8Array(c * r);
    this._src = null;
  }

  drawFromBuffer(rgbaBuffer, srcCols, srcRows) {
    this.blit(rgbaBuffer, srcCols, srcRows);
    return this.draw();
  }

  // The whole pass, pixels included: RGBA8 image (Uint8ClampedArray, cols*cellW x rows*cellH, row 0 = top) of
  // the grid drawn with the caller's glyph atlas - what the reference's draw() leaves on its canvas
  // (js/ascii_pass_shader.js:190-230) with this.alphaGamma, this.grayscaleText and this.transparentBackground.
  // atlasAlpha is the alpha plane of the atlas buildAtlas would bake (ascii_pass.js:20-60) for
  // this.cellWDevice x this.cellHDevice cells: 16 x 16 tiles of (cellW + 4) x (cellH + 4) bytes.
  present(rgbaBuffer, srcCols, srcRows, atlasAlpha) {
    this.blit(rgbaBuffer, srcCols, srcRows);
    if (!this._src) return null;
    if (this._src.length !== this.cols * this.rows * 4) throw new Error('AsciiPass.present: buffer does not match the grid');
    const a = this._atlas;
    if (!atlasAlpha || atlasAlpha.length !== a.count * a.tileW * a.tileH) throw new Error('AsciiPass.present: atlas does not match the cell size');
    this._ensure();
    const N = native();
    N.setAscii(this._h, rampBytes(this.ramp), this.modeFilter, Math.max(1, Math.min(3, this.modeRadius | 0)), this.modeThresh | 0);
    const n = this.cols * this.cellWDevice * this.rows * this.cellHDevice * 4;
    if (!this.image || this.image.length !== n) this.image = new Uint8ClampedArray(n);
    N.present(this._h, this._src, this.cols, this.rows, atlasAlpha, this.cellWDevice, this.cellHDevice, this.alphaGamma,
      !!this.grayscaleText, !!this.transparentBackground, this.image);
    return this.image;
  }

  // Text colour per cell, packed (r << 16) | (g << 8) | b, as the reference's DOM presentation computes it
  // (compositeAndQuantize, js/renderer.js:74-84): channels snapped to max(3, ramp length) levels so that
  // every colour is one of the stylesheet's classes (renderer.js:162-185); UI cells (alpha 2..254) and,
  // with opts.grayscaleText, all cells are black.  Returns this.colors (Uint32Array cols*rows).
  colorsFromBuffer(rgbaBuffer, srcCols, srcRows) {
    this.blit(rgbaBuffer, srcCols, srcRows);
    if (!this._src) return this.colors;
    if (this._src.length !== this.cols * this.rows * 4) throw new Error('AsciiPass.colorsFromBuffer: buffer does not match the grid');
    this._ensure();
    const N = native();
    N.setAscii(this._h, rampBytes(this.ramp), this.modeFilter, Math.max(1, Math.min(3, this.modeRadius | 0)), this.modeThresh | 0);
    if (!this.colors || this.colors.length !== this.cols * this.rows) this.colors = new Uint32Array(this.cols * this.rows);
    N.quantizeColors(this._h, this._src, this.cols, this.rows, 0, !!this.grayscaleText, this.colors);
    return this.colors;
  }

  // Keeps a reference to the caller's RGBA grid (cols*rows*4); auto-resizes on mismatch
  // (ascii_pass.js:243-255).
  blit(rgbaBuffer, srcCols, srcRows) {
    if (!rgbaBuffer) return;
    if ((srcCols | 0) !== (this.cols | 0) || (srcRows | 0) !== (this.rows | 0)) this.setGridSize(srcCols | 0, srcRows | 0);
    this._src = rgbaBuffer;
  }

  // Returns this.codes (Uint8Array cols*rows, row 0 = first row of the blitted buffer).
  draw() {
    if (!this._src) return this.codes;
    if (this._src.length !== this.cols * this.rows * 4) throw new Error('AsciiPass.draw: buffer does not match the grid');
    this._ensure();
    const N = native();
    const r = Math.max(1, Math.min(3, this.modeRadius | 0));   // MAX_MODE_RADIUS = 3 (ascii_pass.js:296-297)
    N.setAscii(this._h, rampBytes(this.ramp), this.modeFilter, r, this.modeThresh | 0);
    if (this.noAlloc) N.quantizeStatic(this._h, this._src, this.cols, this.rows, this.codes);
    else N.quantize(this._h, this._src, this.cols, this.rows, this.codes);
    return this.codes;
  }

  setCellSize(cwCSS, chCSS) {
    if (!cwCSS || !chCSS) return;
    this.cwCSS = Math.max(0.5, +cwCSS);
    this.chCSS = Math.max(0.5, +chCSS);
    this.cellWDevice = Math.max(1, Math.round(this.cwCSS * this.dpr));
    this.cellHDevice = Math.max(1, Math.round(this.chCSS * this.dpr));
    this._atlas = atlasDescriptor(this.cellWDevice, this.cellHDevice);
  }

  // Row y of the code grid as a string: what TextOverlay._computeRowString materialises
  // (js/text_overlay.js:128-148), with smoothing applied when it is on.
  rowString(y) {
    let out = '';
    for (let x = 0; x < this.cols; x++) out += String.fromCharCode(this.codes[y * this.cols + x]);
    return out;
  }
}
