"""Bakes a glyph atlas alpha plane in the reference's layout (js/ascii_pass.js:20-60: 16 x 16 tiles of
(cell_w + 4) x (cell_h + 4) bytes, glyph g at tile (g % 16, g / 16), 2 bytes of padding) with PIL's
built-in font, glyphs centred vertically on the height of 'M' as buildAtlas does.
usage: bake_atlas.py <cell_w> <cell_h> <out.bin>"""
import sys
import numpy as np
from PIL import Image, ImageDraw, ImageFont


def bake(cell_w, cell_h):
    pad, tw, th = 2, cell_w + 4, cell_h + 4
    font = ImageFont.load_default(size=max(4, int(round(cell_h * 0.9))))
    img = Image.new("L", (16 * tw, 16 * th), 0)
    d = ImageDraw.Draw(img)
    l, t, r, b = font.getbbox("M")
    oy = (cell_h - (b - t)) // 2 - t
    for g in range(32, 127):
        ch = chr(g)
        gl, _gt, gr, _gb = font.getbbox(ch)
        ox = (cell_w - (gr - gl)) // 2 - gl
        d.text(((g % 16) * tw + pad + ox, (g // 16) * th + pad + oy), ch, fill=255, font=font)
    return np.asarray(img, dtype=np.uint8)


if __name__ == "__main__":
    cw, chh, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    a = bake(cw, chh)
    a.tofile(out)
    print("atlas %dx%d bytes -> %s, coverage %.1f%%" % (a.shape[1], a.shape[0], out, 100.0 * np.count_nonzero(a) / a.size))
