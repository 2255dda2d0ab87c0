/*
 * cpu_raster.c - CPU ORACLE for the 3D -> ASCII raster path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a checker: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg are the
 * only callers.  Nothing in the product path (the HIP library, the N-API addon, the JS host) links,
 * loads or calls it.
 *
 * It restates, as plain scalar C, what the reference does on one frame of its raster backend:
 *
 *   stage                                   reference                                  pinned by
 *   --------------------------------------  -----------------------------------------  ----------------
 *   vertex shader  clip = P*V*[p,1]         js/render/backends/raster_shader.js:17-22  (WebGL; unpinned)
 *   clip / cull / raster / depth test       WebGL fixed function configured by         (WebGL; unpinned)
 *                                           js/render/backends/raster.js:100-102,
 *                                           276-279,355; target js/render/gl/context.js:88-115
 *   fragment shader (Lambert + <=8 points)  raster_shader.js:42-62                     (WebGL; unpinned)
 *   readPixels + row flip                   raster.js:358-361, context.js:117-124      (WebGL; unpinned)
 *   luminance -> ramp -> ASCII code         js/renderer.js:59-72,                      golden vectors
 *                                           js/text_overlay.js:137-145,                tests/golden/
 *                                           js/ascii_pass_shader.js:12-16,62-68        ref_quantize.json
 *   alpha override (2..254 = literal code)  ascii_pass_shader.js:153-162               golden vectors
 *   modal (majority) smoothing              ascii_pass_shader.js:77-138,169-185        (GLSL; unpinned)
 *   integer-UV atlas texturing of           js/render/backends/shader_utils.js:94-132, (GLSL; unpinned; the
 *   triangles (optional)                    pathtrace_shader.js:125-152,256-258        reference applies it in
 *                                                                                      its path tracer only)
 *
 *   ray backend (orc_rt_render)             js/render/backends/raytrace_shader.js      (GLSL; unpinned; its
 *                                           :62-242                                    uniforms are pinned)
 *
 * PARITY UNPINNED for the rows marked "WebGL"/"GLSL": that arithmetic runs inside the browser's WebGL
 * implementation, which is not part of the reference tree and has no golden outputs there.  For those
 * rows this file is the definition: it follows the OpenGL ES rules the reference's GL state selects
 * and fixes every choice the GL specification leaves open (listed below).  The quantizer rows are
 * pinned against outputs of the reference's own compositeAndQuantize.
 *
 * Numeric contract (every implementation that claims parity follows exactly this):
 *   - IEEE-754 binary32/binary64, round-to-nearest-even, denormals kept, NO fused multiply-add,
 *     correctly rounded / and sqrt.  Expressions are evaluated left to right as parenthesised here.
 *   - PV = P*V in binary32 (column-major, k = 0..3 left to right); clip = PV*[x,y,z,1].
 *   - near plane: a vertex is inside when z_c + w_c >= 0.  Triangles with 1 or 2 vertices outside are
 *     clipped against that plane only (new vertex = I + (O - I) * (dI / (dI - dO)), always computed
 *     from the inside vertex I towards the outside vertex O, for clip coordinates and all varyings),
 *     giving 1 or 2 sub-triangles in the original winding.  No x/y/far geometric clipping: x/y are
 *     handled by scanning only cells of the viewport, far by the depth test against the clear value.
 *   - pixel-homogeneous vertex (X,Y,W) = ((x_c + w_c)*cols/2, (y_c + w_c)*rows/2, w_c) in binary32,
 *     GL window orientation (y up).  Edge functions are the cross products of those vectors taken in
 *     binary64 (a,b,c) and evaluated in binary64 at cell centres (i+0.5, j+0.5): E = a*px + (b*py + c).
 *     det = (X0*a0 + Y0*b0) + W0*c0 in binary64; det <= 0 is culled (back face or zero area).
 *     A cell is covered when every E_k > 0, or E_k == 0 on an edge with a_k > 0 or (a_k == 0 and
 *     b_k > 0) - a top-left style rule that gives exactly one owner on a shared edge because the
 *     neighbour evaluates the exact negation.
 *   - f_k = (float)E_k.  Depth (screen-linear, as GL requires) zn = ((f0*z0 + f1*z1) + f2*z2) /
 *     ((f0*W0 + f1*W1) + f2*W2), zw = zn*0.5 + 0.5, kept only when 0 <= zw < 1; test LESS against a
 *     buffer cleared to 1.0, stored as binary32 (the reference target is DEPTH_COMPONENT16; binary32
 *     is this build's documented deviation).  Equal depth: the earlier triangle wins.
 *   - varyings are perspective-correct: b_k = f_k * (1 / ((f0 + f1) + f2)); v = (b0*v0 + b1*v1) + b2*v2.
 *   - fragment shader in binary32 as written in shade(); colour byte = floor(clamp(c,0,1)*255 + 0.5),
 *     alpha 255; clear colour (0,0,0,255).
 *   - output row 0 is the TOP image row (the reference flips after readPixels).
 *   - texturing (orc_render_textured): u and v are two more varyings (binary32 values of the integer
 *     texel coordinates), clipped and interpolated like the others; texel = floor(uv + 0.5) per axis,
 *     ignored when outside the atlas; the atlas is the array as handed to texImage2D, texel (x, y) is
 *     read from its row height-1-y (shader_utils.js:100-102).  Texel alpha 0: no texture.  Alpha 1: the
 *     texel's rgb / 255 replaces the vertex colour before shading.  Alpha 32..126: the cell's RGB is
 *     the texel's RGB bytes, unshaded, and its alpha is the texel's alpha (an ASCII override for the
 *     quantizer).  Any other alpha: no texture.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_POINT_LIGHTS 8 /* raster_shader.js:5 */

#define ORC_NV 11
typedef struct {
    float c[4];        /* clip x,y,z,w */
    float v[ORC_NV];   /* varyings: world pos, normal, colour (raster_shader.js:18-20), then texel u, v */
} orc_vert;

typedef struct {
    const uint8_t *rgba;   /* as handed to texImage2D (pathtrace.js:352-353) */
    int w, h;
} orc_atlas;

typedef struct {
    float amb[3], ldir[3], lcol[3];
    int n_pl;
    float plpos[3 * ORC_MAX_POINT_LIGHTS], plcol[3 * ORC_MAX_POINT_LIGHTS];
} orc_lights;

/* PV = P*V, column-major 4x4, binary32.  (GLSL evaluates uProj * uView * v as (uProj*uView)*v.) */
void orc_matmul_pv(const float *P, const float *V, float *PV)
{
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            PV[c * 4 + r] = ((P[0 * 4 + r] * V[c * 4 + 0] + P[1 * 4 + r] * V[c * 4 + 1]) +
                             P[2 * 4 + r] * V[c * 4 + 2]) + P[3 * 4 + r] * V[c * 4 + 3];
}

/* raster_shader.js:17-22: gl_Position = uProj * uView * vec4(a_pos, 1.0); varyings copied.
 * With a model matrix M (orc_render_model; the reference has none, raster_shader.js:12-13) the shader reads
 *   gl_Position = uProj * uView * uModel * vec4(a_pos, 1.0)   - GLSL multiplies left to right, matrices first, so PV
 *                                                               here is (P*V)*M, formed by the caller in binary32;
 *   v_pos = (uModel * vec4(a_pos, 1.0)).xyz;  v_nrm = mat3(uModel) * a_nrm;  v_col = a_col
 * (normals go through the upper-left 3x3 as they are: exact for rigid motions and uniform scales, since the
 * fragment shader normalises). */
static void vertex_shader(const float *PV, const float *M, const float *vtx9, orc_vert *o)
{
    const float x = vtx9[0], y = vtx9[1], z = vtx9[2];
    for (int r = 0; r < 4; r++)
        o->c[r] = ((PV[0 * 4 + r] * x + PV[1 * 4 + r] * y) + PV[2 * 4 + r] * z) + PV[3 * 4 + r];
    for (int k = 0; k < 9; k++) o->v[k] = vtx9[k];
    if (M) {
        const float nx = vtx9[3], ny = vtx9[4], nz = vtx9[5];
        for (int r = 0; r < 3; r++) {
            o->v[r] = ((M[0 * 4 + r] * x + M[1 * 4 + r] * y) + M[2 * 4 + r] * z) + M[3 * 4 + r];
            o->v[3 + r] = (M[0 * 4 + r] * nx + M[1 * 4 + r] * ny) + M[2 * 4 + r] * nz;
        }
    }
    o->v[9] = o->v[10] = 0.0f;
}

/* new vertex on the near plane, always from the inside vertex I towards the outside vertex O */
static void clip_vertex(const orc_vert *I, float dI, const orc_vert *O, float dO, orc_vert *out)
{
    const float t = dI / (dI - dO);
    for (int k = 0; k < 4; k++) out->c[k] = I->c[k] + (O->c[k] - I->c[k]) * t;
    for (int k = 0; k < ORC_NV; k++) out->v[k] = I->v[k] + (O->v[k] - I->v[k]) * t;
}

/* Near-plane clip of one triangle; returns the number of sub-triangles (0, 1 or 2). */
static int clip_near(const orc_vert in[3], orc_vert out[2][3])
{
    float d[3];
    int inside[3], n_in = 0;
    for (int k = 0; k < 3; k++) {
        d[k] = in[k].c[2] + in[k].c[3];
        inside[k] = d[k] >= 0.0f;
        n_in += inside[k];
    }
    if (n_in == 0) return 0;
    if (n_in == 3) { memcpy(out[0], in, 3 * sizeof(orc_vert)); return 1; }
    if (n_in == 1) {
        int a = inside[0] ? 0 : (inside[1] ? 1 : 2), b = (a + 1) % 3, c = (a + 2) % 3;
        out[0][0] = in[a];
        clip_vertex(&in[a], d[a], &in[b], d[b], &out[0][1]);
        clip_vertex(&in[a], d[a], &in[c], d[c], &out[0][2]);
        return 1;
    }
    /* two inside: a is the outside vertex, (a,b,c) cyclic */
    {
        int a = !inside[0] ? 0 : (!inside[1] ? 1 : 2), b = (a + 1) % 3, c = (a + 2) % 3;
        orc_vert pab, pca;
        clip_vertex(&in[b], d[b], &in[a], d[a], &pab);
        clip_vertex(&in[c], d[c], &in[a], d[a], &pca);
        out[0][0] = pab; out[0][1] = in[b]; out[0][2] = in[c];
        out[1][0] = pab; out[1][1] = in[c]; out[1][2] = pca;
        return 2;
    }
}

/* raster_shader.js:42-62 in binary32; returns RGBA8 with alpha 255. */
static void shade(const float v[9], const orc_lights *L, uint8_t rgba[4])
{
    const float *pos = v, *nrm = v + 3, *vc = v + 6;
    const float len2 = (nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2];
    const float inv = (len2 > 0.0f) ? 1.0f / sqrtf(len2) : 0.0f;       /* N = normalize(v_nrm) */
    const float N[3] = { nrm[0] * inv, nrm[1] * inv, nrm[2] * inv };
    float col[3];
    for (int k = 0; k < 3; k++) col[k] = vc[k] * L->amb[k];             /* col = v_col * uAmbient */
    const float ndl = fmaxf((N[0] * (-L->ldir[0]) + N[1] * (-L->ldir[1])) + N[2] * (-L->ldir[2]), 0.0f);
    for (int k = 0; k < 3; k++) col[k] = col[k] + (vc[k] * L->lcol[k]) * ndl;
    for (int i = 0; i < ORC_MAX_POINT_LIGHTS; i++) {
        if (i >= L->n_pl) break;
        const float lv[3] = { L->plpos[3 * i] - pos[0], L->plpos[3 * i + 1] - pos[1], L->plpos[3 * i + 2] - pos[2] };
        const float d2 = fmaxf((lv[0] * lv[0] + lv[1] * lv[1]) + lv[2] * lv[2], 1e-4f);
        const float il = 1.0f / sqrtf(d2);                              /* inversesqrt */
        const float Ld[3] = { lv[0] * il, lv[1] * il, lv[2] * il };
        const float ndlp = fmaxf((N[0] * Ld[0] + N[1] * Ld[1]) + N[2] * Ld[2], 0.0f);
        const float atten = 1.0f / (1.0f + d2 * 0.05f);
        const float w = ndlp * atten;
        for (int k = 0; k < 3; k++) col[k] = col[k] + (vc[k] * L->plcol[3 * i + k]) * w;
    }
    for (int k = 0; k < 3; k++) {
        const float c = fminf(fmaxf(col[k], 0.0f), 1.0f);
        rgba[k] = (uint8_t)(int)floorf(c * 255.0f + 0.5f);
    }
    rgba[3] = 255;
}

static int edge_inside(double E, double a, double b)
{
    return E > 0.0 || (E == 0.0 && (a > 0.0 || (a == 0.0 && b > 0.0)));
}

/*
 * Atlas lookup for one fragment (shader_utils.js:108-117 with fetchAtlas :100-105; the texel classes of
 * pathtrace_shader.js:141-152).  Returns 1 when the texel is an ASCII pass-through (out holds the final
 * RGBA), else 0 with col replaced by the texel colour when the texel is solid.
 */
static int apply_texture(const orc_atlas *A, float u, float v, float col[3], uint8_t out[4])
{
    const float tx = floorf(u + 0.5f), ty = floorf(v + 0.5f);
    if (!(tx >= 0.0f) || !(ty >= 0.0f) || !(tx < (float)A->w) || !(ty < (float)A->h)) return 0;
    const uint8_t *texel = A->rgba + 4 * ((size_t)(A->h - 1 - (int)ty) * A->w + (int)tx);
    const int a = texel[3];
    if (a >= 32 && a <= 126) {
        out[0] = texel[0]; out[1] = texel[1]; out[2] = texel[2]; out[3] = (uint8_t)a;
        return 1;
    }
    if (a == 1)
        for (int k = 0; k < 3; k++) col[k] = (float)texel[k] / 255.0f;
    return 0;
}

/* Rasterize one (sub-)triangle into the bottom-up work buffers.  A: atlas, or NULL for an untextured triangle. */
static void raster_tri(const orc_vert t[3], uint32_t tri_id, const orc_lights *L, int cols, int rows,
                       uint8_t *rgba, float *depth, uint32_t *ids, const orc_atlas *A)
{
    float X[3], Y[3], W[3], Z[3];
    const float hw = (float)cols * 0.5f, hh = (float)rows * 0.5f;
    for (int k = 0; k < 3; k++) {
        for (int q = 0; q < 4; q++) if (!isfinite(t[k].c[q])) return;
        if (!(t[k].c[3] > 0.0f)) return;
        W[k] = t[k].c[3];
        Z[k] = t[k].c[2];
        X[k] = (t[k].c[0] + W[k]) * hw;
        Y[k] = (t[k].c[1] + W[k]) * hh;
    }
    double a[3], b[3], c[3];
    for (int k = 0; k < 3; k++) {                   /* e_k = v_{k+1} x v_{k+2} */
        const int p = (k + 1) % 3, q = (k + 2) % 3;
        a[k] = (double)Y[p] * (double)W[q] - (double)Y[q] * (double)W[p];
        b[k] = (double)X[q] * (double)W[p] - (double)X[p] * (double)W[q];
        c[k] = (double)X[p] * (double)Y[q] - (double)X[q] * (double)Y[p];
    }
    const double det = ((double)X[0] * a[0] + (double)Y[0] * b[0]) + (double)W[0] * c[0];
    if (!(det > 0.0)) return;                       /* CULL_FACE BACK, CCW front; zero area */

    /* conservative cell range: any superset of the covered cells gives the same image */
    float minx = X[0] / W[0], maxx = minx, miny = Y[0] / W[0], maxy = miny;
    for (int k = 1; k < 3; k++) {
        const float sx = X[k] / W[k], sy = Y[k] / W[k];
        minx = fminf(minx, sx); maxx = fmaxf(maxx, sx);
        miny = fminf(miny, sy); maxy = fmaxf(maxy, sy);
    }
    const float lox = ceilf(fmaxf(minx - 0.50390625f, 0.0f)), hix = floorf(fminf(maxx - 0.49609375f, (float)(cols - 1)));
    const float loy = ceilf(fmaxf(miny - 0.50390625f, 0.0f)), hiy = floorf(fminf(maxy - 0.49609375f, (float)(rows - 1)));
    if (!(lox <= hix) || !(loy <= hiy)) return;
    const int i0 = (int)lox, i1 = (int)hix, j0 = (int)loy, j1 = (int)hiy;

    for (int j = j0; j <= j1; j++) {
        const double py = (double)j + 0.5;
        for (int i = i0; i <= i1; i++) {
            const double px = (double)i + 0.5;
            const double E0 = a[0] * px + (b[0] * py + c[0]);
            const double E1 = a[1] * px + (b[1] * py + c[1]);
            const double E2 = a[2] * px + (b[2] * py + c[2]);
            if (!edge_inside(E0, a[0], b[0]) || !edge_inside(E1, a[1], b[1]) || !edge_inside(E2, a[2], b[2]))
                continue;
            const float f0 = (float)E0, f1 = (float)E1, f2 = (float)E2;
            const float den = (f0 * W[0] + f1 * W[1]) + f2 * W[2];
            const float num = (f0 * Z[0] + f1 * Z[1]) + f2 * Z[2];
            const float zn = num / den;
            const float zw = zn * 0.5f + 0.5f;
            if (!(zw >= 0.0f) || !(zw < 1.0f)) continue;
            const size_t cell = (size_t)j * cols + i;
            if (!(zw < depth[cell])) continue;      /* depthFunc LESS */
            depth[cell] = zw;
            if (ids) ids[cell] = tri_id;
            const float rS = 1.0f / ((f0 + f1) + f2);
            const float b0 = f0 * rS, b1 = f1 * rS, b2 = f2 * rS;
            float v[ORC_NV];
            for (int q = 0; q < ORC_NV; q++) v[q] = (b0 * t[0].v[q] + b1 * t[1].v[q]) + b2 * t[2].v[q];
            if (A && apply_texture(A, v[9], v[10], v + 6, rgba + cell * 4)) continue;
            shade(v, L, rgba + cell * 4);
        }
    }
}

/*
 * Render one frame.  vbo9: n_verts x (pos3, nrm3, col3) binary32 (raster.js:222-236); idx: n_tris x 3
 * u32 (raster.js:249).  proj/view: column-major binary32 as uploaded by raster.js:303-304.  Lights as
 * packed by raster.js:307-342.  Outputs are TOP-DOWN (row 0 = top): rgba_out cols*rows*4, depth_out
 * cols*rows (1.0 = background), ids_out cols*rows (2*triangle + sub-triangle of the winner,
 * 0xFFFFFFFF = background); any may be NULL.
 * Triangles naming a vertex >= n_verts are skipped.  Returns 0, or -1 on bad arguments.
 */
int orc_render_textured(const float *vbo9, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris,
                        const float *proj, const float *view,
                        const float *amb, const float *ldir, const float *lcol,
                        int n_pl, const float *plpos, const float *plcol,
                        int cols, int rows, uint8_t *rgba_out, float *depth_out, uint32_t *ids_out,
                        const uint16_t *uv8, const uint8_t *atlas_rgba, int atlas_w, int atlas_h);
int orc_render_model(const float *vbo9, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris,
                     const float *proj, const float *view, const float *model,
                     const float *amb, const float *ldir, const float *lcol,
                     int n_pl, const float *plpos, const float *plcol,
                     int cols, int rows, uint8_t *rgba_out, float *depth_out, uint32_t *ids_out,
                     const uint16_t *uv8, const uint8_t *atlas_rgba, int atlas_w, int atlas_h);

int orc_render(const float *vbo9, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris,
               const float *proj, const float *view,
               const float *amb, const float *ldir, const float *lcol,
               int n_pl, const float *plpos, const float *plcol,
               int cols, int rows, uint8_t *rgba_out, float *depth_out, uint32_t *ids_out)
{
    return orc_render_textured(vbo9, n_verts, idx, n_tris, proj, view, amb, ldir, lcol, n_pl, plpos, plcol, cols, rows,
                               rgba_out, depth_out, ids_out, NULL, NULL, 0, 0);
}

/*
 * orc_render with atlas texturing.  uv8: per triangle eight u16 (u0, v0, u1, v1, u2, v2, flags, 0): integer
 * texel coordinates of the three vertices (scene_api.js:139-160) and flags bit 0 = "sample the atlas for
 * this triangle" (the path tracer samples every triangle, and a quad half unless its three coordinate
 * pairs are all zero, shader_utils.js:108-132; the caller encodes that in the bit).  atlas_rgba: RGBA8
 * texels as handed to texImage2D; texturing is on when both sizes exceed 1 (shader_utils.js:94).
 * uv8 == NULL or atlas_rgba == NULL: plain orc_render.
 */
int orc_render_textured(const float *vbo9, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris,
                        const float *proj, const float *view,
                        const float *amb, const float *ldir, const float *lcol,
                        int n_pl, const float *plpos, const float *plcol,
                        int cols, int rows, uint8_t *rgba_out, float *depth_out, uint32_t *ids_out,
                        const uint16_t *uv8, const uint8_t *atlas_rgba, int atlas_w, int atlas_h)
{
    return orc_render_model(vbo9, n_verts, idx, n_tris, proj, view, NULL, amb, ldir, lcol, n_pl, plpos, plcol, cols, rows,
                            rgba_out, depth_out, ids_out, uv8, atlas_rgba, atlas_w, atlas_h);
}

/*
 * orc_render_textured with a model matrix (column-major binary32, or NULL for none): the mesh is drawn as if the
 * vertex shader had a uModel uniform - see vertex_shader().  The reference's rasterizer has no model matrix; this
 * is the definition the build's ar_set_model follows.
 */
int orc_render_model(const float *vbo9, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris,
                     const float *proj, const float *view, const float *model,
                     const float *amb, const float *ldir, const float *lcol,
                     int n_pl, const float *plpos, const float *plcol,
                     int cols, int rows, uint8_t *rgba_out, float *depth_out, uint32_t *ids_out,
                     const uint16_t *uv8, const uint8_t *atlas_rgba, int atlas_w, int atlas_h)
{
    const orc_atlas atlas = { atlas_rgba, atlas_w, atlas_h };
    const int textured = uv8 && atlas_rgba && atlas_w > 1 && atlas_h > 1;
    if (cols <= 0 || rows <= 0 || n_pl < 0 || n_pl > ORC_MAX_POINT_LIGHTS) return -1;
    const size_t n = (size_t)cols * rows;
    uint8_t *rgba = (uint8_t *)malloc(n * 4);
    float *depth = (float *)malloc(n * sizeof(float));
    uint32_t *ids = ids_out ? (uint32_t *)malloc(n * sizeof(uint32_t)) : NULL;
    if (!rgba || !depth || (ids_out && !ids)) { free(rgba); free(depth); free(ids); return -1; }
    for (size_t p = 0; p < n; p++) {                /* clearColor(0,0,0,1), clearDepth 1.0 */
        rgba[4 * p] = rgba[4 * p + 1] = rgba[4 * p + 2] = 0; rgba[4 * p + 3] = 255;
        depth[p] = 1.0f;
        if (ids) ids[p] = 0xFFFFFFFFu;
    }
    orc_lights L;
    memset(&L, 0, sizeof L);
    for (int k = 0; k < 3; k++) { L.amb[k] = amb[k]; L.ldir[k] = ldir[k]; L.lcol[k] = lcol[k]; }
    L.n_pl = n_pl;
    for (int k = 0; k < 3 * n_pl; k++) { L.plpos[k] = plpos[k]; L.plcol[k] = plcol[k]; }

    float PV[16];
    orc_matmul_pv(proj, view, PV);
    if (model) {                                    /* (P*V)*M, the same product routine */
        float PVM[16];
        orc_matmul_pv(PV, model, PVM);
        memcpy(PV, PVM, sizeof PV);
    }
    for (uint32_t t = 0; t < n_tris; t++) {
        const uint32_t i0 = idx[3 * t], i1 = idx[3 * t + 1], i2 = idx[3 * t + 2];
        if (i0 >= n_verts || i1 >= n_verts || i2 >= n_verts) continue;
        orc_vert in[3], sub[2][3];
        vertex_shader(PV, model, vbo9 + (size_t)i0 * 9, &in[0]);
        vertex_shader(PV, model, vbo9 + (size_t)i1 * 9, &in[1]);
        vertex_shader(PV, model, vbo9 + (size_t)i2 * 9, &in[2]);
        const int sample = textured && (uv8[8 * (size_t)t + 6] & 1u);
        if (sample)
            for (int k = 0; k < 3; k++) {
                in[k].v[9] = (float)uv8[8 * (size_t)t + 2 * k];
                in[k].v[10] = (float)uv8[8 * (size_t)t + 2 * k + 1];
            }
        const int ns = clip_near(in, sub);
        for (int s = 0; s < ns; s++)
            raster_tri(sub[s], 2u * t + (uint32_t)s, &L, cols, rows, rgba, depth, ids, sample ? &atlas : NULL);
    }
    for (int r = 0; r < rows; r++) {                /* flipAndCopy: output row r = GL row rows-1-r */
        const size_t src = (size_t)(rows - 1 - r) * cols, dst = (size_t)r * cols;
        if (rgba_out) memcpy(rgba_out + dst * 4, rgba + src * 4, (size_t)cols * 4);
        if (depth_out) memcpy(depth_out + dst, depth + src, (size_t)cols * sizeof(float));
        if (ids_out) memcpy(ids_out + dst, ids + src, (size_t)cols * sizeof(uint32_t));
    }
    free(rgba); free(depth); free(ids);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * ASCII stage.
 * ramp index (js/renderer.js:67-70, text_overlay.js:142-143):
 *     idx = clamp(round(((r+g+b)/3/255) * (L-1)), 0, L-1)
 * evaluated here exactly as the reference does, in binary64 with Math.round = floor(x + 0.5).
 * The product path uses the equivalent integer form floor((2*s*(L-1) + 765) / 1530); tests check both
 * against the reference's recorded outputs.
 * ---------------------------------------------------------------------------------------------- */
static int ramp_index(int r, int g, int b, int ramp_len)
{
    const double intensity = (double)(r + g + b) / 3.0;
    double idx = floor((intensity / 255.0) * (double)(ramp_len - 1) + 0.5);
    if (idx < 0.0) idx = 0.0;
    if (idx > (double)(ramp_len - 1)) idx = (double)(ramp_len - 1);
    return (int)idx;
}

static int is_override(int a) { return a >= 2 && a <= 254; } /* ascii_pass_shader.js:98,155 */

/*
 * RGBA8 grid (row-major, as handed to AsciiPass.blit, js/ascii_pass.js:243-255) -> one ASCII code per
 * cell.  mode_on / mode_radius / mode_thresh are the uniforms uModeOn / uModeRadius / uModeThresh
 * (ascii_pass.js:294-299); radius is clamped to 1..3 as the shader does (ascii_pass_shader.js:173-175).
 * Neighbour order dy outer, dx inner, coordinates clamped to the grid, centre skipped, override cells
 * skipped (ascii_pass_shader.js:84-105).  Returns 0, or -1 on bad arguments.
 */
int orc_quantize(const uint8_t *rgba, int cols, int rows, const uint8_t *ramp, int ramp_len,
                 int mode_on, int mode_radius, int mode_thresh, uint8_t *ascii_out)
{
    if (cols <= 0 || rows <= 0 || ramp_len <= 0) return -1;
    const size_t n = (size_t)cols * rows;
    int *base = (int *)malloc(n * sizeof(int));
    if (!base) return -1;
    for (size_t p = 0; p < n; p++)
        base[p] = ramp_index(rgba[4 * p], rgba[4 * p + 1], rgba[4 * p + 2], ramp_len);
    int radius = mode_radius;
    if (radius < 1) radius = 1;
    if (radius > 3) radius = 3;
    for (int y = 0; y < rows; y++) {
        for (int x = 0; x < cols; x++) {
            const size_t p = (size_t)y * cols + x;
            const int a = rgba[4 * p + 3];
            if (is_override(a)) { ascii_out[p] = (uint8_t)a; continue; }
            int final_idx = base[p];
            if (mode_on == 1) {
                int cand = -1, cnt = 0, votes = 0;
                for (int pass = 0; pass < 2; pass++) {
                    if (pass == 1 && cand < 0) break;
                    for (int dy = -radius; dy <= radius; dy++) {
                        for (int dx = -radius; dx <= radius; dx++) {
                            if (dx == 0 && dy == 0) continue;
                            int nx = x + dx, ny = y + dy;
                            nx = nx < 0 ? 0 : (nx > cols - 1 ? cols - 1 : nx);
                            ny = ny < 0 ? 0 : (ny > rows - 1 ? rows - 1 : ny);
                            const size_t q = (size_t)ny * cols + nx;
                            if (is_override(rgba[4 * q + 3])) continue;
                            const int ni = base[q];
                            if (pass == 0) {                    /* Boyer-Moore candidate */
                                if (cnt == 0) { cand = ni; cnt = 1; }
                                else if (ni == cand) cnt += 1;
                                else cnt -= 1;
                            } else if (ni == cand) votes += 1;  /* true vote count */
                        }
                    }
                }
                if (cand >= 0 && votes >= mode_thresh && cand != base[p]) final_idx = cand;
            }
            /* asciiFromRampIndex: out-of-range falls back to space (ascii_pass_shader.js:15) */
            ascii_out[p] = (final_idx >= 0 && final_idx < ramp_len) ? ramp[final_idx] : 32;
        }
    }
    free(base);
    return 0;
}

/* Math.round of a non-negative double: nearest integer, halves up (ECMA-262 Math.round). */
static double js_round(double x)
{
    double r = floor(x);
    if (x - r >= 0.5) r += 1.0;
    return r;
}

/*
 * Colour half of compositeAndQuantize (js/renderer.js:13-19,28-30,38-44,74-84): per cell the packed
 * text colour (r << 16) | (g << 8) | b of the DOM path.  Each channel is snapped to `steps` levels,
 * qChan(u8) = round(round(u8 / 255 * (steps - 1)) * (255 / (steps - 1))), in binary64 as JavaScript
 * does; steps <= 0 selects the reference's own choice max(3, max(2, ramp_len)).  UI override cells
 * (alpha 2..254, see is_override) and every cell when grayscale is set are black (0).
 * Pinned by colors_u32 of tests/golden/ref_quantize.json.  Returns 0, or -1 on bad arguments.
 */
int orc_quantize_colors(const uint8_t *rgba, int cols, int rows, int ramp_len, int steps, int grayscale,
                        uint32_t *colors_out)
{
    if (cols <= 0 || rows <= 0) return -1;
    if (steps <= 0) {
        int rl = ramp_len < 2 ? 2 : ramp_len;
        steps = rl < 3 ? 3 : rl;
    }
    if (steps < 2) return -1;
    const double step_val = 255.0 / (double)(steps - 1);
    uint32_t lut[256];
    for (int u = 0; u < 256; u++)
        lut[u] = (uint32_t)js_round(js_round(((double)u / 255.0) * (double)(steps - 1)) * step_val);
    const size_t n = (size_t)cols * rows;
    for (size_t p = 0; p < n; p++) {
        if (grayscale || is_override(rgba[4 * p + 3])) { colors_out[p] = 0; continue; }
        colors_out[p] = (lut[rgba[4 * p]] << 16) | (lut[rgba[4 * p + 1]] << 8) | lut[rgba[4 * p + 2]];
    }
    return 0;
}

/*
 * Glyph compositing of the ASCII pass (js/ascii_pass_shader.js:190-230) for a canvas of exactly
 * cols*cell_w x rows*cell_h device pixels.  PARITY UNPINNED (the reference does this in WebGL and
 * bakes its atlas with a browser font); this function is the definition the HIP kernel follows.
 *   codes       cols*rows glyph indices (output of orc_quantize), rgba the grid they came from
 *   atlas       alpha plane of a 16 x 16 tile atlas, tile = (cell_w + 4) x (cell_h + 4) with 2 pixels of
 *               padding, glyph g at tile (g % 16, g / 16) (js/ascii_pass.js:20-60), row 0 = top of the glyphs
 *   image       RGBA8, row 0 = top; pixel (X, Y) belongs to cell (X / cell_w, Y / cell_h) and reads the
 *               atlas texel (pad + X % cell_w, pad + Y % cell_h) of its glyph's tile (nearest, :46-47)
 *   cov = pow(alpha / 255, gamma) evaluated in binary64 and rounded to binary32 (gamma == 1: alpha / 255 in
 *   binary32); tint = cell colour / 255 (0 when grayscale); out = (1 - cov) + tint * cov per channel in
 *   binary32 (mix(vec3(1), tint, cov)), byte = floor(clamp(out, 0, 1) * 255 + 0.5), alpha 255;
 *   with transparent_bg a pixel whose cov <= 0 is (0, 0, 0, 0) (the shader discards it).
 */
int orc_present(const uint8_t *rgba, const uint8_t *codes, int cols, int rows, const uint8_t *atlas, int cell_w,
                int cell_h, double gamma, int grayscale, int transparent_bg, uint8_t *image)
{
    if (cols <= 0 || rows <= 0 || cell_w <= 0 || cell_h <= 0) return -1;
    const int pad = 2, tile_w = cell_w + 2 * pad, tile_h = cell_h + 2 * pad, atlas_w = 16 * tile_w;
    float cov_of[256];
    for (int a = 0; a < 256; a++)
        cov_of[a] = (gamma == 1.0) ? (float)a / 255.0f : (float)pow((double)a / 255.0, gamma);
    const size_t W = (size_t)cols * cell_w;
    for (int cy = 0; cy < rows; cy++)
        for (int py = 0; py < cell_h; py++)
            for (int cx = 0; cx < cols; cx++) {
                const size_t cell = (size_t)cy * cols + cx;
                const int g = codes[cell];
                const uint8_t *trow = atlas + (size_t)((g / 16) * tile_h + pad + py) * atlas_w + (g % 16) * tile_w + pad;
                uint8_t *out = image + (((size_t)cy * cell_h + py) * W + (size_t)cx * cell_w) * 4;
                for (int px = 0; px < cell_w; px++, out += 4) {
                    const float cov = cov_of[trow[px]];
                    if (transparent_bg && cov <= 0.0f) { out[0] = out[1] = out[2] = out[3] = 0; continue; }
                    for (int ch = 0; ch < 3; ch++) {
                        const float tint = grayscale ? 0.0f : (float)rgba[4 * cell + ch] / 255.0f;
                        float v = (1.0f - cov) + tint * cov;
                        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                        out[ch] = (uint8_t)floorf(v * 255.0f + 0.5f);
                    }
                    out[3] = 255;
                }
            }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Ray backend: the reference's deterministic single-bounce ray tracer
 * (js/render/backends/raytrace_shader.js:62-242; uniforms as js/render/backends/raytrace.js:195-389
 * uploads them).  PARITY UNPINNED: the arithmetic is GLSL inside the browser's WebGL implementation and
 * the reference holds no output of it, so this restatement is the definition.  Choices GLSL leaves open:
 *   - binary32 throughout, no fused multiply-add, correctly rounded / and sqrt, expressions left to right;
 *     dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z; cross by components; length = sqrt(dot);
 *     normalize(v) = v * (1/sqrt(dot(v,v))), and the zero vector for dot == 0; inversesqrt(x) = 1/sqrt(x);
 *     reflect(I,N) = I - (2*dot(N,I))*N; clamp(x,a,b) = min(max(x,a),b); max/min are fmaxf/fminf.
 *   - the per-frame camera terms (cos/sin of yaw and pitch, tan(fovy/2)) are uniform over the frame; they
 *     are evaluated once on the host in binary64 from the binary32 uniform values and rounded to binary32.
 *   - a fragment sits at (i + 0.5, j + 0.5) in GL window coordinates; output row 0 is the top image row
 *     (flipAndCopy after readPixels, raytrace.js:398-402); colour byte = floor(clamp(c,0,1)*255 + 0.5),
 *     alpha 255.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int n_spheres; const float *spheres4; const int32_t *sphere_mat;
    int n_planes;  const float *planes4;  const int32_t *plane_mat;
    int n_tris;    const float *tri_a, *tri_b, *tri_c; const int32_t *tri_mat;
    int n_mats;    const float *mat_albedo; const int32_t *mat_reflective;
    int n_pl;      const float *pl_pos, *pl_col;
    int n_dl;      const float *dl_dir, *dl_col;
    float env[3];
} orc_rt_scene;

#define RT_EPS 1e-4f
#define RT_FAR 1e6f

typedef struct { float x, y, z; } v3;
static v3 v3_(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static v3 vsub(v3 a, v3 b) { return v3_(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 vadd(v3 a, v3 b) { return v3_(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 vscale(v3 a, float s) { return v3_(a.x * s, a.y * s, a.z * s); }
static float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static v3 vcross(v3 a, v3 b) { return v3_(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static v3 vnorm(v3 a) { const float d = vdot(a, a); return d > 0.0f ? vscale(a, 1.0f / sqrtf(d)) : v3_(0.0f, 0.0f, 0.0f); }
static v3 vload(const float *p) { return v3_(p[0], p[1], p[2]); }

static float rt_sphere(v3 ro, v3 rd, const float *s)            /* raytrace_shader.js:89-101 */
{
    const v3 oc = vsub(ro, vload(s));
    const float b = vdot(oc, rd);
    const float c = vdot(oc, oc) - s[3] * s[3];
    const float h = b * b - c;
    if (h < 0.0f) return -1.0f;
    const float q = sqrtf(h);
    const float t1 = -b - q, t2 = -b + q;
    if (t1 > RT_EPS) return t1;
    if (t2 > RT_EPS) return t2;
    return -1.0f;
}

static float rt_plane(v3 ro, v3 rd, const float *pl)             /* :104-109 */
{
    const v3 n = vload(pl);
    const float denom = vdot(n, rd);
    if (fabsf(denom) < 1e-6f) return -1.0f;
    const float t = (-pl[3] - vdot(n, ro)) / denom;
    return t > RT_EPS ? t : -1.0f;
}

static float rt_tri(v3 ro, v3 rd, v3 a, v3 b, v3 c, v3 *n)       /* :111-125 */
{
    const v3 e1 = vsub(b, a), e2 = vsub(c, a);
    const v3 p = vcross(rd, e2);
    const float det = vdot(e1, p);
    *n = v3_(0.0f, 0.0f, 0.0f);
    if (fabsf(det) < 1e-6f) return -1.0f;
    const float inv = 1.0f / det;
    const v3 t = vsub(ro, a);
    const float u = vdot(t, p) * inv;
    if (u < 0.0f || u > 1.0f) return -1.0f;
    const v3 q = vcross(t, e1);
    const float v = vdot(rd, q) * inv;
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    const float tt = vdot(e2, q) * inv;
    if (tt <= RT_EPS) return -1.0f;
    *n = vnorm(vcross(e1, e2));
    if (vdot(*n, rd) > 0.0f) *n = v3_(-n->x, -n->y, -n->z);
    return tt;
}

typedef struct { float t; int m; v3 n; } rt_hit;

static rt_hit rt_intersect(const orc_rt_scene *S, v3 ro, v3 rd)  /* :129-150 */
{
    rt_hit H = { RT_FAR, -1, { 0.0f, 0.0f, 0.0f } };
    for (int i = 0; i < S->n_spheres; i++) {
        const float *s = S->spheres4 + 4 * i;
        const float t = rt_sphere(ro, rd, s);
        if (t > RT_EPS && t < H.t) {
            H.t = t; H.m = S->sphere_mat[i];
            const v3 pos = vadd(ro, vscale(rd, t));
            const float r = fmaxf(s[3], 1e-6f);
            const v3 d = vsub(pos, vload(s));
            H.n = v3_(d.x / r, d.y / r, d.z / r);
        }
    }
    for (int i = 0; i < S->n_planes; i++) {
        const float t = rt_plane(ro, rd, S->planes4 + 4 * i);
        if (t > RT_EPS && t < H.t) { H.t = t; H.m = S->plane_mat[i]; H.n = vload(S->planes4 + 4 * i); }
    }
    for (int i = 0; i < S->n_tris; i++) {
        v3 n;
        const float t = rt_tri(ro, rd, vload(S->tri_a + 3 * i), vload(S->tri_b + 3 * i), vload(S->tri_c + 3 * i), &n);
        if (t > RT_EPS && t < H.t) { H.t = t; H.m = S->tri_mat[i]; H.n = n; }
    }
    return H;
}

static int rt_occluded(const orc_rt_scene *S, v3 ro, v3 rd, float max_t)   /* :152-166: spheres and triangles only */
{
    for (int i = 0; i < S->n_spheres; i++) {
        const float t = rt_sphere(ro, rd, S->spheres4 + 4 * i);
        if (t > RT_EPS && t < max_t) return 1;
    }
    for (int i = 0; i < S->n_tris; i++) {
        v3 n;
        const float t = rt_tri(ro, rd, vload(S->tri_a + 3 * i), vload(S->tri_b + 3 * i), vload(S->tri_c + 3 * i), &n);
        if (t > RT_EPS && t < max_t) return 1;
    }
    return 0;
}

static v3 rt_shade_diffuse(const orc_rt_scene *S, v3 pos, v3 N, v3 albedo)   /* :168-196 */
{
    const float z = fmaxf(N.y * 0.0f, 0.0f);
    v3 Lo = v3_(S->env[0] * z, S->env[1] * z, S->env[2] * z);
    const v3 org = vadd(pos, vscale(N, RT_EPS));
    for (int i = 0; i < S->n_dl; i++) {
        const v3 d = vload(S->dl_dir + 3 * i);
        const v3 L = vnorm(v3_(-d.x, -d.y, -d.z));
        const float ndl = fmaxf(vdot(N, L), 0.0f);
        if (ndl > 0.0f && !rt_occluded(S, org, L, 1e5f)) {
            const float *c = S->dl_col + 3 * i;
            Lo = v3_(Lo.x + (albedo.x * c[0]) * ndl, Lo.y + (albedo.y * c[1]) * ndl, Lo.z + (albedo.z * c[2]) * ndl);
        }
    }
    for (int i = 0; i < S->n_pl; i++) {
        const v3 Lv = vsub(vload(S->pl_pos + 3 * i), pos);
        const float dist2 = fmaxf(vdot(Lv, Lv), 1e-6f);
        const v3 L = vscale(Lv, 1.0f / sqrtf(dist2));
        const float ndl = fmaxf(vdot(N, L), 0.0f);
        if (ndl > 0.0f && !rt_occluded(S, org, L, sqrtf(dist2) - 2.0f * RT_EPS)) {
            const float att = 1.0f / (1.0f + dist2 * 0.05f);
            const float w = ndl * att;
            const float *c = S->pl_col + 3 * i;
            Lo = v3_(Lo.x + (albedo.x * c[0]) * w, Lo.y + (albedo.y * c[1]) * w, Lo.z + (albedo.z * c[2]) * w);
        }
    }
    return Lo;
}

static void rt_material(const orc_rt_scene *S, int m, v3 *albedo, int *reflective)   /* :71-86,218-220 */
{
    const int mi = (int)fminf(fmaxf((float)m, 0.0f), (float)(S->n_mats - 1));
    *albedo = v3_(0.8f, 0.8f, 0.8f);
    *reflective = 0;
    if (mi >= 0 && mi < S->n_mats) { *albedo = vload(S->mat_albedo + 3 * mi); *reflective = S->mat_reflective[mi] != 0; }
}

/* Camera terms of raytrace_shader.js:66-70,203-207, evaluated once per frame: basis[0..2] = uu, [3..5] = vv,
 * [6..8] = ww, [9] = focal. */
void orc_rt_camera(float yaw, float pitch, float fovy, float *basis)
{
    const double cp = cos((double)pitch), sp = sin((double)pitch), cy = cos((double)yaw), sy = sin((double)yaw);
    const v3 ww = vnorm(v3_((float)(cp * cy), (float)sp, (float)(cp * sy)));
    v3 uu = vnorm(vcross(ww, v3_(0.0f, 1.0f, 0.0f)));
    if (sqrtf(vdot(uu, uu)) < 1e-3f) uu = v3_(1.0f, 0.0f, 0.0f);
    const v3 vv = vnorm(vcross(uu, ww));
    basis[0] = uu.x; basis[1] = uu.y; basis[2] = uu.z;
    basis[3] = vv.x; basis[4] = vv.y; basis[5] = vv.z;
    basis[6] = ww.x; basis[7] = ww.y; basis[8] = ww.z;
    basis[9] = 1.0f / fmaxf(1e-6f, (float)tan(0.5 * (double)fovy));
}

/* One frame of the ray backend; rgba_out cols*rows*4, row 0 = top.  Returns 0, or -1 on bad arguments. */
int orc_rt_render(const orc_rt_scene *S, int cols, int rows, const float *cam_pos, float yaw, float pitch, float fovy,
                  float pix_aspect, uint8_t *rgba_out)
{
    if (!S || cols <= 0 || rows <= 0 || !cam_pos || !rgba_out) return -1;
    float B[10];
    orc_rt_camera(yaw, pitch, fovy, B);
    const v3 uu = vload(B), vv = vload(B + 3), ww = vload(B + 6);
    const float focal = B[9];
    const float aspect = ((float)cols / (float)rows) * fmaxf(pix_aspect, 1e-6f);
    const v3 ro = vload(cam_pos);
    for (int j = 0; j < rows; j++) {
        for (int i = 0; i < cols; i++) {
            float px = -1.0f + 2.0f * (((float)i + 0.5f) / (float)cols);
            const float py = -1.0f + 2.0f * (((float)j + 0.5f) / (float)rows);
            px = px * aspect;
            const v3 rd = vnorm(v3_((px * uu.x + py * vv.x) + focal * ww.x, (px * uu.y + py * vv.y) + focal * ww.y,
                                    (px * uu.z + py * vv.z) + focal * ww.z));
            v3 col;
            const rt_hit H = rt_intersect(S, ro, rd);
            if (H.m < 0) {
                col = vload(S->env);
            } else {
                const v3 hit = vadd(ro, vscale(rd, H.t));
                const v3 N = vnorm(H.n);
                v3 albedo; int refl;
                rt_material(S, H.m, &albedo, &refl);
                if (!refl) {
                    col = rt_shade_diffuse(S, hit, N, albedo);
                } else {
                    const float d2 = 2.0f * vdot(N, rd);
                    const v3 rdir = vsub(rd, vscale(N, d2));
                    const v3 org = vadd(hit, vscale(N, RT_EPS));
                    const rt_hit H2 = rt_intersect(S, org, rdir);
                    if (H2.m < 0) {
                        col = vload(S->env);
                    } else {
                        const v3 hit2 = vadd(org, vscale(rdir, H2.t));
                        const v3 N2 = vnorm(H2.n);
                        v3 alb2; int r2;
                        rt_material(S, H2.m, &alb2, &r2);
                        col = rt_shade_diffuse(S, hit2, N2, alb2);
                    }
                }
            }
            uint8_t *o = rgba_out + ((size_t)(rows - 1 - j) * cols + i) * 4;
            const float c3[3] = { col.x, col.y, col.z };
            for (int k = 0; k < 3; k++) o[k] = (uint8_t)(int)floorf(fminf(fmaxf(c3[k], 0.0f), 1.0f) * 255.0f + 0.5f);
            o[3] = 255;
        }
    }
    return 0;
}
