/*
 * ascii_rast.h - C ABI of the MI355X (gfx950) 3D -> ASCII rasterizer library (libascii_rast.so).
 *
 * This is the drop-in boundary for the reference's raster backend + ASCII stage.  Each entry point
 * names the reference interface it replaces (paths relative to the reference tree).  Plain pointers
 * and sizes only; every pointer argument is HOST memory unless its name starts with d_.
 * All functions returning int return 0 on success and a negative AR_E_* code on failure; the message
 * is available from ar_last_error().  One context is used from one host thread at a time.
 */
#ifndef ASCII_RAST_H
#define ASCII_RAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AR_MAX_POINT_LIGHTS 8   /* js/render/backends/raster_shader.js:5 */
#define AR_MAX_RAMP 256

#define AR_E_ARG     (-1)  /* invalid argument */
#define AR_E_HIP     (-2)  /* a HIP runtime call failed */
#define AR_E_STATE   (-3)  /* call made in the wrong state (e.g. read before any render) */
#define AR_E_NOMEM   (-4)

typedef struct ar_ctx ar_ctx;

/* Creates a renderer bound to one GPU and one HIP stream of its own.
 * Replaces: RasterBackend constructor + _ensureGL (js/render/backends/raster.js:52-103), i.e. what
 * the router's factory() call instantiates (js/gpu_renderer.js:74).  NULL on failure
 * (ar_last_error(NULL) tells why). */
ar_ctx *ar_create(int device_id);

/* Replaces: RasterBackend.dispose (raster.js:116-127).  NULL is accepted. */
void ar_destroy(ar_ctx *ctx);

/* Message of the most recent failure on ctx (or of the last failed ar_create when ctx is NULL). */
const char *ar_last_error(ar_ctx *ctx);

/* Uploads the geometry: vbo9 = n_verts x (pos3, nrm3, col3) binary32, idx = n_tris x 3 u32 - the
 * exact buffers RasterBackend._rebuildGeometry hands to gl.bufferData (raster.js:222-251).
 * n_tris == 0 clears the geometry (render then leaves the outputs untouched, raster.js:274).
 * Indices >= n_verts make their triangle invisible. */
int ar_set_mesh(ar_ctx *ctx, const float *vbo9, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris);

/* ar_set_mesh for an indexed triangle list, with the vertex buffer built on the GPU: positions = n_verts x 3
 * binary32, indices = n_tris x 3 u32, tri_colors = n_tris x 3 binary32 albedo or NULL for (0.8, 0.8, 0.8).
 * Produces exactly the buffers RasterBackend._rebuildGeometry builds by pushing every triangle through
 * pushTri (raster.js:158-169,222-236): three unshared vertices per triangle carrying the flat normal
 * normalize(cross(b-a, c-a)) evaluated in binary64 ((0,1,0) when its length is <= 1e-8) and the albedo,
 * indices 0..3*n_tris-1.  Replaces that O(triangles) JavaScript loop and shrinks the upload from 120 to
 * 24-36 bytes per triangle.  An index >= n_verts gives its triangle NaN positions (invisible). */
int ar_set_mesh_indexed(ar_ctx *ctx, const float *positions, uint32_t n_verts, const uint32_t *indices, uint32_t n_tris,
                        const float *tri_colors);

/* New vertex data for the current mesh, in place: data holds n_verts (the mesh's count) rows of floats_per_vertex
 * floats - 9: whole vertices (pos3, nrm3, col3) as ar_set_mesh takes them; 3: positions only, normals and colours
 * stay.  Topology, triangle indices (and with them winner ids and the tie rule) and the spatial order chosen at the
 * last upload are kept; only the streams derived from the vertices are rewritten, on the GPU, ordered after the frames
 * in flight and before the frames that follow without the host waiting for either (the copy from `data` is complete
 * when the call returns).  The reference can only move a mesh by rebuilding and uploading it whole
 * (RasterBackend.setScene -> _rebuildGeometry, raster.js:130-133,148-257); this is that per-frame rebuild for a
 * deforming mesh at a fraction of ar_set_mesh's cost: no allocation (after the first call for a mesh of this size),
 * no sort, and the host waits for nothing but its own copy.  flags:
 *   AR_UPDATE_FLAT_NORMALS  every triangle's three vertices get normalize(cross(b-a, c-a)) of the new positions,
 *                           computed as ar_set_mesh_indexed / pushTri do (raster.js:159-169) - for meshes whose
 *                           triangles do not share vertices;
 *   AR_UPDATE_RESORT        also choose a new spatial order, as an upload does (allocates and waits for the GPU):
 *                           worth it once the mesh has moved far from the shape it was uploaded in. */
#define AR_UPDATE_FLAT_NORMALS 1
#define AR_UPDATE_RESORT 2
int ar_update_positions(ar_ctx *ctx, const float *data, uint32_t n_verts, int floats_per_vertex, int flags);

/* Copies the current mesh back in ar_set_mesh's layout (each pointer may be NULL); n_verts / n_tris receive
 * its size.  Counterpart of reading back the buffers given to gl.bufferData (raster.js:238-251); lets a
 * caller check or keep what ar_set_mesh_indexed built. */
int ar_read_mesh(ar_ctx *ctx, float *vbo9, uint32_t *idx, uint32_t *n_verts, uint32_t *n_tris);

/* Integer texel coordinates for atlas texturing, one row of eight u16 per triangle of the current mesh:
 * (u0, v0, u1, v1, u2, v2, flags, 0) - the uvA/uvB/uvC of scene tris and the uv0..uv3 of quads as
 * SceneBuilder stores them (js/render/scene_api.js:139-160) and the path tracer uploads them
 * (js/render/backends/pathtrace.js:186-206,230-246).  flags bit 0: sample the atlas for this triangle
 * (sampleTriFetch samples every triangle, sampleQuadFetch skips a quad half whose three pairs are all
 * zero, js/render/backends/shader_utils.js:108-132; the caller encodes the outcome in the bit).
 * The reference's raster backend draws no textures; this brings its path tracer's primary-ray atlas
 * rule to the raster path.  uv8 == NULL removes the coordinates; ar_set_mesh removes them too. */
int ar_set_mesh_uv(ar_ctx *ctx, const uint16_t *uv8, uint32_t n_tris);

/* The RGBA8 atlas exactly as PathTraceBackend._ensureAtlasTexture hands it to texImage2D
 * (pathtrace.js:341-355): width*height*4 bytes; texel (x, y) is read from row height-1-y
 * (shader_utils.js:100-102).  Texturing is active while texel coordinates are set and both sizes exceed 1
 * (shader_utils.js:94).  Per covered cell of a sampling triangle the texel at floor(uv + 0.5) decides
 * (pathtrace_shader.js:141-152): alpha 32..126 - the cell's RGB is the texel's, unshaded, and its alpha
 * that code, which the ASCII stage emits verbatim; alpha 1 - the texel's rgb replaces the vertex colour
 * in the lighting; anything else, or a texel outside the atlas - untextured.  rgba == NULL removes it. */
int ar_set_texture(ar_ctx *ctx, const uint8_t *rgba, int width, int height);

/* Column-major projection and view matrices, as RasterBackend.render uploads to uProj / uView
 * (raster.js:296-304). */
int ar_set_camera(ar_ctx *ctx, const float proj[16], const float view[16]);

/* A model matrix for the current mesh (column-major 4 x 4; NULL removes it), applied per frame on the GPU: the
 * mesh is drawn as if the reference's vertex shader (js/render/backends/raster_shader.js:17-22) read
 *     gl_Position = uProj * uView * uModel * vec4(a_pos, 1);  v_pos = (uModel * vec4(a_pos, 1)).xyz;
 *     v_nrm = mat3(uModel) * a_nrm;
 * with the matrix product (uProj * uView) * uModel formed first, as GLSL evaluates it.  The reference's rasterizer
 * has no model matrix (raster_shader.js:12-13) - its scenes move only by being rebuilt and uploaded again
 * (raster.js:130-133) - so this replaces a setScene per frame for rigid motion: a spinning mesh costs sixteen
 * floats per frame instead of a new vertex buffer.  Normals go through the upper-left 3 x 3 unchanged, which is
 * exact for rotations, translations and uniform scales (the fragment shader normalises). */
int ar_set_model(ar_ctx *ctx, const float model[16]);

/* Light uniforms as packed by raster.js:307-342: uAmbient, uLightDir, uLightColor, uPLCount and the
 * first n_pl entries of uPLPos / uPLCol (n_pl <= AR_MAX_POINT_LIGHTS, 3 floats each). */
int ar_set_lights(ar_ctx *ctx, const float amb[3], const float dir[3], const float dircol[3],
                  int n_pl, const float *plpos, const float *plcol);

/* ASCII stage parameters: the ramp string baked into the shader (js/ascii_pass_shader.js:8-16,
 * js/ascii_pass.js:142) and the uModeOn / uModeRadius / uModeThresh uniforms (ascii_pass.js:294-299).
 * Defaults: ramp "@%#*+=-:. ", mode on, radius 2, threshold 12 (js/config.js:15,26-34). */
int ar_set_ascii(ar_ctx *ctx, const uint8_t *ramp, int ramp_len, int mode_on, int mode_radius, int mode_thresh);

/* Restricts rendering to output rows [row0, row1) of the frame (screen-strip partition for multi-GPU
 * runs; no counterpart in the single-GPU reference).  row0 = 0, row1 = 0 means the whole frame.
 * ASCII rows outside the strip are left untouched.  With smoothing on, the rgba / depth (and id) planes are
 * also written for up to three rows on either side of the strip - the smoothing window's halo, rendered
 * exactly as a full frame would render them; rows beyond that are left untouched. */
int ar_set_strip(ar_ctx *ctx, int row0, int row1);

/* Renders one frame at cols x rows cells and blocks until done.
 * Replaces: RasterBackend.render from the clear to readPixels + flipAndCopy (raster.js:276-361,
 * js/render/gl/context.js:117-124) plus the index half of AsciiPass.draw (js/ascii_pass.js:257-302,
 * js/ascii_pass_shader.js:140-188).  Outputs (each may be NULL; all row 0 = top of the image):
 *   rgba  cols*rows*4 bytes, alpha 255 (an atlas texel's code where ar_set_texture says so);
 *   depth cols*rows binary32 window depth, 1.0 = background;
 *   ascii cols*rows ASCII codes (== glyph-atlas tile indices, ascii_pass_shader.js:190-199).
 * With no geometry the call succeeds and leaves the output buffers untouched. */
int ar_render(ar_ctx *ctx, int cols, int rows, uint8_t *rgba, float *depth, uint8_t *ascii);

/* The same frame without the device->host copies: enqueue on the context's stream / wait for it /
 * copy out what is wanted afterwards.  Lets a caller keep the outputs resident in HBM (benchmarks,
 * multi-GPU gathers).  ids (per cell 2*triangle + sub-triangle of the winner, 0xFFFFFFFF background;
 * a debugging output with no counterpart in the reference) is only produced by frames rendered
 * after ar_set_id_output(ctx, 1). */
int ar_set_id_output(ar_ctx *ctx, int on);
int ar_render_async(ar_ctx *ctx, int cols, int rows);
int ar_sync(ar_ctx *ctx);
int ar_read(ar_ctx *ctx, uint8_t *rgba, float *depth, uint8_t *ascii, uint32_t *ids);

/* Several frames in ONE chain of kernel launches: n_frames frames of cols x rows cells that differ in their camera only -
 * cameras holds, per frame, the 16 floats of the projection and the 16 of the view matrix as ar_set_camera takes them
 * (32 floats per frame) - are enqueued together; every kernel of the frame runs once for all of them.  Mesh, model
 * matrix, lights, ASCII parameters, strip and options are the context's, as for ar_render_async; the context's own
 * camera is left as it was.  A frame whose cost is launches and the hand-over between dependent kernels rather than
 * work - a small scene, a strip of the frame on one of several GPUs - costs a fraction of that this way.  The reference
 * renders a frame per call (RasterBackend.render, raster.js:269-362); its application calls that once per animation
 * tick (js/main.js:364-375), so a batch is several ticks' cameras handed over at once.
 * The frames' planes are kept apart from those of single frames: ar_read_batch copies frame `frame` (0 .. n_frames-1) of
 * the most recent batch to host memory (any pointer may be NULL; blocks), ar_batch_device_buffer gives the device address
 * of one of its planes (which: 0 rgba, 1 depth, 2 ascii).  While an ASCII target is set (ar_set_ascii_target) frame f's
 * ASCII rows go to the target + f * ascii_stride bytes instead (ascii_stride is ignored otherwise).  Consecutive
 * batches overlap on the context's frame streams as single frames do; ar_sync waits for them.  At most 256 frames a call. */
int ar_render_batch_async(ar_ctx *ctx, int cols, int rows, int n_frames, const float *cameras, size_t ascii_stride);
int ar_read_batch(ar_ctx *ctx, int frame, uint8_t *rgba, float *depth, uint8_t *ascii, uint32_t *ids);
void *ar_batch_device_buffer(ar_ctx *ctx, int which, int frame);

/* Read-back that does not wait: queues the device->host copies of the most recently enqueued frame's planes (each
 * pointer may be NULL, as for ar_read) behind that frame, on the stream it runs on, and returns at once with a
 * ticket; ar_wait blocks until the frame and its copies are complete, ar_poll asks without blocking (1 complete,
 * 0 still running).  The counterpart in the reference is its caller's frame in flight: js/main.js holds two
 * framebuffers and a gpuInFlight guard (main.js:243-248,364-375) around a render that in fact blocks in readPixels
 * (raster.js:358-361); here up to three frames (AR_OPT_OVERLAP) render and copy while the caller does other work.
 * The host buffers must stay valid until the ticket has been waited for and should be page-locked (ar_host_alloc):
 * into pageable memory the call itself waits for the copy.  The next frame on the same stream - the third
 * ar_render_async after this one while frames overlap, the next one otherwise - starts after the copies, so the
 * device buffers are never overwritten under them.  A ticket stays valid until the context is destroyed; waiting
 * for an old one is harmless.  All pointers NULL gives a ticket for the frame alone. */
int ar_read_async(ar_ctx *ctx, uint8_t *rgba, float *depth, uint8_t *ascii, uint32_t *ids, uint64_t *ticket);
int ar_wait(ar_ctx *ctx, uint64_t ticket);
int ar_poll(ar_ctx *ctx, uint64_t ticket);

/* Device addresses of the output buffers of the most recently enqueued frame: which = 0 rgba, 1 depth,
 * 2 ascii.  NULL before the first render.  While frames overlap (AR_OPT_OVERLAP) consecutive frames write
 * different sets of buffers in rotation, so ask again after each ar_render_async; with AR_OPT_OVERLAP 0, with
 * ar_set_stream and for ar_render the addresses only change with the grid size. */
void *ar_device_buffer(ar_ctx *ctx, int which);

/* The HIP stream (hipStream_t) of frames that do not overlap.  Overlapping frames use further streams of
 * the context (see AR_OPT_OVERLAP); to order other work against frames use ar_sync or the calls below. */
void *ar_stream(ar_ctx *ctx);

/* Page-locked host memory for output buffers.  ar_render / ar_read copy into any host pointer, but into
 * ordinary pageable memory the copy is bounced through the runtime's staging buffers at about a third of
 * the PCIe rate; a framebuffer from ar_host_alloc is written by DMA directly.  No counterpart in the
 * reference (its framebuffers are plain typed arrays, js/main.js:243-248).  ar_host_free(NULL) is a no-op. */
void *ar_host_alloc(ar_ctx *ctx, size_t bytes);
void ar_host_free(void *p);

/* Runs every kernel of the following frames on the caller's stream (hipStream_t of the context's device)
 * instead of the context's own streams: frames are then ordered with the caller's other work on that
 * stream (an RCCL collective, a copy) by the stream itself, at the price of frames no longer overlapping.
 * NULL returns to the context's own streams.  No counterpart in the reference. */
int ar_set_stream(ar_ctx *ctx, void *hip_stream);

/* Lets the following frames store the ASCII plane in the caller's device buffer instead of the
 * context's: frame row r is written at d_ascii + (r - first_row) * cols, only for the rows of the
 * current strip, so a strip can be written densely into a slot of a larger gather buffer.  While a
 * target is set, ar_read / ar_render return the ASCII plane of the last frame rendered without one.
 * d_ascii must be 4-byte aligned for frames whose row length is a multiple of four (codes are then stored four at a
 * time; such a frame fails with AR_E_ARG otherwise); any address will do for other row lengths.  NULL returns to the
 * context's own buffer.  No counterpart in the reference. */
int ar_set_ascii_target(ar_ctx *ctx, void *d_ascii, int first_row);

/* Stream-ordered hand-over of the outputs, without blocking the host (no counterpart in the reference;
 * used to feed the ASCII strips to RCCL):
 *   ar_stream_wait_frame  work enqueued on hip_stream after this call starts only when every output of
 *                         the frames enqueued so far is complete;
 *   ar_frame_wait_stream  frames enqueued after this call write the output buffers only when the work
 *                         enqueued on hip_stream so far is complete.
 * hip_stream is a hipStream_t of the context's device. */
int ar_stream_wait_frame(ar_ctx *ctx, void *hip_stream);
int ar_frame_wait_stream(ar_ctx *ctx, void *hip_stream);

/* RGBA8 grid -> ASCII codes with the context's ASCII parameters; alpha 2..254 is a literal code.
 * Replaces: AsciiPass.drawFromBuffer = blit + draw, index half (ascii_pass.js:238-302) and its CPU
 * twins compositeAndQuantize (js/renderer.js:59-72) / _computeRowString (js/text_overlay.js:128-148). */
int ar_quantize(ar_ctx *ctx, const uint8_t *rgba, int cols, int rows, uint8_t *ascii);

/* ar_quantize for callers that cannot allocate: same arguments, same result, but the grid and the ASCII plane are
 * staged in a fixed device array that is part of the library's code object (AR_STATIC_MAX_CELLS x 6 bytes on every
 * GPU the library runs on, held until the process ends), so the call makes no hipMalloc / hipFree.  It has NO
 * counterpart in the reference; it exists for callers that cannot allocate: tight serving loops, harnesses that
 * forbid allocation, threads that run while another captures a HIP graph (the call itself waits for its stream and
 * copies host memory, so it cannot be part of a capture).  Grids of more than AR_STATIC_MAX_CELLS cells are rejected with AR_E_ARG.  All calls on one GPU
 * share the array (whatever context they are made on): calls that overlap on one GPU race. */
#define AR_STATIC_MAX_CELLS (1920 * 1080)
int ar_quantize_static(ar_ctx *ctx, const uint8_t *rgba, int cols, int rows, uint8_t *ascii);

/* RGBA8 grid -> RGBA8 image of coloured glyphs, cols*cell_w x rows*cell_h pixels, row 0 = top.
 * Replaces: AsciiPass.drawFromBuffer in full - the index half as ar_quantize, then the compositing half of
 * the fragment shader (js/ascii_pass_shader.js:190-230): per pixel the nearest texel of the cell's glyph,
 * coverage = pow(alpha / 255, alpha_gamma) (uAlphaGamma), colour = mix(white, cell colour, coverage)
 * (black text when grayscale, uGray), alpha 255; with transparent_bg (uTransparentBG) pixels of zero
 * coverage are (0,0,0,0) instead of white.  atlas_alpha is the alpha plane of the reference's glyph atlas
 * layout (js/ascii_pass.js:20-60): 16 x 16 tiles of (cell_w + 4) x (cell_h + 4) bytes (2 bytes of padding
 * all round), glyph g at tile (g % 16, g / 16), row 0 = top; the reference bakes it with a canvas font,
 * here the caller supplies it.  The canvas is exactly cols*cell_w x rows*cell_h (at most 2^20 pixels wide and 65535 pixel rows). */
int ar_present(ar_ctx *ctx, const uint8_t *rgba, int cols, int rows, const uint8_t *atlas_alpha, int cell_w, int cell_h,
               double alpha_gamma, int grayscale, int transparent_bg, uint8_t *image);

/* RGBA8 grid -> packed text colours (r << 16) | (g << 8) | b of the reference's DOM presentation.
 * Replaces: the colour half of compositeAndQuantize (js/renderer.js:13-19,28-30,74-84): each channel
 * snapped to `steps` levels, qChan(u8) = round(round(u8 / 255 * (steps - 1)) * 255 / (steps - 1));
 * steps <= 0 takes the reference's max(3, ramp length) (so that colours match its stylesheet classes,
 * renderer.js:162-185); cells with alpha 2..254 (UI cells, renderer.js:38-44) and, with grayscale set
 * (config.USE_GRAYSCALE, renderer.js:76-77), all cells are black.
 * ar_read_colors does the same for the RGBA plane of the last rendered frame. */
int ar_quantize_colors(ar_ctx *ctx, const uint8_t *rgba, int cols, int rows, int steps, int grayscale, uint32_t *colors);
int ar_read_colors(ar_ctx *ctx, int steps, int grayscale, uint32_t *colors);

/* ---- ray backend -----------------------------------------------------------------------------------
 * The reference's deterministic single-bounce ray tracer (js/render/backends/raytrace.js,
 * raytrace_shader.js) behind the same context: spheres, planes and triangles, diffuse or mirror materials,
 * point and directional lights with hard shadows, one mirror bounce, environment colour on a miss. */
typedef struct ar_rt_scene {   /* the uniform arrays RaytraceBackend._packScene builds (raytrace.js:195-311) */
    int n_spheres; const float *spheres4;   const int32_t *sphere_mat;     /* uNumS, uSpheres (xyz, r), uSm */
    int n_planes;  const float *planes4;    const int32_t *plane_mat;      /* uNumP, uPlanes (n, d), uPm */
    int n_tris;    const float *tri_a, *tri_b, *tri_c; const int32_t *tri_mat; /* uNumT, uTa, uTb, uTc, uTm */
    int n_mats;    const float *mat_albedo; const int32_t *mat_reflective; /* uNumMats, uMatAlbedo, uMatReflective */
    int n_pl;      const float *pl_pos, *pl_col;                           /* uNumPL, uPLPos, uPLCol */
    int n_dl;      const float *dl_dir, *dl_col;                           /* uNumDL, uDLDir, uDLCol */
    float env[3];                                                          /* uEnv */
} ar_rt_scene;

/* Replaces the scene uniform uploads of RaytraceBackend.render (raytrace.js:349-389).  Arrays hold exactly
 * n_* entries (3 or 4 floats each as in the shader's declarations); they are copied.  A material index
 * below zero makes its primitive count as a miss, indices beyond the table clamp to its last entry, an empty
 * table means albedo 0.8 diffuse (raytrace_shader.js:71-86,214-220). */
int ar_rt_set_scene(ar_ctx *ctx, const ar_rt_scene *scene);

/* Renders one frame of the ray backend at cols x rows cells and blocks until done.
 * Replaces: RaytraceBackend.render from the camera uniforms to readPixels + flipAndCopy (raytrace.js:330-402)
 * plus the index half of AsciiPass.draw, as ar_render does for the rasterizer.  cam_pos / yaw / pitch / fovy /
 * pix_aspect are uCamPos, uYaw, uPitch, uFovY (radians), uPixAspect.  Outputs (each may be NULL, row 0 = top):
 * rgba cols*rows*4 (alpha 255), ascii cols*rows codes; ar_set_ascii and ar_set_strip apply as for ar_render.
 * Without a scene the call succeeds and leaves the outputs untouched (raytrace.js:321). */
int ar_rt_render(ar_ctx *ctx, int cols, int rows, const float cam_pos[3], float yaw, float pitch, float fovy,
                 float pix_aspect, uint8_t *rgba, uint8_t *ascii);

/* The same frame without the copies: enqueue, then ar_sync / ar_read / ar_device_buffer as after
 * ar_render_async (the depth plane is not written by ray frames). */
int ar_rt_render_async(ar_ctx *ctx, int cols, int rows, const float cam_pos[3], float yaw, float pitch, float fovy,
                       float pix_aspect);

/* Per-kernel timing with HIP events on the context's stream.  While on, every launch is bracketed by
 * events; ar_kernel_stats drains them (blocking) and reports, per kernel slot, the number of launches
 * and the summed duration in milliseconds since the last call.  names receives static strings. */
#define AR_N_KERNELS 12
int ar_set_profiling(ar_ctx *ctx, int on);
int ar_kernel_stats(ar_ctx *ctx, const char *names[AR_N_KERNELS], uint32_t launches[AR_N_KERNELS], float total_ms[AR_N_KERNELS]);

/* Counters of the last rendered frame: [0] triangles submitted; direct path: [1] records queued for
 * the medium-triangle pass, [2] for the large-triangle pass; tiled path: [3] (sub-triangle, bin) entries
 * binned, [4] 1 when the record pool overflowed (the frame is still correct, but slower), [5] 1 when
 * the frame took the tiled path; [6..7] reserved. */
int ar_frame_counters(ar_ctx *ctx, uint32_t out[8]);

/* Tuning / test switches (no counterpart in the reference).
 *   AR_OPT_PATH          0 (default): tile-binned path whenever the grid has at most 8000 tiles of
 *                        64 cells by the frame's tile height, else the direct path; 1: always the direct path (every
 *                        fragment resolved with a 64-bit atomic min in HBM).
 *   AR_OPT_OVERLAP       1 (default): consecutive ar_render_async frames of the tiled path run on three
 *                        streams of the context in rotation, each with its own set of buffers, so that up to
 *                        three frames share the GPU (no events between them); 0: one stream, frames strictly
 *                        one after the other.  ar_render (blocking) never overlaps.
 *   AR_OPT_POOL_RECORDS  capacity of the tiled path's (sub-triangle, bin) entry pool in entries;
 *                        0 (default) sizes it as 2 x triangles + 65536.
 *   AR_OPT_TILE_ROWS     height of the tiled path's tiles: 0 (default) picks 32 rows, or 16 or 8 when the frame
 *                        (a strip, a small grid) would otherwise have fewer than 512 tiles; 8, 16, 32 fix it.
 *   AR_OPT_WORK_CLASSES  1 (default): every tile orders its list of triangles by a work class (rows and width of the
 *                        cell range, carried in spare bits of the bin entries) so that the lanes of a wavefront get
 *                        triangles of one kind; 0: lists stay in entry order.  Same frames either way.
 *   AR_OPT_FOLD_SCAN     1 (default): frames with at most 160 bins (small grids, strips) run without the scan kernel of
 *                        the binning step, its work folded into the kernels before and after it (chains of four or more
 *                        frames, ar_render_batch_async, always keep it: they share the launch); 0: never.
 *   AR_OPT_SMOOTHING_KERNEL  which kernel runs the modal smoothing.  0 (default): the fastest one that covers the
 *                        parameters.  For thresholds of at least half the neighbours (the reference's 12 of 24), ramps of
 *                        at most ten glyphs and radius 1 or 2 there is a bit-sliced kernel (56 cells per lane at a time),
 *                        taken for ar_render_batch_async and for RGBA input (ar_quantize, textured frames); single rendered
 *                        frames and lower thresholds take a kernel that counts per cell; radius 3 and longer ramps the
 *                        shader's two passes as written.  1: never the bit-sliced kernel; 2: only the two passes as
 *                        written; 3: the bit-sliced kernel wherever it covers the parameters.  Same codes whichever runs. */
#define AR_OPT_PATH 1
#define AR_OPT_POOL_RECORDS 2
#define AR_OPT_OVERLAP 3
#define AR_OPT_TILE_ROWS 4
#define AR_OPT_WORK_CLASSES 5
#define AR_OPT_FOLD_SCAN 6
#define AR_OPT_SMOOTHING_KERNEL 7
int ar_set_option(ar_ctx *ctx, int option, long long value);

/* ---- several GPUs in one process ------------------------------------------------------------------------
 * The reference drives one GPU (js/gpu_renderer.js:92-95).  A group renders one frame on n GPUs: the rows are cut into
 * n strips (ar_strip_rows), device i renders strip i from its own copy of the scene on a host thread of its own, and
 * every `batch` frames one RCCL all-gather per device combines the ASCII strips, after which every device holds every
 * frame of the batch.  No other host language is needed for it than the one that calls this C ABI.
 *
 *   ar_strip_rows      rows [row0, row1) of strip `strip` when `rows` rows are cut into n_strips: heights differ by at most
 *                      one (the first rows % n_strips strips are the taller ones), so any count divides; plain arithmetic,
 *                      needs no GPU.
 *   ar_group_create    a context per device of device_ids[0..n) (each at most once), an RCCL communicator over them (RCCL is
 *                      loaded at the first call) and a worker thread per device.  NULL on failure (ar_group_last_error(NULL)).
 *   ar_group_ctx       member i's context, for the per-scene calls - ar_set_mesh*, ar_set_lights, ar_set_ascii, ar_set_model,
 *                      ar_set_mesh_uv, ar_set_texture, ar_set_option - which the caller makes on every member, and only while
 *                      the group is idle (before the first frame or after ar_group_sync).  The group itself sets each member's
 *                      strip, ASCII target and camera per frame.
 *   ar_group_render_async  enqueues one frame of cols x rows cells with the given camera (as ar_set_camera takes it) on every
 *                      member and returns at once; the members' threads issue the launches.  The first frame of a new grid
 *                      size waits for the frames before it and allocates the gather rings.
 *   ar_group_sync      gathers what is left of the current batch and waits for everything; reports the first failure of
 *                      any member since the last call.
 *   ar_group_frames_gathered  frames held by the last completed gather (batch, or fewer after ar_group_sync cut one short).
 *   ar_group_read      after a sync: the whole ASCII grid (cols*rows codes, row 0 = top) of frame `frame` (0 = oldest) of the
 *                      last gathered batch, assembled from the strips member `member` received - every member holds them all. */
typedef struct ar_group ar_group;
int ar_strip_rows(int rows, int n_strips, int strip, int *row0, int *row1);
ar_group *ar_group_create(int n, const int *device_ids, int batch);
void ar_group_destroy(ar_group *group);
const char *ar_group_last_error(ar_group *group);
int ar_group_size(ar_group *group);
ar_ctx *ar_group_ctx(ar_group *group, int member);
int ar_group_render_async(ar_group *group, int cols, int rows, const float proj[16], const float view[16]);
int ar_group_sync(ar_group *group);
int ar_group_frames_gathered(ar_group *group);
int ar_group_read(ar_group *group, int member, int frame, uint8_t *ascii);

#ifdef __cplusplus
}
#endif
#endif /* ASCII_RAST_H */
