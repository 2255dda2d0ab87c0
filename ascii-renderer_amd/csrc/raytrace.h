// raytrace.h - ray backend: the reference's single-bounce ray tracer as one kernel
// Part of libascii_rast.so; included by ascii_rast.hip (one translation unit), not compiled on its own.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------
// Ray backend: the reference's deterministic single-bounce ray tracer (raytrace_shader.js:62-242) as one
// kernel, a lane per cell.  Binary32 throughout, no fused multiply-add, expressions left to right;
// normalize(v) = v * (1 / sqrt(dot(v, v))) with the zero vector staying zero, reflect(I, N) =
// I - (2 dot(N, I)) N, max / min = fmaxf / fminf.  The scene arrays are read at addresses that are the same
// for every lane (scalar loads); only the hit tests diverge.
// ------------------------------------------------------------------------------------------------
struct RtScene {  // device pointers into one allocation; counts as uploaded by raytrace.js:351-389
    int n_spheres, n_planes, n_tris, n_mats, n_pl, n_dl;
    const float *spheres4, *planes4, *tri_a, *tri_b, *tri_c, *mat_albedo, *pl_pos, *pl_col, *dl_dir, *dl_col;
    const int32_t *sphere_mat, *plane_mat, *tri_mat, *mat_reflective;
    float env[3];
};

struct RtCamera {  // per-frame terms of raytrace_shader.js:200-210, evaluated on the host
    float ro[3], uu[3], vv[3], ww[3];
    float focal, aspect;  // aspect = (cols / rows) * max(pixel aspect, 1e-6)
};

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 vload(const float *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 vnorm(V3 a) {
    const float d = vdot(a, a);
    return d > 0.0f ? a * (1.0f / sqrtf(d)) : V3{0.0f, 0.0f, 0.0f};
}

constexpr float kRtEps = 1e-4f, kRtFar = 1e6f;

__device__ __forceinline__ float rt_sphere(V3 ro, V3 rd, const float *s) {  // raytrace_shader.js:89-101
    const V3 oc = ro - vload(s);
    const float b = vdot(oc, rd);
    const float c = vdot(oc, oc) - s[3] * s[3];
    const float h = b * b - c;
    if (h < 0.0f) return -1.0f;
    const float q = sqrtf(h);
    const float t1 = -b - q, t2 = -b + q;
    if (t1 > kRtEps) return t1;
    if (t2 > kRtEps) return t2;
    return -1.0f;
}

__device__ __forceinline__ float rt_plane(V3 ro, V3 rd, const float *pl) {  // :104-109
    const V3 n = vload(pl);
    const float denom = vdot(n, rd);
    if (fabsf(denom) < 1e-6f) return -1.0f;
    const float t = (-pl[3] - vdot(n, ro)) / denom;
    return t > kRtEps ? t : -1.0f;
}

__device__ __forceinline__ float rt_tri(V3 ro, V3 rd, V3 a, V3 b, V3 c, V3 &n) {  // :111-125
    const V3 e1 = b - a, e2 = c - a;
    const V3 p = vcross(rd, e2);
    const float det = vdot(e1, p);
    n = v3(0.0f, 0.0f, 0.0f);
    if (fabsf(det) < 1e-6f) return -1.0f;
    const float inv = 1.0f / det;
    const V3 t = ro - a;
    const float u = vdot(t, p) * inv;
    if (u < 0.0f || u > 1.0f) return -1.0f;
    const V3 q = vcross(t, e1);
    const float v = vdot(rd, q) * inv;
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    const float tt = vdot(e2, q) * inv;
    if (tt <= kRtEps) return -1.0f;
    n = vnorm(vcross(e1, e2));
    if (vdot(n, rd) > 0.0f) n = v3(-n.x, -n.y, -n.z);
    return tt;
}

struct RtHit { float t; int m; V3 n; };

__device__ __forceinline__ RtHit rt_intersect(const RtScene &S, V3 ro, V3 rd) {  // :129-150
    RtHit H{kRtFar, -1, V3{0.0f, 0.0f, 0.0f}};
    for (int i = 0; i < S.n_spheres; i++) {
        const float *s = S.spheres4 + 4 * i;
        const float t = rt_sphere(ro, rd, s);
        if (t > kRtEps && t < H.t) {
            H.t = t; H.m = S.sphere_mat[i];
            const V3 pos = ro + rd * t;
            const float r = fmaxf(s[3], 1e-6f);
            const V3 d = pos - vload(s);
            H.n = v3(d.x / r, d.y / r, d.z / r);
        }
    }
    for (int i = 0; i < S.n_planes; i++) {
        const float t = rt_plane(ro, rd, S.planes4 + 4 * i);
        if (t > kRtEps && t < H.t) { H.t = t; H.m = S.plane_mat[i]; H.n = vload(S.planes4 + 4 * i); }
    }
    for (int i = 0; i < S.n_tris; i++) {
        V3 n;
        const float t = rt_tri(ro, rd, vload(S.tri_a + 3 * i), vload(S.tri_b + 3 * i), vload(S.tri_c + 3 * i), n);
        if (t > kRtEps && t < H.t) { H.t = t; H.m = S.tri_mat[i]; H.n = n; }
    }
    return H;
}

__device__ __forceinline__ bool rt_occluded(const RtScene &S, V3 ro, V3 rd, float max_t) {  // :152-166: no planes
    for (int i = 0; i < S.n_spheres; i++) {
        const float t = rt_sphere(ro, rd, S.spheres4 + 4 * i);
        if (t > kRtEps && t < max_t) return true;
    }
    for (int i = 0; i < S.n_tris; i++) {
        V3 n;
        const float t = rt_tri(ro, rd, vload(S.tri_a + 3 * i), vload(S.tri_b + 3 * i), vload(S.tri_c + 3 * i), n);
        if (t > kRtEps && t < max_t) return true;
    }
    return false;
}

__device__ __forceinline__ V3 rt_shade_diffuse(const RtScene &S, V3 pos, V3 N, V3 albedo) {  // :168-196
    const float z = fmaxf(N.y * 0.0f, 0.0f);
    V3 Lo = v3(S.env[0] * z, S.env[1] * z, S.env[2] * z);
    const V3 org = pos + N * kRtEps;
    for (int i = 0; i < S.n_dl; i++) {
        const V3 d = vload(S.dl_dir + 3 * i);
        const V3 L = vnorm(v3(-d.x, -d.y, -d.z));
        const float ndl = fmaxf(vdot(N, L), 0.0f);
        if (ndl > 0.0f && !rt_occluded(S, org, L, 1e5f)) {
            const float *c = S.dl_col + 3 * i;
            Lo = v3(Lo.x + (albedo.x * c[0]) * ndl, Lo.y + (albedo.y * c[1]) * ndl, Lo.z + (albedo.z * c[2]) * ndl);
        }
    }
    for (int i = 0; i < S.n_pl; i++) {
        const V3 Lv = vload(S.pl_pos + 3 * i) - pos;
        const float dist2 = fmaxf(vdot(Lv, Lv), 1e-6f);
        const V3 L = Lv * (1.0f / sqrtf(dist2));
        const float ndl = fmaxf(vdot(N, L), 0.0f);
        if (ndl > 0.0f && !rt_occluded(S, org, L, sqrtf(dist2) - 2.0f * kRtEps)) {
            const float att = 1.0f / (1.0f + dist2 * 0.05f);
            const float w = ndl * att;
            const float *c = S.pl_col + 3 * i;
            Lo = v3(Lo.x + (albedo.x * c[0]) * w, Lo.y + (albedo.y * c[1]) * w, Lo.z + (albedo.z * c[2]) * w);
        }
    }
    return Lo;
}

__device__ __forceinline__ void rt_material(const RtScene &S, int m, V3 &albedo, bool &reflective) {  // :71-86,218-220
    const int mi = int(fminf(fmaxf(float(m), 0.0f), float(S.n_mats - 1)));
    albedo = v3(0.8f, 0.8f, 0.8f);
    reflective = false;
    if (mi >= 0 && mi < S.n_mats) { albedo = vload(S.mat_albedo + 3 * mi); reflective = S.mat_reflective[mi] != 0; }
}

// One lane per cell of output rows [ya, yb): primary ray, nearest hit, direct light with hard shadows, or one
// mirror bounce (raytrace_shader.js:198-242).  RGBA8, alpha 255, row 0 = top.
__global__ __launch_bounds__(256) void k_raytrace(RtScene S, RtCamera cam, int cols, int rows, int ya, int yb,
                                                  uint32_t *__restrict__ rgba) {
    const size_t n = size_t(yb - ya) * cols;
    const size_t p = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n) return;
    const int r = ya + int(p / size_t(cols)), i = int(p % size_t(cols));
    const int j = rows - 1 - r;  // GL window row
    float px = -1.0f + 2.0f * ((float(i) + 0.5f) / float(cols));
    const float py = -1.0f + 2.0f * ((float(j) + 0.5f) / float(rows));
    px = px * cam.aspect;
    const V3 ro = vload(cam.ro), uu = vload(cam.uu), vv = vload(cam.vv), ww = vload(cam.ww);
    const V3 rd = vnorm(v3((px * uu.x + py * vv.x) + cam.focal * ww.x, (px * uu.y + py * vv.y) + cam.focal * ww.y,
                           (px * uu.z + py * vv.z) + cam.focal * ww.z));
    V3 col;
    const RtHit H = rt_intersect(S, ro, rd);
    if (H.m < 0) {
        col = vload(S.env);
    } else {
        const V3 hit = ro + rd * H.t;
        const V3 N = vnorm(H.n);
        V3 albedo; bool refl;
        rt_material(S, H.m, albedo, refl);
        if (!refl) {
            col = rt_shade_diffuse(S, hit, N, albedo);
        } else {
            const float d2 = 2.0f * vdot(N, rd);
            const V3 rdir = rd - N * d2;
            const V3 org = hit + N * kRtEps;
            const RtHit H2 = rt_intersect(S, org, rdir);
            if (H2.m < 0) {
                col = vload(S.env);
            } else {
                const V3 hit2 = org + rdir * H2.t;
                const V3 N2 = vnorm(H2.n);
                V3 alb2; bool r2;
                rt_material(S, H2.m, alb2, r2);
                col = rt_shade_diffuse(S, hit2, N2, alb2);
            }
        }
    }
    const float c3[3] = {col.x, col.y, col.z};
    uint32_t out = 0xFF000000u;
#pragma unroll
    for (int k = 0; k < 3; k++) out |= uint32_t(int(floorf(fminf(fmaxf(c3[k], 0.0f), 1.0f) * 255.0f + 0.5f))) << (8 * k);
    rgba[size_t(r) * cols + i] = out;
}

}  // namespace
