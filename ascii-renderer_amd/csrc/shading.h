// shading.h - ramp index, the fragment shader, the atlas lookup
// Part of libascii_rast.so; included by ascii_rast.hip (one translation unit), not compiled on its own.
#pragma once

namespace {

__device__ __forceinline__ int ramp_index(int r, int g, int b, int ramp_len) {
    // == clamp(round(((r+g+b)/3/255) * (L-1)), 0, L-1): the rounded value is never a tie
    return (2 * (r + g + b) * (ramp_len - 1) + 765) / 1530;
}

// raster_shader.js:42-62 in binary32.
// FP: FrameParams, or the same struct read in place from the kernel-argument segment (kernarg_params).
template <typename FP>
__device__ __forceinline__ void shade(const FP &fp, const float *v, uint8_t (&out)[4]) {
    const float len2 = (v[3] * v[3] + v[4] * v[4]) + v[5] * v[5];
    const float inv = (len2 > 0.0f) ? 1.0f / sqrtf(len2) : 0.0f;
    const float N[3] = {v[3] * inv, v[4] * inv, v[5] * inv};
    float col[3];
#pragma unroll
    for (int k = 0; k < 3; k++) col[k] = v[6 + k] * fp.amb[k];
    const float ndl = fmaxf((N[0] * (-fp.ldir[0]) + N[1] * (-fp.ldir[1])) + N[2] * (-fp.ldir[2]), 0.0f);
#pragma unroll
    for (int k = 0; k < 3; k++) col[k] = col[k] + (v[6 + k] * fp.lcol[k]) * ndl;
    for (int i = 0; i < AR_MAX_POINT_LIGHTS; i++) {
        if (i >= fp.n_pl) break;
        const float lv[3] = {fp.plpos[3 * i] - v[0], fp.plpos[3 * i + 1] - v[1], fp.plpos[3 * i + 2] - v[2]};
        const float d2 = fmaxf((lv[0] * lv[0] + lv[1] * lv[1]) + lv[2] * lv[2], 1e-4f);
        const float il = 1.0f / sqrtf(d2);
        const float Ld[3] = {lv[0] * il, lv[1] * il, lv[2] * il};
        const float ndlp = fmaxf((N[0] * Ld[0] + N[1] * Ld[1]) + N[2] * Ld[2], 0.0f);
        const float atten = 1.0f / (1.0f + d2 * 0.05f);
        const float w = ndlp * atten;
#pragma unroll
        for (int k = 0; k < 3; k++) col[k] = col[k] + (v[6 + k] * fp.plcol[3 * i + k]) * w;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float c = fminf(fmaxf(col[k], 0.0f), 1.0f);
        out[k] = uint8_t(int(floorf(c * 255.0f + 0.5f)));
    }
    out[3] = 255;
}

// Atlas lookup of one fragment at interpolated texel coordinates (u, v) (shader_utils.js:108-117: nearest
// texel floor(uv + 0.5), nothing outside the atlas; fetchAtlas :100-105 reads row height-1-y of the
// uploaded array).  Texel classes as the reference's primary ray treats them (pathtrace_shader.js:141-152):
// alpha 32..126 - the texel's RGB goes out unshaded with that alpha, an ASCII override for the
// quantizer (returns true, px is final); alpha 1 - the texel's rgb / 255 replaces the colour varying;
// anything else - untextured.
__device__ __forceinline__ bool apply_texture(const FrameParams &fp, const uint32_t *__restrict__ atlas, float u, float v,
                                              float *col, uint8_t (&px)[4]) {
    const float tx = floorf(u + 0.5f), ty = floorf(v + 0.5f);
    if (!(tx >= 0.0f) || !(ty >= 0.0f) || !(tx < float(fp.tex_w)) || !(ty < float(fp.tex_h))) return false;
    const uint32_t q = atlas[size_t(fp.tex_h - 1 - int(ty)) * fp.tex_w + int(tx)];
    const uint32_t a = q >> 24;
    if (a >= 32u && a <= 126u) {
        px[0] = uint8_t(q); px[1] = uint8_t(q >> 8); px[2] = uint8_t(q >> 16); px[3] = uint8_t(a);
        return true;
    }
    if (a == 1u) {
        col[0] = float(q & 255u) / 255.0f; col[1] = float((q >> 8) & 255u) / 255.0f; col[2] = float((q >> 16) & 255u) / 255.0f;
    }
    return false;
}

constexpr uint32_t kUvSample = 1u;  // flags word of a triangle's texel-coordinate row: sample the atlas

}  // namespace
