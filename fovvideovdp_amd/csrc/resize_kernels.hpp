// Full-screen resize of planar YUV frames ("optional interpolate resize"): what the reference's file reader does when the
// CLI is given --full-screen-resize (video_reader_yuv_pytorch.unpack with resize_fn, pyfvvdp/video_source_file.py:219-244): the frame is
// unpacked to RGB at its own resolution WITHOUT clipping, resized in RGB with torch.nn.functional.interpolate(mode = bilinear | bicubic |
// nearest | area, align_corners=False, no antialiasing) to the display's resolution, clipped to [0,1]; _prepare_frame (:355-363) then applies
// the display model and the RGB -> luminance weights.  Two kernels per frame and stream:
//   yuv_rgb_planar_kernel   YUV planes -> fp32 RGB planes [3][H][W] at the source resolution (fixed -> float, 4:2:0 chroma bilinear x2, matrix)
//   resize_lum_kernel<MODE> one thread per OUTPUT pixel: interpolate the three planes with torch's index arithmetic, clip, display model,
//                           luminance -> one fp32 luminance frame [Ho][Wo] (what the metric's per-frame feeder takes), optionally the
//                           clipped RGB planes as well (parity tests against the reference's unpack)
// Not on the benchmarked path and not tuned beyond coalesced accesses: the source planes of a frame (3 x 4 B per source pixel) stay in the
// L2 / memory-side cache between the two kernels; the luminance frame then takes the float temporal kernels.
#pragma once

struct YuvRgbArgs {
    const void* src;         // one frame: Y plane, U plane, V plane
    int W, H, uvw, uvh, chroma420;
    float wy, wc;            // 1/(2^(b-8)*219), 1/(2^(b-8)*224)
    float m[9];              // ycbcr2rgb, row-major
    float* out;              // [3][H][W]
};

template <typename T>
__global__ __launch_bounds__(256) void yuv_rgb_planar_kernel(const YuvRgbArgs a) {
    const int HW = a.W * a.H;
    const int p = (int)(blockIdx.x * 256 + threadIdx.x);
    if (p >= HW) return;
    const T* f = reinterpret_cast<const T*>(a.src);
    const int y = p / a.W, x = p - y * a.W;
    const float Yf = fminf(fmaxf(a.wy * (float)f[p] - (16.0f / 219.0f), 0.0f), 1.0f);
    const T* U = f + HW;
    const T* V = U + a.uvw * a.uvh;
    auto cf = [&](const T* pl, int yy, int xx) {
        return fminf(fmaxf(a.wc * (float)pl[yy * a.uvw + xx] - (128.0f / 224.0f), -0.5f), 0.5f);
    };
    float u, v;
    if (a.chroma420) {       // torch bilinear x2, align_corners=False (video_source_file.py:268-270), as yuv_lum in temporal_kernels.hpp
        const float sy = fmaxf(((float)y + 0.5f) * 0.5f - 0.5f, 0.0f), sx = fmaxf(((float)x + 0.5f) * 0.5f - 0.5f, 0.0f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = min(y0 + 1, a.uvh - 1), x1 = min(x0 + 1, a.uvw - 1);
        const float fy = sy - (float)y0, fx = sx - (float)x0;
        const float gy = 1.0f - fy, gx = 1.0f - fx;
        u = gy * (gx * cf(U, y0, x0) + fx * cf(U, y0, x1)) + fy * (gx * cf(U, y1, x0) + fx * cf(U, y1, x1));
        v = gy * (gx * cf(V, y0, x0) + fx * cf(V, y0, x1)) + fy * (gx * cf(V, y1, x0) + fx * cf(V, y1, x1));
    } else {
        u = cf(U, y, x);
        v = cf(V, y, x);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) a.out[(size_t)c * HW + p] = a.m[3 * c] * Yf + a.m[3 * c + 1] * u + a.m[3 * c + 2] * v;     // no clip (:237)
}

struct ResizeArgs {
    const float* rgb;        // [3][H][W]
    int W, H, Wo, Ho;
    float sx, sy;            // W / Wo, H / Ho in fp32 (torch: area_pixel_compute_scale / compute_scales_value without a scale factor)
    EotfDev e;
    float w[3];
    float* lum;              // [Ho][Wo]
    float* rgb_out;          // optional [3][Ho][Wo]: the clipped RGB the reference's unpack returns
};

// torch's cubic convolution coefficients, A = -0.75 (ATen/native/UpSample.h: cubic_convolution1 / 2, get_cubic_upsample_coefficients)
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
    const float A = -0.75f;
    auto cc1 = [&](float x) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; };
    auto cc2 = [&](float x) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; };
    c[0] = cc2(t + 1.0f);
    c[1] = cc1(t);
    c[2] = cc1(1.0f - t);
    c[3] = cc2(2.0f - t);
}

// MODE: FVVDP_RESIZE_NEAREST / BILINEAR / BICUBIC / AREA
template <int MODE>
__global__ __launch_bounds__(256) void resize_lum_kernel(const ResizeArgs a) {
    const int o = (int)(blockIdx.x * 256 + threadIdx.x);
    if (o >= a.Wo * a.Ho) return;
    const int oy = o / a.Wo, ox = o - oy * a.Wo;
    const size_t HW = (size_t)a.W * a.H;
    float v[3];
    if constexpr (MODE == FVVDP_RESIZE_NEAREST) {
        // nearest_neighbor_compute_source_index: min(floor(dst * scale), in - 1)   (mode='nearest', not 'nearest-exact')
        const int iy = min((int)floorf((float)oy * a.sy), a.H - 1), ix = min((int)floorf((float)ox * a.sx), a.W - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = a.rgb[c * HW + (size_t)iy * a.W + ix];
    } else if constexpr (MODE == FVVDP_RESIZE_BILINEAR) {
        // area_pixel_compute_source_index(cubic=false): scale * (dst + 0.5) - 0.5, negative -> 0; guard_index_and_lambda
        const float fy = fmaxf(a.sy * ((float)oy + 0.5f) - 0.5f, 0.0f), fx = fmaxf(a.sx * ((float)ox + 0.5f) - 0.5f, 0.0f);
        const int y0 = min((int)fy, a.H - 1), x0 = min((int)fx, a.W - 1);
        const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0), x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
        const float ly1 = fminf(fmaxf(fy - (float)y0, 0.0f), 1.0f), lx1 = fminf(fmaxf(fx - (float)x0, 0.0f), 1.0f);
        const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* P = a.rgb + c * HW;
            v[c] = ly0 * (lx0 * P[(size_t)y0 * a.W + x0] + lx1 * P[(size_t)y0 * a.W + x1]) +
                   ly1 * (lx0 * P[(size_t)y1 * a.W + x0] + lx1 * P[(size_t)y1 * a.W + x1]);
        }
    } else if constexpr (MODE == FVVDP_RESIZE_BICUBIC) {
        // area_pixel_compute_source_index(cubic=true): no clamp of the coordinate; every tap's index is clamped to the image
        const float ry = a.sy * ((float)oy + 0.5f) - 0.5f, rx = a.sx * ((float)ox + 0.5f) - 0.5f;
        const float fly = floorf(ry), flx = floorf(rx);
        const int iy = (int)fly, ix = (int)flx;
        float cy[4], cx[4];
        cubic_coeffs(ry - fly, cy);
        cubic_coeffs(rx - flx, cx);
        int xs[4], ys[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xs[k] = min(max(ix - 1 + k, 0), a.W - 1);
            ys[k] = min(max(iy - 1 + k, 0), a.H - 1);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* P = a.rgb + c * HW;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {         // x first along each of the four rows, then y (upsample_bicubic2d)
                const float* R = P + (size_t)ys[k] * a.W;
                const float row = ((R[xs[0]] * cx[0] + R[xs[1]] * cx[1]) + R[xs[2]] * cx[2]) + R[xs[3]] * cx[3];
                acc = (k == 0) ? row * cy[0] : acc + row * cy[k];
            }
            v[c] = acc;
        }
    } else {
        // 'area' = adaptive_avg_pool2d: start = floor(o * in / out), end = ceil((o + 1) * in / out)
        const int y0 = (int)floorf((float)(oy * a.H) / (float)a.Ho), y1 = (int)ceilf((float)((oy + 1) * a.H) / (float)a.Ho);
        const int x0 = (int)floorf((float)(ox * a.W) / (float)a.Wo), x1 = (int)ceilf((float)((ox + 1) * a.W) / (float)a.Wo);
        const float n = (float)((y1 - y0) * (x1 - x0));
        float acc[3] = {0.0f, 0.0f, 0.0f};
        for (int yy = y0; yy < y1; ++yy)               // the three planes side by side: three loads in flight per step instead of one
            for (int xx = x0; xx < x1; ++xx) {
                const size_t q = (size_t)yy * a.W + xx;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] += a.rgb[c * HW + q];
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = acc[c] / n;
    }
    const size_t OHW = (size_t)a.Wo * a.Ho;
    bool bad = false;
    float L = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float V = fminf(fmaxf(v[c], 0.0f), 1.0f);                 // RGB.clip(0, 1) (:244)
        if (a.rgb_out) a.rgb_out[c * OHW + o] = V;
        const float l = __fmul_rn(eotf_f32(V, a.e, bad), a.w[c]);       // display model + luminance weights (_prepare_frame, :355-363)
        L = (c == 0) ? l : __fadd_rn(L, l);
    }
    a.lum[o] = L;
}
