/* oracle_sim.c -- TEST INFRASTRUCTURE (see oracle/__init__.py): plain-C restatements of the simulator pieces
 * whose numpy restatement is too slow for rollout-length checks and for bench.py's cpu_baseline leg.
 *
 *   oracle_raster_zbuf   == oracle/raster.py::raster_zbuf (PyTorch3D MeshRasterizer as called at
 *                           macarons/utility/macarons_utils.py:905-937, 2743-2786; PARITY UNPINNED against the
 *                           library, validated bit-for-bit against raster.py on small cases)
 *   oracle_unproject     == oracle/camera.py::unproject (Camera.project_depth_in_3D, mu:2788-2809)
 *   oracle_coverage_count== oracle/planner.py::coverage's inner loop (calculate_coverage_percentage,
 *                           next_best_path/utility/long_term_utils.py:437-468: G x M distances, row minimum, threshold)
 *
 * fp32 throughout with the operation order of the numpy files (compile with -ffp-contract=off, no -ffast-math).
 * The raster only visits the pixels of a conservative screen box of each face (full image when the face crosses the
 * clip plane); the per-pixel ray / triangle algebra is raster.py's. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void to_view(const float* p, const float* R, const float* T, float* o) {
    for (int j = 0; j < 3; ++j) o[j] = ((p[0] * R[0 + j] + p[1] * R[3 + j]) + p[2] * R[6 + j]) + T[j];
}

/* rows [rlo, rhi) of the image only (a band owns its pixels: bands can run on different threads, the face order per pixel
 * -- hence the result -- is that of the whole-image loop) */
static void raster_rows(const float* verts, const int* faces, int n_faces, const float* R, const float* T, int H, int W,
                        float tan_half_fov, float z_clip, float eps, float* zbuf, float* ub, float* vb, int* fb, int rlo, int rhi) {
    const int s = H < W ? H : W;
    float* dxs = (float*)malloc(sizeof(float) * (size_t)W);
    float* dys = (float*)malloc(sizeof(float) * (size_t)H);
    for (int c = 0; c < W; ++c) dxs[c] = ((float)W - (2.f * (float)c + 1.f)) / (float)s * tan_half_fov;
    for (int r = 0; r < H; ++r) dys[r] = ((float)H - (2.f * (float)r + 1.f)) / (float)s * tan_half_fov;
    for (size_t i = (size_t)rlo * W; i < (size_t)rhi * W; ++i) { zbuf[i] = 3.0e38f; if (fb) fb[i] = -1; }
    for (int fi = 0; fi < n_faces; ++fi) {
        float v[3][3];
        for (int k = 0; k < 3; ++k) to_view(verts + 3 * (size_t)faces[3 * (size_t)fi + k], R, T, v[k]);
        if (v[0][2] <= z_clip && v[1][2] <= z_clip && v[2][2] <= z_clip) continue;
        int c0 = 0, c1 = W - 1, r0 = 0, r1 = H - 1;
        if (v[0][2] > z_clip && v[1][2] > z_clip && v[2][2] > z_clip) {
            float cmin = 1e30f, cmax = -1e30f, rmin = 1e30f, rmax = -1e30f;
            for (int k = 0; k < 3; ++k) {
                const float nx = v[k][0] / (v[k][2] * tan_half_fov), ny = v[k][1] / (v[k][2] * tan_half_fov);
                const float col = ((float)W - (float)s * nx - 1.f) * 0.5f, row = ((float)H - (float)s * ny - 1.f) * 0.5f;
                cmin = fminf(cmin, col); cmax = fmaxf(cmax, col); rmin = fminf(rmin, row); rmax = fmaxf(rmax, row);
            }
            cmin = fmaxf(cmin, -1e6f); rmin = fmaxf(rmin, -1e6f); cmax = fminf(cmax, 1e6f); rmax = fminf(rmax, 1e6f);
            c0 = (int)floorf(cmin) - 2; c1 = (int)ceilf(cmax) + 2; r0 = (int)floorf(rmin) - 2; r1 = (int)ceilf(rmax) + 2;
            if (c0 < 0) c0 = 0;
            if (r0 < 0) r0 = 0;
            if (c1 > W - 1) c1 = W - 1;
            if (r1 > H - 1) r1 = H - 1;
            if (c0 > c1 || r0 > r1) continue;
        }
        if (r0 < rlo) r0 = rlo;
        if (r1 > rhi - 1) r1 = rhi - 1;
        if (r0 > r1) continue;
        float e1[3], e2[3], q[3];
        for (int c = 0; c < 3; ++c) { e1[c] = v[1][c] - v[0][c]; e2[c] = v[2][c] - v[0][c]; }
        const float* v0 = v[0];
        q[0] = e1[1] * v0[2] - e1[2] * v0[1];
        q[1] = e1[2] * v0[0] - e1[0] * v0[2];
        q[2] = e1[0] * v0[1] - e1[1] * v0[0];
        const float tnum = (e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2];
        /* plane forms (oracle/raster.py::plane_forms): det and the u numerator are linear in the pixel's ray (dx, dy, 1) */
        const float a[3] = {e1[2] * e2[1] - e1[1] * e2[2], e1[0] * e2[2] - e1[2] * e2[0], e1[1] * e2[0] - e1[0] * e2[1]};
        const float un[3] = {v0[1] * e2[2] - v0[2] * e2[1], v0[2] * e2[0] - v0[0] * e2[2], v0[0] * e2[1] - v0[1] * e2[0]};
        for (int r = r0; r <= r1; ++r) {
            const float dy = dys[r];
            float* zrow = zbuf + (size_t)r * W;
            for (int c = c0; c <= c1; ++c) {
                const float dx = dxs[c];
                const float det = (a[0] * dx + a[1] * dy) + a[2];
                if (!(fabsf(det) >= 1e-12f)) continue;
                const float inv = 1.f / det;
                const float u = ((un[0] * dx + un[1] * dy) + un[2]) * inv;
                const float vv = ((dx * q[0] + dy * q[1]) + q[2]) * inv;
                const float z = tnum * inv;
                if (u >= -eps && vv >= -eps && u + vv <= 1.f + eps && z > z_clip && z < zrow[c]) {
                    zrow[c] = z;
                    if (fb) { const size_t i = (size_t)r * W + c; fb[i] = fi; ub[i] = u; vb[i] = vv; }
                }
            }
        }
    }
    free(dxs); free(dys);
}

static void raster_core(const float* verts, const int* faces, int n_faces, const float* R, const float* T, int H, int W,
                        float tan_half_fov, float z_clip, float eps, float* zbuf, float* ub, float* vb, int* fb) {
    raster_rows(verts, faces, n_faces, R, T, H, W, tan_half_fov, z_clip, eps, zbuf, ub, vb, fb, 0, H);
}

/* n_frames z-buffers [n_frames][H][W], cameras Rs [n_frames][9] / Ts [n_frames][3]; (frame, band of `band_rows` rows) tasks
 * over OpenMP threads when compiled with -fopenmp (bench.py's cpu_baseline); identical to oracle_raster_zbuf per frame. */
void oracle_raster_zbuf_frames(const float* verts, int n_verts, const int* faces, int n_faces, int n_frames, const float* Rs,
                               const float* Ts, int H, int W, float tan_half_fov, float z_clip, float eps, int band_rows,
                               float* zbufs) {
    (void)n_verts;
    if (band_rows < 1) band_rows = H;
    const int bands = (H + band_rows - 1) / band_rows;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int t = 0; t < n_frames * bands; ++t) {
        const int f = t / bands, b = t % bands;
        const int rlo = b * band_rows, rhi = rlo + band_rows < H ? rlo + band_rows : H;
        float* z = zbufs + (size_t)f * H * W;
        raster_rows(verts, faces, n_faces, Rs + 9 * f, Ts + 3 * f, H, W, tan_half_fov, z_clip, eps, z, NULL, NULL, NULL, rlo, rhi);
        for (size_t i = (size_t)rlo * W; i < (size_t)rhi * W; ++i)
            if (!(z[i] < 1.0e38f)) z[i] = -1.f;
    }
}

void oracle_raster_zbuf(const float* verts, int n_verts, const int* faces, int n_faces, const float* R, const float* T,
                        int H, int W, float tan_half_fov, float z_clip, float eps, float* zbuf) {
    (void)n_verts;
    raster_core(verts, faces, n_faces, R, T, H, W, tan_half_fov, z_clip, eps, zbuf, NULL, NULL, NULL);
    for (size_t i = 0; i < (size_t)H * W; ++i)
        if (!(zbuf[i] < 1.0e38f)) zbuf[i] = -1.f;
}

/* == oracle/raster.py::raster_rgbz: depth + ambient x interpolated vertex colours, white background, adjust_contrast. */
void oracle_raster_rgbz(const float* verts, int n_verts, const int* faces, int n_faces, const float* colors, const float* R,
                        const float* T, int H, int W, float tan_half_fov, float z_clip, float eps, float ambient, float contrast,
                        float* zbuf, float* rgb) {
    (void)n_verts;
    const size_t n = (size_t)H * W;
    float* ub = (float*)malloc(sizeof(float) * n);
    float* vb = (float*)malloc(sizeof(float) * n);
    int* fb = (int*)malloc(sizeof(int) * n);
    raster_core(verts, faces, n_faces, R, T, H, W, tan_half_fov, z_clip, eps, zbuf, ub, vb, fb);
    double gsum = 0.0;
    for (size_t i = 0; i < n; ++i) {
        float c[3] = {1.f, 1.f, 1.f};
        if (fb[i] >= 0) {
            const int* f = faces + 3 * (size_t)fb[i];
            const float w0 = (1.f - ub[i]) - vb[i];
            for (int k = 0; k < 3; ++k)
                c[k] = ambient * ((w0 * colors[3 * (size_t)f[0] + k] + ub[i] * colors[3 * (size_t)f[1] + k]) +
                                  vb[i] * colors[3 * (size_t)f[2] + k]);
        } else {
            zbuf[i] = -1.f;
        }
        rgb[3 * i] = c[0]; rgb[3 * i + 1] = c[1]; rgb[3 * i + 2] = c[2];
        gsum += (double)((0.299f * c[0] + 0.587f * c[1]) + 0.114f * c[2]);
    }
    if (contrast != 1.f) {
        const float mean = (float)(gsum / (double)n);
        for (size_t i = 0; i < 3 * n; ++i) {
            const float v = contrast * rgb[i] + (1.f - contrast) * mean;
            rgb[i] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        }
    }
    free(ub); free(vb); free(fb);
}

/* All pixels of one depth frame -> world points [H*W,3] (oracle/camera.py::unproject). */
void oracle_unproject(const float* depth, int H, int W, float tan_half_fov, const float* R, const float* T, float* out) {
    const int s = H < W ? H : W;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const float ndc_x = (float)((double)W / s) - ((float)c / (float)(s - 1)) * 2.f;
            const float ndc_y = (float)((double)H / s) - ((float)r / (float)(s - 1)) * 2.f;
            const float z = depth[(size_t)r * W + c];
            const float xv = (ndc_x * z) * tan_half_fov, yv = (ndc_y * z) * tan_half_fov;
            const float dx = xv - T[0], dy = yv - T[1], dz = z - T[2];
            float* o = out + 3 * ((size_t)r * W + c);
            for (int j = 0; j < 3; ++j) o[j] = (dx * R[3 * j] + dy * R[3 * j + 1]) + dz * R[3 * j + 2];
        }
}

/* Number of gt points whose nearest pc point is closer than threshold (brute force G x M, like torch.cdist + min). */
long long oracle_coverage_count(const float* gt, long long G, const float* pc, long long M, float threshold) {
    long long cnt = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : cnt) schedule(static)
#endif
    for (long long i = 0; i < G; ++i) {
        const float gx = gt[3 * i], gy = gt[3 * i + 1], gz = gt[3 * i + 2];
        float best = 3.0e38f;
        for (long long j = 0; j < M; ++j) {
            const float ex = gx - pc[3 * j], ey = gy - pc[3 * j + 1], ez = gz - pc[3 * j + 2];
            const float d2 = (ex * ex + ey * ey) + ez * ez;
            if (d2 < best) best = d2;
        }
        if (M > 0 && sqrtf(best) < threshold) ++cnt;
    }
    return cnt;
}

/* == oracle/maps.py::accumulate_step_maps (utils.py:166-223 + the slab split / height band of nbp_planning.py:114-127,178-183):
 * out [6][S][S] counts.  bounds = y_bins[:-1] (nb entries; slab = #bounds < y, minus 1), lo_t / hi_t = the height band's
 * open interval, scale = (float)(S / (hi - lo)).  Threads own private count planes that are summed at the end (counts are
 * integers < 2^24: any summation order gives the same floats). */
void oracle_accumulate_step_maps(const float* pts, long long n, float cx, float cz, const float* bounds, int nb, int n_pieces,
                                 float lo_t, float hi_t, int S, float lo, float scale, int max_threads, float* out) {
    const size_t plane = (size_t)S * S;
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
    if (max_threads > 0 && nt > max_threads) nt = max_threads;
#else
    (void)max_threads;
#endif
    unsigned** priv = (unsigned**)calloc((size_t)nt, sizeof(unsigned*));
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
    {
        int me = 0, team = 1;
#ifdef _OPENMP
        me = omp_get_thread_num(); team = omp_get_num_threads();
#endif
        unsigned* mine = priv[me] = (unsigned*)calloc(6 * plane, sizeof(unsigned));
        const long long i0 = n * me / team, i1 = n * (me + 1) / team;
        for (long long i = i0; i < i1; ++i) {
            const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
            const float fr = rintf((-(z - cz) - lo) * scale), fc = rintf((-(x - cx) - lo) * scale);
            if (!(fr >= 0.f && fr < (float)S && fc >= 0.f && fc < (float)S)) continue;
            const size_t cell = (size_t)fr * S + (size_t)fc;
            int bin = -1;
            for (int k = 0; k < nb; ++k) bin += bounds[k] < y;
            mine[(size_t)((bin < 0 || bin >= n_pieces) ? 4 : bin) * plane + cell] += 1u;
            if (y < hi_t && y > lo_t) mine[5 * plane + cell] += 1u;
        }
#ifdef _OPENMP
#pragma omp barrier
#endif
        const size_t c0 = 6 * plane * (size_t)me / (size_t)team, c1 = 6 * plane * (size_t)(me + 1) / (size_t)team;
        for (size_t c = c0; c < c1; ++c) {
            unsigned acc = 0;
            for (int t = 0; t < team; ++t) acc += priv[t][c];
            out[c] = (float)acc;
        }
#ifdef _OPENMP
#pragma omp barrier
#endif
        free(mine);
    }
    free(priv);
}
