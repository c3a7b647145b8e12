// Per-Gaussian and per-(pixel,splat) arithmetic of the rasteriser, shared by the
// HIP kernels (device) and by tests/host_emul (host, g++) so that the exact same
// formulas are checked against the autograd oracle on the CPU before they run on
// the GPU.  Nothing here touches memory layout or parallelisation.
//
// Contract being implemented: the `diff_gaussian_rasterization` extension as called
// at /root/reference gaussian_splatting/gaussian_renderer/__init__.py:61-75,151-168
// (source of the extension itself is absent from the reference tree; constants are
// the published 3DGS/MonoGS ones, listed in oracle/torch_raster.py CONSTANTS).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MGS_HD __host__ __device__ __forceinline__
#else
#define MGS_HD inline
#endif

namespace mgs {

constexpr float kNearZ = 0.2f;
constexpr float kFovClamp = 1.3f;
constexpr float kLowpass = 0.3f;
constexpr float kLambdaFloor = 0.1f;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTStop = 1e-4f;
constexpr float kTouchT = 0.5f;
constexpr float kWEps = 1e-7f;
constexpr int kTile = 16;

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f,
                SH_C2_2 = 0.31539156525252005f, SH_C2_3 = -1.0925484305920792f,
                SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f,
                SH_C3_2 = -0.4570457994644658f, SH_C3_3 = 0.3731763325901154f,
                SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;

// Record flags
constexpr uint32_t kFlagVisible = 1u << 3;   // bits 0..2: colour channel clamped at 0

// 48-byte projected record, one per Gaussian (HBM layout: 3 x float4).
struct alignas(16) SplatRec {
  float x, y, depth, opacity;   // pixel-space mean, p_view.z, opacity
  float ca, cb, cc;             // conic (inverse 2-D covariance): A, B, C
  int32_t radius;               // ceil(3 sigma_max); 0 = culled
  float r, g, b;                // colour after SH + 0.5 and clamp
  uint32_t flags;
};

struct Camera {
  float V[16];     // viewmatrix (row-major torch layout; p_view = [p,1] @ V)
  float PM[16];    // full projection  ([p,1] @ PM)
  float Praw[16];  // projection_matrix ([p_view,1] @ Praw)
  float campos[3];
  int W, H;
  float tanfovx, tanfovy, focal_x, focal_y;
  float scale_modifier;
  int sh_degree;   // active degree
  int sh_coeffs;   // K = number of coefficient triples stored per Gaussian
  int grid_x, grid_y;
  int clamp_grad_upstream;   // backward only: 0 = exact derivative of the EWA clamp, 1 = t.x constant in z
};

MGS_HD float clampf(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); }

// 1/x: one v_rcp_f32 on the device (1 ulp), exact division on the host.
MGS_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}

// R(q) for q = (r,x,y,z), used as given (no normalisation), row-major R[3*i+j].
MGS_HD void quat_to_rot(const float q[4], float R[9]) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
  R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
  R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = R S S^T R^T as packed upper triangle (xx,xy,xz,yy,yz,zz).
MGS_HD void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], float c6[6]) {
  float R[9];
  quat_to_rot(q, R);
  float L[9];
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) L[3 * i + k] = R[3 * i + k] * (s[k] * mod);
  c6[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
  c6[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
  c6[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
  c6[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
  c6[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
  c6[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
}

// p_view = Rv p + tv with Rv[j][i] = V[4*i+j].
MGS_HD void to_view(const float* V, const float p[3], float pc[3]) {
  for (int j = 0; j < 3; j++)
    pc[j] = p[0] * V[0 + j] + p[1] * V[4 + j] + p[2] * V[8 + j] + V[12 + j];
}

struct Cov2D {
  float a, b, c;        // after the low-pass
  float M[6];           // J * Rv (2x3), row-major
  float txc, tyc, tz;   // clamped camera-space x,y and z used in J
  bool clamp_x, clamp_y;
};

MGS_HD void ewa_cov2d(const Camera& cam, const float pc[3], const float c6[6], Cov2D& o) {
  const float tz = pc[2];
  const float limx = kFovClamp * cam.tanfovx, limy = kFovClamp * cam.tanfovy;
  const float txtz = pc[0] / tz, tytz = pc[1] / tz;
  o.clamp_x = (txtz < -limx) || (txtz > limx);
  o.clamp_y = (tytz < -limy) || (tytz > limy);
  o.txc = clampf(txtz, -limx, limx) * tz;
  o.tyc = clampf(tytz, -limy, limy) * tz;
  o.tz = tz;
  const float J00 = cam.focal_x / tz, J02 = -(cam.focal_x * o.txc) / (tz * tz);
  const float J11 = cam.focal_y / tz, J12 = -(cam.focal_y * o.tyc) / (tz * tz);
  // Rv[k][i] = V[4*i+k];  M[r][i] = sum_k J[r][k] Rv[k][i]
  for (int i = 0; i < 3; i++) {
    o.M[i] = J00 * cam.V[4 * i + 0] + J02 * cam.V[4 * i + 2];
    o.M[3 + i] = J11 * cam.V[4 * i + 1] + J12 * cam.V[4 * i + 2];
  }
  const float S00 = c6[0], S01 = c6[1], S02 = c6[2], S11 = c6[3], S12 = c6[4], S22 = c6[5];
  // MS = M * Sigma (2x3)
  float MS[6];
  for (int r = 0; r < 2; r++) {
    const float m0 = o.M[3 * r], m1 = o.M[3 * r + 1], m2 = o.M[3 * r + 2];
    MS[3 * r + 0] = m0 * S00 + m1 * S01 + m2 * S02;
    MS[3 * r + 1] = m0 * S01 + m1 * S11 + m2 * S12;
    MS[3 * r + 2] = m0 * S02 + m1 * S12 + m2 * S22;
  }
  o.a = MS[0] * o.M[0] + MS[1] * o.M[1] + MS[2] * o.M[2] + kLowpass;
  o.b = MS[0] * o.M[3] + MS[1] * o.M[4] + MS[2] * o.M[5];
  o.c = MS[3] * o.M[3] + MS[4] * o.M[4] + MS[5] * o.M[5] + kLowpass;
}

// Degree-0..3 SH colour for unit direction d; sh is [K][3].
// The coefficients of a band are read for all three channels at once, before they are used (on the GPU: one
// batch of independent loads per band; a loop over the channels with the bands nested inside it waited for
// memory once per channel and band - three round trips at degree 0, twelve at degree 3).  The arithmetic per
// channel is the left-to-right sum it always was.
MGS_HD void sh_to_rgb(int deg, const float* sh, const float d[3], float rgb[3]) {
  const float x = d[0], y = d[1], z = d[2];
  float res[3];
  {
    const float s0 = sh[0], s1 = sh[1], s2 = sh[2];
    res[0] = SH_C0 * s0; res[1] = SH_C0 * s1; res[2] = SH_C0 * s2;
  }
  if (deg > 0) {
    float b1[9];
    for (int i = 0; i < 9; i++) b1[i] = sh[3 + i];
    for (int c = 0; c < 3; c++) res[c] = res[c] - SH_C1 * y * b1[c] + SH_C1 * z * b1[3 + c] - SH_C1 * x * b1[6 + c];
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      float b2[15];
      for (int i = 0; i < 15; i++) b2[i] = sh[12 + i];
      for (int c = 0; c < 3; c++)
        res[c] = res[c] + SH_C2_0 * xy * b2[c] + SH_C2_1 * yz * b2[3 + c] +
                 SH_C2_2 * (2.f * zz - xx - yy) * b2[6 + c] + SH_C2_3 * xz * b2[9 + c] +
                 SH_C2_4 * (xx - yy) * b2[12 + c];
      if (deg > 2) {
        float b3[21];
        for (int i = 0; i < 21; i++) b3[i] = sh[27 + i];
        for (int c = 0; c < 3; c++)
          res[c] = res[c] + SH_C3_0 * y * (3.f * xx - yy) * b3[c] + SH_C3_1 * xy * z * b3[3 + c] +
                   SH_C3_2 * y * (4.f * zz - xx - yy) * b3[6 + c] +
                   SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy) * b3[9 + c] +
                   SH_C3_4 * x * (4.f * zz - xx - yy) * b3[12 + c] +
                   SH_C3_5 * z * (xx - yy) * b3[15 + c] + SH_C3_6 * x * (xx - 3.f * yy) * b3[18 + c];
      }
    }
  }
  rgb[0] = res[0]; rgb[1] = res[1]; rgb[2] = res[2];
}

// Tile rectangle [min,max) of a splat of integer radius at pixel (x,y).
MGS_HD void tile_rect(float x, float y, int radius, int grid_x, int grid_y, int rmin[2],
                      int rmax[2]) {
  const float big = 1e8f;
  const float fx = clampf(x, -big, big), fy = clampf(y, -big, big);
  const float rr = (float)radius;
  const float inv = 1.0f / kTile;
  rmin[0] = (int)clampf(truncf((fx - rr) * inv), 0.f, (float)grid_x);
  rmin[1] = (int)clampf(truncf((fy - rr) * inv), 0.f, (float)grid_y);
  rmax[0] = (int)clampf(truncf((fx + rr + (kTile - 1)) * inv), 0.f, (float)grid_x);
  rmax[1] = (int)clampf(truncf((fy + rr + (kTile - 1)) * inv), 0.f, (float)grid_y);
}

// Forward of one Gaussian.  Returns false when culled (rec.radius = 0).
// `sh` points at this Gaussian's [K][3] coefficients (or null with `precol`).
MGS_HD bool project_gaussian(const Camera& cam, const float p[3], const float* scale,
                             const float* quat, const float* cov_pre, const float* sh,
                             const float* precol, float opacity, SplatRec& rec) {
  rec.x = rec.y = rec.depth = 0.f;
  rec.opacity = opacity;
  rec.ca = rec.cb = rec.cc = 0.f;
  rec.radius = 0;
  rec.r = rec.g = rec.b = 0.f;
  rec.flags = 0;
  float pc[3];
  to_view(cam.V, p, pc);
  if (!(pc[2] > kNearZ)) return false;
  float hom[4];
  for (int j = 0; j < 4; j++)
    hom[j] = p[0] * cam.PM[0 + j] + p[1] * cam.PM[4 + j] + p[2] * cam.PM[8 + j] + cam.PM[12 + j];
  const float pw = 1.0f / (hom[3] + kWEps);
  const float ndcx = hom[0] * pw, ndcy = hom[1] * pw;
  float c6[6];
  if (cov_pre) {
    for (int i = 0; i < 6; i++) c6[i] = cov_pre[i];
  } else {
    cov3d_from_scale_rot(scale, cam.scale_modifier, quat, c6);
  }
  Cov2D cv;
  ewa_cov2d(cam, pc, c6, cv);
  const float det = cv.a * cv.c - cv.b * cv.b;
  if (det == 0.0f) return false;
  const float det_inv = 1.0f / det;
  const float mid = 0.5f * (cv.a + cv.c);
  const float root = sqrtf(fmaxf(kLambdaFloor, mid * mid - det));
  const float lam = fmaxf(mid + root, mid - root);
  const float rad_f = fminf(ceilf(3.0f * sqrtf(lam)), 1e7f);
  const float px = ((ndcx + 1.0f) * cam.W - 1.0f) * 0.5f;
  const float py = ((ndcy + 1.0f) * cam.H - 1.0f) * 0.5f;
  const int radius = (int)rad_f;
  int rmin[2], rmax[2];
  tile_rect(px, py, radius, cam.grid_x, cam.grid_y, rmin, rmax);
  if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) return false;

  float rgb[3];
  uint32_t flags = kFlagVisible;
  if (precol) {
    rgb[0] = precol[0]; rgb[1] = precol[1]; rgb[2] = precol[2];
  } else {
    float d[3] = {p[0] - cam.campos[0], p[1] - cam.campos[1], p[2] - cam.campos[2]};
    const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] *= inv; d[1] *= inv; d[2] *= inv;
    sh_to_rgb(cam.sh_degree, sh, d, rgb);
    for (int c = 0; c < 3; c++) {
      rgb[c] += 0.5f;
      if (rgb[c] < 0.f) { flags |= (1u << c); rgb[c] = 0.f; }
    }
  }
  rec.x = px; rec.y = py; rec.depth = pc[2]; rec.opacity = opacity;
  rec.ca = cv.c * det_inv; rec.cb = -cv.b * det_inv; rec.cc = cv.a * det_inv;
  rec.radius = radius;
  rec.r = rgb[0]; rec.g = rgb[1]; rec.b = rgb[2];
  rec.flags = flags;
  return true;
}

// Exact tile culling: can any pixel centre of tile (tx,ty) reach alpha >= 1/255 ?
// q(d) = A dx^2 + 2 B dx dy + C dy^2 must be <= qmax = 2 ln(255 * opacity) somewhere
// in the tile's pixel box.  Conservative (never rejects a contributing pair): the
// slack covers fp32 rounding of the per-pixel evaluation in the blend kernels.
MGS_HD float splat_qmax(float opacity) { return 2.0f * logf(255.0f * opacity); }

// Pixel-centre box [x0,x1] x [y0,y1] (inclusive, already clipped to the image).
MGS_HD bool box_reachable(float x, float y, float A, float B, float C, float qmax, float x0,
                          float y0, float x1, float y1) {
  if (!(qmax >= 0.f)) return false;  // opacity < 1/255 (or NaN): contributes nowhere
  // d = mean - pixel  =>  dx in [x-x1, x-x0]
  const float lx = x - x1, hx = x - x0, ly = y - y1, hy = y - y0;
  if (lx <= 0.f && hx >= 0.f && ly <= 0.f && hy >= 0.f) return true;
  float qmin = 3.4e38f;
  {  // edges dx = lx, dx = hx
    const float invC = fast_rcp(C);
    float dy = clampf(-B * lx * invC, ly, hy);
    qmin = fminf(qmin, A * lx * lx + 2.f * B * lx * dy + C * dy * dy);
    dy = clampf(-B * hx * invC, ly, hy);
    qmin = fminf(qmin, A * hx * hx + 2.f * B * hx * dy + C * dy * dy);
  }
  {  // edges dy = ly, dy = hy
    const float invA = fast_rcp(A);
    float dx = clampf(-B * ly * invA, lx, hx);
    qmin = fminf(qmin, A * dx * dx + 2.f * B * dx * ly + C * ly * ly);
    dx = clampf(-B * hy * invA, lx, hx);
    qmin = fminf(qmin, A * dx * dx + 2.f * B * dx * hy + C * hy * hy);
  }
  const float dm2 = fmaxf(lx * lx, hx * hx) + fmaxf(ly * ly, hy * hy);
  const float slack = 0.01f * qmax + 0.02f + 4e-6f * (A + C) * dm2;
  return !(qmin > qmax + slack);   // NaN-safe: keep the pair
}

MGS_HD bool tile_reachable(float x, float y, float A, float B, float C, float qmax, int tx,
                           int ty, int W, int H) {
  const float x0 = (float)(tx * kTile), y0 = (float)(ty * kTile);
  return box_reachable(x, y, A, B, C, qmax, x0, y0, fminf(x0 + (kTile - 1), (float)(W - 1)),
                       fminf(y0 + (kTile - 1), (float)(H - 1)));
}

// ------------------------------------------------------------------------- //
// Backward of one Gaussian.
// Inputs: screen-space gradients summed over all pixels:
//   g_xy   dL/d(pixel-space mean)         (2)
//   g_con  dL/d(conic A, B, C)  (B = the single off-diagonal parameter) (3)
//   g_op   dL/d(opacity), g_rgb dL/d(rgb) (3), g_depth dL/d(depth)
// Outputs (all written, not accumulated): dmean[3], dmean2D_ndc[2], dscale[3],
// drot[4], dcov6[6] (when cov_pre), dsh[K*3] or dcol[3], dop, dtau[6] = [rho; theta].
// ------------------------------------------------------------------------- //
struct GaussGrad {
  float dmean[3];
  float dndc[2];
  float dscale[3];
  float drot[4];
  float dcov6[6];
  float dop;
  float dtau[6];
};

// Gradient of the SH colour w.r.t. the coefficients and (through the view direction) the mean.  Every
// coefficient gradient of the ACTIVE bands (k < 3 (deg+1)^2) is handed to `emit(k, value)` exactly once, with a
// literal k - a sink that stores straight to memory keeps no coefficient array alive (k_preprocess_bwd's mapping
// mode held a 48-entry array in scratch memory until round 4); bands above the active degree are the caller's.
template <typename Emit>
MGS_HD void sh_backward_emit(int deg, const float* sh, const float p[3], const float campos[3], uint32_t flags,
                             const float g_rgb_in[3], float dmean[3], Emit&& emit) {
  float g[3];
  for (int c = 0; c < 3; c++) g[c] = (flags & (1u << c)) ? 0.f : g_rgb_in[c];
  float dv[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
  const float n2 = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
  const float inv = 1.0f / sqrtf(n2);
  const float x = dv[0] * inv, y = dv[1] * inv, z = dv[2] * inv;
  float ddir[3] = {0.f, 0.f, 0.f};
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int c = 0; c < 3; c++) {
    const float gc = g[c];
    emit(c, SH_C0 * gc);
    if (deg > 0) {
      emit(3 + c, -SH_C1 * y * gc);
      emit(6 + c, SH_C1 * z * gc);
      emit(9 + c, -SH_C1 * x * gc);
      float dx = -SH_C1 * sh[9 + c], dy = -SH_C1 * sh[3 + c], dz = SH_C1 * sh[6 + c];
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        emit(12 + c, SH_C2_0 * xy * gc);
        emit(15 + c, SH_C2_1 * yz * gc);
        emit(18 + c, SH_C2_2 * (2.f * zz - xx - yy) * gc);
        emit(21 + c, SH_C2_3 * xz * gc);
        emit(24 + c, SH_C2_4 * (xx - yy) * gc);
        dx += SH_C2_0 * y * sh[12 + c] + SH_C2_2 * (-2.f * x) * sh[18 + c] +
              SH_C2_3 * z * sh[21 + c] + SH_C2_4 * 2.f * x * sh[24 + c];
        dy += SH_C2_0 * x * sh[12 + c] + SH_C2_1 * z * sh[15 + c] +
              SH_C2_2 * (-2.f * y) * sh[18 + c] + SH_C2_4 * (-2.f * y) * sh[24 + c];
        dz += SH_C2_1 * y * sh[15 + c] + SH_C2_2 * 4.f * z * sh[18 + c] + SH_C2_3 * x * sh[21 + c];
        if (deg > 2) {
          emit(27 + c, SH_C3_0 * y * (3.f * xx - yy) * gc);
          emit(30 + c, SH_C3_1 * xy * z * gc);
          emit(33 + c, SH_C3_2 * y * (4.f * zz - xx - yy) * gc);
          emit(36 + c, SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy) * gc);
          emit(39 + c, SH_C3_4 * x * (4.f * zz - xx - yy) * gc);
          emit(42 + c, SH_C3_5 * z * (xx - yy) * gc);
          emit(45 + c, SH_C3_6 * x * (xx - 3.f * yy) * gc);
          dx += SH_C3_0 * sh[27 + c] * 6.f * xy + SH_C3_1 * sh[30 + c] * yz +
                SH_C3_2 * sh[33 + c] * (-2.f * xy) + SH_C3_3 * sh[36 + c] * (-6.f * xz) +
                SH_C3_4 * sh[39 + c] * (4.f * zz - 3.f * xx - yy) +
                SH_C3_5 * sh[42 + c] * 2.f * xz + SH_C3_6 * sh[45 + c] * 3.f * (xx - yy);
          dy += SH_C3_0 * sh[27 + c] * 3.f * (xx - yy) + SH_C3_1 * sh[30 + c] * xz +
                SH_C3_2 * sh[33 + c] * (4.f * zz - xx - 3.f * yy) +
                SH_C3_3 * sh[36 + c] * (-6.f * yz) + SH_C3_4 * sh[39 + c] * (-2.f * xy) +
                SH_C3_5 * sh[42 + c] * (-2.f * yz) + SH_C3_6 * sh[45 + c] * (-6.f * xy);
          dz += SH_C3_1 * sh[30 + c] * xy + SH_C3_2 * sh[33 + c] * 8.f * yz +
                SH_C3_3 * sh[36 + c] * (6.f * zz - 3.f * xx - 3.f * yy) +
                SH_C3_4 * sh[39 + c] * 8.f * xz + SH_C3_5 * sh[42 + c] * (xx - yy);
        }
      }
      ddir[0] += dx * gc; ddir[1] += dy * gc; ddir[2] += dz * gc;
    }
  }
  if (deg > 0) {
    // dir = v/|v|:  d(dir)/dv = (I - dir dir^T)/|v|
    const float dot = ddir[0] * x + ddir[1] * y + ddir[2] * z;
    dmean[0] += (ddir[0] - x * dot) * inv;
    dmean[1] += (ddir[1] - y * dot) * inv;
    dmean[2] += (ddir[2] - z * dot) * inv;
  }
}

// array form: dsh[3 K], bands above the active degree zero
MGS_HD void sh_backward(int deg, int K, const float* sh, const float p[3], const float campos[3],
                        uint32_t flags, const float g_rgb_in[3], float* dsh, float dmean[3]) {
  for (int k = 3 * (deg + 1) * (deg + 1); k < K * 3; k++) dsh[k] = 0.f;
  sh_backward_emit(deg, sh, p, campos, flags, g_rgb_in, dmean, [&](int k, float v) { dsh[k] = v; });
}

MGS_HD void project_gaussian_backward(const Camera& cam, const float p[3], const float* scale,
                                      const float* quat, const float* cov_pre,
                                      const float g_xy[2], const float g_con[3], float g_op,
                                      float g_depth, GaussGrad& o) {
  for (int i = 0; i < 3; i++) { o.dmean[i] = 0.f; o.dscale[i] = 0.f; }
  for (int i = 0; i < 4; i++) o.drot[i] = 0.f;
  for (int i = 0; i < 6; i++) { o.dcov6[i] = 0.f; o.dtau[i] = 0.f; }
  o.dop = g_op;
  float pc[3];
  to_view(cam.V, p, pc);
  // ---- screen position path -------------------------------------------------
  float hom[4];
  for (int j = 0; j < 4; j++)
    hom[j] = pc[0] * cam.Praw[0 + j] + pc[1] * cam.Praw[4 + j] + pc[2] * cam.Praw[8 + j] +
             cam.Praw[12 + j];
  const float mw = 1.0f / (hom[3] + kWEps);
  const float gnx = g_xy[0] * 0.5f * cam.W, gny = g_xy[1] * 0.5f * cam.H;  // dL/d ndc
  o.dndc[0] = gnx; o.dndc[1] = gny;
  const float gh0 = gnx * mw, gh1 = gny * mw;
  const float gh3 = -(gnx * hom[0] + gny * hom[1]) * mw * mw;
  float gpc[3];
  for (int i = 0; i < 3; i++)
    gpc[i] = cam.Praw[4 * i + 0] * gh0 + cam.Praw[4 * i + 1] * gh1 + cam.Praw[4 * i + 3] * gh3;
  // ---- depth path -------------------------------------------------------------
  gpc[2] += g_depth;
  // ---- covariance path ----------------------------------------------------------
  float c6[6];
  if (cov_pre) {
    for (int i = 0; i < 6; i++) c6[i] = cov_pre[i];
  } else {
    cov3d_from_scale_rot(scale, cam.scale_modifier, quat, c6);
  }
  Cov2D cv;
  ewa_cov2d(cam, pc, c6, cv);
  const float a = cv.a, b = cv.b, c = cv.c;
  const float det = a * c - b * b;
  const float d2 = 1.0f / (det * det + 1e-30f);
  const float gA = g_con[0], gB = g_con[1], gC = g_con[2];
  const float ga = (-c * c * gA + b * c * gB - b * b * gC) * d2;
  const float gb = (2.f * b * c * gA - (det + 2.f * b * b) * gB + 2.f * a * b * gC) * d2;
  const float gc = (-b * b * gA + a * b * gB - a * a * gC) * d2;
  // Gc = [[ga, gb/2],[gb/2, gc]];  GM = Gc * M (2x3)
  const float hb = 0.5f * gb;
  float GM[6];
  for (int i = 0; i < 3; i++) {
    GM[i] = ga * cv.M[i] + hb * cv.M[3 + i];
    GM[3 + i] = hb * cv.M[i] + gc * cv.M[3 + i];
  }
  // dL/dSigma (full symmetric, entries independent) = M^T Gc M
  float GS[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) GS[3 * i + j] = cv.M[i] * GM[j] + cv.M[3 + i] * GM[3 + j];
  // dL/dM = 2 Gc M Sigma
  const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
  float dM[6];
  for (int r = 0; r < 2; r++)
    for (int j = 0; j < 3; j++)
      dM[3 * r + j] = 2.f * (GM[3 * r] * S[j] + GM[3 * r + 1] * S[3 + j] + GM[3 * r + 2] * S[6 + j]);
  // M = J Rv:  dL/dJ[r][k] = sum_i dM[r][i] Rv[k][i] ; Rv[k][i] = V[4i+k]
  float dJ[6];
  for (int r = 0; r < 2; r++)
    for (int k = 0; k < 3; k++)
      dJ[3 * r + k] = dM[3 * r] * cam.V[k] + dM[3 * r + 1] * cam.V[4 + k] + dM[3 * r + 2] * cam.V[8 + k];
  const float tz = cv.tz, itz = 1.0f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
  const float fx = cam.focal_x, fy = cam.focal_y;
  const float g_txc = dJ[2] * (-fx * itz2);
  const float g_tyc = dJ[5] * (-fy * itz2);
  float g_tz = dJ[0] * (-fx * itz2) + dJ[4] * (-fy * itz2) + dJ[2] * (2.f * fx * cv.txc * itz3) +
               dJ[5] * (2.f * fy * cv.tyc * itz3);
  // t.x = clamp(x/z) z: unclamped it equals x (no z dependence); clamped it is c z, whose exact
  // derivative feeds z.  The public CUDA lineage is believed to multiply the x-gradient by 0 when
  // clamped and to keep t.x constant in z (mode 1); the forward is the same.
  if (cv.clamp_x) { if (!cam.clamp_grad_upstream) g_tz += g_txc * cv.txc * itz; } else gpc[0] += g_txc;
  if (cv.clamp_y) { if (!cam.clamp_grad_upstream) g_tz += g_tyc * cv.tyc * itz; } else gpc[1] += g_tyc;
  gpc[2] += g_tz;
  // ---- mean and pose ---------------------------------------------------------------
  // p_c = Rv p + tv:  dL/dp = Rv^T gpc ;  Rv^T[i][k] = Rv[k][i] = V[4i+k]
  for (int i = 0; i < 3; i++)
    o.dmean[i] = cam.V[4 * i] * gpc[0] + cam.V[4 * i + 1] * gpc[1] + cam.V[4 * i + 2] * gpc[2];
  o.dtau[0] = gpc[0]; o.dtau[1] = gpc[1]; o.dtau[2] = gpc[2];
  o.dtau[3] = pc[1] * gpc[2] - pc[2] * gpc[1];
  o.dtau[4] = pc[2] * gpc[0] - pc[0] * gpc[2];
  o.dtau[5] = pc[0] * gpc[1] - pc[1] * gpc[0];
  // rotation of the frame inside the EWA Jacobian product: W(theta) = Exp(theta) Rv
  // dL/dRv = J^T dM (3x3); A = dL/dRv * Rv^T
  {
    const float J00 = fx * itz, J02 = -(fx * cv.txc) * itz2;
    const float J11 = fy * itz, J12 = -(fy * cv.tyc) * itz2;
    float dR[9];
    for (int i = 0; i < 3; i++) {
      dR[0 + i] = J00 * dM[i];
      dR[3 + i] = J11 * dM[3 + i];
      dR[6 + i] = J02 * dM[i] + J12 * dM[3 + i];
    }
    float A[9];
    for (int k = 0; k < 3; k++)
      for (int l = 0; l < 3; l++)
        A[3 * k + l] = dR[3 * k] * cam.V[l] + dR[3 * k + 1] * cam.V[4 + l] + dR[3 * k + 2] * cam.V[8 + l];
    o.dtau[3] += A[7] - A[5];
    o.dtau[4] += A[2] - A[6];
    o.dtau[5] += A[3] - A[1];
  }
  // ---- 3-D covariance parameters ------------------------------------------------------
  if (cov_pre) {
    o.dcov6[0] = GS[0]; o.dcov6[1] = GS[1] + GS[3]; o.dcov6[2] = GS[2] + GS[6];
    o.dcov6[3] = GS[4]; o.dcov6[4] = GS[5] + GS[7]; o.dcov6[5] = GS[8];
  } else {
    float R[9];
    quat_to_rot(quat, R);
    const float mod = cam.scale_modifier;
    float L[9], dL[9];
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 3; k++) L[3 * i + k] = R[3 * i + k] * (scale[k] * mod);
    // Sigma = L L^T  =>  dL/dL = (GS + GS^T) L
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 3; k++) {
        float acc = 0.f;
        for (int j = 0; j < 3; j++) acc += (GS[3 * i + j] + GS[3 * j + i]) * L[3 * j + k];
        dL[3 * i + k] = acc;
      }
    float G[9];
    for (int k = 0; k < 3; k++) {
      o.dscale[k] = mod * (R[k] * dL[k] + R[3 + k] * dL[3 + k] + R[6 + k] * dL[6 + k]);
      const float sk = scale[k] * mod;
      G[k] = dL[k] * sk; G[3 + k] = dL[3 + k] * sk; G[6 + k] = dL[6 + k] * sk;
    }
    const float r = quat[0], x = quat[1], y = quat[2], z = quat[3];
    o.drot[0] = 2.f * (-z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7]);
    o.drot[1] = 2.f * (y * G[1] + z * G[2] + y * G[3] - 2.f * x * G[4] - r * G[5] + z * G[6] + r * G[7] - 2.f * x * G[8]);
    o.drot[2] = 2.f * (-2.f * y * G[0] + x * G[1] + r * G[2] + x * G[3] + z * G[5] - r * G[6] + z * G[7] - 2.f * y * G[8]);
    o.drot[3] = 2.f * (-2.f * z * G[0] - r * G[1] + x * G[2] + r * G[3] - 2.f * z * G[4] + y * G[5] + x * G[6] + y * G[7]);
  }
}

// ------------------------------------------------------------------------- //
// per (pixel, splat) evaluation used by the blend kernels
// ------------------------------------------------------------------------- //
// power = -0.5 (A dx^2 + C dy^2) - B dx dy with d = mean - pixel.
MGS_HD float splat_power(float dx, float dy, float A, float B, float C) {
  return -0.5f * (A * dx * dx + C * dy * dy) - B * dx * dy;
}

// What the blend kernels need of a record.
struct SplatLite {
  float x, y, A, B, C, o, r, g, b, depth;
};

MGS_HD SplatLite lite_of(const SplatRec& q) {
  SplatLite s;
  s.x = q.x; s.y = q.y; s.A = q.ca; s.B = q.cb; s.C = q.cc; s.o = q.opacity;
  s.r = q.r; s.g = q.g; s.b = q.b; s.depth = q.depth;
  return s;
}

// One front-to-back step for one pixel.  Returns 0 = splat skipped, 1 = blended,
// 2 = pixel saturated (the caller stops; this splat is NOT blended).
MGS_HD int blend_forward_step(float px, float py, const SplatLite& s, float& T, float C[3],
                              float& D, bool& touched) {
  touched = false;
  const float dx = s.x - px, dy = s.y - py;
  const float power = splat_power(dx, dy, s.A, s.B, s.C);
  if (power > 0.f) return 0;
  const float alpha = fminf(kAlphaMax, s.o * expf(power));
  if (alpha < kAlphaMin) return 0;
  const float test_T = T * (1.f - alpha);
  if (test_T < kTStop) return 2;
  const float w = alpha * T;
  C[0] += s.r * w; C[1] += s.g * w; C[2] += s.b * w;
  D += s.depth * w;
  touched = test_T > kTouchT;
  T = test_T;
  return 1;
}

// Per-pixel state of the back-to-front replay.
struct PixBwd {
  float T;            // transmittance in front of the splat being visited
  float T_final;
  float acc[3];       // colour accumulated behind the current splat
  float acc_d;
  float last_alpha;
  float last_c[3];
  float last_d;
  float gpix[3];      // dL/d(pixel colour)
  float gdepth;       // dL/d(pixel depth)
  float bg_dot;       // bg . gpix
};

// The 10 screen-space gradients one (pixel, splat) pair contributes.
struct SplatGrad {
  float gx, gy;          // dL/d(pixel-space mean)
  float gA, gB, gC;      // dL/d(conic), B = the single off-diagonal parameter
  float gop;             // dL/d(opacity)
  float gr, gg, gb;      // dL/d(rgb)
  float gdepth;          // dL/d(depth)
};

MGS_HD void pixbwd_init(PixBwd& st, float T_final, const float gpix[3], float gdepth,
                        const float bg[3]) {
  st.T = T_final; st.T_final = T_final;
  st.acc[0] = st.acc[1] = st.acc[2] = 0.f; st.acc_d = 0.f;
  st.last_alpha = 0.f; st.last_c[0] = st.last_c[1] = st.last_c[2] = 0.f; st.last_d = 0.f;
  st.gpix[0] = gpix[0]; st.gpix[1] = gpix[1]; st.gpix[2] = gpix[2];
  st.gdepth = gdepth;
  st.bg_dot = bg[0] * gpix[0] + bg[1] * gpix[1] + bg[2] * gpix[2];
}

// One back-to-front step.  Returns false when the splat did not contribute to this
// pixel in the forward pass (g is then left untouched).
MGS_HD bool blend_backward_step(float px, float py, const SplatLite& s, PixBwd& st,
                                SplatGrad& g) {
  const float dx = s.x - px, dy = s.y - py;
  const float power = splat_power(dx, dy, s.A, s.B, s.C);
  if (power > 0.f) return false;
  const float G = expf(power);
  const float alpha = fminf(kAlphaMax, s.o * G);
  if (alpha < kAlphaMin) return false;
  const float om = 1.f - alpha;
  st.T = st.T / om;
  const float w = alpha * st.T;
  float dL_dalpha = 0.f;
  const float col[3] = {s.r, s.g, s.b};
  for (int ch = 0; ch < 3; ch++) {
    st.acc[ch] = st.last_alpha * st.last_c[ch] + (1.f - st.last_alpha) * st.acc[ch];
    st.last_c[ch] = col[ch];
    dL_dalpha += (col[ch] - st.acc[ch]) * st.gpix[ch];
  }
  g.gr = w * st.gpix[0]; g.gg = w * st.gpix[1]; g.gb = w * st.gpix[2];
  st.acc_d = st.last_alpha * st.last_d + (1.f - st.last_alpha) * st.acc_d;
  st.last_d = s.depth;
  dL_dalpha += (s.depth - st.acc_d) * st.gdepth;
  g.gdepth = w * st.gdepth;
  dL_dalpha *= st.T;
  st.last_alpha = alpha;
  dL_dalpha += (-st.T_final / om) * st.bg_dot;
  // alpha = min(0.99, o*G) is straight-through in backward (see oracle header)
  const float dL_dG = s.o * dL_dalpha;
  const float gdx = G * dx, gdy = G * dy;
  g.gx = dL_dG * (-gdx * s.A - gdy * s.B);
  g.gy = dL_dG * (-gdy * s.C - gdx * s.B);
  g.gA = -0.5f * gdx * dx * dL_dG;
  g.gB = -gdx * dy * dL_dG;
  g.gC = -0.5f * gdy * dy * dL_dG;
  g.gop = G * dL_dalpha;
  return true;
}

}  // namespace mgs
