// Host emulation of the rasteriser pipeline (TEST INFRASTRUCTURE ONLY).
//
// Runs the per-Gaussian / per-pixel arithmetic of monogs_amd/csrc/raster_math.h
// (the very header the HIP kernels include) on the CPU, tile by tile, so that the
// hand-derived backward and the exact tile culling can be checked against the
// autograd oracle (oracle/torch_raster.py) in this GPU-less container, and so that
// bench.py has a multi-threaded CPU baseline ("port") at full problem size.
//
// It restates the contract of `diff_gaussian_rasterization` as called at
// /root/reference gaussian_splatting/gaussian_renderer/__init__.py:151-168.
// PARITY UNPINNED (no reference fixture exists for the rasteriser; see the oracle
// header).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library; the product never does.
//
// Build: g++ -O2 -fopenmp -shared -fPIC oracle/host_emul.cpp -o oracle/libhost_emul.so
#include <algorithm>
#include <cstdint>
#include <omp.h>
#include <cstring>
#include <vector>

#include "../monogs_amd/csrc/raster_math.h"

using namespace mgs;

namespace {

struct Ctx {
  Camera cam;
  int N = 0;
  std::vector<SplatRec> rec;
  std::vector<int32_t> tile_offset;           // [T+1]
  std::vector<uint64_t> keys;                 // sorted per tile: depth_bits<<32 | id
  std::vector<float> final_T;
  std::vector<int32_t> n_contrib;
};

Camera make_camera(const float* V, const float* PM, const float* Praw, const float* campos,
                   int W, int H, float tanfovx, float tanfovy, float scale_modifier, int sh_degree,
                   int sh_coeffs) {
  Camera c;
  memcpy(c.V, V, 64); memcpy(c.PM, PM, 64); memcpy(c.Praw, Praw, 64);
  c.campos[0] = campos[0]; c.campos[1] = campos[1]; c.campos[2] = campos[2];
  c.W = W; c.H = H; c.tanfovx = tanfovx; c.tanfovy = tanfovy;
  c.focal_x = W / (2.0f * tanfovx); c.focal_y = H / (2.0f * tanfovy);
  c.scale_modifier = scale_modifier; c.sh_degree = sh_degree; c.sh_coeffs = sh_coeffs;
  c.grid_x = (W + kTile - 1) / kTile; c.grid_y = (H + kTile - 1) / kTile;
  c.clamp_grad_upstream = 1;
  return c;
}

uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

}  // namespace

extern "C" {

void* emul_create() { return new Ctx(); }
void emul_destroy(void* h) { delete (Ctx*)h; }

// exact_cull = 0 reproduces the reference's bounding-square binning; 1 applies the
// product's exact tile culling (results must be identical, pair count smaller).
int64_t emul_forward(void* h, int N, int W, int H, float tanfovx, float tanfovy,
                     float scale_modifier, int sh_degree, int sh_coeffs, const float* V,
                     const float* PM, const float* Praw, const float* campos, const float* bg,
                     const float* means3D, const float* scales, const float* rotations,
                     const float* cov_pre, const float* opacities, const float* shs,
                     const float* precol, int exact_cull, float* out_color, float* out_depth,
                     float* out_opacity, int32_t* radii, int32_t* n_touched) {
  Ctx& c = *(Ctx*)h;
  c.cam = make_camera(V, PM, Praw, campos, W, H, tanfovx, tanfovy, scale_modifier, sh_degree,
                      sh_coeffs);
  c.N = N;
  c.rec.resize(N);
  const Camera& cam = c.cam;
  const int T = cam.grid_x * cam.grid_y;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++) {
    project_gaussian(cam, means3D + 3 * i, scales ? scales + 3 * i : nullptr,
                     rotations ? rotations + 4 * i : nullptr, cov_pre ? cov_pre + 6 * i : nullptr,
                     shs ? shs + 3 * (size_t)sh_coeffs * i : nullptr,
                     precol ? precol + 3 * i : nullptr, opacities[i], c.rec[i]);
    radii[i] = c.rec[i].radius;
    n_touched[i] = 0;
  }
  // binning
  std::vector<std::vector<uint64_t>> lists(T);
  for (int i = 0; i < N; i++) {
    const SplatRec& r = c.rec[i];
    if (r.radius <= 0) continue;
    int rmin[2], rmax[2];
    tile_rect(r.x, r.y, r.radius, cam.grid_x, cam.grid_y, rmin, rmax);
    const float qmax = splat_qmax(r.opacity);
    for (int ty = rmin[1]; ty < rmax[1]; ty++)
      for (int tx = rmin[0]; tx < rmax[0]; tx++) {
        if (exact_cull && !tile_reachable(r.x, r.y, r.ca, r.cb, r.cc, qmax, tx, ty, W, H)) continue;
        lists[ty * cam.grid_x + tx].push_back(((uint64_t)fbits(r.depth) << 32) | (uint32_t)i);
      }
  }
  c.tile_offset.assign(T + 1, 0);
  for (int t = 0; t < T; t++) c.tile_offset[t + 1] = c.tile_offset[t] + (int)lists[t].size();
  c.keys.resize(c.tile_offset[T]);
  c.final_T.assign((size_t)W * H, 1.f);
  c.n_contrib.assign((size_t)W * H, 0);
  std::vector<int32_t> touched_acc(N, 0);
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < T; t++) {
    std::sort(lists[t].begin(), lists[t].end());
    std::copy(lists[t].begin(), lists[t].end(), c.keys.begin() + c.tile_offset[t]);
    const int tx = t % cam.grid_x, ty = t / cam.grid_x;
    const int n = (int)lists[t].size();
    for (int ly = 0; ly < kTile; ly++)
      for (int lx = 0; lx < kTile; lx++) {
        const int px = tx * kTile + lx, py = ty * kTile + ly;
        if (px >= W || py >= H) continue;
        float Tr = 1.f, C[3] = {0, 0, 0}, D = 0.f;
        int last = 0;
        for (int k = 0; k < n; k++) {
          const uint32_t id = (uint32_t)lists[t][k];
          bool touched;
          const int rc = blend_forward_step((float)px, (float)py, lite_of(c.rec[id]), Tr, C, D, touched);
          if (rc == 2) break;
          if (rc == 1) {
            last = k + 1;
            if (touched) {
#pragma omp atomic
              touched_acc[id]++;
            }
          }
        }
        const size_t pix = (size_t)py * W + px;
        c.final_T[pix] = Tr;
        c.n_contrib[pix] = last;
        for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * W * H + pix] = C[ch] + Tr * bg[ch];
        out_depth[pix] = D;
        out_opacity[pix] = 1.f - Tr;
      }
  }
  for (int i = 0; i < N; i++) n_touched[i] = touched_acc[i];
  return (int64_t)c.keys.size();
}

// Backward for the state left by the last emul_forward on this handle.
void emul_backward(void* h, const float* bg, const float* means3D, const float* scales,
                   const float* rotations, const float* cov_pre, const float* shs,
                   const float* grad_color, const float* grad_depth, float* dmeans3D,
                   float* dmeans2D, float* dshs_or_col, float* dopacity, float* dscales,
                   float* drot, float* dcov, float* dtau) {
  Ctx& c = *(Ctx*)h;
  const Camera& cam = c.cam;
  const int N = c.N, W = cam.W, H = cam.H, T = cam.grid_x * cam.grid_y;
  std::vector<double> acc((size_t)N * 10, 0.0);
#pragma omp parallel
  {
    std::vector<double> loc((size_t)N * 10, 0.0);
#pragma omp for schedule(dynamic, 1)
    for (int t = 0; t < T; t++) {
      const int tx = t % cam.grid_x, ty = t / cam.grid_x;
      const uint64_t* list = c.keys.data() + c.tile_offset[t];
      for (int ly = 0; ly < kTile; ly++)
        for (int lx = 0; lx < kTile; lx++) {
          const int px = tx * kTile + lx, py = ty * kTile + ly;
          if (px >= W || py >= H) continue;
          const size_t pix = (size_t)py * W + px;
          const float gp[3] = {grad_color[pix], grad_color[(size_t)W * H + pix],
                               grad_color[2 * (size_t)W * H + pix]};
          PixBwd st;
          pixbwd_init(st, c.final_T[pix], gp, grad_depth ? grad_depth[pix] : 0.f, bg);
          for (int k = c.n_contrib[pix] - 1; k >= 0; k--) {
            const uint32_t id = (uint32_t)list[k];
            SplatGrad g;
            if (!blend_backward_step((float)px, (float)py, lite_of(c.rec[id]), st, g)) continue;
            double* a = loc.data() + (size_t)id * 10;
            a[0] += g.gx; a[1] += g.gy; a[2] += g.gA; a[3] += g.gB; a[4] += g.gC;
            a[5] += g.gop; a[6] += g.gr; a[7] += g.gg; a[8] += g.gb; a[9] += g.gdepth;
          }
        }
    }
#pragma omp critical
    for (size_t i = 0; i < acc.size(); i++) acc[i] += loc[i];
  }
  double tau[6] = {0, 0, 0, 0, 0, 0};
  const int K = cam.sh_coeffs;
  for (int i = 0; i < N; i++) {
    const SplatRec& r = c.rec[i];
    float* dm = dmeans3D + 3 * i;
    dm[0] = dm[1] = dm[2] = 0.f;
    dmeans2D[3 * i] = dmeans2D[3 * i + 1] = dmeans2D[3 * i + 2] = 0.f;
    dopacity[i] = 0.f;
    if (dscales) dscales[3 * i] = dscales[3 * i + 1] = dscales[3 * i + 2] = 0.f;
    if (drot) drot[4 * i] = drot[4 * i + 1] = drot[4 * i + 2] = drot[4 * i + 3] = 0.f;
    if (dcov) for (int k = 0; k < 6; k++) dcov[6 * i + k] = 0.f;
    const int ncol = shs ? 3 * K : 3;
    for (int k = 0; k < ncol; k++) dshs_or_col[(size_t)ncol * i + k] = 0.f;
    if (r.radius <= 0) continue;
    const double* a = acc.data() + (size_t)i * 10;
    const float g_xy[2] = {(float)a[0], (float)a[1]};
    const float g_con[3] = {(float)a[2], (float)a[3], (float)a[4]};
    const float g_rgb[3] = {(float)a[6], (float)a[7], (float)a[8]};
    GaussGrad gg;
    project_gaussian_backward(cam, means3D + 3 * i, scales ? scales + 3 * i : nullptr,
                              rotations ? rotations + 4 * i : nullptr,
                              cov_pre ? cov_pre + 6 * i : nullptr, g_xy, g_con, (float)a[5],
                              (float)a[9], gg);
    if (shs) {
      sh_backward(cam.sh_degree, K, shs + 3 * (size_t)K * i, means3D + 3 * i, cam.campos, r.flags,
                  g_rgb, dshs_or_col + 3 * (size_t)K * i, gg.dmean);
    } else {
      for (int k = 0; k < 3; k++) dshs_or_col[3 * i + k] = g_rgb[k];
    }
    for (int k = 0; k < 3; k++) dm[k] = gg.dmean[k];
    dmeans2D[3 * i] = gg.dndc[0]; dmeans2D[3 * i + 1] = gg.dndc[1];
    dopacity[i] = gg.dop;
    if (dscales) for (int k = 0; k < 3; k++) dscales[3 * i + k] = gg.dscale[k];
    if (drot) for (int k = 0; k < 4; k++) drot[4 * i + k] = gg.drot[k];
    if (dcov) for (int k = 0; k < 6; k++) dcov[6 * i + k] = gg.dcov6[k];
    for (int k = 0; k < 6; k++) tau[k] += gg.dtau[k];
  }
  for (int k = 0; k < 6; k++) dtau[k] = (float)tau[k];
}

// number of OpenMP threads the emulation runs on (bench.py reports it as cpu_baseline.cores)
// exclusive per-tile pair offsets of the last emul_forward (T + 1 entries)
void emul_tile_offsets(void* h, int32_t* out, int n) {
  Ctx& c = *(Ctx*)h;
  for (int i = 0; i < n && i < (int)c.tile_offset.size(); i++) out[i] = c.tile_offset[i];
}

int emul_num_threads() { return omp_get_max_threads(); }

}  // extern "C"
