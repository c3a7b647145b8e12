"""Literal numpy restatement of the reference's OWN in-tree EWA splat code (TEST
INFRASTRUCTURE ONLY): the OpenGL viewer shaders

    /root/reference/gui/gl_render/shaders/gau_vert.glsl:60-159   computeCov3D, computeCov2D,
                                                                 conic inversion
    /root/reference/gui/gl_render/shaders/gau_frag.glsl:20-26    per-fragment alpha rule

The rasteriser extension itself is an absent submodule, but these shaders are the
reference authors' restatement of the same per-Gaussian projection and alpha rule, so they
pin the constants the oracle otherwise takes from upstream knowledge: the +-1.3 tan(fov)
clamp of the EWA Jacobian, the +0.3 px^2 low-pass, conic = inverse 2-D covariance,
alpha = min(0.99, opacity * exp(power)), `power > 0` and `alpha < 1/255` discarded.

GLSL matrices are column-major: `mat3(a, b, c, d, e, f, g, h, i)` has COLUMNS (a,b,c),
(d,e,f), (g,h,i) and `M[i][j]` is column i, row j.  `glsl_mat3` reproduces that so the
shader source can be followed line by line.  The viewer works in OpenGL camera coordinates
(x right, y up, camera looks down -z): its view matrix is cv_gl @ T_w2c with
cv_gl = diag(1, -1, -1, 1) (gui/gui_utils.py:14, gui/slam_gui.py:515), so 2-D quantities
come out with the y axis flipped - callers compare (a, -b, c).
"""
from __future__ import annotations

import numpy as np

CV_GL = np.diag([1.0, -1.0, -1.0, 1.0])          # gui/gui_utils.py:14


def glsl_mat3(*v):
    """GLSL mat3 constructor from 9 scalars (column-major) as a math-convention array."""
    return np.array(v, dtype=np.float64).reshape(3, 3).T


def compute_cov3d(scale, q):
    """gau_vert.glsl:60-82."""
    S = np.diag(np.asarray(scale, dtype=np.float64))
    r, x, y, z = (float(t) for t in q)
    R = glsl_mat3(
        1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y - r * z), 2.0 * (x * z + r * y),
        2.0 * (x * y + r * z), 1.0 - 2.0 * (x * x + z * z), 2.0 * (y * z - r * x),
        2.0 * (x * z - r * y), 2.0 * (y * z + r * x), 1.0 - 2.0 * (x * x + y * y))
    M = S @ R
    return M.T @ M


def compute_cov2d(mean_view, focal_x, focal_y, tan_fovx, tan_fovy, cov3d, viewmatrix):
    """gau_vert.glsl:84-111; returns (cov[0][0], cov[0][1], cov[1][1])."""
    t = np.array(mean_view, dtype=np.float64)
    limx, limy = 1.3 * tan_fovx, 1.3 * tan_fovy
    txtz, tytz = t[0] / t[2], t[1] / t[2]
    t[0] = min(limx, max(-limx, txtz)) * t[2]
    t[1] = min(limy, max(-limy, tytz)) * t[2]
    J = glsl_mat3(
        focal_x / t[2], 0.0, -(focal_x * t[0]) / (t[2] * t[2]),
        0.0, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2]),
        0.0, 0.0, 0.0)
    W = np.asarray(viewmatrix, dtype=np.float64)[:3, :3].T     # transpose(mat3(viewmatrix))
    T = W @ J
    cov = T.T @ cov3d.T @ T
    cov[0, 0] += 0.3
    cov[1, 1] += 0.3
    # GLSL cov[0][1] = column 0, row 1
    return np.array([cov[0, 0], cov[1, 0], cov[1, 1]])


def conic_from_cov2d(cov2d):
    """gau_vert.glsl:149-154."""
    det = cov2d[0] * cov2d[2] - cov2d[1] * cov2d[1]
    det_inv = 1.0 / det
    return np.array([cov2d[2] * det_inv, -cov2d[1] * det_inv, cov2d[0] * det_inv])


def quad_half_extent(cov2d):
    """gau_vert.glsl:156: the viewer draws the splat on a +-3 sigma axis-aligned quad."""
    return 3.0 * np.sqrt(cov2d[0]), 3.0 * np.sqrt(cov2d[2])


def fragment_alpha(conic, coordxy, opacity):
    """gau_frag.glsl:20-26; 0.0 stands for `discard`."""
    x, y = coordxy
    power = -0.5 * (conic[0] * x * x + conic[2] * y * y) - conic[1] * x * y
    if power > 0.0:
        return 0.0
    a = min(0.99, opacity * np.exp(power))
    if a < 1.0 / 255.0:
        return 0.0
    return a


def splat(mean_world, scale, quat, T_w2c, focal, tan_fovx, tan_fovy, scale_modifier=1.0):
    """The vertex shader's per-Gaussian path (gau_vert.glsl:123-154) for a computer-vision
    world-to-camera pose: returns (cov2d, conic) in the viewer's y-up convention."""
    view_gl = CV_GL @ np.asarray(T_w2c, dtype=np.float64)
    p_view = view_gl @ np.append(np.asarray(mean_world, dtype=np.float64), 1.0)
    cov3d = compute_cov3d(np.asarray(scale, dtype=np.float64) * scale_modifier, quat)
    # the shader passes hfovxy_focal.z for BOTH focal lengths (:140-146): square pixels
    cov2d = compute_cov2d(p_view, focal, focal, tan_fovx, tan_fovy, cov3d, view_gl)
    return cov2d, conic_from_cov2d(cov2d)
