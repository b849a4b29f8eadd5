"""Independent float64 torch re-statement of the rasterizer maths in conventional
(row-major, column-vector) notation, differentiated by torch.autograd.

Third implementation used only to validate the C oracle's forward values and hand-written backward
formulas on tiny scenes: the tile lists / visibility decisions are taken from the oracle, every
differentiable quantity is recomputed here.  Quirks of the reference that autograd would not
reproduce are mimicked explicitly (each one cites the reference line):
  * alpha = min(0.99, o*G) has NO gradient mask (backward.cu:543,618,638)  -> straight-through;
  * the tan-fov clamp of t.x/t.z zeroes only dL/dt.x (backward.cu:175-176,262-264) -> detach;
  * dL/dscale omits the scale_modifier chain factor (backward.cu:321-325) -> only mod == 1 is tested.
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def sh_to_rgb(deg, shs, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6]
               + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return res + 0.5


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)
    return R


def cov3d_from_scale_rot(scales, rots, mod):
    R = quat_to_rot(rots)
    s = scales * mod
    M = R * s[:, None, :]
    return M @ M.transpose(1, 2)


def sym6_to_mat(c):
    return torch.stack([torch.stack([c[:, 0], c[:, 1], c[:, 2]], -1), torch.stack([c[:, 1], c[:, 3], c[:, 4]], -1),
                        torch.stack([c[:, 2], c[:, 4], c[:, 5]], -1)], 1)


def preprocess(means3D, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H, deg, opacities, shs=None,
               colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0):
    Vt, Pt = viewmatrix.double(), projmatrix.double()
    P = means3D.shape[0]
    ph = torch.cat([means3D, torch.ones(P, 1, dtype=torch.float64)], 1)
    t = (ph @ Vt)[:, :3]
    hom = ph @ Pt
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    Sigma = sym6_to_mat(cov3D_precomp) if cov3D_precomp is not None else cov3d_from_scale_rot(scales, rotations,
                                                                                            scale_modifier)
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = t[:, 2]
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    tx = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * tz).detach(), t[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * tz).detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -fx * tx / (tz * tz)], -1),
                     torch.stack([zero, fy / tz, -fy * ty / (tz * tz)], -1)], 1)
    Rwc = Vt[:3, :3].t()
    A = J @ Rwc
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a, b, c = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], -1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3 * torch.sqrt(lam))
    pix = torch.stack([((ndc[:, 0] + 1) * W - 1) * 0.5, ((ndc[:, 1] + 1) * H - 1) * 0.5], -1)
    if colors_precomp is None:
        d = means3D - campos.double()[None]
        d = d / d.norm(dim=1, keepdim=True)
        raw = sh_to_rgb(deg, shs, d)
        rgb = raw.clamp_min(0.0)
        clamped = raw < 0
    else:
        rgb, clamped = colors_precomp, None
    return dict(depth=t[:, 2], pix=pix, conic=conic, radius=radius, rgb=rgb, clamped=clamped, cov3D=Sigma,
                cov2=torch.stack([a, b, c], -1), opacity=opacities[:, 0])


def render(pre, point_list, ranges, W, H, bg, semantics=None):
    """Per-tile front-to-back compositing, vectorised over the pixels of a tile."""
    S = 0 if semantics is None else semantics.shape[1]
    color = torch.zeros(3, H, W, dtype=torch.float64)
    depth = torch.zeros(1, H, W, dtype=torch.float64)
    alpha_o = torch.zeros(1, H, W, dtype=torch.float64)
    sem = torch.zeros(S, H, W, dtype=torch.float64)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    gx = (W + 15) // 16
    outs = []
    for tile in range(ranges.shape[0]):
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        ty, tx = divmod(tile, gx)
        ys = torch.arange(ty * 16, min(ty * 16 + 16, H))
        xs = torch.arange(tx * 16, min(tx * 16 + 16, W))
        if len(ys) == 0 or len(xs) == 0:
            continue
        py, px = torch.meshgrid(ys, xs, indexing="ij")
        pxf, pyf = px.reshape(-1).double(), py.reshape(-1).double()
        n = pxf.numel()
        T = torch.ones(n, dtype=torch.float64)
        Cc = torch.zeros(n, 3, dtype=torch.float64)
        Dd = torch.zeros(n, dtype=torch.float64)
        Aa = torch.zeros(n, dtype=torch.float64)
        Ss = torch.zeros(n, S, dtype=torch.float64)
        done = torch.zeros(n, dtype=torch.bool)
        last = torch.zeros(n, dtype=torch.int64)
        for k, gi in enumerate(point_list[r0:r1].tolist()):
            dx = pre["pix"][gi, 0] - pxf
            dy = pre["pix"][gi, 1] - pyf
            cn = pre["conic"][gi]
            power = -0.5 * (cn[0] * dx * dx + cn[2] * dy * dy) - cn[1] * dx * dy
            og = pre["opacity"][gi] * torch.exp(power)
            alpha = og + (og.clamp(max=0.99) - og).detach()
            valid = (~done) & (power <= 0) & (alpha >= 1.0 / 255.0)
            testT = T * (1 - alpha)
            newdone = valid & (testT < 1e-4)
            contrib = valid & ~newdone
            w = torch.where(contrib, alpha * T, torch.zeros_like(T))
            Cc = Cc + w[:, None] * pre["rgb"][gi][None]
            Dd = Dd + w * pre["depth"][gi]
            Aa = Aa + w
            if S:
                Ss = Ss + w[:, None] * semantics[gi][None]
            T = torch.where(contrib, testT, T)
            last = torch.where(contrib, torch.full_like(last, k + 1), last)
            done = done | newdone
        outs.append((py.reshape(-1), px.reshape(-1), Cc + T[:, None] * bg.double()[None], Dd, Aa, Ss, last))
    for py, px, Cc, Dd, Aa, Ss, last in outs:
        color[:, py, px] = Cc.t()
        depth[0, py, px] = Dd
        alpha_o[0, py, px] = Aa
        if S:
            sem[:, py, px] = Ss.t()
        n_contrib[py, px] = last
    return color, depth, alpha_o, sem, n_contrib
