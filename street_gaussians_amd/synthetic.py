"""Synthetic street-like Gaussian scenes and cameras (SURVEY.md 8d recipe).

Used by bench.py and the tests so that every number is quoted on the same inputs.  The camera
matrices follow the reference's convention (/root/reference/lib/utils/camera_utils.py:52-61 and
lib/utils/graphics_utils.py:51-70): ``viewmatrix = W2C.T`` and ``projmatrix = W2C.T @ P.T`` --
i.e. the tensors hold the TRANSPOSED (row-vector) matrices, flat index ``m[4*r+c]`` is element
(c, r) of the column-vector matrix.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class Camera:
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    viewmatrix: torch.Tensor  # [4,4] = W2C.T
    projmatrix: torch.Tensor  # [4,4] = W2C.T @ P.T
    campos: torch.Tensor      # [3]

    def to(self, device):
        return Camera(self.image_height, self.image_width, self.tanfovx, self.tanfovy,
                      self.viewmatrix.to(device), self.projmatrix.to(device), self.campos.to(device))


def projection_matrix(znear: float, zfar: float, tanfovx: float, tanfovy: float) -> torch.Tensor:
    """getProjectionMatrix of the reference (lib/utils/graphics_utils.py:51-70), column-vector form."""
    top = tanfovy * znear
    bottom = -top
    right = tanfovx * znear
    left = -right
    P = torch.zeros(4, 4, dtype=torch.float64)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(width: int = 1920, height: int = 1280, fx: float = 2050.0, fy: Optional[float] = None,
                yaw_deg: float = 0.0, znear: float = 0.01, zfar: float = 1000.0,
                translation=(0.0, 0.0, 0.0)) -> Camera:
    """Waymo-front-like pinhole camera; view k of the multi-view benchmark is yawed by k*5 degrees."""
    fy = fx if fy is None else fy
    tanfovx = width / (2.0 * fx)
    tanfovy = height / (2.0 * fy)
    a = math.radians(yaw_deg)
    # world -> camera: rotate about the camera's y axis, then translate
    R = torch.tensor([[math.cos(a), 0.0, -math.sin(a)], [0.0, 1.0, 0.0], [math.sin(a), 0.0, math.cos(a)]],
                     dtype=torch.float64)
    W2C = torch.eye(4, dtype=torch.float64)
    W2C[:3, :3] = R
    W2C[:3, 3] = torch.tensor(translation, dtype=torch.float64)
    P = projection_matrix(znear, zfar, tanfovx, tanfovy)
    view_t = W2C.t().contiguous()
    proj_t = (W2C.t() @ P.t()).contiguous()
    campos = torch.linalg.inv(W2C)[:3, 3]
    return Camera(height, width, tanfovx, tanfovy, view_t.float(), proj_t.float(), campos.float().contiguous())


@dataclass
class Scene:
    means3D: torch.Tensor    # [P,3]
    scales: torch.Tensor     # [P,3]  (post-activation)
    rotations: torch.Tensor  # [P,4]  (w,x,y,z) normalised
    opacities: torch.Tensor  # [P,1]
    shs: torch.Tensor        # [P,M,3]
    semantics: torch.Tensor  # [P,S]

    def to(self, device):
        return Scene(*[t.to(device) for t in (self.means3D, self.scales, self.rotations, self.opacities, self.shs,
                                              self.semantics)])

    @property
    def P(self):
        return self.means3D.shape[0]


def make_scene(P: int, cam: Camera, sh_degree_max: int = 3, S: int = 0, seed: int = 0,
               zmin: float = 1.0, zmax: float = 80.0, scale_px: float = 0.0015, scale_sigma: float = 0.6,
               margin: float = 1.1) -> Scene:
    """means uniform in the view-0 frustum with depth uniform in 1/z over [zmin, zmax],
    x,y = z*tanfov*U(-margin, margin); scales = exp(N(log(scale_px*z), scale_sigma^2)) per axis;
    rotations = normalize(N(0,1)^4); opacity = sigmoid(N(0, 2^2)); shs = N(0, 0.3^2) with higher bands
    damped and a DC offset so colours sit in range; semantics = N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(P, generator=g, dtype=torch.float64)
    inv_z = 1.0 / zmax + u * (1.0 / zmin - 1.0 / zmax)
    z = 1.0 / inv_z
    ux = (torch.rand(P, generator=g, dtype=torch.float64) * 2 - 1) * margin
    uy = (torch.rand(P, generator=g, dtype=torch.float64) * 2 - 1) * margin
    x = z * cam.tanfovx * ux
    y = z * cam.tanfovy * uy
    means = torch.stack([x, y, z], dim=1).float()
    logs = torch.log(scale_px * z)[:, None] + scale_sigma * torch.randn(P, 3, generator=g, dtype=torch.float64)
    scales = torch.exp(logs).float()
    q = torch.randn(P, 4, generator=g, dtype=torch.float64)
    q = (q / q.norm(dim=1, keepdim=True)).float()
    opac = torch.sigmoid(2.0 * torch.randn(P, 1, generator=g, dtype=torch.float64)).float()
    M = (sh_degree_max + 1) ** 2
    shs = 0.3 * torch.randn(P, M, 3, generator=g, dtype=torch.float64)
    if M > 1:
        shs[:, 1:, :] *= 0.3
    shs[:, 0, :] += 0.5  # DC offset
    sem = torch.randn(P, S, generator=g, dtype=torch.float64).float()
    return Scene(means.contiguous(), scales.contiguous(), q.contiguous(), opac.contiguous(),
                 shs.float().contiguous(), sem.contiguous())


def loss_weights(cam: Camera, S: int = 0, seed: int = 1):
    """Fixed random per-pixel weights so every dL_dout_* is dense."""
    g = torch.Generator().manual_seed(seed)
    H, W = cam.image_height, cam.image_width
    return dict(color=torch.rand(3, H, W, generator=g) - 0.5, depth=(torch.rand(1, H, W, generator=g) - 0.5) * 0.1,
                alpha=torch.rand(1, H, W, generator=g) - 0.5, semantic=torch.rand(S, H, W, generator=g) - 0.5)


def smoke_test_camera() -> dict:
    """The hard-coded KITTI-like camera of the reference's only test
    (/root/reference/script/test_gaussian_rasterization.py:6-20)."""
    return dict(
        image_height=375, image_width=1242, FoVx=1.416, FoVy=0.506,
        world_view_transform=torch.tensor([[0.9598, 0.0081, 0.2806, 0.0000], [-0.0123, 0.9998, 0.0134, 0.0000],
                                           [-0.2804, -0.0163, 0.9597, 0.0000], [-2.0954, -0.0935, 4.9320, 1.0000]]),
        full_proj_transform=torch.tensor([[1.1205, 0.0312, 0.2806, 0.2806], [-0.0144, 3.8661, 0.0134, 0.0134],
                                          [-0.3274, -0.0632, 0.9598, 0.9597], [-2.4464, -0.3614, 4.9225, 4.9320]]),
        camera_center=torch.tensor([6.2808e-01, 1.4572e-03, -5.3226e+00]))
