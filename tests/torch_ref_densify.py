"""Reference of adaptive density control for street_gaussians_amd/densify.py in plain torch ops on plain tensors --
test infrastructure only.  It performs the steps of GaussianModel.densify_and_prune
(/root/reference/lib/models/gaussian_model.py:522-553) in the reference's order, with the optimiser surgery
(cat_optimizer / prune_optimizer, :363-407) applied to explicit (exp_avg, exp_avg_sq) pairs, and with the split's
normal samples supplied by the caller as standard normals."""
import torch

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "semantic")


def _rotmat(q):  # utils/general_utils.py:125-146
    q = q / torch.sqrt((q * q).sum(1, keepdim=True))
    w, x, y, z = q.unbind(-1)
    return torch.stack([torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)), -1),
                        torch.stack((2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)), -1),
                        torch.stack((2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), -1)], -2)


class Model:
    """Just enough of GaussianModel: raw parameters, Adam moments, the three statistics."""

    def __init__(self, params, states, accum, denom):
        self.p = {k: v.clone() for k, v in params.items()}
        self.s = {k: (a.clone(), b.clone()) for k, (a, b) in states.items()}
        self.accum, self.denom = accum.clone(), denom.clone()

    def _extend(self, new):  # densification_postfix + cat_optimizer (:383-407, :429-446)
        for k in NAMES:
            self.p[k] = torch.cat((self.p[k], new[k]), 0)
            a, b = self.s[k]
            self.s[k] = (torch.cat((a, torch.zeros_like(new[k])), 0), torch.cat((b, torch.zeros_like(new[k])), 0))

    def _prune(self, mask):  # prune_points + prune_optimizer (:363-381, :409-427)
        keep = ~mask
        for k in NAMES:
            self.p[k] = self.p[k][keep]
            self.s[k] = (self.s[k][0][keep], self.s[k][1][keep])

    def reset_opacity(self):  # gaussian_model.py:410-414 with reset_optimizer :344-361
        op = torch.sigmoid(self.p["opacity"])
        x = torch.min(op, torch.ones_like(op) * 0.01)
        self.p["opacity"] = torch.log(x / (1 - x))
        self.s["opacity"] = (torch.zeros_like(self.p["opacity"]), torch.zeros_like(self.p["opacity"]))

    def densify_and_prune(self, max_grad, min_opacity, extent, percent_dense, percent_big_ws, prune_big, normals,
                          grad_column=0, N=2, variant=None, sphere_center=None, sphere_radius=None, box_min=None,
                          box_max=None, box_normals=None):
        """variant None: GaussianModel (gaussian_model.py:522-553); "bkgd": GaussianModelBkgd
        (gaussian_model_bkgd.py:74-114); "actor": GaussianModelActor (gaussian_model_actor.py:204-261), whose
        torch.normal(mean=0, std=scale) samples are box_normals * scale."""
        scalars = {"points_total": self.p["xyz"].shape[0]}
        grads = self.accum[:, grad_column:grad_column + 1] / self.denom      # :523
        grads[grads.isnan()] = 0.0                                           # :524
        scale = lambda: torch.exp(self.p["scaling"])
        # clone (:494-520)
        sel = (torch.norm(grads, dim=-1) >= max_grad) & (scale().max(dim=1).values <= percent_dense * extent)
        scalars["points_clone"] = int(sel.sum())
        self._extend({k: self.p[k][sel] for k in NAMES})
        # split (:448-492)
        n_now = self.p["xyz"].shape[0]
        padded = torch.zeros(n_now, dtype=grads.dtype, device=grads.device)
        padded[:grads.shape[0]] = grads.squeeze(-1)
        sel = (padded >= max_grad) & (scale().max(dim=1).values > percent_dense * extent)
        scalars["points_split"] = int(sel.sum())
        stds = scale()[sel].repeat(N, 1)
        samples = normals[:stds.shape[0]].to(stds.dtype) * stds              # normal(mean=0, std=stds)
        rots = _rotmat(self.p["rotation"][sel]).repeat(N, 1, 1)
        new = {k: self.p[k][sel].repeat(N, *([1] * (self.p[k].dim() - 1))) for k in NAMES}
        new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.p["xyz"][sel].repeat(N, 1)
        new["scaling"] = torch.log(scale()[sel].repeat(N, 1) / (0.8 * N))
        self._extend(new)
        self._prune(torch.cat((sel, torch.zeros(N * int(sel.sum()), dtype=torch.bool, device=sel.device))))
        # prune (:532-543)
        mask = (torch.sigmoid(self.p["opacity"]) < min_opacity).squeeze(-1)
        if variant == "bkgd":
            scalars["points_below_min_opacity"] = int(mask.sum())
        if prune_big:
            big = scale().max(dim=1).values > extent * percent_big_ws
            if variant == "bkgd":  # gaussian_model_bkgd.py:95-101
                dists = torch.linalg.norm(self.p["xyz"] - sphere_center, dim=1)
                big[dists > 2 * sphere_radius] = False
                scalars["points_big_ws"] = int(big.sum())
            mask = mask | big
            if variant == "actor":  # gaussian_model_actor.py:231-249
                stds = scale()[:, None, :].expand(-1, 2, -1)
                samples = box_normals.to(stds.dtype) * stds
                rots = _rotmat(self.p["rotation"])[:, None, :, :].expand(-1, 2, -1, -1)
                origins = self.p["xyz"][:, None, :].expand(-1, 2, -1)
                sx = torch.matmul(rots, samples.unsqueeze(-1)).squeeze(-1) + origins
                n = sx.shape[0]
                inside = torch.all((sx >= box_min).view(n, -1), dim=-1) & torch.all((sx <= box_max).view(n, -1), dim=-1)
                mask = mask | ~inside
        self._prune(mask)
        scalars["points_pruned"] = int(mask.sum())
        return scalars
