"""Restatement, in plain torch ops on any device / dtype, of how the reference flattens its scene graph every
iteration -- test infrastructure for street_gaussians_amd/scene.py, never imported by the product.  Each function
cites the reference lines it follows (/root/reference/lib/...); tests/test_scene_cpu.py pins the quaternion helpers
against the reference's own functions when /root/reference is present."""
import torch
import torch.nn.functional as F


def quaternion_raw_multiply(a, b):
    """Hamilton product of (w, x, y, z) quaternions with broadcasting (utils/general_utils.py:220-238)."""
    w1, x1, y1, z1 = a.unbind(-1)
    w2, x2, y2, z2 = b.unbind(-1)
    return torch.stack((w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2), dim=-1)


def quaternion_to_matrix(quat):
    """[n, 4] (w, x, y, z), not necessarily unit -> [n, 3, 3] rotation matrices; the quaternion is divided by its norm
    first (utils/general_utils.py:125-146)."""
    length = torch.sqrt(quat[:, 0] * quat[:, 0] + quat[:, 1] * quat[:, 1] + quat[:, 2] * quat[:, 2] + quat[:, 3] * quat[:, 3])
    w, x, y, z = (quat / length[:, None]).unbind(-1)
    rows = [torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)), dim=-1),
            torch.stack((2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)), dim=-1),
            torch.stack((2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), dim=-1)]
    return torch.stack(rows, dim=-2)


def compose(segs, M, S):
    """segs: list of dicts with the fields of street_gaussians_amd.scene.Segment.  Follows
    models/street_gaussian_model.py:287-449 (get_scaling, get_rotation, get_xyz, get_features, get_semantic,
    get_opacity), models/gaussian_model.py:224-251 (activations) and models/gaussian_model_actor.py:62-80."""
    xyzs, rots, scales, opacs, feats, sems = [], [], [], [], [], []
    for s in segs:
        n = s["xyz"].shape[0]
        actor = s.get("pose") is not None
        rot = F.normalize(s["rotation"])                      # gaussian_model.py:222,229-230
        xyz = s["xyz"]
        if actor:
            pose = s["pose"]
            obj_rot = pose[:4].unsqueeze(0).expand(n, -1)     # street_gaussian_model.py:266-267
            obj_trans = pose[4:].unsqueeze(0).expand(n, -1)
            fm = s.get("flip_mask")
            if fm is not None:                                # :322-327, :353-356 (training mode)
                fq = torch.tensor(s.get("flip_quat", (0.0, 0.0, 1.0, 0.0)), dtype=rot.dtype, device=rot.device)
                rot = rot.clone()
                rot[fm] = quaternion_raw_multiply(fq.unsqueeze(0), rot[fm])
                xyz = xyz.clone()
                xyz[fm, s.get("flip_axis", 1)] *= -1
            rot = F.normalize(quaternion_raw_multiply(obj_rot, rot))                         # :328-329
            xyz = torch.einsum("bij, bj -> bi", quaternion_to_matrix(obj_rot), xyz) + obj_trans  # :357-358
        xyzs.append(xyz)
        rots.append(rot)
        scales.append(torch.exp(s["scaling"]))                # gaussian_model.py:214,225-226
        opacs.append(torch.sigmoid(s["opacity"]))             # :219,250-251
        dc = s["features_dc"]
        if actor and s.get("idft") is not None:               # gaussian_model_actor.py:71-80
            dc = torch.sum(dc * s["idft"][None, :, None], dim=1, keepdim=True)
        feats.append(torch.cat((dc[:, :1], s["features_rest"]), dim=1))   # gaussian_model.py:237-241
        if S > 0:
            sem = s.get("semantic")
            mode = s.get("semantic_mode", "logits")
            if sem is None:
                sems.append(torch.zeros(n, S, dtype=xyz.dtype, device=xyz.device))
            elif actor:                                       # gaussian_model_actor.py:62-69
                full = torch.zeros(n, S, dtype=xyz.dtype, device=xyz.device)
                full[:, s.get("class_label", 0)] = sem[:, 0] if mode == "logits" else torch.sigmoid(sem[:, 0])
                sems.append(full)
            else:                                             # gaussian_model.py:243-248
                sems.append(sem if mode == "logits" else F.softmax(sem, dim=1))
    cat = lambda l: torch.cat(l, dim=0)
    N = sum(x.shape[0] for x in xyzs)
    sem_out = cat(sems) if S > 0 else torch.zeros(N, 0, dtype=xyzs[0].dtype, device=xyzs[0].device)
    return cat(xyzs), cat(rots), cat(scales), cat(opacs), cat(feats), sem_out


def densification_stats(models, grad2d, radii):
    """street_gaussian_model.py:551-571 (set_max_radii2D, add_densification_stats), in place."""
    vis = radii > 0
    start = 0
    for m in models:
        end = start + m["denom"].shape[0]
        v = vis[start:end]
        g = grad2d[start:end]
        m["max_radii2D"][v] = torch.max(m["max_radii2D"][v], radii[start:end].float()[v])
        m["xyz_gradient_accum"][v, 0:1] += torch.norm(g[v, :2], dim=-1, keepdim=True)
        m["xyz_gradient_accum"][v, 1:2] += torch.norm(g[v, 2:], dim=-1, keepdim=True)
        m["denom"][v] += 1
        start = end
