"""The reference's checkpoint files: ``torch.save`` of the dictionary
``StreetGaussianModel.save_state_dict`` builds (/root/reference/lib/models/street_gaussian_model.py:138-159,
train.py:218-223) -- one entry per sub-model holding ``GaussianModel.state_dict`` (gaussian_model.py:182-205: the raw
parameters ``xyz, feature_dc, feature_rest, scaling, rotation, opacity, semantic`` and, unless ``is_final``, the
densification statistics, ``spatial_lr_scale``, ``active_sh_degree`` and the optimiser state), plus the entries of the
modules outside the rasterizer path (``actor_pose``, ``sky_cubemap``, ``color_correction``, ``pose_correction``) and
``iter``.  This module maps that layout to and from the plain per-model dictionaries plyio.py uses and to
``scene.Segment``s, so a scene trained with the reference can be rendered (and its densification continued) here.
Host-side: nothing touches the GPU until ``segments`` is asked for a device."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

# GaussianModel.state_dict key -> plyio / densify name
PARAM_KEYS = OrderedDict([("xyz", "xyz"), ("feature_dc", "features_dc"), ("feature_rest", "features_rest"),
                          ("scaling", "scaling"), ("rotation", "rotation"), ("opacity", "opacity"),
                          ("semantic", "semantic")])
EXTRA_KEYS = ("spatial_lr_scale", "denom", "max_radii2D", "xyz_gradient_accum", "active_sh_degree", "optimizer")
NON_MODEL_ENTRIES = ("actor_pose", "sky_cubemap", "color_correction", "pose_correction", "iter")


def is_model_entry(name: str, value) -> bool:
    return name not in NON_MODEL_ENTRIES and isinstance(value, dict) and all(k in value for k in ("xyz", "feature_dc"))


def models_from_state(state: Dict) -> "OrderedDict[str, Dict[str, torch.Tensor]]":
    """{model name: {xyz, features_dc, features_rest, scaling, rotation, opacity, semantic (+ the training extras that are
    present)}} from a loaded checkpoint; tensors are detached, not copied."""
    out = OrderedDict()
    for name, sd in state.items():
        if not is_model_entry(name, sd):
            continue
        m = {ours: torch.as_tensor(sd[theirs]).detach() for theirs, ours in PARAM_KEYS.items()}
        for k in EXTRA_KEYS:
            if k in sd:
                m[k] = sd[k]
        out[name] = m
    return out


def state_from_models(models: Dict[str, Dict], is_final: bool = True, iteration: Optional[int] = None,
                      others: Optional[Dict] = None, device=None) -> Dict:
    """The inverse: a dictionary with the reference's layout, ready for ``torch.save``.  With ``is_final`` only the raw
    parameters are written, like ``state_dict(is_final=True)``.

    The seven parameters are written the way the reference's own ``state_dict`` holds them -- ``nn.Parameter`` with
    ``requires_grad=True`` -- because ``GaussianModel.load_state_dict`` (gaussian_model.py:157-164) ASSIGNS the entries to
    ``_xyz``, ``_features_dc``, ... as they are and ``training_setup`` builds the optimiser over them; the reference reads
    the file with ``torch.load`` without ``map_location`` and renders from the loaded tensors directly, so a file that is
    to be resumed / rendered BY THE REFERENCE must be saved from GPU tensors (``device="cuda"`` moves them; the default
    keeps the device the tensors are on, so that saving works on a machine without a GPU)."""
    state = {}
    for name, m in models.items():
        sd = {}
        for theirs, ours in PARAM_KEYS.items():
            t = torch.as_tensor(m[ours]).detach()
            if device is not None:
                t = t.to(device)
            sd[theirs] = torch.nn.Parameter(t.float().contiguous().clone(), requires_grad=True)
        if not is_final:
            for k in EXTRA_KEYS:
                if k in m:
                    sd[k] = m[k]
        state[name] = sd
    for k, v in (others or {}).items():
        state[k] = v
    if iteration is not None:
        state["iter"] = int(iteration)
    return state


def load(path: str) -> "OrderedDict[str, Dict[str, torch.Tensor]]":
    return models_from_state(torch.load(path, map_location="cpu", weights_only=False))


def exact_storage(obj):
    """torch.save writes a tensor's WHOLE storage.  Tensors that come out of this library's densify step (and its gradient
    outputs) are views of ladder-sized storages (street_gaussians_amd/_alloc.py: up to 12.5 % of uninitialised slack behind
    the data), so they are cloned into storages of exactly their size here -- the file then holds what the reference's
    ``torch.cat`` outputs would (gaussian_model.py:363-407).  Recurses through dicts / lists / tuples."""
    if isinstance(obj, torch.Tensor):
        exact = obj.numel() * obj.element_size()
        if obj.layout == torch.strided and obj.untyped_storage().nbytes() != exact:
            return obj.detach().clone(memory_format=torch.contiguous_format)
        return obj
    if isinstance(obj, dict):
        return type(obj)((k, exact_storage(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(exact_storage(v) for v in obj)
    return obj


def save(path: str, models, **kw) -> None:
    torch.save(exact_storage(state_from_models(models, **kw)), path)


def segments(models: Dict[str, Dict], device, poses: Optional[Dict[str, torch.Tensor]] = None, idfts=None):
    """scene.Segment list in the order of ``models`` (background first, like parse_camera orders the graph).  A model
    with an entry in ``poses`` ([7] = obj_rot wxyz + obj_trans in world space for the rendered frame) becomes an actor
    segment; ``idfts`` gives each actor's IDFT row for its Fourier DC features (gaussian_model_actor.py:71-80)."""
    from . import scene
    segs = []
    for name, m in models.items():
        t = lambda k: (m[k] if torch.is_tensor(m[k]) else torch.from_numpy(np.array(m[k]))).float().to(device).contiguous()
        sem = t("semantic") if "semantic" in m and np.prod(tuple(m["semantic"].shape)) > 0 else None
        pose = poses.get(name) if poses else None
        segs.append(scene.Segment(xyz=t("xyz"), rotation=t("rotation"), scaling=t("scaling"), opacity=t("opacity"),
                                  features_dc=t("features_dc"), features_rest=t("features_rest"), semantic=sem,
                                  pose=None if pose is None else torch.as_tensor(pose).float().to(device),
                                  idft=None if pose is None or idfts is None else torch.as_tensor(idfts[name]).float().to(device)))
    return segs
