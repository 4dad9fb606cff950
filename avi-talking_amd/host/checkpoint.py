"""On-disk formats of the reference entry points, written from / read into the MI355X objects.

* ``{best,last}.pth`` of ``train_diffusion_prior.py`` ``save_ckpt`` (:155-168) / ``resume_ckpt`` (:238-251): a dict with
  ``epoch``, ``model_state_dict`` (reference key names), ``optimizer_state_dict`` (``torch.optim.AdamW.state_dict()`` over
  the FOUR parameter groups built at :997-1003: net decay / net no-decay / voxel2clip decay / voxel2clip no-decay, in
  ``named_parameters()`` order), ``lr_scheduler``, ``train_losses``, ``val_losses``, ``lrs``.
* ``flame_*.pkl`` of ``evaluation_functions.py:624-638``: ``{"shape", "expression", "jaw_pose", "global_pose"}`` numpy arrays.

Parameter ORDER inside a group matters for ``optimizer_state_dict`` (torch indexes parameters by position) and is the order
of ``Module.named_parameters()``: a module's OWN parameters before its sub-modules', sub-modules in registration order
(``torch_parameter_order``).  So in ``net`` the three direct parameters ``learned_query`` / ``null_brain_embeds`` /
``null_image_embed`` (models/diffusion_prior.py:196,203-204) come before ``to_time_embeds`` and ``causal_transformer``, and
inside every dalle2 ``Attention`` its direct ``null_kv`` before ``norm`` / ``to_q`` / ... (round 4: found by cross-loading
with the reference's own ``save_ckpt`` / ``resume_ckpt``, tests/test_reference_live.py; the given key order used to be taken
as is).  Sibling order: for ``voxel2clip`` the registration order of the reference ``BrainNetwork`` (pinned by
tests/golden/brain.npz key order); for ``net`` the reference's constructors (:169-207, :119-152) and, inside the dalle2 blocks,
the order of ``avi_talking_amd.weights.make_prior_weights`` keys, which restates dalle2's registration order from memory
(dalle2_pytorch is absent: unpinned, like the rest of that class - DESIGN.md section 2).
Loading uses ``torch.load(..., weights_only=True)``: checkpoints are data, never code."""
import os
import pickle

import numpy as np
import torch

from .training import no_decay

NET, V2C = "net.", "voxel2clip."


def torch_parameter_order(names):
    """``names`` re-ordered the way ``torch.nn.Module.named_parameters()`` yields the parameters of the module tree the
    dotted names imply: at every module its own parameters first (in the given order), then its sub-modules in the order
    they first appear."""
    tree = {"p": [], "c": {}}
    for n in names:
        node = tree
        parts = n.split(".")
        for part in parts[:-1]:
            node = node["c"].setdefault(part, {"p": [], "c": {}})
        node["p"].append(n)
    out = []

    def walk(node):
        out.extend(node["p"])
        for child in node["c"].values():
            walk(child)
    walk(tree)
    return out


def param_groups(names):
    """The four groups of train_diffusion_prior.py:997-1003 as lists of parameter names, each in ``named_parameters()``
    order of its sub-model (``torch_parameter_order``)."""
    g = [[], [], [], []]
    for n in torch_parameter_order(names):
        if n.startswith(NET):
            g[1 if no_decay(n[len(NET):]) else 0].append(n)
        elif n.startswith(V2C):
            g[3 if no_decay(n[len(V2C):]) else 2].append(n)
    return g


def model_state_dict(trainer, ordered_names):
    """Named fp32 tensors on the CPU under the reference's key names (+ the NoiseScheduler buffers the sampler mirrors)."""
    sd = {n: trainer.store.view(n).detach().cpu().clone() for n in ordered_names}
    for k, v in trainer.sched.items():
        sd["noise_scheduler." + k] = v.detach().cpu().clone()
    return sd


def optimizer_state_dict(trainer, ordered_names):
    S = trainer.store
    groups = param_groups(ordered_names)
    state, pg, idx = {}, [], 0
    b1, b2 = trainer.betas
    for gi, names in enumerate(groups):
        ids = []
        for n in names:
            if trainer.step_count > 0:
                state[idx] = {"step": torch.tensor(float(trainer.step_count)),
                              "exp_avg": S.view(n, S.M).detach().cpu().clone(),
                              "exp_avg_sq": S.view(n, S.V).detach().cpu().clone()}
            ids.append(idx)
            idx += 1
        pg.append({"lr": trainer.lr, "betas": (b1, b2), "eps": trainer.eps,
                   "weight_decay": trainer.wd if gi in (0, 2) else 0.0, "amsgrad": False, "maximize": False,
                   "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": ids})
    return {"state": state, "param_groups": pg}


def save_ckpt(tag, outdir, epoch, trainer, ordered_names, lr_scheduler=None, losses=(), val_losses=(), lrs=()):
    """train_diffusion_prior.py:155-168."""
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, f"{tag}.pth")
    torch.save({"epoch": epoch,
                "model_state_dict": model_state_dict(trainer, ordered_names),
                "optimizer_state_dict": optimizer_state_dict(trainer, ordered_names),
                "lr_scheduler": lr_scheduler if lr_scheduler is not None else {},
                "train_losses": list(losses), "val_losses": list(val_losses), "lrs": list(lrs)}, path)
    return path


def resume_ckpt(path, trainer, ordered_names):
    """train_diffusion_prior.py:238-251: restores the model and the optimizer (not the lr scheduler), returns epoch."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    S = trainer.store
    sd = ck["model_state_dict"]
    missing = [n for n in S.names if n not in sd]
    if missing:
        raise KeyError(f"checkpoint lacks parameters {missing[:3]}...")
    for n in S.names:
        S.view(n).copy_(sd[n].to(S.P.device, torch.float32).reshape(S.shape[n]))
    opt = ck["optimizer_state_dict"]
    flat = [n for g in param_groups(ordered_names) for n in g]
    step = 0
    S.M.zero_()
    S.V.zero_()
    for i, n in enumerate(flat):
        st = opt["state"].get(i)
        if st is None:
            continue
        S.view(n, S.M).copy_(st["exp_avg"].to(S.P.device, torch.float32).reshape(S.shape[n]))
        S.view(n, S.V).copy_(st["exp_avg_sq"].to(S.P.device, torch.float32).reshape(S.shape[n]))
        step = max(step, int(float(st["step"])))
    trainer.step_count = step
    if opt["param_groups"]:
        trainer.lr = float(opt["param_groups"][0]["lr"])
    trainer.reload_planes()
    return ck["epoch"]


def flame_dict(gt_shape, predicted_exp, predicted_jaw):
    """evaluation_functions.py:627-631 for ONE clip: shape (n_shape,), exp (T,50), jaw (T,3)."""
    jaw = predicted_jaw.detach().cpu().numpy()
    return {"shape": gt_shape.detach().cpu().numpy(), "expression": predicted_exp.detach().cpu().numpy(),
            "jaw_pose": jaw, "global_pose": np.zeros_like(jaw)}


def save_flame_pkl(path, gt_shape, predicted_exp, predicted_jaw, overwrite=True):
    """evaluation_functions.py:633-637 (the file holds only numpy arrays in a dict)."""
    if os.path.exists(path) and not overwrite:
        return path
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(flame_dict(gt_shape, predicted_exp, predicted_jaw), f)
    return path
