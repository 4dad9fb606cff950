"""CPU, build container only: cross-checks against the reference's OWN Python, imported live from /root/reference (skipped
where that tree does not exist - the GPU box).  Nothing here is needed by the product or by the GPU tests.

* on-disk format of ``{best,last}.pth``: ``host/checkpoint.py`` against the reference entry point's own
  ``save_ckpt`` / ``resume_ckpt`` (train_diffusion_prior.py:155-168,238-251) in BOTH directions, with the optimizer built the
  way ``main()`` builds it (:997-1004: four AdamW groups by substring match on parameter names).  The model is a stand-in
  module tree that registers parameters under the reference's key names in ``make_prior_weights`` order (small tensors: the
  format does not depend on sizes); the trainer is a CPU stand-in with the attributes ``checkpoint.py`` touches."""
import math
import os
import sys
from types import SimpleNamespace
from unittest.mock import MagicMock

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "train_diffusion_prior.py")),
                                reason="the reference tree is not present on this machine")


def _reference_entry_point():
    import importlib.util
    import transformers  # noqa: F401
    keep = dict(sys.modules)
    try:
        for name in ["cv2", "easydict", "omegaconf", "torchvision", "torchvision.transforms", "clip", "PIL", "dalle2_pytorch",
                     "inferno_apps", "inferno_apps.TalkingHead", "inferno_apps.TalkingHead.evaluation",
                     "inferno_apps.TalkingHead.evaluation.TalkingHeadWrapper",
                     "inferno_apps.TalkingHead.evaluation.evaluation_functions", "inferno", "inferno.datasets",
                     "inferno.datasets.FaceVideoDataModule", "dataset", "dataset.data_loader", "talkclip_text_generation",
                     "talkclip_text_generation.text_gen", "emoca_utils", "models", "models.diffusion_prior", "third_party",
                     "third_party.pirender", "third_party.pirender.util", "third_party.pirender.util.meters"]:
            m = MagicMock()
            m.__all__ = []
            sys.modules[name] = m
        spec = importlib.util.spec_from_file_location("ref_train_entry_live", os.path.join(REF, "train_diffusion_prior.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:                       # leave no stub behind for the other tests of this process
        for k in list(sys.modules):
            if k not in keep:
                del sys.modules[k]
        sys.modules.update(keep)


class Bag(torch.nn.Module):
    """A module tree built from dotted parameter names (registration order = the order of the names)."""

    def add(self, dotted, tensor):
        head, _, rest = dotted.partition(".")
        if not rest:
            self.register_parameter(head, torch.nn.Parameter(tensor.clone()))
            return
        if head not in self._modules:
            self.add_module(head, Bag())
        self._modules[head].add(rest, tensor)


def _names_and_small_weights():
    """The trainable parameter names of the prior in ``make_prior_weights`` order, each with a SMALL tensor of its rank."""
    from avi_talking_amd.weights import make_prior_weights
    w = make_prior_weights(3)
    names = [k for k, v in w.items() if v.is_floating_point() and not k.startswith("noise_scheduler")]
    g = torch.Generator().manual_seed(11)
    small = {n: torch.randn(tuple(min(d, 3) for d in w[n].shape) or (1,), generator=g) for n in names}
    return names, small


class CpuStore:
    def __init__(self, names, tensors):
        self.names, self.shape, self.offset = list(names), {n: tuple(tensors[n].shape) for n in names}, {}
        off = 0
        for n in names:
            self.offset[n] = off
            off += tensors[n].numel()
        self.P = torch.cat([tensors[n].reshape(-1) for n in names]).clone()
        self.M, self.V = torch.zeros_like(self.P), torch.zeros_like(self.P)

    def view(self, n, buf=None):
        buf = self.P if buf is None else buf
        return buf[self.offset[n]:self.offset[n] + math.prod(self.shape[n])].view(self.shape[n])


def _trainer(names, tensors, lr=1e-3):
    return SimpleNamespace(store=CpuStore(names, tensors), betas=(0.9, 0.999), eps=1e-8, lr=lr, wd=1e-2, step_count=0,
                           sched={"betas": torch.zeros(3)}, reload_planes=lambda: None)


def _reference_optimizer(model, lr):
    """train_diffusion_prior.py:997-1004, restated: four groups, weight decay 1e-2 except names matching the substrings."""
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    groups = [
        {"params": [p for n, p in model.net.named_parameters() if not any(nd in n for nd in no_decay)], "weight_decay": 1e-2},
        {"params": [p for n, p in model.net.named_parameters() if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
        {"params": [p for n, p in model.voxel2clip.named_parameters() if not any(nd in n for nd in no_decay)], "weight_decay": 1e-2},
        {"params": [p for n, p in model.voxel2clip.named_parameters() if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    return torch.optim.AdamW(groups, lr=lr)


def test_checkpoints_cross_load_with_the_reference_functions(tmp_path):
    from avi_talking_amd.host import checkpoint as CK
    ref = _reference_entry_point()
    names, small = _names_and_small_weights()

    # ---- the reference writes, host/checkpoint.py reads
    model = Bag()
    for n in names:
        model.add(n, small[n])
    # torch yields a module's own parameters before its sub-modules': net.learned_query / null_*_embed(s) before
    # net.to_time_embeds..., every Attention's null_kv before its norm.g - checkpoint.torch_parameter_order restates that rule
    assert [k for k, _ in model.named_parameters()] == CK.torch_parameter_order(names) != names
    opt = _reference_optimizer(model, lr=3e-4)
    assert sum(len(g["params"]) for g in opt.param_groups) == len(names)
    g = torch.Generator().manual_seed(5)
    for step in range(2):
        for p in model.parameters():
            p.grad = torch.randn(p.shape, generator=g)
        opt.step()
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-4, total_steps=50, final_div_factor=1000, pct_start=0.4)
    ref.save_ckpt("last", str(tmp_path / "ref"), 7, model, opt, sched, [1.0, 0.5], [0.7], [3e-4, 2e-4])
    tr = _trainer(names, {n: torch.zeros_like(small[n]) for n in names}, lr=1.0)
    assert CK.resume_ckpt(str(tmp_path / "ref" / "last.pth"), tr, names) == 7
    sd = model.state_dict()
    flat = [n for grp in CK.param_groups(names) for n in grp]
    by_param = {id(p): n for n, p in model.named_parameters()}
    assert [by_param[id(p)] for grp in opt.param_groups for p in grp["params"]] == flat      # same grouping, same order
    for n in names:
        assert torch.equal(tr.store.view(n), sd[n])
        st = opt.state[dict(model.named_parameters())[n]]
        assert torch.equal(tr.store.view(n, tr.store.M), st["exp_avg"]) and torch.equal(tr.store.view(n, tr.store.V), st["exp_avg_sq"])
    assert tr.step_count == 2 and tr.lr == opt.param_groups[0]["lr"]

    # ---- host/checkpoint.py writes, the reference reads
    tr2 = _trainer(names, small, lr=5e-4)
    tr2.step_count = 3
    tr2.store.M.copy_(torch.randn(tr2.store.P.shape, generator=g))
    tr2.store.V.copy_(torch.rand(tr2.store.P.shape, generator=g))
    path = CK.save_ckpt("best", str(tmp_path / "ours"), 4, tr2, names, lr_scheduler={"last_epoch": 9}, losses=[2.0], lrs=[5e-4])
    model2 = Bag()
    for n in names:
        model2.add(n, torch.zeros_like(small[n]))
    for extra in ("betas",):                      # the NoiseScheduler buffers ride along, as in the reference's state_dict
        model2.add_module("noise_scheduler", Bag()) if "noise_scheduler" not in model2._modules else None
        model2._modules["noise_scheduler"].register_buffer(extra, torch.ones(3))
    opt2 = _reference_optimizer(model2, lr=1.0)
    assert ref.resume_ckpt(path, opt2, None, model2) == 4
    params2 = dict(model2.named_parameters())
    for n in names:
        assert torch.equal(params2[n].detach(), tr2.store.view(n))
        st = opt2.state[params2[n]]
        assert float(st["step"]) == 3.0
        assert torch.equal(st["exp_avg"], tr2.store.view(n, tr2.store.M)) and torch.equal(st["exp_avg_sq"], tr2.store.view(n, tr2.store.V))
    assert opt2.param_groups[0]["lr"] == 5e-4 and [g_["weight_decay"] for g_ in opt2.param_groups] == [1e-2, 0.0, 1e-2, 0.0]
    assert torch.equal(model2.noise_scheduler.betas, torch.zeros(3))
