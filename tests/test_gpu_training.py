"""GPU parity of the training step (C4-C6): losses, every parameter gradient and one AdamW update against
torch autograd over the CPU oracle (oracle/prior.py train_loss) with identical random draws."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, seed=71):
    g = torch.Generator().manual_seed(seed)
    voxel = torch.randn(B, 768, generator=g)
    target = torch.randn(B, 1, 128, generator=g) * 0.3
    times = torch.randint(0, 100, (B,), generator=g)
    noise = torch.randn(B, 1, 128, generator=g)
    bk = torch.rand(B, generator=g) < 0.8
    ik = torch.rand(B, generator=g) < 0.8
    masks = [(torch.rand(B, 4096, generator=g) >= 0.5).float() / 0.5] + \
            [(torch.rand(B, 4096, generator=g) >= 0.15).float() / 0.85 for _ in range(4)]
    return voxel, target, times, noise, bk, ik, masks


@pytest.fixture(scope="module")
def setup():
    from avi_talking_amd.weights import make_prior_weights
    from oracle import prior as OP
    B, temp = 64, 0.005
    w = make_prior_weights(3)
    voxel, target, times, noise, bk, ik, masks = _inputs(B)
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    loss, l_nce, l_prior, pred = OP.train_loss(wr, voxel, target, times, noise, temp, bk, ik, masks)
    loss.backward()
    return dict(B=B, temp=temp, w=w, wr=wr, inputs=(voxel, target, times, noise, bk, ik, masks),
                l_nce=l_nce.item(), l_prior=l_prior.item(), pred=pred.detach())


def test_losses_and_gradients(gpu, setup):
    from avi_talking_amd.host.training import PriorTrainer
    voxel, target, times, noise, bk, ik, masks = setup["inputs"]
    tr = PriorTrainer(setup["w"], device=gpu)
    out = tr.forward_backward(voxel.to(gpu), target.to(gpu), times.to(gpu), noise.to(gpu), setup["temp"],
                              bk.to(gpu), ik.to(gpu), [m.to(gpu) for m in masks])
    lp, ln = out["loss_prior"].item(), out["loss_nce"].item()
    print(f"loss_prior {lp:.6f} (oracle {setup['l_prior']:.6f})  loss_nce {ln:.6f} (oracle {setup['l_nce']:.6f})")
    assert abs(lp - setup["l_prior"]) < 1e-4 * max(1, abs(setup["l_prior"]))
    assert abs(ln - setup["l_nce"]) < 1e-4 * max(1, abs(setup["l_nce"]))
    assert (out["pred"].cpu() - setup["pred"].reshape(-1, 128)).abs().max().item() < 1e-3
    worst = (0.0, "")
    for name, ref in setup["wr"].items():
        g = tr.store.grad(name).cpu()
        r = ref.grad if ref.grad is not None else torch.zeros_like(ref)
        scale = r.abs().max().item() + 1e-12
        err = (g - r).abs().max().item() / scale
        if err > worst[0]:
            worst = (err, name)
        assert err < 2e-3, f"{name}: relative grad error {err:.2e} (scale {scale:.2e})"
    print(f"worst relative gradient error {worst[0]:.2e} at {worst[1]}")


def test_adamw_kernel_matches_torch(gpu):
    """avi_adamw on identical (p, g) sequences vs torch.optim.AdamW, 3 steps, with and without weight decay;
    also checks the emitted bf16 hi/lo planes."""
    import avi_talking_amd.lib as L
    n = 4096 * 33
    g0 = torch.Generator().manual_seed(5)
    for wd in (1e-2, 0.0):
        p = torch.randn(n, generator=g0)
        ref = p.clone().requires_grad_(True)
        opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=wd)
        dp, m, v = p.to(gpu), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
        hi, lo = torch.zeros(n, dtype=torch.int16, device=gpu), torch.zeros(n, dtype=torch.int16, device=gpu)
        for step in (1, 2, 3):
            grad = torch.randn(n, generator=g0) * 10 ** torch.randint(-6, 1, (n,), generator=g0).float()
            ref.grad = grad.clone()
            opt.step()
            dg = grad.to(gpu)
            L.check(L.load().avi_adamw(dp.data_ptr(), dg.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999,
                                       1e-8, wd, step, 1.0, None, hi.data_ptr(), lo.data_ptr(), L.stream_ptr()), "adamw")
        err = (dp.cpu() - ref.detach()).abs().max().item()
        assert err < 2e-6, (wd, err)
        recon = hi.view(torch.bfloat16).float() + lo.view(torch.bfloat16).float()
        assert (recon.cpu() - dp.cpu()).abs().max().item() < 2e-5 * dp.abs().max().item()


def test_adamw_follows_one_cycle_lr_and_momentum(gpu):
    """The reference's scheduler (train_diffusion_prior.py:351-357) has torch's default cycle_momentum=True: every step
    changes the rate AND AdamW's beta1.  avi_adamw reading {lr, bias corrections, beta1} from the `dyn` device buffer
    (what a replayed hipGraph sees) against torch.optim.AdamW driven by torch's own OneCycleLR, 12 steps."""
    import math
    import avi_talking_amd.lib as L
    from avi_talking_amd.host.schedule import reference_schedule
    n, epochs, per_epoch = 4096 * 5, 3, 1
    g0 = torch.Generator().manual_seed(7)
    p = torch.randn(n, generator=g0)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=1e-2)
    total = epochs * per_epoch * 5
    tsched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=total, final_div_factor=1000,
                                                 pct_start=2 / epochs)
    mine = reference_schedule(1e-3, epochs, per_epoch)
    dp, m, v = p.to(gpu), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    dyn = torch.zeros(4, device=gpu)
    for step in range(1, total):
        grad = torch.randn(n, generator=g0)
        ref.grad = grad.clone()
        opt.step()
        lr, b1 = mine.lr_at(step - 1), mine.momentum_at(step - 1)
        assert abs(b1 - opt.param_groups[0]["betas"][0]) < 1e-12
        dyn.copy_(torch.tensor([lr, 1 - b1 ** step, 1 / math.sqrt(1 - 0.999 ** step), b1]))
        # scalar arguments deliberately stale (the graph's captured values): dyn overrides them
        L.check(L.load().avi_adamw(dp.data_ptr(), grad.to(gpu).data_ptr(), m.data_ptr(), v.data_ptr(), n, 123.0, 0.9, 0.999,
                                   1e-8, 1e-2, 1, 1.0, dyn.data_ptr(), None, None, L.stream_ptr()), "adamw")
        tsched.step()
    err = (dp.cpu() - ref.detach()).abs().max().item()
    print(f"AdamW under OneCycleLR with cycled beta1, {total - 1} steps: max-abs parameter difference {err:.2e}")
    assert err < 5e-6
    # and with beta1 held at 0.9 the result differs: the cycle is not a no-op
    assert abs(mine.momentum_at(0) - 0.95) < 1e-12 and abs(mine.momentum_at(int(mine.up_end)) - 0.85) < 1e-2


def test_train_step_updates_parameters(gpu, setup):
    from avi_talking_amd.host.training import PriorTrainer, no_decay
    voxel, target, times, noise, bk, ik, masks = setup["inputs"]
    tr = PriorTrainer(setup["w"], device=gpu, lr=1e-4)
    rand = dict(times=times.to(gpu), noise=noise.to(gpu), brain_keep=bk.to(gpu), image_keep=ik.to(gpu),
                dropout_masks=[m.to(gpu) for m in masks])
    # reference optimizer on the oracle's parameters (train_diffusion_prior.py:997-1004)
    named = list(setup["wr"].items())
    groups = [{"params": [p for n, p in named if not no_decay(n)], "weight_decay": 1e-2},
              {"params": [p for n, p in named if no_decay(n)], "weight_decay": 0.0}]
    before = {n: p.detach().clone() for n, p in named}
    opt = torch.optim.AdamW(groups, lr=1e-4)
    opt.step()
    tr.train_step(voxel.to(gpu), target.to(gpu), setup["temp"], rand=rand)
    worst = 0.0
    for n, p in named:
        got = tr.store.view(n).cpu()
        upd_ref, upd = p.detach() - before[n], got - before[n]
        # Adam's first update is ~ -lr*sign(g): compare where the gradient is well above its own error bar
        gref = p.grad if p.grad is not None else torch.zeros_like(p)
        sel = gref.abs() > 0.05 * gref.abs().max()
        if sel.any():
            err = (upd - upd_ref)[sel].abs().max().item() / 1e-4
            worst = max(worst, err)
            assert err < 5e-2, f"{n}: AdamW update differs by {err:.2e} of lr"
    print(f"worst AdamW update error (fraction of lr) {worst:.2e}")
    # a second step runs on the refreshed bf16 planes / transposed packs and keeps the loss finite
    out = tr.train_step(voxel.to(gpu), target.to(gpu), setup["temp"], rand=rand)
    assert torch.isfinite(out["loss_prior"]).all() and torch.isfinite(out["loss_nce"]).all()


def test_checkpoint_round_trip_and_flame_pkl(gpu, tmp_path):
    """{best,last}.pth in the reference's layout (train_diffusion_prior.py:155-168,238-251): resuming reproduces the
    uninterrupted run (up to the float-atomic accumulation order of dnull_kv / drel_bias, ~1e-8), and the optimizer
    part loads into a real torch.optim.AdamW built like :997-1004."""
    import pickle
    from avi_talking_amd.weights import make_prior_weights
    from avi_talking_amd.host.training import PriorTrainer
    from avi_talking_amd.host import checkpoint as CK
    wp = make_prior_weights(3)
    names = [k for k, v in wp.items() if v.is_floating_point() and not k.startswith("noise_scheduler")]
    B = 64
    g = torch.Generator(device=gpu).manual_seed(7)
    voxel = torch.randn(B, 768, device=gpu, generator=g)
    target = torch.randn(B, 1, 128, device=gpu, generator=g) * 0.3

    def run(tr, rands):
        return [tr.train_step(voxel, target, 0.005, rand=r)["loss_prior"].item() for r in rands]

    tr = PriorTrainer(wp, device=gpu, lr=1e-3)
    rands = [tr.draw(B, generator=g) for _ in range(4)]
    run(tr, rands[:2])
    path = CK.save_ckpt("last", str(tmp_path), 3, tr, names, losses=[1.0, 0.9], lrs=[1e-3])
    tail_ref = run(tr, rands[2:])
    ref_params = tr.store.P.clone()

    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "lr_scheduler", "train_losses",
                       "val_losses", "lrs"}
    assert all(n in ck["model_state_dict"] for n in names)
    assert len(ck["optimizer_state_dict"]["param_groups"]) == 4
    # a torch AdamW built the reference's way accepts the optimizer state
    params = {n: torch.nn.Parameter(ck["model_state_dict"][n].clone()) for n in names}
    groups = CK.param_groups(names)
    opt = torch.optim.AdamW([{"params": [params[n] for n in gn], "weight_decay": wd}
                             for gn, wd in zip(groups, (1e-2, 0.0, 1e-2, 0.0))], lr=1e-3)
    opt.load_state_dict(ck["optimizer_state_dict"])
    assert float(opt.state[params[groups[2][0]]]["step"]) == 2.0

    tr2 = PriorTrainer(make_prior_weights(99), device=gpu, lr=5e-4)       # different weights and lr: all overwritten
    assert CK.resume_ckpt(path, tr2, names) == 3
    assert tr2.step_count == 2 and tr2.lr == 1e-3
    tail = run(tr2, rands[2:])
    assert all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(tail, tail_ref)), (tail, tail_ref)
    worst = max(((tr2.store.view(n) - ref_params[tr.store.offset[n]:tr.store.offset[n] + tr.store.view(n).numel()]
                  .view(tr.store.shape[n])).abs().max().item(), n) for n in names)
    assert worst[0] < 1e-6, worst

    fp = CK.save_flame_pkl(str(tmp_path / "flame" / "flame_x.pkl"), torch.zeros(300), torch.ones(5, 50), torch.ones(5, 3))
    with open(fp, "rb") as f:
        d = pickle.load(f)
    assert set(d) == {"shape", "expression", "jaw_pose", "global_pose"} and d["global_pose"].shape == (5, 3)
    assert not d["global_pose"].any()


def test_fused_forward_and_backward_equal_the_launch_chain(gpu, monkeypatch):
    """The one-launch training forward of the denoiser (avi_prior_train_forward), the one-launch dX chain of its backward
    (avi_prior_train_backward) and the launch chains they replace give the same losses, predictions and gradients - every
    element of the flat gradient buffer (all 3-term bf16: differences at rounding level)."""
    from avi_talking_amd.host.training import PriorTrainer
    from avi_talking_amd.weights import make_prior_weights
    w = make_prior_weights(3)
    g = torch.Generator().manual_seed(123)
    B = 64
    voxel, target = torch.randn(B, 768, generator=g).to(gpu), (torch.randn(B, 1, 128, generator=g) * 0.3).to(gpu)
    outs = []
    for fwd, bwd in (("1", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("AVI_TRAIN_FUSED_FWD", fwd)
        monkeypatch.setenv("AVI_TRAIN_FUSED_BWD", bwd)
        tr = PriorTrainer(w, device=gpu)
        assert tr.fused_forward == (fwd == "1") and tr.fused_backward == (bwd == "1")
        rand = tr.draw(B, generator=torch.Generator(device=gpu).manual_seed(9))
        o = tr.forward_backward(voxel, target, rand["times"], rand["noise"], 0.005, rand["brain_keep"], rand["image_keep"],
                                rand["dropout_masks"])
        tr.allreduce_grads()
        outs.append((o["loss_prior"].item(), o["loss_nce"].item(), o["pred"].clone(), tr.store.G.clone(), tr))
    lp0, ln0, p0, g0, tr0 = outs[-1]
    for lp1, ln1, p1, g1, _ in outs[:-1]:
        assert abs(lp1 - lp0) < 1e-4 * abs(lp0) and abs(ln1 - ln0) < 1e-5
        assert (p1 - p0).abs().max().item() < 1e-4
        diff = (g1 - g0).abs()
        worst = int(diff.argmax())
        name = max((n for n in tr0.store.names if tr0.store.offset[n] <= worst), key=lambda n: tr0.store.offset[n])
        print(f"max gradient difference {diff.max().item():.2e} (scale {g0.abs().max().item():.2e}) in {name}")
        assert diff.max().item() < 2e-4 * g0.abs().max().item()
        # small parameters too: relative to their own scale
        for n in ("net.causal_transformer.layers.0.0.null_kv", "net.causal_transformer.layers.3.0.norm.g",
                  "net.causal_transformer.layers.5.1.0.g", "net.causal_transformer.rel_pos_bias.relative_attention_bias.weight",
                  "net.causal_transformer.layers.2.0.to_out.1.g"):
            a_, b_ = tr0.store.view(n, g1), tr0.store.view(n, g0)
            assert (a_ - b_).abs().max().item() < 1e-3 * b_.abs().max().item() + 1e-7, n


def test_gradient_buffer_needs_no_clearing(gpu):
    """The fused path skips the 311 MB clear of the gradient buffer: every element is stored by exactly one launch.  Poison
    the buffer, run the same step twice: both runs must give - bit for bit - the gradients of a run on a zeroed buffer."""
    from avi_talking_amd.host.training import PriorTrainer
    from avi_talking_amd.weights import make_prior_weights
    tr = PriorTrainer(make_prior_weights(3), device=gpu)
    assert tr.fused_backward
    g = torch.Generator().manual_seed(321)
    B = 64
    voxel, target = torch.randn(B, 768, generator=g).to(gpu), (torch.randn(B, 1, 128, generator=g) * 0.3).to(gpu)
    rand = tr.draw(B, generator=torch.Generator(device=gpu).manual_seed(4))
    run = lambda: tr.forward_backward(voxel, target, rand["times"], rand["noise"], 0.005, rand["brain_keep"],
                                      rand["image_keep"], rand["dropout_masks"]) and tr.allreduce_grads()
    tr.store.G.zero_()
    run()
    ref = tr.store.G.clone()
    for poison in (7.0, -3.0e4):
        tr.store.G.fill_(poison)
        run()
        diff = (tr.store.G - ref).abs()
        worst = int(diff.argmax())
        name = max((n for n in tr.store.names if tr.store.offset[n] <= worst), key=lambda n: tr.store.offset[n])
        assert diff.max().item() == 0.0, (name, diff.max().item())


def test_dp_segment_step_equals_single_graph_step(gpu, setup):
    """capture_step_dp (the data-parallel step: 7 hipGraph segments cut at the gradient-bucket announcements, the bucket
    exchanges between them, one fused-AdamW launch per bucket) at world size 1 against capture_step (one graph, one AdamW
    over the whole buffer): same recorded draws, three steps.  The reordering of the backward pass and the per-bucket optimizer
    change no arithmetic, and since the null-kv / relative-bias gradients are summed in sample order (no float atomics:
    AviPriorTrainBwd.attn_part) the step is run-to-run DETERMINISTIC: two runs of one path agree bit for bit, and so do the
    two paths."""
    from avi_talking_amd.host.training import PriorTrainer, grad_spans
    voxel, target, times, noise, bk, ik, masks = setup["inputs"]
    rand = dict(times=times.to(gpu).to(torch.int32), noise=noise.to(gpu), brain_keep=bk.to(gpu), image_keep=ik.to(gpu),
                dropout_masks=[m.to(gpu) for m in masks])
    def run(dp):
        tr = PriorTrainer(setup["w"], device=gpu, lr=1e-3)
        (tr.capture_step_dp if dp else tr.capture_step)(voxel.to(gpu), target.to(gpu), 0.005, rand, warmup=1)
        losses = []
        for k in range(3):
            o = (tr.replay_step_dp if dp else tr.replay_step)(lr=1e-3 * (k + 1), beta1=0.9 + 0.01 * k)
            torch.cuda.synchronize()
            losses.append((float(o["loss_prior"]), float(o["loss_nce"])))
        return tr, losses

    one, l_one = run(False)
    again, l_again = run(False)              # the same path twice: bit-identical
    seg, l_seg = run(True)
    assert len(seg._segs) == len(grad_spans()) == 7 and one.step_count == seg.step_count == 4
    for (a0, a1), (b0, b1) in zip(l_one, l_seg):
        assert abs(a0 - b0) <= 1e-6 * max(1.0, abs(a0)) and abs(a1 - b1) <= 1e-6 * max(1.0, abs(a1))
    for name in ("P", "M", "V", "G"):
        x, y, z = getattr(one.store, name), getattr(seg.store, name), getattr(again.store, name)
        floor = (x - z).abs().max().item()
        err = (x - y).abs().max().item()
        print(f"{name}: segments vs one graph {err:.2e}, one graph vs itself {floor:.2e} (scale {x.abs().max().item():.2e})")
        assert floor == 0.0, (name, floor)                 # deterministic run to run
        assert err == 0.0, (name, err)                     # same launches on the same operands, another order of launches
    recon = seg.store.HI.view(torch.bfloat16).float() + seg.store.LO.view(torch.bfloat16).float()
    assert (recon - seg.store.P).abs().max().item() <= 2e-5 * seg.store.P.abs().max().item()   # planes follow the update

def test_dp_segment_step_with_in_graph_draws(gpu, setup):
    """The segment chain with its random draws inside the first segment: replays advance the stream, a reset reproduces."""
    from avi_talking_amd.host.rng import DeviceRng
    from avi_talking_amd.host.training import PriorTrainer
    voxel, target = setup["inputs"][0].to(gpu), setup["inputs"][1].to(gpu)
    tr = PriorTrainer(setup["w"], device=gpu, lr=1e-4)
    rng = DeviceRng(5, gpu)
    tr.capture_step_dp(voxel, target, 0.005, rng=rng, warmup=1)
    rng.set_state(offset=10)
    l0 = float(tr.replay_step_dp()["loss_prior"])
    l1 = float(tr.replay_step_dp()["loss_prior"])
    assert rng.get_state() == (5, 12) and l0 == l0 and l1 == l1 and l0 != l1


def test_soft_clip_loss_hip_matches_the_reference_entry_point(gpu):
    """avi_soft_clip_loss against the reference's OWN soft_clip_loss (train_diffusion_prior.py:125-133 run on the imported
    entry point, tests/golden/train_helpers.npz): no oracle in between.  B = 64 rows, the reference's temperatures."""
    import os
    import numpy as np
    import avi_talking_amd.lib as L
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_helpers.npz"))
    preds, targs = torch.from_numpy(g["preds"]).to(gpu), torch.from_numpy(g["targs"]).to(gpu)
    B, D = preds.shape
    for t in (0.004, 0.005, 0.0075, 0.125):
        loss = torch.zeros(1, device=gpu)
        dproj = torch.empty_like(preds)
        scratch = torch.empty(2 * B * D + 2 * B + 3 * B * B, dtype=torch.float32, device=gpu)
        L.check(L.load().avi_soft_clip_loss(preds.data_ptr(), targs.data_ptr(), B, D, float(t), 1.0, loss.data_ptr(),
                                            dproj.data_ptr(), scratch.data_ptr(), L.stream_ptr()), "avi_soft_clip_loss")
        ref = float(g[f"soft_clip_loss_T{t}"])
        got = float(loss.item())
        print(f"soft_clip_loss T={t}: HIP {got:.6f} reference {ref:.6f}")
        assert abs(got - ref) <= 2e-5 * max(1.0, abs(ref))
