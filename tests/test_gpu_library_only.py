"""GPU: the captured passes launch nothing but libavi_talking_hip.so kernels.

Why it matters: the library is built without packed-FP32 instructions because kernels using them were corrupted by
matrix-core kernels of a second stream (avi-talking_amd/build.py); torch's own elementwise / copy / fill kernels are
not built that way, so none of them may run inside the two-stream sampling pass or the training step.  The check
records every ATen operator torch dispatches while a pass is enqueued: only operators that allocate or re-view memory
(no device code) are allowed - every byte of arithmetic and data movement then went through the C ABI."""
import pytest
import torch
from torch.utils._python_dispatch import TorchDispatchMode

pytestmark = pytest.mark.gpu

# operators that launch no kernel: allocation, views, metadata
NO_KERNEL = {
    "aten.empty.memory_format", "aten.empty_like.default", "aten.empty_strided.default", "aten.view.default",
    "aten._unsafe_view.default", "aten.reshape.default", "aten.slice.Tensor", "aten.select.int", "aten.as_strided.default",
    "aten.expand.default", "aten.permute.default", "aten.transpose.int", "aten.t.default", "aten.unsqueeze.default",
    "aten.squeeze.dim", "aten.squeeze.default", "aten.detach.default", "aten.alias.default", "aten.unbind.int",
    "aten.split.Tensor", "aten.view.dtype", "aten.lift_fresh.default", "aten._reshape_alias.default",
}


class Recorder(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.ops.append(str(func))
        return func(*args, **(kwargs or {}))


def kernel_ops(fn):
    rec = Recorder()
    with rec:
        fn()
    torch.cuda.synchronize()
    return sorted({o for o in rec.ops if o not in NO_KERNEL})


def test_sampling_pass_launches_only_library_kernels(gpu):
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline
    pipe = SamplingPipeline(W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3), device=gpu)
    g = torch.Generator().manual_seed(3)
    pcm = (torch.randn(4, 16000, generator=g) * 3000).to(torch.int16).to(gpu)
    voxel, noise = torch.randn(4, 768, generator=g).to(gpu), torch.randn(101, 4, 1, 128, generator=g).to(gpu)
    pipe.run(pcm, voxel, noise)                                  # first call builds cached tables
    bad = kernel_ops(lambda: pipe.run(pcm, voxel, noise))
    assert bad == [], f"torch kernels inside the sampling pass: {bad}"


def test_training_step_launches_only_library_kernels(gpu):
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.training import PriorTrainer
    tr = PriorTrainer(W.make_prior_weights(3), device=gpu, lr=1e-4)
    g = torch.Generator(device=gpu).manual_seed(5)
    voxel = torch.randn(64, 768, device=gpu, generator=g)
    target = torch.randn(64, 1, 128, device=gpu, generator=g) * 0.3
    rand = tr.draw(64, generator=g)
    step = lambda: (tr.forward_backward(voxel, target, rand["times"], rand["noise"], 0.005, rand["brain_keep"],
                                        rand["image_keep"], rand["dropout_masks"]),
                    tr.allreduce_grads(), tr.optimizer_step(use_dyn=True, _in_graph=True))
    step()
    bad = kernel_ops(step)
    assert bad == [], f"torch kernels inside the training step: {bad}"
