"""Experiment: what a pass costs when the DDPM sampler is NOT tied to its pass - several samples per workgroup on few CUs
(unpaired kernel, samples_per_group = 2 ... 5), each launch allowed to finish during the NEXT pass (two launches in flight on
two streams) - against today's paired sampler that ends with its pass.  The audio branch's graphs replay back to back on
their own streams; sampler launches are enqueued one per pass, alternating between two high-priority streams, with no
dependency between the branches (the head is left out).  Prints, per configuration, the pass period and the sampler's duration.
usage: python scripts/ab_deferred.py [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avi_talking_amd as pkg
pkg.request_hw_queues(8)
from avi_talking_amd import weights as W
from avi_talking_amd.host.pipeline import SamplingPipeline
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
wa, wh, wp = W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3)
B = 32
pcm = bench.synth_audio(B, 160000, 1234).to(dev)
voxel = torch.randn(B, 768, generator=torch.Generator().manual_seed(1235)).to(dev)
noise = torch.randn(101, B, 1, 128, generator=torch.Generator().manual_seed(0)).to(dev)
p = SamplingPipeline(wa, wh, wp, device=dev, rng_seed=4242)
p.capture_pipelined(pcm, voxel, None, arrangements=[(2, True)])
torch.cuda.synchronize()
b = p._pbody
te = torch.randn(B, 1, 128, device=dev)
S = [torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=-1)]


TL = []


def audio_pass():
    """front + chains of the captured body on their streams (what _enqueue_body does, without the sampler's branch)"""
    A = p._s_body
    with torch.cuda.stream(A):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record(A)
        b.g_front.replay()
        ev[1].record(A)
        TL.append(ev)
        b.e_front.record(A)
        for i, C_ in enumerate(b.streams, 1):
            with torch.cuda.stream(C_):
                C_.wait_event(b.e_front)
                b.g_chain[i].replay()
                b.e_chain[i].record(C_)
        b.g_chain[0].replay()
        for i in range(1, b.chains):
            A.wait_event(b.e_chain[i])
        ev[2].record(A)


def run(spg, paired, inflight):
    prior = p.prior
    prior.paired, prior.samples_per_group = paired, spg
    graphs = []
    for j in range(2):                       # one graph per stream: sampler launch alone (time table prebuilt)
        prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": te}, noise=noise)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": te}, noise=noise)
        graphs.append(g)
    evs = []
    for k in range(N + 4):
        if k == 4:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        j = k % inflight
        if spg:
            with torch.cuda.stream(S[j]):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(S[j])
                graphs[j].replay()
                e1.record(S[j])
                evs.append((e0, e1))
        audio_pass()
    p._s_body.synchronize()
    t_audio = (time.perf_counter() - t0) / N * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / N * 1e3
    smp = sorted(a.elapsed_time(b_) for a, b_ in evs[4:]) if evs else [0.0]
    print(f"sampler: {'paired' if paired else 'unpaired'} spg={spg} ({prior.cus_held(B)} CUs per launch, {inflight} stream(s)): "
          f"audio passes every {t_audio:.3f} ms, everything done after {t_all:.3f} ms per pass, sampler launch median "
          f"{smp[len(smp) // 2]:.2f} ms (max {smp[-1]:.2f})", flush=True)


import avi_talking_amd.lib as L
so = L.load()


def run_dummy(blocks, lds, spin, label, stream=None):
    """audio passes beside a kernel that only HOLDS CUs (sleeps): what the sampler costs by occupying, not by what it does"""
    sink = torch.zeros(blocks, dtype=torch.int32, device=dev)
    for k in range(N + 4):
        if k == 4:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        st = stream or S[0]
        with torch.cuda.stream(st):
            L.check(so.avi_debug_where(sink.data_ptr(), blocks, 512, lds, spin, st.cuda_stream), "where")
        audio_pass()
    p._s_body.synchronize()
    t_audio = (time.perf_counter() - t0) / N * 1e3
    torch.cuda.synchronize()
    fr = sorted(e[0].elapsed_time(e[1]) for e in TL[-N:])[N // 2]
    ch = sorted(e[1].elapsed_time(e[2]) for e in TL[-N:])[N // 2]
    print(f"dummy {label}: {blocks} workgroups x {lds // 1024} KB LDS sleeping: audio passes every {t_audio:.3f} ms "
          f"(front {fr:.3f} ms with ~20 launches, chains {ch:.3f} ms with ~2 x 150 launches)", flush=True)


run_dummy(1, 1024, 0, "none (1 workgroup that leaves at once)")
run_dummy(1, 128 * 1024, 5000, "1 CU held ~10 ms")
run_dummy(8, 128 * 1024, 5000, "8 CUs held ~10 ms")
run_dummy(1, 1024, 5000, "1 small workgroup (1 KB LDS) held ~10 ms")
run_dummy(1, 128 * 1024, 2000, "1 CU held ~4 ms (the front's length)")
