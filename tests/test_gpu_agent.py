"""GPU: agent-level plumbing on the HIP path -- JSON -> process_config -> agent -> validate, checkpoint round trip with
identical bits, the folder data pipeline with on-device conversion, and train_step inside an initialised nccl (RCCL)
process group (FlatGradBucket -> all-reduce -> Adam under the real autograd tape)."""
import os

import numpy as np
import pytest
import torch

from test_host_plumbing import _make_images, reference_shaped_json

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _agent(**over):
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.agents.liftingDWT_agent import LiftingBasedDWTAgent
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    return LiftingBasedDWTAgent(make_config(**over))


def test_json_to_agent_validate(tmp_path, monkeypatch):
    """configs[0] plumbing: a JSON with the reference's key set (netType CDF97 + SubbandAutoEncoderBerk, as the shipped
    liftingDWT.json) through get_config_from_json -> process_config -> LiftingBasedDWTAgent -> run() in validate mode."""
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import agents
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import get_config_from_json, process_config
    monkeypatch.chdir(tmp_path)
    config, _ = get_config_from_json(reference_shaped_json(tmp_path, mode="validate"))
    config = process_config(config)
    agent_class = getattr(agents, config.agent)                       # main.py:30 resolves the class by name
    agent = agent_class(config)                                        # no checkpoint yet: logged, not fatal
    assert agent.data_loader.synthetic
    agent.run()
    agent.finalize()
    assert agent.valid_logger.current_epoch == 1
    # the reference's default path: train mode for one tiny epoch writes checkpoint.pth.tar + model_best.pth.tar
    config, _ = get_config_from_json(reference_shaped_json(tmp_path, mode="train", entropy_layer="factorized"))
    config = process_config(config)
    agent = agent_class(config)
    agent.run()
    agent.finalize()
    assert os.path.exists(os.path.join(config.checkpoint_dir, "model_best.pth.tar"))
    assert agent.current_epoch == 1 and agent.current_iteration == 2


def test_checkpoint_save_load_validate_identical_bits(tmp_path):
    ck = str(tmp_path) + "/"
    a = _agent(dwtlevels=2, mode="train", checkpoint_dir=ck, patch_size=64, batch_size=2, val_patch_size=64)
    a.train_one_epoch()                                                # moves the weights away from the seed init
    va = a.validate()
    a.current_epoch = 3
    a.save_checkpoint(is_best=1)
    b = _agent(dwtlevels=2, mode="validate", checkpoint_dir=ck, patch_size=64, batch_size=2, val_patch_size=64, seed=99)
    assert b.current_epoch == 3 and b.current_iteration == a.current_iteration      # loaded in __init__ (mode validate)
    for (k, v), (_, v2) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert torch.equal(v, v2), k
    b.data_loader = a.data_loader
    assert b.validate() == va                                          # bit-identical rate + distortion after reload


def test_u8_conversion_and_device_loader(tmp_path):
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.dataloaders.image_dl import ImageDataLoader
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    a = torch.randint(0, 256, (3, 17, 29, 3), dtype=torch.uint8)
    got = ops.u8hwc_to_f32chw(a.to(DEV))
    assert torch.equal(got.cpu(), a.permute(0, 3, 1, 2).float().div(255))          # ToTensor arithmetic, bit-exact
    arrs = _make_images(str(tmp_path / "tr"), [(64, 48)] * 5, seed=4)
    cfg = make_config(train_data_1=str(tmp_path / "tr"), test_data=str(tmp_path / "tr"), num_train_dirs=1, patch_size=32,
                      batch_size=2, test_patch_size=0, seed=5)
    dl = ImageDataLoader(cfg, torch.device(DEV))
    assert not dl.synthetic
    seen = 0
    for x in dl.train_loader:
        assert x.is_cuda and x.dtype == torch.float32 and x.shape[1:] == (3, 32, 32) and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
        u8 = (x * 255).round().to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
        for img in u8:                                                 # every sample is an exact window of a source image
            assert any(np.array_equal(img, s[t:t + 32, l:l + 32]) for s in arrs for t in range(0, 17) for l in range(0, 33))
        seen += x.shape[0]
    assert seen == 5
    full = [x for x in dl.valid_loader]
    assert len(full) == 5 and full[0].shape == (1, 3, 48, 64)
    assert torch.equal(full[0][0].cpu(), torch.from_numpy(arrs[0]).permute(2, 0, 1).float() / 255)
    # an agent on the folder pipeline: one epoch + validation on full images
    ag = _agent(dwtlevels=2, mode="train", train_data_1=str(tmp_path / "tr"), test_data=str(tmp_path / "tr"),
                num_train_dirs=1, patch_size=32, batch_size=2, test_patch_size=0, entropy_layer="factorized")
    ag.train_one_epoch()
    assert ag.current_iteration == 3 and ag.validate() > 0


def test_train_step_inside_nccl_group():
    """world_size 1 over nccl (= RCCL): the flat bucket really goes through the collective (all_reduce on the device
    buffer), gradients stay aliased under the real autograd tape, Adam moves every parameter."""
    import torch.distributed as dist
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        assert parallel.backend_name() == "nccl"
        ag = _agent(dwtlevels=2, mode="train", patch_size=64, batch_size=2)
        ag.model.train()
        before = [p.detach().clone() for p in ag.model.parameters()]
        x = torch.rand(2, 3, 64, 64, device=DEV)
        l0 = float(ag.train_step(x)[0])
        flat = ag._bucket.flat
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        for p in ag.model.parameters():
            assert lo <= p.grad.data_ptr() < hi                        # AccumulateGrad kept the bucket views
        assert float(flat.abs().sum()) > 0
        moved = sum(int(not torch.equal(a, b)) for a, b in zip(before, ag.model.parameters()))
        assert moved > 100
        for _ in range(5):
            l1 = float(ag.train_step(x)[0])
        assert l1 < l0
        assert parallel.mean_over_ranks([2.5], ag.device) == [2.5]
        # what bench.py's N > 1 training leg does besides the bucket exchange: the agreement flag, the broadcast of every
        # parameter and buffer (all their dtypes must be ones RCCL takes), the max-over-ranks clock
        assert parallel.all_ranks_ok(True, ag.device) is True and parallel.all_ranks_ok(False, ag.device) is False
        snap = [t.detach().clone() for t in list(ag.model.parameters()) + list(ag.model.buffers())]
        parallel.broadcast_parameters(ag.model)
        for a, b in zip(snap, list(ag.model.parameters()) + list(ag.model.buffers())):
            assert torch.equal(a, b)
        assert parallel.max_over_ranks(1.25, ag.device) == 1.25
        parallel.barrier()
    finally:
        dist.destroy_process_group()


def test_iwave_matches_the_reference_fixture():
    """PostProcessingiWave against tests/golden/ref_iwave.npz = the REFERENCE'S OWN module run on CPU
    (tests/golden/make_golden_iwave.py imports graphs/layers/post_processing_networks.py:39-77): output, input gradient and
    four parameter gradients of sum(y^2), same by-name weights."""
    import zlib  # noqa: F401
    from helpers import load_golden
    from oracle import weights as oweights
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.layers.post_processing_networks import \
        PostProcessingiWave
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    z = load_golden("ref_iwave")
    net = PostProcessingiWave(make_config(resnetlevel=2, postprocess="iwave"))
    sd = {k: (oweights.fill_value("iwave." + k, v).to(v.dtype).reshape(v.shape) * 0.25) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    assert abs(sum(float(v.double().abs().sum()) for v in sd.values()) - float(z["wsum"])) < 1e-6 * float(z["wsum"])
    net = net.to(DEV).train()
    x = z["x"].to(DEV).requires_grad_(True)
    y = net(x)
    assert float((y.detach().cpu() - z["y"]).abs().max()) < 2e-5
    (y ** 2).sum().backward()
    assert float((x.grad.cpu() - z["gx"]).abs().max()) < 2e-4 * float(z["gx"].abs().max())
    g = dict(net.named_parameters())
    for key, name in (("g_convFilter_weight", "convFilter.weight"), ("g_res1_conv2_weight", "resNetList.1.resNet.2.weight"),
                      ("g_res0_conv0_bias", "resNetList.0.resNet.0.bias"), ("g_outputConvFilter_weight", "outputConvFilter.weight")):
        r = z[key]
        assert float((g[name].grad.cpu() - r).abs().max()) < 5e-4 * float(r.abs().max()) + 1e-6, name
    net.eval()
    with torch.no_grad():
        assert float((net(z["x"].to(DEV)).cpu() - z["y"]).abs().max()) < 2e-5          # eval path (cached packs)


def test_postprocess_iwave_forward_backward_and_agent_mode(tmp_path):
    """PostProcessingiWave (post_processing_networks.py:54-77) on the conv engine vs the same module's torch maths, its
    gradients vs torch autograd, and the agent's train_postprocess mode (frozen codec, MSE-only training, :113-153)."""
    import torch.nn.functional as F
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.layers.post_processing_networks import \
        PostProcessingiWave
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    cfg = make_config(resnetlevel=2, postprocess="iwave")
    torch.manual_seed(3)
    net = PostProcessingiWave(cfg)
    for blk in net.resNetList:                         # the reference's 0.01 init makes the blocks ~identity: use a livelier one
        for m in (blk.resNet[0], blk.resNet[2]):
            torch.nn.init.normal_(m.weight, std=0.05)
            torch.nn.init.normal_(m.bias, std=0.05)
    assert sorted(k for k in net.state_dict() if "resNetList.1" in k) == [
        "resNetList.1.resNet.0.bias", "resNetList.1.resNet.0.weight", "resNetList.1.resNet.2.bias", "resNetList.1.resNet.2.weight"]
    x = torch.rand(2, 3, 24, 40)

    def torch_forward(m, inp):
        c = lambda conv, t: F.conv2d(t, conv.weight, conv.bias, padding=1)
        t1 = c(m.convFilter, inp)
        t2 = t1
        for blk in m.resNetList:
            t2 = c(blk.resNet[2], F.relu(c(blk.resNet[0], t2))) + t2
        return c(m.outputConvFilter, c(m.interConvFilter, t2) + t1) + inp
    import copy
    ref = copy.deepcopy(net).double()
    xr = x.double().requires_grad_(True)
    yr = torch_forward(ref, xr)
    (yr ** 2).sum().backward()
    net = net.to(DEV).train()
    xg = x.to(DEV).requires_grad_(True)
    yg = net(xg)
    assert float((yg.detach().cpu().double() - yr.detach()).abs().max()) < 2e-5
    (yg ** 2).sum().backward()
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 2e-3 * float(xr.grad.abs().max())
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert float((p.grad.cpu().double() - q.grad).abs().max()) < 2e-3 * float(q.grad.abs().max()) + 1e-6, n
    net.eval()
    with torch.no_grad():
        assert float((net(x.to(DEV)).cpu().double() - yr.detach()).abs().max()) < 2e-5       # eval path (cached packs)
    ag = _agent(dwtlevels=2, mode="train_postprocess", postprocess="iwave", resnetlevel=1, patch_size=32, batch_size=2,
                val_patch_size=32, checkpoint_dir=str(tmp_path) + "/", entropy_layer="factorized")
    before = [p.detach().clone() for p in ag.postprocess.parameters()]
    codec = [p.detach().clone() for p in ag.model.parameters()]
    ag.run()
    assert any(not torch.equal(a, b) for a, b in zip(before, ag.postprocess.parameters()))   # the post-filter trained
    assert all(torch.equal(a, b) for a, b in zip(codec, ag.model.parameters()))              # the codec did not
    ck = torch.load(os.path.join(str(tmp_path), "checkpoint.pth.tar"), weights_only=True)
    assert "state_dict_postprocess" in ck                                                     # agents/base.py:112-124


def test_parameter_arena_matches_plain_tensors(monkeypatch, tmp_path):
    """param_arena: after the first step the stacked per-plane parameters are slices of one arena and their gradients slices of the
    flat bucket (no torch.stack, one accumulation per stack).  Same seed, same batch, same injected noise: the trajectory equals the
    one with plain tensors (LLDWT_PARAM_ARENA=0) up to the order of float atomics; every p.data / p.grad aliases the flat buffers;
    a checkpoint round trip (state_dict -> save -> weights-only load -> load_state_dict) keeps values and aliasing.
    Reference: agents/liftingDWT_agent.py:78-98 (the step), graphs/models/LiftingBasedDWT_net.py:43-62 (one net per plane)."""
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.agents import liftingDWT_agent as la
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(5)).to(DEV)

    def run(use):
        monkeypatch.setattr(la, "_USE_ARENA", use)
        torch.manual_seed(1234)
        ag = _agent(dwtlevels=2, mode="train", patch_size=64, batch_size=2)
        ag.model.train()
        gen = torch.Generator(device=DEV).manual_seed(77)
        noise = lambda t: torch.rand(t.shape, device=t.device, generator=gen) - 0.5
        losses = [float(ag.train_step(x, noise_fn=noise)[0])]
        g1 = {n: p.grad.detach().clone() for n, p in ag.model.named_parameters()}     # gradients of the first step (same weights)
        losses += [float(ag.train_step(x, noise_fn=noise)[0]) for _ in range(2)]
        return ag, losses, g1
    ag1, l1, g1 = run(True)
    ag0, l0, g0 = run(False)
    for a, b in zip(l1, l0):
        assert abs(a - b) < 1e-4 * abs(b), (l1, l0)
    assert list(g1) == list(g0)
    for n in g1:        # the first step fell back to torch.stack; the SECOND and third ran on the arena (their losses are compared above)
        assert float((g1[n] - g0[n]).abs().max()) <= 1e-4 * max(1e-6, float(g0[n].abs().max())), n
    # Adam's first steps are lr * sign(g): an element whose gradient is float noise around zero may move the other way, so the
    # parameters are compared in bulk, not element by element
    num = sum(float((p1 - p0).abs().sum()) for p1, p0 in zip(ag1.model.parameters(), ag0.model.parameters()))
    den = sum(float(p0.abs().sum()) for p0 in ag0.model.parameters())
    assert num < 1e-3 * den, (num, den)
    b = ag1._bucket
    assert b.flat_p is not None and len(b.group_views) > 50 and ag0._bucket.flat_p is None
    lo, hi = b.flat_p.data_ptr(), b.flat_p.data_ptr() + 4 * b.flat_p.numel()
    glo, ghi = b.flat.data_ptr(), b.flat.data_ptr() + 4 * b.flat.numel()
    for p in ag1.model.parameters():
        assert lo <= p.data_ptr() < hi and glo <= p.grad.data_ptr() < ghi
    f = tmp_path / "ck.pth"
    torch.save({"state_dict": ag1.model.state_dict()}, f)
    sd = torch.load(f, weights_only=True)["state_dict"]
    with torch.no_grad():
        for p in ag1.model.parameters():
            p.add_(1.0)
    ag1.model.load_state_dict(sd)
    for n, p in ag1.model.named_parameters():
        assert lo <= p.data_ptr() < hi and torch.equal(p.detach().cpu(), sd[n].cpu()), n
    l_next = float(ag1.train_step(x)[0])
    assert l_next == l_next and len(b.group_views) > 50            # still on the fast path after the reload
