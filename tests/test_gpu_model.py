"""GPU parity: the host-side mirror (graphs/models, agents) on the HIP kernels vs the reference golden vectors."""
import pytest
import torch

from helpers import filled, load_golden, maxdiff
from oracle import model, weights

pytestmark = pytest.mark.gpu
PKG = "imagecompressionlearnedliftingandlearnedtreebasedmodels_amd"


def _wrapper(cfg):
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net import \
        LiftingBasedDWTNetWrapper
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import Config
    net = LiftingBasedDWTNetWrapper(Config(cfg))
    sd = filled(weights.wrapper_template(cfg))
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected
    return net.to("cuda:0").eval(), sd


@pytest.mark.parametrize("name", ["ref_wrapper_cond2_L3", "ref_wrapper_ezwt_L3", "ref_wrapper_fact_L2",
                                  "ref_wrapper_cond2_berk_L2", "ref_wrapper_ztblock_L3"])
def test_wrapper_vs_reference(name):
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.layers.lifting_dwt_nets import encode_planes
    g = load_golden(name)
    cfg = g["cfg"]
    net, sd = _wrapper(cfg)
    L = cfg["dwtlevels"]
    dev = "cuda:0"
    nets = net.nets()
    with torch.no_grad():
        # (1) entropy model on the REFERENCE's coefficients: identical rounding, so bits must agree to 1e-4
        oxe = torch.stack([g["p%d_out_xe" % c] for c in range(3)], 0).to(dev).contiguous()
        oxo = [torch.stack([g["p%d_out_xo%d" % (c, i)] for c in range(3)], 0).to(dev).contiguous() for i in range(L)]
        em = [n.entropymodel for n in nets]
        si_xe, si_xo, xe_q, xo_q = type(em[0]).forward_planes(em, oxe, oxo, False)
        for c in range(3):
            assert maxdiff(xe_q[c].cpu(), g["p%d_q_xe" % c]) < 1e-4   # round(x-mu)+mu: mu is a network output
            assert maxdiff(si_xe[c].cpu(), g["p%d_si_xe" % c]) < 5e-4
            for i in range(L):
                assert maxdiff(xo_q[i][c].cpu(), g["p%d_q_xo%d" % (c, i)]) < 1e-4
                d = (si_xo[i][c].cpu() - g["p%d_si_xo%d" % (c, i)]).abs()
                assert float(d.max()) < 5e-4, (c, i, float(d.max()))
        # estimated rate per tensor: relative 1e-4
        tot = sum(float(t.double().sum()) for t in si_xo) + float(si_xe.double().sum())
        ref = sum(float(g["p%d_si_xo%d" % (c, i)].double().sum()) for c in range(3) for i in range(L)) + \
            sum(float(g["p%d_si_xe" % c].double().sum()) for c in range(3))
        assert abs(tot - ref) < 1e-4 * ref
        # (2) encode from pixels: coefficients within 1e-4
        y = g["y"].to(dev)
        y_pm = y.permute(1, 0, 2, 3).unsqueeze(2).contiguous()
        e_xe, e_xo = encode_planes([n.autoencoder for n in nets], y_pm)
        for c in range(3):
            assert maxdiff(e_xe[c].cpu(), g["p%d_out_xe" % c]) < 1e-4
            for i in range(L):
                assert maxdiff(e_xo[i][c].cpu(), g["p%d_out_xo%d" % (c, i)]) < 1e-4
        # (3) full forward through the reference-shaped API
        yhat, s_xe, s_xo = net(y)
        assert len(s_xo) == 3 * L and s_xo[L].shape == g["p1_si_xo0"].shape
        flips = sum(int((torch.round(e_xo[i][c].cpu()) != torch.round(g["p%d_out_xo%d" % (c, i)])).sum())
                    for c in range(3) for i in range(L))
        if flips == 0:
            assert maxdiff(yhat.cpu(), g["yhat"]) < 2e-4
            bpp = (sum(float(t.double().sum()) for t in s_xo) + float(s_xe.double().sum())) / g["x"].numel() * 3
            assert abs(bpp - (float(g["rate1"]) + float(g["rate2"]))) < 1e-4 * max(1.0, bpp)


def test_agent_batch_bpp_psnr_match():
    """Same weights + same input => same bpp / PSNR as the reference CPU path (SURVEY 6, 8d parity gates)."""
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.agents.liftingDWT_agent import LiftingBasedDWTAgent
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    g = load_golden("ref_wrapper_cond2_L3")
    cfg = make_config(**g["cfg"])
    cfg.mode = "validate"
    agent = LiftingBasedDWTAgent(cfg)
    agent.model.load_state_dict(filled(weights.wrapper_template(g["cfg"])), strict=False)
    agent.model.eval()
    with torch.no_grad():
        loss, mse, r1, r2, xhat = agent.batch_forward(g["x"].to(agent.device), agent.valid_loss, clamp=False)
    assert maxdiff(xhat.cpu(), g["xhat"]) < 2e-4
    assert abs(float(r1) - float(g["rate1"])) < 1e-4 and abs(float(r2) - float(g["rate2"])) < 1e-4 * max(1, float(g["rate2"]))
    psnr = 10 * torch.log10(1.0 / mse.cpu())
    psnr_ref = 10 * torch.log10(1.0 / torch.tensor(float(g["mse"])))
    assert abs(float(psnr) - float(psnr_ref)) < 0.01
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    # validate() runs the synthetic loader end to end
    assert agent.validate() > 0


def test_single_plane_module_api():
    """Per-module entry points (P=1): P_block_v2.forward, one_level_lifting, encode/decode."""
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.layers.lifting_dwt_nets import \
        LiftingBasedNeuralWaveletv4
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import Config
    g = load_golden("ref_lifting_L2_k5")
    cfg = g["cfg"]
    net = LiftingBasedNeuralWaveletv4(Config(cfg))
    net.load_state_dict(filled(weights.autoencoder_template(cfg)), strict=False)
    net = net.to("cuda:0").eval()
    x = g["x"].to("cuda:0")
    with torch.no_grad():
        LL, LH, HL, HH = net.waveletForward[0].one_level_lifting(x)
        for a, n in ((LL, "LL"), (LH, "LH"), (HL, "HL"), (HH, "HH")):
            assert maxdiff(a.cpu(), g[n]) < 1e-4, n
        rec = net.waveletInverse[0].one_level_lifting(LL, LH.contiguous(), HL.contiguous(), HH.contiguous())
        assert maxdiff(rec.cpu(), g["rec1"]) < 1e-4
        out_xe, out_xo = net.encode(x)
        assert maxdiff(out_xe.cpu(), g["out_xe"]) < 1e-4
        for i, t in enumerate(out_xo):
            assert maxdiff(t.cpu(), g["out_xo%d" % i]) < 1e-4
        assert maxdiff(net.decode(out_xe, out_xo).cpu(), g["xr"]) < 2e-4
        pb = load_golden("ref_pblock_k5")
        from oracle import lifting as olift
        y = net.P_blocks[0](pb["x"].to("cuda:0"))
        assert maxdiff(y.cpu(), olift.p_block(pb["x"], filled(weights.autoencoder_template(cfg)), "P_blocks.0.")) < 1e-4
