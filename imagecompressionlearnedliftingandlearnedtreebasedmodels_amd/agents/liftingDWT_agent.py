"""LiftingBasedDWTAgent -- per-batch maths, validation loop, optimizer set-up (reference agents/liftingDWT_agent.py).

Same class name, constructor signature and method set as the reference agent (:14-366) so the JSON's
``"agent": "LiftingBasedDWTAgent"`` resolves to it (main.py:30).  ``batch_forward`` is the forward half of
``train_one_epoch`` (:84-96) / ``validate`` (:171-183): RGB->YCbCr, Y-0.5, model, +0.5, YCbCr->RGB, -0.5, forward3.
"""
import os

import torch
from torch import optim

from .. import ops
from ..graphs.losses.rate_dist import TrainDLoss, TrainRDLoss
from .. import autograd as ag
from .. import parallel
from .. import param_arena
from ..graphs.models.LiftingBasedDWT_net import (LiftingBasedDWTNetWrapper, byte_extractor, compress_planes,
                                                  forward_planes, forward_planes_train)
from ..dataloaders.image_dl import ImageDataLoader, SyntheticLoader  # noqa: F401  (SyntheticLoader: re-export)
from ..loggers import RDLogger
from .base import BaseAgent


# LLDWT_PARAM_ARENA=0: parameters and gradients keep their own tensors (torch.stack per step, one gradient add per parameter)
_USE_ARENA = os.environ.get("LLDWT_PARAM_ARENA", "1") != "0"


class LiftingBasedDWTAgent(BaseAgent):
    def __init__(self, config):
        super().__init__(config)
        self.clrch = config.clrch
        self.lr = config.learning_rate
        self.model = LiftingBasedDWTNetWrapper(config).to(self.device)
        self.postprocessflag = config.get("postprocess", "none")
        self.postprocess = None
        self._bucket_postprocess = None
        if config.mode == "train_postprocess":                                   # :26-41
            # built BEFORE the per-rank noise stream is seeded: every rank initialises the same replica from the config seed
            # (and the broadcast makes that independent of how many random numbers the codec's constructor drew)
            from ..graphs.layers.post_processing_networks import make_postprocess
            self.postprocess = make_postprocess(config).to(self.device)
            parallel.broadcast_parameters(self.postprocess)
        self.seed_noise_stream()
        self.optimizer = configure_optimizers(self.model, self.lr)
        if self.postprocess is not None:
            self.optimizer_postprocess = optim.Adam(self.postprocess.parameters(), lr=0.0001)
            self.scheduler_postprocess = optim.lr_scheduler.ReduceLROnPlateau(
                self.optimizer_postprocess, factor=0.5, patience=5, threshold=0.0001, threshold_mode="rel", cooldown=0,
                min_lr=1e-06, eps=1e-08)
        self.scheduler = optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, factor=0.5, patience=5, threshold=0.0001,
                                                              threshold_mode="rel", cooldown=0, min_lr=1e-06, eps=1e-08)
        self.grad_acc_iters = config.grad_acc_iters
        self.loss_prnt_iters = config.loss_prnt_iters
        self.data_loader = ImageDataLoader(config, self.device)       # folder datasets when configured, else synthetic
        self.lambda_ = config.lambda_
        self.loss_switch_thr = config.loss_switch_thr
        self.training_loss_switch = config.training_loss_switch
        self.train_loss = TrainDLoss(config.lambda_) if self.training_loss_switch == 0 else TrainRDLoss(config.lambda_)
        self.valid_loss = TrainRDLoss(config.lambda_)
        self.train_logger, self.trnit_logger = RDLogger(), RDLogger()
        self.aux_logger, self.valid_logger, self.test_logger = RDLogger(), RDLogger(), RDLogger()
        self._bucket = None
        self.imshow_validation = False
        if config.mode in ("test", "validate", "validate_recu_reco") and "checkpoint_dir" in config:
            self.load_checkpoint("model_best.pth.tar")                           # :65-67
        elif config.get("resume_training") and "checkpoint_dir" in config:
            self.load_checkpoint(config.checkpoint_file)                         # :68-69
        self.model_size_estimation()                                             # :73

    def batch_forward(self, x, loss_fn, clamp=False):
        """x (B,3,H,W) RGB in [0,1] -> (loss, mse, rate1, rate2, xhat)."""
        if self.clrch != 1:
            xs = (x - 0.5).contiguous()
            xhat, si_xe, si_xo = self.model(xs)
        else:
            y = ops.rgb_to_ycc(x.contiguous())                                   # :86-87, plane-major
            yhat, si_xe, si_xo = forward_planes(self.model.nets(), y, self.model.training)
            xhat = ops.ycc_to_rgb(yhat, clamp=clamp)                             # :90-94 (+ clamp :181)
            xs = (x - 0.5).contiguous()
        loss, mse, r1, r2 = loss_fn.forward3(xs, xhat, si_xe, si_xo)
        return loss, mse, r1, r2, xhat

    def train_step(self, x, noise_fn=None, allreduce=True):
        """One optimisation step on a batch x (B,3,H,W) in [0,1] (agents/liftingDWT_agent.py:78-98): zero_grad, forward
        (training noise), loss / grad_acc_iters, backward (HIP kernels), gradient all-reduce over ranks, Adam step.
        allreduce=False keeps the step rank-local (no collective: bench.py's pre-flight step, after which the ranks agree
        whether all of them can run the leg at all)."""
        if self._bucket is None:
            self._bucket = parallel.FlatGradBucket(self.model.parameters())     # one flat fp32 bucket for RCCL
        self._bucket.zero_()
        param_arena.set_active(self._bucket if _USE_ARENA else None)            # stacks of per-plane parameters as arena slices
        try:
            return self._train_step(x, noise_fn, allreduce)
        finally:
            param_arena.set_active(None)
            groups = param_arena.take_pending()
            if groups and _USE_ARENA:                                           # first step (or after .to()): learn the layout
                self._bucket.relayout(list(getattr(self._bucket, "_kept_groups", [])) + groups)

    def _train_step(self, x, noise_fn, allreduce):
        xs = (x - 0.5).contiguous()
        if self.clrch == 1:
            y = ops.rgb_to_ycc(x.contiguous())                                 # :86-87
            yhat, si_xe, si_xo = forward_planes_train(self.model.nets(), y, noise_fn)
            xhat = ag.YccToRgbFn.apply(yhat)                                    # :90-94
        else:                                                                   # rgb processed together (:80-84)
            xh, si_xe, si_xo = forward_planes_train(self.model.nets(), xs[None].contiguous(), noise_fn)
            xhat, si_xe, si_xo = xh[0], si_xe[0], [t[0] for t in si_xo]
        loss, mse, r1, r2 = self.train_loss.forward3_train(xs, xhat, si_xe, si_xo)
        (loss / self.grad_acc_iters).float().backward()                        # :97
        if allreduce:
            self._bucket.all_reduce_mean()                                      # data-parallel: mean gradient over ranks
        self.optimizer.step()                                                   # :98
        self._bucket.bump_version()
        self.current_iteration += 1
        return loss, mse, r1, r2

    def train_one_epoch(self):
        """agents/liftingDWT_agent.py:75-111.  Data-parallel: only the gradients are all-reduced inside train_step, so
        every decision that steers training -- the D -> RD loss switch (:104-109) and the ReduceLROnPlateau step (:111)
        -- is taken on the MEAN OVER RANKS of the logged value; with rank-local values the replicas would run different
        losses / learning rates and drift apart.  All ranks see the same number of batches (loader contract)."""
        self.model.train()
        for x in self.data_loader.train_loader:
            x = x.to(self.device)
            loss, mse, r1, r2 = self.train_step(x)
            vals = (loss.item(), mse.item(), r1.item(), r2.item())
            self.train_logger(*vals)
            self.trnit_logger(*vals)
            if (self.current_iteration + 1) % self.loss_prnt_iters == 0:
                _, trnit_mse, _, _ = self.trnit_logger.display(lr=self.optimizer.param_groups[0]["lr"], typ="it")
                trnit_mse, = parallel.mean_over_ranks([trnit_mse], self.device)
                if trnit_mse < self.loss_switch_thr and self.training_loss_switch == 0:     # :104-109
                    self.train_loss = TrainRDLoss(self.lambda_)
                    print("Switching training loss to Rate+lambda*Distortion (it was only lambda*Distortion up to here)")
                    self.training_loss_switch = 1
        train_rd_loss, _, _, _ = self.train_logger.display(lr=self.optimizer.param_groups[0]["lr"], typ="tr")
        train_rd_loss, = parallel.mean_over_ranks([train_rd_loss], self.device)
        self.scheduler.step(train_rd_loss)                                      # :111

    # ---------------------------------------------------------------- mode train_postprocess (:113-153, :203-250)
    def _postprocess_batch(self, x, train):
        """Frozen codec (eval, no grad) -> reconstructed RGB -> post-processing net -> loss terms."""
        with torch.no_grad():
            if self.clrch == 1:
                y = ops.rgb_to_ycc(x.contiguous())
                yhat, si_xe, si_xo = forward_planes(self.model.nets(), y, False)
                xrec = ops.ycc_to_rgb(yhat) + 0.5                              # RGB in [0,1] (:131-132)
            else:
                xh, si_xe, si_xo = self.model((x - 0.5).contiguous())
                xrec = xh + 0.5
        xhat = self.postprocess(xrec) - 0.5                                    # :134, then the -0.5 shift (:137)
        xs = (x - 0.5).contiguous()
        if not train:
            xhat = xhat.clamp(-0.5, 0.5)
            return self.valid_loss.forward3(xs, xhat.contiguous(), si_xe, si_xo)
        return self.train_loss.forward3_train(xs, xhat.contiguous(), si_xe, si_xo)

    def _postprocess_zero_grad(self):
        """Data-parallel like train_step: the post-processing net's gradients live in one flat bucket."""
        if self._bucket_postprocess is None:
            self._bucket_postprocess = parallel.FlatGradBucket(self.postprocess.parameters())
        self._bucket_postprocess.zero_()

    def _postprocess_backward_and_step(self, mse):
        """backward of the MSE (:141), ONE all-reduce (mean over ranks) of the bucket, Adam: the replicas of the
        post-processing net stay identical (rank 0's is the one that is checkpointed and every rank's metrics steer
        scheduler_postprocess through mean_over_ranks)."""
        (mse / self.grad_acc_iters).float().backward()
        self._bucket_postprocess.all_reduce_mean()
        self.optimizer_postprocess.step()

    def train_one_epoch_postprocess(self):
        """agents/liftingDWT_agent.py:113-153: the codec is frozen (eval), only the post-processing net trains, on the MSE."""
        self.model.eval()
        self.postprocess.train()
        for x in self.data_loader.train_loader:
            x = x.to(self.device)
            self._postprocess_zero_grad()
            loss, mse, r1, r2 = self._postprocess_batch(x, True)
            self._postprocess_backward_and_step(mse)
            self.current_iteration += 1
            vals = (loss.item(), mse.item(), r1.item(), r2.item())
            self.train_logger(*vals)
            self.trnit_logger(*vals)
        _, train_mse, _, _ = self.train_logger.display(lr=self.optimizer.param_groups[0]["lr"], typ="tr")
        train_mse, = parallel.mean_over_ranks([train_mse], self.device)
        self.scheduler_postprocess.step(train_mse)                             # :153

    @torch.no_grad()
    def validate_postprocess(self):
        """agents/liftingDWT_agent.py:203-250."""
        self.model.eval()
        self.postprocess.eval()
        psnr, r1s, r2s = [], [], []
        for x in self.data_loader.valid_loader:
            x = x.to(self.device)
            loss, mse, r1, r2 = self._postprocess_batch(x, False)
            self.valid_logger(loss.item(), mse.item(), r1.item(), r2.item())
            psnr.append(10.0 * torch.log10(1.0 / torch.tensor(mse.item())))
            r1s.append(r1.item())
            r2s.append(r2.item())
        valid_rd_loss, _, _, _ = self.valid_logger.display(lr=0.0, typ="va")
        m = lambda v: float(torch.tensor(v).mean()) if v else 0.0
        valid_rd_loss, ps, a1, a2 = parallel.mean_over_ranks([valid_rd_loss, m(psnr), m(r1s), m(r2s)], self.device)
        print(" avg_psnr = %.2f, rate_1 = %g, rate_2 = %g, total_rate = %g" % (ps, a1, a2, a1 + a2))
        return valid_rd_loss

    @torch.no_grad()
    def validate(self):
        """agents/liftingDWT_agent.py:155-201."""
        self.model.eval()
        psnr, r1s, r2s = [], [], []
        for x in self.data_loader.valid_loader:
            x = x.to(self.device)
            loss, mse, r1, r2, _ = self.batch_forward(x, self.valid_loss, clamp=True)
            self.valid_logger(loss.item(), mse.item(), r1.item(), r2.item())
            psnr.append(10.0 * torch.log10(1.0 / torch.tensor(mse.item())))
            r1s.append(r1.item())
            r2s.append(r2.item())
        valid_rd_loss, _, _, _ = self.valid_logger.display(lr=0.0, typ="va")
        m = lambda v: float(torch.tensor(v).mean()) if v else 0.0
        # mean over ranks: is_best / best_valid_loss (agents/base.py:161-164) must agree on every replica
        valid_rd_loss, ps, a1, a2 = parallel.mean_over_ranks([valid_rd_loss, m(psnr), m(r1s), m(r2s)], self.device)
        print(" avg_psnr = %.2f, rate_1 = %g, rate_2 = %g, total_rate = %g" % (ps, a1, a2, a1 + a2))
        return valid_rd_loss

    def model_size_estimation(self, print_params=False):
        """agents/liftingDWT_agent.py:313-366: parameter + buffer bytes of the model (and the post-processing net)."""
        def size(mod):
            ps = sum(p.nelement() * p.element_size() for p in mod.parameters())
            bs = sum(b.nelement() * b.element_size() for b in mod.buffers())
            if print_params:
                for n, t in list(mod.named_parameters()) + list(mod.named_buffers()):
                    print(n, type(t), t.size())
            return ps, bs
        ps, bs = size(self.model)
        if parallel.is_rank0():
            print(" model param+buffer=total size: %.2f+%.2f=%.2fMB" % (ps / 2 ** 20, bs / 2 ** 20, (ps + bs) / 2 ** 20))
        post = getattr(self, "postprocess", None)
        if post is not None:
            pp, pb = size(post)
            if parallel.is_rank0():
                print(" postprocess param+buffer=total size: %.2f+%.2f=%.2fMB" % (pp / 2 ** 20, pb / 2 ** 20, (pp + pb) / 2 ** 20))
            ps, bs = ps + pp, bs + pb
        return ps + bs

    @torch.no_grad()
    def test(self):
        """agents/liftingDWT_agent.py:262-311: real entropy coding of the test set -- compress (range-ANS streams),
        decompress FROM the streams, PSNR of the reconstruction, bits per pixel from the stream lengths."""
        self.model.eval()
        psnr, r_hi, r_lo = [], [], []
        for x in self.data_loader.test_loader:
            x = x.to(self.device)
            if self.clrch != 1:
                xs = (x - 0.5).contiguous()
                xhat, byte_xe, byte_xo = self.model.compress(xs)
            else:
                y = ops.rgb_to_ycc(x.contiguous())                               # :281-283
                yhat, s_xe, s_xo = compress_planes(self.model.nets(), y)
                xhat = ops.ycc_to_rgb(yhat, clamp=True)                          # :286-294
                n = x.shape[0] * x.shape[2] * x.shape[3]
                byte_xe = 8.0 * sum(byte_extractor(r) for r in s_xe) / n
                byte_xo = 8.0 * sum(byte_extractor(r) for lv in s_xo for r in lv) / n
                xs = (x - 0.5).contiguous()
            mse = float(torch.mean((xs - xhat.clamp(-0.5, 0.5)) ** 2))
            psnr.append(10.0 * torch.log10(1.0 / torch.tensor(mse)))
            r_hi.append(byte_xo)
            r_lo.append(byte_xe)
        m = lambda v: float(torch.tensor(v).mean()) if v else 0.0
        ps, hi, lo = parallel.mean_over_ranks([m(psnr), m(r_hi), m(r_lo)], self.device)
        print(" avg_psnr = %.2f, rate_high = %g, rate_low = %g, total_rate = %g" % (ps, hi, lo, hi + lo))
        self.test_result = {"psnr": ps, "rate_high": hi, "rate_low": lo}
        return True


def configure_optimizers(net, lr):
    """Adam over all trainable parameters sorted by name (agents/liftingDWT_agent.py:369-389).  On the device the update runs as
    torch's FUSED multi-tensor Adam (one kernel family over the ~600 parameter tensors instead of ten foreach passes, 1.5 ms per
    step); same state_dict layout (step, exp_avg, exp_avg_sq per parameter), LLDWT_ADAM=foreach keeps the default form."""
    params = dict(net.named_parameters())
    names = sorted(n for n, p in params.items() if p.requires_grad)
    plist = [params[n] for n in names]
    fused = bool(plist) and all(p.is_cuda for p in plist) and os.environ.get("LLDWT_ADAM", "fused") != "foreach"
    return optim.Adam([{"params": plist, "lr": lr}], **({"fused": True} if fused else {}))
