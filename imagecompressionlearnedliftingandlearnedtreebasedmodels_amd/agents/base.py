"""BaseAgent -- device selection, mode dispatch, epoch loop, checkpoint I/O (reference agents/base.py:13-188).

One process per GPU: the device is ``cuda:LOCAL_RANK`` (the reference is single-GPU, agents/base.py:21,28).
"""
import logging
import os
import shutil

import torch


class BaseAgent:
    def __init__(self, config):
        self.config = config
        self.logger = logging.getLogger("Agent")
        self.best_valid_loss = float("inf")
        self.current_epoch = 0
        self.current_iteration = 0
        if not torch.cuda.is_available():
            raise RuntimeError("the agent needs a GPU: the product path has no CPU fallback (the reference also "
                               "requires one, agents/base.py:27-28)")
        local = int(os.environ.get("LOCAL_RANK", config.get("gpu_device", 0)))
        torch.cuda.set_device(local)
        self.device = torch.device("cuda", local)
        self.cuda = True
        self.manual_seed = config.seed
        self.lr = config.learning_rate
        # the replicated model must be initialised identically on every rank: seed with the config seed here; the agent
        # switches to a per-rank stream for the quantisation noise after the model is built (seed_noise_stream)
        torch.manual_seed(self.manual_seed)
        torch.cuda.manual_seed(self.manual_seed)

    def seed_noise_stream(self):
        rank = int(os.environ.get("RANK", 0))
        torch.manual_seed(self.manual_seed + 7919 * rank)
        torch.cuda.manual_seed(self.manual_seed + 7919 * rank)

    # the four hooks of agents/base.py:30-61
    def train_one_epoch(self):
        raise NotImplementedError

    def validate(self):
        raise NotImplementedError

    def test(self):
        raise NotImplementedError

    LOGGERS = ("train_logger", "trnit_logger", "valid_logger", "test_logger")

    def load_checkpoint(self, file_name):
        """Restores model + counters + loggers, NOT optimizer/scheduler (agents/base.py:63-95, :74-75 commented out).

        The file may come from this code base or from the reference (same key set), i.e. from a third
        party: it is read with the weights-only unpickler (tensors, dicts, lists, numbers -- nothing executable).  The
        model's key set must match exactly; a checkpoint of another architecture is an error, not a silent partial load."""
        filename = os.path.join(self.config.checkpoint_dir, file_name)
        try:
            ckpt = torch.load(filename, map_location=self.device, weights_only=True)
        except OSError:
            self.logger.info("No checkpoint exists from '%s'. Skipping...", self.config.checkpoint_dir)
            self.logger.info("**First time to train**")
            return False
        self.current_epoch = ckpt["epoch"]                    # agents/base.py:70 (the epoch loop restarts AT this epoch)
        self.current_iteration = ckpt["iteration"]
        self.best_valid_loss = ckpt["best_valid_loss"]
        missing, unexpected = self.model.load_state_dict(ckpt["state_dict"], strict=False)
        if missing or unexpected:
            raise RuntimeError("checkpoint %s does not match the model: %d missing keys (e.g. %s), %d unexpected (e.g. %s)"
                               % (filename, len(missing), missing[:3], len(unexpected), unexpected[:3]))
        for name in self.LOGGERS:
            if name in ckpt and hasattr(self, name):
                getattr(self, name).load_state_dict(ckpt[name])
        self.logger.info("Checkpoint loaded successfully from '%s' at (epoch %s) at (iteration %s)",
                         self.config.checkpoint_dir, ckpt["epoch"], ckpt["iteration"])
        return True

    def save_checkpoint(self, file_name="checkpoint.pth.tar", is_best=0, collective=True):
        """agents/base.py:97-128.  Data-parallel: the replicas are identical, so rank 0 alone writes; everyone then
        meets at a barrier so that no rank reads a half-written file.  ``collective=False`` is the emergency save of a
        rank that is about to die (run()'s exception handler): its peers sit in some other collective, so it must not
        enter a barrier -- it writes its own rank-suffixed file (whatever its rank) and returns."""
        from .. import parallel
        if not collective:
            file_name = "%s.rank%d%s" % (file_name[:-len(".pth.tar")], parallel.rank(), ".pth.tar") \
                if file_name.endswith(".pth.tar") else "%s.rank%d" % (file_name, parallel.rank())
        if parallel.is_rank0() or not collective:
            state = {"epoch": self.current_epoch, "iteration": self.current_iteration,
                     "best_valid_loss": float(self.best_valid_loss), "state_dict": self.model.state_dict(),
                     "optimizer": self.optimizer.state_dict(), "scheduler": self.scheduler.state_dict()}
            if getattr(self, "postprocess", None) is not None:
                state["state_dict_postprocess"] = self.postprocess.state_dict()
            for name in self.LOGGERS:
                if hasattr(self, name):
                    state[name] = getattr(self, name).state_dict()
            os.makedirs(self.config.checkpoint_dir, exist_ok=True)
            path = os.path.join(self.config.checkpoint_dir, file_name)
            torch.save(state, path + ".tmp")
            os.replace(path + ".tmp", path)
            if is_best and collective:
                shutil.copyfile(path, os.path.join(self.config.checkpoint_dir, "model_best.pth.tar"))
        if collective:
            parallel.barrier()

    def run(self):
        """Mode dispatch (agents/base.py:130-154): exceptions save a checkpoint and re-raise, Ctrl-C is swallowed."""
        try:
            mode = self.config.mode
            if mode == "test":
                self.test()
            elif mode == "validate":
                self.validate()
            elif mode == "train":
                self.train()
            elif mode == "train_postprocess":
                self.train_postprocess()
            elif mode == "debug":
                with torch.autograd.detect_anomaly():
                    self.train()
            else:
                raise NameError("'" + mode + "' is not a valid training mode.")
        except KeyboardInterrupt:
            self.logger.info("You have entered CTRL+C.. Wait to finalize")
        except AssertionError:
            raise
        except Exception:
            # only a training run has state worth rescuing (a failing test / validate run must not overwrite
            # checkpoint.pth.tar with the weights it loaded and a fresh optimizer); the save is rank-local -- no barrier: the
            # other ranks are inside the gradient all-reduce or a mean_over_ranks, and the re-raise lets the launcher tear
            # them down -- and goes to checkpoint.rank<r>.pth.tar, never over the collective checkpoint
            if self.config.mode in ("train", "train_postprocess", "debug") and getattr(self, "optimizer", None) is not None \
                    and "checkpoint_dir" in self.config:
                try:
                    self.save_checkpoint(collective=False)
                except Exception as e:                         # the original error is the one to report
                    self.logger.error("emergency checkpoint failed: %s", e)
            raise

    def _epoch_loop(self, train_one, validate):
        every = int(self.config.get("validate_every", 1))
        for epoch in range(self.current_epoch, self.config.max_epoch):
            self.current_epoch = epoch
            train_one()
            if not (self.current_epoch + 1) % every:
                valid_loss = validate()              # already the mean over ranks: every replica takes the same branch
                is_best = valid_loss < self.best_valid_loss
                if is_best:
                    self.best_valid_loss = valid_loss
                if "checkpoint_dir" in self.config:
                    self.save_checkpoint(is_best=is_best)
            self.current_epoch += 1

    def train(self):
        """Epoch loop (agents/base.py:156-168)."""
        self._epoch_loop(self.train_one_epoch, self.validate)

    def train_postprocess(self):
        """agents/base.py:170-182."""
        self._epoch_loop(self.train_one_epoch_postprocess, self.validate_postprocess)

    def finalize(self):
        """agents/base.py:184-187: a final checkpoint in the training modes."""
        self.logger.info("Please wait while finalizing the operation.. Thank you")
        if self.config.mode in ("train", "train_postprocess", "debug") and "checkpoint_dir" in self.config \
                and getattr(self, "optimizer", None) is not None:
            self.save_checkpoint()
