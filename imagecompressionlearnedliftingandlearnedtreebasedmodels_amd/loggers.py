"""Mean-meters for (loss, mse, rate, rate2) -- the part of loggers/rate.py:50-151 the agent calls.

The state_dict schema is the reference's (``{'loss','mse','rate','rate2','it','ep'}``, loggers/rate.py:84-93) so that a
checkpoint written by either code base restores the other's loggers; the round-1 schema of this repo (``{'n','sums'}``)
is still accepted on load.  Values are plain Python floats (loadable with ``torch.load(weights_only=True)``).
"""
import logging
import math
from datetime import datetime


class RateDistortionMeter:
    def __init__(self):
        self.current_iteration = 0
        self.current_epoch = 0
        self.reset()

    def reset(self):
        self.loss, self.mse, self.rate, self.rate2 = [], [], [], []

    def append(self, loss, mse, rate, rate2=0):
        self.current_iteration += 1
        self.loss.append(float(loss))
        self.mse.append(float(mse))
        self.rate.append(float(rate))
        if rate2 > 0:                       # loggers/rate.py:64-65: rate2 is only recorded when positive
            self.rate2.append(float(rate2))

    def mean(self):
        self.current_epoch += 1
        m = lambda v: sum(v) / len(v) if v else 0.0
        out = (m(self.loss), m(self.mse), m(self.rate), m(self.rate2))
        self.reset()
        return out

    def state_dict(self):
        return {"loss": list(self.loss), "mse": list(self.mse), "rate": list(self.rate), "rate2": list(self.rate2),
                "it": self.current_iteration, "ep": self.current_epoch}

    def load_state_dict(self, info):
        if "sums" in info and "n" in info:          # round-1 files of this repo: running sums -> one mean sample
            n = int(info["n"])
            self.reset()
            if n:
                self.append(*[s / n for s in info["sums"]])
            self.current_iteration = n
            return
        as_floats = lambda v: [float(t) for t in v]
        self.loss, self.mse = as_floats(info["loss"]), as_floats(info["mse"])
        self.rate, self.rate2 = as_floats(info["rate"]), as_floats(info["rate2"])
        self.current_iteration = int(info["it"])
        self.current_epoch = int(info["ep"])


class RDLogger(RateDistortionMeter):
    def __init__(self):
        super().__init__()
        self.logger = logging.getLogger("Loss")

    def __call__(self, *args):
        self.append(*args)

    def display(self, lr=0.0, typ="tr"):
        """-> (loss, mse, rate, rate2) means since the last display (loggers/rate.py:106-110)."""
        loss, mse, rate, rate2 = self.mean()
        psnr = 10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf")
        name = {"tr": "  Train Epoch", "te": "   Test Epoch", "va": "  Valid Epoch", "it": "Train Itera"}.get(typ, typ)
        msg = "%s: %3d  RDLoss: %.6f MSE/PSNR: %.6f/%.2f Rate: %.3f+%.3f  (lr: %.6f) (%s)" % (
            name, self.current_epoch, loss, mse, psnr, rate, rate2, lr, datetime.now().strftime("%H:%M:%S"))
        self.logger.info(msg)
        print(msg)
        return loss, mse, rate, rate2
