"""Mean-meters for (loss, mse, rate1, rate2) -- the part of loggers/rate.py:50-151 the agent calls."""


class RDLogger:
    def __init__(self):
        self.reset()

    def reset(self):
        self.n = 0
        self.sums = [0.0, 0.0, 0.0, 0.0]

    def __call__(self, loss, mse, rate1, rate2):
        self.n += 1
        for i, v in enumerate((loss, mse, rate1, rate2)):
            self.sums[i] += float(v)

    def display(self, lr=0.0, typ="tr"):
        n = max(self.n, 1)
        loss, mse, r1, r2 = (s / n for s in self.sums)
        print("[%s] n=%d lr=%g loss=%.6f mse=%.6g rate1=%.5f rate2=%.5f" % (typ, self.n, lr, loss, mse, r1, r2))
        self.reset()
        return loss, mse, r1 + r2, None

    def state_dict(self):
        return {"n": self.n, "sums": list(self.sums)}

    def load_state_dict(self, d):
        self.n, self.sums = d["n"], list(d["sums"])
