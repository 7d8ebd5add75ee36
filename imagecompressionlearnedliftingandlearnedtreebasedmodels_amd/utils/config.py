"""Config loading -- the small part of utils/config.py:50-103 the hot path needs (easydict is not a dependency here)."""
import json
import os


class Config(dict):
    """Attribute-style dict (stands in for EasyDict, utils/config.py:7,64)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


DEFAULTS = dict(  # hot-path keys of liftingDWT.json:5-44
    agent="LiftingBasedDWTAgent", mode="train", resume_training=False, imshow_validation=False, cuda=True, gpu_device=0,
    seed=1337, clrch=1, netType="LiftingBasedNeuralWaveletv4", entropy_layer="conditioned2ZTsepSubbands",
    autoencoder="SubbandAutoEncoder", dwtlevels=4, num_lifting_perlayer=2, filtersize=5, block_property="same", scale=0,
    linearity_flag=1, depth_scale=2, res_connection_weight=0.1, batch_size=4, patch_size=256, grad_acc_iters=1,
    loss_prnt_iters=3600, learning_rate=1e-4, lambda_=11700, loss_switch_thr=0.0015, training_loss_switch=1,
    max_epoch=1, postprocess="none", checkpoint_file="checkpoint.pth.tar")


def make_config(**over):
    c = Config(DEFAULTS)
    c.update(over)
    return c


def get_config_from_json(json_file):
    with open(json_file, "r") as f:
        d = json.load(f)
    return Config(d), d


def process_config(json_file):
    config, _ = get_config_from_json(json_file)
    exp = config.get("exp_name", "exp")
    config.summary_dir = os.path.join("experiments", exp, "summaries/")
    config.checkpoint_dir = os.path.join("experiments", exp, "checkpoints/")
    config.out_dir = os.path.join("experiments", exp, "out/")
    config.log_dir = os.path.join("experiments", exp, "logs/")
    for d in (config.summary_dir, config.checkpoint_dir, config.out_dir, config.log_dir):
        os.makedirs(d, exist_ok=True)
    return config
