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
    """utils/config.py:50-66: -> (config with attribute access, plain dict)."""
    with open(json_file, "r") as f:
        try:
            d = json.load(f)
        except ValueError:
            print("INVALID JSON file format.. Please provide a good json file")
            raise SystemExit(-1)
    return Config(d), d


def setup_logging(log_dir):
    """utils/config.py:25-47: console + rotating file handlers (debug / error files under log_dir)."""
    import logging
    from logging.handlers import RotatingFileHandler
    fmt = logging.Formatter("[%(levelname)s] - %(asctime)s - %(name)s - : %(message)s in %(pathname)s:%(lineno)d")
    main_logger = logging.getLogger()
    main_logger.setLevel(logging.INFO)
    for h in list(main_logger.handlers):
        if getattr(h, "_lldwt", False):
            main_logger.removeHandler(h)
    console = logging.StreamHandler()
    console.setLevel(logging.INFO)
    console.setFormatter(logging.Formatter("[%(levelname)s]: %(message)s"))
    debug = RotatingFileHandler(os.path.join(log_dir, "exp_debug.log"), maxBytes=10 ** 6, backupCount=5)
    debug.setLevel(logging.DEBUG)
    debug.setFormatter(fmt)
    err = RotatingFileHandler(os.path.join(log_dir, "exp_error.log"), maxBytes=10 ** 6, backupCount=5)
    err.setLevel(logging.WARNING)
    err.setFormatter(fmt)
    for h in (console, debug, err):
        h._lldwt = True
        main_logger.addHandler(h)


def process_config(config):
    """utils/config.py:69-103.  ``config``: the object returned by get_config_from_json (the reference's signature,
    main.py:16,26); a JSON path is accepted too.  exp_name is mandatory; the experiment directories are created and
    logging is set up."""
    if isinstance(config, (str, os.PathLike)):
        config, _ = get_config_from_json(config)
    exp = config.get("exp_name")
    if not exp:
        print("ERROR!!..Please provide the exp_name in json file..")
        raise SystemExit(-1)
    config.summary_dir = os.path.join("experiments", exp, "summaries/")
    config.checkpoint_dir = os.path.join("experiments", exp, "checkpoints/")
    config.out_dir = os.path.join("experiments", exp, "out/")
    config.log_dir = os.path.join("experiments", exp, "logs/")
    for d in (config.summary_dir, config.checkpoint_dir, config.out_dir, config.log_dir):
        os.makedirs(d, exist_ok=True)
    setup_logging(config.log_dir)
    return config
