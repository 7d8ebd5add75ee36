"""Input pipeline -- mirrors dataloaders/image_dl.py of the reference (``ImageDataLoader(config)`` with
``train_loader / valid_loader / test_loader``; folder of .png/.jpg; RandomCrop(patch_size) for training,
CenterCrop(test_patch_size) or the full image for validation/test; images smaller than the crop are ``ImageOps.fit``-ed
up first, :87-105; ToTensor range [0,1]).

MI355X-first layout of the same semantics: the host only decodes (PIL) and cuts the crop window as uint8 HWC into a
PINNED staging batch (3 B/pixel); a background thread keeps ``prefetch`` batches ahead; the batch crosses PCIe once as
uint8 on a side stream and ``lldwt_u8hwc_to_f32chw`` turns it into the (B,3,H,W) fp32 tensor on the device
(x / 255 in fp32 == ToTensor, bit-exact).  The reference's per-sample float tensors (12 B/pixel, 4 worker processes,
pinned copies) are never built.

Data-parallel: every rank walks the same seeded permutation and takes its ``rank::world`` slice, truncated so that all
ranks see the same number of batches (collectives in the agent line up).

No image folder configured (``train_data_1`` / ``test_data`` absent or not a directory): ``SyntheticLoader`` -- seeded
uniform RGB crops generated on the device (what bench.py and the GPU tests use; there is no dataset in the image).
"""
import logging
import os
import queue
import threading

import numpy as np
import torch
from PIL import Image, ImageOps

from .. import ops, parallel

_EXT = (".png", ".jpg")


def pil_loader(path):
    with open(path, "rb") as f:            # open as a file to avoid ResourceWarning (dataloaders/image_dl.py:153-158)
        img = Image.open(f)
        return img.convert("RGB")


def _list_images(root):
    roots = [root] if isinstance(root, str) else list(root)
    files = []
    for r in roots:
        files += sorted(os.path.join(r, f) for f in os.listdir(r) if f.endswith(_EXT))
    return files


class ImageDataset:
    """Folder dataset (dataloaders/image_dl.py:63-105).  ``get(i, rng)`` -> uint8 (H,W,3) numpy crop."""

    def __init__(self, root, size, train=True):
        self.size = int(size)
        self.train = bool(train)
        try:
            self.image_files = _list_images(root)
        except OSError:
            logging.getLogger().exception("Dataset could not be found. Drive might be unmounted.", exc_info=False)
            raise
        if not self.image_files:
            raise FileNotFoundError("no .png/.jpg images under %r" % (root,))

    def __len__(self):
        return len(self.image_files)

    def load_fitted(self, i):
        """Decode + enlarge images smaller than the requested crop (ImageOps.fit, :87-99)."""
        img = pil_loader(self.image_files[i])
        w, h = img.size
        s = self.size
        if s > 0 and (w < s or h < s):
            img = ImageOps.fit(img, (max(w, s) if w >= s else s, max(h, s) if h >= s else s))
        return img

    def window(self, w, h, rng):
        """Crop window (top, left, height, width): RandomCrop draws top/left uniformly (torchvision RandomCrop.get_params),
        CenterCrop rounds the centred offset like torchvision (round((h - s) / 2))."""
        s = self.size
        if s == 0:
            return 0, 0, h, w
        if self.train:
            top = int(rng.integers(0, h - s + 1))
            left = int(rng.integers(0, w - s + 1))
        else:
            top, left = int(round((h - s) / 2.0)), int(round((w - s) / 2.0))
        return top, left, s, s

    def get(self, i, rng):
        img = self.load_fitted(i)
        w, h = img.size
        top, left, ch, cw = self.window(w, h, rng)
        return np.array(img.crop((left, top, left + cw, top + ch)), dtype=np.uint8)      # writable copy

    def __getitem__(self, i):
        """Reference-shaped item: float (3,H,W) tensor in [0,1] on the HOST (API parity; the loaders below do not use it)."""
        a = self.get(i, np.random.default_rng())
        return torch.from_numpy(a.copy()).permute(2, 0, 1).float().div(255)


class ImageDataset_test(ImageDataset):
    """dataloaders/image_dl.py:107-150: always CenterCrop (or the full image when size == 0)."""

    def __init__(self, root, size, train=False):
        super().__init__(root, size, train=False)


class DeviceBatchLoader:
    """Iterates (B,3,H,W) fp32 device batches; decode/crop on host threads, uint8 pinned staging, conversion on device."""

    def __init__(self, dataset, batch_size, shuffle, device, seed=0, drop_last=False, prefetch=2, rank=0, world=1):
        self.ds, self.bs, self.shuffle, self.dev = dataset, int(batch_size), bool(shuffle), device
        self.seed, self.drop_last, self.prefetch = int(seed), drop_last, int(prefetch)
        self.rank, self.world = int(rank), int(world)
        self.epoch = 0
        self._copy_stream = None

    def _indices(self):
        n = len(self.ds)
        if self.shuffle:
            order = np.random.default_rng(self.seed + self.epoch).permutation(n)      # same on every rank
        else:
            order = np.arange(n)
        per_rank = n // self.world if self.world > 1 else n
        mine = order[self.rank::self.world][:per_rank] if self.world > 1 else order
        return mine

    def __len__(self):
        n = len(self.ds) // self.world if self.world > 1 else len(self.ds)
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def _produce(self, idx, q, stop):
        rng = np.random.default_rng((self.seed + 1) * 1000003 + self.epoch * 7919 + self.rank)
        try:
            for b0 in range(0, len(idx), self.bs):
                chunk = idx[b0:b0 + self.bs]
                if self.drop_last and len(chunk) < self.bs:
                    break
                crops = [self.ds.get(int(i), rng) for i in chunk]
                h, w = crops[0].shape[:2]
                if any(c.shape[:2] != (h, w) for c in crops):
                    raise ValueError("images of different sizes in one batch: use a crop size or batch_size 1")
                stage = torch.empty(len(crops), h, w, 3, dtype=torch.uint8, pin_memory=self.dev.type == "cuda")
                for k, c in enumerate(crops):
                    stage[k] = torch.from_numpy(c)
                if stop.is_set():
                    return
                q.put(stage)
            q.put(None)
        except BaseException as e:          # surface decode errors in the consumer
            q.put(e)

    def __iter__(self):
        idx = self._indices()
        q = queue.Queue(maxsize=max(1, self.prefetch))
        stop = threading.Event()
        t = threading.Thread(target=self._produce, args=(idx, q, stop), daemon=True)
        t.start()
        self.epoch += 1
        if self._copy_stream is None and self.dev.type == "cuda":
            self._copy_stream = torch.cuda.Stream(device=self.dev)
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                if self.dev.type == "cuda":
                    with torch.cuda.stream(self._copy_stream):
                        d8 = item.to(self.dev, non_blocking=True)          # uint8 over PCIe, side stream
                    torch.cuda.current_stream().wait_stream(self._copy_stream)
                    d8.record_stream(torch.cuda.current_stream())
                    yield ops.u8hwc_to_f32chw(d8)
                else:
                    raise RuntimeError("DeviceBatchLoader needs a GPU device (no CPU fallback for the conversion kernel)")
        finally:
            stop.set()
            while t.is_alive():            # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                t.join(timeout=0.05)


class SyntheticLoader:
    """Seeded uniform RGB crops in [0,1] (ToTensor range) generated on the device; per-rank seed under data parallelism."""

    def __init__(self, n_batches, batch, size, device, seed):
        self.n, self.b, self.s, self.dev, self.seed = n_batches, batch, size, device, seed

    def __iter__(self):
        g = torch.Generator(device=self.dev).manual_seed(self.seed)
        for _ in range(self.n):
            yield torch.rand(self.b, 3, self.s, self.s, device=self.dev, generator=g)

    def __len__(self):
        return self.n


class ImageDataLoader:
    """dataloaders/image_dl.py:13-60.  Like the reference, validation runs on the TEST dataset with test_patch_size and
    batch 1 (:29-31,55-60)."""

    def __init__(self, config, device=None):
        device = device if device is not None else torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        rank, world, _ = parallel.env_rank()
        seed = int(config.get("seed", 0))
        train_dirs = [config.get("train_data_%d" % k) for k in range(1, int(config.get("num_train_dirs", 1)) + 1)]
        train_dirs = [d for d in train_dirs if d]
        test_dir = config.get("test_data")
        have_train = bool(train_dirs) and all(os.path.isdir(d) for d in train_dirs)
        have_test = bool(test_dir) and os.path.isdir(test_dir)
        self.synthetic = not (have_train and have_test)
        if self.synthetic:
            n = int(config.get("synthetic_batches", 2))
            vs = config.get("val_patch_size", config.patch_size) or config.patch_size
            self.train_loader = SyntheticLoader(n, config.batch_size, config.patch_size, device,
                                                parallel.rank_seed(seed, rank))
            self.valid_loader = SyntheticLoader(n, config.get("val_batch_size", 1), vs, device, seed + 1)
            self.test_loader = self.valid_loader
            return
        self.train_dataset = ImageDataset(train_dirs, config.patch_size, train=True)
        self.test_dataset = ImageDataset_test(test_dir, config.get("test_patch_size", 0), train=False)
        self.valid_dataset = self.test_dataset
        self.train_loader = DeviceBatchLoader(self.train_dataset, config.batch_size, True, device, seed=seed,
                                              drop_last=world > 1, rank=rank, world=world)
        # validation / test: every rank walks the whole (small) set, so the reported means need no gather
        self.test_loader = DeviceBatchLoader(self.test_dataset, 1, False, device, seed=seed)
        self.valid_loader = DeviceBatchLoader(self.test_dataset, 1, False, device, seed=seed)
