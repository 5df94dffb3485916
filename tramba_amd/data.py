"""Dataset reader and augmentation in front of the hot path: the reference's `RGB_Dataset` / `ImageLoader`
(data/dataloader.py:42-133) and its sample transforms (data/custom_transforms.py:22-198), which train.py:288-293 and
test_TSOD.py:48-52 wrap in a `torch.utils.data.DataLoader`.

Host-side image decoding and PIL geometry: there is no kernel here.  What matters for a drop-in is that a run seeded
like the reference's (`random_seed(1026)`, train.py:284) sees the SAME tensors, so the augmentation draws from numpy's
legacy generator in exactly the reference's order:

    scale-crop   random() -> scale,  random() -> apply?
    flip         random() -> left/right?,  random() -> up/down? (drawn even though train mode never flips vertically)
    rotate       randint(lo, hi) -> degrees,  random() -> apply?
    enhance      shuffle(the three enhancers, in place, carried over from sample to sample),
                 then per enhancer random() -> apply?  and, if applied, random() -> factor

`tests/golden/make_golden_data.py` runs the reference's transform classes on seeded synthetic images and
`tests/test_data.py` checks this module against those tensors bit for bit.

Sample layout (same keys as the reference): `image` float32 (3, S, S) normalised with the ImageNet mean / std, `gt`
float32 (1, S, S) in [0, 1], `name` (file stem), `shape` (W, H) of the original mask.
"""
import os
import re

import numpy as np
import torch
from PIL import Image, ImageEnhance, ImageOps
from torch.utils.data import DataLoader, Dataset

Image.MAX_IMAGE_PIXELS = None            # dataloader.py:19: remote-sensing tiles can be huge

IMAGENET_MEAN = (0.485, 0.456, 0.406)    # dataloader.py:32
IMAGENET_STD = (0.229, 0.224, 0.225)
_PIXEL_KEYS = ("image", "gt")


def natural_sorted(paths):
    """File order of the reference (dataloader.py:130-133): digit runs compare as numbers, the rest case-folded."""
    def key(p):
        return [int(tok) if tok.isdigit() else tok.lower() for tok in re.split("([0-9]+)", p)]
    return sorted(paths, key=key)


def _centre_box(outer, inner):
    """Box of size `inner` centred in an image of size `outer`, with the reference's floor divisions."""
    return ((outer[0] - inner[0]) // 2, (outer[1] - inner[1]) // 2,
            (outer[0] + inner[0]) // 2, (outer[1] + inner[1]) // 2)


class Augment:
    """The train-mode chain between the fixed resize and the tensor conversion (dataloader.py:26-30 with the
    parameters given there).  `rng`: anything with numpy's legacy `random / randint / shuffle` (default: the global
    `numpy.random`, which is what `random_seed` seeds); the enhancer order is state, as in the reference, so keep one
    instance per dataset."""

    def __init__(self, rng=None, scale=(0.75, 1.25), degrees=(-10, 10)):
        self.rng = np.random if rng is None else rng
        self.scale = scale
        self.degrees = degrees
        # custom_transforms.py:109-116 fills the list in this fixed order whatever order the caller names them in
        self.enhancers = [ImageEnhance.Contrast, ImageEnhance.Brightness, ImageEnhance.Sharpness]

    def _scale_crop(self, s):                                       # custom_transforms.py:46-66
        factor = self.rng.random() * (self.scale[1] - self.scale[0]) + self.scale[0]
        if not self.rng.random() < 0.5:
            return
        for k in _PIXEL_KEYS:
            if k not in s:
                continue
            w, h = s[k].size
            grown = s[k].resize((int(np.round(w * factor)), int(np.round(h * factor))))   # PIL's default filter
            box = _centre_box(grown.size, (w, h))
            pad = -min(0, box[0], box[1])                           # a shrunk image is padded with black first
            grown = ImageOps.expand(grown, border=pad)
            s[k] = grown.crop(tuple(v + pad for v in box))

    def _flip(self, s):                                             # custom_transforms.py:73-86
        mirror = self.rng.random() < 0.5
        self.rng.random()                                           # the vertical draw is made and discarded
        if mirror:
            for k in _PIXEL_KEYS:
                if k in s:
                    s[k] = s[k].transpose(Image.FLIP_LEFT_RIGHT)

    def _rotate(self, s):                                           # custom_transforms.py:93-107
        deg = int(self.rng.randint(self.degrees[0], self.degrees[1]))
        if deg < 0:
            deg += 360
        if not self.rng.random() < 0.5:
            return
        for k in _PIXEL_KEYS:
            if k in s:
                size = s[k].size
                turned = s[k].rotate(deg, expand=True)
                s[k] = turned.crop(_centre_box(turned.size, size))

    def _enhance(self, s):                                          # custom_transforms.py:118-128
        self.rng.shuffle(self.enhancers)
        for make in self.enhancers:
            if self.rng.random() > 0.5:
                op = make(s["image"])                               # built before the factor is drawn, as there
                s["image"] = op.enhance(float(1 + self.rng.random() / 10))

    def __call__(self, s):
        self._scale_crop(s)
        self._flip(s)
        self._rotate(s)
        self._enhance(s)
        return s


def to_tensors(s):
    """PIL -> normalised tensors (custom_transforms.py:130-198).  The arithmetic keeps the reference's rounding:
    /255 in float32, then mean and std as float64 operands of in-place float32 updates."""
    img = np.array(s["image"], dtype=np.float32)
    img /= 255
    img -= np.asarray(IMAGENET_MEAN)
    img /= np.asarray(IMAGENET_STD)
    s["image"] = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).float()
    if "gt" in s:
        gt = np.array(s["gt"], dtype=np.float32)
        gt /= 255
        s["gt"] = torch.from_numpy(gt).unsqueeze(0)
    return s


class Transform:
    """`get_transform(img_size, mode)` of dataloader.py:22-39: resize to (S, S) (bilinear image, nearest mask), the
    augmentation chain when mode == 'train', tensors.  Any other mode string is the test chain, as in the reference
    (which passes 'Test')."""

    def __init__(self, img_size=384, mode="train", rng=None):
        self.size = (img_size, img_size)
        self.augment = Augment(rng) if mode == "train" else None

    def __call__(self, s):
        s["image"] = s["image"].resize(self.size, Image.BILINEAR)
        if "gt" in s:
            s["gt"] = s["gt"].resize(self.size, Image.NEAREST)
        if self.augment is not None:
            self.augment(s)
        return to_tensors(s)


def get_transform(img_size=384, mode="train", rng=None):
    return Transform(img_size, mode, rng)


_IMAGE_EXT = (".jpg", ".png")


def _listing(folder, ext=_IMAGE_EXT):
    if not os.path.isdir(folder):
        raise FileNotFoundError(f"dataset folder missing: {folder}")
    return natural_sorted([os.path.join(folder, f) for f in os.listdir(folder) if f.lower().endswith(ext)])


def _stem(path):
    return os.path.splitext(os.path.basename(path))[0]


class RGB_Dataset(Dataset):
    """`<root>/<set>/image/*.{jpg,png}` paired by file stem with `<root>/<set>/mask/*` (dataloader.py:42-88).
    Pairs whose image and mask sizes differ are dropped (filter_files); a stem mismatch is an error."""

    def __init__(self, root, sets, img_size, mode, rng=None):
        self.images, self.gts = [], []
        for name in sets:
            imgs = _listing(os.path.join(root, name, "image"))
            gts = _listing(os.path.join(root, name, "mask"))
            if len(imgs) != len(gts):
                raise ValueError(f"{name}: {len(imgs)} images but {len(gts)} masks")
            for ip, gp in zip(imgs, gts):
                if _stem(ip) != _stem(gp):
                    raise ValueError(f"{name}: image {ip} is paired with mask {gp}")
                with Image.open(ip) as a, Image.open(gp) as b:      # header only: no pixel decode
                    same = a.size == b.size
                if same:
                    self.images.append(ip)
                    self.gts.append(gp)
        self.size = len(self.images)
        self.transform = get_transform(img_size, mode, rng)

    def __len__(self):
        return self.size

    def __getitem__(self, index):
        image = Image.open(self.images[index]).convert("RGB")
        gt = Image.open(self.gts[index]).convert("L")
        sample = {"image": image, "gt": gt, "name": _stem(self.images[index]), "shape": gt.size}
        return self.transform(sample)


class ImageLoader:
    """Iterate a folder (or one file) of images without masks (dataloader.py:91-127): `image` comes back with a batch
    axis, `shape` is (H, W) of the original, `original` is the PIL image."""

    def __init__(self, root, img_size=384):
        if os.path.isdir(root):
            self.images = _listing(root, (".jpg", ".png", ".jpeg"))
        elif os.path.isfile(root):
            self.images = [root]
        else:
            raise FileNotFoundError(root)
        self.size = len(self.images)
        self.transform = get_transform(img_size, "Test")

    def __len__(self):
        return self.size

    def __iter__(self):
        for path in self.images:
            image = Image.open(path).convert("RGB")
            s = self.transform({"image": image, "name": _stem(path), "shape": image.size[::-1]})
            s["original"] = image
            s["image"] = s["image"].unsqueeze(0)
            yield s


_RANK_STRIDE = 1000003     # per-rank offset of the augmentation streams (data-parallel ranks are all seeded 1026)


def _seed_worker(worker_id, rank=0):
    """Each DataLoader worker is a fork of the parent and would replay the parent's numpy stream: give every worker
    its own stream derived from torch's per-worker seed (the reference leaves the workers on identical streams), offset
    by the data-parallel rank so that two ranks do not augment their shards with the same draws."""
    np.random.seed((torch.initial_seed() + _RANK_STRIDE * rank) % (1 << 32))


def train_loader(root, img_size=384, batch_size=4, num_workers=8, rank=0, world_size=1, seed=1026, distinct_workers=True):
    """train.py:288-293: Train split, shuffled, pinned.  With world_size > 1 every rank draws a disjoint shard of each
    epoch's permutation (call `loader.sampler.set_epoch(epoch)` per epoch)."""
    ds = RGB_Dataset(root, ["Train"], img_size, "train")
    sampler = None
    if world_size > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(ds, num_replicas=world_size, rank=rank, shuffle=True, seed=seed, drop_last=True)
        if num_workers == 0:     # the samples are drawn in this process: move its numpy stream off the other ranks'
            np.random.seed((seed + _RANK_STRIDE * rank) % (1 << 32))
    import functools
    return DataLoader(ds, batch_size=batch_size, shuffle=sampler is None, sampler=sampler, pin_memory=True,
                      num_workers=num_workers, drop_last=world_size > 1,
                      # workers start at iter() time, by then the process usually holds a HIP context and must not be forked:
                      # on a GPU build they are always spawned, and kept across epochs (a spawn re-imports torch per worker)
                      multiprocessing_context=("spawn" if num_workers > 0 and torch.cuda.device_count() > 0 else None),
                      persistent_workers=num_workers > 0 and torch.cuda.device_count() > 0,
                      worker_init_fn=functools.partial(_seed_worker, rank=rank) if (distinct_workers and num_workers > 0) else None)


def eval_loader(root, img_size=384, num_workers=8):
    """test_TSOD.py:48-52 / train.py:114-115: Test split, batch 1, in file order."""
    return DataLoader(RGB_Dataset(root, ["Test"], img_size, "Test"), batch_size=1, shuffle=False, num_workers=num_workers)


def device_batches(loader, device="cuda"):
    """Adapter from a loader of sample dictionaries to the `batches(epoch)` callable `tramba_amd.train.fit` consumes:
    (images, label) on `device` (train.py:47-50 `data_batch['image']` / `['gt']` moved with `.cuda()`), copies asynchronous from
    the pinned batches, and the epoch handed to a distributed sampler so that every epoch reshuffles."""
    def batches(epoch):
        if hasattr(loader.sampler, "set_epoch"):
            loader.sampler.set_epoch(epoch)
        for b in loader:
            yield b["image"].to(device, non_blocking=True), b["gt"].to(device, non_blocking=True)
    return batches
