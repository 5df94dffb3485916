#!/usr/bin/env python3
"""Generate tests/golden/golden_data.npz by IMPORTING the reference's data/custom_transforms.py (numpy + PIL + torch,
nothing stubbed) and chaining its transform classes exactly as data/dataloader.py:22-39 does (that file itself needs
torchvision for `transforms.Compose`, which this image lacks: the chain below is a plain loop over the same objects
with the same arguments).

    python tests/golden/make_golden_data.py               # TRAMBA_REFERENCE=/root/reference

Train mode: numpy seeded with 1026 (train.py:284), DATA_TRAIN_SAMPLES consecutive samples through ONE chain, so the
carried enhancer order and every branch of the augmentation are exercised.  Test mode: one sample per source size.
Run in the build container only; the fixture (tensors only) is what travels."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = os.environ.get("TRAMBA_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import data.custom_transforms as T  # noqa: E402


def chain(mode, s):
    c = [T.static_resize(size=[s, s])]
    if mode == "train":
        c += [T.random_scale_crop(range=[0.75, 1.25]), T.random_flip(lr=True, ud=False),
              T.random_rotate(range=[-10, 10]),
              T.random_image_enhance(methods=["contrast", "sharpness", "brightness"])]
    return c + [T.tonumpy(), T.normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225]), T.totensor()]


def run(c, sample):
    for t in c:
        sample = t(sample)
    return sample


def main():
    out = {}
    np.random.seed(1026)
    c = chain("train", synth.DATA_SIZE)
    for i in range(synth.DATA_TRAIN_SAMPLES):
        w, h = synth.DATA_SOURCES[i % len(synth.DATA_SOURCES)]
        img, gt = synth.image_pair(f"train{i}", w, h)
        s = run(c, {"image": img, "gt": gt, "name": f"train{i}", "shape": gt.size})
        out[f"train{i}_image"], out[f"train{i}_gt"] = s["image"].numpy(), s["gt"].numpy()
    out["rng_after_train"] = np.array([np.random.random()])          # the stream position must agree too
    c = chain("Test", synth.DATA_SIZE)
    for i, (w, h) in enumerate(synth.DATA_SOURCES):
        img, gt = synth.image_pair(f"test{i}", w, h)
        s = run(c, {"image": img, "gt": gt, "name": f"test{i}", "shape": gt.size})
        out[f"test{i}_image"], out[f"test{i}_gt"] = s["image"].numpy(), s["gt"].numpy()
    np.savez_compressed(os.path.join(HERE, "golden_data.npz"), **out)
    print("wrote golden_data.npz:", len(out), "arrays", sum(v.nbytes for v in out.values()), "bytes")


if __name__ == "__main__":
    main()
