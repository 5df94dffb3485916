"""Host-side data path in front of the model (tramba_amd/data.py) against tensors produced by the reference's own
transform classes (tests/golden/make_golden_data.py): bit-exact, including the position of the numpy stream."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import synth  # noqa: E402

from tramba_amd import data as D  # noqa: E402


@pytest.fixture(scope="module")
def golden_data():
    return np.load(os.path.join(HERE, "golden", "golden_data.npz"))


def _train_pass(tf):
    out = []
    for i in range(synth.DATA_TRAIN_SAMPLES):
        w, h = synth.DATA_SOURCES[i % len(synth.DATA_SOURCES)]
        img, gt = synth.image_pair(f"train{i}", w, h)
        out.append(tf({"image": img, "gt": gt, "name": f"train{i}", "shape": gt.size}))
    return out


def test_train_augmentation_is_bit_exact_with_reference(golden_data):
    np.random.seed(1026)                                             # train.py:284 random_seed(1026)
    tf = D.get_transform(synth.DATA_SIZE, "train")
    for i, s in enumerate(_train_pass(tf)):
        assert s["image"].dtype == torch.float32 and tuple(s["image"].shape) == (3, synth.DATA_SIZE, synth.DATA_SIZE)
        assert tuple(s["gt"].shape) == (1, synth.DATA_SIZE, synth.DATA_SIZE)
        assert np.array_equal(s["image"].numpy(), golden_data[f"train{i}_image"]), i
        assert np.array_equal(s["gt"].numpy(), golden_data[f"train{i}_gt"]), i
    assert np.random.random() == golden_data["rng_after_train"][0]   # same number of draws, same order


def test_private_generator_replays_the_global_stream(golden_data):
    tf = D.get_transform(synth.DATA_SIZE, "train", rng=np.random.RandomState(1026))
    for i, s in enumerate(_train_pass(tf)):
        assert np.array_equal(s["image"].numpy(), golden_data[f"train{i}_image"]), i


def test_golden_cases_cover_the_augmentation_branches(golden_data):
    """The fixture would pin nothing if the 12 draws never took a branch: count what a replay of the stream takes."""
    class Spy(D.Augment):
        def __init__(self):
            super().__init__(np.random.RandomState(1026))
            self.hits = dict(scale=0, flip=0, rotate=0, enhance=0)

    spy = Spy()
    for i in range(synth.DATA_TRAIN_SAMPLES):
        w, h = synth.DATA_SOURCES[i % len(synth.DATA_SOURCES)]
        img, gt = synth.image_pair(f"train{i}", w, h)
        s = {"image": img.resize((32, 32)), "gt": gt.resize((32, 32))}
        for name, step in (("scale", spy._scale_crop), ("flip", spy._flip), ("rotate", spy._rotate), ("enhance", spy._enhance)):
            before = (s["image"].tobytes(), s["gt"].tobytes())
            step(s)
            spy.hits[name] += before != (s["image"].tobytes(), s["gt"].tobytes())
    assert all(v >= 2 for v in spy.hits.values()), spy.hits
    assert all(v < synth.DATA_TRAIN_SAMPLES for v in spy.hits.values()), spy.hits


def test_test_mode_is_bit_exact_with_reference(golden_data):
    tf = D.get_transform(synth.DATA_SIZE, "Test")
    state = np.random.get_state()[1].copy()
    for i, (w, h) in enumerate(synth.DATA_SOURCES):
        img, gt = synth.image_pair(f"test{i}", w, h)
        s = tf({"image": img, "gt": gt, "name": f"test{i}", "shape": gt.size})
        assert np.array_equal(s["image"].numpy(), golden_data[f"test{i}_image"])
        assert np.array_equal(s["gt"].numpy(), golden_data[f"test{i}_gt"])
        assert set(np.unique(s["gt"].numpy())) <= {0.0, 1.0}         # nearest resize keeps the mask binary
    assert np.array_equal(np.random.get_state()[1], state)          # the test chain draws nothing


def _write_split(root, split, names, sizes, mask_sizes=None, mask_ext=".png"):
    os.makedirs(os.path.join(root, split, "image"))
    os.makedirs(os.path.join(root, split, "mask"))
    for n, (w, h), ms in zip(names, sizes, mask_sizes or sizes):
        img, _ = synth.image_pair(n, w, h)
        _, gt = synth.image_pair(n, *ms)
        img.save(os.path.join(root, split, "image", n + ".png"))
        gt.save(os.path.join(root, split, "mask", n + mask_ext))


def test_dataset_layout_order_and_filter(tmp_path):
    root = str(tmp_path)
    names = ["P10", "p2", "P1", "p33"]
    _write_split(root, "Test", names, [(40, 30)] * 4, mask_sizes=[(40, 30), (40, 30), (20, 30), (40, 30)])
    ds = D.RGB_Dataset(root, ["Test"], 32, "Test")
    assert [D._stem(p) for p in ds.images] == ["p2", "P10", "p33"]   # natural order; P1's mask has another size
    s = ds[1]
    assert s["name"] == "P10" and tuple(s["shape"]) == (40, 30)
    img, gt = synth.image_pair("P10", 40, 30)
    want = D.get_transform(32, "Test")({"image": img, "gt": gt})
    assert torch.equal(s["image"], want["image"]) and torch.equal(s["gt"], want["gt"])
    batch = next(iter(D.eval_loader(root, 32, num_workers=0)))       # what test_TSOD.py:53-58 unpacks
    assert tuple(batch["image"].shape) == (1, 3, 32, 32) and tuple(batch["gt"].shape) == (1, 1, 32, 32)
    assert batch["name"] == ["p2"] and [int(v) for v in batch["shape"]] == [40, 30]


def test_dataset_rejects_mismatched_pairs(tmp_path):
    root = str(tmp_path)
    _write_split(root, "Train", ["a1", "a2"], [(33, 33)] * 2)
    os.rename(os.path.join(root, "Train", "mask", "a2.png"), os.path.join(root, "Train", "mask", "a3.png"))
    with pytest.raises(ValueError, match="paired"):
        D.RGB_Dataset(root, ["Train"], 32, "train")
    os.remove(os.path.join(root, "Train", "mask", "a3.png"))
    with pytest.raises(ValueError, match="2 images but 1 masks"):
        D.RGB_Dataset(root, ["Train"], 32, "train")
    with pytest.raises(FileNotFoundError):
        D.RGB_Dataset(root, ["Val"], 32, "Test")


def test_train_loader_shards_across_ranks(tmp_path):
    root = str(tmp_path)
    names = [f"t{i}" for i in range(10)]
    _write_split(root, "Train", names, [(36, 36)] * 10)
    seen = []
    for rank in range(2):
        dl = D.train_loader(root, 32, batch_size=2, num_workers=0, rank=rank, world_size=2)
        dl.sampler.set_epoch(3)
        got = [n for b in dl for n in b["name"]]
        assert len(got) == 4 and all(tuple(b["image"].shape) == (2, 3, 32, 32) for b in dl)   # 5 per rank, drop_last
        seen.append(set(got))
    assert not (seen[0] & seen[1])
    single = D.train_loader(root, 32, batch_size=5, num_workers=0)
    assert sorted(n for b in single for n in b["name"]) == sorted(names)


def test_image_loader_iterates_a_folder_or_a_file(tmp_path):
    root = str(tmp_path)
    _write_split(root, "Test", ["x2", "x10"], [(50, 20)] * 2)
    folder = os.path.join(root, "Test", "image")
    got = list(D.ImageLoader(folder, 32))
    assert [s["name"] for s in got] == ["x2", "x10"] and got[0]["shape"] == (20, 50)
    assert tuple(got[0]["image"].shape) == (1, 3, 32, 32) and got[0]["original"].size == (50, 20)
    assert len(D.ImageLoader(os.path.join(folder, "x2.png"), 32)) == 1


def test_device_batches_feed_the_epoch_loop(tmp_path):
    root = str(tmp_path)
    _write_split(root, "Train", [f"t{i}" for i in range(6)], [(36, 36)] * 6)
    dl = D.train_loader(root, 32, batch_size=3, num_workers=0, rank=1, world_size=2)
    batches = D.device_batches(dl, "cpu")
    first = [(x.shape, y.shape) for x, y in batches(0)]
    assert first == [(torch.Size([3, 3, 32, 32]), torch.Size([3, 1, 32, 32]))]
    assert dl.sampler.epoch == 0 and list(batches(4)) and dl.sampler.epoch == 4


def test_worker_processes_draw_distinct_streams(tmp_path):
    """Forked workers inherit the parent's numpy state; `_seed_worker` gives each its own, so two workers do not apply
    the same augmentation to their samples (the same image read by two workers must come back different)."""
    root = str(tmp_path)
    _write_split(root, "Train", ["a", "b"], [(48, 48)] * 2)
    for n in ("a", "b"):                                             # both files hold the same picture
        img, gt = synth.image_pair("same", 48, 48)
        img.save(os.path.join(root, "Train", "image", n + ".png"))
        gt.save(os.path.join(root, "Train", "mask", n + ".png"))
    torch.manual_seed(7)
    dl = D.train_loader(root, 32, batch_size=1, num_workers=2)
    got = {b["name"][0]: b["image"] for b in dl}
    assert set(got) == {"a", "b"} and not torch.equal(got["a"], got["b"])
