"""Epoch loop + checkpoint formats of the reference's fit() (train.py:212-263) -- CPU, tiny module."""
import os

import numpy as np
import torch
import torch.nn as nn

from tramba_amd import train


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.vssm_encoder = nn.Conv2d(3, 4, 3, padding=1)
        self.decoder = nn.Conv2d(4, 1, 1)

    def forward(self, x):
        y = self.decoder(torch.relu(self.vssm_encoder(x)))
        return [nn.functional.avg_pool2d(y, 2), y]


def _data(epoch):
    g = torch.Generator().manual_seed(epoch)
    for _ in range(3):
        yield torch.randn(2, 3, 8, 8, generator=g), (torch.rand(2, 1, 8, 8, generator=g) > 0.5).float()


def test_fit_writes_reference_checkpoint_files_and_resumes(tmp_path):
    torch.manual_seed(0)
    m = Tiny()
    opt = train.get_opt(1e-2, m)
    maes = iter([0.5, 0.4, 0.45, 0.3, 0.35, 0.2])
    hist = train.fit(m, opt, _data, epochs=6, base_lr=1e-2, decay_epochs=[3], decay_factors=[0.1],
                     save_model=str(tmp_path), method="Tramba-V-TSOD", evaluate=lambda mod, e: next(maes), see=2,
                     best_mae=None)
    d = tmp_path / "Tramba-V-TSOD"
    names = sorted(os.listdir(d))
    assert "Tramba-V-TSOD_resume.pth" in names                       # written at epoch 5 (index 4)
    assert any(n.startswith("Tramba-V-TSOD_MAE_0.5_2") for n in names)  # first evaluated epoch: index 1 -> "_2.pth"
    assert [h["lr"] for h in hist] == [1e-2] * 3 + [1e-3] * 3           # utils/lr.py: set at the listed epoch, kept after
    assert opt.param_groups[0]["lr"] == 1e-4                            # encoder group at a tenth
    ck = torch.load(d / "Tramba-V-TSOD_resume.pth")
    assert set(ck) == {"model", "optimizer", "epoch"} and ck["epoch"] == 4
    # resume "last": model + optimizer restored, continues at epoch + 1
    m2 = Tiny()
    opt2 = train.get_opt(1e-2, m2)
    start = train.load_resume("last", str(tmp_path), "Tramba-V-TSOD", m2, opt2)
    assert start == 5
    for k, v in ck["model"].items():
        assert torch.equal(m2.state_dict()[k], v)
    assert opt2.state_dict()["state"].keys() == ck["optimizer"]["state"].keys()
    # resume from a best-MAE file: bare state_dict, epoch parsed from the file name
    best = [n for n in names if "_MAE_" in n][0]
    m3 = Tiny()
    start3 = train.load_resume(str(d / best), str(tmp_path), "Tramba-V-TSOD", m3, train.get_opt(1e-2, m3))
    assert start3 == int(best.split("_")[-1].split(".")[0])
    assert train.load_resume(None, str(tmp_path), "x", m3, opt2) == 0


def test_fit_only_main_rank_writes(tmp_path):
    m = Tiny()
    opt = train.get_opt(1e-2, m)
    train.fit(m, opt, _data, epochs=5, base_lr=1e-2, decay_epochs=[], decay_factors=[], save_model=str(tmp_path),
              method="T", evaluate=lambda mod, e: 0.1, see=0, is_main=False)
    assert not (tmp_path / "T").exists()


def test_bf16_split_is_exact_to_fp32_resolution():
    """_split16: the fp32 validation mode feeds the 16-bit matrix-core kernels hi + lo bf16 pieces (three passes); the
    two pieces must carry the fp32 value to 2^-16 relative, so that hi.hi + hi.lo + lo.hi is an fp32-grade product."""
    from tramba_amd.modules import _split16
    t = torch.randn(4096, generator=torch.Generator().manual_seed(3)) * torch.logspace(-3, 3, 4096)
    hi, lo = _split16(t)
    assert hi.dtype == lo.dtype == torch.bfloat16
    rel = ((hi.double() + lo.double()) - t.double()).abs() / t.double().abs()
    assert float(rel.max()) < 2.0 ** -15
