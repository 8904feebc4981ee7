"""validation metrics against the reference's own pytorch_ssim (golden), image conversion,
checkpoint interop (reference key names, 'module.' prefixes, training-state resume) and inference-time weight folding.
The net runs on the CPU kernel emulator here; tests/test_gpu_parity.py repeats the net-level checks on the GPU."""
import math
import os

import numpy as np
import pytest
import torch

import dasr_amd  # noqa: F401
from dasr_amd import harness, synth, validate
from tests.emu_fixture import emu  # noqa: F401
from tests.golden_cases import ssim_inputs
from tests import parity_checks as pc


def test_ssim_oracle_matches_reference(golden_dir):
    """The CPU restatement (oracle.ssim_ref) against the reference's own pytorch_ssim values."""
    from oracle import depthnet_oracle as O
    g = np.load(os.path.join(golden_dir, "ssim.npz"))
    for name, (a, b) in ssim_inputs().items():
        assert abs(float(O.ssim_ref(a, b)) - float(g[name + ".mean"])) <= 2e-6, name
        per = O.ssim_ref(a, b, size_average=False).numpy()
        assert np.abs(per - g[name + ".per_image"]).max() <= 2e-6, name
    assert abs(float(g["identical.mean"]) - 1.0) <= 1e-6


def test_ssim_kernel_matches_reference(emu):  # noqa: F811
    print(pc.check_ssim_kernel("cpu"))


def test_tensor2img_and_psnr():
    t = torch.tensor([[[0.0, 0.5, 1.2]], [[0.25, -0.1, 0.999]], [[1.0, 0.002, 0.4980392]]]).repeat(1, 2, 1)   # [3,2,3] RGB
    img = validate.tensor2img(t)
    assert img.dtype == np.uint8 and img.shape == (2, 3, 3) and (img[0] == img[1]).all()
    # BGR order, clamp to [0,1], round-half-even of x*255 (numpy round)
    assert img[0, 0].tolist() == [255, 64, 0]
    assert img[0, 1].tolist() == [1, 0, 128]
    assert img[0, 2].tolist() == [127, 255, 255]
    assert validate.tensor2img(torch.rand(5, 7)).shape == (5, 7)
    with pytest.raises(TypeError):
        validate.tensor2img(torch.rand(2, 3, 4, 5))
    a = np.zeros((4, 4, 3)); b = np.full((4, 4, 3), 5.0)
    assert abs(validate.calculate_psnr(a, b) - 20 * math.log10(255.0 / 5.0)) < 1e-12
    assert validate.calculate_psnr(a, a) == float("inf")
    assert abs(harness.calculate_psnr(torch.from_numpy(a), torch.from_numpy(b)) - validate.calculate_psnr(a, b)) < 1e-12


def test_validation_loop_and_folding(emu):  # noqa: F811
    print(pc.check_validation_and_folding("cpu"))


def test_checkpoint_interop(emu, tmp_path):  # noqa: F811
    print(pc.check_checkpoint_interop("cpu", str(tmp_path)))
