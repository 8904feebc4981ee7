import os
"""The HIP kernel sources, executed by the CPU kernel emulator (tests/hipemu), against the reference's
golden vectors and the oracle.  These run without a GPU; tests/test_gpu_parity.py repeats them on the
MI355X through libdasr_hip.so."""
import pytest

from tests import parity_checks as pc
from tests.emu_fixture import emu  # noqa: F401
from tests.golden_cases import DEPTHNET_CASES


def test_conv_variants(emu):
    pc.check_conv_variants("cpu")


@pytest.mark.parametrize("pieces", [3, 2], ids=["bf16x3", "fp16x2"])
def test_split_conv(emu, pieces):
    print(pc.check_split_conv("cpu", pieces=pieces))


def test_conv9_split(emu):
    print(pc.check_conv9_split("cpu"))


def test_absmax(emu):
    print(pc.check_absmax("cpu"))


def test_fused_amax(emu):
    print(pc.check_fused_amax("cpu"))


def test_fused_amax_net(emu):
    print(pc.check_fused_amax_net("cpu", "x2_nb4"))        # (the GPU suite runs the x8 case)


def test_prepack_ops(emu):
    print(pc.check_prepack_ops("cpu"))


def test_prepack_net(emu):
    print(pc.check_prepack_net("cpu", "x2_one_depth_block", steps=2))


def test_pixel_shuffle_bit_exact(emu):
    pc.check_pixel_shuffle_bit_exact("cpu")


def test_sean(emu):
    pc.check_sean_golden("cpu")


def test_region_pool(emu):
    pc.check_region_pool("cpu")


def test_blocks(emu):
    pc.check_blocks("cpu")


def test_dgrad_act(emu):
    pc.check_dgrad_act("cpu")


def test_fused_loss(emu):
    pc.check_fused_loss("cpu")


def test_encoder_geometry(emu):
    pc.check_encoder_geometry("cpu")


# x3 / x2 run on the GPU only (tests/test_gpu_parity.py): the emulator needs ~25 s per whole-net case
_EMU_CASES = [c for c in DEPTHNET_CASES if c["name"] in ("x8_nb4", "x4_nb4", "x8_nb5_odd")]


@pytest.mark.parametrize("case", _EMU_CASES, ids=[c["name"] for c in _EMU_CASES])
def test_depthnet(emu, case):
    # gates per case in parity_checks.DEPTHNET_GATES: gradients against the reference's float64 run, <= 10x measured
    print(case["name"], pc.check_depthnet_case(case, "cpu"))


def test_soft_masks_whole_net(emu):
    print(pc.check_soft_masks_whole_net("cpu"))


def test_constant_alpha_and_mask_resize(emu):
    print(pc.check_constant_alpha_and_mask_resize("cpu"))


def test_batch_independence_and_determinism(emu):
    pc.check_batch_independence_and_determinism("cpu")


def test_state_dict_roundtrip():
    pc.check_state_dict_roundtrip("cpu")


def test_reference_assertion(emu):
    pc.check_reference_assertion("cpu")


def test_depth_prep(emu):
    print(pc.check_depth_prep("cpu"))


def test_other_region_counts(emu):
    print(pc.check_other_region_counts("cpu"))


def test_conv_fwd_stats(emu):
    print(pc.check_conv_fwd_stats("cpu"))


def test_train_step_matches_reference(emu):
    print(pc.check_train_step("cpu"))


def test_define_g(emu):
    print(pc.check_define_g("cpu"))


def test_depth_mask_golden(emu):
    print(pc.check_depth_mask_golden("cpu"))


def test_replica_protocol(emu):
    print(pc.check_replica_protocol("cpu"))


def test_region_shortcut_invalidation(emu):
    print(pc.check_region_shortcut_invalidation("cpu"))


# (the 4-wave form with one workgroup per XCD - impl 2 + 16 - runs on the GPU only: the emulator needs ~40 s per parameter)
@pytest.mark.parametrize("impl", [0, 2 + 32, 1], ids=["persistent", "8wave_1wg_per_xcd", "first_kernel"])
def test_bf16_conv_variants(emu, impl):
    print(pc.check_bf16_conv_variants("cpu", impl=impl))


def test_bf16_encoder_s2d(emu):
    print(pc.check_bf16_encoder_s2d("cpu"))


def test_bf16_encoder_paths_agree(emu):
    print(pc.check_bf16_encoder_paths_agree("cpu"))


@pytest.mark.parametrize("name", ["x8_nb4", "x4_nb4"])
def test_bf16_depthnet(emu, name):
    case = [c for c in DEPTHNET_CASES if c["name"] == name][0]
    print(name, pc.check_bf16_depthnet_case(case, "cpu"))
    if os.environ.get("DASR_TEST_FP32_ENCODER"):       # the same check with the encoder on the fp32 gather kernels
        from dasr_amd import graph
        graph.ENCODER_S2D = False
        try:
            print(name, "fp32 encoder", pc.check_bf16_depthnet_case(case, "cpu"))
        finally:
            graph.ENCODER_S2D = True


def test_bf16_ops_vs_fp32_kernels(emu):
    print(pc.check_bf16_ops_vs_fp32_kernels("cpu"))
