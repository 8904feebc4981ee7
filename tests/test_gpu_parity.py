"""GPU parity tests: the same checks as tests/test_hip_path_emu.py, through libdasr_hip.so on the MI355X."""
import os

import pytest
import torch

from tests import parity_checks as pc
from tests.golden_cases import DEPTHNET_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def device_lib():
    from dasr_amd import _lib
    os.environ.pop("DASR_HIPEMU_LIB", None)
    _lib.reset_for_tests()
    assert torch.cuda.is_available()
    assert _lib.is_device_build()
    yield


def test_conv_variants():
    pc.check_conv_variants("cuda")


def test_pixel_shuffle_bit_exact():
    pc.check_pixel_shuffle_bit_exact("cuda")


@pytest.mark.parametrize("pieces", [3, 2], ids=["bf16x3", "fp16x2"])
def test_split_conv(pieces):
    print(pc.check_split_conv("cuda", pieces=pieces))


def test_conv9_split():
    print(pc.check_conv9_split("cuda"))


def test_absmax():
    print(pc.check_absmax("cuda"))


def test_fused_amax():
    print(pc.check_fused_amax("cuda"))


def test_fused_amax_net():
    print(pc.check_fused_amax_net("cuda"))


def test_prepack_ops():
    print(pc.check_prepack_ops("cuda"))


def test_prepack_net():
    print(pc.check_prepack_net("cuda"))


def test_ssim_kernel():
    print(pc.check_ssim_kernel("cuda"))


def test_graphed_trainer_matches_eager():
    """harness.Trainer(use_graph=True): three eager steps, one captured, then replays - against the eager trainer on the same
    batches (which change every step: the loader's tensors are copied into the captured ones, region bytes included).
    Training is chaotic at this scale (Adam turns the rounding noise of mathematically-zero gradients into +-lr steps, and
    the weight-gradient slabs are summed with atomics), so the two runs may drift: the loss of every step agrees to 3e-3
    (consecutive steps see different batches and differ by 10-30 %: stale captured inputs would show), and the accumulated
    parameter updates point the same way (cosine >= 0.98 over all parameters with a defined gradient)."""
    from dasr_amd import harness, prep
    case = dict(name="graph", scale=8, which=[0, 1, 2], L=32, nb=5, B=2, H=16, W=20)
    runs = {}
    net0, _ = pc.build_net(case, "cuda")
    init = {k: v.detach().clone() for k, v in net0.state_dict().items()}
    for use_graph in (False, True):
        net, cfg = pc.build_net(case, "cuda")
        tr = harness.Trainer(net, use_graph=use_graph)
        losses = []
        for step in range(7):
            lq, gt, dm, _ = pc.synth.seeded_batch(10 * step, 2, 16, 20, 8)
            lq, gt, dm = lq.cuda(), gt.cuda(), dm.cuda()
            mk = prep.depth_to_masks(dm, 10)
            log = tr.optimize_parameters(lq, gt, dm, mk)
            losses.append(float(log["l_all"]))
        assert (tr._graph is not None) == use_graph
        runs[use_graph] = (losses, {k: v.detach().clone() for k, v in net.state_dict().items()})
    le, lg = runs[False][0], runs[True][0]
    assert all(abs(a - b) <= 3e-3 * max(1.0, abs(a)) for a, b in zip(le, lg)), (le, lg)
    assert max(le) - min(le) > 0.2, le                        # the batches do differ
    dot = na = nb = 0.0
    for k, v0 in init.items():
        if any(z in k for z in pc.ZERO_GRAD_KEYS):
            continue
        ua, ub = (runs[False][1][k] - v0).double().flatten(), (runs[True][1][k] - v0).double().flatten()
        dot += float(ua @ ub); na += float(ua @ ua); nb += float(ub @ ub)
    worst = dot / (na * nb) ** 0.5
    assert worst >= 0.98, worst
    print("graphed vs eager: losses", le, lg, "update cosine", worst)


# x8_nb4 holds a ReLU input within one fp32 rounding of zero (the reference's fp32 and float64 runs and this repo's exact-fp32
# kernels happen to land on the same side; the all-fp32 tree of round 2 did not).  Any change of summation
# order can flip it; the split kernels do, on the MI355X and on the emulator alike: the FORWARD stays inside its gates
# (5e-6), the gradient of the linear functional moves by 5.7e-4 (norm1.alpha_beta by 16 %).  That is a discontinuity of the
# test function, not an accuracy figure (kernel-level accuracy vs float64: check_split_conv), so the case is reported as an
# expected failure instead of widening its gate; the other four cases hold the unchanged fp32 gates.
_SPLIT_CASES = [pytest.param(c, id=c["name"], marks=pytest.mark.xfail(reason="ReLU decision at rounding distance from zero flips",
                                                                      strict=False) if c["name"] == "x8_nb4" else ())
                for c in DEPTHNET_CASES]


@pytest.mark.parametrize("pieces", [3, 2], ids=["bf16x3", "fp16x2"])
@pytest.mark.parametrize("case", _SPLIT_CASES)
def test_depthnet_split_bf16(case, pieces):
    """The whole-net golden cases with the split convolutions FORCED on (by default they take over above
    graph.SPLIT_MIN_PIXELS pixels only, i.e. never at these tiny frames), in both schemes (three bf16 pieces / two scaled fp16
    pieces): the fp32 gates, unchanged."""
    from dasr_amd import graph
    old = graph.SPLIT_MIN_PIXELS, graph.SPLIT_PIECES
    graph.SPLIT_MIN_PIXELS, graph.SPLIT_PIECES = 0, pieces
    try:
        print(case["name"], "split", pieces, pc.check_depthnet_case(case, "cuda"))
    finally:
        graph.SPLIT_MIN_PIXELS, graph.SPLIT_PIECES = old


def test_sean():
    pc.check_sean_golden("cuda")


def test_region_pool():
    pc.check_region_pool("cuda")


def test_blocks():
    pc.check_blocks("cuda")


def test_dgrad_act():
    pc.check_dgrad_act("cuda")


def test_fused_loss():
    pc.check_fused_loss("cuda")


def test_encoder_geometry():
    pc.check_encoder_geometry("cuda")


@pytest.mark.parametrize("case", DEPTHNET_CASES, ids=[c["name"] for c in DEPTHNET_CASES])
def test_depthnet(case):
    print(case["name"], pc.check_depthnet_case(case, "cuda"))


def test_soft_masks_whole_net():
    print(pc.check_soft_masks_whole_net("cuda"))


def test_constant_alpha_and_mask_resize():
    print(pc.check_constant_alpha_and_mask_resize("cuda"))


def test_batch_independence_and_determinism():
    pc.check_batch_independence_and_determinism("cuda")


def test_reference_assertion():
    pc.check_reference_assertion("cuda")


def test_full_size_x8_vs_oracle():
    print(pc.check_full_size_x8("cuda"))


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.gpu
def test_depth_prep():
    print(pc.check_depth_prep("cuda"))


@pytest.mark.gpu
def test_validation_and_folding():
    print(pc.check_validation_and_folding("cuda"))


@pytest.mark.gpu
def test_checkpoint_interop(tmp_path):
    print(pc.check_checkpoint_interop("cuda", str(tmp_path)))


@pytest.mark.gpu
def test_rccl_bucketed_allreduce_single_rank():
    """The N > 1 code path on real device memory: RCCL (backend "nccl") initialises on this box, and the trainer's
    bucketed gradient exchange (tape marks -> pack -> async all_reduce per bucket -> wait -> / world -> unpack) round-trips
    through the HIP DepthNet's own backward.  A one-rank group cannot show the sum, so the trainer is told the world is 2:
    every gradient must come back exactly halved, the buckets must be the four the tape defines, and a c4-shaped step
    (x8, 16 frames, bf16) must not get measurably slower with the exchange on (HIP events; printed)."""
    import os
    import torch.distributed as dist
    from dasr_amd import harness, networks, prep
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = torch.arange(8, dtype=torch.float32, device="cuda")
        dist.all_reduce(t)
        dist.barrier()
        assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
        case = dict(name="rccl", scale=8, which=[0, 1, 2, 3], L=16, nb=6, B=1, H=8, W=12)
        net, cfg = pc.build_net(case, "cuda")
        tr = harness.Trainer(net, group=dist.group.WORLD)
        lq, gt, dm, mk = [x.cuda() for x in pc.synth.seeded_batch(0, 1, 8, 12, 8)]
        sr = net(lq, dm, mk)
        (sr * sr).mean().backward()
        before = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        net.zero_grad(set_to_none=True)
        tr._enable_dp(2)
        sizes = []
        orig = tr._submit_bucket
        tr._submit_bucket = lambda idx, grads: (sizes.append(len(idx)), orig(idx, grads))[1]
        sr = net(lq, dm, mk)
        (sr * sr).mean().backward()
        assert len(sizes) == 3 and all(n > 0 for n in sizes), sizes      # the three tape marks fired inside backward
        tr._finish_allreduce()
        torch.cuda.synchronize()
        assert len(sizes) == 4 and sizes[3] > 0, sizes
        n = 0
        for k, p in net.named_parameters():
            if p.grad is not None:
                # (two backward passes differ in the last bits: the weight-gradient slabs are summed with atomics)
                d = (p.grad - before[k] / 2).abs().max().item()
                assert d <= 1e-4 * before[k].abs().max().item() + 1e-9, (k, d)
                n += 1
        assert n == sum(sizes), (n, sizes)
        del net, tr
        # c4's per-GPU step with and without the exchange
        opt = {"network_G": dict(networks.X8_NETWORK_G), "datasets": {"train": {"depthMaskNum": 10}}}
        net = networks.define_G(opt)
        pc.synth.closed_form_fill_(net.state_dict().items())
        net = net.cuda().set_compute_dtype(torch.bfloat16)
        lq, gt, dm, _ = pc.synth.seeded_batch(0, 16, 128, 160, 8)
        lq, gt, dm = lq.cuda(), gt.cuda(), dm.cuda()
        mk = prep.depth_to_masks(dm, 10)
        ms = {1: [], 2: []}
        for world in (1, 2, 1, 2):               # interleaved: allocator growth and clocks hit both arms alike
            tr = harness.Trainer(net)
            tr.group = dist.group.WORLD
            if world > 1:
                tr._enable_dp(world)
            else:
                tr.world = 1
                object.__setattr__(net, "_grad_bucket_hook", None)
            for _ in range(3):
                tr.optimize_parameters(lq, gt, dm, mk)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                tr.optimize_parameters(lq, gt, dm, mk)
            e1.record()
            torch.cuda.synchronize()
            ms[world].append(e0.elapsed_time(e1) / 5)
            assert bool(torch.isfinite(tr.log["l_all"]))
        print("c4 step: %s ms without, %s ms with the bucketed exchange (one-rank RCCL group, world pretended 2)"
              % (["%.2f" % v for v in ms[1]], ["%.2f" % v for v in ms[2]]))
        assert min(ms[2]) - min(ms[1]) < 1.5, ms          # measured +0.3 ms (tools/time_exchange.py steps)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_large_frame_x2_properties():
    print(pc.check_large_frame_x2("cuda"))


@pytest.mark.gpu
def test_x4_config_shape_properties():
    print(pc.check_x4_config_shape("cuda"))


@pytest.mark.gpu
def test_other_region_counts():
    print(pc.check_other_region_counts("cuda"))


@pytest.mark.gpu
def test_conv_fwd_stats():
    print(pc.check_conv_fwd_stats("cuda"))


@pytest.mark.gpu
def test_train_step_matches_reference():
    print(pc.check_train_step("cuda"))


@pytest.mark.gpu
def test_define_g():
    print(pc.check_define_g("cuda"))


@pytest.mark.gpu
def test_depth_mask_golden():
    print(pc.check_depth_mask_golden("cuda"))


@pytest.mark.gpu
def test_replica_protocol():
    print(pc.check_replica_protocol("cuda"))


@pytest.mark.gpu
def test_data_parallel_replicate():
    print(pc.check_data_parallel_gpu())


@pytest.mark.gpu
def test_ddp_single_rank():
    print(pc.check_ddp_single_rank_gpu())


@pytest.mark.gpu
def test_region_shortcut_invalidation():
    print(pc.check_region_shortcut_invalidation("cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("impl", [0, 16, 32, 2 + 16, 2 + 32, 1],
                         ids=["persistent", "4wave", "8wave", "4wave_1wg_per_xcd", "8wave_1wg_per_xcd", "first_kernel"])
def test_bf16_conv_variants(impl):
    print(pc.check_bf16_conv_variants("cuda", impl=impl))


@pytest.mark.gpu
def test_bf16_encoder_s2d():
    print(pc.check_bf16_encoder_s2d("cuda"))


@pytest.mark.gpu
def test_bf16_encoder_paths_agree():
    print(pc.check_bf16_encoder_paths_agree("cuda"))


@pytest.mark.gpu
def test_bf16_ops_vs_fp32_kernels():
    print(pc.check_bf16_ops_vs_fp32_kernels("cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["x8_nb4", "x4_nb4"])
def test_bf16_depthnet(name):
    case = [c for c in DEPTHNET_CASES if c["name"] == name][0]
    print(name, pc.check_bf16_depthnet_case(case, "cuda"))


@pytest.mark.gpu
def test_bf16_c3_full_frame():
    print(pc.check_bf16_c3_full_frame("cuda"))


@pytest.mark.gpu
def test_bf16_c4_full_frame():
    """BASELINE configs[3]'s per-GPU network (x8, nb=16, L=256, 128x160 LR) on the bf16 path."""
    print(pc.check_bf16_c3_full_frame("cuda", scale=8, H=128, W=160))
