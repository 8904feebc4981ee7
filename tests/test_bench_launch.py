"""bench.py's launch / rank plumbing, end to end on the CPU (kernel emulator + gloo): `python bench.py --gpus 2` with no
launcher around it must start its own two ranks (torch.distributed.run as a child process, rendezvous on 127.0.0.1), run the
N > 1 branch - barriers, max-over-ranks time, the bucketed gradient exchange of harness.Trainer - and print ONE JSON line
from rank 0.  Timing claims are made on the MI355X only; `--device cpu --config tiny` exists for this test."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=900):
    from dasr_amd import build
    build.build_emu()                         # once, here: the ranks then find it up to date
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.pop("DASR_HIPEMU_LIB", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--device", "cpu", "--config", "tiny", "--steps", "1",
           "--warmup", "0"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), r.stderr


def test_bench_self_launches_two_ranks():
    out, err = _run(["--gpus", "2"])
    assert "launching 2 ranks" in err
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["warmup"] == 0
    assert out["config"]["global_batch"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["scaling"] == "weak" and out["unit"] == "frames/s" and out["value"] > 0
    assert out["loss"] is not None and out["loss"] == out["loss"]
    # (value is printed with three decimals; on a loaded build container the emulated step takes tens of seconds)
    assert abs(out["value"] - 2 * 1 / (out["ms_per_step"] * 1e-3)) <= 1e-2 * out["value"] + 1e-3


def test_bench_single_rank_line_and_infer_mode():
    out, _ = _run(["--gpus", "1", "--mode", "infer"])
    assert out["n_gpus"] == 1 and out["config"]["global_batch"] == 1 and out["config"]["mode"] == "infer"
    assert out["loss"] is None and out["value"] > 0


def test_self_launch_command(monkeypatch):
    """The child command is the driver's own launch line (torch.distributed.run, one node, N ranks, 127.0.0.1)."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    args = bench.parse_args(["--gpus", "4", "--steps", "2"])
    assert bench.self_launch(args) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
