"""Root conftest: the CPU suite (`pytest tests -m "not gpu"`) spends its time in the CPU kernel emulator, one core per
test; when pytest-xdist is importable and the caller did not choose a worker count, it is spread over six workers
(~35 min serial -> ~7 min on the build container's 8 cores).  GPU runs (`-m gpu`) are never touched: they stay in ONE process.  DASR_TESTS_SERIAL=1
keeps the CPU suite serial too."""
import os

import pytest


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    opt = config.option
    if hasattr(config, "workerinput") or os.environ.get("DASR_TESTS_SERIAL"):
        return None
    if not hasattr(opt, "numprocesses") or opt.numprocesses not in (None, 0) or getattr(opt, "dist", "no") != "no":
        return None                                    # no xdist, or the caller already chose
    if (getattr(opt, "markexpr", "") or "").replace(" ", "") != "notgpu" or getattr(opt, "usepdb", False):
        return None
    if getattr(opt, "collectonly", False):
        return None
    opt.numprocesses = 6
    opt.dist = "load"
    opt.tx = ["popen"] * 6
    return None
