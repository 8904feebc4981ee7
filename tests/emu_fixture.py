"""Load the CPU kernel-emulator build of csrc/ for unit tests (see tests/hipemu/hipemu.h)."""
import os

import pytest


def use_emulator():
    from dasr_amd import _lib, build
    path = build.build_emu()
    os.environ["DASR_HIPEMU_LIB"] = path
    _lib.reset_for_tests()
    return path


def use_device():
    from dasr_amd import _lib
    os.environ.pop("DASR_HIPEMU_LIB", None)
    _lib.reset_for_tests()


@pytest.fixture
def emu():
    use_emulator()
    yield
    use_device()
