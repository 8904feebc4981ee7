"""dasr_amd — MI355X-native hot path of the depth-aware endoscopy SR generator.

Scope: the ``DepthNet`` generator forward/backward of
CUHK-AIM-Group/Depth-Aware-Endoscopy-SR (reference:
codes/models/modules/sftmd_arch.py:837-950, normalization.py:7-92) behind the
reference's own ``nn.Module`` API, running on hand-written HIP kernels for gfx950
through the C ABI declared in ``include/dasr.h``.

Importing the package does not load the HIP library; the first kernel call does,
and raises if ``libdasr_hip.so`` has not been built (``__graft_entry__.build()``).
"""
__version__ = "0.1.0"
