"""Data-parallel back ends behind the reference's Accelerator contract (accelerators/accelerator.py:15-32,
accelerators/__init__.py:12-15)."""
from .accelerator import Accelerator
from .rccl_ddp_accelerator import RCCLDDPAccelerator

ACCELERATOR_MAP = {"RCCLDDP": RCCLDDPAccelerator, "DDP": RCCLDDPAccelerator, "ApexDDP": RCCLDDPAccelerator,
                   "TorchAMPDDP": RCCLDDPAccelerator}

# ApexDDP / TorchAMPDDP / DDP of the reference (accelerators/__init__.py:12-15) all map to the one bf16 data-parallel back end:
# FP16_OPT_LEVEL, FP16_LOSS_SCALE, AUTO_CAST and SYNCBN have no effect here (bf16 compute with fp32 master weights needs no loss
# scaling; the path has no BatchNorm); RCCLDDPAccelerator.__init__ logs the ones a config sets.
