"""Data-parallel back ends behind the reference's Accelerator contract (accelerators/accelerator.py:15-32,
accelerators/__init__.py:12-15)."""
from .accelerator import Accelerator
from .rccl_ddp_accelerator import RCCLDDPAccelerator

ACCELERATOR_MAP = {"RCCLDDP": RCCLDDPAccelerator, "DDP": RCCLDDPAccelerator, "ApexDDP": RCCLDDPAccelerator,
                   "TorchAMPDDP": RCCLDDPAccelerator}
