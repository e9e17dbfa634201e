"""distributed-sgd_amd -- MI355X-native engine for the hot path of zifeo/distributed-sgd.

    Engine            one libdsgd_hip context (HIP kernels behind the C ABI of include/dsgd.h)
    EngineGroup       several contexts (one per device) driven by one host thread: dsgd_*_devices
    synth             synthetic RCV1-like CSR generator
    host              host-side mirror of the reference's Master / Slave / SparseSVM surface
    rcv1              RCV1-v2 text files <-> CSR with the reference loader's semantics (Dataset.rcv1)
    wire              the reference's `Slave` gRPC service in front of an engine (proto.proto), for unmodified masters
    DenseLogistic     the optional dense mini-batch variant of BASELINE.json configs[4] (no reference counterpart)

The directory name is not an importable identifier; `import dsgd_amd` (repo root) aliases it.
"""

from . import _build, _lib, host, rcv1, synth, wire  # noqa: F401
from ._lib import DsgdError, DsgdIndexError, DsgdInvalidArgument  # noqa: F401
from .dense import DenseLogistic  # noqa: F401
from .engine import Engine, EngineGroup, Plan, device_count  # noqa: F401

__all__ = ["Engine", "EngineGroup", "Plan", "DenseLogistic", "device_count", "synth", "host", "rcv1", "wire", "DsgdError", "DsgdIndexError", "DsgdInvalidArgument"]
