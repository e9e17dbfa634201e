"""Test helper: the CPU oracle behind the backend interface of distributed-sgd_amd/host.py (oracle/backend.py), so that the
host-side orchestration (Master.fit mirror, host-owned all-reduce) can be exercised without a GPU."""

from oracle.backend import OracleBackend  # noqa: F401
