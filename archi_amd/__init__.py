"""archi_amd -- MI355X (gfx950) embedding + retrieval backend that drops in behind
archi's embedding-provider / vector-store plugin surface
(src/data_manager/vectorstore/postgres_vectorstore.py, manager.py:66-73,
src/archi/utils/vectorstore_connector.py:28-69 in the reference).

Python host code -> ctypes -> libarchi_hip.so (hand-written HIP for gfx950).
"""
from ._lib import HipBackendError, StaleFilterError  # noqa: F401

__all__ = ["HipBackendError", "StaleFilterError"]
