"""Config plug-in point: lets archi's YAML select this backend.

Mirrors `ConfigService._resolve_embedding_classes`
(/root/reference/src/utils/config_service.py:470-496): class names in
`data_manager.embedding_class_map[*].class` are mapped to callables. A maintainer merges
EMBEDDING_MAPPING below into the reference's table; the functions here
let a deployment do the same without touching the reference, and keep the dimension bookkeeping
(src/cli/managers/templates_manager.py:408-421, src/utils/config_service.py:888-901) that feeds
`vector({{embedding_dimensions}})` in src/cli/templates/init.sql:266.
"""
from __future__ import annotations

from typing import Any, Dict

# model name -> embedding dimension (the reference's table knows MiniLM/OpenAI sizes; bge-base adds 768)
EMBEDDING_DIMENSIONS = {
    "sentence-transformers/all-MiniLM-L6-v2": 384,
    "all-MiniLM-L6-v2": 384,
    "BAAI/bge-base-en": 768,
    "BAAI/bge-base-en-v1.5": 768,
}


def embedding_mapping() -> Dict[str, Any]:
    from .embeddings import ArchiHipEmbeddings
    return {"ArchiHipEmbeddings": ArchiHipEmbeddings}


def resolve_embedding_classes(embedding_class_map: Dict[str, Any]) -> Dict[str, Any]:
    """Same contract as the reference's resolver: entries whose `class` (or whose key, when `class` is
    absent) names a known embedder get the callable; everything else passes through untouched."""
    if not embedding_class_map:
        return {}
    mapping = embedding_mapping()
    resolved: Dict[str, Any] = {}
    for name, cfg in embedding_class_map.items():
        entry = dict(cfg or {})
        cls_name = entry.get("class")
        if isinstance(cls_name, str) and cls_name in mapping:
            entry["class"] = mapping[cls_name]
        elif cls_name is None and name in mapping:
            entry["class"] = mapping[name]
        resolved[name] = entry
    return resolved


def embedding_dimensions(entry: Dict[str, Any]) -> int:
    """`dimensions` override first (reference behaviour), else the table, else 384 (init.sql default)."""
    if "dimensions" in entry:
        return int(entry["dimensions"])
    model = (entry.get("kwargs") or {}).get("model_name") or (entry.get("kwargs") or {}).get("model")
    return EMBEDDING_DIMENSIONS.get(model, 384)


def map_distance_metric(manager_metric: str) -> str:
    """manager config uses l2|cosine|ip and maps ip -> inner_product (manager.py:21,160-165)."""
    if manager_metric not in ("l2", "cosine", "ip"):
        raise ValueError(f"The selected distance metrics, '{manager_metric}', is not supported. "
                         f"Must be one of ['l2', 'cosine', 'ip']")
    return "inner_product" if manager_metric == "ip" else manager_metric
