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


def resolve_embedding_classes(embedding_class_map: Dict[str, Any], mapping: Dict[str, Any] = None) -> Dict[str, Any]:
    """Same contract as the reference's resolver: entries whose `class` (or whose key, when `class` is
    absent) names a known embedder get the callable; everything else passes through untouched.
    `mapping`: the name -> callable table; default = this backend's own entry. A deployment that keeps the reference's
    embedders passes their table merged with embedding_mapping(). Pinned against the
    reference resolver's recorded output in tests/golden/reference_wrapper.json["config"]."""
    if not embedding_class_map:
        return {}
    mapping = dict(embedding_mapping() if mapping is None else mapping)
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


# src/cli/managers/templates_manager.py:408-413 (name -> dimensions of the rendered `vector(D)` column)
_TEMPLATE_DEFAULT_DIMENSIONS = {
    "all-MiniLM-L6-v2": 384,
    "text-embedding-ada-002": 1536,
    "text-embedding-3-small": 1536,
    "text-embedding-3-large": 3072,
}


def init_sql_dimensions(data_manager_config: Dict[str, Any]) -> int:
    """The D of `embedding vector(D)` the reference renders into init.sql (templates_manager.py:403-421): table lookup by
    `embedding_name` (default all-MiniLM-L6-v2, unknown names -> 384), overridden by
    `embedding_class_map[embedding_name].dimensions`. An ArchiHipEmbeddings entry for a 768-d model therefore MUST carry
    `dimensions: 768` (embedding_dimensions() above fills it in from the model name when a config is generated)."""
    cmap = data_manager_config.get("embedding_class_map", {}) or {}
    name = data_manager_config.get("embedding_name", "all-MiniLM-L6-v2")
    dims = _TEMPLATE_DEFAULT_DIMENSIONS.get(name, 384)
    if name in cmap:
        dims = (cmap[name] or {}).get("dimensions", dims)
    return dims


def map_distance_metric(manager_metric: str) -> str:
    """manager config uses l2|cosine|ip and maps ip -> inner_product (manager.py:21,160-165)."""
    if manager_metric not in ("l2", "cosine", "ip"):
        raise ValueError(f"The selected distance metrics, '{manager_metric}', is not supported. "
                         f"Must be one of ['l2', 'cosine', 'ip']")
    return "inner_product" if manager_metric == "ip" else manager_metric
