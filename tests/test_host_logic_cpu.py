"""CPU suite: host-side pieces of the path that need no GPU."""
import numpy as np
import pytest

from archi_amd import config_plugin as cp


def test_embedding_text_round_trip_is_lossless():
    """a4: the reference ships vectors to pgvector as text, "[" + ",".join(str(x)) + "]" then
    ::vector (float4) (postgres_vectorstore.py:313-314). For float32-origin values that round trip
    is exact, so the store may take the list[float] straight to float32."""
    rng = np.random.default_rng(0)
    v32 = (rng.standard_normal(5000) * np.exp(rng.uniform(-20, 20, 5000))).astype(np.float32)
    as_python = [float(x) for x in v32]                      # what embed_query returns
    text = "[" + ",".join(str(x) for x in as_python) + "]"
    back = np.array([float(t) for t in text.strip("[]").split(",")], dtype=np.float64).astype(np.float32)
    assert np.array_equal(back, v32)
    assert str(float(np.float32(0.1))) == "0.10000000149011612"


def test_config_plugin_resolves_like_the_reference():
    m = {"ArchiHipEmbeddings": {"class": "ArchiHipEmbeddings", "kwargs": {"model_name": "BAAI/bge-base-en"}},
         "OpenAIEmbeddings": {"class": "OpenAIEmbeddings", "kwargs": {"model": "text-embedding-3-small"}},
         "Bare": None}
    r = cp.resolve_embedding_classes(m)
    from archi_amd.embeddings import ArchiHipEmbeddings
    assert r["ArchiHipEmbeddings"]["class"] is ArchiHipEmbeddings
    assert r["OpenAIEmbeddings"]["class"] == "OpenAIEmbeddings" and r["Bare"] == {}
    assert cp.resolve_embedding_classes({}) == {}
    assert cp.embedding_dimensions(r["ArchiHipEmbeddings"]) == 768
    assert cp.embedding_dimensions({"dimensions": 1024}) == 1024
    assert cp.map_distance_metric("ip") == "inner_product" and cp.map_distance_metric("l2") == "l2"
    with pytest.raises(ValueError, match="is not supported"):
        cp.map_distance_metric("manhattan")
