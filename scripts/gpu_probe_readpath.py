#!/usr/bin/env python3
"""The read path of one chat request through the drop-in classes: similarity_search_with_score(query, k) on a collection
built by the ingestor -- embed_query + ak_index_search + Document assembly, host buffers and Python included."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd import vectorstore as vs
from archi_amd.embeddings import ArchiHipEmbeddings
from archi_amd.ingest import BatchedIngestor
from tests.synth_text import make_files, make_vocab_file
td = tempfile.mkdtemp()
emb = ArchiHipEmbeddings("sentence-transformers/all-MiniLM-L6-v2",
                         model_kwargs={"synthetic_seed": 0, "device": "cuda:0", "vocab_file": make_vocab_file(os.path.join(td, "v.txt"))},
                         encode_kwargs={"normalize_embeddings": True})
store = vs.ArchiHipVectorStore({"hip": {"dtype": "f32", "capacity": 200000}}, emb, collection_name="docs")
files = make_files(7, int(os.environ.get("AK_FILES", "4000")))
t0 = time.perf_counter()
done = BatchedIngestor(store, "docs").ingest(files, document_ids={f[0]: i + 1 for i, f in enumerate(files)})
print(f"ingested {store.count()} chunks of {len(files)} files in {time.perf_counter() - t0:.2f} s")
q = "kalo miren stavor quzen phitor elan droxi bune sygra"
for k in (4, 10):
    for _ in range(20):
        store.similarity_search_with_score(q, k=k)
    t0 = time.perf_counter()
    for _ in range(200):
        res = store.similarity_search_with_score(q, k=k)
    print(f"similarity_search_with_score(k={k}): {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms  (top score {res[0][1]:.4f})")
t0 = time.perf_counter()
for _ in range(200):
    v = emb.embed_query(q)
print(f"  embed_query alone: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
t0 = time.perf_counter()
for _ in range(200):
    store.similarity_search_by_vector_with_score(v, k=4)
print(f"  similarity_search_by_vector_with_score alone: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
t0 = time.perf_counter()
for _ in range(200):
    store.similarity_search_with_score(q, k=4, filter={"filename": files[3][1]})
print(f"  with a metadata filter (cached WHERE mask): {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
col = store._collection()
qv = np.asarray(v, np.float32)[None]
for k in (4, 10, 20):
    _, d, _, st = col.index.search(qv, k, return_stats=True)
    print(f"  index.search k={k}: {st}; distances {np.round(d[0][:k], 5).tolist()}")
