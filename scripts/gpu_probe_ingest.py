#!/usr/bin/env python3
"""End-to-end ingestion rate: files -> split -> tokenise -> embed -> store (BatchedIngestor + ArchiHipVectorStore)."""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from archi_amd import vectorstore as vs
from archi_amd.embeddings import ArchiHipEmbeddings
from archi_amd.ingest import BatchedIngestor
from tests.synth_text import make_files, make_vocab_file
td = tempfile.mkdtemp()
emb = ArchiHipEmbeddings("sentence-transformers/all-MiniLM-L6-v2",
                         model_kwargs={"synthetic_seed": 0, "device": "cuda:0", "vocab_file": make_vocab_file(os.path.join(td, "v.txt"))},
                         encode_kwargs={"normalize_embeddings": True})
files = make_files(7, int(os.environ.get("AK_FILES", "180")))
for rep in range(3):
    vs.reset_collections()
    store = vs.ArchiHipVectorStore({"hip": {"dtype": "f32", "capacity": 100000}}, emb, collection_name=f"c{rep}")
    ing = BatchedIngestor(store, f"c{rep}")
    t0 = time.perf_counter()
    if rep == 2:
        pr = cProfile.Profile(); pr.enable()
    done = ing.ingest(files, document_ids={f[0]: i + 1 for i, f in enumerate(files)})
    if rep == 2:
        pr.disable()
    dt = time.perf_counter() - t0
    n = sum(len(v) for v in done.values())
    print(f"{len(files)} files, {n} chunks: {dt * 1e3:.0f} ms -> {n / dt:.0f} chunks/s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
