import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.embeddings import ArchiHipEmbeddings
from archi_amd.ingest import prepare_file
from tests.synth_text import make_files, make_vocab_file
td = tempfile.mkdtemp()
prov = ArchiHipEmbeddings("sentence-transformers/all-MiniLM-L6-v2",
                          model_kwargs={"synthetic_seed": 0, "device": "cuda:0", "vocab_file": make_vocab_file(os.path.join(td, "v.txt"))},
                          encode_kwargs={"normalize_embeddings": True})
chunks = []
for fh, fn, text in make_files(7, 180):
    chunks += prepare_file(fh, fn, text, "bench")[0]
ids, lens = prov.tokenizer.encode_batch_array(chunks, prov.max_seq_length)
for i in range(8):
    t0 = time.perf_counter(); prov.embed_token_arrays(ids, lens); print(f"arrays only #{i}: {1e3*(time.perf_counter()-t0):.1f} ms")
for i in range(4):
    t0 = time.perf_counter(); ids, lens = prov.tokenizer.encode_batch_array(chunks, prov.max_seq_length); t1 = time.perf_counter()
    prov.embed_token_arrays(ids, lens); print(f"tok {1e3*(t1-t0):.1f} + arrays #{i}: {1e3*(time.perf_counter()-t1):.1f} ms")
prov.tokenizer._threads = 8
for i in range(4):
    t0 = time.perf_counter(); ids, lens = prov.tokenizer.encode_batch_array(chunks, prov.max_seq_length); t1 = time.perf_counter()
    prov.embed_token_arrays(ids, lens); print(f"8-thread tok {1e3*(t1-t0):.1f} + arrays #{i}: {1e3*(time.perf_counter()-t1):.1f} ms")
