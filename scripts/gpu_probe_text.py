#!/usr/bin/env python3
"""Where the text -> embedding time goes: tokenizer, tile preparation, GPU (HIP events per tile), D2H."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

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
prov.embed_documents_array(chunks)
for _ in range(2):
    t0 = time.perf_counter()
    ids, lens = prov.tokenizer.encode_batch_array(chunks, prov.max_seq_length)
    t1 = time.perf_counter()
    out = prov.embed_token_arrays(ids, lens)
    t2 = time.perf_counter()
    print(f"{len(chunks)} chunks, mean {lens.mean():.0f} tokens: tokenizer {1e3 * (t1 - t0):.1f} ms, tiles+GPU+D2H {1e3 * (t2 - t1):.1f} ms "
          f"-> {len(chunks) / (t2 - t0):.0f} chunks/s")
# GPU-only: the same tiles, inputs resident
order = np.argsort(-lens.astype(np.int64), kind="stable")
i, tot = 0, 0.0
while i < len(order):
    S = max(32, (int(lens[order[i]]) + 31) // 32 * 32)
    nb = max(1, prov.batch_tokens // S)
    n = len(order[i:i + nb])
    tid = torch.randint(1000, 30000, (n, S), dtype=torch.int32, device="cuda")
    m = torch.ones((n, S), dtype=torch.int32, device="cuda")
    prov.encoder.forward(tid, m)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        prov.encoder.forward(tid, m)
    e1.record()
    torch.cuda.synchronize()
    tot += e0.elapsed_time(e1) / 3
    i += nb
print(f"sum of tile GPU times {tot:.1f} ms -> {len(chunks) / tot * 1e3:.0f} chunks/s if the host kept up")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); prov.embed_token_arrays(ids, lens); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
