"""Ingestion harness: file text -> chunks -> cross-file batched embedding -> store upsert.

The build's counterpart of `VectorStoreManager._add_to_postgres`
(/root/reference/src/data_manager/vectorstore/manager.py:262-449):

  * chunking follows the reference's splitter configuration -- `CharacterTextSplitter(chunk_size,
    chunk_overlap)` (manager.py:75-78; defaults 1000 / 0, src/cli/templates/base-config.yaml:153-154).
    The splitter itself is third-party (langchain-text-splitters, not in the image); `split_text`
    restates its published algorithm: split on the separator, greedily merge pieces up to chunk_size,
    carry `chunk_overlap` characters of pieces into the next chunk, strip whitespace, drop empties.
  * per-chunk metadata follows manager.py:300-322 (`chunk_index`, `filename`, `resource_hash`,
    `collection`; NUL bytes removed; blank chunks skipped but still counted in `chunk_index`).
  * behind a row-sharded index (pg_config["hip"]["shards"]) every rank runs this loop on the same files (SPMD) and embeds only
    the chunks whose rows land on its shard: row ids are the table's SERIAL key, so the ids of a group are known before it
    is embedded (store.embed_for_rows / add_texts_batch(plan=...)); no collective.
  * the reference embeds one file per call (manager.py:362-373), i.e. tiny batches. Here the chunks of many
    files (groups of ~2048 chunks, whole files) go through ONE `embed_documents` call, so the embedder can
    sort by length and fill `[B,S]` tiles, and group g+1 is embedded on a helper thread while group g's rows
    are written into the store; the per-file failure semantics (manager.py:374-389: a file whose embedding raises is
    marked failed, the others continue) are kept by retrying file by file when the joint call raises.
"""
from __future__ import annotations

import re
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple


def _join(pieces: Sequence[str], separator: str) -> Optional[str]:
    text = separator.join(pieces).strip()
    return text if text else None


def split_text(text: str, chunk_size: int = 1000, chunk_overlap: int = 0, separator: str = "\n\n") -> List[str]:
    """CharacterTextSplitter.split_text [upstream langchain-text-splitters]: a piece longer than
    chunk_size is kept whole (this splitter never cuts inside a piece)."""
    if chunk_overlap > chunk_size:
        raise ValueError(f"Got a larger chunk overlap ({chunk_overlap}) than chunk size ({chunk_size}), should be smaller.")
    pieces = [p for p in (re.split(re.escape(separator), text) if separator else list(text)) if p != ""]
    sep_len = len(separator)
    out: List[str] = []
    cur: List[str] = []
    total = 0
    for piece in pieces:
        n = len(piece)
        if total + n + (sep_len if cur else 0) > chunk_size and cur:
            doc = _join(cur, separator)
            if doc is not None:
                out.append(doc)
            # drop pieces from the front until what is carried over fits the overlap and the new piece fits
            while total > chunk_overlap or (total + n + (sep_len if cur else 0) > chunk_size and total > 0):
                total -= len(cur[0]) + (sep_len if len(cur) > 1 else 0)
                cur = cur[1:]
        cur.append(piece)
        total += n + (sep_len if len(cur) > 1 else 0)
    doc = _join(cur, separator)
    if doc is not None:
        out.append(doc)
    return out


def prepare_file(filehash: str, filename: str, text: str, collection: str,
                 file_metadata: Optional[Dict[str, Any]] = None, chunk_size: int = 1000,
                 chunk_overlap: int = 0) -> Tuple[List[str], List[Dict[str, Any]]]:
    """One file -> (chunks, metadatas), manager.py:300-322."""
    chunks: List[str] = []
    metadatas: List[Dict[str, Any]] = []
    for index, chunk in enumerate(split_text(text, chunk_size, chunk_overlap)):
        chunk = chunk.replace("\x00", "")
        if not chunk.strip():
            continue
        chunks.append(chunk)
        meta = dict(file_metadata or {})
        meta.update(chunk_index=index, filename=filename, resource_hash=filehash, collection=collection)
        metadatas.append(meta)
    return chunks, metadatas


class BatchedIngestor:
    """files -> store, embedding across files in one call.

    `store` is an ArchiHipVectorStore (or anything with `.embeddings` and
    `add_texts(texts, metadatas, document_id=..., embeddings=...)`); `on_status(filehash, status, error)`
    receives what the reference writes to `documents.ingestion_status` (manager.py:374-389,440-447)."""

    def __init__(self, store: Any, collection: str = "default", chunk_size: int = 1000, chunk_overlap: int = 0,
                 on_status: Optional[Callable[[str, str, Optional[str]], None]] = None, group_chunks: int = 2048):
        self.store = store
        self.group_chunks = group_chunks
        self.collection = collection
        self.chunk_size = chunk_size
        self.chunk_overlap = chunk_overlap
        self.on_status = on_status or (lambda h, s, e: None)

    def ingest(self, files: Iterable[Tuple[str, str, str]], document_ids: Optional[Dict[str, Any]] = None,
               file_metadata: Optional[Dict[str, Dict[str, Any]]] = None) -> Dict[str, List[str]]:
        """files: (filehash, filename, text). Returns {filehash: chunk ids} for the files embedded."""
        embedder = self.store.embeddings
        has_array = callable(getattr(type(embedder), "embed_documents_array", None))
        embed = embedder.embed_documents_array if has_array else embedder.embed_documents      # float32 rows if offered

        # groups of whole files, ~group_chunks chunks each, split and prepared lazily: while the GPU embeds group g
        # (helper thread; the call releases the GIL while it waits) this thread writes group g-1's rows into the store and
        # prepares group g+1
        def groups_of(it):
            group, count = [], 0
            for filehash, filename, text in it:
                chunks, metas = prepare_file(filehash, filename, text, self.collection,
                                             (file_metadata or {}).get(filehash), self.chunk_size, self.chunk_overlap)
                if not chunks:
                    self.on_status(filehash, "failed", "No text chunks could be extracted")   # manager.py:324-327
                    continue
                group.append((filehash, chunks, metas))
                count += len(chunks)
                if count >= self.group_chunks:
                    yield group
                    group, count = [], 0
            if group:
                yield group

        # row-sharded store: a group's chunks become rows rid0, rid0 + 1, ... and this rank embeds only its own (1 / world of
        # them). The ids of group g + 1 are predicted while group g is still being written; a wrong guess (a failed file in
        # between) is refused by add_texts_batch and that group is embedded again, file by file
        sharded = callable(getattr(type(self.store), "embed_for_rows", None)) and self.store.shard_layout()[0] > 1
        next_rid = self.store.next_row_id() if sharded else None

        def embed_group(group, rid0=None):
            try:
                texts = [c for _, chunks, _ in group for c in chunks]
                if rid0 is not None:
                    vecs, mine, rid0 = self.store.embed_for_rows(texts, rid0)
                    return vecs, (rid0, mine)
                return embed(texts), None
            except Exception as exc:            # isolate the failing file in the caller
                return None, (("failed", exc) if rid0 is not None else None)

        done: Dict[str, List[str]] = {}
        from concurrent.futures import ThreadPoolExecutor
        source = groups_of(files)
        group = next(source, None)
        if group is None:
            return {}
        size = lambda g: sum(len(chunks) for _, chunks, _ in g)      # noqa: E731
        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = pool.submit(embed_group, group, next_rid)
            while group is not None:
                nxt = next(source, None)                  # split + prepare the next group while the GPU works on this one
                vectors, plan = pending.result()
                if nxt is not None:      # predicted first row id of the next group: this group's rows come first
                    pending = pool.submit(embed_group, nxt, None if next_rid is None else next_rid + size(group))
                self._write_group(group, vectors, embed, document_ids, done, plan)
                if next_rid is not None:
                    next_rid = self.store.next_row_id()       # where the table really stands (a failed file shifts the ids)
                group = nxt
        return done

    def _write_group(self, group, vectors, embed, document_ids, done, plan=None) -> None:
        """Store one embedded group. vectors is None when the joint embed call raised: then every file is embedded on its
        own so that only the failing one is marked failed (manager.py:374-389). plan: (rid0, mine) of a row-sharded store
        (only this rank's rows are real); the file-by-file path then lets the store embed each file's share itself."""
        if plan is not None:
            # SPMD: every rank must take the same path from here. A share that failed to embed on ANY rank sends ALL of them
            # file by file (store.agree_embedded: one int32 all-reduce), where each file's add_texts agrees again
            err = plan[1] if plan[0] == "failed" else None
            try:
                self.store.agree_embedded(err)
            except Exception:
                vectors = None
            if vectors is None:
                plan = ("file by file", None)
        if vectors is not None and callable(getattr(type(self.store), "add_texts_batch", None)):
            # one index update for the whole group; if it fails, fall through to file-by-file to isolate the culprit
            try:
                items, pos = [], 0
                for filehash, chunks, metas in group:
                    items.append((chunks, metas, (document_ids or {}).get(filehash), vectors[pos: pos + len(chunks)]))
                    pos += len(chunks)
                for (filehash, _, _), ids in zip(group, self.store.add_texts_batch(items, plan=plan) if plan is not None
                                                 else self.store.add_texts_batch(items)):      # (plan: a real (rid0, mine) here)
                    done[filehash] = ids
                    self.on_status(filehash, "embedded", None)
                return
            except Exception:
                for filehash, _, _ in group:
                    done.pop(filehash, None)
        pos = 0
        for filehash, chunks, metas in group:
            try:
                doc_id = (document_ids or {}).get(filehash)
                if plan is not None:             # the plan died with the batch: the store embeds this rank's share of the file
                    done[filehash] = self.store.add_texts(chunks, metas, document_id=doc_id)
                else:
                    vecs = vectors[pos: pos + len(chunks)] if vectors is not None else embed(chunks)
                    done[filehash] = self.store.add_texts(chunks, metas, document_id=doc_id, embeddings=vecs)
                self.on_status(filehash, "embedded", None)
            except Exception as exc:            # manager.py:374-389: mark failed, keep going
                self.on_status(filehash, "failed", str(exc))
            pos += len(chunks)
