"""HipEncoder -- Python handle of the HIP BERT encoder (ak_encoder_*).

PyTorch-ROCm only HOLDS the weights in HBM (bf16 matrices, fp32 vectors) and hands raw device
pointers to the C ABI; every arithmetic step of the forward pass runs in hand-written HIP kernels
(archi_amd/csrc/encoder.hip, gemm.hip, attention.hip).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import numpy as np

from . import _lib
from ._lib import POOLING, AkBertConfig, HipBackendError, check

# name -> (vocab, hidden, layers, heads, intermediate, max_position, pooling, max_seq_length)
MODEL_SHAPES = {
    "sentence-transformers/all-MiniLM-L6-v2": (30522, 384, 6, 12, 1536, 512, "mean", 256),
    "all-MiniLM-L6-v2": (30522, 384, 6, 12, 1536, 512, "mean", 256),
    "BAAI/bge-base-en": (30522, 768, 12, 12, 3072, 512, "cls", 512),
    "BAAI/bge-base-en-v1.5": (30522, 768, 12, 12, 3072, 512, "cls", 512),
}

LAYER_KEYS = ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")
MATRIX_KEYS = {"wq", "wk", "wv", "wo", "w1", "w2"}


def weight_order(layers: int):
    names = ["word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b"]
    for l in range(layers):
        names += [f"l{l}.{k}" for k in LAYER_KEYS]
    return names


PRECISIONS = {"bf16": 0, "f32": 1, "bf16x3": 2}        # AkBertConfig.precision


class HipEncoder:
    def __init__(self, vocab: int, hidden: int, layers: int, heads: int, intermediate: int, max_position: int,
                 weights: Dict[str, np.ndarray], ln_eps: float = 1e-12, device: Optional[int] = None,
                 residual: str = "bf16", precision: str = "bf16"):
        """residual: "bf16" keeps the residual stream between layers in bf16 only (hidden size 384: 60% less epilogue
        traffic; adds ~1e-6 of cosine deviation from the fp32 reference to the ~2e-6 the bf16 GEMM inputs already
        cost); "f32" keeps it in fp32 like the reference's CPU path. ARCHI_ENCODER_RESIDUAL overrides.
        precision: "bf16" = the measured MFMA path; "f32" = parity mode: float32 weights and arithmetic throughout, on
        v_mfma_f32_32x32x2_f32 (~1e-6 from the reference's torch-fp32 CPU embedder, ~1/9 of the bf16 rate); "bf16x3" = split-bf16
        parity mode: float32 weights, every GEMM operand split x = hi + lo into two bf16 values and every product run as
        hi.hi + lo.hi + hi.lo on the bf16 matrix cores with one float32 accumulator, everything between the GEMMs in float32
        (~1e-6 per component from float64, scores within 1e-5 of the CPU path, ~3x the "f32" mode's rate)."""
        import os
        import torch
        residual = os.environ.get("ARCHI_ENCODER_RESIDUAL", residual)
        if residual not in ("bf16", "f32"):
            raise ValueError("residual must be 'bf16' or 'f32'")
        self.residual = residual
        if precision not in PRECISIONS:
            raise ValueError("precision must be 'bf16', 'f32' or 'bf16x3'")
        self.precision = precision
        self._lib = _lib.init(device)
        self.hidden, self.layers, self.max_position, self.vocab = hidden, layers, max_position, vocab
        dev = torch.device("cuda", _lib.bound_device())      # the library's device, not torch's per-thread default
        self._tensors = []   # keeps the device memory alive
        ptrs = []
        for name in weight_order(layers):
            if name not in weights:
                raise HipBackendError(f"encoder weight {name!r} missing")
            arr = weights[name]
            t = arr if isinstance(arr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(arr))
            is_matrix = name in ("word_emb", "pos_emb", "type_emb") or name.split(".")[-1] in MATRIX_KEYS
            t = t.to(device=dev, dtype=torch.bfloat16 if (is_matrix and precision == "bf16") else torch.float32).contiguous()
            self._tensors.append(t)
            ptrs.append(t.data_ptr())
        cfg = AkBertConfig(vocab, hidden, layers, heads, intermediate, max_position, 2, ln_eps, int(residual == "bf16"),
                           PRECISIONS[precision])
        arr_t = ctypes.c_void_p * len(ptrs)
        h = ctypes.c_void_p()
        torch.cuda.synchronize(dev)
        check(self._lib.ak_encoder_create(ctypes.byref(cfg), arr_t(*ptrs), len(ptrs), ctypes.byref(h)),
              "ak_encoder_create")
        self._h = h
        self._dev = dev

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ak_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, ids, mask, pooling: str = "mean", normalise: bool = True):
        """ids, mask: [B,S] integer arrays/tensors. Returns a [B,hidden] float32 CUDA tensor."""
        import torch
        ids_t = torch.as_tensor(ids, dtype=torch.int32, device=self._dev)
        mask_t = torch.as_tensor(mask, dtype=torch.int32, device=self._dev)
        B, S = ids_t.shape
        if S > self.max_position or S > 512:
            raise ValueError(f"sequence length {S} exceeds the encoder limit")
        Sp = (S + 31) // 32 * 32
        if Sp != S:   # pad with masked tokens (attention ignores them, pooling skips them)
            ids_t = torch.nn.functional.pad(ids_t, (0, Sp - S))
            mask_t = torch.nn.functional.pad(mask_t, (0, Sp - S))
        ids_t, mask_t = ids_t.contiguous(), mask_t.contiguous()
        out = torch.empty((B, self.hidden), dtype=torch.float32, device=self._dev)
        check(self._lib.ak_encoder_forward(self._h, ctypes.c_void_p(ids_t.data_ptr()),
                                           ctypes.c_void_p(mask_t.data_ptr()), B, Sp, POOLING[pooling],
                                           int(normalise), ctypes.c_void_p(out.data_ptr()),
                                           ctypes.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)),
              "ak_encoder_forward")
        return out


    def forward_lens(self, stage, n_rows: int, S: int, out, pooling: str = "mean", normalise: bool = True) -> None:
        """Right-padded rows given by their lengths (ak_encoder_forward_lens): `stage` is an int32 CUDA tensor [n_rows, S + 1] --
        S token ids per row, the row's length in column S (the provider's tile layout) --, `out` a float32 CUDA tensor view
        [n_rows, hidden] inside the caller's result buffer. No torch kernel runs: the library lays the mask out itself."""
        import torch
        if stage.dtype != torch.int32 or not stage.is_cuda or not stage.is_contiguous() or tuple(stage.shape) != (n_rows, S + 1):
            raise ValueError("forward_lens: stage must be a contiguous int32 CUDA tensor [n_rows, S + 1]")
        if out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous() or tuple(out.shape) != (n_rows, self.hidden):
            raise ValueError("forward_lens: out must be a contiguous float32 CUDA tensor [n_rows, hidden]")
        if S % 32 or S > self.max_position or S > 512:
            raise ValueError(f"sequence length {S} must be a multiple of 32 within the encoder limit")
        base = stage.data_ptr()
        check(self._lib.ak_encoder_forward_lens(self._h, ctypes.c_void_p(base), S + 1, ctypes.c_void_p(base + 4 * S), S + 1, n_rows, S,
                                                POOLING[pooling], int(normalise), ctypes.c_void_p(out.data_ptr()),
                                                ctypes.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)),
              "ak_encoder_forward_lens")


def random_init_weights(vocab, hidden, layers, intermediate, max_position, seed: int = 0) -> Dict[str, "np.ndarray"]:
    """Seeded random-init weights of a given architecture (benchmarks: no checkpoints exist offline). LayerNorm weights are drawn
    around (1, 0), not set to it: a trained checkpoint's are not trivial either, and paths that fold the LayerNorm into their
    neighbours (csrc/gemm.hip, lazy LayerNorm) would be measured and checked on a degenerate case otherwise."""
    import torch
    g = torch.Generator().manual_seed(seed)
    w = {}

    def mat(r, c, std=0.02):
        return (torch.randn(r, c, generator=g) * std).numpy()

    w["word_emb"], w["pos_emb"], w["type_emb"] = mat(vocab, hidden), mat(max_position, hidden), mat(2, hidden)
    def ln():
        return ((1.0 + 0.1 * torch.randn(hidden, generator=g)).numpy().astype(np.float32),
                (0.1 * torch.randn(hidden, generator=g)).numpy().astype(np.float32))

    w["emb_ln_g"], w["emb_ln_b"] = ln()
    for l in range(layers):
        p = f"l{l}."
        for k, (r, c) in (("wq", (hidden, hidden)), ("wk", (hidden, hidden)), ("wv", (hidden, hidden)),
                          ("wo", (hidden, hidden)), ("w1", (intermediate, hidden)), ("w2", (hidden, intermediate))):
            w[p + k] = mat(r, c)
        for k, d in (("bq", hidden), ("bk", hidden), ("bv", hidden), ("bo", hidden), ("b1", intermediate), ("b2", hidden)):
            w[p + k] = (torch.randn(d, generator=g) * 0.02).numpy()
        for k in ("ln1", "ln2"):
            w[p + k + "_g"], w[p + k + "_b"] = ln()
    return w


def load_hf_weights(model_dir: str):
    """Load a local HF BERT checkpoint directory (config.json + model.safetensors | pytorch_model.bin). No network.
    Anything the HIP encoder does not implement (non-BERT, non-GELU, relative positions) is refused here."""
    import json
    import os
    cfg = json.load(open(os.path.join(model_dir, "config.json")))
    if cfg.get("model_type", "bert") != "bert":
        raise ValueError(f"{model_dir}: model_type {cfg.get('model_type')!r} is not BERT")
    if cfg.get("hidden_act", "gelu") != "gelu":
        raise ValueError(f"{model_dir}: hidden_act {cfg.get('hidden_act')!r} (the HIP encoder implements erf GELU)")
    if cfg.get("position_embedding_type", "absolute") != "absolute":
        raise ValueError(f"{model_dir}: position_embedding_type {cfg.get('position_embedding_type')!r} is not supported")
    st, pt = os.path.join(model_dir, "model.safetensors"), os.path.join(model_dir, "pytorch_model.bin")
    if os.path.exists(st):
        from safetensors.torch import load_file     # torch loader: checkpoints stored in f16/bf16 load too
        sd = load_file(st)
    elif os.path.exists(pt):
        import torch
        sd = torch.load(pt, map_location="cpu", weights_only=True)
    else:
        raise FileNotFoundError(f"{model_dir}: neither model.safetensors nor pytorch_model.bin")
    sd = {(k[5:] if k.startswith("bert.") else k): v.float() for k, v in sd.items()}
    L = cfg["num_hidden_layers"]
    w = {"word_emb": sd["embeddings.word_embeddings.weight"], "pos_emb": sd["embeddings.position_embeddings.weight"],
         "type_emb": sd["embeddings.token_type_embeddings.weight"], "emb_ln_g": sd["embeddings.LayerNorm.weight"],
         "emb_ln_b": sd["embeddings.LayerNorm.bias"]}
    for l in range(L):
        p, q = f"encoder.layer.{l}.", f"l{l}."
        for hf, m in (("attention.self.query", "q"), ("attention.self.key", "k"), ("attention.self.value", "v")):
            w[q + "w" + m], w[q + "b" + m] = sd[p + hf + ".weight"], sd[p + hf + ".bias"]
        w[q + "wo"], w[q + "bo"] = sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"]
        w[q + "ln1_g"], w[q + "ln1_b"] = sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"]
        w[q + "w1"], w[q + "b1"] = sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]
        w[q + "w2"], w[q + "b2"] = sd[p + "output.dense.weight"], sd[p + "output.dense.bias"]
        w[q + "ln2_g"], w[q + "ln2_b"] = sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"]
    shape = (cfg["vocab_size"], cfg["hidden_size"], L, cfg["num_attention_heads"], cfg["intermediate_size"],
             cfg["max_position_embeddings"])
    return shape, w, float(cfg.get("layer_norm_eps", 1e-12))


def read_sentence_transformers_config(model_dir: str):
    """What SentenceTransformer(model_dir) -- the engine under the reference's HuggingFaceEmbeddings [upstream] --
    reads beside the BERT weights: `modules.json` (is there a Normalize module), `1_Pooling/config.json`
    (cls | mean) and `sentence_bert_config.json` (max_seq_length). A plain HF directory without them gets
    sentence-transformers' defaults: mean pooling, no Normalize module, max_seq_length None (= the model limit).
    Returns (pooling, max_seq_length | None, always_normalise)."""
    import json
    import os
    pooling, max_len, norm = "mean", None, False
    mj = os.path.join(model_dir, "modules.json")
    pool_dir = "1_Pooling"
    if os.path.exists(mj):
        for m in json.load(open(mj)):
            kind = m.get("type", "")
            if kind.endswith("Normalize"):
                norm = True
            elif kind.endswith("Pooling"):
                pool_dir = m.get("path", pool_dir)
    pj = os.path.join(model_dir, pool_dir, "config.json")
    if os.path.exists(pj):
        pc = json.load(open(pj))
        modes = [k for k in ("cls_token", "mean_tokens", "max_tokens", "mean_sqrt_len_tokens", "weightedmean_tokens",
                             "lasttoken") if pc.get("pooling_mode_" + k)]
        if modes == ["cls_token"]:
            pooling = "cls"
        elif modes != ["mean_tokens"]:
            raise ValueError(f"{model_dir}: pooling modes {modes} (the HIP encoder implements cls and mean)")
    sj = os.path.join(model_dir, "sentence_bert_config.json")
    if os.path.exists(sj):
        max_len = json.load(open(sj)).get("max_seq_length")
    return pooling, max_len, norm
