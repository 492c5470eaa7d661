#!/usr/bin/env python3
"""Generate tests/golden/encoder_*.npz by running transformers.BertModel (the engine
sentence-transformers drives for the reference's HuggingFaceEmbeddings, [upstream]) on the
synthetic weights of oracle/encoder_oracle.synth_weights, in the build container.

Stored per case: token ids, attention mask, and the HF outputs after the sentence-transformers
pooling (mean | cls) + L2-normalise steps. Weights are NOT stored (regenerated from the seed).
    python tests/golden/make_encoder_fixtures.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import encoder_oracle as eo  # noqa: E402


def hf_model(shape, w):
    from transformers import BertConfig, BertModel
    vocab, H, L, heads, I, max_pos, _ = eo.SHAPES[shape]
    cfg = BertConfig(vocab_size=vocab, hidden_size=H, num_hidden_layers=L, num_attention_heads=heads,
                     intermediate_size=I, max_position_embeddings=max_pos, hidden_act="gelu",
                     layer_norm_eps=1e-12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    sd = {}
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    sd["embeddings.word_embeddings.weight"] = t(w["word_emb"])
    sd["embeddings.position_embeddings.weight"] = t(w["pos_emb"])
    sd["embeddings.token_type_embeddings.weight"] = t(w["type_emb"])
    sd["embeddings.LayerNorm.weight"] = t(w["emb_ln_g"]); sd["embeddings.LayerNorm.bias"] = t(w["emb_ln_b"])
    for l in range(L):
        p, q = f"encoder.layer.{l}.", f"l{l}."
        for hf, mine in (("attention.self.query", "q"), ("attention.self.key", "k"), ("attention.self.value", "v")):
            sd[p + hf + ".weight"] = t(w[q + "w" + mine]); sd[p + hf + ".bias"] = t(w[q + "b" + mine])
        sd[p + "attention.output.dense.weight"] = t(w[q + "wo"]); sd[p + "attention.output.dense.bias"] = t(w[q + "bo"])
        sd[p + "attention.output.LayerNorm.weight"] = t(w[q + "ln1_g"]); sd[p + "attention.output.LayerNorm.bias"] = t(w[q + "ln1_b"])
        sd[p + "intermediate.dense.weight"] = t(w[q + "w1"]); sd[p + "intermediate.dense.bias"] = t(w[q + "b1"])
        sd[p + "output.dense.weight"] = t(w[q + "w2"]); sd[p + "output.dense.bias"] = t(w[q + "b2"])
        sd[p + "output.LayerNorm.weight"] = t(w[q + "ln2_g"]); sd[p + "output.LayerNorm.bias"] = t(w[q + "ln2_b"])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "position_ids" not in k], missing
    assert not unexpected, unexpected
    return m


def hf_embed(shape, w, ids, mask, pooling):
    m = hf_model(shape, w)
    with torch.no_grad():
        h = m(input_ids=torch.from_numpy(ids).long(), attention_mask=torch.from_numpy(mask).long()).last_hidden_state
    mk = torch.from_numpy(mask).float()
    if pooling == "cls":
        out = h[:, 0]
    else:  # sentence_transformers.models.Pooling (mean): sum / clamp(sum_mask, 1e-9)
        out = (h * mk[:, :, None]).sum(1) / mk.sum(1, keepdim=True).clamp(min=1e-9)
    return torch.nn.functional.normalize(out, p=2, dim=1).numpy()


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name, shape, B, S, pooling, seed in (("tiny_B5_S24_mean", "tiny", 5, 24, "mean", 3),
                                              ("tiny_B3_S64_cls", "tiny", 3, 64, "cls", 4),
                                              ("minilm_B4_S32_mean", "minilm-l6", 4, 32, "mean", 5),
                                              ("minilm_B2_S256_mean", "minilm-l6", 2, 256, "mean", 6),
                                              ("bge_B2_S64_cls", "bge-base", 2, 64, "cls", 8)):
        w = eo.synth_weights(shape, seed=7)
        ids, mask = eo.synth_tokens(B, S, seed=seed, vocab=eo.SHAPES[shape][0])
        ref = hf_embed(shape, w, ids, mask, pooling)
        mine = eo.forward(shape, w, ids, mask, pooling=pooling)
        err = np.abs(ref - mine).max()
        print(f"{name}: HF vs restatement max|diff| = {err:.3e}")
        assert err < 2e-6
        np.savez_compressed(os.path.join(here, f"encoder_{name}.npz"), ids=ids, mask=mask, expected=ref,
                            shape=shape, pooling=pooling, weight_seed=7)


if __name__ == "__main__":
    main()
