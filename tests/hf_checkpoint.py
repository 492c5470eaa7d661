"""Test helper: write a small sentence-transformers-style BERT checkpoint directory with transformers
(config.json, model.safetensors, vocab.txt, modules.json, 1_Pooling/config.json, sentence_bert_config.json) --
the on-disk layout the reference's HuggingFaceEmbeddings loads [upstream sentence-transformers] -- and embed text
with transformers.BertModel in fp32 as the reference engine would."""
import json
import os

import numpy as np

WORDS = ["the", "muon", "detector", "cal", "##ib", "##ration", "run", "grid", ".", ",", "beam", "trigger", "##s",
         "jet", "energy", "of", "a", "is", "in", "and", "data", "##set", "tier", "site", "job", "fail", "##ed"]


def write_checkpoint(path, pooling="mean", max_seq_length=32, normalize=True, dtype="float32", seed=0):
    import torch
    from transformers import BertConfig, BertModel
    torch.manual_seed(seed)
    cfg = BertConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                     intermediate_size=256, max_position_embeddings=64, hidden_act="gelu", layer_norm_eps=1e-12,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, initializer_range=0.1)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():                      # non-trivial LayerNorm parameters and biases
        for n, p in model.named_parameters():
            if "LayerNorm.weight" in n:
                p.add_(0.1 * torch.randn_like(p))
            elif "bias" in n:
                p.add_(0.05 * torch.randn_like(p))
    model = model.to(getattr(torch, dtype))
    os.makedirs(path, exist_ok=True)
    model.save_pretrained(path, safe_serialization=True)
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS
    vocab += [f"tok{i}" for i in range(1000 - len(vocab))]
    with open(os.path.join(path, "vocab.txt"), "w") as f:
        f.write("\n".join(vocab) + "\n")
    modules = [{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
               {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]
    if normalize:
        modules.append({"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"})
    json.dump(modules, open(os.path.join(path, "modules.json"), "w"))
    os.makedirs(os.path.join(path, "1_Pooling"), exist_ok=True)
    json.dump({"word_embedding_dimension": 128, "pooling_mode_cls_token": pooling == "cls",
               "pooling_mode_mean_tokens": pooling == "mean", "pooling_mode_max_tokens": False,
               "pooling_mode_mean_sqrt_len_tokens": False}, open(os.path.join(path, "1_Pooling", "config.json"), "w"))
    json.dump({"max_seq_length": max_seq_length, "do_lower_case": False},
              open(os.path.join(path, "sentence_bert_config.json"), "w"))
    return model.float()


def hf_embed(model, ids, mask, pooling="mean", normalize=True):
    import torch
    with torch.no_grad():
        h = model(input_ids=torch.from_numpy(np.asarray(ids)).long(),
                  attention_mask=torch.from_numpy(np.asarray(mask)).long()).last_hidden_state
    mk = torch.from_numpy(np.asarray(mask)).float()
    out = h[:, 0] if pooling == "cls" else (h * mk[:, :, None]).sum(1) / mk.sum(1, keepdim=True).clamp(min=1e-9)
    if normalize:
        out = torch.nn.functional.normalize(out, p=2, dim=1)
    return out.numpy()


TEXTS = ["The muon detector calibration run.", "grid, grid grid", "jet energy of a beam trigger",
         "datasets in the tier site and jobs failed", "run " * 40, "unknownword the data"]
