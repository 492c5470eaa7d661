"""Context: the same MiniLM-L6-shape forward through PyTorch-ROCm (transformers.BertModel, bf16, SDPA attention) on this GPU."""
import time, torch
from transformers import BertConfig, BertModel
cfg = BertConfig(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12, intermediate_size=1536,
                 max_position_embeddings=512, hidden_act="gelu", layer_norm_eps=1e-12)
m = BertModel(cfg, add_pooling_layer=False).cuda().eval()
ids = torch.randint(1000, 30000, (256, 256), device="cuda"); mask = torch.ones_like(ids)
for dt in (torch.bfloat16, torch.float16):
    mm = m.to(dt)
    with torch.no_grad():
        for _ in range(3): out = mm(input_ids=ids, attention_mask=mask).last_hidden_state
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): out = mm(input_ids=ids, attention_mask=mask).last_hidden_state
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t) / 10
    print(f"transformers.BertModel {dt} 256x256: {dtm*1e3:.2f} ms/batch  {256/dtm:.0f} chunks/s", flush=True)
