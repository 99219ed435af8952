"""TEST INFRASTRUCTURE -- tests/golden/t5.npz from the REAL `transformers.T5EncoderModel` (random init, reduced T5 v1.1 config with head_dim 64).

    python -m oracle.gen_golden_t5"""
import os

import numpy as np
import torch


def main():
    import transformers
    from transformers import T5Config, T5EncoderModel
    torch.manual_seed(4321)
    cfg = T5Config(vocab_size=200, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2, relative_attention_num_buckets=32, relative_attention_max_distance=128,
                   feed_forward_proj="gated-gelu", layer_norm_epsilon=1e-6, dropout_rate=0.0, tie_word_embeddings=False)
    m = T5EncoderModel(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            p.copy_(p.to(torch.bfloat16).float())                       # bf16-exact weights: shared bit for bit with the HIP path
    ids = torch.randint(0, 200, (2, 226))
    ids[:, 150:] = 0                                                    # padded to max_length like the pipeline's tokenizer call (pad id 0), NO attention mask
    with torch.no_grad():
        y = m(input_ids=ids).last_hidden_state
        mask = torch.ones(2, 226, dtype=torch.long); mask[0, 150:] = 0; mask[1, 90:] = 0
        y_masked = m(input_ids=ids, attention_mask=mask).last_hidden_state
    out = {"transformers_version": np.array(transformers.__version__), "ids": ids.numpy(), "y": y.numpy(), "mask": mask.numpy(), "y_masked": y_masked.numpy(),
           "cfg": np.array([128, 2, 64, 256, 2, 200], dtype=np.int64)}
    for k, v in m.state_dict().items():
        out["sd." + k] = v.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "t5.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", sorted(k for k in out if k.startswith("sd."))[:4])


if __name__ == "__main__":
    main()
