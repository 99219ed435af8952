"""The retrieval row's text embedder (gte-base-en-v1.5's NewModel: third-party remote code, parity UNPINNED) on the GPU against the fp32 restatement
oracle/gte_ref.py: a reduced model at several unpadded lengths, the base configuration (12 x 768, 136.8 M parameters), the sentence embedder's length grouping,
and the embedder plugged into RAGDatabase.text_search."""
import os

import numpy as np
import pytest
import torch

from oracle import gte_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
SMALL = dict(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, layer_norm_eps=1e-12, max_position_embeddings=8192,
             rope_theta=500000.0, rope_scaling_factor=2.0)


def rel(got, want):
    g, w = got.float().cpu(), want.float().cpu()
    assert g.shape == w.shape and torch.isfinite(g).all()
    return ((g - w).norm() / w.norm()).item()


def _model(cfg, seed):
    from motionrag_amd.text_embedder import NewModel
    sd = {k: v.to(torch.bfloat16).float() for k, v in R.seeded_state(cfg, seed).items()}
    m = NewModel(**cfg)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV, torch.bfloat16), sd


@pytest.mark.parametrize("B,S", [(1, 5), (3, 37), (2, 200), (1, 700)])
def test_reduced_model_matches_oracle(hip, B, S):
    m, sd = _model(SMALL, 3)
    ids = torch.randint(1, 512, (B, S), generator=torch.Generator().manual_seed(S))
    want = R.encoder(sd, SMALL, ids)
    got = m(ids.to(DEV)).last_hidden_state
    assert got.dtype == torch.bfloat16 and rel(got, want) < 2e-2
    assert rel(torch.nn.functional.normalize(got[:, 0].float(), dim=1), R.sentence_embedding(sd, SMALL, ids)) < 1.5e-2


def test_base_configuration_matches_oracle(hip):
    m, sd = _model(R.CONFIG_BASE, 4)
    assert sum(p.numel() for p in m.parameters()) == 136_776_192
    ids = torch.randint(1, 30528, (3, 24), generator=torch.Generator().manual_seed(1))
    want = R.sentence_embedding(sd, R.CONFIG_BASE, ids)
    got = torch.nn.functional.normalize(m(ids.to(DEV))[0][:, 0].float(), dim=1)
    assert rel(got, want) < 4e-2                                               # 24 post-norm sublayers with bf16 activations (the reference runs the model in bf16 too): 2.7 % measured
    assert (got.cpu() * want).sum(1).min().item() > 0.999                      # cosine between the two embeddings of every sentence


def test_sentence_embedder_groups_by_length_and_feeds_text_search(hip, tmp_path):
    from motionrag_amd import rag
    from motionrag_amd.text_embedder import SentenceEmbedder
    m, sd = _model(SMALL, 5)

    def tok(texts):                                                            # stand-in WordPiece: [CLS] = 101 % 512, one id per word, [SEP]
        return [[101] + [2 + (sum(map(ord, w)) % 500) for w in t.split()] + [102] for t in texts]
    emb = SentenceEmbedder(m, tok)
    texts = ["a dog runs", "pour the milk into the bowl", "a cat sleeps", "cut the onion", "open the door slowly and walk in", "stir"]
    E = emb.encode(texts)
    assert E.shape == (6, 128) and torch.allclose(E.norm(dim=1), torch.ones(6, device=DEV), atol=1e-5)
    for i, t in enumerate(texts):                                              # grouped batches == one text at a time == the oracle
        ids = torch.tensor(tok([t]))
        assert rel(E[i:i + 1], R.sentence_embedding(sd, SMALL, ids)) < 1.5e-2
        assert np.allclose(emb(t), E[i].cpu().numpy(), atol=2e-2)
    annos = [{"llm_caption": t, "id": i, "video": f"v{i}.mp4", "start_sec": 0.0, "end_sec": 1.0} for i, t in enumerate(texts)]
    rag.add_to_db(rag.prepare_annotations(annos), embedder=emb, db_path=str(tmp_path))
    db = rag.RAGDatabase(str(tmp_path), "llm_caption", device="cuda", embedder=emb)
    hits = db.text_search("cut the onion", top_k=2)
    assert hits[0]["text"] == "cut the onion" and hits[0]["_distance"] < 1e-3


def test_build_rag_database_from_a_local_gte_snapshot(hip, tmp_path):
    """tools/build_rag_database.py --gte_dir (the reference's build path: captions embedded by the table's embedding function,
    tools/build_rag_database.py:16-52 there): a synthetic local snapshot -- config.json, a random `model.safetensors` under the checkpoint's `new.`
    key prefix, a WordPiece tokenizer saved by `transformers` -- is loaded, 300 synthetic captions are embedded on the GPU, written, and searched"""
    import json
    import subprocess
    import sys
    import safetensors.torch
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    from motionrag_amd import rag
    from motionrag_amd.text_embedder import NewModel
    snap = tmp_path / "gte"
    os.makedirs(snap)
    words = sorted({w for a in rag.synthetic_captions(300) for w in a["motion_caption"].replace(":", " : ").split()})
    vocab = list(dict.fromkeys(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", ":"] + words + [f"##{d}" for d in "0123456789"] + list("0123456789")))
    tk = Tokenizer(models.WordPiece({w: i for i, w in enumerate(vocab)}, unk_token="[UNK]"))          # a BERT-style WordPiece tokenizer, as the checkpoint ships
    tk.normalizer = normalizers.BertNormalizer(lowercase=True)
    tk.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tk.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", 2), ("[SEP]", 3)])
    PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]",
                            mask_token="[MASK]").save_pretrained(str(snap))
    cfg = dict(vocab_size=len(vocab) + 7, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, layer_norm_eps=1e-12,
               max_position_embeddings=512, rope_theta=500000.0, rope_scaling={"type": "ntk", "factor": 2.0})
    with open(snap / "config.json", "w") as f:
        json.dump(cfg, f)
    torch.manual_seed(0)
    model = NewModel(**{k: v for k, v in cfg.items() if k != "rope_scaling"}, rope_scaling_factor=2.0)
    sd = {"new." + k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}
    sd["new.embeddings.position_ids"] = torch.arange(512)[None]                  # buffers of the published checkpoint that the loader must skip
    sd["pooler.dense.weight"] = torch.zeros(4, 4)
    safetensors.torch.save_file(sd, str(snap / "model.safetensors"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "build_rag_database.py"), "--db_path", str(tmp_path / "o.db"), "--n_synthetic", "300",
                          "--gte_dir", str(snap)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = out.stdout.decode().strip().splitlines()[-1]
    assert "300 rows x 128" in line and "all self-free: True" in line, line
    db = rag.RAGDatabase(str(tmp_path / "o.db"), "motion_caption", device="cuda")
    v = np.asarray(db.vectors_host)
    np.testing.assert_allclose(np.linalg.norm(v, axis=1), 1.0, atol=2e-2)         # L2-normalised sentence embeddings (bf16 model, fp32 normalisation)
    assert db.text_search(text=v[17], top_k=1)[0]["id"] == 17
