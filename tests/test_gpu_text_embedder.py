"""The retrieval row's text embedder (gte-base-en-v1.5's NewModel: third-party remote code, parity UNPINNED) on the GPU against the fp32 restatement
oracle/gte_ref.py: a reduced model at several unpadded lengths, the base configuration (12 x 768, 136.8 M parameters), the sentence embedder's length grouping,
and the embedder plugged into RAGDatabase.text_search."""
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
