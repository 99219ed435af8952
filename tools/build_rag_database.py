#!/usr/bin/env python3
"""Build the retrieval table the way tools/build_rag_database.py:77-88 of the reference does -- `prepare_annotations` -> `add_to_db` --
over an annotation file or, with no file (BASELINE config #1: "text-embedding + cosine top-k over 10k synthetic captions"), over 10 000
synthetic captions with hash-seeded unit embeddings (the gte-base model is third-party and there is no network).  Then run the reference's
RAG fan-out (src/data/datamodule.py:225-265) as batched GPU top-k and print a summary.

    python tools/build_rag_database.py --db_path /tmp/rag/openvid.db --dataset openvid --caption_name motion_caption [--annotations_path x.pt]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from motionrag_amd import rag  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--db_path", type=str, default="/tmp/motionrag_rag/openvid.db")
    ap.add_argument("--dataset", type=str, default="openvid")
    ap.add_argument("--annotations_path", type=str, default=None, help=".pt list of annotation dicts (torch.load), as the reference; default: 10 000 synthetic captions")
    ap.add_argument("--caption_name", type=str, default="motion_caption")
    ap.add_argument("--n_synthetic", type=int, default=10000)
    ap.add_argument("--ref_video_num", type=int, default=9)
    ap.add_argument("--gte_dir", type=str, default=None, help="local snapshot of Alibaba-NLP/gte-base-en-v1.5 (model.safetensors + tokenizer files): captions are embedded by "
                    "motionrag_amd.text_embedder on the GPU, as the reference's LanceDB embedding function does (tools/build_rag_database.py:28-31); default: hash-seeded unit vectors")
    args = ap.parse_args()

    annotations = torch.load(args.annotations_path) if args.annotations_path else rag.synthetic_captions(args.n_synthetic)
    t0 = time.perf_counter()
    if args.gte_dir:
        import safetensors.torch
        from transformers import AutoTokenizer
        from motionrag_amd.text_embedder import NewModel, SentenceEmbedder
        cfg = {}
        if os.path.exists(os.path.join(args.gte_dir, "config.json")):      # the snapshot's own geometry (absent: the published gte-base-en-v1.5 one)
            import json
            with open(os.path.join(args.gte_dir, "config.json")) as f:
                hf = json.load(f)
            cfg = {k: hf[k] for k in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "layer_norm_eps",
                                      "max_position_embeddings", "rope_theta") if k in hf}
            if isinstance(hf.get("rope_scaling"), dict) and "factor" in hf["rope_scaling"]:
                cfg["rope_scaling_factor"] = float(hf["rope_scaling"]["factor"])
        model = NewModel(**cfg)
        sd = safetensors.torch.load_file(os.path.join(args.gte_dir, "model.safetensors"))
        model.load_state_dict({k[len("new."):] if k.startswith("new.") else k: v for k, v in sd.items() if "position_ids" not in k and not k.startswith("pooler")}, strict=True)
        tok = AutoTokenizer.from_pretrained(args.gte_dir)
        embedder = SentenceEmbedder(model.to("cuda", torch.bfloat16), lambda texts: tok(list(texts), truncation=True, max_length=8192)["input_ids"])
        texts = [a[args.caption_name] or "" for a in annotations]
        emb = torch.cat([embedder.encode(texts[i:i + 4096]) for i in range(0, len(texts), 4096)]).cpu().numpy()
    else:
        embed = rag.hash_embedder(768)
        emb = np.stack([np.asarray(a["text_embedding"], np.float32) if "text_embedding" in a else embed(a[args.caption_name] or "") for a in annotations])
    rows = rag.prepare_annotations(annotations, text_name=args.caption_name, dataset_name=args.dataset)
    rag.add_to_db(rows, emb, text_name=args.caption_name, db_path=args.db_path)
    t1 = time.perf_counter()
    db = rag.RAGDatabase(args.db_path, args.caption_name, device="cuda")
    for a, e in zip(annotations, emb):
        a["text_embedding"] = e
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rag.attach_ref_videos(annotations, db, ref_video_num=args.ref_video_num)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    d = np.array([[r["_distance"] for r in a["ref_videos"]] for a in annotations])
    print(f"table {args.caption_name}: {len(db)} rows x {emb.shape[1]} (built in {t1 - t0:.2f} s); "
          f"{len(annotations)} queries x top-{args.ref_video_num + 3} with self-exclusion in {t3 - t2:.3f} s "
          f"({len(annotations) / (t3 - t2):.0f} queries/s); mean nearest _distance {d[:, 0].mean():.4f}, all self-free: "
          f"{all(r['video'] != a['video'] for a in annotations for r in a['ref_videos'])}")
