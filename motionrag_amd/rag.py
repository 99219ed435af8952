"""Retrieval over the reference-motion database: flat-scan top-k on the GPU.

Host-side mirror of the reference's retrieval interface:

    reference                                              here
    src/data/rag.py:11-80      RAGDatabase.text_search       RAGDatabase.text_search (same arguments / result rows)
    tools/build_rag_database.py:16-74  add_to_db, prepare_annotations   add_to_db, prepare_annotations
    src/data/datamodule.py:231-236     per-annotation search fan-out     RAGDatabase.text_search_batch (one launch)

The reference delegates storage and search to lancedb==0.14.0 (Rust) and embedding to
sentence-transformers (both third-party, not installed here).  This module keeps the table as
`<db_path>/<table_name>/{vectors.npy, meta.json}` (fp32 [N, D] + the reference's row schema) and scores on
the GPU with libmrag_hip.so's `mrag_topk_f32` (sequential-fmaf distances, deterministic ties).  Text
inputs need an `embedder` callable (text -> [D] fp32); the shipped data path always passes embeddings
(datamodule.py:233 hands `anno['text_embedding']`).
"""
from __future__ import annotations

import json
import os
import re
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import ops

_WHERE_RE = re.compile(r"^\s*video\s*!=\s*([\"'])(.*)\1\s*$")
SCHEMA = ("text", "id", "uid", "dataset", "video", "start_sec", "end_sec")


def prepare_annotations(annotations: List[dict], text_name: str = "llm_caption", dataset_name: str = "coin") -> List[dict]:
    """tools/build_rag_database.py:55-74"""
    return [{"text": a[text_name] if a[text_name] is not None else "", "id": a["id"], "uid": dataset_name + "/" + str(a["id"]),
             "dataset": dataset_name, "video": a["video"], "start_sec": a["start_sec"], "end_sec": a["end_sec"]} for a in annotations]


def add_to_db(annotations: List[dict], embeddings: Optional[np.ndarray] = None, embedder: Optional[Callable] = None,
              text_name: str = "llm_caption", db_path: str = "../data/rag.db") -> None:
    """tools/build_rag_database.py:16-52: append rows (+ their text embeddings) to table `text_name`."""
    tdir = os.path.join(db_path, text_name)
    os.makedirs(tdir, exist_ok=True)
    if embeddings is None:
        if embedder is None:
            raise ValueError("add_to_db needs `embeddings` or an `embedder` (sentence-transformers is third-party)")
        embeddings = np.stack([np.asarray(embedder(a["text"]), dtype=np.float32) for a in annotations])
    embeddings = np.ascontiguousarray(embeddings, dtype=np.float32)
    if embeddings.shape[0] != len(annotations):
        raise ValueError("one embedding per annotation")
    vec_path, meta_path = os.path.join(tdir, "vectors.npy"), os.path.join(tdir, "meta.json")
    rows = [{k: a[k] for k in SCHEMA} for a in annotations]
    if os.path.exists(vec_path):
        embeddings = np.concatenate([np.load(vec_path), embeddings], axis=0)
        with open(meta_path) as f:
            rows = json.load(f) + rows
    np.save(vec_path, embeddings)
    with open(meta_path, "w") as f:
        json.dump(rows, f)


class RAGDatabase:
    """src/data/rag.py:11-15; `metric` is LanceDB's: 'l2' (its default without an index; `_distance` is the squared
    L2 distance) or 'dot' (`_distance = 1 - dot`, the metric of the index build_rag_database.py:52 creates)."""

    def __init__(self, db_path: str, table_name: str, device: str = "cuda", metric: str = "l2", embedder: Optional[Callable] = None):
        tdir = os.path.join(db_path, table_name)
        self.vectors_host = np.load(os.path.join(tdir, "vectors.npy"))
        with open(os.path.join(tdir, "meta.json")) as f:
            self.rows = json.load(f)
        self._init_device(device, metric, embedder)

    @classmethod
    def from_arrays(cls, vectors: np.ndarray, rows: List[dict], device: str = "cuda", metric: str = "l2", embedder=None) -> "RAGDatabase":
        self = cls.__new__(cls)
        self.vectors_host = np.ascontiguousarray(vectors, dtype=np.float32)
        self.rows = rows
        self._init_device(device, metric, embedder)
        return self

    def _init_device(self, device, metric, embedder):
        if metric not in ("l2", "dot"):
            raise ValueError(f"Invalid metric: {metric}")
        self.metric, self.embedder = metric, embedder
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ops.HipOnly("RAGDatabase scores on the GPU only (libmrag_hip.so); there is no CPU search path")
        videos = [r["video"] for r in self.rows]
        self.video_ids = {v: i for i, v in enumerate(dict.fromkeys(videos))}
        self.group = torch.tensor([self.video_ids[v] for v in videos], dtype=torch.int32, device=self.device)
        self.vectors = torch.from_numpy(self.vectors_host).to(self.device)   # resident in HBM for every search
        self._plans = {}

    def __len__(self):
        return len(self.rows)

    # ---- result formatting (rag.py:17-34) ----
    def _format(self, rows_idx: np.ndarray, dist: np.ndarray, select: Optional[Sequence[str]], output_format: str):
        cols = list(select) if select is not None else list(SCHEMA)
        out = []
        for r, d in zip(rows_idx.tolist(), dist.tolist()):
            if r < 0:
                continue
            rec = {c: self.rows[r][c] for c in cols}
            rec["_distance"] = d
            out.append(rec)
        if output_format in ("dict", "list"):
            return out
        if output_format == "pandas":
            import pandas as pd
            return pd.DataFrame(out)
        if output_format == "pyarrow":
            import pyarrow as pa
            return pa.Table.from_pylist(out)
        raise ValueError(f"Invalid format: {output_format}")

    def _exclude_id(self, where: Optional[str]) -> int:
        if where is None:
            return -1
        m = _WHERE_RE.match(where)
        if not m:
            raise NotImplementedError(f"only the reference's `video != \"<name>\"` filter is supported, got: {where!r}")
        return self.video_ids.get(m.group(2), -1)

    def _embed(self, text) -> np.ndarray:
        if isinstance(text, str):
            if self.embedder is None:
                raise ValueError("text queries need an `embedder`; the data path passes embeddings (datamodule.py:233)")
            text = self.embedder(text)
        if isinstance(text, torch.Tensor):
            text = text.detach().float().cpu().numpy()
        return np.ascontiguousarray(text, dtype=np.float32).reshape(-1)

    def text_search_batch(self, embeddings, top_k: int = 10, where: Optional[Sequence[Optional[str]]] = None,
                          select: Optional[Sequence[str]] = None, output_format: str = "dict"):
        """all queries of datamodule.py:231-236 in one launch: embeddings [Q, D]; `where` one filter per query."""
        if isinstance(embeddings, torch.Tensor):
            q = embeddings.to(self.device, torch.float32).contiguous()
        else:
            q = torch.from_numpy(np.ascontiguousarray(embeddings, dtype=np.float32)).to(self.device)
        Q = q.shape[0]
        group = exclude = None
        if where is not None and any(w is not None for w in where):
            exclude = torch.tensor([self._exclude_id(w) for w in where], dtype=torch.int32, device=self.device)
            group = self.group
        if Q <= 4 and top_k <= 64:       # the interactive search (rag.py:63-80): a prepared plan -- one C-ABI call = one launch, no allocation
            plan = self._plans.get((Q, top_k))
            if plan is None:
                plan = self._plans[(Q, top_k)] = ops.TopkPlan(self.vectors, Q, top_k, metric=self.metric, group=self.group)
            plan.queries.copy_(q)
            if exclude is not None:
                plan.exclude.copy_(exclude)
            else:
                plan.exclude.fill_(-1)
            rows, dist = plan.run()
        else:
            rows, dist = ops.topk(self.vectors, q, top_k, metric=self.metric, group=group, exclude=exclude)
        rows, dist = rows.cpu().numpy(), dist.cpu().numpy()
        return [self._format(rows[i], dist[i], select, output_format) for i in range(Q)]

    def vector_search(self, vector, vector_column_name: str = None, top_k: int = 10, table=None, where: str = None,
                      select: List[str] = None, nprobes: int = 50, refine_factor: int = 30, output_format: str = "dict"):
        """rag.py:36-61 (flat scan: nprobes / refine_factor only matter for LanceDB's IVF index and are accepted for
        signature compatibility)."""
        if vector_column_name not in (None, "text_embedding"):
            raise NotImplementedError("only the text_embedding column is on the shipped path (ref_video_type: rag_text)")
        if table is not None:
            raise NotImplementedError("temporary tables are only used by text_image_search (not on the shipped path)")
        emb = self._embed(vector)[None]
        return self.text_search_batch(emb, top_k, [where], select, output_format)[0]

    def text_search(self, text, top_k: int = 10, table=None, where: str = None, select: List[str] = None, nprobes: int = 50,
                    refine_factor: int = 30, output_format: str = "dict"):
        """rag.py:63-80"""
        return self.vector_search(text, vector_column_name="text_embedding", top_k=top_k, table=table, where=where, select=select,
                                  nprobes=nprobes, refine_factor=refine_factor, output_format=output_format)


# ---------------------------------------------------------------------------------------------- callers either side of the search
def attach_ref_videos(annotations: List[dict], db: "RAGDatabase", ref_video_num: int = 9, *, ref_video_type: str = "rag_text",
                      chunk: int = 256) -> List[dict]:
    """The RAG fan-out of src/data/datamodule.py:225-265 (`ref_video_type == 'rag_text'`): every annotation searches with its own
    `text_embedding`, over-fetches `ref_video_num + 3` rows (:234), excludes its own video (`where = 'video != "<self>"'`, :235) and keeps
    `['video', 'start_sec', 'end_sec']` (+ `_distance`) in `anno['ref_videos']`.  The reference runs one LanceDB query per annotation in a
    64-process pool; here the queries of a chunk are ONE batched top-k launch."""
    if ref_video_type != "rag_text":
        raise NotImplementedError("only the shipped `ref_video_type: rag_text` (configs/*/MotionRAG_open.yml) runs on the GPU scan")
    for i in range(0, len(annotations), chunk):
        part = annotations[i:i + chunk]
        q = np.stack([np.asarray(a["text_embedding"], dtype=np.float32) for a in part])
        res = db.text_search_batch(q, top_k=ref_video_num + 3, where=[f'video != "{a["video"]}"' for a in part], select=["video", "start_sec", "end_sec"])
        for a, r in zip(part, res):
            a["ref_videos"] = r
    return annotations


def get_ref_videos(video_info: dict, video: torch.Tensor, load_clip: Callable[[dict], torch.Tensor], ref_video_num: int = 9,
                   video_length: Optional[int] = None, uncond_video_ratio: float = 0.0, rng=None):
    """The consumer contract of src/data/dataset.py:285-312 (`VideoDataset.get_ref_videos`): video [1, T, C, H, W] ->
    (ref_videos [K, T, C, H, W], distance list).  The first `ref_video_num` retrieved rows are used (most similar first); a reference that
    IS the target clip re-uses `video`; a dropped (unconditional-training) reference or one whose clip fails to load leaves a ZERO video and
    distance 1.0 (:292, :306-310).  `load_clip(row) -> [1, T, C, H, W]` stands for the dataset's video reader (out of scope)."""
    import random as _random
    rng = rng or _random
    T = video_length if video_length is not None else video.shape[1]
    ref_videos = torch.zeros(ref_video_num, T, *video.shape[2:], dtype=video.dtype, device=video.device)
    distance: List[float] = []
    for i, v in enumerate(video_info.get("ref_videos", [])[:ref_video_num]):
        if rng.random() > uncond_video_ratio:
            try:
                ref = video if v["video"] == video_info["video"] else load_clip(v)
                ref_videos[i] = ref
                distance.append(v["_distance"])
            except Exception as e:          # noqa: BLE001 -- the reference swallows reader errors the same way
                print(f"Rag read video Error: {e}")
                distance.append(1.0)
        else:
            distance.append(1.0)
    return ref_videos, distance


_ADJ = ("slow", "fast", "shaky", "smooth", "sudden", "gentle", "circular", "zigzag", "rising", "falling")
_NOUN = ("camera", "person", "dog", "car", "bird", "wave", "hand", "crowd", "leaf", "train")
_VERB = ("pans left", "pans right", "zooms in", "zooms out", "walks forward", "turns around", "jumps", "rotates", "drifts", "stops")


def synthetic_captions(n: int = 10000) -> List[dict]:
    """BASELINE config #1 / SURVEY 8d: `n` synthetic motion captions `clip {i}: a {adj} {noun} {verb}` in the reference's annotation schema"""
    return [{"motion_caption": f"clip {i}: a {_ADJ[i % 10]} {_NOUN[(i // 10) % 10]} {_VERB[(i // 100) % 10]}", "id": i, "video": f"clip_{i:06d}.mp4",
             "start_sec": 0.0, "end_sec": 4.0} for i in range(n)]


def hash_embedder(dim: int = 768) -> Callable[[str], np.ndarray]:
    """Offline stand-in for sentence-transformers' gte-base-en-v1.5 (third-party weights, no network): a unit-normalised N(0, 1) vector
    seeded by the caption's hash -- deterministic, `dim`-dimensional, fp32."""
    import hashlib

    def embed(text: str) -> np.ndarray:
        seed = int.from_bytes(hashlib.sha256(text.encode("utf-8")).digest()[:8], "little")
        v = np.random.default_rng(seed).standard_normal(dim).astype(np.float32)
        return v / np.linalg.norm(v)

    return embed
