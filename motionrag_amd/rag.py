"""Retrieval over the reference-motion database: flat-scan top-k on the GPU.

Host-side mirror of the reference's retrieval interface:

    reference                                              here
    src/data/rag.py:11-80      RAGDatabase.text_search       RAGDatabase.text_search (same arguments / result rows)
    tools/build_rag_database.py:16-74  add_to_db, prepare_annotations   add_to_db, prepare_annotations
    src/data/datamodule.py:231-236     per-annotation search fan-out     RAGDatabase.text_search_batch (one launch)

The reference delegates storage and search to lancedb==0.14.0 (Rust) and embedding to
sentence-transformers (both third-party, not installed here).  This module keeps the table as
`<db_path>/<table_name>/{vectors.npy, meta.arrow}`: fp32 [N, D] read through a memory map and uploaded to HBM in
chunks (the host never holds a second copy), and the reference's row schema as an Arrow IPC file that is memory-mapped
and read LAZILY -- only the `video` column is touched when the table opens (dictionary-encoded into the int32 group ids
of the `video != self` filter); result rows are `take`n from the mapped columns.  Sized for what the scan was measured
at (10^7 rows: 30.7 GB of vectors, no per-row Python objects); tables written by round 2 (`meta.json`) still open.
Scoring: libmrag_hip.so's `mrag_topk_f32` (sequential-fmaf distances, deterministic ties).

Filter order.  The reference builds `table.search(v).limit(k)...where(where)` (src/data/rag.py:54-58), i.e. lancedb 0.14.0's
`LanceQueryBuilder.where(where, prefilter=False)` (lancedb/query.py of that release: "prefilter: bool, default False -- if True, apply the
filter before vector search, otherwise the filter is applied on the result of vector search"; the default only became True in later
releases).  That is a POST-filter: the k nearest rows are taken first, rows failing `video != "<self>"` are then dropped, and FEWER than k
rows can come back.  `RAGDatabase(prefilter=False)` (the default) reproduces this order in the kernel's final merge; `prefilter=True`
excludes rows before selection (always k results while k rows pass).  lancedb is not installed here, so the default is cited, not executed:
if a deployment's lancedb applies the filter first, construct with `prefilter=True`.  The reference over-fetches `ref_video_num + 3`
(datamodule.py:234) and keeps the first `ref_video_num` (dataset.py:296): the two orders differ for a query only when more than 3 of its 12
nearest rows are clips of its own video; `get_ref_videos` then pads with zero videos (and, like the reference, a shorter distance list).
Text
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


_ARROW_TYPES = {"text": "string", "id": "int64", "uid": "string", "dataset": "string", "video": "string", "start_sec": "float64", "end_sec": "float64"}
UPLOAD_CHUNK_BYTES = 256 << 20


def _rows_to_table(rows: List[dict]):
    import pyarrow as pa
    cols = {}
    for k in SCHEMA:
        vals = [r.get(k) for r in rows]                  # partial rows (tests, ad-hoc tables): missing fields are nulls
        try:
            cols[k] = pa.array(vals, type=getattr(pa, _ARROW_TYPES[k])())
        except (pa.ArrowInvalid, pa.ArrowTypeError):      # e.g. string ids: keep what the annotations carry
            cols[k] = pa.array(vals)
    return pa.table(cols)


def _read_meta(tdir: str):
    """the table's metadata as a (memory-mapped, zero-copy) Arrow table; round-2 tables carry meta.json instead"""
    import pyarrow as pa
    apath = os.path.join(tdir, "meta.arrow")
    if os.path.exists(apath):
        return pa.ipc.open_file(pa.memory_map(apath, "r")).read_all()
    with open(os.path.join(tdir, "meta.json")) as f:
        return _rows_to_table(json.load(f))


def add_to_db(annotations: List[dict], embeddings: Optional[np.ndarray] = None, embedder: Optional[Callable] = None,
              text_name: str = "llm_caption", db_path: str = "../data/rag.db") -> None:
    """tools/build_rag_database.py:16-52: append rows (+ their text embeddings) to table `text_name`.  Needs pyarrow >= 14 (as RAGDatabase does).
    Append order: vectors.npy is replaced first, meta.arrow second (two atomic renames); a reader that opens the table between them -- or after a
    crash there -- sees more vectors than rows and opens the table as it was before the append (RAGDatabase.__init__)."""
    import pyarrow as pa
    tdir0 = os.path.join(db_path, text_name)
    if os.path.exists(os.path.join(tdir0, "vectors.npy")) and os.path.exists(os.path.join(tdir0, "meta.arrow")):
        n_vec, n_meta = np.load(os.path.join(tdir0, "vectors.npy"), mmap_mode="r").shape[0], _read_meta(tdir0).num_rows
        if n_vec > n_meta:                                  # an interrupted append: drop its orphan vectors before appending again
            keep = np.ascontiguousarray(np.load(os.path.join(tdir0, "vectors.npy"), mmap_mode="r")[:n_meta])
            np.save(os.path.join(tdir0, "vectors.npy.repair.npy"), keep)
            os.replace(os.path.join(tdir0, "vectors.npy.repair.npy"), os.path.join(tdir0, "vectors.npy"))
    tdir = os.path.join(db_path, text_name)
    os.makedirs(tdir, exist_ok=True)
    if embeddings is None:
        if embedder is None:
            raise ValueError("add_to_db needs `embeddings` or an `embedder` (sentence-transformers is third-party)")
        embeddings = np.stack([np.asarray(embedder(a["text"]), dtype=np.float32) for a in annotations])
    embeddings = np.ascontiguousarray(embeddings, dtype=np.float32)
    if embeddings.shape[0] != len(annotations):
        raise ValueError("one embedding per annotation")
    vec_path, meta_path = os.path.join(tdir, "vectors.npy"), os.path.join(tdir, "meta.arrow")
    table = _rows_to_table(annotations)
    if os.path.exists(vec_path):                            # append: the old rows are streamed through the page cache, never loaded whole
        old = np.load(vec_path, mmap_mode="r")
        if old.shape[1] != embeddings.shape[1]:
            raise ValueError(f"table holds {old.shape[1]}-dimensional vectors, got {embeddings.shape[1]}")
        tmp = vec_path + ".tmp.npy"
        out = np.lib.format.open_memmap(tmp, mode="w+", dtype=np.float32, shape=(old.shape[0] + embeddings.shape[0], old.shape[1]))
        step = max(1, UPLOAD_CHUNK_BYTES // (4 * old.shape[1]))
        for i in range(0, old.shape[0], step):
            j = min(i + step, old.shape[0])
            out[i:j] = old[i:j]
        out[old.shape[0]:] = embeddings
        out.flush()
        del out, old
        os.replace(tmp, vec_path)
        table = pa.concat_tables([_read_meta(tdir), table], promote_options="default")
    else:
        np.save(vec_path, embeddings)
    tmp = meta_path + ".tmp"
    with pa.OSFile(tmp, "wb") as sink, pa.ipc.new_file(sink, table.schema) as w:
        w.write_table(table)
    os.replace(tmp, meta_path)
    legacy = os.path.join(tdir, "meta.json")
    if os.path.exists(legacy):
        os.remove(legacy)


class RAGDatabase:
    """src/data/rag.py:11-15; `metric` is LanceDB's: 'l2' (its default without an index; `_distance` is the squared
    L2 distance) or 'dot' (`_distance = 1 - dot`, the metric of the index build_rag_database.py:52 creates)."""

    def __init__(self, db_path: str, table_name: str, device: str = "cuda", metric: str = "l2", embedder: Optional[Callable] = None,
                 prefilter: bool = False):
        self.prefilter = bool(prefilter)          # lancedb's `where(..., prefilter=)`; False = its 0.14.0 default (module docstring)
        tdir = os.path.join(db_path, table_name)
        self.vectors_host = np.load(os.path.join(tdir, "vectors.npy"), mmap_mode="c")      # a copy-on-write view of the file (never written): pages stream through on upload
        self.meta = _read_meta(tdir)
        if self.vectors_host.shape[0] > self.meta.num_rows:
            # add_to_db appends by replacing vectors.npy and THEN meta.arrow: a crash (or an open) between the two renames leaves the new vectors
            # without their rows.  The table is append-only, so its first `num_rows` vectors are exactly the table before that append: open that.
            import warnings
            warnings.warn(f"{tdir}: {self.vectors_host.shape[0]} vectors but {self.meta.num_rows} metadata rows -- an interrupted add_to_db; "
                          f"opening the {self.meta.num_rows} complete rows (re-run the append)")
            self.vectors_host = self.vectors_host[:self.meta.num_rows]
        self._init_device(device, metric, embedder)

    @classmethod
    def from_arrays(cls, vectors: np.ndarray, rows, device: str = "cuda", metric: str = "l2", embedder=None, prefilter: bool = False) -> "RAGDatabase":
        """`rows`: list of row dicts in the reference's schema, or an Arrow table with those columns"""
        self = cls.__new__(cls)
        self.prefilter = bool(prefilter)
        self.vectors_host = vectors if (isinstance(vectors, np.ndarray) and vectors.dtype == np.float32) else np.ascontiguousarray(vectors, dtype=np.float32)
        self.meta = _rows_to_table(rows) if isinstance(rows, list) else rows
        self._init_device(device, metric, embedder)
        return self

    def _init_device(self, device, metric, embedder):
        import pyarrow as pa
        import pyarrow.compute as pc
        if metric not in ("l2", "dot"):
            raise ValueError(f"Invalid metric: {metric}")
        self.metric, self.embedder = metric, embedder
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ops.HipOnly("RAGDatabase scores on the GPU only (libmrag_hip.so); there is no CPU search path")
        N, D = self.vectors_host.shape
        if self.meta.num_rows != N:
            raise ValueError(f"{N} vectors but {self.meta.num_rows} metadata rows")
        # group id of a row = index of its video among the distinct videos in order of first appearance (the `video != "<name>"` filter of
        # datamodule.py:235 compares ids in the kernel): one dictionary encoding of ONE column, no per-row Python objects
        enc = pc.dictionary_encode(self.meta.column("video").combine_chunks() if self.meta.num_rows else pa.array([], pa.string()))
        self._videos = enc.dictionary
        self.group = torch.from_numpy(np.array(enc.indices.to_numpy(zero_copy_only=False), dtype=np.int32)).to(self.device)     # np.array: a writable copy
        self.vectors = torch.empty(N, D, dtype=torch.float32, device=self.device)          # resident in HBM for every search
        step = max(1, UPLOAD_CHUNK_BYTES // (4 * D))
        for i in range(0, N, step):                                                        # chunked: the host side is the page cache of the mapped file
            self.vectors[i:i + step].copy_(torch.from_numpy(np.ascontiguousarray(self.vectors_host[i:i + step])))
        self._plans = {}
        import threading
        self._lock = threading.Lock()

    def __len__(self):
        return self.meta.num_rows

    @property
    def rows(self) -> List[dict]:
        """every row as a dict (debugging / small tables; searches never build this)"""
        return self.meta.to_pylist()

    # ---- result formatting (rag.py:17-34) ----
    def _format(self, rows_idx: np.ndarray, dist: np.ndarray, select: Optional[Sequence[str]], output_format: str):
        import pyarrow as pa
        cols = list(select) if select is not None else list(SCHEMA)
        keep = rows_idx >= 0
        hit = self.meta.select(cols).take(pa.array(rows_idx[keep].astype(np.int64)))      # only the requested columns of the hit rows leave the map
        hit = hit.append_column("_distance", pa.array(dist[keep].astype(np.float64)))
        if output_format in ("dict", "list"):
            return hit.to_pylist()
        if output_format == "pandas":
            return hit.to_pandas()
        if output_format == "pyarrow":
            return hit
        raise ValueError(f"Invalid format: {output_format}")

    def _exclude_id(self, where: Optional[str]) -> int:
        """`where` of the reference is an SQL string handed to LanceDB (src/data/rag.py:36-61); the one filter the shipped data path builds is
        `video != "<name>"` (datamodule.py:235), evaluated here as an id comparison inside the scan.  Anything else is refused up front."""
        if where is None:
            return -1
        m = _WHERE_RE.match(where)
        if not m:
            raise ValueError(f"unsupported `where` filter {where!r}: this scan evaluates only the reference's own `video != \"<name>\"` "
                             f"(src/data/datamodule.py:235); a general SQL predicate needs LanceDB")
        import pyarrow as pa
        import pyarrow.compute as pc
        idx = pc.index_in(pa.scalar(m.group(2)), value_set=self._videos).as_py()
        return -1 if idx is None else int(idx)

    def _exclude_ids(self, where: Sequence[Optional[str]]) -> List[int]:
        """one id per query; the names of a batch are looked up in ONE pass over the distinct videos"""
        import pyarrow as pa
        import pyarrow.compute as pc
        names = []
        for w in where:
            if w is None:
                names.append(None)
                continue
            m = _WHERE_RE.match(w)
            if not m:
                self._exclude_id(w)          # raises the descriptive error
            names.append(m.group(2))
        idx = pc.index_in(pa.array(names, pa.string()), value_set=self._videos).to_pylist()
        return [-1 if (i is None or n is None) else int(i) for i, n in zip(idx, names)]

    def _embed(self, text) -> np.ndarray:
        if isinstance(text, str):
            if self.embedder is None:
                raise ValueError("text queries need an `embedder`; the data path passes embeddings (datamodule.py:233)")
            text = self.embedder(text)
        if isinstance(text, torch.Tensor):
            text = text.detach().float().cpu().numpy()
        return np.ascontiguousarray(text, dtype=np.float32).reshape(-1)

    def text_search_batch(self, embeddings, top_k: int = 10, where: Optional[Sequence[Optional[str]]] = None,
                          select: Optional[Sequence[str]] = None, output_format: str = "dict", prefilter: Optional[bool] = None):
        """all queries of datamodule.py:231-236 in one launch: embeddings [Q, D]; `where` one filter per query.  `prefilter` overrides the
        database's filter order for this call (None = `self.prefilter`); with the post-filter a query may return fewer than `top_k` rows."""
        post = not (self.prefilter if prefilter is None else prefilter)
        if isinstance(embeddings, torch.Tensor):
            q = embeddings.to(self.device, torch.float32).contiguous()
        else:
            q = torch.from_numpy(np.ascontiguousarray(embeddings, dtype=np.float32)).to(self.device)
        Q = q.shape[0]
        group = exclude = None
        if where is not None and any(w is not None for w in where):
            exclude = torch.tensor(self._exclude_ids(where), dtype=torch.int32, device=self.device)       # every filter parsed BEFORE anything launches
            group = self.group
        if Q <= 4 and top_k <= 64:       # the interactive search (rag.py:63-80): a prepared plan -- one C-ABI call = one launch, no allocation
            # a plan owns its workspace, arrival counters included: one per (Q, k, STREAM) so that searches issued from different streams never share
            # counters, and a lock so that two host threads on one stream cannot interleave the copy-in / launch / read-out of one plan
            key = (Q, top_k, torch.cuda.current_stream(self.device).cuda_stream, post)
            with self._lock:
                plan = self._plans.get(key)
                if plan is None:
                    plan = self._plans[key] = ops.TopkPlan(self.vectors, Q, top_k, metric=self.metric, group=self.group, postfilter=post)
                plan.queries.copy_(q)
                if exclude is not None:
                    plan.exclude.copy_(exclude)
                else:
                    plan.exclude.fill_(-1)
                rows, dist = plan.run()
                rows, dist = rows.cpu().numpy(), dist.cpu().numpy()
            self._note_short(rows, top_k, post, exclude is not None)
            return [self._format(rows[i], dist[i], select, output_format) for i in range(Q)]
        else:
            rows, dist = ops.topk(self.vectors, q, top_k, metric=self.metric, group=group, exclude=exclude, postfilter=post)
        rows, dist = rows.cpu().numpy(), dist.cpu().numpy()
        self._note_short(rows, top_k, post, exclude is not None)
        return [self._format(rows[i], dist[i], select, output_format) for i in range(Q)]

    def _note_short(self, rows: np.ndarray, top_k: int, post: bool, filtered: bool) -> None:
        """ONE warning per database when the post-filter order hands back fewer than `top_k` rows: a deployment whose LanceDB pre-filters would have
        got `top_k` there (`RAGDatabase(prefilter=True)` is that order)."""
        if post and filtered and top_k <= len(self) and not getattr(self, "_short_noted", False) and (rows < 0).any():
            import warnings
            self._short_noted = True
            short = int((rows < 0).any(axis=1).sum())
            warnings.warn(f"RAGDatabase: {short} of {rows.shape[0]} filtered searches returned fewer than top_k={top_k} rows -- the `where` filter is applied AFTER the "
                          "k nearest rows were taken (lancedb 0.14.0's `where(..., prefilter=False)`, src/data/rag.py:57-58); the consumer pads the missing "
                          "references with zero clips.  RAGDatabase(prefilter=True) filters before the selection.  (reported once per database)")

    def vector_search(self, vector, vector_column_name: str = None, top_k: int = 10, table=None, where: str = None,
                      select: List[str] = None, nprobes: int = 50, refine_factor: int = 30, output_format: str = "dict",
                      prefilter: Optional[bool] = None):
        """rag.py:36-61 (flat scan: nprobes / refine_factor only matter for LanceDB's IVF index and are accepted for
        signature compatibility)."""
        if vector_column_name not in (None, "text_embedding"):
            raise NotImplementedError("only the text_embedding column is on the shipped path (ref_video_type: rag_text)")
        if table is not None:
            raise NotImplementedError("temporary tables are only used by text_image_search (not on the shipped path)")
        emb = self._embed(vector)[None]
        return self.text_search_batch(emb, top_k, [where], select, output_format, prefilter=prefilter)[0]

    def text_search(self, text, top_k: int = 10, table=None, where: str = None, select: List[str] = None, nprobes: int = 50,
                    refine_factor: int = 30, output_format: str = "dict", prefilter: Optional[bool] = None):
        """rag.py:63-80 (`prefilter`: not a reference argument; None = the database's filter order)"""
        return self.vector_search(text, vector_column_name="text_embedding", top_k=top_k, table=table, where=where, select=select,
                                  nprobes=nprobes, refine_factor=refine_factor, output_format=output_format, prefilter=prefilter)


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
