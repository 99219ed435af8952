"""CPU suite: the C-ABI library builds, loads and exports every symbol include/mrag_hip.h declares (no compute calls),
host-side logic (scheduler tables, RoPE tables, masks, retrieval table I/O), and the product path's refusal to run
without the GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from motionrag_amd import _lib
    _lib.build()
    hdr = open(os.path.join(ROOT, "include", "mrag_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(mrag_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(_lib.SYMBOLS), (declared, sorted(_lib.SYMBOLS))
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    lib = _lib.lib()
    assert lib.mrag_abi_version() == _lib.ABI_VERSION and lib.mrag_target_arch() == b"gfx950"
    assert lib.mrag_topk_workspace_bytes(10000, 256) > 0
    # the fan-out form's ONE-launch plan (tables of one resident round of workgroups) keeps every (query, row) first score + the 32-row groups' minima in the
    # workspace: BASELINE config #1's table = 79 row blocks of 128 -> 256 x 10 112 scores + 256 x 316 minima behind the 64 counter bytes; a 10^6-row table streams
    assert lib.mrag_topk_workspace_bytes(10000, 256) >= 64 + 256 * 10112 * 4 + 256 * 316 * 4
    assert lib.mrag_topk_workspace_bytes(1000000, 256) < 256 * 1000000 * 4


def test_struct_layouts_match_the_header():
    """field order / count of the ctypes structs follows the header's typedefs"""
    from motionrag_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mrag_hip.h")).read()
    for cname, st in (("mrag_gemm_args", _lib.GemmArgs), ("mrag_attn_args", _lib.AttnArgs), ("mrag_ln_args", _lib.LnArgs),
                      ("mrag_qknorm_rope_args", _lib.QkNormRopeArgs), ("mrag_groupnorm_args", _lib.GroupNormArgs), ("mrag_conv_args", _lib.ConvArgs)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
        assert names == [f[0] for f in st._fields_], (cname, names)


def test_tuning_constants_match_the_header():
    """the developer knobs ops.py passes in `tuning` are the header's enumerators (one set of meanings on both sides of the C ABI)"""
    import re
    from motionrag_amd import ops
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mrag_hip.h")).read()
    vals = {}
    for name, expr in re.findall(r"(MRAG_(?:GEMM|ATTN)_TUNE_\w+)\s*=\s*([^,}/]+)", hdr):
        vals[name] = eval(expr.strip(), {}, {})          # "8", "1 << 16"
    for py in ("GEMM_TUNE_NO_WIDE", "GEMM_TUNE_NO_STAGED", "GEMM_TUNE_GEGLU_NO_STAGED", "GEMM_TUNE_STREAMK", "GEMM_TUNE_NO_W4", "GEMM_TUNE_NO_SKINNY", "GEMM_TUNE_TAIL_RECT",
               "ATTN_TUNE_NO_TINY", "ATTN_TUNE_LEGACY"):
        assert vals["MRAG_" + py] == getattr(ops, py), py


def test_ops_refuse_cpu_tensors():
    from motionrag_amd import ops, rag
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    with pytest.raises(ops.HipOnly):
        ops.linear(x, x)
    with pytest.raises(ops.HipOnly):
        ops.layernorm(x, None, None, 1e-5)
    with pytest.raises(ops.HipOnly):
        ops.topk(torch.zeros(4, 32), torch.zeros(1, 32), 2)
    with pytest.raises(ops.HipOnly):
        rag.RAGDatabase.from_arrays(np.zeros((4, 32), np.float32), [{"video": "a"}] * 4, device="cpu")


def test_missing_library_fails_loudly(monkeypatch):
    from motionrag_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmrag_hip.so")
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.lib()


def test_scheduler_and_rope_tables_match_oracle():
    from motionrag_amd.cogvideox import CogVideoXDDIMScheduler, get_3d_rotary_pos_embed
    from oracle import cogvideox_ref
    s = CogVideoXDDIMScheduler()
    np.testing.assert_array_equal(s.alphas_cumprod, cogvideox_ref.ddim_alphas_cumprod())
    for n in (50, 25, 3):
        np.testing.assert_array_equal(s.set_timesteps(n), cogvideox_ref.ddim_timesteps(n))
        for t in s.timesteps:
            assert s.coeffs(int(t)) == cogvideox_ref.ddim_coeffs(cogvideox_ref.ddim_alphas_cumprod(), int(t), n)
    cos, sin = get_3d_rotary_pos_embed(64, 13, 30, 45)
    rc, rs = cogvideox_ref.rope_3d(64, 13, 30, 45)
    assert cos.shape == (17550, 64) and torch.equal(cos, rc) and torch.equal(sin, rs)


def test_cama_host_tables_match_reference_golden():
    from motionrag_amd import cama
    g = np.load(os.path.join(ROOT, "tests", "golden", "sinusoid.npz"))
    pe = cama.SinusoidPositionalEmbeddings(1024, 256)
    np.testing.assert_array_equal(pe.pos_table[0][g["rows"]].numpy(), g["t256"])
    m = np.load(os.path.join(ROOT, "tests", "golden", "mask.npz"))
    at = cama.build_cama(None, None, layers=1, depth=1)
    np.testing.assert_array_equal(at.get_mask(4, 3).numpy(), m["m4x3"])
    np.testing.assert_array_equal(np.packbits(at.get_mask(10, 25).numpy()), m["m10x25"])


def test_state_dict_keys_follow_the_checkpoint_layout():
    from motionrag_amd import cama
    from motionrag_amd.cogvideox import CogVideoXTransformer3DModel
    from oracle import cama_ref, cogvideox_ref
    at = cama.build_cama(None, None)
    assert set(at.state_dict().keys()) == set(cama_ref.random_cama_sd(0, layers=4).keys())
    cfg = cogvideox_ref.DiTConfig(num_layers=1, heads=2, in_channels=16, out_channels=8, time_embed_dim=64, text_embed_dim=64,
                                  max_text_len=10, ip_dim=64, frames=3, height=8, width=12)
    m = CogVideoXTransformer3DModel(num_layers=1, num_attention_heads=2, in_channels=16, out_channels=8, time_embed_dim=64, text_embed_dim=64,
                                    max_text_seq_length=10, sample_frames=3, sample_height=8, sample_width=12).install_motion_adapters(64)
    assert set(m.state_dict().keys()) == set(cogvideox_ref.random_dit_sd(cfg).keys())
    assert list(m.attn_processors) == ["transformer_blocks.0.attn1.processor"]


def test_rag_table_io_and_filter_parsing(tmp_path):
    from motionrag_amd import rag
    annos = [{"llm_caption": None if i == 1 else f"c{i}", "id": i, "video": f"v{i}", "start_sec": 0.0, "end_sec": 1.0} for i in range(3)]
    rows = rag.prepare_annotations(annos)
    assert rows[1]["text"] == "" and rows[2]["uid"] == "coin/2" and set(rows[0]) == set(rag.SCHEMA)
    rag.add_to_db(rows, np.eye(3, 32, dtype=np.float32), text_name="t", db_path=str(tmp_path))
    rag.add_to_db(rows[:1], np.ones((1, 32), np.float32), text_name="t", db_path=str(tmp_path))
    vec = np.load(tmp_path / "t" / "vectors.npy", mmap_mode="r")
    assert vec.shape == (4, 32) and np.array_equal(vec[:3], np.eye(3, 32, dtype=np.float32)) and np.all(vec[3] == 1)
    meta = rag._read_meta(str(tmp_path / "t"))                              # Arrow IPC file, memory-mapped; the reference's row schema
    assert meta.num_rows == 4 and meta.column_names == list(rag.SCHEMA) and meta.column("video").to_pylist() == ["v0", "v1", "v2", "v0"]
    assert meta.column("text").to_pylist()[1] == "" and meta.column("id").to_pylist() == [0, 1, 2, 0]
    with pytest.raises(ValueError):
        rag.add_to_db(rows[:1], np.ones((1, 16), np.float32), text_name="t", db_path=str(tmp_path))    # dimension mismatch on append
    # a round-2 table (meta.json beside vectors.npy) still opens, and its next append converts it
    import json
    os.makedirs(tmp_path / "old")
    np.save(tmp_path / "old" / "vectors.npy", np.eye(3, 32, dtype=np.float32))
    with open(tmp_path / "old" / "meta.json", "w") as f:
        json.dump(rows, f)
    assert rag._read_meta(str(tmp_path / "old")).column("uid").to_pylist() == ["coin/0", "coin/1", "coin/2"]
    rag.add_to_db(rows[2:], np.ones((1, 32), np.float32), text_name="old", db_path=str(tmp_path))
    assert not os.path.exists(tmp_path / "old" / "meta.json") and rag._read_meta(str(tmp_path / "old")).num_rows == 4
    assert rag._WHERE_RE.match('video != "a b/c.mp4"').group(2) == "a b/c.mp4"
    assert rag._WHERE_RE.match("video != 'x'").group(2) == "x"
    assert rag._WHERE_RE.match("start_sec > 3") is None


def test_dynamicrafter_state_dict_keys_match_reference_golden():
    """the UNet mirrors load the reference's checkpoints: key sets (and shapes) equal those dumped from the reference classes"""
    import json
    from motionrag_amd import dynamicrafter as dc
    g = np.load(os.path.join(ROOT, "tests", "golden", "dc_blocks.npz"))
    meta = json.loads(str(g["meta"]))
    C, cd = 64, 96
    mods = {
        "st": dc.SpatialTransformer(C, 1, 64, depth=1, context_dim=cd, use_linear=True, image_cross_attention=True, action_cross_attention=True),
        "tt": dc.TemporalTransformer(C, 2, 64, depth=1, context_dim=cd, use_linear=True, temporal_length=4),
        "rb": dc.ResBlock(C, 128, 0.0, out_channels=96, use_temporal_conv=True), "rb2": dc.ResBlock(C, 128, 0.0, out_channels=C),
        "dn": dc.Downsample(C, True, out_channels=C), "up": dc.Upsample(C, True, out_channels=C)}
    for name, m in mods.items():
        sd = m.state_dict()
        assert sorted(sd) == meta[name]["keys"], name
        assert [list(sd[k].shape) for k in sorted(sd)] == meta[name]["shapes"], name
    u = np.load(os.path.join(ROOT, "tests", "golden", "dc_unet.npz"))
    unet = dc.UNetModel(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=(1, 2), num_res_blocks=1, channel_mult=(1, 2),
                        num_head_channels=64, transformer_depth=1, context_dim=64, use_linear=True, temporal_conv=True, temporal_attention=True,
                        temporal_self_att_only=True, use_relative_position=False, temporal_length=4, addition_attention=True,
                        image_cross_attention=True, action_cross_attention=True, default_fs=10, fs_condition=True)
    sd = unet.state_dict()
    assert sorted(sd) == [str(k) for k in u["keys"]]
    assert [list(sd[k].shape) for k in sorted(sd)] == [s[:n].tolist() for s, n in zip(u["shapes"], u["ndims"])]


def test_dc_sampler_tables_match_reference_golden():
    from motionrag_amd.dynamicrafter import DDIMSampler, make_alphas_cumprod
    g = np.load(os.path.join(ROOT, "tests", "golden", "dc_schedule.npz"))
    np.testing.assert_allclose(make_alphas_cumprod(), g["alphas_cumprod"], rtol=1e-12, atol=1e-15)
    s = DDIMSampler()
    np.testing.assert_array_equal(s.make_schedule(30, 1.0), g["t30"])
    np.testing.assert_allclose(s.ddim_sigmas, g["sigmas"], rtol=1e-6); np.testing.assert_allclose(s.ddim_alphas, g["alphas"], rtol=1e-6)
    np.testing.assert_allclose(s.ddim_alphas_prev, g["alphas_prev"], rtol=1e-6)
    np.testing.assert_array_equal(s.scale_arr, g["scale_arr"])
    np.testing.assert_array_equal(s.make_schedule(50, 0.0), g["t50"])


def test_tuple_tensor_semantics():
    """svd/pipelines/pipeline.py:25-57: both members move / repeat together, indexing / shape / dtype see the first"""
    from motionrag_amd.svd import TupleTensor
    a, b = torch.arange(6.0).view(2, 1, 3), torch.ones(2, 25, 4)
    tt = TupleTensor([a, b])
    assert isinstance(tt, tuple) and tt.shape == a.shape and tt.dtype == a.dtype and tt.size(2) == 3
    assert torch.equal(tt[1], a[1]) and torch.equal(tt[None, :].reshape(1, 2, 1, 3)[:, 0], a[None, 0])
    r = tt.repeat_interleave(3, dim=0)
    assert isinstance(r, TupleTensor) and r.shape == (6, 1, 3) and r.to_tuple()[1].shape == (6, 25, 4)
    h = tt.to(torch.float16)
    assert h.dtype == torch.float16 and h.to_tuple()[1].dtype == torch.float16
    e, ip = tt                      # how attn_processor.py:34-37 unpacks it
    assert e is a and ip is b


def test_dpm_scheduler_multipliers_match_the_restated_step():
    """CogVideoXDPMScheduler.dpm_coeffs (host float64, with the infinities of the zero-terminal-SNR first step and the alpha = 1 last step) against the oracle's
    statement-by-statement step: a scalar chain through 25 / 50 / 3 steps lands on the same values, with two noise draws exactly on the second-order steps"""
    import torch
    from motionrag_amd.cogvideox import CogVideoXDPMScheduler
    from oracle import cogvideox_ref as R
    ac = R.ddim_alphas_cumprod()
    for steps in (25, 50, 3):
        s = CogVideoXDPMScheduler()
        ts = s.set_timesteps(steps)
        g = torch.Generator().manual_seed(steps)
        x = torch.randn(4, 5, generator=g)
        xo, x0_old, mine_old = x.clone(), None, torch.zeros_like(x)
        for i, t in enumerate(ts):
            v = torch.randn(4, 5, generator=g)
            draws = []

            def nf():
                draws.append(torch.randn(4, 5, generator=g))
                return draws[-1]
            t_back = int(ts[i - 1]) if i > 0 else None
            want, want_x0 = R.dpm_step(ac, v, x0_old, int(t), t_back, xo, steps, nf)
            sa, sb, m1, m2, m3, m4, mn, second = s.dpm_coeffs(int(t), t_back)
            assert second == (0 < i < steps - 1) and len(draws) == (2 if second else 1)
            x0 = sa * x - sb * v
            got = m1 * x - m2 * (m3 * x0 - m4 * mine_old if second else x0) + mn * draws[-1]
            assert torch.isfinite(got).all() and torch.allclose(got, want, rtol=1e-4, atol=1e-5)
            x, mine_old, xo, x0_old = got, x0, want, want_x0


def test_add_to_db_recovers_from_an_interrupted_append(tmp_path):
    """add_to_db replaces vectors.npy, then meta.arrow: a crash between the two renames leaves orphan vectors.  The next append drops them first
    (the table is append-only, so its first `num_rows` vectors are the table before the interrupted append); host logic only, no GPU."""
    import numpy as np
    from motionrag_amd import rag
    rows = [{"text": "a", "id": i, "uid": f"x/{i}", "dataset": "x", "video": f"v{i}", "start_sec": 0.0, "end_sec": 1.0} for i in range(5)]
    rag.add_to_db(rows[:3], np.ones((3, 8), np.float32), text_name="t", db_path=str(tmp_path))
    vp = tmp_path / "t" / "vectors.npy"
    np.save(vp, np.concatenate([np.load(vp), 2 * np.ones((2, 8), np.float32)]))          # the state after a crash: 5 vectors, 3 rows
    rag.add_to_db(rows[3:], 3 * np.ones((2, 8), np.float32), text_name="t", db_path=str(tmp_path))
    assert np.load(vp)[:, 0].tolist() == [1.0, 1.0, 1.0, 3.0, 3.0]
    assert rag._read_meta(str(tmp_path / "t")).column("id").to_pylist() == [0, 1, 2, 3, 4]


def test_verify_unpinned_lists_the_unpinned_pieces_and_degrades_to_package_missing():
    """tools/verify_unpinned.py is the one-command check of the oracle's unpinned restatements for a machine that HAS diffusers / lancedb / sentence-transformers /
    kornia.  Here those packages are absent: `--list` names every unpinned piece with its reference call site, and a run reports
    "package missing" per check with exit code 2 -- it never crashes and never silently passes."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "verify_unpinned.py")
    lst = subprocess.run([sys.executable, tool, "--list"], capture_output=True, text=True, timeout=120)
    assert lst.returncode == 0
    for piece in ("CogVideoXTransformer3DModel", "CogVideoXDDIMScheduler", "CogVideoXDPMScheduler", "UNetSpatioTemporalConditionModel", "EulerDiscreteScheduler",
                  "AutoencoderKLCogVideoX", "AutoencoderKLTemporalDecoder", "lancedb 0.14.0", "gte-base-en-v1.5", "kornia"):
        assert piece in lst.stdout, piece
    assert "src/projects/cogvideox/module.py:23-48" in lst.stdout and "src/data/rag.py:36-61" in lst.stdout
    absent = [m for m in ("diffusers", "lancedb", "kornia") if importlib.util.find_spec(m) is None]
    if absent:                                          # this image: every check that needs an absent package says so
        run = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=300)
        assert run.returncode == 2, run.stdout + run.stderr
        assert run.stdout.count("package missing") >= 3 and "Traceback" not in run.stdout + run.stderr
