"""CPU-only checks of the drop-in boundary: the gfx950 shared library loads, exports every symbol that
include/visualbert_hip.h declares (no compute calls without a GPU), and the Python surface keeps the
reference's names and state-dict contract."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "visualbert_amd", "libvisualbert_hip.so")
HDR = os.path.join(ROOT, "include", "visualbert_hip.h")
DEV_HDR = os.path.join(ROOT, "include", "visualbert_hip_dev.h")
DEV_SO = os.path.join(ROOT, "visualbert_amd", "libvisualbert_hip_dev.so")


def header_symbols(path=HDR):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vb_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_table_agree():
    from visualbert_amd import _lib
    assert header_symbols() == sorted(_lib.SIGNATURES.keys())


def test_library_loads_and_exports_every_declared_symbol():
    if not os.path.isfile(SO):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(SO)
    for name in header_symbols():
        assert hasattr(lib, name), name
    lib.vb_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.vb_version()


def test_layer_backward_insists_on_the_forward_output():
    """argument checks only (they run on the host, before any launch): with H <= 768 the LayerNorm forward may have skipped its pre-LN
    sum, so the backward cannot run without h_out; the saved-workspace size answers without a device too"""
    from visualbert_amd import _lib
    L = _lib.lib()
    B, S, H, I, NH = 2, 16, 768, 3072, 12
    assert L.vb_bert_layer_saved_bytes(_lib.VB_BF16, B, S, H, I, NH, 0.1) > 0
    assert L.vb_bert_layer_saved_bytes(_lib.VB_BF16, B, S, H + 4, I, NH, 0.1) == -1          # H is not heads x 64
    buf = (ctypes.c_char * 4096)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    arr12, arr4 = (ctypes.c_void_p * 12)(*([p.value] * 12)), (ctypes.c_void_p * 4)(*([p.value] * 4))
    ld = (ctypes.c_int64 * 4)(H, H, H, I)
    rc = L.vb_bert_layer_bwd(_lib.VB_BF16, p, None, p, p, p, p, p, arr12, arr12, arr4, ld, B, S, H, I, NH, 0.1, 0.1, 1, 1, None)
    assert rc == -1, rc                                                                      # VB_ERR_ARG: no h_out


def test_developer_knobs_are_not_in_the_product_library():
    """ablation bits, timelines and measurement kernels (include/visualbert_hip_dev.h) exist only in the developer build."""
    from visualbert_amd import _lib
    dev_syms = [s for s in header_symbols(DEV_HDR) if s not in header_symbols()]
    assert sorted(dev_syms) == sorted(_lib.DEV_SIGNATURES.keys())
    if not (os.path.isfile(SO) and os.path.isfile(DEV_SO)):
        import __graft_entry__ as g
        g.build()
    lib, dev = ctypes.CDLL(SO), ctypes.CDLL(DEV_SO)
    for name in dev_syms:
        assert not hasattr(lib, name), name
        assert hasattr(dev, name), name
    for name in header_symbols():
        assert hasattr(dev, name), name
    # the vendor-library yardstick (csrc/vendor_gemm.hip) and the experiment arms of vb_stream_opts.nt_kernel live in the developer
    # library only: the product contains no hipBLASLt binding at all and refuses those kernel ids when they are SET
    blob = open(SO, "rb").read()
    assert b"hipblasLt" not in blob and b"hipblaslt" not in blob
    assert b"hipblasLtMatmul" in open(DEV_SO, "rb").read()
    for L in (lib, dev):
        L.vb_stream_set_opts.restype = ctypes.c_int
        L.vb_stream_set_opts.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    fake_stream = ctypes.c_void_p(0x1234)                       # options are a host-side table keyed by the handle: no GPU needed
    for k in (0, 1, 14, 22, 24, 42, 81, 90) + tuple(_lib.DEV_NT_KERNELS):
        o = _lib.StreamOpts(0, k, 0, 0)
        want_product = 0 if k not in _lib.DEV_NT_KERNELS else -1                 # VB_ERR_ARG
        assert lib.vb_stream_set_opts(fake_stream, ctypes.byref(o)) == want_product, k
        assert dev.vb_stream_set_opts(fake_stream, ctypes.byref(o)) == 0, k
    lib.vb_stream_set_opts(fake_stream, None)
    dev.vb_stream_set_opts(fake_stream, None)


def test_missing_library_fails_loudly(tmp_path):
    from visualbert_amd import _lib
    old = (_lib._lib, _lib._lib_path, _lib._device_type)
    try:
        _lib.set_library(str(tmp_path / "nope.so"))
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            _lib.lib()
    finally:
        _lib._lib, _lib._lib_path, _lib._device_type = old


def test_cpu_tensor_is_rejected_by_the_product_path():
    from visualbert_amd import _lib
    if _lib.device_type() != "cuda":
        pytest.skip("simulator session")
    with pytest.raises(RuntimeError, match="library runs on cuda"):
        _lib.ptr(torch.zeros(4))


def test_state_dict_contract_matches_reference_names():
    """key names and shapes are the checkpoint compatibility contract (models/model_wrapper.py:201-221)."""
    from oracle import visualbert_oracle as vo
    from visualbert_amd.modeling import BertConfig, TrainVisualBERTObjective
    cfg = vo.OracleConfig(**vo.CONFIGS["micro"])
    for head in ("pretraining", "vqa", "nlvr"):
        bc = BertConfig(cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size)
        m = TrainVisualBERTObjective(bc, head, visual_embedding_dim=cfg.visual_embedding_dim)
        sd = m.state_dict()
        want = vo.param_shapes(cfg, head)
        for k, shp in want.items():
            assert k in sd and tuple(sd[k].shape) == tuple(shp), k
        extra = set(sd.keys()) - set(want.keys())
        assert extra <= {"cls.predictions.decoder.weight"}, extra
        if head == "pretraining":
            assert m.cls.predictions.decoder.weight is m.bert.embeddings.word_embeddings.weight
        # q/k/v storage is packed: one [3H, H] matrix in the arena
        sa = m.bert.encoder.layer[0].attention.self
        assert sa._adjacent()
        n_params = sum(p.numel() for p in m.parameters())
        assert n_params == sum(int(torch.tensor(s).prod()) for s in want.values())


def test_bert_base_parameter_count():
    from visualbert_amd.modeling import BertConfig, TrainVisualBERTObjective
    m = TrainVisualBERTObjective(BertConfig(30522), "pretraining", visual_embedding_dim=2048)
    assert sum(p.numel() for p in m.parameters()) == 112074812          # SURVEY.md section 8b [measured]
    names = [n for n, _ in m.named_parameters() if "pooler" not in n]
    assert sum(dict(m.named_parameters())[n].numel() for n in names) == 111484220


def test_schedules_match_oracle():
    from oracle import visualbert_oracle as vo
    from visualbert_amd.optimization import WarmupLinearSchedule
    s = WarmupLinearSchedule(warmup=0.1, t_total=100)
    for step in (0, 1, 5, 10, 11, 50, 99, 100, 150):
        assert abs(s.get_lr(step) - vo.schedule_lr(step, 100, 0.1)) < 1e-12


def test_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under visualbert_amd/ (Python or kernel sources) may import, load or name it."""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for path in glob.glob(os.path.join(root, "visualbert_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".h", "Makefile")):
            if "oracle" in open(path, errors="ignore").read():
                offenders.append(os.path.relpath(path, root))
    assert not offenders, offenders
