"""Host-side boundary checks that need no GPU: the C-ABI library exports every symbol include/clt_hip.h declares
(and the ctypes table binds exactly those), and the drop-in classes keep the reference's constructor arguments,
parameter names / logical shapes, config schema and error behaviour (reference tests/test_save_to_pretrained.py:64-91,
tests/test_load_from_pretrained.py:52-54, model/topk.py:56, model/standardize.py:54-55)."""
import ctypes
import os
import re

import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CONFIGS = "/root/reference/config"


def _declared():
    text = open(os.path.join(ROOT, "include", "clt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(clt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from crosslayer_transcoder_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/clt_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert _lib.lib.clt_arch() == b"gfx950" and _lib.lib.clt_abi_version() == _lib.ABI_VERSION
    # size queries are pure host code
    assert _lib.lib.clt_select_lists_workspace_bytes(4096, 12, 32768, 16, 1) > 0
    assert _lib.lib.clt_sparse_build_workspace_bytes(16, 2, 32) == (5 * 32 + 3 * 64 + 64) * 4


def _model(nl, dec="clt", L=2, d=110, F=32):
    from crosslayer_transcoder_amd.model import CrossLayerTranscoder, CrosslayerDecoder, Decoder, Encoder
    from crosslayer_transcoder_amd.model.standardize import DimensionwiseInputStandardizer, DimensionwiseOutputStandardizer
    D = CrosslayerDecoder if dec == "clt" else Decoder
    return CrossLayerTranscoder(nl, Encoder(d, F, L), D(d, F, L), DimensionwiseInputStandardizer(L, d),
                                DimensionwiseOutputStandardizer(L, d))


def test_parameter_names_shapes_and_config_schema():
    from crosslayer_transcoder_amd.model import BatchTopK, JumpReLU, PerLayerBatchTopK, PerLayerTopK, ReLU
    L, d, F = 2, 110, 32
    for dec in ("clt", "plt"):
        for nl in (ReLU(), JumpReLU(theta=0.03, bandwidth=0.01, n_layers=L, d_features=F), PerLayerTopK(k=4),
                   BatchTopK(k=4, e=0.1), PerLayerBatchTopK(k=4, e=0.1, n_layers=L)):
            m = _model(nl, dec, L, d, F)
            shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            assert shapes["encoder.W"] == (L, d, F) and shapes["encoder.b"] == (L, F) and shapes["decoder.b"] == (L, d)
            if dec == "clt":
                assert all(shapes[f"decoder.W_{l}"] == (l + 1, F, d) for l in range(L))
            else:
                assert shapes["decoder.W"] == (L, F, d)
            assert shapes["input_standardizer.mean"] == (L, d) and shapes["output_standardizer.std"] == (L, d)
            if isinstance(nl, JumpReLU):
                assert shapes["nonlinearity.theta"] == (1, L, F)
            if isinstance(nl, (BatchTopK, PerLayerBatchTopK)):
                assert shapes["nonlinearity.threshold"] == ((L,) if isinstance(nl, PerLayerBatchTopK) else (1,))
            cfg = m.to_config()
            assert cfg["class_path"] == "crosslayer_transcoder.model.clt.CrossLayerTranscoder"
            assert cfg["init_args"]["encoder"] == {"class_path": "crosslayer_transcoder.model.clt.Encoder",
                                                   "init_args": {"d_acts": d, "d_features": F, "n_layers": L}}
            again = type(m).from_config(yaml.safe_load(yaml.dump(cfg)))
            assert {k: tuple(v.shape) for k, v in again.state_dict().items()} == shapes
    # the encoder weight is stored feature-major but keeps the reference's logical layout
    enc = _model(ReLU()).encoder
    assert enc.W.permute(0, 2, 1).is_contiguous()


def test_error_conventions_without_gpu():
    from crosslayer_transcoder_amd.model import PerLayerTopK
    from crosslayer_transcoder_amd.model.clt_lightning import CrossLayerTranscoderModule
    from crosslayer_transcoder_amd.model.standardize import DimensionwiseInputStandardizer
    with pytest.raises(AssertionError):                      # k > d_features (model/topk.py:56)
        PerLayerTopK(k=6)(torch.zeros(3, 4, 5))
    with pytest.raises(ValueError):                          # "Standardizer not initialized" (standardize.py:54-55)
        DimensionwiseInputStandardizer(2, 4)(torch.zeros(1, 2, 4))
    from crosslayer_transcoder_amd.model import ReLU
    mod = CrossLayerTranscoderModule(model=_model(ReLU()), optimizer="sgd")
    with pytest.raises(ValueError):                          # unknown optimizer (clt_lightning.py:353-354)
        mod.configure_optimizers()
    with pytest.raises(RuntimeError):                        # no CPU fallback in the product path
        mod.compute_step(torch.zeros(4, 2, 2, 110), 0)


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="reference configs not present on this machine")
@pytest.mark.parametrize("name", ["debug-cpu.yaml", "default.yaml", "jumprelu-clt.yaml", "topk-clt.yaml"])
def test_reference_yaml_configs_instantiate(name):
    """The reference's own LightningCLI YAML files build the drop-in module tree unchanged (class_path prefixes are
    mapped; eval-time metric objects are dropped).  Large configs are only checked structurally (meta device)."""
    from crosslayer_transcoder_amd import trainer
    path = os.path.join(REF_CONFIGS, name)
    if not os.path.exists(path):
        pytest.skip("config not in this reference version")
    cfg = trainer.load_config(path)
    small = name == "debug-cpu.yaml"
    with torch.device("cpu" if small else "meta"):
        mod = trainer.build_module(cfg)
    want = cfg["model"]["class_path"].rsplit(".", 1)[1]
    assert type(mod).__name__ == want
    enc_args = cfg["model"]["init_args"]["model"]["init_args"]["encoder"]["init_args"]
    assert tuple(mod.model.encoder.W.shape) == (enc_args["n_layers"], enc_args["d_acts"], enc_args["d_features"])
    conf = mod.configure_optimizers() if small else None
    if small:
        opt = conf["optimizer"] if isinstance(conf, dict) else conf
        # (PyYAML reads the reference's `1e-3` spelling as a string; the loader turns it into the float jsonargparse would)
        assert opt.param_groups[0]["lr"] == float(cfg["model"]["init_args"]["learning_rate"]) == 1e-3
        assert isinstance(mod.pre_actv_loss, float) and mod.pre_actv_loss == 1e-6


def test_own_debug_config_builds_module_and_datamodule():
    """config/debug-cpu.yaml of this repository (BASELINE.json configs[0]): model tree, optimizer / scheduler and the
    TensorDataModule of its data: section."""
    from crosslayer_transcoder_amd import trainer
    cfg = trainer.load_config(os.path.join(ROOT, "config", "debug-cpu.yaml"))
    mod = trainer.build_module(cfg)
    assert type(mod).__name__ == "JumpReLUCrossLayerTranscoderModule" and mod.fused_step and not mod.automatic_optimization
    conf = mod.configure_optimizers()
    assert conf["optimizer"].param_groups[0]["lr"] == 1e-3 and conf["lr_scheduler"]["name"] == "step_decay"
    dm = trainer.build_datamodule(cfg)
    assert dm.batch_size == 16 and dm.shuffle and dm.tensor_file.endswith("debug_actvs.pt")
    # trainer.precision of the reference's production YAMLs: fp16 operands + dynamic loss scaling (GradScaler's rule)
    trainer.apply_trainer_precision({"trainer": {"precision": "16-mixed"}})
    import torch
    from crosslayer_transcoder_amd import get_precision, set_precision
    from crosslayer_transcoder_amd.precision import LossScaler, mixed16, operand_dtype
    assert get_precision() == "fp16" and mixed16() and operand_dtype() == torch.float16
    set_precision("bf16-mixed")
    assert operand_dtype() == torch.bfloat16 and mixed16()
    set_precision("32")
    assert not mixed16()
    sc = LossScaler(growth_interval=3)
    assert sc.scale == 65536.0
    sc.update(True)
    assert sc.scale == 32768.0 and sc.skipped_steps == 1
    for _ in range(3):
        sc.update(False)
    assert sc.scale == 65536.0 and sc.growth_tracker == 0


def test_forward_ops_are_registered_with_the_dispatcher():
    """custom_ops.py: the module forward's operators exist as torch.ops.clt_hip.* with schemas, fake-tensor rules (shape
    propagation without a device) and no CPU kernel (there is no CPU fallback)."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from crosslayer_transcoder_amd import custom_ops as C
    for name in C.REGISTERED:
        op = getattr(torch.ops.clt_hip, name).default
        assert op._schema.name == f"clt_hip::{name}"
    with FakeTensorMode():
        xs, W, b = torch.empty(8, 3, 16, device="cuda"), torch.empty(3, 16, 32, device="cuda"), torch.empty(3, 32, device="cuda")
        pre = torch.ops.clt_hip.encoder(xs, W, b)
        assert pre.shape == (8, 3, 32) and pre.dtype == torch.float32
        feat, thr = torch.ops.clt_hip.batch_topk(pre, 4, True, torch.empty(3, device="cuda"), 0.1)
        r = torch.ops.clt_hip.clt_decoder(feat, torch.empty(3, 16, device="cuda"), [torch.empty(l + 1, 32, 16, device="cuda") for l in range(3)])
        assert r.shape == (8, 3, 16) and thr.shape == (3,)
    import pytest
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.clt_hip.relu(torch.zeros(4, 4))


def test_fp16_fused_step_advances_the_lightning_progress_hooks(monkeypatch):
    """`precision: "16-mixed"` with fused_step: the library's own loss scaling steps the RAW FusedAdam (an outer GradScaler
    would unscale a second time), so Lightning's manual-optimisation progress — trainer.global_step is the count of
    LightningOptimizer._on_after_step calls — is advanced by calling the wrapper's _on_before_step / _on_after_step hooks
    around the raw step.  Lightning is not installed here ("parity unpinned" for the trainer integration): the wrapper is a
    stub with LightningOptimizer's attribute names; the engine call is stubbed out (no GPU in this test)."""
    import crosslayer_transcoder_amd as cta
    from crosslayer_transcoder_amd.model import BatchTopK
    from crosslayer_transcoder_amd.model.clt_lightning import CrossLayerTranscoderModule
    from crosslayer_transcoder_amd.precision import LossScaler
    mod = CrossLayerTranscoderModule(model=_model(BatchTopK(k=2, e=0.1)))
    raw = mod.configure_optimizers()
    calls = []

    class Wrapper:                       # lightning.pytorch.core.optimizer.LightningOptimizer's surface
        optimizer = raw

        def _on_before_step(self):
            calls.append("before")

        def _on_after_step(self):
            calls.append("after")

        def step(self, *a, **k):
            raise AssertionError("the wrapper's step() would route through the precision plugin's GradScaler")

    monkeypatch.setattr(mod, "compute_step", lambda *a, **k: {"grads": {}, "loss": torch.tensor(1.0)})
    monkeypatch.setattr(raw, "step", lambda *a, **k: calls.append("raw-step"))
    mod.loss_scaler = LossScaler()
    try:
        cta.set_precision("16-mixed")
        batch = torch.zeros(4, 2, 2, 110)
        for i in range(3):
            mod.fused_train_step(batch, i, raw, None, stepper=Wrapper())
    finally:
        cta.set_precision("32")
    assert calls == ["before", "raw-step", "after"] * 3
    assert mod.loss_scaler.growth_tracker == 3 and mod.loss_scaler.skipped_steps == 0


def test_tail_split_cost_model():
    """ops._tail_split (host logic of the per-layer decoder forward): which of a persistent batched GEMM's problems leave the
    launch for a grouped one over pieces of the contraction.  Pieces are whole pairs of k-tiles, keep 2048 of the contraction,
    and a split is taken only where the model prices it 10 % (or the asked gain) below the one launch."""
    from crosslayer_transcoder_amd import ops
    assert ops._tail_split(12, 48, 32768, 256) == (2, 8)            # B 4096, d 768: 576 tiles = 2.25 rounds -> 2 + 3/8
    assert ops._tail_split(12, 24, 32768, 256) == (2, 16)           # B 2048: 1.125 rounds -> 1 + 3/16
    assert ops._tail_split(12, 96, 32768, 256) == (0, 1)            # B 8192: 4.5 rounds of 5, priced at 7 %
    assert ops._tail_split(12, 96, 32768, 256, 0.05) == (4, 2)      # ... taken when 5 % is enough
    assert ops._tail_split(12, 128, 32768, 256) == (0, 1)           # whole rounds: nothing to gain
    assert ops._tail_split(12, 12, 32768, 256) == (0, 1)            # B 1024: the pieces' fixed costs eat the gain
    assert ops._tail_split(12, 48, 768, 256) == (0, 1)              # short contraction: no pieces
    assert ops._tail_split(12, 48, 10000, 256) == (0, 1)            # no piece count divides the contraction into pairs of k-tiles
    for items, tiles, K, blocks in ((12, 48, 32768, 256), (7, 30, 16384, 304), (3, 2, 8192, 256), (16, 5, 65536, 64)):
        t, s = ops._tail_split(items, tiles, K, blocks)
        assert 0 <= t <= items and (t == 0) == (s == 1)
        if t:
            assert K % (128 * s) == 0 and K // s >= 2048
