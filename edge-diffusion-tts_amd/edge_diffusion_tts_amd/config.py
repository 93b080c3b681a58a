"""CFG / TrainPhase / get_device / set_seed -- host-side mirror of the reference configuration surface.

Mirrors /root/reference/edge_diffusion_tts/config.py (CFG :51-153, __post_init__ :155-170,
from_dict/to_dict :197-213, get_device :18-32, set_seed :35-41).  Every field name and default of the
reference dataclass is kept so that ``CFG(**ckpt["cfg"])`` / ``CFG.from_dict`` round-trip; only the fields
listed in HOT_PATH_FIELDS are read by the MI355X sampler path.
"""
from __future__ import annotations

import dataclasses as _dc
import enum
import os
import random
import time
from typing import Any, Dict

import numpy as np
import torch

# Fields the DDIM sampler / decoder actually consume ("Config / flags").
HOT_PATH_FIELDS = (
    "n_mels", "diff_steps", "device", "hidden", "codebook_size", "semantic_dim", "heads", "ffn_mult",
    "dropout", "use_adaln", "attn_window_size", "layers",
)


def get_device() -> str:
    """'cuda' on PyTorch-ROCm with a visible GPU (that is the MI355X path), else mps / xla / cpu."""
    if torch.cuda.is_available():
        return "cuda"
    mps = getattr(torch.backends, "mps", None)
    if mps is not None and mps.is_available():
        return "mps"
    try:  # TPU probe, same precedence as the reference
        import torch_xla.core.xla_model  # noqa: F401
        return "xla"
    except ImportError:
        return "cpu"


def set_seed(seed: int) -> None:
    """Seed python, numpy and torch (all devices)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class TrainPhase(enum.Enum):
    DIFFUSION = "diffusion"
    PROGRESSIVE = "progressive"
    CONSISTENCY = "consistency"


def _default_fsq_levels():
    return [4, 4, 3, 3, 2, 2, 2, 2]


def _default_run_name():
    return time.strftime("run_%Y%m%d_%H%M%S")


@_dc.dataclass
class CFG:
    # system
    seed: int = 42
    device: str = _dc.field(default_factory=get_device)
    out_dir: str = "run_edge_diffusion"
    run_name: str = _dc.field(default_factory=_default_run_name)
    # data
    data_root: str = "./data"
    ljspeech_dir: str = "./data/LJSpeech-1.1"
    sample_rate: int = 16000
    orig_sr: int = 22050
    segment_secs: float = 2.0
    segment_len: int = 32000
    num_workers: int = 0
    pin_memory: bool = False
    # mel front-end
    n_fft: int = 1024
    hop_length: int = 160
    win_length: int = 1024
    n_mels: int = 80
    f_min: float = 0.0
    f_max: float = 8000.0
    # semantic tokens
    hubert_id: str = "facebook/hubert-base-ls960"
    hubert_layer: int = 9
    semantic_dim: int = 128
    codebook_size: int = 512
    vq_commit: float = 1.0
    use_fsq: bool = True
    fsq_levels: list = _dc.field(default_factory=_default_fsq_levels)
    # decoder architecture
    hidden: int = 160
    layers: int = 4
    heads: int = 4
    ffn_mult: int = 2
    use_depthwise: bool = True
    use_flash_attn: bool = True
    use_adaln: bool = True
    dropout: float = 0.2
    attn_window_size: int = 64
    # diffusion
    diff_steps: int = 1000
    beta_start: float = 1e-4
    beta_end: float = 2e-2
    use_v_prediction: bool = True
    max_timestep: int = 950
    # training phases
    phase: TrainPhase = TrainPhase.DIFFUSION
    diffusion_epochs: int = 50
    progressive_epochs_per_halving: int = 5
    progressive_target_steps: int = 4
    consistency_epochs: int = 10
    consistency_weight: float = 1.0
    # optimisation
    batch_size: int = 4
    grad_accumulation: int = 8
    lr: float = 2e-4
    lr_consistency: float = 1e-4
    weight_decay: float = 0.01
    grad_clip: float = 1.0
    # logging
    log_every_steps: int = 50
    val_every_steps: int = 200
    plot_every_steps: int = 100
    val_batches: int = 4
    # inference / checkpoint
    inference_steps: int = 4
    ckpt_path: str = ""

    def __post_init__(self) -> None:
        # segment length snapped down to a multiple of 320 samples (config.py:157-162)
        self.segment_len = (int(self.sample_rate * self.segment_secs) // 320) * 320
        # the reference creates both directories as a construction side effect (config.py:165-166)
        for d in (self.data_root, self.out_dir):
            os.makedirs(d, exist_ok=True)
        if not self.ckpt_path:
            self.ckpt_path = os.path.join(self.out_dir, "checkpoint_latest.pt")

    # -- helpers kept for API compatibility --------------------------------------------------
    def setup_environment(self) -> None:
        set_seed(self.seed)
        if hasattr(torch, "set_float32_matmul_precision"):
            torch.set_float32_matmul_precision("high")

    def print_config(self) -> None:
        bar = "=" * 60
        print(f"{bar}\n   EDGE DIFFUSION TTS -- MI355X sampler path\n{bar}")
        print(f"Device: {self.device}")
        print(f"Segment: {self.segment_len} samples ({self.segment_len / self.sample_rate:.2f}s)")
        print(f"Decoder: hidden={self.hidden} layers={self.layers} heads={self.heads}")
        print(f"Target inference steps: {self.inference_steps}\n{bar}\n")

    def get_run_dir(self) -> str:
        return os.path.join(self.out_dir, self.run_name)

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "CFG":
        known = {f.name for f in _dc.fields(cls)}
        kw = {k: v for k, v in d.items() if k in known}
        if isinstance(kw.get("phase"), str):
            kw["phase"] = TrainPhase(kw["phase"])
        return cls(**kw)

    def to_dict(self) -> Dict[str, Any]:
        out = {}
        for f in _dc.fields(self):
            v = getattr(self, f.name)
            out[f.name] = v.value if isinstance(v, TrainPhase) else v
        return out
