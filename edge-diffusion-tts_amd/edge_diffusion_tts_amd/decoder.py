"""EdgeDiffusionDecoder -- API mirror of /root/reference/edge_diffusion_tts/models/decoder.py:14-109 on MI355X.

The module owns parameters and buffers under the reference's state-dict key names, so
``load_state_dict(reference_decoder.state_dict())`` works, but it has no sub-module forward code: ``forward`` hands
device pointers to the C ABI (include/edtts.h, edtts_decoder_forward) where the whole network runs as hand-written
gfx950 kernels.  Inference only (the reference path this replaces runs under ``torch.no_grad``, inference.py:23);
dropout is the identity as in ``decoder.eval()``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import native
from .synth import decoder_shapes, sinusoidal_table, time_frequencies


class _Node(nn.Module):
    """Parameter container; the tree of _Node objects reproduces the reference's dotted key names."""


def _attach(root: nn.Module, key: str, tensor: torch.Tensor, is_buffer: bool) -> None:
    *path, leaf = key.split(".")
    mod = root
    for name in path:
        if name not in mod._modules:
            mod.add_module(name, _Node())
        mod = mod._modules[name]
    if is_buffer:
        mod.register_buffer(leaf, tensor)
    else:
        mod.register_parameter(leaf, nn.Parameter(tensor, requires_grad=False))


class EdgeDiffusionDecoder(nn.Module):
    def __init__(self, cfg, max_len: int = 1000, max_context_len: int = 512, compute_dtype: str = "f32"):
        """``max_len`` / ``max_context_len`` size the two sinusoidal tables (reference: 1000 / 512, decoder.py:38,41);
        they are pure functions of position, so larger values only lift the reference's length limit.
        ``compute_dtype``: "f32" (the reference's arithmetic) or "bf16" -- contractions on bf16 MFMA with fp32 accumulation,
        residual stream / norms / softmax in fp32 (the reference's AMP precedent, utils/speed_utils.py:70; compiled for
        head_dim 32, i.e. BASELINE config 3: hidden=256, heads=8).  Parameters stay fp32 either way."""
        super().__init__()
        self.cfg = cfg
        if compute_dtype not in native.COMPUTE_DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(native.COMPUTE_DTYPES)}, got {compute_dtype!r}")
        self.compute_dtype = compute_dtype
        self.max_len, self.max_context_len, self.n_step_emb = int(max_len), int(max_context_len), 16
        H = cfg.hidden
        for key, shape in decoder_shapes(cfg, self.max_len, self.max_context_len, self.n_step_emb).items():
            if key == "pos_emb.pe":
                _attach(self, key, sinusoidal_table(self.max_len, H), True)
            elif key == "context_pos_emb.pe":
                _attach(self, key, sinusoidal_table(self.max_context_len, H), True)
            else:
                _attach(self, key, self._default_init(key, shape), False)
        self._fix_bias_init()
        self.register_buffer("_time_freqs", time_frequencies(H), persistent=False)
        self._zero_mod = None
        self._packed: Optional[torch.Tensor] = None
        self._packed_sig: Optional[Tuple] = None
        self._workspaces: Dict[Tuple, torch.Tensor] = {}
        self._pinned_workspaces = set()  # keys handed out during graph capture (never evicted)

    # same distributions as the reference's default construction (nn.Linear / nn.Embedding defaults, ones for norm
    # gains, zeros for final out_proj and the AdaLN projections -- decoder.py:63-64, transformer.py:61-62)
    @staticmethod
    def _default_init(key: str, shape) -> torch.Tensor:
        leaf = key.rsplit(".", 1)[-1]
        if key.startswith("out_proj.") or ".norm1.proj." in key or ".norm3.proj." in key:
            return torch.zeros(shape)
        if key.endswith("emb.weight"):
            return torch.randn(shape)
        if "norm" in key and leaf == "weight" and len(shape) == 1:
            return torch.ones(shape)
        if key == "final_norm.bias":
            return torch.zeros(shape)
        if leaf == "weight":
            bound = 1.0 / math.sqrt(shape[-1])
            return torch.empty(shape).uniform_(-bound, bound)
        return torch.zeros(shape)  # Linear biases: drawn in _fix_bias_init (needs the sibling weight's fan-in)

    def _fix_bias_init(self) -> None:
        sd = dict(self.named_parameters())
        for k, p in sd.items():
            if k.endswith(".bias") and not (k.startswith("out_proj.") or ".norm1.proj." in k or ".norm3.proj." in k or k == "final_norm.bias"):
                w = sd[k[:-4] + "weight"]
                bound = 1.0 / math.sqrt(w.shape[-1])
                with torch.no_grad():
                    p.uniform_(-bound, bound)

    def reset_parameters(self) -> None:
        self._fix_bias_init()

    # ------------------------------------------------------------------------------------------ checkpoint interop
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts the reference's decoder state-dicts as they are saved in practice: plain keys
        (train.py:195,291-297; generate_sample.py:51) and keys prefixed with ``_orig_mod.`` when the decoder had been
        wrapped by torch.compile before saving (train.py:84-86 + :195; only train_v2.py:339 unwraps it).  Tensors are
        converted to fp32; positional tables shorter/longer than this module's are rejected by the normal shape check."""
        sd = {}
        for k, v in state_dict.items():
            k = k[len("_orig_mod."):] if k.startswith("_orig_mod.") else k
            if k.endswith("rope.cos_cached") or k.endswith("rope.sin_cached"):  # non-persistent in the reference, tolerated
                continue
            sd[k] = v.to(torch.float32) if torch.is_tensor(v) and v.is_floating_point() else v
        return super().load_state_dict(sd, strict=strict, **kw)

    @classmethod
    def from_checkpoint(cls, checkpoint, cfg=None, device=None, **kw) -> "EdgeDiffusionDecoder":
        """Build a decoder from a reference checkpoint dict (or a path to one): ``{"decoder": state_dict, "cfg": dict, ...}``
        as written by train.py:291-297 / train_v2.py:335-341 and read by generate_sample.py:38-51.  ``cfg`` overrides the
        stored config; ``codebook_size`` follows the checkpoint's token embedding (FSQ runs have 2304 codes, train_v2.py:246)."""
        from .config import CFG
        if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
            checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
        sd = checkpoint["decoder"] if "decoder" in checkpoint else checkpoint
        if cfg is None:
            stored = checkpoint.get("cfg") if isinstance(checkpoint, dict) else None
            if stored is None:
                cfg = CFG()
            elif isinstance(stored, dict):
                cfg = CFG.from_dict(dict(stored))
            else:  # a pickled reference CFG object: copy the fields both classes share
                cfg = CFG.from_dict({k: getattr(stored, k) for k in CFG.__dataclass_fields__ if hasattr(stored, k) and k != "phase"})
        tok = next((v for k, v in sd.items() if k.endswith("token_emb.weight")), None)
        if tok is not None and tok.shape[0] != cfg.codebook_size:
            cfg.codebook_size = int(tok.shape[0])
        pe = next((v for k, v in sd.items() if k.endswith("pos_emb.pe") and "context" not in k), None)
        cpe = next((v for k, v in sd.items() if k.endswith("context_pos_emb.pe")), None)
        dec = cls(cfg, max_len=int(pe.shape[0]) if pe is not None else 1000,
                  max_context_len=int(cpe.shape[0]) if cpe is not None else 512, **kw)
        dec.load_state_dict(sd)
        if device is not None:
            dec = dec.to(device)
        return dec.eval()

    # ------------------------------------------------------------------------------------------ native state
    def dims(self) -> native.EdttsDims:
        c = self.cfg
        window = -1 if c.attn_window_size is None else int(c.attn_window_size)
        return native.EdttsDims(c.hidden, c.layers, c.heads, c.n_mels, c.ffn_mult, c.codebook_size, c.semantic_dim, window,
                                self.max_len, self.max_context_len, self.n_step_emb, native.COMPUTE_DTYPES[self.compute_dtype])

    def _state_tensors(self) -> Dict[str, torch.Tensor]:
        sd = {k: v for k, v in self.named_parameters()}
        sd.update({k: v for k, v in self.named_buffers()})
        sd["time_freqs"] = sd.pop("_time_freqs")
        if not self.cfg.use_adaln:
            # Plain RMSNorm blocks (layers/transformer.py:101-104,119-122,142-157): the kernels' AdaLN slots get the RMSNorm
            # gain and an all-zero modulation projection, i.e. (1 + scale, shift) = (1, 0) -- y * 1 + 0 is exact in fp32.
            H = self.cfg.hidden
            dev = sd["in_proj.weight"].device
            if self._zero_mod is None or self._zero_mod[0].device != dev:
                self._zero_mod = (torch.zeros(2 * H, H, device=dev), torch.zeros(2 * H, device=dev))
            for l in range(self.cfg.layers):
                for n in ("norm1", "norm3"):
                    sd[f"layers.{l}.{n}.norm.weight"] = sd.pop(f"layers.{l}.{n}.weight")
                    sd[f"layers.{l}.{n}.proj.weight"], sd[f"layers.{l}.{n}.proj.bias"] = self._zero_mod
        return sd

    def _slot_tensors(self):
        """(slot names, tensors in slot order).  Building the name -> tensor map through named_parameters() walks the module tree and
        formats ~90 dotted names (245 us per call on the build container's host, a third of a B = 1 sampler call).  The tree of
        containers is fixed after construction, so each slot is resolved ONCE to (the owning module's _parameters / _buffers dict,
        key) and read from there on every call: a parameter that was written into, moved by .to() or replaced by a new Parameter
        object is picked up all the same (12 us)."""
        if not self.cfg.use_adaln:  # (plain-RMSNorm decoders substitute tensors for the AdaLN slots: the general path)
            sd = self._state_tensors()
            names = native.slot_names(self.cfg.layers)
            return names, [sd[n] for n in names]
        refs = getattr(self, "_slot_refs", None)
        if refs is None:
            names = native.slot_names(self.cfg.layers)
            pairs = []
            for n in names:
                key = "_time_freqs" if n == "time_freqs" else n
                *path, leaf = key.split(".")
                mod = self
                for part in path:
                    mod = mod._modules[part]
                pairs.append((mod._parameters if leaf in mod._parameters else mod._buffers, leaf))
            refs = self._slot_refs = (names, pairs)
        return refs[0], [d[k] for d, k in refs[1]]

    def _ensure_packed(self) -> torch.Tensor:
        names, tensors = self._slot_tensors()
        sig = tuple((t.data_ptr(), t._version) for t in tensors)
        if self._packed is None or sig != self._packed_sig:
            dev = tensors[0].device
            for n, t in zip(names, tensors):
                if t.device != dev or t.dtype != torch.float32:
                    raise native.EdttsError(f"weight {n}: expected fp32 on {dev}, got {t.dtype} on {t.device}")
            dims = self.dims()
            nbytes = native.packed_bytes(dims)
            if self._packed is None or self._packed.numel() != nbytes or self._packed.device != dev:
                self._packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            native.pack_weights(dims, [t.contiguous() for t in tensors], self._packed)
            self._packed_sig = sig
        return self._packed

    WORKSPACE_CACHE = 8

    def workspace(self, B: int, T: int, S: int, cond_rows: int, device, tag: str = "") -> torch.Tensor:
        """Cached scratch memory for one (shape, device, tag).  At most WORKSPACE_CACHE entries are kept; the least recently USED one
        is dropped to make room -- never one that a captured hipGraph points at (a workspace handed out while the stream was
        capturing is pinned for the life of the decoder: a replay writes into it)."""
        key = (B, T, S, cond_rows, str(device), tag)
        ws = self._workspaces.pop(key, None)
        if ws is None:
            evictable = [k for k in self._workspaces if k not in self._pinned_workspaces]  # insertion order = least recently used first
            while len(self._workspaces) >= self.WORKSPACE_CACHE and evictable:
                del self._workspaces[evictable.pop(0)]
            nbytes = native.workspace_bytes(self.dims(), B, T, S, cond_rows)
            ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)  # must start zero-filled (padding lanes)
        self._workspaces[key] = ws  # (re-)inserted last = most recently used
        if ws.is_cuda and torch.cuda.is_current_stream_capturing():
            self._pinned_workspaces.add(key)
            if len(self._pinned_workspaces) > self.WORKSPACE_CACHE:
                import warnings
                warnings.warn(f"EdgeDiffusionDecoder: {len(self._pinned_workspaces)} workspaces are pinned by captured graphs (more than "
                              f"WORKSPACE_CACHE = {self.WORKSPACE_CACHE}); call release_pinned() for shapes whose graphs are gone",
                              RuntimeWarning, stacklevel=3)
        return ws

    def release_pinned(self, B: Optional[int] = None, T: Optional[int] = None, S: Optional[int] = None) -> int:
        """Un-pin (and drop) the workspaces that were handed out during graph capture -- all of them, or those of one (B, T, S).
        A pin is keyed by shape, not by graph, and the decoder cannot see a hipGraph die: call this once the graphs that replay
        into those workspaces have been destroyed (replaying one afterwards would write into freed memory).  Returns the number
        of workspaces released."""
        keys = [k for k in self._pinned_workspaces if (B is None or k[0] == B) and (T is None or k[1] == T) and (S is None or k[2] == S)]
        for k in keys:
            self._pinned_workspaces.discard(k)
            self._workspaces.pop(k, None)
        return len(keys)

    # ------------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x_t: torch.Tensor, t: torch.Tensor, sem_idx: Optional[torch.Tensor] = None,
                step_idx: Optional[torch.Tensor] = None, sem_features: Optional[torch.Tensor] = None) -> torch.Tensor:
        """eps = decoder(x_t [B,T,n_mels], t [B], sem_idx [B,S] | sem_features [B,S,semantic_dim], step_idx [B] | None)."""
        if sem_idx is None and sem_features is None:
            raise ValueError("Either sem_idx or sem_features must be provided")
        B, T, _ = x_t.shape
        S = sem_features.shape[1] if sem_features is not None else sem_idx.shape[1]
        packed = self._ensure_packed()
        ws = self.workspace(B, T, S, B, x_t.device)
        return native.decoder_forward(self.dims(), packed, ws, x_t.contiguous(), t.contiguous(),
                                      None if step_idx is None else step_idx.contiguous(),
                                      None if sem_features is not None or sem_idx is None else sem_idx.contiguous(),
                                      None if sem_features is None else sem_features.contiguous(), S)
