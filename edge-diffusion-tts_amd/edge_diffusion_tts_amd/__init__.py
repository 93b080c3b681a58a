"""edge_diffusion_tts_amd -- MI355X-native drop-in for the DDIM sampler path of Krabbens/edge-diffusion-tts.

Same public names as the reference package for this path (edge_diffusion_tts/__init__.py:17-21):
CFG, TrainPhase, get_device, set_seed, DiffusionSchedule, EdgeDiffusionDecoder, EdgeInference, plus the exported
DepthwiseSeparableConv layer.  Everything numerical runs in libedtts_hip.so (hand-written gfx950 kernels behind the
C ABI of include/edtts.h); there is no CPU fallback.
"""
__version__ = "0.1.0"

from .config import CFG, TrainPhase, get_device, set_seed
from .schedule import DiffusionSchedule, DPMSolverPP
from .decoder import EdgeDiffusionDecoder
from .inference import EdgeInference
from .conv import DepthwiseSeparableConv
from .synth import synth_state_dict
from .longform import InpaintSampler
from .melpost import GriffinLim, InverseMelScale, MelVocoder, denormalize_mel, normalize_mel

__all__ = [
    "CFG", "TrainPhase", "get_device", "set_seed", "DiffusionSchedule", "DPMSolverPP", "EdgeDiffusionDecoder", "EdgeInference",
    "DepthwiseSeparableConv", "synth_state_dict", "InpaintSampler",
    "GriffinLim", "InverseMelScale", "MelVocoder", "denormalize_mel", "normalize_mel",
]
