"""mgen: MI355X-native training step and sampler for decoder-only MIDI language models.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed) over libmgen_hip.so, a C-ABI library of
hand-written gfx950 HIP kernels (include/mgen.h)."""
from .config import LMConfig, default_decoder, aria_style_llama, pos4d_decoder, gpt2_decoder  # noqa: F401
from .model import MidiLM, CausalLMOutput  # noqa: F401
from .gpt2 import MidiGPT2LM  # noqa: F401
from .optim import FusedAdamW, get_optim  # noqa: F401
from .collate import DataCollator, MidiDataset4D  # noqa: F401
