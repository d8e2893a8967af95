"""Hyper-parameters of the decoder-only MIDI language model.

Mirrors the Qwen3Config fields that the reference's default decoder actually reads
(/root/reference/src/model/model.py:126-136 overrides hidden/layers/heads/kv-heads/intermediate/max-positions and the
special token ids on top of the Qwen/Qwen3-0.6B base config; vocab_size 151936, head_dim 128, tied embeddings,
rope_theta 1e6 and rms_norm_eps 1e-6 are inherited from that base config)."""
from dataclasses import dataclass, asdict


@dataclass
class LMConfig:
    vocab_size: int = 151936
    hidden_size: int = 512
    intermediate_size: int = 2048
    num_hidden_layers: int = 14
    num_attention_heads: int = 8
    num_key_value_heads: int = 8
    head_dim: int = 128
    max_position_embeddings: int = 4096
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    tie_word_embeddings: bool = True
    qk_norm: bool = True           # Qwen3 blocks normalise q and k per head; Llama blocks do not
    rope_interleaved: bool = False  # rotation pairs (2i, 2i+1) instead of (i, i + head_dim/2) (token models)
    # "rms": RMSNorm blocks (Qwen3 / Llama). "layer": LayerNorm with gain and bias (eps = layer_norm_eps) in the same
    # pre-norm block - the other half of the Aria block variant (LayerNorm + adjacent-pair RoPE + fused QKV + SwiGLU;
    # module names in /root/reference/src/model/model.py:239,263-269)
    norm_kind: str = "rms"
    bos_token_id: int = 1
    eos_token_id: int = 2
    pad_token_id: int = 0
    # rows of logits alive at once in the fused lm_head + cross-entropy (32768 x 151936 bf16 = 10 GB; large chunks
    # give the dgrad GEMM, whose output is only [chunk, d], enough tiles to fill 256 CUs)
    lm_head_chunk_tokens: int = 32768
    # 4-D position model (/root/reference/src/model/modeling.py): notes enter as continuous [start, duration, pitch,
    # velocity] rows through an adjacent-pair rotary embedding of q, k AND v; no token embedding; a 16-wide head
    # predicting 8 unit angle pairs trained with a cosine loss
    pos4d: bool = False
    # "llama": RMSNorm + RoPE + SwiGLU blocks (Qwen3 / Llama); "gpt2": LayerNorm + learned positions + gelu_new blocks
    # with biased Conv1D layers (mgen.gpt2.MidiGPT2LM)
    arch: str = "llama"
    layer_norm_eps: float = 1e-5
    # GPT-2 dropout probabilities (training mode only): summed embeddings, both residual projections, attention
    # probabilities. transformers' GPT2Config defaults are 0.1 each and the reference does not override them.
    embd_pdrop: float = 0.0
    resid_pdrop: float = 0.0
    attn_pdrop: float = 0.0
    rope4d_theta: float = 10000.0     # fixed in the reference (:510), independent of rope_theta
    rope4d_nope: int = 4              # zero frequencies ("NoPE" pairs) at the end of each axis' block (:522-525)

    def to_dict(self):
        return asdict(self)

    @property
    def qkv_width(self):
        return (self.num_attention_heads + 2 * self.num_key_value_heads) * self.head_dim


def default_decoder() -> LMConfig:
    """The reference's default decoder (MidiQwen, src/model/model.py:120-139)."""
    return LMConfig()


def pos4d_decoder() -> LMConfig:
    """The reference's 4-D position model (MidiQwenNew, src/model/model.py:273-290): the default decoder's shape
    without a vocabulary."""
    return LMConfig(vocab_size=1, tie_word_embeddings=False, pos4d=True)


def gpt2_decoder(vocab_size: int = 30000, n_positions: int = 4096, pdrop: float = 0.1) -> LMConfig:
    """The reference's GPT-2 variant (MidiGPT2, src/model/model.py:54-64): n_embd 512, 6 layers, 8 heads; the
    vocabulary is the trained tokenizer's (30 000), n_positions = CONTEXT_SIZE = 4096 (src/utils.py:4); dropout 0.1
    everywhere (the GPT2Config defaults, transformers models/gpt2/configuration_gpt2.py:85-87)."""
    return LMConfig(arch="gpt2", vocab_size=vocab_size, hidden_size=512, intermediate_size=2048, num_hidden_layers=6,
                    num_attention_heads=8, num_key_value_heads=8, head_dim=64, max_position_embeddings=n_positions,
                    tie_word_embeddings=True, qk_norm=False, embd_pdrop=pdrop, resid_pdrop=pdrop, attn_pdrop=pdrop)


def aria_style_llama(aria_block: bool = False) -> LMConfig:
    """~650M Llama-architecture stand-in for the remote-code Aria model used by src/train_aria.py (the benchmark
    configurations' model: RMSNorm, split-half RoPE). aria_block=True: the block as the checkpoint's own code is
    recalled to have it - LayerNorm with gain and bias, adjacent-pair RoPE - unverifiable offline either way."""
    cfg = LMConfig(vocab_size=17727, hidden_size=1536, intermediate_size=6144, num_hidden_layers=16,
                   num_attention_heads=24, num_key_value_heads=24, head_dim=64, max_position_embeddings=8192,
                   rms_norm_eps=1e-6, rope_theta=10000.0, tie_word_embeddings=False, qk_norm=False)
    if aria_block:
        cfg.norm_kind, cfg.rope_interleaved = "layer", True
    return cfg
