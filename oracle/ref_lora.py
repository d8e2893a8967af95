"""CPU restatement of the low-rank adapter arithmetic. TEST INFRASTRUCTURE ONLY: nothing in the product path may import
this file.

PARITY UNPINNED: the reference gets its adapters from the third-party package `peft` (`get_peft_model` with
`LoraConfig(r=32, lora_alpha=64, lora_dropout=0.05, bias="none")`, /root/reference/src/model/model.py:251-271), which is
not installed here and is un-pinned in the reference. peft's Linear computes
    y = x W^T + (alpha / r) * (dropout(x) A^T) B^T,    A: [r, in] ~ U(-1/sqrt(in), 1/sqrt(in)),  B: [out, r] = 0,
with every base weight frozen. Without dropout that is a plain linear layer with the merged weight W + (alpha/r) B A,
which gives the two identities the GPU tests check against the (pinned) plain-model path:
    loss(adapters) == loss(plain model with merged weights)
    dA = (alpha/r) B^T dW',   dB = (alpha/r) dW' A^T      (dW' = weight gradient of the merged plain model)
"""
import torch


def merged_weight(W, A, B, scale):
    return W.double() + scale * (B.double() @ A.double())


def adapter_grads_from_merged(dW, A, B, scale):
    dW = dW.double()
    return scale * (B.double().t() @ dW), scale * (dW @ A.double().t())


def lora_linear(x, W, A, B, scale):
    return x @ W.t() + scale * ((x @ A.t()) @ B.t())
