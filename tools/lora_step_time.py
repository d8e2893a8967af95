import sys, time, torch
sys.path.insert(0, "symbolic-music-generation_amd")
"""Adapter-training step time: python tools/lora_step_time.py [dropout] [default|aria] [batch] [seq]"""
from mgen import MidiLM, default_decoder, aria_style_llama
from mgen.trainer import Trainer
kind = sys.argv[2] if len(sys.argv) > 2 else "default"
Bsz, S = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (32, 2048)
m = MidiLM(aria_style_llama() if kind == "aria" else default_decoder(), device="cuda:0", seed=0)
m.to_lora(lora_dropout=float(sys.argv[1]) if len(sys.argv) > 1 else 0.0)
tr = Trainer(m, lr=1e-4, num_epochs=1, steps_per_epoch=100, warmup=1)
ids = torch.randint(3, 17000, (Bsz, S)).cuda()
b = {"input_ids": ids, "labels": ids}
for _ in range(3): tr.training_step(b)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(6): l = tr.training_step(b)
torch.cuda.synchronize(); print(kind, Bsz, S, "lora step ms", (time.perf_counter() - t0) / 6 * 1e3, "loss", float(l))
