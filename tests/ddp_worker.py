"""Worker of tests/test_gpu_00_two_ranks.py::test_two_ranks_average_like_one_rank_on_the_whole_batch: one rank of a
data-parallel step on a tiny decoder; rank 0 writes the averaged gradient to the file named on the command line."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "symbolic-music-generation_amd"))


def main(out_path):
    from mgen import MidiLM
    from mgen.config import LMConfig
    from mgen.trainer import GradReducer, init_distributed_from_env
    rank, local, world = init_distributed_from_env()
    cfg = LMConfig(vocab_size=1000, hidden_size=256, intermediate_size=512, num_hidden_layers=3,
                   num_attention_heads=2, num_key_value_heads=2, head_dim=128)
    m = MidiLM(cfg, device=torch.device("cuda", local), seed=0)
    red = GradReducer(m, min_bucket_elems=300_000)      # small buckets: several collectives per step
    # ids below 400 of 1000: the embedding bucket is cut down to the rows some rank can have touched
    ids = torch.randint(3, 400, (world, 256), generator=torch.Generator().manual_seed(4))
    mine = ids[rank:rank + 1].cuda()
    for _ in range(2):                                  # two steps: the buckets are re-used
        m.zero_grad()
        red.bucket_log.clear()
        red.note_token_ids(mine)
        loss = m(input_ids=mine, labels=mine).loss
        loss.backward()
        red.wait()
    if rank == 0:
        torch.save({"grad": m.flat_grad.cpu(), "ids": ids, "buckets": torch.tensor([b for b, _ in red.bucket_log])},
                   out_path)


if __name__ == "__main__":
    main(sys.argv[1])
