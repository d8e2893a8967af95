import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "tools", "attn_ab.py")).read().replace("\nmain()\n", "\n")
ns = {"__file__": os.path.join(ROOT, "tools", "attn_ab.py"), "__name__": "ab"}
exec(compile(src, "attn_ab", "exec"), ns)
import torch
lib = ns["load"](sys.argv[1])
B, S, H, D = 32, 2048, 8, 128
t = ns["make"](B, S, H, H, D, seed=1)
r = ns["Run"](lib, B, S, H, H, D, t)
r.fwd(); r.fwd(); torch.cuda.synchronize()
st = r.lse.view(torch.int32).view(B * H, S // 256, 4, 64)[..., :10].cpu().long() & 0xFFFFFFFF   # [bh, qblock, wave, field]
names = ["start", "pro_wait", "pro_comp", "loop", "epi", "J", "jl", "hwid", "xcc", "start_hi"]
for qb in range(8):
    x = st[:, qb]
    print("qblock", qb, "J", int(x[0, 0, 5]), {names[i]: [int(x[:, w, i].median()) for w in range(4)] for i in (1, 2, 3, 4)})
# per CU timeline from wave 0 of every workgroup
w0 = st[:, :, 0].reshape(-1, 10)
cu = collections.defaultdict(list)
for row in w0.tolist():
    start = row[0] | (row[9] << 32)
    dur = row[1] + row[2] + row[3] + row[4]
    hw = row[7]
    key = (row[8], (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
    cu[key].append((start, dur, row[5]))
gaps, spans, busy = [], [], []
for k, v in cu.items():
    v.sort()
    spans.append(v[-1][0] + v[-1][1] - v[0][0])
    busy.append(sum(d for _, d, _ in v))
    for (s0, d0, _), (s1, _, _) in zip(v, v[1:]):
        gaps.append(s1 - s0 - d0)
gaps.sort(); spans.sort()
print("CUs seen", len(cu), "WGs per CU", sorted(collections.Counter(len(v) for v in cu.values()).items()))
print("gap between consecutive WGs on a CU: median", gaps[len(gaps) // 2], "p10", gaps[len(gaps) // 10], "p90", gaps[9 * len(gaps) // 10])
print("per-CU span: min", spans[0], "median", spans[len(spans) // 2], "max", spans[-1], " busy median", sorted(busy)[len(busy) // 2])
one = sorted(cu.items())[0]
print("one CU", one[0], [(s - one[1][0][0], d, j) for s, d, j in one[1]])
