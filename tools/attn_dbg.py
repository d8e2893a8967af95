import sys, os
sys.argv = [sys.argv[0]] + sys.argv[1:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib.util, torch
spec = importlib.util.spec_from_file_location("ab", os.path.join(ROOT, "tools", "attn_ab.py"))
src = open(os.path.join(ROOT, "tools", "attn_ab.py")).read().replace("\nmain()\n", "\n")
ns = {"__file__": os.path.join(ROOT, "tools", "attn_ab.py"), "__name__": "ab"}
exec(compile(src, "attn_ab", "exec"), ns)
base, cand = ns["load"](sys.argv[1]), ns["load"](sys.argv[2])
B, S, H, KVH, D = (int(x) for x in sys.argv[3:8])
t = ns["make"](B, S, H, KVH, D, seed=S)
a, b = ns["Run"](base, B, S, H, KVH, D, t), ns["Run"](cand, B, S, H, KVH, D, t)
a.fwd(); b.fwd(); torch.cuda.synchronize()
el = (a.lse - b.lse).abs()          # [B,H,S]
eo = (a.o.float() - b.o.float()).abs().view(B, S, H, D).amax(-1).permute(0, 2, 1)   # [B,H,S]
for name, e in (("lse", el), ("o", eo)):
    blocks = e[0, 0].view(-1, 32).amax(-1) if S % 32 == 0 else e[0, 0]
    print(name, "per 32-row block, b0 h0:", [float("%.3g" % x) for x in blocks.tolist()])
print("lse base", a.lse[0, 0, :8].tolist()); print("lse cand", b.lse[0, 0, :8].tolist())
if S <= 64:
    q, k = t[0].float().view(S, H, D)[:, 0], t[1].float().view(S, KVH, D)[:, 0]
    sc = (q @ k.t()) * D ** -0.5
    print("s[0,:4]", sc[0, :4].tolist(), "s[1,:4]", sc[1, :4].tolist(), "s[:4,0]", sc[:4, 0].tolist())
    print("lse base", [round(x, 3) for x in a.lse[0, 0].tolist()])
    print("lse cand", [round(x, 3) for x in b.lse[0, 0].tolist()])
if S == 128:
    print("base 96..", [round(x, 3) for x in a.lse[0, 0, 96:].tolist()])
    print("cand 96..", [round(x, 3) for x in b.lse[0, 0, 96:].tolist()])
    q, k = t[0].float().view(S, H, D)[:, 0], t[1].float().view(S, KVH, D)[:, 0]
    sc = (q @ k.t()) * D ** -0.5
    m = torch.tril(torch.ones(S, S, dtype=torch.bool, device=sc.device))
    for lo, hi in ((0, 64), (0, 96), (64, 128), (64, 96), (96, 128)):
        mm = m.clone(); mm[:, :lo] = False; mm[:, hi:] = False
        print("lse keys", lo, hi, [round(x, 3) for x in torch.logsumexp(sc.masked_fill(~mm, -1e30), -1)[96:104].tolist()])
